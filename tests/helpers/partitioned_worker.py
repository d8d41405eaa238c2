"""Worker of tests/test_gpu_partitioned.py (one process per rank, torch.distributed "gloo", all ranks may share GPU 0).
Each rank first runs the SINGLE-partition step on the whole pi mesh (phase by phase, same kernel sequence) and records the
global fields after every phase, then runs its part of the partitioned step and compares its OWNED values:
  * step 1 up to the SSH right-hand side: bit for bit (no global reduction involved yet);
  * everything after the SSH solve and the state after NSTEPS steps: to the solver tolerance (the partitioned dot products
    are summed in another order, exactly like the reference run on another number of ranks)."""
import json, os, sys
import numpy as np
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from fesom2_amd.mesh import Mesh
from fesom2_amd.config import make_params
from fesom2_amd.core import OceanCore
from fesom2_amd.synthetic import analytic_ts
from fesom2_amd import parallel

PI = os.path.join(REPO, "tests", "golden", "meshes", "pi")
NSTEPS = int(os.environ.get("PART_NSTEPS", "4"))
# (field, kind, values per item as multiple of nl-1 / nl / 1)
NODE3 = ["Unode", "density_m_rho0", "Unode_rhs", "U_c", "hnode_new", "hnode"]
PROBES = {
    "vel_nodes": [("Unode", "n", "2*n1")],
    "pressure": [("density_m_rho0", "n", "n1"), ("bvfreq", "n", "nl"), ("pgf_x", "e", "n1"), ("pgf_y", "e", "n1")],
    "mixing": [("Kv", "n", "nl"), ("Av", "e", "nl")],
    "vel_rhs": [("UV_rhs", "e", "2*n1"), ("UV_rhsAB", "e", "2*n1")],
    "impl_visc": [("UV_rhs", "e", "2*n1"), ("U_b", "e", "2*n1"), ("U_c", "n", "2*n1")],
    "ssh_rhs": [("ssh_rhs", "n", "1"), ("ssh_values", "z", "1")],
    "solve": [("d_eta", "n", "1")],
    "vert_vel": [("UV", "e", "2*n1"), ("eta_n", "n", "1"), ("hbar", "n", "1"), ("Wvel", "n", "nl"), ("hnode_new", "n", "n1"), ("dhe", "e", "1")],
    "tracers": [("tr_arr", "t", "n1"), ("fct_LO", "n", "n1")],
    "thickness": [("hnode", "n", "n1"), ("helem", "e", "n1"), ("zbar_3d_n", "n", "nl")],
}


def widths(mesh):
    n1 = mesh.nl - 1
    return {"n1": n1, "nl": mesh.nl, "2*n1": 2 * n1, "1": 1}


def grab(core, mesh, label, store, step):
    W = widths(mesh)
    N = mesh.myDim_nod2D + mesh.eDim_nod2D
    E = mesh.myDim_elem2D + mesh.eDim_elem2D
    for f, kind, w in PROBES[label]:
        w = W[w]
        if kind == "n":
            a = core.get(f, N * w).reshape(N, w)
        elif kind == "e":
            a = core.get(f, E * w).reshape(E, w)
        elif kind == "t":
            a = core.get(f, 2 * N * w).reshape(2, N, w)
        else:
            a = core.get(f, mesh.ssh_nza)
        store[(step, label, f)] = a


def main():
    dist.init_process_group(os.environ.get("PART_BACKEND", "gloo"))
    if dist.get_backend() == "nccl":
        import torch
        torch.cuda.set_device(0)
    rank, world = dist.get_rank(), dist.get_world_size()
    opts = os.environ.get("PART_OPTS", "").split(",")
    par = make_params(dt=900.0, mix_scheme="KPP" if "kpp" in opts else "PP", Fer_GM="gm" in opts, Redi="redi" in opts, scaling_Ferreira="gm" in opts or "redi" in opts,
                      visc_option=7 if "visc7" in opts else 6 if "visc6" in opts else 5)
    T, S = analytic_ts(PI)
    # ---- single partition (whole mesh) on this rank
    gm = Mesh.load(PI, dt=900.0)
    st = gm.initial_state(2); st.tr_arr[0], st.tr_arr[1] = T, S; st.tr_arr_old[...] = st.tr_arr
    g = OceanCore(gm, par); g.upload_state(st)
    if "kpp" in opts:
        from fesom2_amd.synthetic import analytic_forcing
        g.set_forcing(**analytic_forcing(gm))
    ref = {}
    first = True
    for n in range(1, NSTEPS + 1):
        parallel.run_step(g, par, lambda kind, names: None, lambda: g.call("solve_ssh"), first, lambda lab, n=n: grab(g, gm, lab, ref, n))
        first = False
    g.close()
    # ---- this rank's partition
    TRANSPORT = os.environ.get("PART_TRANSPORT") or None      # "rccl": the library's built-in transport (real RCCL, or the test double)
    if "ras_off" in opts:
        # rank 1 pretends its block does not qualify for the RAS preconditioner: the library must notice at the first step (global sum of
        # the ranks' flags) and take the Jacobi phases on EVERY rank -- mismatched solver paths would hang or scramble d_eta.  Library-
        # driven steps only; compared with the single-partition run to the solver tolerance.
        os.environ["FESOM_GPU_RAS_OFF_ON_RANK"] = "1"
        pc = parallel.PartitionedCore(PI, par, dt=900.0, transport=TRANSPORT)
        lm = pc.mesh
        ln = lm.myList_nod2D - 1
        st = lm.initial_state(2); st.tr_arr[0], st.tr_arr[1] = T[ln], S[ln]; st.tr_arr_old[...] = st.tr_arr
        pc.core.upload_state(st)
        kinds = [pc.core.lib.fesom_gpu_solver_kind()]
        for n in range(1, NSTEPS + 1):
            pc.step_native(n)
        kinds.append(pc.core.lib.fesom_gpu_solver_kind())
        myN = lm.myDim_nod2D
        N = myN + lm.eDim_nod2D
        eta = pc.core.get("eta_n", N)[:myN]
        tr = pc.core.get("tr_arr", 2 * (lm.nl - 1) * N).reshape(2, N, lm.nl - 1)[:, :myN]
        report = {"rank": rank, "kinds": kinds, "iters": int(pc.solver_iterations),
                  "d_eta": float(np.abs(eta - ref[(NSTEPS, "vert_vel", "eta_n")][ln[:myN], 0]).max()),
                  "d_T": float(np.abs(tr - ref[(NSTEPS, "tracers", "tr_arr")][:, ln[:myN]]).max())}
        pc.close()
        sys.stdout.write("PARTREPORT " + json.dumps(report) + chr(10)); sys.stdout.flush()
        dist.destroy_process_group()
        return
    pc = parallel.PartitionedCore(PI, par, dt=900.0, transport=TRANSPORT)
    lm = pc.mesh
    ln = lm.myList_nod2D - 1
    st = lm.initial_state(2); st.tr_arr[0], st.tr_arr[1] = T[ln], S[ln]; st.tr_arr_old[...] = st.tr_arr
    pc.core.upload_state(st)
    if "kpp" in opts:
        pc.core.set_forcing(**analytic_forcing(lm))
    mine = {}
    for n in range(1, NSTEPS + 1):
        pc.step(n, probe=lambda lab, n=n: grab(pc.core, lm, lab, mine, n))
    myN, myE = lm.myDim_nod2D, lm.myDim_elem2D
    le = lm.myList_elem2D - 1
    report = {"rank": rank, "bitwise_fail": [], "maxdiff": {}, "iters": pc.solver_iterations}
    pre_solver = ("vel_nodes", "pressure", "mixing", "vel_rhs", "impl_visc", "ssh_rhs")
    for (step, label, f), a in mine.items():
        r = ref[(step, label, f)]
        if f == "ssh_values":
            continue
        if a.ndim == 3:
            a, r = a[:, :myN], r[:, ln[:myN]]
        elif a.shape[0] == myN + lm.eDim_nod2D:
            a, r = a[:myN], r[ln[:myN]]
        else:
            a, r = a[:myE], r[le[:myE]]
        d = float(np.abs(a - r).max()) if a.size else 0.0
        if step == 1 and label in pre_solver:
            eq = (a.view(np.int64) == r.view(np.int64)) | ((a == 0) & (r == 0))
            if not eq.all():
                report["bitwise_fail"].append(f"{label}:{f} {int((~eq).sum())}/{eq.size} max {d:.3e}")
        key = f"{label}:{f}"
        report["maxdiff"][key] = max(report["maxdiff"].get(key, 0.0), d)
    # halo consistency at the end: halo values of T equal the owners' values
    Tfin = mine[(NSTEPS, "tracers", "tr_arr")]
    report["halo_T_maxdiff"] = float(np.abs(Tfin[:, myN:] - ref[(NSTEPS, "tracers", "tr_arr")][:, ln[myN:]]).max()) if lm.eDim_nod2D else 0.0
    # ---- the same steps driven by the library (fesom_gpu_step_partitioned + transport callbacks): bit-identical to the
    # phase-by-phase Python loop above (same kernels, same order, same transport)
    fin = {f: pc.core.get(f, cnt).copy() for f, cnt in (("tr_arr", 2 * (lm.nl - 1) * (myN + lm.eDim_nod2D)), ("eta_n", myN + lm.eDim_nod2D),
                                                           ("UV", 2 * (lm.nl - 1) * (myE + lm.eDim_elem2D)))}
    its_py = pc.solver_iterations
    pc.close()
    pc = parallel.PartitionedCore(PI, par, dt=900.0, transport=TRANSPORT)
    pc.core.upload_state(st)
    if "kpp" in opts:
        pc.core.set_forcing(**analytic_forcing(lm))
    for n in range(1, NSTEPS + 1):
        pc.step_native(n)
    report["native_mismatch"] = [f for f, a in fin.items() if not np.array_equal(a.view(np.int64), pc.core.get(f, a.size).view(np.int64))]
    report["native_iters"] = [int(its_py), int(pc.solver_iterations)]
    report["transport"] = pc.transport_name
    report["comm_stats"] = list(pc.comm_stats())
    pc.close()
    sys.stdout.write("PARTREPORT " + json.dumps(report) + chr(10)); sys.stdout.flush()
    dist.destroy_process_group()


main()
