"""Worker of tests/test_ice.py::test_gpu_partitioned_evp_equals_reference (one process per rank, gloo rendezvous, ranks share GPU 0):
EVPdynamics_m on the reference's dist_2 partition of pi, halo of (u_ice_aux, v_ice_aux) exchanged after every subcycle, against the
rank-local outputs of the REFERENCE's 2-rank run of its own routine (tests/golden/ice_evp_reference.npz: r2/<rank>/...), bit for bit."""
import ctypes as C, json, os, sys
import numpy as np
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from fesom2_amd import parallel, ice, _lib
from fesom2_amd.config import make_params

PI = os.path.join(REPO, "tests", "golden", "meshes", "pi")
STATE = ("u_ice", "v_ice", "a_ice", "m_ice", "m_snow", "elevation", "u_w", "v_w", "stress_atmice_x", "stress_atmice_y", "sigma11", "sigma12", "sigma22")


def bits(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64).ravel(); b = np.ascontiguousarray(b, dtype=np.float64).ravel()
    return bool(((a.view(np.int64) == b.view(np.int64)) | ((a == 0) & (b == 0))).all())


def main():
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    adv = os.environ.get("ICE_ADV") == "1"                    # + the FCT advection after the EVP call (tests/golden/ice_adv_reference.npz)
    aevp = os.environ.get("ICE_AEVP") == "1"                  # the adaptive EVP, EVPdynamics_a (tests/golden/ice_aevp_reference.npz)
    evp0 = os.environ.get("ICE_EVP0") == "1"                  # the classic EVP, EVPdynamics (tests/golden/ice_evp0_reference.npz)
    g = np.load(os.path.join(REPO, "tests", "golden", "ice_evp0_reference.npz" if evp0 else ("ice_aevp_reference.npz" if aevp else ("ice_adv_reference.npz" if adv else "ice_evp_reference.npz"))))
    transport = os.environ.get("PART_TRANSPORT") or None
    pc = parallel.PartitionedCore(PI, make_params(dt=900.0), dt=900.0, transport=transport)     # the partition's transport (ocean context = same com lists)
    mesh = pc.mesh
    pv = g["in/ice_params"]
    par = ice.ice_params(ice_dt=pv[0], ellipse=pv[1], alpha_evp=pv[2], beta_evp=pv[3], Pstar=pv[4], c_pressure=pv[5], delta_min=pv[6], cd_oce_ice=pv[7],
                         evp_rheol_steps=int(pv[8]), max_ice_loading=pv[9], **(dict(whichEVP=2, c_aevp=pv[10]) if aevp else (dict(whichEVP=0, theta_io=pv[11], Tevp_inv=pv[12]) if evp0 else {})))
    myE, N = mesh.myDim_elem2D, mesh.myDim_nod2D + mesh.eDim_nod2D
    names = STATE + (("alpha_evp_array", "beta_evp_array") if aevp else ())
    fields = ice.IceFields(**{k: (g[f"r2/{rank}/in/{k}"][:myE] if (k.startswith("sigma") or k.startswith("alpha")) else g[f"r2/{rank}/in/{k}"]) for k in names})
    core = ice.IceCore(mesh, par)
    core.upload(fields)
    core.lib.fesom_gpu_ice_evp_partitioned.argtypes = [C.c_int, C.c_void_p]
    tr = None if (pc.transport == "rccl") else C.byref(pc._get_transport())
    core._chk(core.lib.fesom_gpu_ice_evp_partitioned(1, tr), "ice_evp_partitioned")
    if adv:
        core.lib.fesom_gpu_ice_advect_partitioned.argtypes = [C.c_int, C.c_void_p]
        core._chk(core.lib.fesom_gpu_ice_advect_partitioned(1, tr), "ice_advect_partitioned")
    core.download(fields)
    rep = {"rank": rank, "transport": pc.transport_name, "bad": []}
    myN = mesh.myDim_nod2D
    for k in (("u_ice", "v_ice", "a_ice", "m_ice", "m_snow") if adv else (("u_ice", "v_ice", "sigma11", "sigma12", "sigma22") + (("alpha_evp_array", "beta_evp_array") if aevp else ()))):
        ref = g[f"r2/{rank}/out1/{k}"]
        ref = ref[:myE] if (k.startswith("sigma") or k.startswith("alpha")) else ref[:N]
        if k == "beta_evp_array":                             # (the reference forms it for the owned nodes and never exchanges it)
            if not bits(fields[k][:myN], ref[:myN]):
                rep["bad"].append(f"{k}: max |d| {float(np.abs(fields[k][:myN] - ref[:myN]).max()):.3e}")
            continue
        if not bits(fields[k], ref):
            rep["bad"].append(f"{k}: max |d| {float(np.abs(fields[k] - ref).max()):.3e}")
    rep["changed"] = float(np.abs(fields["m_ice"] - g[f"r2/{rank}/in/m_ice"]).max()) * 1e2 if adv else float(np.abs(fields["u_ice"] - g[f"r2/{rank}/in/u_ice"]).max())
    core.close(); pc.close()
    sys.stdout.write("ICEREPORT " + json.dumps(rep) + chr(10)); sys.stdout.flush()
    dist.destroy_process_group()


main()
