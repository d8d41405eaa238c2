#!/usr/bin/env python3
"""bench.py -- SYPD of the MI355X ocean dynamical core on the pi mesh (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one pass of the hot path (compute_vel_nodes + oce_timestep_ale: EOS/PGF, momentum, SSH solve,
ALE vertical velocity, 2x FCT tracer advection + diffusion, thickness update) over the pi mesh
(3140 nodes, 47 layers, T/S, no sea ice), synthetic analytic initial state resident in HBM.
SYPD = 86400 / (365*96 * seconds_per_step)  (pi: step_per_day=96, setups/pi/setup.yml:12).

N>1: one process per GPU under torch.distributed.run.  `value` is the SYPD of ONE pi simulation partitioned over the N GPUs
(reference node partition, halo exchange over RCCL, partitioned SSH solve; fesom2_amd/parallel.py + fesom_gpu_step_partitioned;
scaling "strong").  pi has ~390 surface nodes per GPU at N = 8 and is latency-bound: expect it BELOW the N = 1 value.  The
aggregate of N independent replicas is reported beside it in "replicas" (weak scaling, no communication); it only becomes
`value` if the partitioned run fails (error kept in "partitioned").

--physics pp (default, the workload of this round's profiles) | default (KPP + GM + Redi + surface forcing, the
reference's namelist defaults); at N = 1 a short run of the other set is reported in "other_physics".

Besides the contract fields the JSON line carries
  roofline     : dominant kernel, algorithmic bytes (SURVEY 8d counting rule) / HIP-event time vs 8 TB/s
  cpu_baseline : the reference Fortran/MPI build (oracle/_ref, kind "reference") timed on the host cores,
                 or the scalar C restatement (kind "port") if the reference binary cannot run here.
"""
import argparse
import json
import os
import subprocess
import sys
import time
import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

STEPS_PER_YEAR = 365 * 96
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s measured-achievable

# algorithmic traffic per launch in 8-byte values per wet cell (N3 nodes, E3 prisms, D3 edge cells):
# every distinct 3-D array once per read and once per write, gathers once per gathered value (SURVEY 8d rule)
KERNEL_VALUES = {
    "k_vel_nodes": (2, 2, 0), "k_pressure_bv": (8, 0, 0), "k_pgf": (2, 3, 0), "k_sigma_slope": (13, 0, 0),
    "k_pp": (5, 22, 0), "k_momadv_node": (5, 2, 0),
    "k_vel_rhs": (2, 10, 0), "k_visc_elem": (0, 4, 0), "k_visc_node": (2, 2, 0), "k_impl_visc": (3, 13, 0),
    "k_edge_transport": (0, 5, 0), "k_update_vel": (0, 6, 0), "k_vert_vel": (8, 3, 0),
    "k_tr_ab": (3, 0, 0), "k_tr_z": (3, 0, 0), "k_tr_grad_elem": (2, 4, 0), "k_updn_grad": (0, 2, 4), "k_flux_hor": (2, 3, 6),
    "k_fct_lo_node": (12, 0, 1), "k_fct_node": (10, 0, 1), "k_fct_edge_limit": (2, 0, 2),
    "k_tr_update": (18, 0, 2), "k_diff_flux": (2, 6, 1), "k_thick_node": (5, 0, 0), "k_thick_elem": (1, 1, 0),
}
PER_TRACER = ("k_tr_ab", "k_tr_z", "k_tr_grad_elem", "k_updn_grad", "k_flux_hor", "k_fct_lo_node", "k_fct_node",
              "k_fct_edge_limit", "k_tr_update", "k_diff_flux")


PHYSICS = {   # --physics: options of the hot path; "pp" is the headline workload of this round's profiles, "default" the reference's
    "pp": dict(kw=dict(), ref_cfg="pi_pp", text="PP mixing, no GM/Redi, no surface forcing"),
    "default": dict(kw=dict(mix_scheme="KPP", Fer_GM=True, Redi=True), ref_cfg="pi_default",
                    text="KPP mixing + GM + Redi (namelist.oce defaults), analytic wind/heat/fresh-water forcing"),
}


def cpu_baseline(nsteps_ref=400, physics="pp", refine=0):
    """Reference Fortran/MPI hot path (oracle/_ref/fesom_oracle.x, built from the reference's own sources) on the
    host cores: same mesh, same options, same initial state; 8 MPI ranks (dist_8)."""
    ncpu = os.cpu_count() or 1
    exe = os.path.join(REPO, "oracle", "_ref", "fesom_oracle.x")
    try:
        if not os.path.exists(exe):
            raise RuntimeError("no reference binary")
        from oracle.ref import run_ref
        ranks = 8 if ncpu >= 8 else 2
        cfg = PHYSICS[physics]["ref_cfg"]
        if refine > 0:            # the reference on the same refined mesh: edge files + partition written in its own formats
            cfg, _ = run_ref.refined_case(refine, ranks, base=cfg)
            nsteps_ref = max(20, nsteps_ref // 4 ** refine)
        rd, rc, lines = run_ref.run(cfg, ranks, nsteps_ref, mode="step", dump=(), dump_mesh=False)
        tl = [l for l in lines if l.startswith("ORACLE_TIMING")]
        if rc != 0 or not tl:
            raise RuntimeError(f"reference run failed rc={rc}")
        sps = float(tl[0].split("s_per_step=")[1])
        return {"value": 86400.0 / (STEPS_PER_YEAR * sps), "unit": "simulated_years/day", "cores": ranks, "kind": "reference",
                "sample": f"{nsteps_ref} steps of oce_timestep_ale on pi{' refined ' + str(refine) + 'x' if refine else ''} ({PHYSICS[physics]['text']}), {ranks} MPI ranks, {sps*1e3:.2f} ms/step"}
    except Exception as e:          # reference cannot run here: time the scalar C restatement instead
        from fesom2_amd.mesh import Mesh
        from fesom2_amd.config import make_params
        from fesom2_amd.synthetic import analytic_ts
        from oracle_lib import Oracle
        pi = os.path.join(REPO, "tests", "golden", "meshes", "pi")
        mesh = Mesh.load(pi, dt=900.0)
        orc = Oracle(mesh, make_params(dt=900.0, **PHYSICS[physics]["kw"]))
        if physics == "default":
            from fesom2_amd.synthetic import analytic_forcing
            for k, v in analytic_forcing(mesh).items():
                orc.set(k, v)
        st = mesh.initial_state(2)
        st.tr_arr[0], st.tr_arr[1] = analytic_ts(pi)
        st.tr_arr_old[...] = st.tr_arr
        orc.set_state(st)
        n = 40
        orc.call("step", 1)
        t0 = time.perf_counter()
        for k in range(n):
            orc.call("step", 2 + k)
        sps = (time.perf_counter() - t0) / n
        return {"value": 86400.0 / (STEPS_PER_YEAR * sps), "unit": "simulated_years/day", "cores": 1, "kind": "port",
                "sample": f"{n} steps of the scalar C restatement on pi, {sps*1e3:.2f} ms/step (reference binary unavailable: {e})"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--physics", choices=sorted(PHYSICS), default="pp", help="options of the timed step (default: pp; a short run of the other set is reported in 'other_physics' at N=1)")
    ap.add_argument("--refine", type=int, default=0, help="supplementary workload: pi refined uniformly L times (4^L x the cells, same dt = 900 s); the headline metric is L = 0")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    local_rank %= max(1, torch.cuda.device_count())        # (identity on a node with >= N GPUs; lets N ranks rehearse on one GPU)
    os.environ.setdefault("FESOM_GPU_DEVICE", str(local_rank))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a wedged point-to-point in the partitioned leg must raise (and fall back to the replicas line), not abort the job
        os.environ.setdefault("TORCH_NCCL_BLOCKING_WAIT", "1")
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
        dist.init_process_group(backend=os.environ.get("FESOM_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)   # (gloo: 1-GPU rehearsal)

    import __graft_entry__ as ge
    if rank == 0 and not os.path.exists(os.path.join(REPO, "fesom2_amd", "libfesom_gpu.so")):
        ge.build()
    if world > 1:
        dist.barrier()
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.core import OceanCore
    from fesom2_amd.synthetic import analytic_ts, analytic_forcing

    pi = os.path.join(REPO, "tests", "golden", "meshes", "pi")
    dt = 900.0
    if args.refine > 0:
        import tempfile
        from fesom2_amd import mesh_refine
        pi_r = os.path.join(tempfile.gettempdir(), f"fesom_pi_refined_{args.refine}_{os.getpid()}")
        mesh_refine.refine(pi, pi_r, args.refine)
        pi = pi_r                      # dt stays 900 s: the untuned viscosity of this synthetic set-up is unstable for shorter steps
    steps_per_year = 365 * 86400.0 / dt
    mesh = Mesh.load(pi, dt=dt)
    par = make_params(dt=dt, **PHYSICS[args.physics]["kw"])
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(pi)
    st.tr_arr_old[...] = st.tr_arr

    def new_core(physics):
        c = OceanCore(mesh, make_params(dt=dt, **PHYSICS[physics]["kw"]))
        c.upload_state(st)
        if physics == "default":
            c.set_forcing(**analytic_forcing(mesh))
        return c

    core = new_core(args.physics)

    def barrier():
        if world > 1:
            dist.barrier()

    core.run_steps(1, args.warmup)
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    core.run_steps(1 + args.warmup, args.steps)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    its = core.solver_iterations
    sps = elapsed / args.steps
    sypd_one = 86400.0 / (steps_per_year * sps)

    # ---- N > 1: additionally ONE simulation partitioned over the N GPUs (reference node partition, halo exchange over
    # RCCL, partitioned SSH solve; fesom2_amd/parallel.py).  pi has 3140 surface nodes, i.e. ~390 per GPU at N = 8: the step
    # is launch/latency-bound and every one of its ~110 exchanges costs more than the kernels between them, so this leg is
    # reported next to the replicas line, not instead of it.  Any failure here leaves the replicas line intact.
    partitioned = None
    if world > 1 and os.environ.get("FESOM_BENCH_PARTITIONED", "1") != "0":
        core.close()
        try:
            import datetime
            from fesom2_amd import parallel
            pg = dist.new_group(timeout=datetime.timedelta(seconds=90))
            pc = parallel.PartitionedCore(pi, par, group=pg, dt=dt)
            ln = pc.mesh.myList_nod2D - 1
            T0, S0 = analytic_ts(pi)
            lst = pc.mesh.initial_state(2)
            lst.tr_arr[0], lst.tr_arr[1] = T0[ln], S0[ln]
            lst.tr_arr_old[...] = lst.tr_arr
            pc.core.upload_state(lst)
            if args.physics == "default":
                pc.core.set_forcing(**analytic_forcing(pc.mesh))
            pw, pk = args.warmup, args.steps          # this leg is the headline at N > 1: the contract's W warm-up and K timed steps
            for n in range(1, pw + 1):
                pc.step_native(n)                 # phase + solver loops in the library, torch.distributed only moves the bytes
            torch.cuda.synchronize(); dist.barrier(group=pg)
            tp = time.perf_counter()
            for n in range(pw + 1, pw + pk + 1):
                pc.step_native(n)
            pc.core.lib.fesom_gpu_sync(); torch.cuda.synchronize(); dist.barrier(group=pg)
            pel = torch.tensor([time.perf_counter() - tp], dtype=torch.float64, device="cuda")
            dist.all_reduce(pel, op=dist.ReduceOp.MAX, group=pg)
            psps = float(pel.item()) / pk
            eta_own = pc.owned("eta_n", 1)[1]
            assert np.isfinite(eta_own).all()
            partitioned = {"ms_per_step": round(psps * 1e3, 4), "value": round(86400.0 / (steps_per_year * psps), 2), "unit": "simulated_years/day",
                           "scaling": "strong", "steps": pk, "warmup": pw, "solver_iterations": pc.solver_iterations, "transport": "rccl (torch.distributed nccl)" if dist.get_backend() == "nccl" else "gloo, host-staged",
                           "owned_nodes_per_gpu": int(pc.mesh.myDim_nod2D), "error": None}
            pc.close()
        except Exception as e:          # noqa: BLE001 - keep the replicas line whatever happens in this leg
            partitioned = {"error": f"{type(e).__name__}: {e}"[:300]}
        if rank != 0:
            os._exit(0)                 # no further collective: rank 0 finishes the report alone
        core = new_core(args.physics)
        core.run_steps(1, 60)
        torch.cuda.synchronize()

    if rank == 0:
        eta = core.get("eta_n", mesh.myDim_nod2D)
        T = core.get("tr_arr", 2 * mesh.myDim_nod2D * (mesh.nl - 1))
        assert np.isfinite(eta).all() and np.isfinite(T).all(), "model state blew up"
        # per-kernel device times (HIP events on the library's stream) -> dominant kernel + roofline
        N3, E3, D3 = mesh.wet_counts()
        times, kbytes = {}, {}
        # SSH solve replayed on a real (operator, rhs, warm-start) triple of one more step
        for r in ("compute_vel_nodes", "pressure_bv", "pressure_force", "compute_sigma_xy", "mixing_pp" if args.physics == "pp" else "mixing_kpp", "compute_vel_rhs",
                  "visc_filt_bcksct", "impl_vert_visc_ale", "update_stiff_mat_ale", "compute_ssh_rhs_ale", "solver_snapshot"):
            core.call(r)
        times["k_solver"] = core.kernel_time_ms("k_solver_replay", 10) * 1e-3
        its = core.solver_iterations
        for k, (a, b, c) in KERNEL_VALUES.items():
            if args.physics != "pp" and k == "k_pp":
                continue
            # per-tracer kernels: timed as the step launches them, T and S in one launch (grid.y = 2)
            times[k] = core.kernel_time_ms(k + (":all" if k in PER_TRACER else ""), 50) * 1e-3
            kbytes[k] = 8.0 * (a * N3 + b * E3 + c * D3) * (2 if k in PER_TRACER else 1)
        share = dict(times)
        share["k_edge_transport"] = times["k_edge_transport"] * 2
        dom = max((k for k in share if k != "k_solver"), key=lambda k: share[k])
        achieved = kbytes[dom] / times[dom] / 1e9
        step_bytes = 8.0 * ((77 * N3 + 67 * E3) + 2 * (77 * N3 + 16 * E3 + 16 * D3))       # 0.405 GB on pi
        # HBM-side bytes per launch of that kernel from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 passes,
        # gfx950 correction): they cannot be read from inside the process, so they come from the committed summary of the
        # last profiled run of this same command (tools/pmc_summary.py -> profiles/*pmc_summary.json); null when absent
        traffic = None
        try:
            import glob
            pm = sorted(glob.glob(os.path.join(REPO, "profiles", "*pmc_summary.json")))
            if pm:
                kern = json.load(open(pm[-1]))["kernels"]
                key = dom if dom in kern else dom + ("<false>" if args.physics == "pp" else "<true>")      # templated on Redi
                traffic = kern[key]["traffic_bytes_max"]      # both tracers per launch, like times[dom]
        except Exception:
            traffic = None
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "kernel_us": round(times[dom] * 1e6, 2), "algorithmic_bytes_per_launch": kbytes[dom],
                    "whole_step": {"algorithmic_GB_per_step": round(step_bytes / 1e9, 4),
                                   "achieved_GBs": round(step_bytes / sps / 1e9, 1),
                                   "frac": round(step_bytes / sps / 1e9 / HBM_PEAK_GBS, 4),
                                   "sum_kernel_us": round(sum(share.values()) * 1e6, 1),
                                   "solver_us": round(times["k_solver"] * 1e6, 1), "solver_iterations": its},
                    "top5_us": {k: round(v * 1e6, 2) for k, v in sorted(share.items(), key=lambda kv: -kv[1])[:5]}}
        # the SSH solve is the largest single launch by time, but it is a 2-D problem (N2 rows) solved by ONE workgroup on ONE CU:
        # its traffic is the ELL operator re-read from L2 twice per iteration, bounded by one CU's L2 port, not by HBM
        if mesh.myDim_nod2D <= 4096:
            ell_bytes = 8.0 * 10 * ((mesh.myDim_nod2D + 63) // 64 * 64)
            sol_bytes = its * 2 * ell_bytes
            roofline["solver"] = {"kernel": "k_solver_reg", "us": round(times["k_solver"] * 1e6, 1), "iterations": its,
                                  "operator_bytes_per_spmv": ell_bytes, "algorithmic_bytes": sol_bytes,
                                  "achieved_GBs": round(sol_bytes / times["k_solver"] / 1e9, 1),
                                  "bound": "latency / one CU's L2 port (64 B/clk ~ 134 GB/s): one 1024-thread workgroup, 2 barriers-separated reductions per iteration",
                                  "frac_of_one_cu_l2": round(sol_bytes / times["k_solver"] / 1e9 / 134.0, 3)}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(physics=args.physics, refine=args.refine)
        # the other option set, short run (N = 1 only): same mesh and state, its own CPU reference timing
        other = None
        if world == 1 and args.refine == 0:
            oph = "default" if args.physics == "pp" else "pp"
            core.close()
            core = new_core(oph)
            ow, ok_ = 100, 500
            core.run_steps(1, ow); torch.cuda.synchronize()
            t1 = time.perf_counter()
            core.run_steps(1 + ow, ok_); torch.cuda.synchronize()
            osps = (time.perf_counter() - t1) / ok_
            assert np.isfinite(core.get("eta_n", mesh.myDim_nod2D)).all(), "model state blew up (other physics)"
            other = {"physics": PHYSICS[oph]["text"], "ms_per_step": round(osps * 1e3, 5), "value": round(86400.0 / (steps_per_year * osps), 2),
                     "unit": "simulated_years/day", "steps": ok_, "warmup": ow, "solver_iterations": core.solver_iterations,
                     "cpu_baseline": None if args.no_cpu_baseline else cpu_baseline(200, oph)}
        part_ok = world > 1 and partitioned is not None and partitioned.get("error") is None
        # N > 1: the metric is the SYPD of ONE pi simulation partitioned over the N GPUs (strong scaling).  Only if that leg failed
        # does the line fall back to the aggregate of N independent replicas (weak), with the error recorded in "partitioned".
        out = {"metric": "SYPD (simulated years/day) on pi mesh, 47 z-levels", "value": partitioned["value"] if part_ok else round(sypd_one * world, 2),
               "unit": "simulated_years/day", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": partitioned["ms_per_step"] if part_ok else round(sps * 1e3, 5), "higher_is_better": True,
               "scaling": "strong" if part_ok else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": ("pi mesh (3140 nodes, 5839 elements, 47 layers)" if args.refine == 0 else
                                       f"pi mesh refined {args.refine}x ({mesh.nod2D} nodes, {mesh.elem2D} elements, 47 layers)") +
                                      ", T/S tracers, zstar ALE, JM EOS, MFCT/QR4C/FCT advection, no sea ice, " + PHYSICS[args.physics]["text"],
                          "steps_per_day": int(round(86400.0 / dt)),
                          "parallelism": "single GPU" if world == 1 else (f"one simulation partitioned over {world} GPUs (reference node partition, halo exchange + partitioned SSH solve over torch.distributed/RCCL); {world} independent replicas in 'replicas'" if part_ok else f"{world} independent replicas (the partitioned run failed, see 'partitioned')"),
                          "wet_cells": {"N3": N3, "E3": E3, "D3": D3}},
               "roofline": roofline, "cpu_baseline": cpu}
        if world > 1:
            out["partitioned"] = partitioned
            out["replicas"] = {"value": round(sypd_one * world, 2), "unit": "simulated_years/day", "ms_per_step": round(sps * 1e3, 5), "scaling": "weak",
                               "note": f"aggregate of {world} independent pi simulations, one per GPU (no communication)"}
        if other is not None:
            out["other_physics"] = other
        print(json.dumps(out), flush=True)
    if world > 1 and partitioned is not None:
        sys.stdout.flush()
        os._exit(0)                     # the other ranks left after the partitioned leg
    barrier()
    core.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
