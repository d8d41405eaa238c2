#!/usr/bin/env python3
"""GPU probe of the large-mesh SSH solve: RAS-Chebyshev (solver_precond=1) against Jacobi (0) on the channel workload.
usage: ras_probe.py [levels=3] [steps=30]"""
import os, sys, time, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from fesom2_amd import workloads
from fesom2_amd.core import OceanCore

lev = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
wl = workloads.channel(lev)
mesh = wl.load_mesh()
out = {}
configs = [(1, {}), (0, {})]
if "--sweep" in sys.argv:
    configs = [(1, dict(FESOM_GPU_RAS_PATCH=str(pm), FESOM_GPU_RAS_DEG=str(dg), FESOM_GPU_RAS_OVL=str(ov), FESOM_GPU_RAS_KAPPA=str(kp)))
               for pm, dg, ov, kp in ((768, 16, 4, 200), (768, 12, 4, 150), (768, 20, 4, 300), (768, 24, 4, 400), (768, 16, 3, 200), (384, 16, 4, 200), (384, 12, 3, 150), (512, 16, 4, 200), (768, 8, 3, 80))]
for pc, env in configs:
    for k in [k for k in os.environ if k.startswith("FESOM_GPU_RAS_")]:
        del os.environ[k]
    os.environ.update(env)
    gpu = OceanCore(mesh, wl.params(solver_precond=pc))
    wl.start(gpu, mesh)
    its = []
    gpu.run_steps(1, 5); gpu.sync()
    t0 = time.time()
    for n in range(6, 6 + steps):
        gpu.run_steps(n, 1)
        its.append(gpu.solver_iterations)
    gpu.sync()
    dt = (time.time() - t0) / steps
    gpu.call("solver_snapshot")
    t_solve = gpu.kernel_time_ms("k_solver_replay", 10)
    rec = dict(env=env, kind=gpu.lib.fesom_gpu_solver_kind(), ms_per_step=dt * 1e3, solver_ms=t_solve, iterations=its[-10:], resid=gpu.solver_residual)
    if pc:
        for k in ("ras_apply0", "ras_apply1", "ras_spmv1", "ras_spmv2", "dsr_update"):
            gpu.call("ras_arm")
            try:
                rec[k + "_us"] = gpu.kernel_time_ms(k, 50) * 1e3
            except RuntimeError as e:
                rec[k + "_us"] = str(e)
    out[pc] = rec
    e = gpu.get("eta_n", mesh.nod2D)
    rec["eta_minmax"] = [float(e.min()), float(e.max())]
    print(json.dumps(rec), flush=True)
    gpu.close()
