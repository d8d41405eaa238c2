"""Experiment: explicit-inverse BiCGstab on pi with ONE iteration + the Jacobi continuation (solver_xinv_its=1) over a long run: iterations the solve reports,
sampled every 50 steps, for both physics sets.  Usage: python tools/xinv_k1_check.py [K]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fesom2_amd import workloads
from fesom2_amd.core import OceanCore

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for physics in ("default", "pp"):
    wl = workloads.pi(physics)
    mesh = wl.load_mesh()
    gpu = OceanCore(mesh, wl.params(solver_xinv_its=K))
    st, aux, forcing = wl.initial_state(mesh)
    gpu.upload_state(st)
    if forcing:
        gpu.set_forcing(**forcing)
    its, n = [], 1
    for blk in range(60):
        gpu.run_steps(n, 49); n += 49
        gpu.run_steps(n, 1); n += 1
        its.append(gpu.solver_iterations)
    gpu.sync()
    t0 = time.perf_counter(); gpu.run_steps(n, 500); gpu.sync()
    ms = (time.perf_counter() - t0) / 500 * 1e3
    print(json.dumps(dict(physics=physics, K=K, steps=n, iterations=its, ms_per_step=round(ms, 4), eta_absmax=float(np.abs(gpu.get("eta_n", mesh.nod2D)).max()))), flush=True)
    gpu.close()
