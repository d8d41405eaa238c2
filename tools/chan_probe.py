#!/usr/bin/env python3
"""Probe a workload on the GPU: ms/step, solver iterations, device-side step monitor, per-kernel HIP-event times.
usage: chan_probe.py [--workload channel|basin|pi] [--levels L] [--steps K] [--warmup W] [--kernels] [--monitor-every M]"""
import argparse, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from fesom2_amd import workloads
from fesom2_amd.core import OceanCore

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="channel")
ap.add_argument("--levels", type=int, default=3)
ap.add_argument("--layers", type=int, default=47)
ap.add_argument("--physics", default="default")
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--warmup", type=int, default=20)
ap.add_argument("--kernels", action="store_true")
ap.add_argument("--monitor-every", type=int, default=0)
a = ap.parse_args()
t0 = time.time()
wl = workloads.channel(a.levels, a.layers) if a.workload == "channel" else workloads.basin(a.levels, a.layers) if a.workload == "basin" else workloads.pi(a.physics, a.levels)
mesh = wl.load_mesh()
N3, E3, D3 = mesh.wet_counts()
print("mesh", mesh.nod2D, mesh.elem2D, mesh.edge2D, "nl", mesh.nl, "wet", N3, E3, D3, "setup s", round(time.time() - t0, 1), flush=True)
core = OceanCore(mesh, wl.params())
wl.start(core, mesh)
core.run_steps(1, a.warmup); core.lib.fesom_gpu_sync()
print("warmup done, its", core.solver_iterations, flush=True)
n = 1 + a.warmup
if a.monitor_every:
    done = 0
    while done < a.steps:
        k = min(a.monitor_every, a.steps - done)
        t1 = time.perf_counter(); core.run_steps(n, k); core.lib.fesom_gpu_sync(); el = time.perf_counter() - t1
        n += k; done += k
        si = core.step_info()
        print(n - 1, f"ms/step {el / k * 1e3:.3f}", "its", core.solver_iterations, "eta", f"{si['min_eta']:.3e} {si['max_eta']:.3e}", "T", f"{si['min_temp']:.2f} {si['max_temp']:.2f}",
              "u", f"{si['min_uvel']:.2e} {si['max_uvel']:.2e}", "cfl_z", f"{si['max_cfl_z']:.3f}", "blowup", si["blowup"], flush=True)
else:
    t1 = time.perf_counter(); core.run_steps(n, a.steps); core.lib.fesom_gpu_sync(); el = time.perf_counter() - t1
    print(f"ms/step {el / a.steps * 1e3:.4f}  its {core.solver_iterations}", flush=True)
if a.kernels:
    sys.path.insert(0, REPO)
    import bench
    res = bench.kernel_table(core, mesh, wl)
    print(json.dumps(res, indent=1))
core.close()
