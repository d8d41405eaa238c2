"""Experiment: drop threshold of the sparsified explicit inverse x BiCGstab iterations on pi (default physics).
For every (drop, K): entries per row, iterations the solve reports, ms per step over 300 steps.  Usage: python tools/xinv_sweep.py"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def run(drop, K, steps=300):
    os.environ["FESOM_GPU_XINV_DROP"] = drop
    from fesom2_amd import workloads
    from fesom2_amd.core import OceanCore
    wl = workloads.pi("default")
    mesh = wl.load_mesh()
    gpu = OceanCore(mesh, wl.params(solver_xinv_its=K))
    st, aux, forcing = wl.initial_state(mesh)
    gpu.upload_state(st)
    if forcing:
        gpu.set_forcing(**forcing)
    gpu.run_steps(1, 40); gpu.sync()
    its = []
    for n in range(20):
        gpu.run_steps(41 + n, 1); its.append(gpu.solver_iterations)
    gpu.sync()
    t0 = time.perf_counter()
    gpu.run_steps(61, steps); gpu.sync()
    ms = (time.perf_counter() - t0) / steps * 1e3
    res = gpu.solver_residual if hasattr(gpu, "solver_residual") else None
    eta = gpu.get("eta_n", mesh.nod2D)
    gpu.close()
    return dict(drop=drop, K=K, ms_per_step=round(ms, 4), iterations=its, residual=res, eta_absmax=float(np.abs(eta).max()))


if __name__ == "__main__":
    if len(sys.argv) == 3:                       # one configuration per process (the library keeps the built inverse for the life of the process)
        print(json.dumps(run(sys.argv[1], int(sys.argv[2]))), flush=True)
    else:
        import subprocess
        for drop in ("1e-4", "1e-5", "1e-6", "1e-7"):
            for K in (1, 2):
                subprocess.run([sys.executable, os.path.abspath(__file__), drop, str(K)], check=False)
