import os, sys, time
sys.path.insert(0, os.getcwd())
from fesom2_amd import workloads
from fesom2_amd.core import OceanCore
wl = workloads.pi("default"); mesh = wl.load_mesh()
core = OceanCore(mesh, wl.params()); wl.start(core, mesh)
core.run_steps(1, 50); core.lib.fesom_gpu_sync()
for k in range(6):
    t0 = time.perf_counter(); core.run_steps(51 + k, 1); t1 = time.perf_counter(); core.lib.fesom_gpu_sync(); t2 = time.perf_counter()
    print(f"1 step: host enqueue {(t1-t0)*1e6:.0f} us, until done {(t2-t0)*1e6:.0f} us")
for n in (4, 16):
    t0 = time.perf_counter(); core.run_steps(100, n); t1 = time.perf_counter(); core.lib.fesom_gpu_sync(); t2 = time.perf_counter()
    print(f"{n} steps: host enqueue {(t1-t0)*1e6/n:.0f} us/step, until done {(t2-t0)*1e6/n:.0f} us/step")
core.close()
