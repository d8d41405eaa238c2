#!/usr/bin/env python3
"""Per-kernel HIP-event times of selected kernels on a workload.  usage: kernel_times.py LEVELS k1,k2,... [nrep]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from fesom2_amd import workloads
from fesom2_amd.core import OceanCore
import bench
L = int(sys.argv[1]); ks = sys.argv[2].split(","); nrep = int(sys.argv[3]) if len(sys.argv) > 3 else 10
wl = workloads.channel(L) if L >= 0 else workloads.pi("default")
mesh = wl.load_mesh()
core = OceanCore(mesh, wl.params())
wl.start(core, mesh)
core.run_steps(1, 5); core.lib.fesom_gpu_sync()
N3, E3, D3 = mesh.wet_counts()
out = []
for k in ks:
    a, b, c = bench.KERNEL_VALUES.get(k, (0, 0, 0))
    name = {"k_thick": "update_thickness_ale"}.get(k, k)
    t = core.kernel_time_ms(name + (":all" if k in bench.PER_TRACER else ""), nrep) * 1e-3
    by = 8.0 * (a * N3 + b * E3 + c * D3) * (2 if k in bench.PER_TRACER else 1)
    out.append(f"{k} {t*1e6:.1f} us {by/t/1e9 if t else 0:.0f} GB/s")
print("TILE", os.environ.get("FESOM_GPU_TILE", "-"), " | ".join(out), flush=True)
core.close()
