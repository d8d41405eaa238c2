#!/usr/bin/env python3
"""Kernel experiments on a large workload: time selected kernels (HIP events, as bench.kernel_table does) in THIS process; the caller sets the
experiment's environment switches.  usage: kexp.py [--workload channel|basin] [--levels L] --kernels k1,k2:all,... [--reps R] [--tag T]"""
import argparse, json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from fesom2_amd import workloads
from fesom2_amd.core import OceanCore

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="channel")
ap.add_argument("--levels", type=int, default=3)
ap.add_argument("--kernels", required=True)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--tag", default="")
a = ap.parse_args()
wl = workloads.channel(a.levels) if a.workload == "channel" else workloads.basin(a.levels)
mesh = wl.load_mesh()
core = OceanCore(mesh, wl.params())
wl.start(core, mesh)
core.run_steps(1, a.steps); core.lib.fesom_gpu_sync()
p = core.params
for r in ("compute_vel_nodes", "pressure_bv", "pressure_force", "compute_sigma_xy", "mixing_pp" if p.mix_scheme == 2 else "mixing_kpp",
          "compute_vel_rhs", "visc_filt_bcksct", "impl_vert_visc_ale", "update_stiff_mat_ale", "compute_ssh_rhs_ale", "solver_snapshot"):
    core.call(r)
out = {}
for k in a.kernels.split(","):
    out[k] = round(core.kernel_time_ms(k, a.reps) * 1e3, 1)
import time
t1 = time.perf_counter(); core.run_steps(1 + a.steps, 20); core.lib.fesom_gpu_sync(); el = (time.perf_counter() - t1) / 20 * 1e3
print(json.dumps({"tag": a.tag, "env": {k: v for k, v in os.environ.items() if k.startswith("FESOM_GPU_")}, "us": out, "ms_per_step": round(el, 3)}), flush=True)
core.close()
