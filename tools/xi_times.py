#!/usr/bin/env python3
"""Per-phase device times of the explicit-inverse SSH solve on pi (HIP events, each phase relaunched inside one captured graph)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from fesom2_amd import workloads
from fesom2_amd.core import OceanCore
wl = workloads.pi("default")
mesh = wl.load_mesh()
for K in (1, 2, 3):
    core = OceanCore(mesh, wl.params(solver_xinv_its=K))
    wl.start(core, mesh)
    core.run_steps(1, 200); core.lib.fesom_gpu_sync()
    its = core.solver_iterations
    for r in ("compute_vel_nodes", "pressure_bv", "pressure_force", "compute_sigma_xy", "mixing_kpp", "compute_vel_rhs", "visc_filt_bcksct", "impl_vert_visc_ale",
              "update_stiff_mat_ale", "compute_ssh_rhs_ale", "solver_snapshot"):
        core.call(r)
    t = core.kernel_time_ms("k_solver_replay", 20) * 1e3
    print(f"K={K}: whole solve {t:.1f} us, iterations {core.solver_iterations} (running step: {its})", flush=True)
    if K == 2:
        core.call("xi_arm")
        out = []
        for k in ("xi_setup", "xi_init", "xi_gemv0", "xi_spmv1", "xi_gemv1", "xi_spmv2"):
            out.append(f"{k} {core.kernel_time_ms(k, 50) * 1e3:.2f} us")
        print(" | ".join(out), flush=True)
    core.close()
