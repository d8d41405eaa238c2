import os, sys, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from fesom2_amd.mesh import Mesh
from fesom2_amd.config import make_params
from fesom2_amd.core import OceanCore
from fesom2_amd.synthetic import analytic_ts
pi = "tests/golden/meshes/pi"
mesh = Mesh.load(pi, dt=900.0); par = make_params(dt=900.0)
st = mesh.initial_state(2); st.tr_arr[0], st.tr_arr[1] = analytic_ts(pi); st.tr_arr_old[...] = st.tr_arr
core = OceanCore(mesh, par); core.upload_state(st)
core.run_steps(1, 50)
for r in ("compute_vel_nodes", "pressure_bv", "pressure_force", "compute_sigma_xy", "mixing_pp", "compute_vel_rhs",
          "visc_filt_bcksct", "impl_vert_visc_ale", "update_stiff_mat_ale", "compute_ssh_rhs_ale", "solver_snapshot"):
    core.call(r)
t = core.kernel_time_ms("k_solver_replay", 10) * 1e3
print("MAXITS", os.environ.get("FESOM_SOLVER_MAXITS"), "GEN", os.environ.get("FESOM_GPU_GENERIC_SOLVER"), "us", round(t, 1), "its", core.solver_iterations)
