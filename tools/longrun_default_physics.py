import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from fesom2_amd.mesh import Mesh
from fesom2_amd.config import make_params
from fesom2_amd.core import OceanCore
from fesom2_amd.synthetic import analytic_ts, analytic_forcing, analytic_sw_3d
PI = os.path.join(os.getcwd(), "tests", "golden", "meshes", "pi")
mesh = Mesh.load(PI, dt=900.0)
par = make_params(dt=900.0, mix_scheme="KPP", Fer_GM=True, Redi=True, use_sw_pene=True)
st = mesh.initial_state(2); st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI); st.tr_arr_old[...] = st.tr_arr
core = OceanCore(mesh, par); core.upload_state(st)
f = analytic_forcing(mesh); f = {k: v * 0.3 for k, v in f.items()}; f["sw_3d"] = analytic_sw_3d(mesh, f["heat_flux"])
core.set_forcing(**f)
t0 = time.time(); n = 1
for chunk in range(10):
    core.run_steps(n, 3504); n += 3504
    si = core.step_info()
    print(f"day {n*900/86400:7.1f} its {core.solver_iterations} (safety net so far: {core.lib.fesom_gpu_solver_safety_net_count()}) eta {si['min_eta']:.2f} {si['max_eta']:.2f} T {si['min_temp']:.2f} {si['max_temp']:.2f} S {si['min_salt']:.2f} {si['max_salt']:.2f} umax {max(abs(si['min_uvel']), si['max_uvel']):.2f} cfl_z {si['max_cfl_z']:.3f} blowup {si['blowup']}", flush=True)
    if si["blowup"]: break
print("wall s", round(time.time() - t0, 1), "for", n - 1, "steps =", round((n - 1) * 900 / 86400 / 365, 2), "years")
