#!/usr/bin/env python3
"""Diagnose a refined-mesh run with the device-side step monitor: usage refine_probe.py LEVELS NSTEPS [dt] [physics]"""
import os, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from fesom2_amd import mesh_refine
from fesom2_amd.mesh import Mesh
from fesom2_amd.config import make_params
from fesom2_amd.core import OceanCore
from fesom2_amd.synthetic import analytic_ts
L, nsteps = int(sys.argv[1]), int(sys.argv[2])
dt = float(sys.argv[3]) if len(sys.argv) > 3 else 900.0
kw = dict(sys.argv[4:] and [a.split("=") for a in sys.argv[4:]])
kw = {k: (float(v) if v.replace(".", "").replace("e-", "").isdigit() else v) for k, v in kw.items()}
pi = os.path.join(REPO, "tests", "golden", "meshes", "pi")
d = os.path.join(tempfile.gettempdir(), f"pi_ref_{L}")
t0 = time.time()
if L > 0:
    mesh_refine.refine(pi, d, L)
else:
    d = pi
mesh = Mesh.load(d, dt=dt)
print("mesh", mesh.nod2D, mesh.elem2D, "load s", round(time.time() - t0, 1), "min resolution km", float(np.sqrt(mesh.elem_area.min() * 2) / 1e3), flush=True)
contrast = float(kw.pop("contrast", 1.0)); forcing = float(kw.pop("forcing", 0.0))
par = make_params(dt=dt, **kw)
st = mesh.initial_state(2)
st.tr_arr[0], st.tr_arr[1] = analytic_ts(d, contrast)
st.tr_arr_old[...] = st.tr_arr
core = OceanCore(mesh, par)
core.upload_state(st)
if forcing:
    from fesom2_amd.synthetic import analytic_forcing
    core.set_forcing(**{k: v * forcing for k, v in analytic_forcing(mesh).items()})
for n0 in range(1, nsteps + 1, 10):
    core.run_steps(n0, 10)
    si = core.step_info()
    print(n0 + 9, "its", core.solver_iterations, "eta", f"{si['min_eta']:.3e} {si['max_eta']:.3e}", "T", f"{si['min_temp']:.2f} {si['max_temp']:.2f}",
          "u", f"{si['min_uvel']:.2e} {si['max_uvel']:.2e}", "w2", f"{si['max_wvel2']:.2e}", "cfl_z", f"{si['max_cfl_z']:.3f}", "Av", f"{si['max_av']:.2e}", "blowup", si["blowup"], flush=True)
    if si["blowup"]:
        break
core.close()
