"""Isolated device time of named kernels / routines (graph of nrep launches) on the pi mesh after 50 spin-up steps."""
import os, sys
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
for k in sys.argv[1:]:
    print(k, round(core.kernel_time_ms(k, 20) * 1e3, 2), "us")
