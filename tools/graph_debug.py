"""Step time without torch in the process (system HIP runtime): eager DAG vs FESOM_GPU_GRAPH=1 vs FESOM_GPU_SERIAL=1."""
import sys, os, time
sys.path.insert(0, os.getcwd())
from fesom2_amd.mesh import Mesh
from fesom2_amd.config import make_params
from fesom2_amd.core import OceanCore
from fesom2_amd.synthetic import analytic_ts
PI = os.path.join(os.getcwd(), "tests", "golden", "meshes", "pi")
mesh = Mesh.load(PI, dt=900.0)
st = mesh.initial_state(2); st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI); st.tr_arr_old[...] = st.tr_arr
core = OceanCore(mesh, make_params(dt=900.0)); core.upload_state(st)
core.run_steps(1, 200); core.get("eta_n", 3140)
t0 = time.perf_counter(); core.run_steps(201, 2000); e = core.get("eta_n", 3140); t1 = time.perf_counter()
print(os.environ.get("FESOM_GPU_GRAPH"), os.environ.get("FESOM_GPU_SERIAL"), "ms/step", round((t1 - t0) / 2000 * 1e3, 4), "eta0", e[0])
