import os, sys, numpy as np
sys.path.insert(0, "/root/repo/tests")
from fesom2_amd.mesh import Mesh
from fesom2_amd.config import make_params
from fesom2_amd.core import OceanCore
from fesom2_amd.synthetic import analytic_ts, analytic_forcing
from oracle_lib import Oracle
from test_gpu_parity import full_chain, compare
V = sys.argv[1]
D = "/root/repo/tests/golden/meshes/" + ("pi" if V == "v1" else "pi_cavity")
mesh = Mesh.load(D, dt=900.0, use_cavity=(V != "v1"))
KW = {"v1": dict(mix_scheme="KPP", Fer_GM=True, Redi=True), "v2": dict(mix_scheme="KPP", Redi=True), "v3": dict(Fer_GM=True, Redi=True), "v4": dict(mix_scheme="KPP"), "v5": dict(Redi=True), "v6": dict(Fer_GM=True)}[V]
print("variant", V, KW)
par = make_params(dt=900.0, use_cavity=True, **KW)
st = mesh.initial_state(2)
st.tr_arr[0], st.tr_arr[1] = analytic_ts(D)
st.tr_arr_old[...] = st.tr_arr
gpu, orc = OceanCore(mesh, par), Oracle(mesh, par)
gpu.upload_state(st); orc.set_state(st)
forcing = analytic_forcing(mesh)
gpu.set_forcing(**forcing)
for k, v in forcing.items():
    orc.set(k, v)
nl = mesh.nl
nbad = 0
for step in range(1, 4):
    for routine, arg, fields in full_chain(2):
        gpu.call(routine, arg); orc.call(routine, arg)
        for f in list(fields) + (["hpressure"] if routine == "pressure_bv" else []):
            a, b = gpu.get(f, orc.count(f)), orc.get(f)
            ok, msg = compare(f, a, b)
            if not ok:
                nbad += 1
                print(f"step {step} {routine}({arg}) {msg}")
                a = np.asarray(a).ravel(); b = np.asarray(b).ravel()
                idx = np.flatnonzero(~((a == b) | (np.isnan(a) & np.isnan(b))))
                for i in idx[:3]:
                    for lev in (nl - 1, nl):
                        pass
                    print("    flat", i, "nlm1:", (i % (nl - 1)) + 1, i // (nl - 1), " nl:", (i % nl) + 1, i // nl, "gpu", a[i], "orc", b[i])
    if nbad:
        break
print("ulev_n max", mesh.ulevels_nod2D.max())
