"""GPU parity tests: the HIP path (through the C ABI of libfesom_gpu.so) against the CPU oracle on the
same inputs, routine by routine and over whole steps, on the pi mesh (config #2 of BASELINE.json:
3140 nodes, 47 layers, T/S tracers).  Bar: bit-exact for every field (fp64, no FMA contraction, reference
summation order), except slope_tapered which passes through tanh (device libm vs glibc: rel <= 1e-12)."""
import os
import numpy as np
import pytest

from parity_chain import full_chain, compare

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PI = os.path.join(REPO, "tests", "golden", "meshes", "pi")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(built):
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.core import OceanCore
    from fesom2_amd.synthetic import analytic_ts
    from oracle_lib import Oracle
    mesh = Mesh.load(PI, dt=900.0)
    par = make_params(dt=900.0)
    st = mesh.initial_state(2)
    T, S = analytic_ts(PI)
    st.tr_arr[0], st.tr_arr[1] = T, S
    st.tr_arr_old[...] = st.tr_arr
    gpu = OceanCore(mesh, par)
    orc = Oracle(mesh, par)
    gpu.upload_state(st)
    orc.set_state(st)
    yield mesh, par, gpu, orc
    gpu.close()


def test_routine_chain_bitwise(setup):
    mesh, par, gpu, orc = setup
    failures = []
    for step in range(1, 4):
        for routine, arg, fields in full_chain(2):
            gpu.call(routine, arg)
            orc.call(routine, arg)
            for f in fields:
                n = orc.count(f)
                ok, msg = compare(f, gpu.get(f, n), orc.get(f))
                if not ok:
                    failures.append(f"step {step} {routine}({arg}) {msg}")
            if failures:
                break
        if failures:
            break
    assert not failures, "\n".join(failures)
    assert gpu.solver_iterations == orc.solver_iterations


def test_whole_steps_and_conservation(setup):
    """20 further steps through the graph-replayed step vs the oracle's step: prognostic state bitwise;
    tracer content sum(T*h*A) conserved to 1e-12 relative (no surface fluxes)."""
    mesh, par, gpu, orc = setup
    nlm1 = mesh.nl - 1
    asv = mesh.areasvol[:, :nlm1]

    def content(T, h):
        return float((T * h * asv).sum())
    h0 = gpu.get("hnode", orc.count("hnode")).reshape(-1, nlm1)
    T0 = gpu.get("tr_arr", orc.count("tr_arr")).reshape(2, -1, nlm1)
    c0 = [content(T0[i], h0) for i in range(2)]
    gpu.run_steps(4, 20)
    for n in range(20):
        orc.call("step", 4 + n)
    for f in ("tr_arr", "UV", "eta_n", "hnode", "hbar", "Wvel", "ssh_rhs_old", "UV_rhsAB", "tr_arr_old"):
        ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
        assert ok, msg
    h1 = gpu.get("hnode", orc.count("hnode")).reshape(-1, nlm1)
    T1 = gpu.get("tr_arr", orc.count("tr_arr")).reshape(2, -1, nlm1)
    for i in range(2):
        c1 = content(T1[i], h1)
        assert abs(c1 - c0[i]) / abs(c0[i]) < 1e-12, (i, c0[i], c1)
    eta = gpu.get("eta_n", orc.count("eta_n"))
    assert np.isfinite(eta).all() and np.abs(eta).max() < 5.0


def test_solver_iteration_schedule_across_the_spinup_bitwise(built):
    """The default schedule of the explicit-inverse solve (solver_xinv_its = 0): two enqueued iterations for the first 300 solves after init, one
    afterwards (csrc/solver.hip:launch_solver_xinv; the oracle counts the same way).  310 whole steps from rest: HIP == oracle bit for bit on both
    sides of the switch; a fixed K = 1 run is a different (equally converged) iterate sequence."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.core import OceanCore
    from fesom2_amd.synthetic import analytic_ts
    from oracle_lib import Oracle
    mesh = Mesh.load(PI, dt=900.0)
    par = make_params(dt=900.0)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    gpu, orc = OceanCore(mesh, par), Oracle(mesh, par)
    gpu.upload_state(st); orc.set_state(st)
    for lo, hi in ((1, 299), (299, 311)):            # up to step 298, then across the switch
        gpu.run_steps(lo, hi - lo)
        for n in range(lo, hi):
            orc.call("step", n)
        for f in ("eta_n", "d_eta", "UV", "tr_arr", "hnode"):
            ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
            assert ok, (hi, msg)
    eta = gpu.get("eta_n", orc.count("eta_n"))
    gpu.close()
    par1 = make_params(dt=900.0, solver_xinv_its=1)
    g1 = OceanCore(mesh, par1)
    g1.upload_state(st)
    g1.run_steps(1, 310)
    e1 = g1.get("eta_n", orc.count("eta_n"))
    g1.close()
    assert 0.0 < np.abs(e1 - eta).max() < 1e-8


def test_psolve_abi(setup):
    """psolver_init/psolve with the reference's C signatures (src/psolve.c:16,152): residual of the
    row-scaled system below the reference's tolerance 1e-10."""
    import ctypes as C
    mesh, par, gpu, orc = setup
    lib = gpu.lib
    n, nza = mesh.myDim_nod2D, mesh.ssh_nza
    rp = (mesh.ssh_rowptr - mesh.ssh_rowptr[0]).astype(np.int32)
    ci = (mesh.ssh_colind_loc - 1).astype(np.int32)
    vals = mesh.ssh_values.copy()
    rng = np.random.default_rng(7)
    xt = rng.standard_normal(n)
    rhs = np.zeros(n)
    for i in range(n):
        rhs[i] = (vals[rp[i]:rp[i + 1]] * xt[ci[rp[i]:rp[i + 1]]]).sum()
    sol = np.zeros(n)
    part = np.array([0, n], dtype=np.int32)
    ip = lambda v: C.byref(C.c_int(v))
    dp = lambda v: C.byref(C.c_double(v))
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    lib.psolver_init(ip(1), ip(6), ip(2), ip(1), ip(2), ip(3), dp(1e-8), ip(2000), ip(15), dp(1e-10), P(part, C.c_int), P(rp, C.c_int),
                     P(ci, C.c_int), P(vals, C.c_double), ip(1), ip(0))
    lib.psolve(ip(1), P(rhs, C.c_double), P(vals, C.c_double), P(sol, C.c_double), ip(1))
    lib.psolver_final()
    scale = np.array([1.0 / np.abs(vals[rp[i]:rp[i + 1]]).sum() for i in range(n)])
    res = np.array([(vals[rp[i]:rp[i + 1]] * sol[ci[rp[i]:rp[i + 1]]]).sum() for i in range(n)]) - rhs
    assert np.sqrt(((res * scale) ** 2).sum()) < 2e-10
    assert np.abs(sol - xt).max() < 1e-6 * np.abs(xt).max()


def test_gm_chain_and_steps_bitwise(built):
    """Gent-McWilliams bolus velocities (Fer_GM=.true.): routine chain over 3 steps and 10 whole steps, HIP == oracle bitwise
    (the oracle is pinned against a reference run with the same options, tests/test_oracle_vs_reference.py)"""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.core import OceanCore
    from fesom2_amd.synthetic import analytic_ts
    from oracle_lib import Oracle
    mesh = Mesh.load(PI, dt=900.0)
    par = make_params(dt=900.0, Fer_GM=True, scaling_Ferreira=True)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    gpu, orc = OceanCore(mesh, par), Oracle(mesh, par)
    gpu.upload_state(st); orc.set_state(st)
    failures = []
    for step in range(1, 4):
        for routine, arg, fields in full_chain(2, gm=True):
            gpu.call(routine, arg); orc.call(routine, arg)
            for f in fields:
                ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
                if not ok:
                    failures.append(f"step {step} {routine}({arg}) {msg}")
        if failures:
            break
    assert not failures, "\n".join(failures[:10])
    gpu.run_steps(4, 10)
    for n in range(10):
        orc.call("step", 4 + n)
    for f in ("tr_arr", "UV", "eta_n", "hnode", "fer_UV", "fer_Wvel"):
        ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
        assert ok, msg
    gpu.close()


@pytest.mark.parametrize("gm", [True, False])
def test_redi_chain_bitwise_and_steps(built, gm):
    """Isoneutral (Redi) diffusion, with and without GM.  The tapered slope goes through tanh, the one libm call where the
    device library and glibc differ in the last bits (ULP_FIELDS), so:
    (1) routine chain over 3 steps with the oracle's slope_tapered handed to the HIP side after compute_neutral_slope:
        every Redi kernel (Ki scaling, slope terms of the horizontal flux, explicit vertical flux, K33 in the implicit
        solve) must then be BITWISE equal to the oracle;
    (2) 10 free-running steps: relative agreement 1e-9 (tolerance: tanh ulps amplified through 10 steps)."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.core import OceanCore
    from fesom2_amd.synthetic import analytic_ts
    from oracle_lib import Oracle
    mesh = Mesh.load(PI, dt=900.0)
    par = make_params(dt=900.0, Fer_GM=gm, scaling_Ferreira=True, Redi=True)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    gpu, orc = OceanCore(mesh, par), Oracle(mesh, par)
    gpu.upload_state(st); orc.set_state(st)
    failures = []
    for step in range(1, 4):
        for routine, arg, fields in full_chain(2, gm=gm, redi=True):
            gpu.call(routine, arg); orc.call(routine, arg)
            for f in fields:
                ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
                if not ok:
                    failures.append(f"step {step} {routine}({arg}) {msg}")
            if routine == "compute_neutral_slope":
                gpu.set("slope_tapered", orc.get("slope_tapered"))
        if failures:
            break
    assert not failures, "\n".join(failures[:10])
    gpu.run_steps(4, 10)
    for n in range(10):
        orc.call("step", 4 + n)
    for f in ("tr_arr", "UV", "eta_n", "hnode", "Ki"):
        a, b = gpu.get(f, orc.count(f)), orc.get(f)
        err = np.abs(a - b).max() / np.abs(b).max()
        assert err < 1e-9, (f, err)
    gpu.close()


@pytest.mark.parametrize("full,sw,nonlcl", [(False, False, ""), (True, False, ""), (True, True, ""), (False, True, ""), (False, False, "zstar"), (False, True, "linfs"), (False, False, "dd")])
def test_kpp_chain_and_steps(built, full, sw, nonlcl):
    """KPP vertical mixing under surface forcing (full = with GM + Redi, the reference's default physics): routine chain over
    3 steps, HIP == oracle bitwise (with Redi the oracle's tapered slopes are handed over, see the Redi test), then 10 whole
    steps through the step graph: bitwise without Redi, 1e-9 relative with it.  nonlcl: + use_kpp_nonlclflx (non-local transport of heat and,
    with linfs where the reference keeps ref_sss, of salt; oracle pinned on the reference runs pi_kpp_nonlcl / pi_kpp_nonlcl_linfs); "dd": double_diffusion
    (ddmix; oracle pinned on pi_kpp_dd)."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.core import OceanCore
    from fesom2_amd.synthetic import analytic_ts
    from oracle_lib import Oracle
    akw = dict(which_ale="linfs", use_partial_cell=False) if nonlcl == "linfs" else {}
    mesh = Mesh.load(PI, dt=900.0, **akw)
    par = make_params(dt=900.0, mix_scheme="KPP", Fer_GM=full, Redi=full, scaling_Ferreira=full, use_sw_pene=sw, use_kpp_nonlclflx=nonlcl in ("zstar", "linfs"), double_diffusion=(nonlcl == "dd"), **akw)   # sw: short-wave penetration
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    if nonlcl == "dd":
        # ddmix compares alpha*T with beta*S (the tracer values themselves, oce_ale_mixing_kpp.F90:880-883): with oceanic salinities neither branch is ever
        # taken -- the reference run pi_kpp_dd is bit-identical to pi_kpp, which is all that pins the oracle.  A brackish state (S ~ 5) takes the salt-fingering
        # branch (the diffusive-convection branch needs beta*S < alpha*T < 0 and cannot be reached): HIP == oracle there, parity with the reference unpinned.
        st.tr_arr[1] *= 0.15
    st.tr_arr_old[...] = st.tr_arr
    gpu, orc = OceanCore(mesh, par), Oracle(mesh, par)
    gpu.upload_state(st); orc.set_state(st)
    from fesom2_amd.synthetic import analytic_forcing, analytic_sw_3d
    forcing = analytic_forcing(mesh)
    if sw:
        forcing["sw_3d"] = analytic_sw_3d(mesh, forcing["heat_flux"])
    gpu.set_forcing(**forcing)
    for k, v in forcing.items():
        orc.set(k, v)
    failures = []
    for step in range(1, 4):
        for routine, arg, fields in full_chain(2, gm=full, redi=full, kpp=True):
            gpu.call(routine, arg); orc.call(routine, arg)
            for f in fields:
                ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
                if not ok:
                    failures.append(f"step {step} {routine}({arg}) {msg}")
            if full and routine == "compute_neutral_slope":
                gpu.set("slope_tapered", orc.get("slope_tapered"))
        if failures:
            break
    assert not failures, "\n".join(failures[:10])
    if nonlcl == "dd":
        k1, k2 = orc.get("kpp_Kv1"), orc.get("kpp_Kv2")
        assert (k2 > k1).sum() > 1000                        # the salt-fingering branch is taken (Kv2 - Kv1 = 0.3 diffdd)
    gpu.run_steps(4, 10)
    for n in range(10):
        orc.call("step", 4 + n)
    for f in ("tr_arr", "UV", "eta_n", "hnode", "Kv", "Av", "kpp_hbl"):
        a, b = gpu.get(f, orc.count(f)), orc.get(f)
        if full:
            err = np.abs(a - b).max() / np.abs(b).max()
            assert err < (1e-9 if f != "kpp_hbl" else 1e-6), (f, err)
        else:
            ok, msg = compare(f, a, b)
            assert ok, msg
    gpu.close()


def test_step_info_device_monitor(built):
    """fesom_gpu_step_info (write_step_info + check_blowup on the device) against the oracle's restatement after 5 steps of the
    default physics: extrema and the blow-up flag exact, the area-weighted sums to 1e-13 relative of sum|terms| (different, fixed
    summation order); then a NaN / an out-of-range temperature planted in the device state must raise the flag."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.core import OceanCore
    from fesom2_amd.synthetic import analytic_ts, analytic_forcing
    from fesom2_amd._lib import STEP_INFO_FIELDS
    from oracle_lib import Oracle
    mesh = Mesh.load(PI, dt=900.0)
    par = make_params(dt=900.0, mix_scheme="KPP", Fer_GM=True)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    gpu, orc = OceanCore(mesh, par), Oracle(mesh, par)
    gpu.upload_state(st); orc.set_state(st)
    forcing = analytic_forcing(mesh)
    gpu.set_forcing(**forcing)
    for k, v in forcing.items():
        orc.set(k, v)
    gpu.run_steps(1, 5)
    for n in range(1, 6):
        orc.call("step", n)
    a, b = gpu.step_info(), dict(zip(STEP_INFO_FIELDS, orc.step_info()))
    area = mesh.areasvol[:, 0]
    for k in STEP_INFO_FIELDS:
        if k.startswith("sum_"):
            scale = {"sum_eta": np.abs(area * orc.get("eta_n")).sum(), "sum_hbar": np.abs(area * orc.get("hbar")).sum(),
                     "sum_deta": np.abs(area * orc.get("d_eta")).sum(), "sum_dhbar": np.abs(area * orc.get("d_eta")).sum(),
                     "sum_wflux": np.abs(area * forcing["water_flux"]).sum(), "sum_area": area.sum()}[k]
            assert abs(a[k] - b[k]) <= 1e-13 * scale, (k, a[k], b[k])
        else:
            assert a[k] == b[k], (k, a[k], b[k])
    assert a["blowup"] == 0.0 and a["min_temp"] > -2 and a["max_salt"] < 40 and a["max_cfl_z"] > 0
    T = gpu.get("tr_arr", orc.count("tr_arr"))          # (nl-1, N, 2): first level of node 262 = a wet temperature value
    T2 = T.copy(); T2[262 * (mesh.nl - 1)] = 75.0
    gpu.set("tr_arr", T2)
    assert gpu.step_info()["blowup"] == 1.0
    gpu.set("tr_arr", T)
    assert gpu.step_info()["blowup"] == 0.0
    e = gpu.get("eta_n", mesh.myDim_nod2D); e[7] = np.nan
    gpu.set("eta_n", e)
    assert gpu.step_info()["blowup"] == 1.0
    gpu.close()


def test_w_split_chain_and_steps_bitwise(built):
    """w_split=.true. with a threshold that makes the split active on pi (w_max_cfl = 0.0003): explicit/implicit vertical velocity,
    implicit vertical advection in the momentum solve and in the low-order FCT solution (adv_tra_vert_impl); HIP == oracle bitwise
    over the routine chain (3 steps) and 10 graph-replayed steps, under surface forcing."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.core import OceanCore
    from fesom2_amd.synthetic import analytic_ts, analytic_forcing
    from oracle_lib import Oracle
    mesh = Mesh.load(PI, dt=900.0)
    par = make_params(dt=900.0, w_split=True, w_max_cfl=0.0003)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    gpu, orc = OceanCore(mesh, par), Oracle(mesh, par)
    gpu.upload_state(st); orc.set_state(st)
    forcing = analytic_forcing(mesh)
    gpu.set_forcing(**forcing)
    for k, v in forcing.items():
        orc.set(k, v)
    failures = []
    for step in range(1, 4):
        for routine, arg, fields in full_chain(2):
            gpu.call(routine, arg); orc.call(routine, arg)
            for f in fields:
                ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
                if not ok:
                    failures.append(f"step {step} {routine}({arg}) {msg}")
        if failures:
            break
    assert not failures, "\n".join(failures[:10])
    assert np.count_nonzero(orc.get("Wvel_i")) > 1000
    gpu.run_steps(4, 10)
    for n in range(10):
        orc.call("step", 4 + n)
    for f in ("tr_arr", "UV", "eta_n", "hnode", "Wvel_i"):
        ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
        assert ok, msg
    gpu.close()


FUZZ = [
    dict(dt=450.0, which_ale="zstar", use_partial_cell=False, state_equation=1, mix_scheme="PP", w_split=True, w_max_cfl=0.0005),
    dict(dt=1200.0, which_ale="zstar", use_partial_cell=True, state_equation=0, mix_scheme="KPP", Fer_GM=True, scaling_Ferreira=True),
    dict(dt=900.0, which_ale="linfs", use_partial_cell=False, state_equation=1, mix_scheme="KPP", use_sw_pene=True),
    dict(dt=900.0, which_ale="linfs", use_partial_cell=True, state_equation=0, mix_scheme="PP", Fer_GM=True, scaling_FESOM14=True, K_GM_bvref=1),
    dict(dt=600.0, which_ale="zstar", use_partial_cell=True, state_equation=1, mix_scheme="none", use_windmix=True, K_hor=500.0),
    dict(dt=900.0, which_ale="zstar", use_partial_cell=True, state_equation=1, mix_scheme="KPP", Fer_GM=True, K_GM_bvref=0, scaling_Ferreira=True,
         w_split=True, w_max_cfl=0.001, use_sw_pene=True, solver_x0_order=0),
    dict(dt=900.0, which_ale="zstar", use_partial_cell=True, state_equation=1, mix_scheme="PP", visc_option=6),
    dict(dt=900.0, which_ale="linfs", use_partial_cell=False, state_equation=1, mix_scheme="KPP", visc_option=7, Fer_GM=True, scaling_Ferreira=True),
    dict(dt=900.0, which_ale="zstar", use_partial_cell=True, state_equation=1, mix_scheme="PP", visc_option=7, w_split=True, w_max_cfl=0.0005),
    dict(dt=900.0, which_ale="zstar", use_partial_cell=True, state_equation=1, mix_scheme="PP", tra_adv_ver="CDIFF"),
    dict(dt=900.0, which_ale="zstar", use_partial_cell=True, state_equation=1, mix_scheme="KPP", tra_adv_ver="UPW1", w_split=True, w_max_cfl=0.0003),
    dict(dt=900.0, which_ale="linfs", use_partial_cell=False, state_equation=0, mix_scheme="PP", tra_adv_ver="CDIFF", Fer_GM=True, scaling_Ferreira=True, visc_option=6),
    dict(dt=900.0, which_ale="zstar", use_partial_cell=True, state_equation=1, mix_scheme="PP", tra_adv_hor="MUSCL"),
    dict(dt=900.0, which_ale="zstar", use_partial_cell=True, state_equation=1, mix_scheme="KPP", tra_adv_hor="UPW1", tra_adv_ver="CDIFF"),
    dict(dt=900.0, which_ale="linfs", use_partial_cell=True, state_equation=1, mix_scheme="PP", tra_adv_hor="MUSCL", tra_adv_ver="UPW1", w_split=True, w_max_cfl=0.0005),
    dict(dt=900.0, which_ale="zstar", use_partial_cell=True, state_equation=1, mix_scheme="PP", tra_adv_ver="PPM"),
    dict(dt=1200.0, which_ale="zstar", use_partial_cell=False, state_equation=1, mix_scheme="KPP", tra_adv_ver="PPM", Fer_GM=True, scaling_Ferreira=True, w_split=True, w_max_cfl=0.0005),
]


@pytest.mark.parametrize("case", range(len(FUZZ)))
def test_option_combinations_bitwise(built, case):
    """Combinations of the options the library accepts (ALE variant, partial cells, EOS, mixing scheme, GM scalings, w_split, short-wave
    penetration, time step, initial-guess order), 8 whole steps under scaled surface forcing: HIP == oracle bit for bit.  (Redi stays
    out: tanh, see test_redi_chain_bitwise_and_steps.)  The oracle is pinned on reference runs for the configurations of
    tests/test_oracle_vs_reference.py; here the point is that every option path of the HIP code agrees with its restatement."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.core import OceanCore
    from fesom2_amd.synthetic import analytic_ts, analytic_forcing, analytic_sw_3d
    from oracle_lib import Oracle
    kw = dict(FUZZ[case])
    dt, ale, pc = kw.pop("dt"), kw.pop("which_ale"), kw.pop("use_partial_cell")
    mesh = Mesh.load(PI, dt=dt, which_ale=ale, use_partial_cell=pc)
    par = make_params(dt=dt, which_ale=ale, use_partial_cell=pc, **kw)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    gpu, orc = OceanCore(mesh, par), Oracle(mesh, par)
    gpu.upload_state(st); orc.set_state(st)
    forcing = {k: v * (0.5 + 0.25 * case) for k, v in analytic_forcing(mesh).items()}
    if kw.get("use_sw_pene"):
        forcing["sw_3d"] = analytic_sw_3d(mesh, forcing["heat_flux"])
    gpu.set_forcing(**forcing)
    for k, v in forcing.items():
        orc.set(k, v)
    gpu.run_steps(1, 8)
    for n in range(1, 9):
        orc.call("step", n)
    for f in ("tr_arr", "UV", "eta_n", "hnode", "Wvel", "Kv", "Av"):
        ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
        assert ok, (FUZZ[case], msg)
    gpu.close()


@pytest.mark.parametrize("case", [1, 5])
def test_profile_step_is_the_same_step(built, case):
    """fesom_gpu_profile_step (the reference's rtime_oce_* phase timers, oce_ale.F90:2771-2777, taken with HIP events around the
    named routines run one after the other) advances the model exactly as fesom_gpu_step does: 6 profiled steps == oracle bit
    for bit; the phase times are positive, the solver is part of dynssh, and the phases add up to the total."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.core import OceanCore
    from fesom2_amd.synthetic import analytic_ts, analytic_forcing, analytic_sw_3d
    from oracle_lib import Oracle
    kw = dict(FUZZ[case])
    dt, ale, pc = kw.pop("dt"), kw.pop("which_ale"), kw.pop("use_partial_cell")
    mesh = Mesh.load(PI, dt=dt, which_ale=ale, use_partial_cell=pc)
    par = make_params(dt=dt, which_ale=ale, use_partial_cell=pc, **kw)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    gpu, orc = OceanCore(mesh, par), Oracle(mesh, par)
    gpu.upload_state(st); orc.set_state(st)
    forcing = analytic_forcing(mesh)
    if kw.get("use_sw_pene"):
        forcing["sw_3d"] = analytic_sw_3d(mesh, forcing["heat_flux"])
    gpu.set_forcing(**forcing)
    for k, v in forcing.items():
        orc.set(k, v)
    gpu.run_steps(1, 2)                      # normal and profiled steps may be mixed freely
    for n in range(3, 9):
        t = gpu.profile_step(n)
        assert all(v > 0.0 for v in t.values()), t
        assert t["solvessh"] < t["dynssh"], t
        parts = t["mixpres"] + t["dyn"] + t["dynssh"] + t["GMRedi"] + t["solvetra"]
        assert parts <= t["total"] * 1.0001 and parts > 0.8 * t["total"], t
    for n in range(1, 9):
        orc.call("step", n)
    for f in ("tr_arr", "UV", "eta_n", "hnode", "Wvel", "Kv", "Av"):
        ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
        assert ok, (FUZZ[case], msg)
    gpu.close()


@pytest.mark.parametrize("kw", [dict(), dict(w_split=True, w_max_cfl=0.0003), dict(w_split=True, w_max_cfl=0.0003, tile=1)], ids=["non", "non_wsplit", "non_wsplit_tile"])
def test_no_limiter_chain_bitwise(built, kw):
    """tra_adv_lim = 'NON' (oracle pinned on the reference runs pi_pp_non / pi_pp_non_wsplit): HIP == oracle bit for bit after every routine of 3 steps under
    surface forcing, and for the state after 8 further steps through fesom_gpu_run_steps.  With w_split the implicit part of the vertical velocity enters the
    diffusion solve (do_wimpl: tru_coeffs in kernels_tra.hip); `_tile`: the CORE2-class kernel shapes."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.core import OceanCore
    from fesom2_amd.synthetic import analytic_ts, analytic_forcing
    from oracle_lib import Oracle
    mesh = Mesh.load(PI, dt=900.0)
    kw = dict(kw)
    tile = kw.pop("tile", None)
    par = make_params(dt=900.0, tra_adv_lim="NON", **kw)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    old = os.environ.get("FESOM_GPU_TILE")
    if tile:
        os.environ["FESOM_GPU_TILE"] = str(tile)
    try:
        gpu = OceanCore(mesh, par)
    finally:
        if tile:
            if old is None:
                os.environ.pop("FESOM_GPU_TILE", None)
            else:
                os.environ["FESOM_GPU_TILE"] = old
    orc = Oracle(mesh, par)
    gpu.upload_state(st); orc.set_state(st)
    forcing = analytic_forcing(mesh)
    gpu.set_forcing(**forcing)
    for k, v in forcing.items():
        orc.set(k, v)
    failures = []
    for step in range(1, 4):
        for routine, arg, fields in full_chain(2):
            gpu.call(routine, arg); orc.call(routine, arg)
            for f in fields:
                if f in ("fct_LO", "fct_plus", "fct_minus"):        # not formed without the limiter
                    continue
                ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
                if not ok:
                    failures.append(f"step {step} {routine}({arg}) {msg}")
        if failures:
            break
    assert not failures, "\n".join(failures[:10])
    gpu.run_steps(4, 8)
    for n in range(4, 12):
        orc.call("step", n)
    for f in ("tr_arr", "UV", "eta_n", "hnode"):
        ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
        assert ok, msg
    gpu.close()


@pytest.mark.parametrize("opt", [1, 2, 3, 4, 6, 7, 8])
def test_biharmonic_viscosity_chain_bitwise(built, opt):
    """visc_option 1 / 2 / 3 (h_viscosity_leith + visc_filt_harmon / visc_filt_hbhmix / visc_filt_biharm(2); the Leith coefficient and the relative
    vorticity are compared as well), 4 / 6 / 7 (visc_filt_biharm(1), visc_filt_bilapl, visc_filt_bidiff) and 8 (backscatter_coef + visc_filt_dbcksc +
    uke_update; the sub-grid energy budget is compared as well); oracle pinned on the reference runs pi_pp_visc1 .. 8: HIP == oracle bit for bit
    after every routine of 3 steps under surface forcing."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.core import OceanCore
    from fesom2_amd.synthetic import analytic_ts, analytic_forcing
    from oracle_lib import Oracle
    mesh = Mesh.load(PI, dt=900.0)
    par = make_params(dt=900.0, visc_option=opt)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    gpu, orc = OceanCore(mesh, par), Oracle(mesh, par)
    gpu.upload_state(st); orc.set_state(st)
    forcing = analytic_forcing(mesh)
    gpu.set_forcing(**forcing)
    for k, v in forcing.items():
        orc.set(k, v)
    failures = []
    for step in range(1, 4):
        for routine, arg, fields in full_chain(2):
            gpu.call(routine, arg); orc.call(routine, arg)
            extra = []
            if routine == "viscosity_filter":
                extra = ["Visc", "vorticity"] if opt <= 3 else ["v_back", "UV_dis_tend", "UV_back_tend", "uke_dif", "uke_dis", "uke_back", "uke_rhs", "uke_rhs_old", "uke"] if opt == 8 else []
            for f in list(fields) + extra:
                ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
                if not ok:
                    failures.append(f"step {step} {routine}({arg}) {msg}")
        if failures:
            break
    assert not failures, "\n".join(failures[:10])
    gpu.run_steps(4, 8)                                  # the whole-step launch order (stream DAG) as well
    for n in range(4, 12):
        orc.call("step", n)
    for f in ("UV", "eta_n", "tr_arr") + (("uke", "uke_rhs") if opt == 8 else ()):
        ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
        assert ok, msg
    gpu.close()


def test_zlevel_chain_bitwise(built):
    """which_ALE = 'zlevel' (k_vert_vel / k_thick zlevel branches; oracle pinned on the reference run pi_pp_zlevel): HIP == oracle bit for bit after every routine of
    3 steps and after 6 further whole steps; then from a state whose sub-surface layers 2 and 3 are thinner than at rest (what the reference's local-zstar fallback
    leaves behind), so that the "return to zlevel" branch refills them over several layers: again bit for bit."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.core import OceanCore
    from fesom2_amd.synthetic import analytic_ts, analytic_forcing
    from oracle_lib import Oracle
    mesh = Mesh.load(PI, dt=900.0, which_ale="zlevel")
    par = make_params(dt=900.0, which_ale="zlevel")
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    gpu, orc = OceanCore(mesh, par), Oracle(mesh, par)
    gpu.upload_state(st); orc.set_state(st)
    forcing = analytic_forcing(mesh)
    gpu.set_forcing(**forcing)
    for k, v in forcing.items():
        orc.set(k, v)
    failures = []
    for step in range(1, 4):
        for routine, arg, fields in full_chain(2):
            gpu.call(routine, arg); orc.call(routine, arg)
            for f in fields:
                ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
                if not ok:
                    failures.append(f"step {step} {routine}({arg}) {msg}")
        if failures:
            break
    assert not failures, "\n".join(failures[:10])
    gpu.run_steps(4, 6)
    for n in range(6):
        orc.call("step", 4 + n)
    for f in ("tr_arr", "UV", "eta_n", "hnode", "hnode_new", "helem", "zbar_3d_n", "Z_3d_n", "hbar", "Wvel"):
        ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
        assert ok, msg
    gpu.sync()                                                      # (no column needed the local-zstar fallback)
    h0 = st.hnode.reshape(-1, mesh.nl - 1)
    assert np.abs(orc.get("hnode").reshape(-1, mesh.nl - 1)[:, 1:] - h0[:, 1:]).max() == 0.0 and np.abs(orc.get("hnode").reshape(-1, mesh.nl - 1)[:, 0] - h0[:, 0]).max() > 1e-4
    # thinner sub-surface layers: the rising columns refill layer 3, then 2, then the surface layer
    hn = orc.get("hnode").reshape(-1, mesh.nl - 1).copy()
    deep = mesh.nlevels_nod2D_min > 8
    hn[deep, 1] -= 2.0e-4; hn[deep, 2] -= 1.0e-4
    for f in ("hnode", "hnode_new"):
        gpu.set(f, hn.ravel()); orc.set(f, hn.ravel())
    for step in range(10, 13):
        for routine, arg, fields in full_chain(2):
            gpu.call(routine, arg); orc.call(routine, arg)
            for f in list(fields) + (["hnode_new"] if routine == "vert_vel_ale" else []):
                ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
                if not ok:
                    failures.append(f"refill step {step} {routine}({arg}) {msg}")
        if failures:
            break
    assert not failures, "\n".join(failures[:10])
    hn2 = orc.get("hnode").reshape(-1, mesh.nl - 1)
    assert (hn2[deep, 2] > hn[deep, 2]).any() and (hn2[deep, 1] > hn[deep, 1]).any()       # layers 3 and 2 were refilled somewhere
    gpu.close()


def test_zlevel_reports_the_missing_local_zstar_fallback(built):
    """zlevel with min_hnode just below 1: the first falling step of any column asks for the reference's local-zstar fallback (oce_ale.F90:1859-1942), which is not
    built (the reference's own update_thickness_ale stops in a non-conformable PACK there under the compiler it is built with here): the library says so at the next
    synchronising call instead of stepping on silently."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.core import OceanCore
    from fesom2_amd.synthetic import analytic_ts, analytic_forcing
    mesh = Mesh.load(PI, dt=900.0, which_ale="zlevel")
    gpu = OceanCore(mesh, make_params(dt=900.0, which_ale="zlevel", min_hnode=0.999999))
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    gpu.upload_state(st)
    gpu.set_forcing(**analytic_forcing(mesh))
    gpu.run_steps(1, 3)
    with pytest.raises(RuntimeError, match="local-zstar"):
        gpu.sync()
    gpu.close()


def test_spp_chain_bitwise(built):
    """SPP = .true. (k_spp: cal_rejected_salt + app_rejected_salt at the head of solve_tracers_ale, linfs; oracle pinned on the reference run pi_pp_linfs_spp):
    HIP == oracle bit for bit after every routine of 3 steps and after 6 further whole steps.  (The reference leaves 0/0 in the salinity below the bottom of
    columns shallower than the plume depth; the comparison takes NaN == NaN there and is bitwise everywhere else.)"""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.core import OceanCore
    from fesom2_amd.synthetic import analytic_ts, analytic_forcing
    from oracle_lib import Oracle
    mesh = Mesh.load(PI, dt=900.0, which_ale="linfs", use_partial_cell=False)
    par = make_params(dt=900.0, which_ale="linfs", use_partial_cell=False, SPP=True)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    gpu, orc = OceanCore(mesh, par), Oracle(mesh, par)
    gpu.upload_state(st); orc.set_state(st)
    forcing = analytic_forcing(mesh)
    lon, lat = mesh.geo_coord_nod2D[:, 0], mesh.geo_coord_nod2D[:, 1]
    forcing["thdgr"] = 3.0e-7 * (np.abs(np.sin(lat)) - 0.7) * (1.0 + 0.3 * np.cos(2.0 * lon))
    forcing["S_oc_array"] = 33.0 + 1.5 * np.cos(lon)
    gpu.set_forcing(**forcing)
    for k, v in forcing.items():
        orc.set(k, v)
    chain = full_chain(2)
    i0 = [r for r, _, _ in chain].index("init_tracers_AB")
    chain = chain[:i0] + [("spp", 0, ["tr_arr"])] + chain[i0:]
    failures = []
    for step in range(1, 4):
        for routine, arg, fields in chain:
            if routine == "spp":
                S0 = orc.get("tr_arr").copy()
            gpu.call(routine, arg); orc.call(routine, arg)
            for f in fields:
                ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
                if not ok:
                    failures.append(f"step {step} {routine}({arg}) {msg}")
            if routine == "spp":
                d = np.nan_to_num(orc.get("tr_arr") - S0).reshape(2, -1, mesh.nl - 1)
                assert d[0].max() == 0.0 and d[0].min() == 0.0 and d[1].min() < -1e-5 and d[1].max() > 1e-6      # salt leaves the surface cell for the plume
                assert not (d[1][lat <= 0.0] != 0.0).any()                                                            # northern hemisphere only
        if failures:
            break
    assert not failures, "\n".join(failures[:10])
    gpu.run_steps(4, 6)
    for n in range(6):
        orc.call("step", 4 + n)
    for f in ("tr_arr", "UV", "eta_n", "hnode", "hbar", "Wvel"):
        ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
        assert ok, msg
    wet = (np.arange(mesh.nl - 1)[None, :] < (mesh.nlevels_nod2D[:, None] - 1))
    assert np.isfinite(gpu.get("tr_arr", orc.count("tr_arr")).reshape(2, -1, mesh.nl - 1)[:, wet]).all()       # the 0/0 stays below the bottom
    gpu.close()


def test_relax_to_clim_chain_bitwise(built):
    """clim_relax > 0 (k_relax_clim after the tracer update, the salinity clamp behind it; oracle pinned on the reference run pi_pp_climrelax): HIP == oracle bit
    for bit after every routine of 3 steps and after 6 further whole steps."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.core import OceanCore
    from fesom2_amd.synthetic import analytic_ts, analytic_forcing
    from oracle_lib import Oracle
    mesh = Mesh.load(PI, dt=900.0)
    par = make_params(dt=900.0, clim_relax=1.1574e-6)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    gpu, orc = OceanCore(mesh, par), Oracle(mesh, par)
    gpu.upload_state(st); orc.set_state(st)
    forcing = analytic_forcing(mesh)
    gpu.set_forcing(**forcing)
    for k, v in forcing.items():
        orc.set(k, v)
    lat = mesh.geo_coord_nod2D[:, 1]
    clim = {"Tclim": st.tr_arr[0] + 0.5, "Sclim": st.tr_arr[1] - 0.2, "relax2clim": 1.1574e-6 * np.maximum(0.0, 2.0 * np.sin(lat) ** 2 - 0.5)}
    for k, v in clim.items():
        gpu.set(k, v); orc.set(k, v)
    failures = []
    for step in range(1, 4):
        for routine, arg, fields in full_chain(2):
            gpu.call(routine, arg); orc.call(routine, arg)
            if routine == "diff_tracers_ale":
                gpu.call("relax_to_clim", arg); orc.call("relax_to_clim", arg)
            for f in fields:
                ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
                if not ok:
                    failures.append(f"step {step} {routine}({arg}) {msg}")
        if failures:
            break
    assert not failures, "\n".join(failures[:10])
    gpu.run_steps(4, 6)
    for n in range(6):
        orc.call("step", 4 + n)
    for f in ("tr_arr", "UV", "eta_n", "hnode", "hbar", "Wvel"):
        ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
        assert ok, msg
    T0 = st.tr_arr[0]
    assert np.abs(gpu.get("tr_arr", orc.count("tr_arr")).reshape(2, -1, mesh.nl - 1)[0] - T0).max() > 1e-3
    gpu.close()


def test_surface_potentials_chain_bitwise(built):
    """use_floatice + l_mslp + use_global_tides in the surface pressure gradient (surf_pre in k_vel_rhs; oracle pinned on the reference run pi_pp_surfpot):
    HIP == oracle bit for bit after every routine of 3 steps and after 6 further whole steps."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.core import OceanCore
    from fesom2_amd.synthetic import analytic_ts, analytic_forcing, analytic_surface_potentials
    from oracle_lib import Oracle
    mesh = Mesh.load(PI, dt=900.0)
    par = make_params(dt=900.0, use_floatice=True, l_mslp=True, use_global_tides=True)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    gpu, orc = OceanCore(mesh, par), Oracle(mesh, par)
    gpu.upload_state(st); orc.set_state(st)
    forcing = dict(analytic_forcing(mesh), **analytic_surface_potentials(mesh))
    gpu.set_forcing(**forcing)
    for k, v in forcing.items():
        orc.set(k, v)
    failures = []
    for step in range(1, 4):
        for routine, arg, fields in full_chain(2):
            gpu.call(routine, arg); orc.call(routine, arg)
            for f in fields:
                ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
                if not ok:
                    failures.append(f"step {step} {routine}({arg}) {msg}")
        if failures:
            break
    assert not failures, "\n".join(failures[:10])
    gpu.run_steps(4, 6)
    for n in range(6):
        orc.call("step", 4 + n)
    for f in ("tr_arr", "UV", "eta_n", "hnode", "hbar", "Wvel"):
        ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
        assert ok, msg
    gpu.close()


def test_biharmonic_tracer_filter_chain_bitwise(built):
    """smooth_bh_tra (diff_part_bh: k_bh1, k_bh2 after k_tr_update, the salinity clamp moved behind them as in the reference); oracle pinned on the reference run
    pi_pp_bhtra: HIP == oracle bit for bit after every routine of 3 steps and after 6 further whole steps."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.core import OceanCore
    from fesom2_amd.synthetic import analytic_ts, analytic_forcing
    from oracle_lib import Oracle
    mesh = Mesh.load(PI, dt=900.0)
    par = make_params(dt=900.0, smooth_bh_tra=True)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    gpu, orc = OceanCore(mesh, par), Oracle(mesh, par)
    gpu.upload_state(st); orc.set_state(st)
    forcing = analytic_forcing(mesh)
    gpu.set_forcing(**forcing)
    for k, v in forcing.items():
        orc.set(k, v)
    failures = []
    for step in range(1, 4):
        for routine, arg, fields in full_chain(2):
            gpu.call(routine, arg); orc.call(routine, arg)
            for f in fields:
                ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
                if not ok:
                    failures.append(f"step {step} {routine}({arg}) {msg}")
        if failures:
            break
    assert not failures, "\n".join(failures[:10])
    gpu.run_steps(4, 6)
    for n in range(6):
        orc.call("step", 4 + n)
    for f in ("tr_arr", "UV", "eta_n", "hnode", "hbar", "Wvel"):
        ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
        assert ok, msg
    gpu.close()


@pytest.mark.parametrize("akw", [dict(which_ale="linfs", use_partial_cell=False), dict()])
def test_vector_invariant_momentum_chain_bitwise(built, akw):
    """mom_adv = 3 (compute_vel_rhs_vinv: k_vinv_ke, k_leith_vort, k_vinv_elem) with the linear free surface and full cells; oracle pinned on the reference
    run pi_pp_linfs_vinv: HIP == oracle bit for bit after every routine of 3 steps and after 6 further whole steps.  The same with zstar (pi_pp_vinv:
    hpressure stays zero there in the reference)."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.core import OceanCore
    from fesom2_amd.synthetic import analytic_ts, analytic_forcing
    from oracle_lib import Oracle
    mesh = Mesh.load(PI, dt=900.0, **akw)
    par = make_params(dt=900.0, mom_adv=3, **akw)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    gpu, orc = OceanCore(mesh, par), Oracle(mesh, par)
    gpu.upload_state(st); orc.set_state(st)
    forcing = analytic_forcing(mesh)
    gpu.set_forcing(**forcing)
    for k, v in forcing.items():
        orc.set(k, v)
    failures = []
    for step in range(1, 4):
        for routine, arg, fields in full_chain(2):
            gpu.call(routine, arg); orc.call(routine, arg)
            for f in list(fields) + (["KE_node", "vorticity"] if routine == "compute_vel_rhs" else []):
                ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
                if not ok:
                    failures.append(f"step {step} {routine}({arg}) {msg}")
        if failures:
            break
    assert not failures, "\n".join(failures[:10])
    gpu.run_steps(4, 6)
    for n in range(6):
        orc.call("step", 4 + n)
    for f in ("tr_arr", "UV", "eta_n", "hnode", "hbar", "Wvel", "UV_rhsAB"):
        ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
        assert ok, msg
    gpu.close()


@pytest.mark.parametrize("kw", [dict(), dict(which_ale="linfs", use_partial_cell=True), dict(which_ale="linfs", use_partial_cell=True, which_pgf="nemo"), dict(which_pgf="easypgf"), dict(which_ale="linfs", use_partial_cell=True, which_pgf="easypgf")])
def test_cubicspline_pgf_chain_bitwise(built, kw):
    """which_pgf = 'cubicspline' (pressure_force_4_zxxxx_cubicspline with zstar, pressure_force_4_linfs_cubicspline with linfs + partial cells; oracle pinned on
    the reference runs pi_pp_cubicspline / pi_pp_linfs_cubic): HIP == oracle bit for bit after every routine of 3 steps under surface forcing."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.core import OceanCore
    from fesom2_amd.synthetic import analytic_ts, analytic_forcing
    from oracle_lib import Oracle
    mesh = Mesh.load(PI, dt=900.0, **{k: v for k, v in kw.items() if k != "which_pgf"})
    par = make_params(dt=900.0, **dict(dict(which_pgf="cubicspline"), **kw))             # (the third case: 'nemo', pinned on pi_pp_linfs_nemo)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    gpu, orc = OceanCore(mesh, par), Oracle(mesh, par)
    gpu.upload_state(st); orc.set_state(st)
    forcing = analytic_forcing(mesh)
    gpu.set_forcing(**forcing)
    for k, v in forcing.items():
        orc.set(k, v)
    failures = []
    for step in range(1, 4):
        for routine, arg, fields in full_chain(2):
            gpu.call(routine, arg); orc.call(routine, arg)
            for f in fields:
                ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
                if not ok:
                    failures.append(f"step {step} {routine}({arg}) {msg}")
        if failures:
            break
    assert not failures, "\n".join(failures[:10])
    gpu.close()


@pytest.mark.parametrize("kw", [dict(mix_scheme="PP"), dict(mix_scheme="KPP", Fer_GM=True, Redi=True)])
def test_monin_obukhov_mixing_chain(built, kw):
    """use_momix = .true. (the shipped config/namelist.oce; oracle pinned on the reference runs pi_pp_momix / pi_default_momix): k_momix (mo_length, pmlktmo)
    + the momix terms of the fused mixing kernels.  HIP == oracle bit for bit after every routine of 3 steps; the mixing length itself to 1e-13 relative
    (its Newton iteration calls exp: the device's libm against glibc), Kv / Av are compared bitwise (they only see it through |zbar| <= mixlength)."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.core import OceanCore
    from fesom2_amd.synthetic import analytic_ts, analytic_forcing, analytic_ice
    from oracle_lib import Oracle
    kpp = kw["mix_scheme"] == "KPP"
    mesh = Mesh.load(PI, dt=900.0)
    par = make_params(dt=900.0, use_momix=True, **kw)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    gpu, orc = OceanCore(mesh, par), Oracle(mesh, par)
    gpu.upload_state(st); orc.set_state(st)
    forcing = dict(analytic_forcing(mesh), **analytic_ice(mesh))
    gpu.set_forcing(**forcing)
    for k, v in forcing.items():
        orc.set(k, v)
    failures, deep = [], 0
    for step in range(1, 4):
        for routine, arg, fields in full_chain(2, gm=kpp, redi=kpp, kpp=kpp):
            gpu.call(routine, arg); orc.call(routine, arg)
            if routine == "compute_neutral_slope" and kpp:          # tanh: device libm vs glibc (see test_redi_chain_bitwise_and_steps)
                gpu.set("slope_tapered", orc.get("slope_tapered"))
            if routine == "mo_convect":                             # exp inside pmlktmo: compared to 1e-13 below, then the same bits on both sides
                ok, msg = compare("mixlength", gpu.get("mixlength", orc.count("mixlength")), orc.get("mixlength"))
                if not ok:
                    failures.append(f"step {step} {routine}({arg}) {msg}")
                gpu.set("mixlength", orc.get("mixlength"))
            for f in fields:
                ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
                if not ok:
                    failures.append(f"step {step} {routine}({arg}) {msg}")
        if failures:
            break
    assert not failures, "\n".join(failures[:10])
    ml = gpu.get("mixlength", orc.count("mixlength"))
    assert (ml >= 10.0).sum() > 500 and (ml == 0.0).sum() > 1000          # applied south of 50 S only
    gpu.close()


@pytest.mark.parametrize("field,value,msg", [("visc_option", 0, "visc_option"), ("visc_option", 9, "visc_option"), ("which_pgf", -1, "which_pgf"), ("tra_adv_ver", 4, "tra_adv_ver"),
                                             ("tra_adv_ver", -1, "tra_adv_ver"), ("tra_adv_hor", 3, "tra_adv_hor"), ("mom_adv", 1, "mom_adv"),
                                             ("mix_scheme", 3, "mix_scheme")])
def test_init_refuses_options_it_does_not_implement(built, field, value, msg):
    """fesom_gpu_init fails with a message naming the option instead of silently running something else (the Fortran layer maps a
    namelist value it does not know to -1); the reference-side caller keeps its own routine for those runs."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.core import OceanCore
    mesh = Mesh.load(PI, dt=900.0)
    par = make_params(dt=900.0)
    setattr(par, field, value)
    with pytest.raises(RuntimeError, match=msg):
        OceanCore(mesh, par)
    par = make_params(dt=900.0)                      # and the library is usable afterwards
    gpu = OceanCore(mesh, par)
    gpu.close()


@pytest.mark.parametrize("mix", ["PP", "KPP"])
def test_kv0_background_steps(built, mix):
    """Kv0_const=.false. (Kv0_background_qiang): the only difference between the HIP path and the oracle is atan (device libm vs glibc,
    <= 1 ulp of a 1e-5 m2/s diffusivity), so 8 free-running steps agree to 1e-10 relative in Kv (the shear-dependent part of the scheme
    amplifies the last-bit difference of the background: 3e-11 observed) and 1e-9 absolute in the state
    (tolerance stated here; everything else on the path stays bitwise, see the other tests)."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.core import OceanCore
    from fesom2_amd.synthetic import analytic_ts, analytic_forcing
    from oracle_lib import Oracle
    mesh = Mesh.load(PI, dt=900.0)
    par = make_params(dt=900.0, mix_scheme=mix, Kv0_const=False)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    gpu, orc = OceanCore(mesh, par), Oracle(mesh, par)
    gpu.upload_state(st); orc.set_state(st)
    forcing = analytic_forcing(mesh)
    gpu.set_forcing(**forcing)
    for k, v in forcing.items():
        orc.set(k, v)
    gpu.run_steps(1, 8)
    for n in range(1, 9):
        orc.call("step", n)
    kg, ko = gpu.get("Kv", orc.count("Kv")), orc.get("Kv")
    assert np.abs(kg - ko).max() <= 1e-10 * np.abs(ko).max()
    assert np.abs(ko).max() > 2e-5 and len(np.unique(np.round(ko[ko > 0], 12))) > 50        # the background really varies
    for f in ("tr_arr", "UV", "eta_n", "hnode"):
        a, b = gpu.get(f, orc.count(f)), orc.get(f)
        assert np.isfinite(a).all() and np.abs(a - b).max() < 1e-9, f
    gpu.close()


@pytest.mark.parametrize("shape,kw", [(1, dict()), (3, dict()), (1, dict(mix_scheme="KPP", Fer_GM=True, Redi=True)), (2, dict(which_ale="linfs", use_partial_cell=True)),
                                      (1, dict(use_cavity=True)), (3, dict(use_cavity=True, mix_scheme="KPP", Fer_GM=True, Redi=True))])
def test_tile_shapes_chain_and_steps_bitwise(built, shape, kw):
    """The kernel shapes of CORE2-class meshes (DM::use_tile: k_tr_update / k_impl_visc tiles, k_edge_transport_tile, k_pgf_tile, k_flux_hor with
    fill_up_dn_grad on the fly), forced on pi with FESOM_GPU_TILE=<shape>: HIP == oracle bit for bit after every routine of 2 steps and after 6
    further whole steps (the channel tests run them at size; this one pins every routine)."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.core import OceanCore
    from fesom2_amd.synthetic import analytic_ts, analytic_forcing
    from oracle_lib import Oracle
    kpp = kw.get("mix_scheme") == "KPP"
    mkw = {k: v for k, v in kw.items() if k in ("which_ale", "use_partial_cell", "use_cavity")}
    D = os.path.join(os.path.dirname(PI), "pi_cavity") if kw.get("use_cavity") else PI      # (cavities: the shapes with upper levels > 1)
    mesh = Mesh.load(D, dt=900.0, **mkw)
    par = make_params(dt=900.0, **kw)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(D)
    st.tr_arr_old[...] = st.tr_arr
    old = os.environ.get("FESOM_GPU_TILE")
    os.environ["FESOM_GPU_TILE"] = str(shape)
    try:
        gpu = OceanCore(mesh, par)
    finally:
        if old is None:
            os.environ.pop("FESOM_GPU_TILE", None)
        else:
            os.environ["FESOM_GPU_TILE"] = old
    assert gpu.tile_shape == shape
    orc = Oracle(mesh, par)
    gpu.upload_state(st); orc.set_state(st)
    forcing = analytic_forcing(mesh)
    gpu.set_forcing(**forcing)
    for k, v in forcing.items():
        orc.set(k, v)
    failures = []
    for step in range(1, 3):
        for routine, arg, fields in full_chain(2, gm=kpp, redi=kpp, kpp=kpp):
            gpu.call(routine, arg); orc.call(routine, arg)
            if routine == "compute_neutral_slope" and kpp:
                gpu.set("slope_tapered", orc.get("slope_tapered"))
            for f in fields:
                ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
                if not ok:
                    failures.append(f"step {step} {routine}({arg}) {msg}")
        if failures:
            break
    assert not failures, "\n".join(failures[:10])
    if not kpp:                       # (with Redi the free-running steps differ by the tanh of the slopes: covered by test_redi_chain_bitwise_and_steps)
        gpu.run_steps(3, 6)
        for n in range(6):
            orc.call("step", 3 + n)
        for f in ("tr_arr", "UV", "eta_n", "hnode", "hbar", "Wvel"):
            ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
            assert ok, msg
    gpu.close()


@pytest.mark.parametrize("kw", [dict(), dict(mix_scheme="KPP", Fer_GM=True, Redi=True)], ids=["pp", "kpp_gm_redi"])
def test_cavity_chain_bitwise(built, kw):
    """Ice-shelf cavities (use_cavity: upper levels ulevels > 1 from the cavity files of the mesh, the reference density profile of init_ref_density, the
    interface-water fill and the cavity branch of hpressure in pressure_bv, every kernel from its column's upper level): the pi mesh with a synthetic draft
    (tests/golden/make_cavity_mesh.py), oracle pinned on the reference runs pi_pp_cavity / pi_default_cavity (tests/test_oracle_vs_reference.py): HIP == oracle
    bit for bit after every routine of 3 steps under surface forcing and after 8 further whole steps; with GM/Redi also what init_Redi_GM leaves at the rim of
    the draft (kernels_gm.hip:k_gm_coef)."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.core import OceanCore
    from fesom2_amd.synthetic import analytic_ts, analytic_forcing
    from oracle_lib import Oracle
    D = os.path.join(os.path.dirname(PI), "pi_cavity")
    mesh = Mesh.load(D, dt=900.0, use_cavity=True)
    assert (mesh.ulevels_nod2D > 1).sum() > 300 and mesh.ulevels.max() > 10
    par = make_params(dt=900.0, use_cavity=True, **kw)
    redi = bool(kw)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(D)
    st.tr_arr_old[...] = st.tr_arr
    gpu, orc = OceanCore(mesh, par), Oracle(mesh, par)
    gpu.upload_state(st); orc.set_state(st)
    forcing = analytic_forcing(mesh)
    gpu.set_forcing(**forcing)
    for k, v in forcing.items():
        orc.set(k, v)
    ok, msg = compare("density_ref", gpu.get("density_ref", orc.count("density_ref")), orc.get("density_ref"))
    assert ok, msg
    failures = []
    for step in range(1, 4):
        for routine, arg, fields in full_chain(2, gm=redi, redi=redi, kpp=redi):
            gpu.call(routine, arg); orc.call(routine, arg)
            for f in list(fields) + (["hpressure"] if routine == "pressure_bv" else []):
                ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
                if not ok:
                    failures.append(f"step {step} {routine}({arg}) {msg}")
            if routine == "compute_neutral_slope" and redi:           # tanh: device libm vs glibc (see test_redi_chain_bitwise_and_steps)
                gpu.set("slope_tapered", orc.get("slope_tapered"))
        if failures:
            break
    assert not failures, "\n".join(failures[:15])
    gpu.run_steps(4, 8)
    for n in range(4, 12):
        orc.call("step", n)
    for f in ("tr_arr", "UV", "eta_n", "hnode", "hbar", "Wvel"):
        a, b = gpu.get(f, orc.count(f)), orc.get(f)
        if redi:                       # (with Redi the free-running steps differ by the tanh of the slopes)
            err = np.abs(a - b).max() / np.abs(b).max()
            assert err < 1e-9, (f, err)
        else:
            ok, msg = compare(f, a, b)
            assert ok, msg
    gpu.close()


def test_density_ref_chain_bitwise(built):
    """use_density_ref=.true. without cavities (oracle pinned on the reference run pi_pp_dref): k_init_density_ref at the first state upload, k_pressure_bv against
    the profile: HIP == oracle bit for bit after every routine of 2 steps and after 6 further whole steps."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.core import OceanCore
    from fesom2_amd.synthetic import analytic_ts, analytic_forcing
    from oracle_lib import Oracle
    mesh = Mesh.load(PI, dt=900.0)
    par = make_params(dt=900.0, use_density_ref=True)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    gpu, orc = OceanCore(mesh, par), Oracle(mesh, par)
    gpu.upload_state(st); orc.set_state(st)
    forcing = analytic_forcing(mesh)
    gpu.set_forcing(**forcing)
    for k, v in forcing.items():
        orc.set(k, v)
    ok, msg = compare("density_ref", gpu.get("density_ref", orc.count("density_ref")), orc.get("density_ref"))
    assert ok, msg
    assert np.ptp(orc.get("density_ref")) > 1.0                    # (a profile, not density_0)
    failures = []
    for step in range(1, 3):
        for routine, arg, fields in full_chain(2):
            gpu.call(routine, arg); orc.call(routine, arg)
            for f in fields:
                ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
                if not ok:
                    failures.append(f"step {step} {routine}({arg}) {msg}")
    assert not failures, "\n".join(failures[:15])
    gpu.run_steps(3, 6)
    for n in range(3, 9):
        orc.call("step", n)
    for f in ("tr_arr", "UV", "eta_n", "hnode", "density_m_rho0"):
        ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
        assert ok, msg
    gpu.close()


CAVITY_VARIANTS = [
    ("pc", dict(use_cavity_partial_cell=True), dict()),
    ("easypgf", dict(), dict(which_pgf="easypgf")),
    ("cubicspline", dict(), dict(which_pgf="cubicspline")),
    ("zlevel", dict(which_ale="zlevel"), dict()),
    ("linfs", dict(which_ale="linfs", use_partial_cell=False), dict()),
    ("linfs_pc", dict(which_ale="linfs"), dict()),
    ("linfs_easypgf", dict(which_ale="linfs"), dict(which_pgf="easypgf")),
    ("linfs_cubic", dict(which_ale="linfs"), dict(which_pgf="cubicspline")),
    ("linfs_nemo", dict(which_ale="linfs"), dict(which_pgf="nemo")),
    ("linfs_sergey", dict(which_ale="linfs", use_cavity_partial_cell=True), dict(which_pgf="sergey")),
    ("linfs_cavpc_shch", dict(which_ale="linfs", use_cavity_partial_cell=True), dict()),
    ("linfs_cavpc_shch_tile", dict(which_ale="linfs", use_cavity_partial_cell=True), dict()),
]


@pytest.mark.parametrize("name,mkw,kw", CAVITY_VARIANTS, ids=[c[0] for c in CAVITY_VARIANTS])
def test_cavity_variants_chain_bitwise(built, name, mkw, kw):
    """The cavity variants the reference offers, each pinned on its own reference run in tests/test_oracle_vs_reference.py::CAVITY_CASES: partial cells at the
    shelf base, zlevel, linfs, every pressure-gradient scheme under a shelf incl. 'sergey' (pressure_force_4_linfs_cavity): HIP == oracle bit for bit after
    every routine of 2 steps and after 4 further whole steps (`_tile`: the CORE2-class kernel shapes forced on)."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.core import OceanCore
    from fesom2_amd.synthetic import analytic_ts, analytic_forcing
    from oracle_lib import Oracle
    D = os.path.join(os.path.dirname(PI), "pi_cavity")
    mesh = Mesh.load(D, dt=900.0, use_cavity=True, **mkw)
    par = make_params(dt=900.0, use_cavity=True, **mkw, **kw)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(D)
    st.tr_arr_old[...] = st.tr_arr
    old = os.environ.get("FESOM_GPU_TILE")
    if name.endswith("_tile"):
        os.environ["FESOM_GPU_TILE"] = "1"
    try:
        gpu = OceanCore(mesh, par)
    finally:
        if name.endswith("_tile"):
            if old is None:
                os.environ.pop("FESOM_GPU_TILE", None)
            else:
                os.environ["FESOM_GPU_TILE"] = old
    orc = Oracle(mesh, par)
    gpu.upload_state(st); orc.set_state(st)
    forcing = analytic_forcing(mesh)
    gpu.set_forcing(**forcing)
    for k, v in forcing.items():
        orc.set(k, v)
    failures = []
    for step in range(1, 3):
        for routine, arg, fields in full_chain(2):
            gpu.call(routine, arg); orc.call(routine, arg)
            for f in list(fields) + (["hpressure"] if routine == "pressure_bv" else []):
                ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
                if not ok:
                    failures.append(f"step {step} {routine}({arg}) {msg}")
        if failures:
            break
    assert not failures, "\n".join(failures[:15])
    gpu.run_steps(3, 4)
    for n in range(3, 7):
        orc.call("step", n)
    for f in ("tr_arr", "UV", "eta_n", "hnode", "hbar", "Wvel"):
        ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
        assert ok, msg
    gpu.close()


def test_scaling_rossby_chain(built):
    """scaling_Rossby=.true. (oracle pinned on the reference run pi_default_rossby): the Fermi cut-off of K_GM in k_gm_coef goes through exp -- device libm against
    glibc -- so fer_K / Ki / fer_gamma and what follows are compared to 1e-12 relative after init_Redi_GM (the oracle's fer_K, Ki and tapered slopes are then handed
    over); every other routine of 2 steps bit for bit; 6 free-running steps to 1e-9."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.core import OceanCore
    from fesom2_amd.synthetic import analytic_ts, analytic_forcing
    from oracle_lib import Oracle
    mesh = Mesh.load(PI, dt=900.0)
    par = make_params(dt=900.0, mix_scheme="KPP", Fer_GM=True, Redi=True, scaling_Rossby=True)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    gpu, orc = OceanCore(mesh, par), Oracle(mesh, par)
    gpu.upload_state(st); orc.set_state(st)
    forcing = analytic_forcing(mesh)
    gpu.set_forcing(**forcing)
    for k, v in forcing.items():
        orc.set(k, v)
    failures = []
    for step in range(1, 3):
        for routine, arg, fields in full_chain(2, gm=True, redi=True, kpp=True):
            gpu.call(routine, arg); orc.call(routine, arg)
            if routine == "compute_neutral_slope":
                gpu.set("slope_tapered", orc.get("slope_tapered"))
            if routine == "init_Redi_GM":
                for f in ("fer_K", "Ki"):
                    a, b = gpu.get(f, orc.count(f)), orc.get(f)
                    err = np.abs(a - b).max() / np.abs(b).max()
                    assert err < 1e-12, (f, err)
                    gpu.set(f, b)
                assert orc.get("fer_K").min() < 0.5 * orc.get("fer_K").max()
                continue
            for f in fields:
                ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
                if not ok:
                    failures.append(f"step {step} {routine}({arg}) {msg}")
        if failures:
            break
    assert not failures, "\n".join(failures[:10])
    gpu.run_steps(3, 6)
    for n in range(3, 9):
        orc.call("step", n)
    for f in ("tr_arr", "UV", "eta_n", "hnode", "fer_K"):
        a, b = gpu.get(f, orc.count(f)), orc.get(f)
        err = np.abs(a - b).max() / np.abs(b).max()
        assert err < 1e-9, (f, err)
    gpu.close()
