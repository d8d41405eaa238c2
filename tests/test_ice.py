"""Sea-ice mEVP rheology (row f-4 of SURVEY 8: src/ice_maEVP.F90:273-602, EVPdynamics_m).
CPU: the oracle's restatement (oracle/c/orc_ice.c) against the REFERENCE's own routine -- golden vectors from a 1-rank run of
oracle/_ref on the pi mesh with an analytic ice state (tests/golden/make_ice_golden.py): bit for bit after one and after two calls
of 120 subcycles.  GPU: the HIP kernels through the C ABI against the oracle and against the golden vectors, bit for bit."""
import ctypes as C
import os
import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PI = os.path.join(REPO, "tests", "golden", "meshes", "pi")
STATE = ("u_ice", "v_ice", "a_ice", "m_ice", "m_snow", "elevation", "u_w", "v_w", "stress_atmice_x", "stress_atmice_y", "sigma11", "sigma12", "sigma22")
OUT = ("u_ice", "v_ice", "sigma11", "sigma12", "sigma22")


def gold():
    return np.load(os.path.join(REPO, "tests", "golden", "ice_evp_reference.npz"))


def setup(g, **kw):
    from fesom2_amd.mesh import Mesh
    from fesom2_amd import ice
    mesh = Mesh.load(PI, dt=900.0)
    pv = g["in/ice_params"]
    par = ice.ice_params(ice_dt=pv[0], ellipse=pv[1], alpha_evp=pv[2], beta_evp=pv[3], Pstar=pv[4], c_pressure=pv[5], delta_min=pv[6], cd_oce_ice=pv[7],
                         evp_rheol_steps=int(pv[8]), max_ice_loading=pv[9], **kw)
    fields = ice.IceFields(**{k: g["in/" + k] for k in STATE})
    return mesh, par, fields


def bits(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64).ravel(); b = np.ascontiguousarray(b, dtype=np.float64).ravel()
    return bool(((a.view(np.int64) == b.view(np.int64)) | ((a == 0) & (b == 0))).all())


def oracle_evp(mesh, par, fields):
    import oracle_lib
    oracle_lib.build()
    orc = C.CDLL(oracle_lib.ORC_LIB)
    assert orc.orc_ice_evp(mesh.desc_p, C.byref(par), C.byref(fields.desc)) == 0


def test_host_mesh_layer_supplies_the_ice_metrics(built):
    """metric_factor (with the reference's quirk: every entry = tan(latitude of the last element)/r_earth, oce_mesh.F90:2183) and
    coriolis_node of the host mesh layer equal the reference's arrays bit for bit; the coastal nodes formed from the boundary edges
    are the nodes with bc_index_nod2D = 0 (oce_mesh.F90:2404-2413)"""
    from fesom2_amd.mesh import Mesh
    g = gold()
    mesh = Mesh.load(PI, dt=900.0)
    d = mesh.desc_p.contents
    E, N = mesh.myDim_elem2D, mesh.myDim_nod2D
    assert bits(np.ctypeslib.as_array(d.metric_factor, shape=(E,)), g["in/metric_factor"][:E])
    assert bits(np.ctypeslib.as_array(d.coriolis_node, shape=(N,)), g["in/coriolis_node"])
    edges = np.ctypeslib.as_array(d.edges, shape=(mesh.myDim_edge2D, 2)); lst = np.ctypeslib.as_array(d.myList_edge2D, shape=(mesh.myDim_edge2D,))
    bc = np.ones(N); bc[edges[lst > d.edge2D_in].ravel() - 1] = 0
    assert np.array_equal(bc, g["in/bc_index_nod2D"])


def test_oracle_evp_equals_reference_bitwise(built):
    g = gold()
    mesh, par, fields = setup(g)
    assert (fields["a_ice"] >= 0.01).sum() > 2000 and (fields["a_ice"] < 0.01).sum() > 100        # ice-covered and ice-free regions
    for n in (1, 2):
        oracle_evp(mesh, par, fields)
        for k in OUT:
            assert bits(fields[k], g[f"out{n}/{k}"]), (n, k, float(np.abs(fields[k] - g[f"out{n}/{k}"]).max()))
    assert np.abs(fields["u_ice"] - g["in/u_ice"]).max() > 1e-3 and np.abs(fields["sigma11"]).max() > 1e3


def test_oracle_evp_floating_ice_variant_runs(built):
    """use_floatice (ice + snow load in the sea-surface slope term, ice_maEVP.F90:159-185): no reference run pins it (parity unpinned);
    it changes the answer and stays finite"""
    g = gold()
    mesh, par, fields = setup(g, use_floatice=True)
    oracle_evp(mesh, par, fields)
    assert np.isfinite(fields["u_ice"]).all() and not bits(fields["u_ice"], g["out1/u_ice"])


@pytest.mark.gpu
@pytest.mark.parametrize("floatice", [False, True])
def test_gpu_evp_equals_oracle_and_reference_bitwise(built, floatice):
    from fesom2_amd import ice
    g = gold()
    mesh, par, fo = setup(g, use_floatice=floatice)
    _, _, fg = setup(g, use_floatice=floatice)
    core = ice.IceCore(mesh, par)
    core.upload(fg)
    for n in (1, 2):
        core.evp(1); core.download(fg)
        oracle_evp(mesh, par, fo)
        for k in OUT:
            assert bits(fg[k], fo[k]), (n, k, float(np.abs(fg[k] - fo[k]).max()))
            if not floatice:
                assert bits(fg[k], g[f"out{n}/{k}"]), (n, k)
    ms = core.time_ms(3)
    assert 0.0 < ms < 50.0
    core.close()


@pytest.mark.gpu
@pytest.mark.parametrize("transport,adv", [("callback", False), ("builtin", False), ("callback", True), ("builtin", True), ("callback", "aevp"), ("builtin", "aevp"), ("callback", "evp0"), ("builtin", "evp0")])
def test_gpu_partitioned_evp_equals_reference(built, transport, adv):
    """2 ranks (the reference's dist_2 partition of pi, sharing the GPU): halo of (u_ice_aux, v_ice_aux) after every subcycle through the
    host-callback transport (gloo) or the library's built-in transport (shared-memory stand-in for librccl); every rank's u_ice, v_ice
    (owned + halo) and stresses equal the rank-local outputs of the reference's own 2-rank run bit for bit -- including the
    reference's partition-dependent metric_factor (oce_mesh.F90:2183).  adv: + the FCT advection of m_ice, a_ice, m_snow after the EVP call
    (fesom_gpu_ice_advect_partitioned: the reference's exchange_nod calls, three tracers per message) against the 2-rank run of the reference's
    own ice_fct routines (tests/golden/ice_adv_reference.npz), owned + halo nodes bit for bit."""
    import json, re, subprocess, sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", FESOM_GPU_DEVICE="0")
    if transport == "builtin":
        fake = os.path.join(REPO, "tests", "helpers", "libfake_rccl.so")
        assert os.path.exists(fake)
        env.update(FESOM_GPU_RCCL_LIB=fake, PART_TRANSPORT="rccl")
    if adv == "aevp":          # the adaptive EVP, EVPdynamics_a, against the reference's 2-rank run of it (tests/golden/ice_aevp_reference.npz: r2/...)
        env["ICE_AEVP"] = "1"; adv = False; port_off = 4
    elif adv == "evp0":        # the classic EVP, EVPdynamics (tests/golden/ice_evp0_reference.npz: r2/...)
        env["ICE_EVP0"] = "1"; adv = False; port_off = 6
    else:
        port_off = 0
    if adv:
        env["ICE_ADV"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(29830 + (1 if transport == "builtin" else 0) + (2 if adv else 0) + port_off), os.path.join(REPO, "tests", "helpers", "partitioned_ice_worker.py")],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    reps = [json.loads(x) for x in re.findall(r"ICEREPORT (\{.*\})", r.stdout)]
    assert len(reps) == 2
    for rep in reps:
        assert rep["bad"] == [], rep
        assert rep["changed"] > 1e-3


# ---- FCT advection of m_ice, a_ice, m_snow (src/ice_fct.F90: ice_TG_rhs_div, ice_fct_solve, ice_update_for_div; cut_off) --------------------
ADV = ("evp.u_ice", "evp.v_ice", "tg.rhs_m", "tg.rhs_a", "tg.rhs_ms", "tg.rhs_mdiv", "tg.rhs_adiv", "tg.rhs_msdiv", "fct.m_icel", "fct.a_icel", "fct.m_snowl",
       "fct.dm_ice", "fct.da_ice", "fct.dm_snow", "fct.m_ice", "fct.a_ice", "fct.m_snow", "div.m_ice", "div.a_ice", "div.m_snow")


def gold_adv():
    return np.load(os.path.join(REPO, "tests", "golden", "ice_adv_reference.npz"))


def oracle_adv(mesh, par, fields, gamma, want_dbg=False):
    import oracle_lib
    oracle_lib.build()
    orc = C.CDLL(oracle_lib.ORC_LIB)
    N = mesh.myDim_nod2D + mesh.eDim_nod2D
    dbg = [np.zeros(N) for _ in ADV]
    arr = (C.POINTER(C.c_double) * len(ADV))(*[a.ctypes.data_as(C.POINTER(C.c_double)) for a in dbg])
    orc.orc_ice_adv.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    assert orc.orc_ice_adv(C.cast(mesh.desc_p, C.c_void_p), C.c_double(par.ice_dt), C.c_double(gamma), C.cast(C.byref(fields.desc), C.c_void_p),
                           C.cast(arr, C.c_void_p) if want_dbg else None) == 0
    return dict(zip(ADV, dbg))


def test_host_mass_matrix_and_oracle_advection_equal_reference_bitwise(built):
    """Three ice steps (EVPdynamics_m, then the FCT advection) of the oracle against the reference's own routines on pi, one rank
    (tests/golden/make_ice_adv_golden.py): the Taylor-Galerkin right-hand sides, the high- and low-order solutions, the limited update, the
    divergence correction and the state after cut_off, every array bit for bit in each of the three steps."""
    g = gold_adv()
    mesh, par, fields = setup(g)
    gamma = float(g["in/ice_gamma_fct"][0])
    myN = mesh.myDim_nod2D
    for n in (1, 2, 3):
        oracle_evp(mesh, par, fields)
        dbg = oracle_adv(mesh, par, fields, gamma, want_dbg=True)
        for k in ADV:
            assert bits(dbg[k][:myN], g[f"adv{n}/{k}"][:myN]), (n, k, float(np.abs(dbg[k][:myN] - g[f"adv{n}/{k}"][:myN]).max()))
        for k in ("u_ice", "v_ice", "a_ice", "m_ice", "m_snow", "sigma11", "sigma12", "sigma22"):
            assert bits(fields[k], g[f"out{n}/{k}"]), (n, k, float(np.abs(fields[k] - g[f"out{n}/{k}"]).max()))
    assert np.abs(fields["m_ice"] - g["in/m_ice"]).max() > 0.05                  # the ice has moved
    assert fields["a_ice"].max() <= 1.0 and fields["m_ice"].min() >= 0.0        # cut_off


@pytest.mark.gpu
def test_gpu_ice_step_equals_oracle_and_reference_bitwise(built):
    """Three ice steps on the GPU (fesom_gpu_ice_evp + fesom_gpu_ice_advect on the device-resident state; the pressure factor follows the advected
    concentration) against the oracle and against the reference's own run: u_ice, v_ice, stresses, m_ice, a_ice, m_snow bit for bit after every step."""
    from fesom2_amd import ice
    g = gold_adv()
    mesh, par, fo = setup(g, ice_gamma_fct=float(g["in/ice_gamma_fct"][0]))
    _, _, fg = setup(g)
    core = ice.IceCore(mesh, par)
    core.upload(fg)
    for n in (1, 2, 3):
        core.step(1); core.download(fg)
        oracle_evp(mesh, par, fo); oracle_adv(mesh, par, fo, par.ice_gamma_fct)
        for k in ("u_ice", "v_ice", "a_ice", "m_ice", "m_snow", "sigma11", "sigma12", "sigma22"):
            assert bits(fg[k], fo[k]), (n, k, float(np.abs(fg[k] - fo[k]).max()))
            assert bits(fg[k], g[f"out{n}/{k}"]), (n, k, float(np.abs(fg[k] - g[f"out{n}/{k}"]).max()))
    core.close()


# ---- adaptive EVP (whichEVP = 2, EVPdynamics_a, src/ice_maEVP.F90:785-888) ----
AOUT = OUT + ("alpha_evp_array", "beta_evp_array")


def gold_a():
    return np.load(os.path.join(REPO, "tests", "golden", "ice_aevp_reference.npz"))


def setup_a(g):
    from fesom2_amd.mesh import Mesh
    from fesom2_amd import ice
    mesh = Mesh.load(PI, dt=900.0)
    pv = g["in/ice_params"]
    par = ice.ice_params(ice_dt=pv[0], ellipse=pv[1], alpha_evp=pv[2], beta_evp=pv[3], Pstar=pv[4], c_pressure=pv[5], delta_min=pv[6], cd_oce_ice=pv[7],
                         evp_rheol_steps=int(pv[8]), max_ice_loading=pv[9], whichEVP=2, c_aevp=pv[10])
    fields = ice.IceFields(**{k: g["in/" + k] for k in STATE + ("alpha_evp_array", "beta_evp_array")})
    return mesh, par, fields


def test_oracle_aevp_equals_reference_bitwise(built):
    """the oracle's restatement of EVPdynamics_a (ssh2rhs, stress_tensor_a with the element's alpha, stress2rhs_m, node update with the node's beta,
    find_alpha_field_a, find_beta_field_a) against the reference's own routine: velocities, stresses, alpha and beta bit for bit after one, two and three
    calls of 120 subcycles (tests/golden/make_ice_aevp_golden.py)"""
    import oracle_lib
    g = gold_a()
    mesh, par, fields = setup_a(g)
    oracle_lib.build()
    orc = C.CDLL(oracle_lib.ORC_LIB)
    assert np.all(fields["alpha_evp_array"] == par.alpha_evp) and np.all(fields["beta_evp_array"] == par.alpha_evp)
    for n in (1, 2, 3):
        assert orc.orc_ice_evp_a(mesh.desc_p, C.byref(par), C.byref(fields.desc)) == 0
        for k in AOUT:
            assert bits(fields[k], g[f"out{n}/{k}"]), (n, k, float(np.abs(fields[k] - g[f"out{n}/{k}"]).max()))
    a = fields["alpha_evp_array"]
    assert a.min() == 50.0 and (a > 50.0).sum() > 100 and np.abs(fields["u_ice"] - g["in/u_ice"]).max() > 1e-3      # (the adaptive alpha acts)


@pytest.mark.gpu
def test_gpu_aevp_equals_oracle_and_reference_bitwise(built):
    """adaptive EVP on the device (k_ice_a_prep / k_ice_a_stress / k_ice_a_node per subcycle, k_ice_a_alpha / k_ice_a_beta after the call): velocities,
    stresses, alpha_evp_array and beta_evp_array equal the oracle's and the reference's own EVPdynamics_a bit for bit after one, two and three calls;
    the state (stresses, alpha, beta) stays on the device between the calls"""
    from fesom2_amd import ice
    import oracle_lib
    g = gold_a()
    mesh, par, fo = setup_a(g)
    _, _, fg = setup_a(g)
    oracle_lib.build()
    orc = C.CDLL(oracle_lib.ORC_LIB)
    core = ice.IceCore(mesh, par)
    core.upload(fg)
    for n in (1, 2, 3):
        core.evp(1); core.download(fg)
        assert orc.orc_ice_evp_a(mesh.desc_p, C.byref(par), C.byref(fo.desc)) == 0
        for k in AOUT:
            assert bits(fg[k], fo[k]), (n, k, float(np.abs(fg[k] - fo[k]).max()))
            assert bits(fg[k], g[f"out{n}/{k}"]), (n, k)
    ms = core.time_ms(3)
    assert 0.0 < ms < 100.0
    core.close()


# ---- classic EVP (whichEVP = 0, the default of namelist.ice: EVPdynamics, src/ice_EVP.F90:397-667) ----
def gold_0():
    return np.load(os.path.join(REPO, "tests", "golden", "ice_evp0_reference.npz"))


def setup_0(g):
    from fesom2_amd.mesh import Mesh
    from fesom2_amd import ice
    mesh = Mesh.load(PI, dt=900.0)
    pv = g["in/ice_params"]
    par = ice.ice_params(ice_dt=pv[0], ellipse=pv[1], alpha_evp=pv[2], beta_evp=pv[3], Pstar=pv[4], c_pressure=pv[5], delta_min=pv[6], cd_oce_ice=pv[7],
                         evp_rheol_steps=int(pv[8]), max_ice_loading=pv[9], whichEVP=0, theta_io=pv[11], Tevp_inv=pv[12])
    fields = ice.IceFields(**{k: g["in/" + k] for k in STATE})
    return mesh, par, fields


def test_oracle_evp0_equals_reference_bitwise(built):
    """the oracle's restatement of the classic EVP (ice strength and sea-surface-slope term, stress_tensor, stress2rhs, node update in place) against the
    reference's own EVPdynamics: velocities and stresses bit for bit after one and after two calls of 120 subcycles (tests/golden/make_ice_evp0_golden.py)"""
    import oracle_lib
    g = gold_0()
    mesh, par, fields = setup_0(g)
    assert par.Tevp_inv == 3.0 / par.ice_dt
    oracle_lib.build()
    orc = C.CDLL(oracle_lib.ORC_LIB)
    for n in (1, 2):
        assert orc.orc_ice_evp0(mesh.desc_p, C.byref(par), C.byref(fields.desc)) == 0
        for k in OUT:
            assert bits(fields[k], g[f"out{n}/{k}"]), (n, k, float(np.abs(fields[k] - g[f"out{n}/{k}"]).max()))
    assert np.abs(fields["u_ice"] - g["in/u_ice"]).max() > 1e-3 and np.abs(fields["sigma11"]).max() > 1e2
    assert not bits(fields["u_ice"], gold()["out2/u_ice"])                 # (not the mEVP answer)


@pytest.mark.gpu
def test_gpu_evp0_equals_oracle_and_reference_bitwise(built):
    """the classic EVP on the device (k_ice_c_prep_node / _elem once per call, k_ice_c_stress + k_ice_c_node per subcycle, velocities in place): velocities and
    stresses equal the oracle's and the reference's own EVPdynamics bit for bit after one and after two calls"""
    from fesom2_amd import ice
    import oracle_lib
    g = gold_0()
    mesh, par, fo = setup_0(g)
    _, _, fg = setup_0(g)
    oracle_lib.build()
    orc = C.CDLL(oracle_lib.ORC_LIB)
    core = ice.IceCore(mesh, par)
    core.upload(fg)
    for n in (1, 2):
        core.evp(1); core.download(fg)
        assert orc.orc_ice_evp0(mesh.desc_p, C.byref(par), C.byref(fo.desc)) == 0
        for k in OUT:
            assert bits(fg[k], fo[k]), (n, k, float(np.abs(fg[k] - fo[k]).max()))
            assert bits(fg[k], g[f"out{n}/{k}"]), (n, k)
    ms = core.time_ms(3)
    assert 0.0 < ms < 100.0
    core.close()


@pytest.mark.gpu
@pytest.mark.parametrize("which", [0, 2])
def test_gpu_ice_step_other_rheologies_equals_oracle_bitwise(built, which):
    """Whole ice steps (rheology + FCT advection on the device-resident state, the pressure factor following the advected concentration) with the classic and the
    adaptive EVP: u_ice, v_ice, stresses, m_ice, a_ice, m_snow (and alpha / beta) equal the oracle's sequence orc_ice_evp0 / orc_ice_evp_a + orc_ice_adv bit for bit
    after each of three steps.  (Each routine is pinned on the reference by itself; the reference run of this combination exists for mEVP: the test above.)"""
    from fesom2_amd import ice
    import oracle_lib
    g = gold_0() if which == 0 else gold_a()
    mesh, par, fo = (setup_0 if which == 0 else setup_a)(g)
    _, _, fg = (setup_0 if which == 0 else setup_a)(g)
    oracle_lib.build()
    orc = C.CDLL(oracle_lib.ORC_LIB)
    core = ice.IceCore(mesh, par)
    core.upload(fg)
    names = ("u_ice", "v_ice", "a_ice", "m_ice", "m_snow", "sigma11", "sigma12", "sigma22") + (("alpha_evp_array", "beta_evp_array") if which == 2 else ())
    for n in (1, 2, 3):
        core.step(1); core.download(fg)
        assert (orc.orc_ice_evp0 if which == 0 else orc.orc_ice_evp_a)(mesh.desc_p, C.byref(par), C.byref(fo.desc)) == 0
        oracle_adv(mesh, par, fo, par.ice_gamma_fct)
        for k in names:
            assert bits(fg[k], fo[k]), (n, k, float(np.abs(fg[k] - fo[k]).max()))
    assert np.abs(fg["m_ice"] - g["in/m_ice"]).max() > 1e-4
    core.close()
