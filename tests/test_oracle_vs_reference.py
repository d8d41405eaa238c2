"""CPU: the C oracle (oracle/c) and the host mesh layer against the REAL reference's outputs, via the committed
digests in tests/golden/pi_pp_reference.npz (made by tests/golden/make_goldens.py from oracle/_ref, the reference's own
Fortran/pARMS sources compiled with amdflang).  Bar: bit-exact on every sampled value of every routine output over
3 steps of the pi mesh; the SSH solve (pARMS RAS+ILU in the reference, Jacobi-BiCGstab here) is compared to the
solver tolerance and the reference's d_eta is then injected so that every later routine sees identical inputs."""
import os
import numpy as np
import pytest
from golden_util import gold, check_digest, wet_masks

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PI = os.path.join(REPO, "tests", "golden", "meshes", "pi")


@pytest.fixture(scope="module")
def env(built):
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.synthetic import analytic_ts
    from oracle_lib import Oracle
    mesh = Mesh.load(PI, dt=900.0)
    par = make_params(dt=900.0)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    orc = Oracle(mesh, par)
    orc.set_state(st)
    return mesh, par, st, orc


SETUP_FIELDS = ["coord_nod2D", "geo_coord_nod2D", "elem2D_nodes", "edges", "edge_tri", "elem_edges", "elem_area", "edge_dxdy",
                "edge_cross_dxdy", "elem_cos", "metric_factor", "elem_neighbors", "nod_in_elem2D_num", "depth", "gradient_vec",
                "gradient_sca", "zbar", "Z", "nlevels", "nlevels_nod2D", "nlevels_nod2D_min", "area", "area_inv", "areasvol",
                "areasvol_inv", "mesh_resolution", "coriolis", "coriolis_node", "edge_up_dn_tri", "bottom_elem_thickness",
                "bottom_node_thickness", "zbar_n_bot", "zbar_e_bot", "zbar_n_srf", "zbar_e_srf", "ssh_rowptr", "ssh_colind", "ssh_values"]


def test_mesh_layer_bitwise(env):
    """host mesh layer (csrc/mesh_host.cpp) == mesh_setup + ocean_setup of the reference, bit for bit"""
    mesh, par, st, orc = env
    g = gold()
    bad = []
    for f in SETUP_FIELDS:
        ok, msg = check_digest(getattr(mesh, f), g["setup/" + f])
        if not ok:
            bad.append(f"{f}: {msg}")
    for f in ("hnode", "helem", "zbar_3d_n", "Z_3d_n", "tr_arr", "eta_n", "hbar"):
        ok, msg = check_digest(getattr(st, f), g["setup/" + f])
        if not ok:
            bad.append(f"state {f}: {msg}")
    assert not bad, "\n".join(bad)


def test_oracle_chain_bitwise(env):
    mesh, par, st, orc = env
    from ref_chain import run_reference_chain
    bad = run_reference_chain(orc, mesh, gold(), steps=(1, 2, 3))
    assert not bad, "\n".join(bad[:20])


@pytest.mark.parametrize("redi", [False, True])
def test_oracle_chain_bitwise_gm(built, redi):
    """pi with the Gent-McWilliams bolus velocities (Fer_GM=.true., src/oce_fer_gm.F90): init_Redi_GM, fer_solve_Gamma,
    fer_gamma2vel, fer_Wvel of vert_vel_ale and the bolus velocities around the tracer loop, against a reference run; with
    `redi` also the isoneutral diffusion (rotated horizontal fluxes, explicit and implicit vertical parts,
    src/oce_ale_tracer.F90:398-1077)"""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.synthetic import analytic_ts
    from oracle_lib import Oracle
    from ref_chain import run_reference_chain
    mesh = Mesh.load(PI, dt=900.0)
    par = make_params(dt=900.0, Fer_GM=True, Redi=redi)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    orc = Oracle(mesh, par)
    orc.set_state(st)
    bad = run_reference_chain(orc, mesh, gold("pi_pp_gm_redi" if redi else "pi_pp_gm"), steps=(1, 2, 3))
    assert not bad, "\n".join(bad[:20])


FORCING = ("stress_atmoce_x", "stress_atmoce_y", "heat_flux", "water_flux", "stress_surf")


@pytest.mark.parametrize("cfg", ["pi_kpp", "pi_default", "pi_default_sw", "pi_kpp_nonlcl", "pi_kpp_nonlcl_linfs", "pi_kpp_dd"])
def test_oracle_chain_bitwise_kpp_forced(built, cfg):
    """KPP vertical mixing (src/oce_ale_mixing_kpp.F90: ri_iwmix, bldepth, wscale tables, blmix_kpp, enhance, smoothing of blmc)
    under the harness's analytic wind stress / heat / fresh-water forcing, which also pins the surface boundary terms of
    impl_vert_visc_ale, compute_ssh_rhs_ale, compute_hbar_ale and the tracer diffusion; `pi_default` = the reference's
    default physics (KPP + GM + Redi)."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.synthetic import analytic_ts
    from oracle_lib import Oracle
    from ref_chain import run_reference_chain
    full = cfg not in ("pi_kpp", "pi_kpp_nonlcl", "pi_kpp_nonlcl_linfs", "pi_kpp_dd")
    dd = cfg == "pi_kpp_dd"                     # + double diffusion (ddmix, oce_ale_mixing_kpp.F90:857-934)
    nonlcl = cfg.startswith("pi_kpp_nonlcl")    # + the non-local transport of heat and salt (use_kpp_nonlclflx, oce_ale_tracer.F90:688-724); with zstar the
    # reference's ocean_setup zeroes ref_sss (oce_setup_step.F90:42-47) and only the heat term acts, with linfs (+ full cells) both do
    akw = dict(which_ale="linfs", use_partial_cell=False) if cfg.endswith("linfs") else {}
    sw = cfg == "pi_default_sw"                 # + short-wave penetration (use_sw_pene=.true., the default of namelist.config)
    mesh = Mesh.load(PI, dt=900.0, **akw)
    par = make_params(dt=900.0, mix_scheme="KPP", Fer_GM=full, Redi=full, use_sw_pene=sw, use_kpp_nonlclflx=nonlcl, double_diffusion=dd, **akw)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    orc = Oracle(mesh, par)
    orc.set_state(st)
    g = gold(cfg)
    for f in FORCING:
        orc.set(f, g["forcing/" + f])
    if sw:
        from fesom2_amd.synthetic import analytic_sw_3d
        sw3 = analytic_sw_3d(mesh, g["forcing/heat_flux"])
        ok, msg = check_digest(sw3, g["forcing_digest/sw_3d"])     # the Python restatement of the harness's sw_3d: same bits
        assert ok, msg
        orc.set("sw_3d", sw3)
        orc.set("kpp_sw_node", g["part/last_owned_node"].astype(np.float64))      # rank-dependent quirk of the reference, see orc_kpp.c
    bad = run_reference_chain(orc, mesh, g, steps=(1, 2, 3))
    assert not bad, "\n".join(bad[:20])


STEP_INFO = ("sum_eta", "sum_hbar", "sum_deta", "sum_dhbar", "sum_wflux", "sum_area",
             "min_eta", "min_hbar", "min_wflux", "min_hflux", "min_temp", "min_salt", "min_wvel", "min_wvel2", "min_uvel", "min_uvel2",
             "min_vvel", "min_vvel2", "min_deta", "min_hnode", "min_hnode2",
             "max_eta", "max_hbar", "max_wflux", "max_hflux", "max_temp", "max_salt", "max_wvel", "max_wvel2", "max_uvel", "max_uvel2",
             "max_vvel", "max_vvel2", "max_deta", "max_hnode", "max_hnode2", "max_cfl_z", "max_pgfx", "max_pgfy", "max_av", "max_kv", "blowup")
TABLE = {"eta": "eta", "deta": "deta", "hbar": "hbar", "wflux": "wflux", "hflux": "hflux", "temp": "temp", "salt": "salt",
         "wvel(1,:)": "wvel", "wvel(2,:)": "wvel2", "uvel(1,:)": "uvel", "uvel(2,:)": "uvel2", "vvel(1,:)": "vvel", "vvel(2,:)": "vvel2",
         "hnode(1,:)": "hnode", "hnode(2,:)": "hnode2", "cfl_z": "cfl_z", "pgf_x": "pgfx", "pgf_y": "pgfy", "Av": "av", "Kv": "kv"}


def check_step_info(si, ref, int_tol):
    """si: dict of fesom_step_info fields (one partition = global); ref: one step of tests/golden/step_info_pi_default.json"""
    bad = []
    for k in ("eta", "hbar", "deta", "dhbar"):
        v = si["sum_" + k] / si["sum_area"]
        if abs(v - ref["int_" + k]) > int_tol:
            bad.append(f"int_{k}: {v!r} vs {ref['int_' + k]!r}")
    for k in ("min_eta", "max_eta"):
        if abs(si[k] - ref[k]) > 1e-9:
            bad.append(f"{k}: {si[k]!r} vs {ref[k]!r}")
    for name, key in TABLE.items():
        if name in ("pgf_x", "pgf_y", "Av"):      # the reference scans these ELEMENT arrays over each rank's first myDim_nod2D columns:
            continue                              # its printed value depends on the partition of the run (2 ranks here)
        lo, hi = ref["table_ES10.3"][name]
        for tag, r in (("min_", lo), ("max_", hi)):
            if r is None:
                continue
            v = si[tag + key]
            if abs(v - r) > 6e-4 * abs(r) + 1e-30:                   # the reference prints 4 digits (ES10.3)
                bad.append(f"{tag}{key}: {v!r} vs printed {r!r}")
    if si["blowup"] != 0.0:
        bad.append("blowup flag set")
    return bad


def test_oracle_step_info_vs_reference_printout(built):
    """write_step_info (the reference's step monitor, src/write_step_info.F90) restated in the oracle against the numbers the
    reference itself printed in a run of config pi_default: area-mean integrals to 1e-16 m absolute (the oracle state is
    bit-identical, only the 2-rank summation order differs; the terms are 1e4 times the mean), extrema to the 4 printed digits."""
    import json
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.synthetic import analytic_ts
    from oracle_lib import Oracle
    from ref_chain import run_reference_chain
    ref = json.load(open(os.path.join(REPO, "tests", "golden", "step_info_pi_default.json")))["steps"]
    mesh = Mesh.load(PI, dt=900.0)
    par = make_params(dt=900.0, mix_scheme="KPP", Fer_GM=True, Redi=True)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    orc = Oracle(mesh, par)
    orc.set_state(st)
    g = gold("pi_default")
    for f in FORCING:
        orc.set(f, g["forcing/" + f])
    bad = []

    def after(step):
        si = dict(zip(STEP_INFO, orc.step_info()))
        bad.extend(f"step {step}: {m}" for m in check_step_info(si, ref[str(step)], 1e-16))

    chain_bad = run_reference_chain(orc, mesh, g, steps=(1, 2, 3), after_step=after)
    assert not chain_bad and not bad, "\n".join((chain_bad + bad)[:20])


def test_oracle_chain_bitwise_w_split(built):
    """w_split=.true. (namelist.oce): vertical velocity split by CFL_z > w_max_cfl in vert_vel_ale (src/oce_ale.F90:2189-2203), implicit
    part in impl_vert_visc_ale and -- adv_tra_vert_impl, src/oce_adv_tra_ver.F90:83-227 -- in the low-order solution of the FCT
    advection; reference run with w_max_cfl = 0.0003 so that the split is active on pi, with surface forcing."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.synthetic import analytic_ts
    from oracle_lib import Oracle
    from ref_chain import run_reference_chain
    mesh = Mesh.load(PI, dt=900.0)
    par = make_params(dt=900.0, w_split=True, w_max_cfl=0.0003)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    orc = Oracle(mesh, par)
    orc.set_state(st)
    g = gold("pi_pp_wsplit")
    for f in FORCING:
        orc.set(f, g["forcing/" + f])
    bad = run_reference_chain(orc, mesh, g, steps=(1, 2, 3))
    assert not bad, "\n".join(bad[:20])
    wi = orc.get("Wvel_i")
    assert np.count_nonzero(wi) > 1000                     # the split is really active


def test_oracle_chain_bitwise_no_limiter(built):
    """tra_adv_lim = 'NON' (src/oce_adv_tra_driver.F90:137-197: high-order fluxes with init_zero=.true., vertical part with the explicit
    velocity, flux2dtracer without the low-order solution): reference run `pi_pp_non`, every routine of 3 steps bit for bit."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.synthetic import analytic_ts
    from oracle_lib import Oracle
    from ref_chain import run_reference_chain
    mesh = Mesh.load(PI, dt=900.0)
    par = make_params(dt=900.0, tra_adv_lim="NON")
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    orc = Oracle(mesh, par)
    orc.set_state(st)
    g = gold("pi_pp_non")
    for f in FORCING:
        orc.set(f, g["forcing/" + f])
    bad = run_reference_chain(orc, mesh, g, steps=(1, 2, 3))
    assert not bad, "\n".join(bad[:20])
    gf = gold("pi_pp_wsplit")                                # an FCT run: the limiter arrays are formed there, not here
    assert not np.array_equal(g["s2/tr1.adv.fct_plus"], gf["s2/tr1.adv.fct_plus"])


@pytest.mark.parametrize("cfg,kw", [("pi_pp_momix", dict(mix_scheme="PP")), ("pi_default_momix", dict(mix_scheme="KPP", Fer_GM=True, Redi=True))])
def test_oracle_chain_bitwise_monin_obukhov_mixing(built, cfg, kw):
    """use_momix = .true. (shipped config/namelist.oce:48): mo_length / pmlktmo (src/oce_mo_conv.F90:107-182) and the momix branches of mo_convect
    (:22-55, :95) south of 50 S, with the harness's analytic ice state: reference runs `pi_pp_momix` and `pi_default_momix` (= the shipped physics in
    full: KPP + GM + Redi + momix), every routine of 3 steps bit for bit incl. the mixing length carried from step to step."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.synthetic import analytic_ts
    from oracle_lib import Oracle
    from ref_chain import run_reference_chain
    mesh = Mesh.load(PI, dt=900.0)
    par = make_params(dt=900.0, use_momix=True, **kw)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    orc = Oracle(mesh, par)
    orc.set_state(st)
    g = gold(cfg)
    for f in FORCING + ("u_ice", "v_ice", "a_ice"):
        orc.set(f, g["forcing/" + f])
    bad = run_reference_chain(orc, mesh, g, steps=(1, 2, 3))
    assert not bad, "\n".join(bad[:20])
    g0 = gold("pi_pp_wsplit" if cfg == "pi_pp_momix" else "pi_default")
    assert not np.array_equal(g["s2/mixing.Kv"], g0["s2/mixing.Kv"]) and not np.array_equal(g["s2/mixing.Av"], g0["s2/mixing.Av"])


@pytest.mark.parametrize("cfg,akw", [("pi_pp_linfs_vinv", dict(which_ale="linfs", use_partial_cell=False)), ("pi_pp_vinv", dict())])
def test_oracle_chain_bitwise_vector_invariant_momentum(built, cfg, akw):
    """mom_adv = 3 (compute_vel_rhs_vinv, src/oce_vel_rhs_vinv.F90:104-322: kinetic energy at nodes, relative vorticity, gradient of the Bernoulli function) with
    the linear free surface and full cells, the only set-up in which the reference forms hpressure: reference run `pi_pp_linfs_vinv`, every routine of 3 steps
    bit for bit.  `pi_pp_vinv`: the same with zstar and partial cells, where the reference leaves hpressure at the zeros of array_setup
    (oce_setup_step.F90:384): compute_vel_rhs_vinv then runs without a baroclinic pressure term -- kept as the reference has it."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.synthetic import analytic_ts
    from oracle_lib import Oracle
    from ref_chain import run_reference_chain
    mesh = Mesh.load(PI, dt=900.0, **akw)
    par = make_params(dt=900.0, mom_adv=3, **akw)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    orc = Oracle(mesh, par)
    orc.set_state(st)
    g = gold(cfg)
    for f in FORCING:
        orc.set(f, g["forcing/" + f])
    bad = run_reference_chain(orc, mesh, g, steps=(1, 2, 3))
    assert not bad, "\n".join(bad[:20])


@pytest.mark.parametrize("cfg,kw", [("pi_pp_cubicspline", dict()), ("pi_pp_linfs_cubic", dict(which_ale="linfs", use_partial_cell=True)),
                                    ("pi_pp_linfs_nemo", dict(which_ale="linfs", use_partial_cell=True, which_pgf="nemo")), ("pi_pp_easypgf", dict(which_pgf="easypgf")),
                                    ("pi_pp_linfs_easypgf", dict(which_ale="linfs", use_partial_cell=True, which_pgf="easypgf"))])
def test_oracle_chain_bitwise_cubicspline_pgf(built, cfg, kw):
    """which_pgf = 'cubicspline': pressure_force_4_zxxxx_cubicspline (src/oce_ale_pressure_bv.F90:1697-1866, zstar) and pressure_force_4_linfs_cubicspline
    (:1252-1444, linfs with partial cells): reference runs `pi_pp_cubicspline`, `pi_pp_linfs_cubic`, every routine of 3 steps bit for bit."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.synthetic import analytic_ts
    from oracle_lib import Oracle
    from ref_chain import run_reference_chain
    mesh = Mesh.load(PI, dt=900.0, **{k: v for k, v in kw.items() if k != "which_pgf"})
    par = make_params(dt=900.0, **dict(dict(which_pgf="cubicspline"), **kw))         # (pi_pp_linfs_nemo: pressure_force_4_linfs_nemo, :479-635)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    orc = Oracle(mesh, par)
    orc.set_state(st)
    g = gold(cfg)
    for f in FORCING:
        orc.set(f, g["forcing/" + f])
    bad = run_reference_chain(orc, mesh, g, steps=(1, 2, 3))
    assert not bad, "\n".join(bad[:20])
    gz = gold("pi_pp_wsplit" if "which_ale" not in kw else "pi_pp_linfs_pc")      # the shchepetkin runs
    assert not np.array_equal(g["s2/pressure_force.pgf_x"], gz["s2/pressure_force.pgf_x"])


def test_oracle_chain_bitwise_relax_to_clim(built):
    """clim_relax > 0 (relax_to_clim, src/oce_tracer_mod.F90:86-121; Tclim / Sclim = the initial T / S as ocean_setup sets them, oce_setup_step.F90:480-481; the
    harness's analytic nodal rate relax2clim): reference run `pi_pp_climrelax`, every routine of 3 steps bit for bit."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.synthetic import analytic_ts
    from oracle_lib import Oracle
    from ref_chain import run_reference_chain
    mesh = Mesh.load(PI, dt=900.0)
    par = make_params(dt=900.0, clim_relax=1.1574e-6)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    orc = Oracle(mesh, par)
    orc.set_state(st)
    g = gold("pi_pp_climrelax")
    for f in FORCING + ("relax2clim",):
        orc.set(f, g["forcing/" + f])
    orc.set("Tclim", st.tr_arr[0]); orc.set("Sclim", st.tr_arr[1])
    bad = run_reference_chain(orc, mesh, g, steps=(1, 2, 3))
    assert not bad, "\n".join(bad[:20])
    gz = gold("pi_pp_wsplit")
    assert not np.array_equal(g["s3/tr1.end.tr_arr"], gz["s3/tr1.end.tr_arr"])


def test_oracle_chain_bitwise_zlevel(built):
    """which_ALE = 'zlevel' (init_thickness_ale :630-690, vert_vel_ale :1830-2023 incl. the "return to zlevel" branch that pi's 4-layer columns take on every rising
    step, update_thickness_ale :817-943): reference run `pi_pp_zlevel`, every routine of 3 steps bit for bit."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.synthetic import analytic_ts
    from oracle_lib import Oracle
    from ref_chain import run_reference_chain
    mesh = Mesh.load(PI, dt=900.0, which_ale="zlevel")
    par = make_params(dt=900.0, which_ale="zlevel")
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    orc = Oracle(mesh, par)
    orc.set_state(st)
    g = gold("pi_pp_zlevel")
    for f in FORCING:
        orc.set(f, g["forcing/" + f])
    bad = run_reference_chain(orc, mesh, g, steps=(1, 2, 3))
    assert not bad, "\n".join(bad[:20])
    gz = gold("pi_pp_wsplit")                                       # (the same set-up with zstar)
    assert not np.array_equal(g["s2/update_thickness_ale.hnode"], gz["s2/update_thickness_ale.hnode"])
    assert orc.lib.orc_get_ale_flag() == 0
    assert (mesh.nlevels_nod2D == 5).any()                          # columns whose partial bottom cell lies within the lzstar_lev surface layers


def test_oracle_chain_bitwise_spp(built):
    """SPP = .true. (salt plume parameterization, cal_rejected_salt + app_rejected_salt, src/oce_spp.F90, at the head of solve_tracers_ale; linfs as the routine's
    header asks; the harness's analytic ice growth rate thdgr and S_oc_array): reference run `pi_pp_linfs_spp`, every routine of 3 steps bit for bit."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.synthetic import analytic_ts
    from oracle_lib import Oracle
    from ref_chain import run_reference_chain
    mesh = Mesh.load(PI, dt=900.0, which_ale="linfs", use_partial_cell=False)
    par = make_params(dt=900.0, which_ale="linfs", use_partial_cell=False, SPP=True)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    orc = Oracle(mesh, par)
    orc.set_state(st)
    g = gold("pi_pp_linfs_spp")
    for f in FORCING + ("thdgr", "S_oc_array"):
        orc.set(f, g["forcing/" + f])
    bad = run_reference_chain(orc, mesh, g, steps=(1, 2, 3))
    assert not bad, "\n".join(bad[:20])
    gz = gold("pi_pp_linfs_vinv")                                   # (a run without SPP from the same initial salinity)
    assert not np.array_equal(g["s1/tr2.init_AB.tr_arr_old"], gz["s1/tr2.init_AB.tr_arr_old"])      # the plumes moved salt before the tracer loop
    assert (g["forcing/thdgr"] > 0).any() and (g["forcing/thdgr"] < 0).any()


def test_oracle_chain_bitwise_surface_potentials(built):
    """use_floatice (ice + snow load, limited by max_ice_loading), l_mslp (atmospheric pressure) and use_global_tides (tidal potential) in the surface pressure
    gradient of compute_vel_rhs (src/oce_ale_vel_rhs.F90:52-76), with the harness's analytic fields: reference run `pi_pp_surfpot`, every routine of 3 steps bit for bit."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.synthetic import analytic_ts
    from oracle_lib import Oracle
    from ref_chain import run_reference_chain
    mesh = Mesh.load(PI, dt=900.0)
    par = make_params(dt=900.0, use_floatice=True, l_mslp=True, use_global_tides=True)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    orc = Oracle(mesh, par)
    orc.set_state(st)
    g = gold("pi_pp_surfpot")
    for f in FORCING + ("m_ice", "m_snow", "press_air", "ssh_gp"):
        orc.set(f, g["forcing/" + f])
    assert (g["forcing/m_ice"] * 910. + g["forcing/m_snow"] * 290.).max() / 1025. > 5.0        # the max_ice_loading limit acts somewhere
    bad = run_reference_chain(orc, mesh, g, steps=(1, 2, 3))
    assert not bad, "\n".join(bad[:20])
    gz = gold("pi_pp_wsplit")
    assert not np.array_equal(g["s1/compute_vel_rhs.UV_rhs"], gz["s1/compute_vel_rhs.UV_rhs"])


def test_oracle_chain_bitwise_biharmonic_tracer_filter(built):
    """smooth_bh_tra = .true. (diff_part_bh, src/oce_ale_tracer.F90:1081-1150, at the end of diff_tracers_ale): reference run `pi_pp_bhtra`, every routine of 3
    routine of step 1 bit for bit.  The reference applies the filter with the halo values of the PREVIOUS exchange (the tracer is exchanged only after
    diff_tracers_ale, :1104-1118 read ttf at halo nodes), so in its 2-rank run -- where the goldens come from; a 1-rank run of the harness does not exist --
    the nodes within two rings of the partition boundary differ from a one-partition result (measured: 3e-10 in 6 of 262 samples).  They are left out of the
    tracer comparisons here, and only step 1 is compared (afterwards the difference spreads); the partition-faithful behaviour itself is pinned by the 2-rank
    drop-in run of the library against the reference's 2-rank run (tests/test_gpu_dropin.py[pi_pp_bhtra-2])."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.synthetic import analytic_ts
    from oracle_lib import Oracle
    from ref_chain import run_reference_chain
    mesh = Mesh.load(PI, dt=900.0)
    par = make_params(dt=900.0, smooth_bh_tra=True)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    orc = Oracle(mesh, par)
    orc.set_state(st)
    g = gold("pi_pp_bhtra")
    for f in FORCING:
        orc.set(f, g["forcing/" + f])
    # owner rank of every node in the reference's dist_2 partition, nodes within two rings of the other rank's nodes
    N = mesh.nod2D
    tok = np.array(open(os.path.join(PI, "dist_2", "rpart.out")).read().split(), dtype=np.int64)      # npes, owned counts, PE-contiguous index of every node
    assert tok[0] == 2 and tok.size == 3 + N
    rank = (tok[3:] > tok[1]).astype(np.int32)
    ed = np.asarray(mesh.edges).reshape(-1, 2) - 1
    near = np.zeros(N, dtype=bool)
    cut = rank[ed[:, 0]] != rank[ed[:, 1]]
    near[ed[cut].ravel()] = True
    for _ in range(2):
        grow = near[ed[:, 0]] | near[ed[:, 1]]
        near[ed[grow].ravel()] = True
    assert 100 < near.sum() < N // 2
    bad = run_reference_chain(orc, mesh, g, steps=(1,), node_keep=~near)
    assert not bad, "\n".join(bad[:20])
    gz = gold("pi_pp_wsplit")
    assert not np.array_equal(g["s1/tr1.end.tr_arr"], gz["s1/tr1.end.tr_arr"])


def test_oracle_chain_bitwise_linfs_partial_cells(built):
    """which_ALE = 'linfs' with use_partial_cell = .true. on pi: pressure_force_4_linfs_shchepetkin (src/oce_ale_pressure_bv.F90:647-891), the linfs branches
    of compute_hbar_ale / vert_vel_ale / the SSH right-hand side on a mesh with partial bottom cells: reference run `pi_pp_linfs_pc`, every routine of
    3 steps bit for bit."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.synthetic import analytic_ts
    from oracle_lib import Oracle
    from ref_chain import run_reference_chain
    mesh = Mesh.load(PI, dt=900.0, which_ale="linfs", use_partial_cell=True)
    par = make_params(dt=900.0, which_ale="linfs", use_partial_cell=True)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    orc = Oracle(mesh, par)
    orc.set_state(st)
    g = gold("pi_pp_linfs_pc")
    for f in FORCING:
        orc.set(f, g["forcing/" + f])
    bad = run_reference_chain(orc, mesh, g, steps=(1, 2, 3))
    assert not bad, "\n".join(bad[:20])
    gz = gold("pi_pp_wsplit")                                # the zstar run: a different pressure gradient
    assert not np.array_equal(g["s2/pressure_force.pgf_x"], gz["s2/pressure_force.pgf_x"])


@pytest.mark.parametrize("opt", [1, 2, 3, 4, 6, 7, 8])
def test_oracle_chain_bitwise_biharmonic_viscosity(built, opt):
    """visc_option = 1 / 2 / 3 (h_viscosity_leith :461-561 over relative_vorticity src/oce_vel_rhs_vinv.F90:14-102, then visc_filt_harmon :236-273,
    visc_filt_hbhmix :376-458, visc_filt_biharm(2) :275-372; the Leith coefficient and the vorticity are compared too), 4 (visc_filt_biharm(1), src/oce_dyn.F90:275-372), 6 (visc_filt_bilapl, :658-726) and 7 (visc_filt_bidiff, :734-801) instead
    of the easy backscatter: reference runs `pi_pp_visc4` / `pi_pp_visc6` / `pi_pp_visc7` (PP mixing, surface forcing), every routine of 3 steps bit for bit.
    8: backscatter_coef + visc_filt_dbcksc + uke_update (:806-1152) with the prognostic unresolved kinetic energy uke: run `pi_pp_visc8`; v_back, uke and
    uke_rhs are compared too (the result does not depend on the partition: smooth_elem only reads elements of owned nodes, which every rank forms itself)."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.synthetic import analytic_ts
    from oracle_lib import Oracle
    from ref_chain import run_reference_chain
    mesh = Mesh.load(PI, dt=900.0)
    par = make_params(dt=900.0, visc_option=opt)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    orc = Oracle(mesh, par)
    orc.set_state(st)
    g = gold(f"pi_pp_visc{opt}")
    for f in FORCING:
        orc.set(f, g["forcing/" + f])
    # (option 8 is pinned on a SINGLE-DOMAIN replay of the reference: on two ranks its edge loops add the halo edges of an element after the owned ones,
    #  which moves 0.05 % of UV_dis by one ulp against the single-domain order; on one rank, where pARMS cannot run, the harness solves the SSH system
    #  itself (oracle/ref/driver.F90:harness_solve_one_rank); 10 steps, so that the sub-grid energy feeds back into the momentum equation)
    bad = run_reference_chain(orc, mesh, g, steps=range(1, 11), check_steps=(1, 2, 3, 10)) if opt == 8 else run_reference_chain(orc, mesh, g, steps=(1, 2, 3))
    assert not bad, "\n".join(bad[:20])
    # the filter really differs from the default one: UV_rhs after viscosity_filter is not the backscatter result
    g5 = gold("pi_pp_wsplit")
    assert not np.array_equal(g["s2/viscosity_filter.UV_rhs"], g5["s2/viscosity_filter.UV_rhs"])


@pytest.mark.parametrize("ver,cfg,kw", [("CDIFF", "pi_pp_cdiff", {}), ("UPW1", "pi_pp_upw1v", dict(w_split=True, w_max_cfl=0.0003)),
                                        ("CDIFF", "pi_pp_upw1h", dict(tra_adv_hor="UPW1")), ("PPM", "pi_pp_ppm", {})])
def test_oracle_chain_bitwise_vertical_advection_variants(built, ver, cfg, kw):
    """tra_adv_ver = 'CDIFF' (adv_tra_ver_cdiff, src/oce_adv_tra_ver.F90:542-590), 'PPM' (adv_tra_vert_ppm :361-538) and 'UPW1' (:231-282, with w_split) as
    the high-order vertical scheme under FCT, tra_adv_hor = 'UPW1' (src/oce_adv_tra_hor.F90:57-211) as the horizontal one: reference runs
    `pi_pp_cdiff`, `pi_pp_upw1v`, `pi_pp_upw1h`, every routine of 3 steps bit for bit.
    (tra_adv_hor = 'MUSCL' cannot be pinned this way: the reference forms nboundary_lay from the rank's own edges and never exchanges it
    (oce_muscl_adv.F90:74-104), so on 2 ranks 5 halo nodes of pi carry incomplete values and the two ranks disagree about the flux on shared
    edges -- its result depends on the partition.  The library forms the array by the same rank-local rule; MUSCL is pinned by the
    2-rank Fortran drop-in run against the reference's 2-rank CPU run, tests/test_gpu_dropin.py[pi_pp_muscl-2].)"""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.synthetic import analytic_ts
    from oracle_lib import Oracle
    from ref_chain import run_reference_chain
    mesh = Mesh.load(PI, dt=900.0)
    par = make_params(dt=900.0, tra_adv_ver=ver, **kw)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    orc = Oracle(mesh, par)
    orc.set_state(st)
    g = gold(cfg)
    for f in FORCING:
        orc.set(f, g["forcing/" + f])
    bad = run_reference_chain(orc, mesh, g, steps=(1, 2, 3))
    assert not bad, "\n".join(bad[:20])


@pytest.mark.parametrize("mix,cfg", [("PP", "pi_pp_kv0"), ("KPP", "pi_kpp_kv0")])
def test_oracle_chain_bitwise_kv0_background(built, mix, cfg):
    """Kv0_const=.false.: latitude/depth dependent background diffusivity Kv0_background_qiang (src/oce_ale_mixing_pp.F90:91-125) in
    oce_mixing_PP (:73-76) and in KPP's ri_iwmix (oce_ale_mixing_kpp.F90:821-822); reference runs `pi_pp_kv0` / `pi_kpp_kv0`, every
    routine of 3 steps bit for bit (atan: the oracle and the reference build use the same libm here)."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.synthetic import analytic_ts
    from oracle_lib import Oracle
    from ref_chain import run_reference_chain
    mesh = Mesh.load(PI, dt=900.0)
    par = make_params(dt=900.0, mix_scheme=mix, Kv0_const=False)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    orc = Oracle(mesh, par)
    orc.set_state(st)
    g = gold(cfg)
    for f in FORCING:
        orc.set(f, g["forcing/" + f])
    bad = run_reference_chain(orc, mesh, g, steps=(1, 2, 3))
    assert not bad, "\n".join(bad[:20])


CAVITY = os.path.join(REPO, "tests", "golden", "meshes", "pi_cavity")


def test_mesh_layer_bitwise_cavity(built):
    """use_cavity=.true.: the upper levels ulevels / ulevels_nod2D(_max) and the cavity depth read from cavity_elvls / cavity_nlvls / cavity_depth
    (src/oce_mesh.F90:897-1280), areasvol from the lower face under a shelf (mesh_areas, :1986-2050), zbar_e_srf / zbar_n_srf at the upper level
    (src/oce_ale.F90:520-560) and everything else of the set-up == the reference run `pi_pp_cavity` on tests/golden/meshes/pi_cavity
    (tests/golden/make_cavity_mesh.py: a synthetic draft on the pi mesh -- the reference ships no cavity mesh)."""
    from fesom2_amd.mesh import Mesh
    g = gold("pi_pp_cavity")
    mesh = Mesh.load(CAVITY, dt=900.0, use_cavity=True)
    assert (mesh.ulevels > 1).sum() > 700 and (mesh.ulevels_nod2D > 1).sum() > 300
    bad = []
    for f in SETUP_FIELDS + ["ulevels", "ulevels_nod2D", "ulevels_nod2D_max"]:
        if "setup/" + f not in g:
            continue
        ok, msg = check_digest(getattr(mesh, f), g["setup/" + f])
        if not ok:
            bad.append(f"{f}: {msg}")
    st = mesh.initial_state(2)
    for f in ("hnode", "helem", "zbar_3d_n", "Z_3d_n", "eta_n", "hbar"):
        ok, msg = check_digest(getattr(st, f), g["setup/" + f])
        if not ok:
            bad.append(f"state {f}: {msg}")
    assert not bad, "\n".join(bad)


# cavity variants: (reference run, mesh options, parameters)
CAVITY_CASES = [
    ("pi_pp_cavity", dict(), dict()),
    ("pi_default_cavity", dict(), dict(mix_scheme="KPP", Fer_GM=True, Redi=True)),
    ("pi_pp_cavity_pc", dict(use_cavity_partial_cell=True), dict()),                                   # partial cells at the shelf base (set-up only with zstar)
    ("pi_pp_cavity_easypgf", dict(), dict(which_pgf="easypgf")),
    ("pi_pp_cavity_cubicspline", dict(), dict(which_pgf="cubicspline")),
    ("pi_pp_zlevel_cavity", dict(which_ale="zlevel"), dict()),
    ("pi_pp_linfs_cavity", dict(which_ale="linfs", use_partial_cell=False), dict()),                   # full cells: gradient of the hydrostatic pressure (cavity branch of pressure_bv)
    ("pi_pp_linfs_pc_cavity", dict(which_ale="linfs"), dict()),                                        # pressure_force_4_linfs_shchepetkin: the correction stays directly under the shelf
    ("pi_pp_linfs_easypgf_cavity", dict(which_ale="linfs"), dict(which_pgf="easypgf")),
    ("pi_pp_linfs_cubic_cavity", dict(which_ale="linfs"), dict(which_pgf="cubicspline")),              # + pressure boundary term at the shelf base
    ("pi_pp_linfs_nemo_cavity", dict(which_ale="linfs"), dict(which_pgf="nemo")),
    ("pi_pp_linfs_cavity_sergey", dict(which_ale="linfs", use_cavity_partial_cell=True), dict(which_pgf="sergey")),     # pressure_force_4_linfs_cavity
    ("pi_pp_linfs_cavity_pc_shch", dict(which_ale="linfs", use_cavity_partial_cell=True), dict()),
]


@pytest.mark.parametrize("cfg,mkw,kw", CAVITY_CASES, ids=[c[0] for c in CAVITY_CASES])
def test_oracle_chain_bitwise_cavity(built, cfg, mkw, kw):
    """Ice-shelf cavities (use_cavity=.true.): every routine of the step from its column's upper level, the reference density profile of
    init_ref_density (src/oce_ale_pressure_bv.F90:3036-3073; use_density_ref is forced on, src/oce_setup_step.F90), pressure_bv's interface-water fill
    above the shelf base and the cavity branch of hpressure (:214-260, :420-470), the FCT bounds that see the untouched scratch entries above an element's
    upper level (src/oce_adv_tra_fct.F90:110-142) and CFL_z accumulating at the top of a cavity column (src/oce_ale.F90:2141-2152): reference runs
    `pi_pp_cavity` (PP) and `pi_default_cavity` (KPP + GM + Redi), 2 ranks, surface forcing, every routine of 3 steps bit for bit.
    Variants (CAVITY_CASES), each its own reference run on the same mesh, set-up arrays included: use_cavity_partial_cell (init_surface_elem_depth /
    init_surface_node_depth, src/oce_ale.F90:422-545), which_ALE zlevel and linfs, every pressure-gradient scheme the reference offers under a shelf incl.
    'sergey' = pressure_force_4_linfs_cavity (src/oce_ale_pressure_bv.F90:385-403, 1451-1663)."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.synthetic import analytic_ts
    from oracle_lib import Oracle
    from ref_chain import run_reference_chain
    mesh = Mesh.load(CAVITY, dt=900.0, use_cavity=True, **mkw)
    par = make_params(dt=900.0, use_cavity=True, **mkw, **kw)
    g = gold(cfg)
    st = mesh.initial_state(2)
    bad = []
    for f in SETUP_FIELDS:
        ok, msg = check_digest(getattr(mesh, f), g["setup/" + f])
        if not ok:
            bad.append(f"{f}: {msg}")
    for f in ("hnode", "helem", "zbar_3d_n", "Z_3d_n"):
        ok, msg = check_digest(getattr(st, f), g["setup/" + f])
        if not ok:
            bad.append(f"state {f}: {msg}")
    assert not bad, "\n".join(bad)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(CAVITY)
    st.tr_arr_old[...] = st.tr_arr
    orc = Oracle(mesh, par)
    orc.set_state(st)
    g = gold(cfg)
    for f in FORCING:
        orc.set(f, g["forcing/" + f])
    bad = run_reference_chain(orc, mesh, g, steps=(1, 2, 3))
    assert not bad, "\n".join(bad[:20])


def test_oracle_chain_bitwise_density_ref(built):
    """use_density_ref=.true. without cavities (namelist.oce &oce_dyn): density_m_rho0 and the densities of the nemo / cubic-spline PGF against the profile
    init_ref_density forms from (density_ref_T, density_ref_S) at the initial layer depths (src/oce_ale_pressure_bv.F90:3036-3073) instead of density_0;
    reference run `pi_pp_dref`, every routine of 3 steps bit for bit."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.synthetic import analytic_ts
    from oracle_lib import Oracle
    from ref_chain import run_reference_chain
    mesh = Mesh.load(PI, dt=900.0)
    par = make_params(dt=900.0, use_density_ref=True)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    orc = Oracle(mesh, par)
    orc.set_state(st)
    g = gold("pi_pp_dref")
    for f in FORCING:
        orc.set(f, g["forcing/" + f])
    bad = run_reference_chain(orc, mesh, g, steps=(1, 2, 3))
    assert not bad, "\n".join(bad[:20])


def test_oracle_chain_bitwise_scaling_rossby(built):
    """scaling_Rossby=.true. (namelist.oce &oce_dyn): K_GM cut off by a Fermi function of mesh resolution / first baroclinic Rossby radius
    (init_Redi_GM, src/oce_fer_gm.F90:196-200) ahead of the resolution scaling and the ramp; reference run `pi_default_rossby` (KPP + GM + Redi),
    every routine of 3 steps bit for bit (exp: the oracle and the reference build use the same libm here)."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.synthetic import analytic_ts
    from oracle_lib import Oracle
    from ref_chain import run_reference_chain
    mesh = Mesh.load(PI, dt=900.0)
    par = make_params(dt=900.0, mix_scheme="KPP", Fer_GM=True, Redi=True, scaling_Rossby=True)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    orc = Oracle(mesh, par)
    orc.set_state(st)
    g = gold("pi_default_rossby")
    for f in FORCING:
        orc.set(f, g["forcing/" + f])
    bad = run_reference_chain(orc, mesh, g, steps=(1, 2, 3))
    assert not bad, "\n".join(bad[:20])
    assert not np.array_equal(g["s1/gm.fer_K"], gold("pi_default")["s1/gm.fer_K"])      # (the cut-off acts on pi: the reference's own fer_K differs from the run without it)


def test_oracle_chain_bitwise_no_limiter_w_split(built):
    """tra_adv_lim = 'NON' together with w_split: without the FCT low-order solution the implicit part of the vertical velocity, Wvel_i, enters the
    implicit diffusion solve as upwind terms of its three diagonals (do_wimpl, src/oce_ale_tracer.F90:424, 560-572, 604-617, 641-649); reference run
    `pi_pp_non_wsplit` (w_max_cfl = 0.0003 so that the split is active on pi), every routine of 3 steps bit for bit."""
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.synthetic import analytic_ts
    from oracle_lib import Oracle
    from ref_chain import run_reference_chain
    mesh = Mesh.load(PI, dt=900.0)
    par = make_params(dt=900.0, tra_adv_lim="NON", w_split=True, w_max_cfl=0.0003)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    orc = Oracle(mesh, par)
    orc.set_state(st)
    g = gold("pi_pp_non_wsplit")
    for f in FORCING:
        orc.set(f, g["forcing/" + f])
    bad = run_reference_chain(orc, mesh, g, steps=(1, 2, 3))
    assert not bad, "\n".join(bad[:20])
    assert not np.array_equal(g["s3/tr1.end.tr_arr"], gold("pi_pp_non")["s3/tr1.end.tr_arr"])      # (the split acts)
