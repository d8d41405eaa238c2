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


@pytest.mark.parametrize("cfg", ["pi_kpp", "pi_default"])
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
    full = cfg == "pi_default"
    mesh = Mesh.load(PI, dt=900.0)
    par = make_params(dt=900.0, mix_scheme="KPP", Fer_GM=full, Redi=full)
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
