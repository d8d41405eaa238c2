"""CPU: size-independent properties of the hot path, checked on the oracle (the GPU versions of the same checks are in
test_gpu_parity.py): conservation identities the reference monitors in write_step_info (src/write_step_info.F90:14-218),
solver tolerance, boundedness of FCT, linear-EOS and linfs variants (parity UNPINNED for those two: no reference
fixture exists for pi with state_equation=0 / which_ALE='linfs'; they are checked for self-consistency only)."""
import os
import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PI = os.path.join(REPO, "tests", "golden", "meshes", "pi")


def make(which_ale="zstar", state_equation=1, with_diffusion=True, mix="PP", x0=2):
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from fesom2_amd.synthetic import analytic_ts
    from oracle_lib import Oracle
    mesh = Mesh.load(PI, dt=900.0, which_ale=which_ale, use_partial_cell=(which_ale != "linfs"))
    par = make_params(dt=900.0, which_ale=which_ale, use_partial_cell=(which_ale != "linfs"), state_equation=state_equation,
                      with_diffusion=with_diffusion, mix_scheme=mix, solver_x0_order=x0)
    st = mesh.initial_state(2)
    st.tr_arr[0], st.tr_arr[1] = analytic_ts(PI)
    st.tr_arr_old[...] = st.tr_arr
    orc = Oracle(mesh, par)
    orc.set_state(st)
    return mesh, orc


def content(mesh, orc):
    nlm1 = mesh.nl - 1
    T = orc.get("tr_arr").reshape(2, -1, nlm1)
    h = orc.get("hnode").reshape(-1, nlm1)
    a = mesh.areasvol[:, :nlm1]
    wet = np.arange(nlm1)[None, :] < (mesh.nlevels_nod2D[:, None] - 1)
    return [float((T[i] * h * a)[wet].sum()) for i in range(2)]


def test_conservation_and_solver(built):
    mesh, orc = make()
    c0 = content(mesh, orc)
    a1 = mesh.areasvol[:, 0]
    for n in range(1, 11):
        orc.call("step", n)
        assert orc.solver_residual < 1e-10 and orc.solver_iterations < 200
        # sum over nodes of the divergence-form SSH right-hand side vanishes without water flux
        assert abs(orc.get("ssh_rhs_old").sum()) < 1e-3 * np.abs(orc.get("ssh_rhs_old")).sum() * 1e-9 + 1e-2
    c1 = content(mesh, orc)
    for i in range(2):
        assert abs(c1[i] - c0[i]) / abs(c0[i]) < 1e-12, (i, c0[i], c1[i])        # tracer content (north_star: 1e-12)
    # volume: int(eta) == int(hbar) (alpha=1) to round-off, write_step_info prints int(deta)-int(dhbar)
    eta, hbar = orc.get("eta_n"), orc.get("hbar")
    assert abs(((eta - hbar) * a1).sum()) / a1.sum() < 1e-15
    T = orc.get("tr_arr").reshape(2, -1, mesh.nl - 1)
    assert np.isfinite(T).all() and T[1].max() <= 45.0


def test_fct_is_bounded(built):
    """pure advection (no diffusion, constant Av/Kv): FCT keeps T,S inside the initial global range"""
    mesh, orc = make(with_diffusion=False, mix="none")
    nlm1 = mesh.nl - 1
    wet = np.arange(nlm1)[None, :] < (mesh.nlevels_nod2D[:, None] - 1)
    T0 = orc.get("tr_arr").reshape(2, -1, nlm1)
    lo = [T0[i][wet].min() for i in range(2)]; hi = [T0[i][wet].max() for i in range(2)]
    for n in range(1, 9):
        orc.call("step", n)
    T = orc.get("tr_arr").reshape(2, -1, nlm1)
    for i in range(2):
        assert T[i][wet].min() >= lo[i] - 1e-9 and T[i][wet].max() <= hi[i] + 1e-9


@pytest.mark.parametrize("which_ale,eos", [("linfs", 1), ("zstar", 0)])
def test_variants_run(built, which_ale, eos):
    mesh, orc = make(which_ale=which_ale, state_equation=eos)
    for n in range(1, 5):
        orc.call("step", n)
    assert np.isfinite(orc.get("tr_arr")).all() and np.isfinite(orc.get("UV")).all()
    assert np.abs(orc.get("eta_n")).max() < 5.0


def test_solver_zero_rhs_and_idempotence(built):
    """a converged iterate is a fixed point: solving again from the solution performs zero iterations"""
    mesh, orc = make(x0=0)                   # reference initial guess (x0 = previous d_eta)
    orc.call("step", 1)
    for r in ("compute_vel_nodes", "pressure_bv", "pressure_force", "sw_alpha_beta", "compute_sigma_xy", "compute_neutral_slope",
              "mixing_pp", "mo_convect", "compute_vel_rhs", "visc_filt_bcksct", "impl_vert_visc_ale", "update_stiff_mat_ale",
              "compute_ssh_rhs_ale", "solve_ssh"):
        orc.call(r)
    x = orc.get("d_eta").copy()
    orc.call("solve_ssh")
    # y = D x, x = y/D round trip: last-bit changes only
    assert orc.solver_iterations == 0 and np.allclose(orc.get("d_eta"), x, rtol=4e-16, atol=0)
