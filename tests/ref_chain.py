"""Routine-by-routine comparison of the C oracle with the committed digests of a reference run (order of
src/oce_ale.F90:2556-2767).  The SSH solve (pARMS RAS+ILU in the reference, Jacobi-BiCGstab here) is compared to the
solver tolerance and the reference's d_eta is then injected so that every later routine sees identical inputs.
`toy` adds the Soufflet channel hooks (src/toy_channel_soufflet.F90)."""
import numpy as np
from golden_util import check_digest, wet_masks


def run_reference_chain(orc, mesh, g, steps=(1, 2, 3), toy=False, skip=(), after_step=None, node_keep=None, check_steps=None):
    """returns the list of mismatches (empty = bit-identical on every sampled value).  `skip` = set of (step, key)
    entries that are known to differ (documented where used).  `node_keep` (bool per global node): tracer fields are compared at these nodes only
    (options whose result the reference makes depend on the partition: the goldens come from a 2-rank run).  `check_steps`: the steps whose routine outputs the golden file holds
    (the steps between them are run without comparison; the file holds the reference's d_eta of every step)."""
    W = wet_masks(mesh)
    nlm1 = mesh.nl - 1
    bad = []

    def chk(step, field, key, mask=None, sub=None):
        if (step, key) in skip or (check_steps is not None and step not in check_steps):
            return
        a = orc.get(field)
        if sub is not None:
            a = a.reshape(2, -1, nlm1)[sub]
        mk = None if mask is None else W[mask]
        if node_keep is not None and field in ("tr_arr", "del_ttf") and a.size % node_keep.size == 0:
            keep = np.broadcast_to(node_keep[:, None], (node_keep.size, a.size // node_keep.size // (2 if (field == "tr_arr" and sub is None) else 1)))
            keep = np.tile(keep.reshape(-1), 2) if (field == "tr_arr" and sub is None) else keep.reshape(-1)
            mk = keep.reshape(np.shape(mk)) & mk if mk is not None else keep
        ok, msg = check_digest(a, g[f"s{step}/{key}"], mk)
        if not ok:
            bad.append(f"step {step} {key}: {msg}")

    for step in steps:
        for f in ("tr_arr", "UV", "eta_n", "hnode", "helem", "Wvel_e", "zbar_3d_n", "Z_3d_n", "ssh_rhs_old"):
            chk(step, f, "in." + f)
        if toy and step % 10 == 0:
            orc.call("compute_zonal_mean")               # before_oce_step
        orc.call("compute_vel_nodes"); chk(step, "Unode", "compute_vel_nodes.Unode", "n2")
        orc.call("pressure_bv")
        chk(step, "density_m_rho0", "pressure_bv.density_m_rho0", "n"); chk(step, "bvfreq", "pressure_bv.bvfreq", "nl")
        chk(step, "MLD1", "pressure_bv.MLD1"); chk(step, "MLD2", "pressure_bv.MLD2")
        orc.call("pressure_force"); chk(step, "pgf_x", "pressure_force.pgf_x", "e"); chk(step, "pgf_y", "pressure_force.pgf_y", "e")
        orc.call("sw_alpha_beta"); chk(step, "sw_alpha", "sw_alpha_beta.sw_alpha", "n"); chk(step, "sw_beta", "sw_alpha_beta.sw_beta", "n")
        orc.call("compute_sigma_xy"); chk(step, "sigma_xy", "compute_sigma_xy.sigma_xy", "n2")
        orc.call("compute_neutral_slope")
        chk(step, "neutral_slope", "compute_neutral_slope.neutral_sl", "n3"); chk(step, "slope_tapered", "compute_neutral_slope.slope_tape", "n3")
        if orc.params.mix_scheme == 1:                      # KPP (src/oce_ale_mixing_kpp.F90)
            chk(step, "dbsfc", "kpp.dbsfc", "nl")
            orc.call("mixing_kpp")
            chk(step, "kpp_hbl", "kpp.hbl"); chk(step, "kpp_ghats", "kpp.ghats")
            for j in (1, 2, 3):
                chk(step, f"kpp_blmc{j}", f"kpp.blmc{j}")
            chk(step, "kpp_Kv1", "kpp.Kv1"); chk(step, "kpp_Kv2", "kpp.Kv2"); chk(step, "Av", "kpp.Av")
        else:
            orc.call("mixing_pp"); chk(step, "Av", "oce_mixing_PP.Av"); chk(step, "Kv", "oce_mixing_PP.Kv")
        orc.call("mo_convect"); chk(step, "Av", "mixing.Av"); chk(step, "Kv", "mixing.Kv")
        if orc.params.use_momix:
            chk(step, "mixlength", "mixing.mixlength")
        orc.call("compute_vel_rhs"); chk(step, "UV_rhs", "compute_vel_rhs.UV_rhs", "e2"); chk(step, "UV_rhsAB", "compute_vel_rhs.UV_rhsAB", "e2")
        orc.call("viscosity_filter"); chk(step, "UV_rhs", "viscosity_filter.UV_rhs", "e2")
        if orc.params.visc_option == 8:                     # backscatter_coef, uke_update (src/oce_dyn.F90:967-1152)
            chk(step, "v_back", "viscosity_filter.v_back", "e"); chk(step, "uke", "viscosity_filter.uke", "e"); chk(step, "uke_rhs", "viscosity_filter.uke_rhs", "e")
        if orc.params.visc_option <= 3:                     # h_viscosity_leith (src/oce_dyn.F90:461-561)
            chk(step, "vorticity", "viscosity_filter.vorticity", "n"); chk(step, "Visc", "viscosity_filter.Visc", "e")
        orc.call("impl_vert_visc_ale"); chk(step, "UV_rhs", "impl_vert_visc_ale.UV_rhs", "e2")
        if orc.params.which_ale != 0:
            orc.call("update_stiff_mat_ale")
        orc.call("compute_ssh_rhs_ale"); chk(step, "ssh_rhs", "compute_ssh_rhs_ale.ssh_rhs")
        orc.call("solve_ssh")
        ref = g[f"s{step}/full.d_eta"]
        mine = orc.get("d_eta")
        assert orc.solver_residual < 1e-10
        # tolerance: both solves stop at ||scaled residual|| < 1e-10; scaled operator is O(1) -> |dx| ~ 1e-9
        assert np.abs(mine - ref).max() < 5e-9, np.abs(mine - ref).max()
        orc.set("d_eta", ref)
        if toy:
            orc.call("relax_zonal_vel"); chk(step, "UV_rhs", "relax_zonal_vel.UV_rhs", "e2")
        orc.call("update_vel"); chk(step, "UV", "update_vel.UV", "e2"); chk(step, "eta_n", "update_vel.eta_n")
        orc.call("compute_hbar_ale")
        for f in ("hbar", "hbar_old", "ssh_rhs_old", "dhe"):
            chk(step, f, "compute_hbar_ale." + f)
        orc.call("eta_update"); chk(step, "eta_n", "eta_n_update.eta_n")
        gm = bool(orc.params.Fer_GM)
        if orc.params.Redi and not gm:
            orc.call("init_Redi_GM")
        if gm:                                              # oce_ale.F90:2729-2739
            orc.call("init_Redi_GM"); chk(step, "fer_K", "gm.fer_K", "nl"); chk(step, "fer_c", "gm.fer_c")
            orc.call("fer_solve_Gamma"); chk(step, "fer_gamma", "gm.fer_gamma", "nl2")
            orc.call("fer_gamma2vel"); chk(step, "fer_UV", "gm.fer_UV", "e2")
        orc.call("vert_vel_ale")
        for f in ("Wvel", "Wvel_e", "Wvel_i", "CFL_z"):
            chk(step, f, "vert_vel_ale." + f, "nl")
        chk(step, "hnode_new", "vert_vel_ale.hnode_new", "n")
        if gm:
            orc.call("fer_wvel"); chk(step, "fer_Wvel", "gm.fer_Wvel", "nl")
            orc.call("bolus_add")                           # solve_tracers_ale: UV, Wvel(_e) += bolus velocities around the tracer loop
        if orc.params.SPP:                                  # solve_tracers_ale :120-121 (src/oce_spp.F90)
            orc.call("spp"); chk(step, "tr_arr", "spp.salt", "n", sub=1)
        for tr in (1, 2):
            p = f"tr{tr}."
            orc.call("init_tracers_AB", tr)
            chk(step, "tr_arr_old", p + "init_AB.tr_arr_old", "n", sub=tr - 1)
            chk(step, "tr_xy", p + "init_AB.tr_xy", "e2"); chk(step, "tr_z", p + "init_AB.tr_z", "nl")
            chk(step, "edge_up_dn_grad", p + "init_AB.edge_up_dn_grad")
            orc.call("adv_tracers_ale", tr)
            for f in ("fct_LO", "fct_ttf_max", "fct_ttf_min", "fct_plus", "fct_minus", "del_ttf_advhoriz", "del_ttf_advvert", "del_ttf"):
                chk(step, f, p + "adv." + f, "n")
            chk(step, "adv_flux_hor", p + "adv.adv_flux_hor"); chk(step, "adv_flux_ver", p + "adv.adv_flux_ver", "nl")
            orc.call("diff_tracers_ale", tr)
            chk(step, "del_ttf", p + "diff.del_ttf", "n")
            if toy:
                orc.call("relax_zonal_temp")             # after every tracer of the loop, always on tracer 1
            elif orc.params.clim_relax > 1.0e-8:
                orc.call("relax_to_clim", tr)
            chk(step, "tr_arr", p + "end.tr_arr", "n", sub=tr - 1)
        if gm:
            orc.call("bolus_remove")
        orc.call("salinity_clamp")
        orc.call("update_thickness_ale")
        for f in ("hnode", "helem", "zbar_3d_n", "Z_3d_n"):
            chk(step, f, "update_thickness_ale." + f)
        chk(step, "tr_arr", "out.tr_arr"); chk(step, "UV", "out.UV", "e2"); chk(step, "eta_n", "out.eta_n")
        if after_step is not None:
            after_step(step)
        if bad:
            break
    return bad
