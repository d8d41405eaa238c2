#!/usr/bin/env python3
"""Dev/test tool: chain-compare the C oracle (oracle/c) with the real reference (oracle/_ref) routine by
routine on the same inputs, bit for bit.  The reference runs on NP ranks in replay mode (oracle/ref/driver.F90);
owned parts of its dumps are reassembled to global numbering.  After the SSH solve the reference's d_eta is
injected into the oracle (the reference's pARMS RAS+ILU solver is not restated), so that every other routine
is compared on identical inputs.  Usage: compare_oracle.py CFG NP NSTEPS"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
from refdump import read_dump
from oracle.ref import run_ref


def assemble(dumps, setups, name):
    """global array from per-rank dumps (owned parts)."""
    a0 = dumps[0][name]
    dims = setups[0]["dims"]
    out = None
    for d, s in zip(dumps, setups):
        a = d[name]
        nod2D, elem2D, edge2D, _, nl, myN, eN, myE, eE, eX, myD, eD = [int(x) for x in s["dims"]]
        haxis = 1 if name.endswith("tr_arr") or name.endswith("tr_arr_old") and a.ndim == 3 and a.shape[0] == 2 and a.shape[1] == myN + eN else 0
        if a.ndim == 3 and a.shape[0] == 2 and a.shape[1] == myN + eN and a.shape[2] == nl - 1 and "tr_arr" in name:
            haxis = 1
        else:
            haxis = 0
        L = a.shape[haxis]
        if L in (myN + eN, myN):
            glob, own, lst = nod2D, myN, s["myList_nod2D"]
        elif L in (myE + eE, myE, myE + eE + eX):
            glob, own, lst = elem2D, myE, s["myList_elem2D"]
        elif L in (myD + eD, myD):
            glob, own, lst = edge2D, myD, s["myList_edge2D"]
        else:
            return None
        if out is None:
            shp = list(a.shape); shp[haxis] = glob
            out = np.zeros(shp, dtype=a.dtype)
        idx = lst[:own] - 1
        if haxis == 0:
            out[idx] = a[:own]
        else:
            out[:, idx] = a[:, :own]
    return out


def bits_equal(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64); b = np.ascontiguousarray(b, dtype=np.float64)
    return (a.view(np.int64) == b.view(np.int64)) | ((a == 0) & (b == 0))


def report(tag, mine, ref, mask=None):
    eq = bits_equal(mine.reshape(ref.shape), ref)
    if mask is not None:
        eq = eq | ~mask
    nbad = int((~eq).sum())
    if nbad == 0:
        print(f"  ok    {tag:44s} bitwise ({ref.size})")
        return True
    m = mine.reshape(ref.shape)
    d = np.abs(m - ref)[~eq]
    rel = d / (np.abs(ref[~eq]) + 1e-300)
    i = np.argwhere(~eq)[0]
    print(f"  DIFF  {tag:44s} n={nbad}/{ref.size} maxabs={d.max():.3e} maxrel={rel.max():.3e} first={tuple(i)} mine={m[tuple(i)]!r} ref={ref[tuple(i)]!r}")
    return False


def main():
    cfg, np_, nsteps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    rd, rc, lines = run_ref.run(cfg, np_, nsteps, mode="replay", dump=tuple(range(1, nsteps + 1)))
    assert rc == 0, (rc, lines)
    setups = [read_dump(os.path.join(rd, "dumps", f"setup.r{r:05d}.bin")) for r in range(np_)]
    os.environ.setdefault("FESOM_GPU_LIB", "/tmp/libmesh_test.so")
    from fesom2_amd.mesh import Mesh
    from fesom2_amd.config import make_params
    from oracle_lib import Oracle
    c = run_ref.CFGS[cfg]
    meshdir = os.path.join(REPO, "tests", "golden", "meshes", c["mesh"])
    dt = 86400.0 / c["step_per_day"]
    mesh = Mesh.load(meshdir, which_ale=c["which_ale"], use_partial_cell=c["use_partial_cell"] == ".true.",
                     force_rotation=c["force_rotation"] == ".true.", cyclic_length_deg=float(c["cyclic_length"]), dt=dt,
                     K_hor=float(c["k_hor"]))
    par = make_params(dt=dt, which_ale=c["which_ale"], use_partial_cell=c["use_partial_cell"] == ".true.",
                      state_equation=c["state_equation"], mix_scheme=c["mix_scheme"], with_diffusion=True,
                      toy_soufflet=c["toy_ocean"] == ".true.", K_hor=float(c["k_hor"]), Fer_GM=c["fer_gm"] == ".true.", Redi=c["redi"] == ".true.",
                      cyclic_length_deg=float(c["cyclic_length"]))
    orc = Oracle(mesh, par)
    st = mesh.initial_state(2)
    # initial tracers / velocities from the reference's setup dump (same bits)
    for k in ("tr_arr", "tr_arr_old", "UV"):
        st.a[k][...] = assemble(setups, setups, k)
    orc.set_state(st)
    toy = c["toy_ocean"] == ".true."
    if toy:
        # Soufflet channel: Coriolis is redefined by initial_state_soufflet, Tclim/Uclim are the relaxation targets
        # (Uclim is private to the reference's toy module; it is the initial UV(1,:,:), toy_channel_soufflet.F90:327-328);
        # zonal sums are formed per rank and added in rank order like MPI_Allreduce does
        import ctypes as C
        mesh.coriolis[...] = assemble(setups, setups, "coriolis")
        orc.set("Tclim", assemble(setups, setups, "Tclim"))
        orc.set("Uclim", np.ascontiguousarray(st.a["UV"][:, :, 0]))
        rank_of_node = np.zeros(mesh.nod2D, dtype=np.int32)
        for r, s_ in enumerate(setups):
            myN = int(s_["dims"][5])
            rank_of_node[s_["myList_nod2D"][:myN] - 1] = r
        owner = np.ascontiguousarray(rank_of_node[mesh.elem2D_nodes[:, 0] - 1], dtype=np.int32)
        orc.lib.orc_toy_set_partition(owner.ctypes.data_as(C.POINTER(C.c_int)), np_)
        orc.call("compute_zonal_mean_ini"); orc.call("compute_zonal_mean")
    allok = True
    nlm1 = mesh.nl - 1
    lev = np.arange(nlm1)[None, :]
    wet_n = lev < (mesh.nlevels_nod2D[:, None] - 1)
    wet_e = lev < (mesh.nlevels[:, None] - 1)
    levl = np.arange(mesh.nl)[None, :]
    wet_nl = levl < mesh.nlevels_nod2D[:, None]
    for step in range(1, nsteps + 1):
        print(f"=== step {step}")
        d = [read_dump(os.path.join(rd, "dumps", f"replay{step:04d}.r{r:05d}.bin")) for r in range(np_)]
        g = lambda name: assemble(d, setups, name)
        def chk(routine, field, refname, mask=None, arg=None):
            nonlocal allok
            ok = report(f"{routine}:{field}", orc.get(field), g(refname), mask)
            allok &= ok
        # inputs check
        for f in ("tr_arr", "UV", "eta_n", "hnode", "helem", "Wvel_e", "zbar_3d_n", "Z_3d_n", "ssh_rhs_old"):
            chk("in", f, "in." + f)
        if toy and step % 10 == 0:
            orc.call("compute_zonal_mean")               # before_oce_step
        orc.call("compute_vel_nodes"); chk("compute_vel_nodes", "Unode", "compute_vel_nodes.Unode", np.repeat(wet_n[:, :, None], 2, 2))
        orc.call("pressure_bv")
        chk("pressure_bv", "density_m_rho0", "pressure_bv.density_m_rho0", wet_n)
        chk("pressure_bv", "bvfreq", "pressure_bv.bvfreq", wet_nl)
        chk("pressure_bv", "MLD1", "pressure_bv.MLD1"); chk("pressure_bv", "MLD2", "pressure_bv.MLD2")
        orc.call("pressure_force"); chk("pressure_force", "pgf_x", "pressure_force.pgf_x", wet_e); chk("pressure_force", "pgf_y", "pressure_force.pgf_y", wet_e)
        orc.call("sw_alpha_beta"); chk("sw_alpha_beta", "sw_alpha", "sw_alpha_beta.sw_alpha", wet_n); chk("sw_alpha_beta", "sw_beta", "sw_alpha_beta.sw_beta", wet_n)
        orc.call("compute_sigma_xy"); chk("compute_sigma_xy", "sigma_xy", "compute_sigma_xy.sigma_xy", np.repeat(wet_n[:, :, None], 2, 2))
        orc.call("compute_neutral_slope")
        chk("compute_neutral_slope", "neutral_slope", "compute_neutral_slope.neutral_sl", np.repeat(wet_n[:, :, None], 3, 2))
        chk("compute_neutral_slope", "slope_tapered", "compute_neutral_slope.slope_tape", np.repeat(wet_n[:, :, None], 3, 2))
        if par.mix_scheme == 2:
            orc.call("mixing_pp"); chk("mixing_pp", "Av", "oce_mixing_PP.Av"); chk("mixing_pp", "Kv", "oce_mixing_PP.Kv")
            orc.call("mo_convect"); chk("mo_convect", "Av", "mixing.Av"); chk("mo_convect", "Kv", "mixing.Kv")
        orc.call("compute_vel_rhs")
        we2 = np.repeat(wet_e[:, :, None], 2, 2)
        chk("compute_vel_rhs", "UV_rhs", "compute_vel_rhs.UV_rhs", we2); chk("compute_vel_rhs", "UV_rhsAB", "compute_vel_rhs.UV_rhsAB", we2)
        orc.call("visc_filt_bcksct"); chk("visc_filt_bcksct", "UV_rhs", "viscosity_filter.UV_rhs", we2)
        orc.call("impl_vert_visc_ale"); chk("impl_vert_visc_ale", "UV_rhs", "impl_vert_visc_ale.UV_rhs", we2)
        if par.which_ale != 0:
            orc.call("update_stiff_mat_ale")
            # reference values are per rank in local CSR order: compare rank 0..n rows via global reassembly of rows
            vals = np.concatenate([dd["update_stiff_mat_ale.values"] for dd in d]) if np_ > 1 else d[0]["update_stiff_mat_ale.values"]
            if np_ == 1:
                allok &= report("update_stiff_mat_ale:ssh_values", orc.get("ssh_values"), vals)
        orc.call("compute_ssh_rhs_ale"); chk("compute_ssh_rhs_ale", "ssh_rhs", "compute_ssh_rhs_ale.ssh_rhs")
        # solver: compare to tolerance, then inject the reference solution
        orc.call("solve_ssh")
        mine, ref = orc.get("d_eta"), g("solve_ssh_ale.d_eta")
        print(f"  solve_ssh: iterations={orc.solver_iterations} resid={orc.solver_residual:.3e} max|d_eta-ref|={np.abs(mine-ref).max():.3e} max|ref|={np.abs(ref).max():.3e}")
        orc.set("d_eta", ref)
        if toy:
            orc.call("relax_zonal_vel"); chk("relax_zonal_vel", "UV_rhs", "relax_zonal_vel.UV_rhs", we2)
        orc.call("update_vel"); chk("update_vel", "UV", "update_vel.UV", we2); chk("update_vel", "eta_n", "update_vel.eta_n")
        orc.call("compute_hbar_ale")
        for f in ("hbar", "hbar_old", "ssh_rhs_old", "dhe"):
            chk("compute_hbar_ale", f, "compute_hbar_ale." + f)
        orc.call("eta_update"); chk("eta_update", "eta_n", "eta_n_update.eta_n")
        if par.Redi and not par.Fer_GM:
            orc.call("init_Redi_GM")
        if par.Fer_GM:
            orc.call("init_Redi_GM"); chk("gm", "fer_K", "gm.fer_K", wet_nl); chk("gm", "fer_c", "gm.fer_c")
            orc.call("fer_solve_Gamma"); chk("gm", "fer_gamma", "gm.fer_gamma", np.repeat(wet_nl[:, :, None], 2, 2))
            orc.call("fer_gamma2vel"); chk("gm", "fer_UV", "gm.fer_UV", we2)
        orc.call("vert_vel_ale")
        for f in ("Wvel", "Wvel_e", "Wvel_i", "CFL_z"):
            chk("vert_vel_ale", f, "vert_vel_ale." + f, wet_nl)
        chk("vert_vel_ale", "hnode_new", "vert_vel_ale.hnode_new", wet_n)
        if par.Fer_GM:
            orc.call("fer_wvel"); chk("gm", "fer_Wvel", "gm.fer_Wvel", wet_nl)
            orc.call("bolus_add")
        for tr in (1, 2):
            p = f"tr{tr}."
            orc.call("init_tracers_AB", tr)
            allok &= report(f"init_AB{tr}:tr_arr_old", orc.get("tr_arr_old").reshape(2, -1, nlm1)[tr - 1], g(p + "init_AB.tr_arr_old"), wet_n)
            chk(f"init_AB{tr}", "tr_xy", p + "init_AB.tr_xy", we2)
            chk(f"init_AB{tr}", "tr_z", p + "init_AB.tr_z", wet_nl)
            chk(f"init_AB{tr}", "edge_up_dn_grad", p + "init_AB.edge_up_dn_grad")
            orc.call("adv_tracers_ale", tr)
            chk(f"adv{tr}", "fct_LO", p + "adv.fct_LO", wet_n)
            chk(f"adv{tr}", "fct_ttf_max", p + "adv.fct_ttf_max", wet_n); chk(f"adv{tr}", "fct_ttf_min", p + "adv.fct_ttf_min", wet_n)
            chk(f"adv{tr}", "fct_plus", p + "adv.fct_plus", wet_n); chk(f"adv{tr}", "fct_minus", p + "adv.fct_minus", wet_n)
            chk(f"adv{tr}", "adv_flux_hor", p + "adv.adv_flux_hor"); chk(f"adv{tr}", "adv_flux_ver", p + "adv.adv_flux_ver", wet_nl)
            chk(f"adv{tr}", "del_ttf_advhoriz", p + "adv.del_ttf_advhoriz", wet_n); chk(f"adv{tr}", "del_ttf_advvert", p + "adv.del_ttf_advvert", wet_n)
            chk(f"adv{tr}", "del_ttf", p + "adv.del_ttf", wet_n)
            orc.call("diff_tracers_ale", tr)
            chk(f"diff{tr}", "del_ttf", p + "diff.del_ttf", wet_n)
            if toy:
                orc.call("relax_zonal_temp")              # after every tracer, always on tracer 1
            allok &= report(f"diff{tr}:tr_arr", orc.get("tr_arr").reshape(2, -1, nlm1)[tr - 1], g(p + "end.tr_arr"), wet_n)
        if par.Fer_GM:
            orc.call("bolus_remove")
        orc.call("salinity_clamp")
        orc.call("update_thickness_ale")
        for f in ("hnode", "helem", "zbar_3d_n", "Z_3d_n"):
            chk("update_thickness_ale", f, "update_thickness_ale." + f)
        chk("out", "tr_arr", "out.tr_arr"); chk("out", "UV", "out.UV", we2); chk("out", "eta_n", "out.eta_n")
    print("ALL BITWISE" if allok else "DIFFERENCES FOUND")
    return 0 if allok else 1


if __name__ == "__main__":
    sys.exit(main())
