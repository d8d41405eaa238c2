"""Shared sequence of the step (order of src/oce_ale.F90:2556-2767) and the fields each routine writes.
Used by the GPU parity tests (HIP path vs CPU oracle on identical inputs) and by the golden checks."""
import numpy as np

# (routine, arg, [fields written]) ; field names are identical in the oracle and in libfesom_gpu
DYN_PRE = [
    ("compute_vel_nodes", 0, ["Unode"]),
    ("pressure_bv", 0, ["density_m_rho0", "bvfreq", "MLD1", "MLD2"]),
    ("pressure_force", 0, ["pgf_x", "pgf_y"]),
    ("sw_alpha_beta", 0, ["sw_alpha", "sw_beta"]),
    ("compute_sigma_xy", 0, ["sigma_xy"]),
    ("compute_neutral_slope", 0, ["neutral_slope", "slope_tapered"]),
    ("mixing_pp", 0, []),
    ("mo_convect", 0, ["Av", "Kv"]),
    ("compute_vel_rhs", 0, ["UV_rhs", "UV_rhsAB"]),
    ("viscosity_filter", 0, ["UV_rhs"]),
    ("impl_vert_visc_ale", 0, ["UV_rhs"]),
    ("update_stiff_mat_ale", 0, ["ssh_values"]),
    ("compute_ssh_rhs_ale", 0, ["ssh_rhs"]),
    ("solve_ssh", 0, ["d_eta"]),
    ("update_vel", 0, ["UV", "eta_n"]),
    ("compute_hbar_ale", 0, []),
    ("eta_update", 0, ["hbar", "hbar_old", "ssh_rhs_old", "dhe", "eta_n"]),
    ("vert_vel_ale", 0, ["Wvel", "Wvel_e", "Wvel_i", "CFL_z", "hnode_new"]),
]


def tracer_chain(tr):
    return [
        ("init_tracers_AB", tr, ["tr_arr_old", "tr_z", "edge_up_dn_grad", "tr_xy"]),
        ("adv_tracers_ale", tr, ["fct_LO", "fct_ttf_max", "fct_ttf_min", "fct_plus", "fct_minus", "adv_flux_hor", "adv_flux_ver"]),
        ("diff_tracers_ale", tr, ["del_ttf", "tr_arr", "tr_arr_old"]),
    ]


TAIL = [("salinity_clamp", 0, ["tr_arr"]), ("update_thickness_ale", 0, ["hnode", "helem", "zbar_3d_n", "Z_3d_n"])]

# fields whose values go through a libm call that differs between glibc and the device library
ULP_FIELDS = {"slope_tapered": 1e-12, "mixlength": 1e-13}    # mixlength: exp of the device vs glibc inside the Newton iteration of pmlktmo


GM_BEFORE_W = [("init_Redi_GM", 0, ["fer_K", "fer_c"]), ("fer_solve_Gamma", 0, ["fer_gamma"]), ("fer_gamma2vel", 0, ["fer_UV"])]
GM_AFTER_W = [("fer_wvel", 0, ["fer_Wvel"]), ("bolus_add", 0, ["UV", "Wvel", "Wvel_e"])]


KPP_MIX = [("mixing_kpp", 0, ["kpp_hbl", "kpp_ghats", "kpp_blmc1", "kpp_blmc2", "kpp_blmc3", "kpp_viscA", "kpp_Kv1", "kpp_Kv2"]),
           ("mo_convect", 0, ["Av", "Kv"])]


def full_chain(ntr=2, gm=False, redi=False, kpp=False):
    ch = []
    for item in DYN_PRE:
        if kpp and item[0] == "mixing_pp":
            ch += KPP_MIX
            continue
        if kpp and item[0] == "mo_convect":
            continue
        if redi and not gm and item[0] == "vert_vel_ale":
            ch.append(("init_Redi_GM", 0, ["Ki"]))
        if gm and item[0] == "vert_vel_ale":
            ch += [(r, a, f + (["Ki"] if redi and r == "init_Redi_GM" else [])) for r, a, f in GM_BEFORE_W]   # oce_ale.F90:2729-2739
        ch.append(item)
        if gm and item[0] == "vert_vel_ale":
            ch += GM_AFTER_W                       # fer_Wvel is part of vert_vel_ale; bolus added around the tracer loop
    for tr in range(1, ntr + 1):
        ch += tracer_chain(tr)
    if gm:
        ch.append(("bolus_remove", 0, ["UV", "Wvel", "Wvel_e"]))
    return ch + TAIL


def bits_equal(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64).ravel()
    b = np.ascontiguousarray(b, dtype=np.float64).ravel()
    return (a.view(np.int64) == b.view(np.int64)) | ((a == 0) & (b == 0)) | (np.isnan(a) & np.isnan(b))     # (NaN: sign/payload of 0/0 are the machine's)


def compare(name, a, b):
    """returns (ok, message).  Bitwise, except ULP_FIELDS (relative tolerance)."""
    eq = bits_equal(a, b)
    if eq.all():
        return True, f"{name}: bitwise"
    a = a.ravel(); b = b.ravel()
    bad = ~eq
    err = np.abs(a[bad] - b[bad])
    rel = (err / (np.abs(b[bad]) + 1e-300)).max()
    msg = f"{name}: {int(bad.sum())}/{a.size} differ, max abs {err.max():.3e}, max rel {rel:.3e}"
    if name in ULP_FIELDS and rel <= ULP_FIELDS[name]:
        return True, msg + " (within libm tolerance)"
    return False, msg
