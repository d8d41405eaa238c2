"""Namelist-level options of the hot path (host-side mirror of namelist.config / namelist.oce keys
read by oce_timestep_ale; defaults = config/namelist.oce of the reference)."""
from . import _lib
from .mesh import WHICH_ALE


def make_params(dt=900.0, which_ale="zstar", use_partial_cell=True, state_equation=1, num_tracers=2,
                mix_scheme="PP", with_diffusion=True, toy_soufflet=False, K_hor=3000.0, A_ver=1.0e-4, K_ver=1.0e-5,
                cyclic_length_deg=360.0, w_split=False, use_instabmix=True, use_windmix=False, solver_x0_order=3,
                Fer_GM=False, K_GM_max=2000.0, K_GM_min=2.0, K_GM_bvref=2, K_GM_rampmax=-1.0, K_GM_rampmin=-1.0,
                K_GM_resscalorder=1.0, scaling_Ferreira=False, scaling_resolution=True, scaling_FESOM14=False, scaling_Rossby=False, Redi=False,
                visc_sh_limit=5.0e-3, diff_sh_limit=5.0e-3, Ricr=0.3, concv=1.6,
                gamma0=0.003, gamma1=0.1, gamma2=0.285, easy_bs_return=1.5, C_d=0.0025, w_max_cfl=1.0, use_sw_pene=False, visc_option=5, tra_adv_ver="QR4C", tra_adv_hor="MFCT", Kv0_const=True,
                solver_precond=1, solver_xinv_its=0, tra_adv_lim="FCT", Leith_c=0.05, Div_c=0.5, which_pgf="shchepetkin", use_momix=False, momix_lat=-50.0, momix_kv=0.01, mom_adv=2, use_kpp_nonlclflx=False, ref_sss_local=True, ref_sss=34.0, double_diffusion=False, smooth_bh_tra=False, use_floatice=False, l_mslp=False, use_global_tides=False, max_ice_loading=5.0, clim_relax=0.0, SPP=False, Sice=4.0, min_hnode=0.5, lzstar_lev=4,
                c_back=0.1, K_back=600.0, uke_scaling=True, uke_scaling_factor=1.0, rosb_dis=1.0, smooth_back=2, smooth_dis=2, smooth_back_tend=4, scale_area=5.8e9,
                use_cavity=False, use_density_ref=None, density_ref_T=2.0, density_ref_S=34.0, use_cavity_partial_cell=False):
    p = _lib.Params()
    p.dt = dt
    p.which_ale = WHICH_ALE[which_ale]
    p.use_partial_cell = int(use_partial_cell)
    p.state_equation = state_equation
    p.num_tracers = num_tracers
    p.mom_adv = int(mom_adv)         # 2 scalar control volumes (default), 3 vector invariant (linfs only)
    p.visc_option = int(visc_option)
    p.i_vert_visc = 1
    p.i_vert_diff = 1
    p.w_split = int(w_split)
    p.mix_scheme = {"KPP": 1, "PP": 2, "none": 0}[mix_scheme]
    p.use_instabmix = int(use_instabmix)
    p.use_windmix = int(use_windmix)
    p.windmix_nl = 2
    p.toy_soufflet = int(toy_soufflet)
    p.alpha, p.theta, p.epsilon = 1.0, 1.0, 0.1
    p.C_d, p.A_ver, p.K_ver, p.K_hor = C_d, A_ver, K_ver, K_hor
    p.gamma0, p.gamma1, p.gamma2, p.easy_bs_return = gamma0, gamma1, gamma2, easy_bs_return
    p.w_max_cfl = w_max_cfl
    p.tra_adv_ph, p.tra_adv_pv = 1.0, 1.0
    p.instabmix_kv, p.windmix_kv = 0.1, 1.0e-3
    p.cyclic_length = cyclic_length_deg * 3.14159265358979 / 180.0
    p.with_diffusion = int(with_diffusion)
    p.solver_x0_order = int(solver_x0_order)
    p.Fer_GM = int(Fer_GM)
    p.K_GM_max, p.K_GM_min, p.K_GM_bvref = K_GM_max, K_GM_min, int(K_GM_bvref)
    p.K_GM_rampmax, p.K_GM_rampmin, p.K_GM_resscalorder = K_GM_rampmax, K_GM_rampmin, K_GM_resscalorder
    p.scaling_Ferreira, p.scaling_Rossby = int(scaling_Ferreira), int(scaling_Rossby)
    p.scaling_resolution, p.scaling_FESOM14 = int(scaling_resolution), int(scaling_FESOM14)
    p.Redi = int(Redi)
    p.use_sw_pene = int(use_sw_pene)
    p.tra_adv_ver = {"QR4C": 0, "CDIFF": 1, "UPW1": 2, "PPM": 3}[tra_adv_ver]
    p.tra_adv_hor = {"MFCT": 0, "MUSCL": 1, "UPW1": 2}[tra_adv_hor]
    p.Kv0_const = int(Kv0_const)
    p.solver_precond, p.solver_xinv_its = int(solver_precond), int(solver_xinv_its)
    p.tra_adv_lim = {"FCT": 0, "NON": 1}[tra_adv_lim]
    p.Leith_c, p.Div_c = Leith_c, Div_c              # config/namelist.oce:8-9
    p.which_pgf = {"shchepetkin": 0, "cubicspline": 1, "nemo": 2, "easypgf": 3, "sergey": 4}.get(which_pgf, -1)  # oce_modules.F90:172
    p.use_momix, p.momix_lat, p.momix_kv = int(use_momix), momix_lat, momix_kv   # config/namelist.oce:48-50
    # ocean_setup (src/oce_setup_step.F90:42-47): unless which_ALE = 'linfs' the reference sets ref_sss_local = .false., ref_sss = 0 ("this will force the
    # virtual salinity flux to be zero"), whatever the namelist says -- the salt part of the KPP non-local transport vanishes with it
    p.double_diffusion = int(double_diffusion)         # config/namelist.oce:64
    p.smooth_bh_tra = int(smooth_bh_tra)               # config/namelist.oce:57
    p.use_floatice, p.l_mslp, p.use_global_tides, p.max_ice_loading = int(use_floatice and which_ale != "linfs"), int(l_mslp), int(use_global_tides), max_ice_loading
    p.clim_relax = clim_relax                          # config/namelist.oce:70
    p.min_hnode, p.lzstar_lev = min_hnode, lzstar_lev  # gen_modules_config.F90:60,64 (zlevel)
    p.SPP, p.Sice = int(SPP), Sice                     # config/namelist.oce:26, src/ice_modules.F90:132
    # visc_option = 8 (src/oce_modules.F90:34-41; scale_area: config/namelist.oce)
    p.c_back, p.K_back, p.uke_scaling_factor, p.rosb_dis, p.scale_area = c_back, K_back, uke_scaling_factor, rosb_dis, scale_area
    p.uke_scaling, p.smooth_back, p.smooth_dis, p.smooth_back_tend = int(uke_scaling), int(smooth_back), int(smooth_dis), int(smooth_back_tend)
    p.use_cavity = int(use_cavity)
    p.use_cavity_partial_cell = int(use_cavity_partial_cell)
    p.use_density_ref = int(use_cavity if use_density_ref is None else use_density_ref)      # (ocean_setup switches it on with cavities, oce_setup_step.F90:122)
    p.density_ref_T, p.density_ref_S = density_ref_T, density_ref_S
    linfs = (which_ale == "linfs")
    p.use_kpp_nonlclflx, p.ref_sss_local, p.ref_sss = int(use_kpp_nonlclflx), int(ref_sss_local and linfs), (ref_sss if linfs else 0.0)   # config/namelist.oce:71-72, :83
    p.visc_sh_limit, p.diff_sh_limit, p.Ricr, p.concv = visc_sh_limit, diff_sh_limit, Ricr, concv
    return p
