#!/usr/bin/env python3
"""Run the reference oracle binary (oracle/_ref/fesom_oracle.x) on a named configuration.

Test infrastructure only.  Writes namelists (values = the reference's
config/namelist.config + config/namelist.oce defaults with the overrides of
setups/test_souf/setup.yml:9-49 where noted), initial-condition files and a
1-rank partition if needed into oracle/_ref/run_<cfg>_<np>/ and starts
mpiexec.  Usage: run_ref.py CFG NP NSTEPS [mode=step|replay] [dump=1,2,..] [mean]
"""
import os, sys, subprocess, shutil

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, REPO)
OUT = os.path.join(REPO, "oracle", "_ref")
MESHES = os.path.join(REPO, "tests", "golden", "meshes")

CONFIG_TMPL = """&modelname
runid='fesom'
/
&timestep
step_per_day={step_per_day}
run_length=1
run_length_unit='d'
/
&clockinit
timenew=0.0
daynew=1
yearnew=1958
/
&paths
MeshPath='{meshpath}/'
ClimateDataPath='./'
ResultPath='./'
/
&restart_log
restart_length=1
restart_length_unit='y'
logfile_outfreq=100000
/
&ale_def
which_ALE='{which_ale}'
use_partial_cell={use_partial_cell}
min_hnode={min_hnode}
/
&geometry
cartesian=.false.
fplane=.false.
cyclic_length={cyclic_length}
rotated_grid={rotated_grid}
force_rotation={force_rotation}
alphaEuler=50.
betaEuler=15.
gammaEuler=-90.
/
&calendar
include_fleapyear=.false.
/
&run_config
use_ice=.false.
use_cavity={use_cavity}
use_cavity_partial_cell={use_cavity_partial_cell}
use_floatice={use_floatice}
use_sw_pene={use_sw_pene}
toy_ocean={toy_ocean}
which_toy='{which_toy}'
flag_warn_cflz=.false.
/
"""

OCE_TMPL = """&oce_dyn
state_equation={state_equation}
C_d=0.0025
gamma0=0.003
gamma1=0.1
gamma2=0.285
Div_c=.5
Leith_c=.05
visc_option={visc_option}
which_pgf='{which_pgf}'
easy_bs_return=1.5
A_ver=1.e-4
scale_area=5.8e9
mom_adv={mom_adv}
free_slip=.false.
i_vert_visc=.true.
w_split={w_split}
w_max_cfl={w_max_cfl}
SPP={SPP}
Fer_GM={fer_gm}
K_GM_max=2000.0
K_GM_min=2.0
K_GM_bvref=2
K_GM_rampmax=-1.0
K_GM_rampmin=-1.0
K_GM_resscalorder=1
scaling_Ferreira=.false.
scaling_Rossby={scaling_Rossby}
scaling_resolution=.true.
scaling_FESOM14=.false.
Redi={redi}
visc_sh_limit=5.0e-3
mix_scheme='{mix_scheme}'
Ricr=0.3
concv=1.6
use_density_ref={use_density_ref}
/
&oce_tra
use_momix={use_momix}
momix_lat=-50.0
momix_kv=0.01
use_instabmix=.true.
instabmix_kv=0.1
use_windmix=.false.
windmix_kv=1.e-3
windmix_nl=2
smooth_bh_tra={smooth_bh_tra}
gamma0_tra=0.0005
gamma1_tra=0.0125
gamma2_tra=0.
use_kpp_nonlclflx={use_kpp_nonlclflx}
diff_sh_limit=5.0e-3
Kv0_const={Kv0_const}
double_diffusion={double_diffusion}
K_ver=1.0e-5
K_hor={k_hor}
surf_relax_T=0.0
surf_relax_S={surf_relax_s}
balance_salt_water={balance_salt_water}
clim_relax={clim_relax}
ref_sss_local=.true.
ref_sss=34.
i_vert_diff=.true.
tra_adv_hor='{tra_adv_hor}'
tra_adv_ver='{tra_adv_ver}'
tra_adv_lim='{tra_adv_lim}'
tra_adv_ph=1.
tra_adv_pv=1.
num_tracers=2
tracer_ID=0,1
/
&oce_init3d
n_ic3d=2
idlist=1,0
filelist='ic_S.bin','ic_T.bin'
varlist='salt','temp'
t_insitu=.false.
/
"""

_CAV = dict(mesh="pi_cavity", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
            rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
            fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
            balance_salt_water=".true.", use_cavity=".true.", synth_forcing=True)
CFGS = {
    # ice-shelf cavity: the pi mesh with a synthetic draft (tests/golden/make_cavity_mesh.py -> meshes/pi_cavity: cavity_elvls / nlvls / depth), use_cavity=.true.
    "pi_pp_cavity": dict(mesh="pi_cavity", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                         rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                         fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                         balance_salt_water=".true.", use_cavity=".true.", synth_forcing=True),
    # cavity variants on the same mesh: partial cells at the shelf base, easypgf, zlevel, linfs (full cells / partial cells / partial cells at the base with 'sergey' and 'shchepetkin')
    "pi_pp_cavity_pc": dict(_CAV, use_cavity_partial_cell=".true."),
    "pi_pp_cavity_easypgf": dict(_CAV, which_pgf="easypgf"),
    "pi_pp_zlevel_cavity": dict(_CAV, which_ale="zlevel"),
    "pi_pp_linfs_cavity": dict(_CAV, which_ale="linfs", use_partial_cell=".false."),
    "pi_pp_linfs_pc_cavity": dict(_CAV, which_ale="linfs"),
    "pi_pp_linfs_cavity_sergey": dict(_CAV, which_ale="linfs", use_cavity_partial_cell=".true.", which_pgf="sergey"),
    "pi_pp_linfs_cavity_pc_shch": dict(_CAV, which_ale="linfs", use_cavity_partial_cell=".true."),
    "pi_pp_linfs_easypgf_cavity": dict(_CAV, which_ale="linfs", which_pgf="easypgf"),
    "pi_pp_cavity_cubicspline": dict(_CAV, which_pgf="cubicspline"),
    "pi_pp_linfs_cubic_cavity": dict(_CAV, which_ale="linfs", which_pgf="cubicspline"),
    "pi_pp_linfs_nemo_cavity": dict(_CAV, which_ale="linfs", which_pgf="nemo"),
    # the same cavity mesh under the reference's default physics (KPP + GM + Redi)
    "pi_default_cavity": dict(mesh="pi_cavity", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                              rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                              fer_gm=".true.", redi=".true.", mix_scheme="KPP", k_hor="3000.", surf_relax_s="1.929e-06",
                              balance_salt_water=".true.", use_cavity=".true.", synth_forcing=True),
    # use_density_ref=.true. without cavities: density_m_rho0 against the profile of init_ref_density (T=2, S=34) instead of density_0
    "pi_pp_dref": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                       rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                       fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                       balance_salt_water=".true.", use_density_ref=".true.", synth_forcing=True),
    # pi mesh, 47 layers, zstar + partial cells, JM EOS, PP mixing, no GM/Redi (round-1 closure config)
    "pi_pp": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                  rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                  fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                  balance_salt_water=".true."),
    # the same with the Gent-McWilliams bolus velocities (Ferrari et al. 2010) switched on
    "pi_pp_gm": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                     rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                     fer_gm=".true.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                     balance_salt_water=".true."),
    # ... and with isoneutral (Redi) diffusion on top
    "pi_pp_gm_redi": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                          rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                          fer_gm=".true.", redi=".true.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                          balance_salt_water=".true."),
    # pi mesh with the reference's default physics (KPP + GM + Redi) and the harness's analytic surface forcing
    "pi_default": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                       rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                       fer_gm=".true.", redi=".true.", mix_scheme="KPP", k_hor="3000.", surf_relax_s="1.929e-06",
                       balance_salt_water=".true.", synth_forcing=True),
    # scaling_Rossby = .true.: K_GM cut off where the mesh resolves the first baroclinic Rossby radius
    "pi_default_rossby": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                       rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                       fer_gm=".true.", redi=".true.", mix_scheme="KPP", k_hor="3000.", surf_relax_s="1.929e-06",
                       balance_salt_water=".true.", synth_forcing=True, scaling_Rossby=".true."),
    # the physics of the shipped config/namelist.oce in full: KPP + GM + Redi + use_momix (analytic ice state for mo_length)
    "pi_default_momix": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                       rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                       fer_gm=".true.", redi=".true.", mix_scheme="KPP", k_hor="3000.", surf_relax_s="1.929e-06",
                       balance_salt_water=".true.", synth_forcing=True, use_momix=".true."),
    # KPP with its non-local transport terms (use_kpp_nonlclflx = .true., oce_ale_tracer.F90:688-724)
    "pi_kpp_nonlcl": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                       rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                       fer_gm=".false.", redi=".false.", mix_scheme="KPP", k_hor="3000.", surf_relax_s="1.929e-06",
                       balance_salt_water=".true.", synth_forcing=True, use_kpp_nonlclflx=".true."),
    # the surface potentials of compute_vel_rhs beside g*eta (oce_ale_vel_rhs.F90:52-76): floating ice load, atmospheric pressure, tidal potential (analytic fields)
    "pi_pp_surfpot": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                       rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                       fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                       balance_salt_water=".true.", synth_forcing=True, use_floatice=".true.", tides=True, mslp=True),     # (use_global_tides is in no namelist of the reference: the harness sets the module variable)
    # which_ALE = 'zlevel': the ssh change in the surface layer only (oce_ale.F90:630-690, 817-943, 1830-2023); pi has 4-layer columns, whose partial bottom cell
    # sends every rising step through the "return to zlevel" branch
    "pi_pp_zlevel": dict(mesh="pi", step_per_day=96, which_ale="zlevel", use_partial_cell=".true.", cyclic_length=360,
                       rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                       fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                       balance_salt_water=".true.", synth_forcing=True),
    # SPP = .true.: salt plume parameterization at the head of solve_tracers_ale (src/oce_spp.F90), linfs as its header asks; analytic thdgr / S_oc_array
    "pi_pp_linfs_spp": dict(mesh="pi", step_per_day=96, which_ale="linfs", use_partial_cell=".false.", cyclic_length=360,
                       rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                       fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                       balance_salt_water=".true.", synth_forcing=True, SPP=".true."),
    # clim_relax > 0: relax_to_clim after diff_tracers_ale (oce_tracer_mod.F90:86-121), analytic nodal rate relax2clim
    "pi_pp_climrelax": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                       rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                       fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                       balance_salt_water=".true.", synth_forcing=True, clim_relax="1.1574e-6"),
    # smooth_bh_tra = .true.: biharmonic tracer diffusion as a filter (diff_part_bh, oce_ale_tracer.F90:1081-1150)
    "pi_pp_bhtra": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                       rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                       fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                       balance_salt_water=".true.", synth_forcing=True, smooth_bh_tra=".true."),
    # KPP with double diffusion (ddmix, oce_ale_mixing_kpp.F90:857-934)
    "pi_kpp_dd": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                       rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                       fer_gm=".false.", redi=".false.", mix_scheme="KPP", k_hor="3000.", surf_relax_s="1.929e-06",
                       balance_salt_water=".true.", synth_forcing=True, double_diffusion=".true."),
    "pi_kpp_nonlcl_linfs": dict(mesh="pi", step_per_day=96, which_ale="linfs", use_partial_cell=".false.", cyclic_length=360,
                       rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                       fer_gm=".false.", redi=".false.", mix_scheme="KPP", k_hor="3000.", surf_relax_s="1.929e-06",
                       balance_salt_water=".true.", synth_forcing=True, use_kpp_nonlclflx=".true."),
    # PP + w_split: vertical velocity split into an explicit and an implicit part where CFL_z > w_max_cfl (threshold lowered so that
    # the split is active on pi from the first steps)
    "pi_pp_wsplit": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                         rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                         fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                         balance_salt_water=".true.", synth_forcing=True, w_split=".true.", w_max_cfl="0.0003"),
    # the default physics with short-wave penetration (use_sw_pene=.true., the default of namelist.config), sw_3d from the harness
    "pi_default_sw": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                          rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                          fer_gm=".true.", redi=".true.", mix_scheme="KPP", k_hor="3000.", surf_relax_s="1.929e-06",
                          balance_salt_water=".true.", synth_forcing=True, use_sw_pene=".true."),
    # the biharmonic viscosity filters instead of the easy backscatter: visc_option = 6 (visc_filt_bilapl), 7 (visc_filt_bidiff)
    "pi_pp_visc6": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                        rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                        fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                        balance_salt_water=".true.", synth_forcing=True, visc_option=6),
    # tra_adv_lim = 'NON': the high-order tracer fluxes without the FCT limiter (oce_adv_tra_driver.F90:137-197)
    "pi_pp_non": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                      rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                      fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                      balance_salt_water=".true.", synth_forcing=True, tra_adv_lim="NON"),
    # tra_adv_lim = 'NON' with w_split: the implicit part of the vertical velocity inside the diffusion solve (do_wimpl, oce_ale_tracer.F90:424)
    "pi_pp_non_wsplit": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                      rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                      fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                      balance_salt_water=".true.", synth_forcing=True, tra_adv_lim="NON", w_split=".true.", w_max_cfl="0.0003"),
    # visc_option = 1, 2, 3: Leith viscosity (h_viscosity_leith) with the harmonic / harmonic + biharmonic background / biharmonic filter
    "pi_pp_visc1": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                        rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                        fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                        balance_salt_water=".true.", synth_forcing=True, visc_option=1),
    "pi_pp_visc2": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                        rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                        fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                        balance_salt_water=".true.", synth_forcing=True, visc_option=2),
    "pi_pp_visc3": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                        rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                        fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                        balance_salt_water=".true.", synth_forcing=True, visc_option=3),
    # visc_option = 4: visc_filt_biharm(1), the biharmonic "third-order-upwind-like" filter (src/oce_dyn.F90:275-372)
    # use_momix = .true. (the shipped config/namelist.oce:48): Monin-Obukhov mixing of mo_convect south of 50 S, with an analytic ice state
    "pi_pp_momix": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                        rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                        fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                        balance_salt_water=".true.", synth_forcing=True, use_momix=".true."),
    # mom_adv = 3: compute_vel_rhs_vinv (src/oce_vel_rhs_vinv.F90:104-322), linear free surface with full cells (hpressure exists only there)
    "pi_pp_linfs_vinv": dict(mesh="pi", step_per_day=96, which_ale="linfs", use_partial_cell=".false.", cyclic_length=360,
                        rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                        fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                        balance_salt_water=".true.", synth_forcing=True, mom_adv=3),
    # mom_adv = 3 with zstar: the reference leaves hpressure at the zeros of array_setup there (no baroclinic pressure term in compute_vel_rhs_vinv)
    "pi_pp_vinv": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                        rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                        fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                        balance_salt_water=".true.", synth_forcing=True, mom_adv=3),
    # linfs with partial cells and which_pgf = 'cubicspline': pressure_force_4_linfs_cubicspline (src/oce_ale_pressure_bv.F90:1252-1444)
    "pi_pp_linfs_cubic": dict(mesh="pi", step_per_day=96, which_ale="linfs", use_partial_cell=".true.", cyclic_length=360,
                        rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                        fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                        balance_salt_water=".true.", synth_forcing=True, which_pgf="cubicspline"),
    # linfs with partial cells and which_pgf = 'nemo': pressure_force_4_linfs_nemo (src/oce_ale_pressure_bv.F90:479-635)
    "pi_pp_linfs_nemo": dict(mesh="pi", step_per_day=96, which_ale="linfs", use_partial_cell=".true.", cyclic_length=360,
                        rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                        fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                        balance_salt_water=".true.", synth_forcing=True, which_pgf="nemo"),
    # which_pgf = 'easypgf' with zstar: pressure_force_4_zxxxx_easypgf (src/oce_ale_pressure_bv.F90:2116-2546)
    "pi_pp_easypgf": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                        rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                        fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                        balance_salt_water=".true.", synth_forcing=True, which_pgf="easypgf"),
    "pi_pp_linfs_easypgf": dict(mesh="pi", step_per_day=96, which_ale="linfs", use_partial_cell=".true.", cyclic_length=360,
                        rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                        fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                        balance_salt_water=".true.", synth_forcing=True, which_pgf="easypgf"),
    # which_pgf = 'cubicspline': pressure_force_4_zxxxx_cubicspline (src/oce_ale_pressure_bv.F90:1697-1866)
    "pi_pp_cubicspline": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                        rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                        fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                        balance_salt_water=".true.", synth_forcing=True, which_pgf="cubicspline"),
    # linfs with partial cells: pressure_force_4_linfs_shchepetkin (src/oce_ale_pressure_bv.F90:647-891)
    "pi_pp_linfs_pc": dict(mesh="pi", step_per_day=96, which_ale="linfs", use_partial_cell=".true.", cyclic_length=360,
                        rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                        fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                        balance_salt_water=".true.", synth_forcing=True),
    "pi_pp_visc4": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                        rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                        fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                        balance_salt_water=".true.", synth_forcing=True, visc_option=4),
    # visc_option = 8: backscatter_coef + visc_filt_dbcksc + uke_update (src/oce_dyn.F90:806-1152), prognostic unresolved kinetic energy
    "pi_pp_visc8": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                        rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                        fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                        balance_salt_water=".true.", synth_forcing=True, visc_option=8),
    "pi_pp_visc7": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                        rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                        fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                        balance_salt_water=".true.", synth_forcing=True, visc_option=7),
    # vertical high-order advection variants under FCT: tra_adv_ver = 'CDIFF', 'UPW1'
    "pi_pp_cdiff": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                        rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                        fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                        balance_salt_water=".true.", synth_forcing=True, tra_adv_ver="CDIFF"),
    "pi_pp_upw1v": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                        rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                        fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                        balance_salt_water=".true.", synth_forcing=True, tra_adv_ver="UPW1", w_split=".true.", w_max_cfl="0.0003"),
    "pi_pp_ppm": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                      rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                      fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                      balance_salt_water=".true.", synth_forcing=True, tra_adv_ver="PPM"),
    # Kv0_const=.false.: latitude/depth dependent background diffusivity (Kv0_background_qiang) in PP and in KPP
    "pi_pp_kv0": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                      rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                      fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                      balance_salt_water=".true.", synth_forcing=True, Kv0_const=".false."),
    "pi_kpp_kv0": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                       rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                       fer_gm=".false.", redi=".false.", mix_scheme="KPP", k_hor="3000.", surf_relax_s="1.929e-06",
                       balance_salt_water=".true.", synth_forcing=True, Kv0_const=".false."),
    # horizontal high-order advection variants under FCT: tra_adv_hor = 'MUSCL', 'UPW1'
    "pi_pp_muscl": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                        rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                        fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                        balance_salt_water=".true.", synth_forcing=True, tra_adv_hor="MUSCL"),
    "pi_pp_upw1h": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                        rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                        fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="3000.", surf_relax_s="1.929e-06",
                        balance_salt_water=".true.", synth_forcing=True, tra_adv_hor="UPW1", tra_adv_ver="CDIFF"),
    # KPP alone (no GM/Redi) with the same forcing
    "pi_kpp": dict(mesh="pi", step_per_day=96, which_ale="zstar", use_partial_cell=".true.", cyclic_length=360,
                   rotated_grid=".true.", force_rotation=".true.", toy_ocean=".false.", state_equation=1,
                   fer_gm=".false.", redi=".false.", mix_scheme="KPP", k_hor="3000.", surf_relax_s="1.929e-06",
                   balance_salt_water=".true.", synth_forcing=True),
    # Soufflet channel = setups/test_souf/setup.yml overrides
    "souf": dict(mesh="soufflet", step_per_day=72, which_ale="zstar", use_partial_cell=".true.", cyclic_length=4.5,
                 rotated_grid=".false.", force_rotation=".false.", toy_ocean=".true.", state_equation=0,
                 fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="10", surf_relax_s="0.0",
                 balance_salt_water=".false."),
    "souf_linfs": dict(mesh="soufflet", step_per_day=72, which_ale="linfs", use_partial_cell=".false.", cyclic_length=4.5,
                       rotated_grid=".false.", force_rotation=".false.", toy_ocean=".true.", state_equation=0,
                       fer_gm=".false.", redi=".false.", mix_scheme="PP", k_hor="10", surf_relax_s="0.0",
                       balance_salt_water=".false."),
}


def refined_case(levels, npes, base="pi_pp", workdir=None):
    """registers configuration `<base>_r<levels>`: the pi mesh refined `levels` times (fesom2_amd.mesh_refine), with the edge files
    and a `dist_<npes>` partition written in the reference's formats by fesom2_amd.partition_io -- so that the REFERENCE runs on
    the same large mesh as the GPU path (CPU baseline / parity at sizes beyond pi)."""
    import tempfile
    from fesom2_amd import mesh_refine, partition_io
    d = os.path.join(workdir or tempfile.gettempdir(), f"fesom_pi_r{levels}")
    if not os.path.exists(os.path.join(d, "edgenum.out")):
        mesh_refine.refine(os.path.join(MESHES, "pi"), d, levels)
        partition_io.write_edge_files(d)
    for n in sorted({npes, 1}):
        if n > 1 and not os.path.isdir(os.path.join(d, f"dist_{n}")):
            partition_io.write_dist(d, n)
    name = f"{base}_r{levels}"
    CFGS[name] = dict(CFGS[base], mesh=d)
    return name, d


def channel_case(levels, npes, layers=47, base="souf", workdir=None):
    """registers configuration `chan_r<levels>_<layers>`: the Soufflet channel of the reference's CI case refined `levels` times
    (fesom2_amd.channel_mesh: cyclic refinement, `layers` stretched layers, dt = 1200 s / 2**levels) with edge files and a
    `dist_<npes>` partition in the reference's formats -- the CORE2-class workload (BASELINE config #3) that the REFERENCE runs too."""
    import tempfile
    from fesom2_amd import channel_mesh, partition_io
    d = os.path.join(workdir or tempfile.gettempdir(), f"fesom_chan_r{levels}_{layers}")
    kw = dict(force_rotation=False, cyclic_length_deg=channel_mesh.CYCLIC_DEG)
    if not os.path.exists(os.path.join(d, "edgenum.out")):
        channel_mesh.build(os.path.join(MESHES, "soufflet"), d, levels, layers)
        partition_io.write_edge_files(d, **kw)
    for n in sorted({npes, 1}):
        if n > 1 and not os.path.isdir(os.path.join(d, f"dist_{n}")):
            partition_io.write_dist(d, n, **kw)
    name = f"chan_r{levels}_{layers}"
    CFGS[name] = dict(CFGS[base], mesh=d, step_per_day=int(round(86400.0 / channel_mesh.dt_for(levels))))
    return name, d


def basin_case(levels, npes, layers=47, workdir=None):
    """registers configuration `basin_r<levels>_<layers>`: the channel geometry with the analytic bathymetry of fesom2_amd.channel_mesh
    (build_basin) and the reference's default physics (pi_default: KPP + GM + Redi, JM EOS, analytic forcing), edge files and `dist_<npes>` in
    the reference's formats -- the default-physics CORE2-class workload that the REFERENCE runs too."""
    import tempfile
    from fesom2_amd import channel_mesh, partition_io
    d = os.path.join(workdir or tempfile.gettempdir(), f"fesom_basin_r{levels}_{layers}")
    kw = dict(force_rotation=False, cyclic_length_deg=channel_mesh.CYCLIC_DEG)
    if not os.path.exists(os.path.join(d, "nlvls.out")):
        channel_mesh.build_basin(os.path.join(MESHES, "soufflet"), d, levels, layers)
    if not os.path.exists(os.path.join(d, "edgenum.out")):
        partition_io.write_edge_files(d, **kw)
    for n in sorted({npes, 1}):
        if n > 1 and not os.path.isdir(os.path.join(d, f"dist_{n}")):
            partition_io.write_dist(d, n, **kw)
    name = f"basin_r{levels}_{layers}"
    CFGS[name] = dict(CFGS["pi_default"], mesh=d, step_per_day=int(round(86400.0 / channel_mesh.dt_for(levels))), cyclic_length=4.5,
                      rotated_grid=".false.", force_rotation=".false.", toy_ocean=".true.", which_toy="basin")     # (toy_ocean only skips read_mesh's rotation check: see driver.F90)
    return name, d


def prepare(cfg, np_, tag=""):
    c = CFGS[cfg]
    rd = os.path.join(OUT, f"run_{cfg}_{np_}{tag}")
    os.makedirs(os.path.join(rd, "dumps"), exist_ok=True)
    meshdir = c["mesh"] if os.path.isabs(c["mesh"]) else os.path.join(MESHES, c["mesh"])
    if np_ == 1 and not os.path.isdir(os.path.join(meshdir, "dist_1")):
        from oracle.ref.make_dist1 import make_dist1
        make_dist1(meshdir)
    if np_ > 1 and not os.path.isdir(os.path.join(meshdir, f"dist_{np_}")):
        # a partition count the mesh does not ship: written in the reference's format by this package's partition layer into a
        # scratch copy of the mesh directory (the committed fixtures stay as they are)
        import tempfile
        from fesom2_amd import partition_io
        cp = os.path.join(tempfile.gettempdir(), f"fesom_{os.path.basename(meshdir.rstrip('/'))}_dist{np_}")
        if not os.path.isdir(os.path.join(cp, f"dist_{np_}")):
            shutil.copytree(meshdir, cp, dirs_exist_ok=True, copy_function=shutil.copyfile)
            for root, dirs, _ in os.walk(cp):                       # (the source tree may be read-only)
                os.chmod(root, 0o755)
            partition_io.write_dist(cp, np_)
        meshdir = cp
    open(os.path.join(rd, "namelist.config"), "w").write(CONFIG_TMPL.format(meshpath=meshdir, **dict(dict(use_sw_pene=".false.", use_floatice=".false.", min_hnode="0.5", which_toy="soufflet", use_cavity=".false.", use_cavity_partial_cell=".false."), **c)))
    open(os.path.join(rd, "namelist.oce"), "w").write(OCE_TMPL.format(**dict(dict(w_split=".false.", w_max_cfl="1.0", visc_option=5, tra_adv_ver="QR4C", tra_adv_hor="MFCT", Kv0_const=".true.", tra_adv_lim="FCT", use_momix=".false.", which_pgf="shchepetkin", mom_adv=2, use_kpp_nonlclflx=".false.", double_diffusion=".false.", smooth_bh_tra=".false.", clim_relax="0.0", SPP=".false.", use_density_ref=".false.", scaling_Rossby=".false."), **c)))
    if c["toy_ocean"] == ".false." or c.get("which_toy", "soufflet") != "soufflet":
        from fesom2_amd.synthetic import write_ic_files
        write_ic_files(meshdir, rd)
    return rd


def run(cfg, np_, nsteps, mode="step", dump=(), mean=False, dump_mesh=True, quiet=True, exe_name="fesom_oracle.x", step_info=False, gpu_profile=False, ice_adv=False, ice_aevp=False, ice_evp0=False):
    forcing = CFGS[cfg].get("synth_forcing", False)
    rd = prepare(cfg, np_, "" if exe_name == "fesom_oracle.x" else "_" + exe_name.split(".")[0])
    ds = ",".join(str(d) for d in dump) if dump else "-1"
    open(os.path.join(rd, "namelist.oracle"), "w").write(
        f"&oracle\nnsteps={nsteps}\nmode='{mode}'\ndump_dir='dumps'\ndump_steps={ds}\n"
        f"dump_mesh={'.true.' if dump_mesh else '.false.'}\ndo_mean={'.true.' if mean else '.false.'}\n"
        f"synth_forcing={'.true.' if forcing else '.false.'}\nstep_info={'.true.' if step_info else '.false.'}\ngpu_profile={'.true.' if gpu_profile else '.false.'}\nice_adv={'.true.' if ice_adv else '.false.'}\nice_aevp={'.true.' if ice_aevp else '.false.'}\nice_evp0={'.true.' if ice_evp0 else '.false.'}\nmslp={'.true.' if CFGS[cfg].get('mslp') else '.false.'}\ntides={'.true.' if CFGS[cfg].get('tides') else '.false.'}\n/\n")
    exe = os.path.join(OUT, exe_name)
    cmd = ["/opt/conda/bin/mpiexec", "-n", str(np_), exe]
    def big_stack():        # the reference keeps (nl, nodes) work arrays on the stack: larger meshes overflow the default 8 MB
        import resource
        try:
            resource.setrlimit(resource.RLIMIT_STACK, (resource.RLIM_INFINITY, resource.RLIM_INFINITY))
        except (ValueError, OSError):
            pass
    r = subprocess.run(cmd, cwd=rd, capture_output=True, text=True, preexec_fn=big_stack)
    open(os.path.join(rd, "stdout.log"), "w").write(r.stdout + "\n--- stderr ---\n" + r.stderr)
    lines = [l for l in r.stdout.splitlines() if l.startswith("ORACLE") or "ERROR" in l]
    if not quiet or r.returncode != 0:
        print(r.stdout[-3000:]); print(r.stderr[-2000:])
    return rd, r.returncode, lines


if __name__ == "__main__":
    cfg, np_, nsteps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    mode, dump, mean = "step", (), False
    for a in sys.argv[4:]:
        if a.startswith("mode="): mode = a[5:]
        elif a.startswith("dump="): dump = tuple(int(x) for x in a[5:].split(","))
        elif a == "mean": mean = True
    rd, rc, lines = run(cfg, np_, nsteps, mode, dump, mean)
    print(rd, "rc=", rc)
    print("\n".join(lines))
