/* fesom_gpu.h -- C ABI of the MI355X-native FESOM2 ocean dynamical core.
 *
 * Drop-in boundary for the hot path behind `oce_timestep_ale(n, mesh)`
 * (reference: src/oce_ale.F90:2521-2799, called at src/fvom_main.F90:250).
 * The reference has no plugin API: its state lives in Fortran module globals
 * (o_ARRAYS src/oce_modules.F90:221-353, t_mesh src/MOD_MESH.F90:19-95,
 * g_PARSUP src/gen_modules_partitioning.F90:2-76).  A Fortran caller hands those
 * arrays over by `c_loc` in the descriptor structs below.
 *
 * Conventions (identical to the Fortran side):
 *   - all reals are IEEE binary64, all integers 32 bit,
 *   - arrays are column-major with the vertical index fastest: (nz, horiz) or (comp, nz, horiz),
 *   - connectivity holds 1-based LOCAL indices (0 / <=0 = "none", e.g. edge_tri(2,ed) on boundary edges),
 *   - node arrays span myDim_nod2D+eDim_nod2D, element arrays myDim_elem2D+eDim_elem2D
 *     (+eXDim_elem2D where noted), edge arrays myDim_edge2D+eDim_edge2D.
 * Host arrays stay owned by the caller; the library owns device mirrors.
 * All entry points return 0 on success, non-zero on error (caller sets pe_status,
 * src/gen_comm.F90:644-657).  One process <-> one GPU; not re-entrant.
 */
#ifndef FESOM_GPU_H
#define FESOM_GPU_H
#ifdef __cplusplus
extern "C" {
#endif

/* ---- mesh + partition description: t_mesh (MOD_MESH.F90:19-95) and myDim/eDim
 *      (gen_modules_partitioning.F90:30-60) ------------------------------------------- */
typedef struct fesom_mesh_desc {
  int nod2D, elem2D, edge2D, edge2D_in, nl;
  int myDim_nod2D, eDim_nod2D;
  int myDim_elem2D, eDim_elem2D, eXDim_elem2D;
  int myDim_edge2D, eDim_edge2D;
  int max_nod_in_elem;                 /* leading extent of nod_in_elem2D */
  int ssh_nza;                         /* local number of CSR entries */
  const int    *myList_nod2D, *myList_elem2D, *myList_edge2D;   /* global ids */
  const double *coord_nod2D;           /* (2,N)  rotated lon/lat [rad] */
  const double *geo_coord_nod2D;       /* (2,N)  geographic lon/lat [rad] */
  const int    *elem2D_nodes;          /* (3,E+eX) */
  const int    *edges;                 /* (2,D) */
  const int    *edge_tri;              /* (2,D) second <=0 on boundary edges */
  const int    *elem_edges;            /* (3,myE) */
  const int    *elem_neighbors;        /* (3,myE) */
  const int    *nod_in_elem2D;         /* (max_nod_in_elem,N) */
  const int    *nod_in_elem2D_num;     /* (N) */
  const int    *nlevels, *ulevels;     /* (E+eX) */
  const int    *nlevels_nod2D, *ulevels_nod2D, *nlevels_nod2D_min, *ulevels_nod2D_max; /* (N) */
  const double *zbar;                  /* (nl)   level interfaces, negative down */
  const double *Z;                     /* (nl-1) mid-levels */
  const double *depth;                 /* (N) */
  const double *elem_area;             /* (E+eX) */
  const double *area, *area_inv, *areasvol, *areasvol_inv;      /* (nl,N) */
  const double *mesh_resolution;       /* (N) */
  const double *gradient_sca;          /* (6,myE) */
  const double *gradient_vec;          /* (6,myE) */
  const double *edge_dxdy;             /* (2,D) */
  const double *edge_cross_dxdy;       /* (4,D) */
  const double *elem_cos, *metric_factor; /* (E+eX) */
  const double *coriolis;              /* (myE)  o_ARRAYS */
  const double *coriolis_node;         /* (N) */
  const int    *ssh_rowptr;            /* (myN+1) 1-based, global offset as in oce_ale.F90:1297 */
  const int    *ssh_colind;            /* (nza)  1-based PE-contiguous global columns (oce_ale.F90:1317-1344) */
  const int    *ssh_colind_loc;        /* (nza)  1-based local columns */
  const double *ssh_values;            /* (nza)  initial operator (init_stiff_mat_ale) */
  const int    *edge_up_dn_tri;        /* (2,myD) o_MESH, oce_muscl_adv.F90:124-281 */
  /* static ALE geometry (oce_ale.F90:82-420) */
  const double *zbar_n_bot, *zbar_n_srf, *bottom_node_thickness;  /* (N) */
  const double *zbar_e_bot, *zbar_e_srf;                          /* (myE+eE) */
  const double *bottom_elem_thickness;                            /* (myE) */
} fesom_mesh_desc;

/* ---- halo exchange lists: com_struct (gen_modules_partitioning.F90:17-29), files
 *      dist_N/com_infoNNNNN.out.  Lists hold 1-based local indices, ptr arrays are 1-based CSR. */
typedef struct fesom_com_desc {
  int rPEnum, sPEnum;
  const int *rPE, *rptr, *rlist;
  const int *sPE, *sptr, *slist;
} fesom_com_desc;

typedef struct fesom_part_desc {
  int npes, mype;
  fesom_com_desc com_nod2D, com_elem2D, com_elem2D_full;
} fesom_part_desc;

/* ---- namelist scalars the path reads (namelist.config / namelist.oce; defaults
 *      src/oce_modules.F90:7-190, src/gen_modules_config.F90:8-127) ------------------- */
typedef struct fesom_params {
  double dt;                 /* 86400/step_per_day (gen_model_setup.F90:44) */
  int    which_ale;          /* 0 linfs, 1 zlevel, 2 zstar */
  int    use_partial_cell;
  int    state_equation;     /* 1 Jackett-McDougall, 0 linear */
  int    num_tracers;
  int    mom_adv;            /* 2 (scalar control volumes) */
  int    visc_option;        /* 5 (easy backscatter, visc_filt_bcksct; default), 1 (Leith + visc_filt_harmon), 2 (Leith + visc_filt_hbhmix), 3 (Leith + visc_filt_biharm(2)),
                                4 (visc_filt_biharm(1)), 6 (visc_filt_bilapl), 7 (visc_filt_bidiff), 8 (backscatter_coef + visc_filt_dbcksc + uke_update); oce_dyn.F90:196-228 */
  int    i_vert_visc, i_vert_diff, w_split;
  int    mix_scheme;         /* 1 = KPP (oce_ale_mixing_kpp.F90) ; 2 = PP ; 0 = constant A_ver/K_ver (no mixing scheme) */
  int    use_instabmix, use_windmix, windmix_nl;
  int    toy_soufflet;       /* 1: linear EOS branch of the Soufflet channel (oce_ale_pressure_bv.F90:2992) */
  double alpha, theta, epsilon;
  double C_d, A_ver, K_ver, K_hor;
  double gamma0, gamma1, gamma2, easy_bs_return;
  double w_max_cfl;
  double tra_adv_ph, tra_adv_pv;
  double instabmix_kv, windmix_kv;
  double cyclic_length;      /* [rad] */
  int    with_diffusion;     /* 1: run diff_tracers_ale closure (rows f-1 i); 0: advection only */
  int    solver_x0_order;    /* SSH solver initial guess: 0 = previous d_eta (reference, psolve.c:206-212);
                                2 / 3 = quadratic / cubic extrapolation of the last three / four solutions (fewer
                                iterations, same tolerance) */
  /* Gent-McWilliams bolus velocities after Ferrari et al. 2010 (src/oce_fer_gm.F90; namelist.oce &oce_dyn) */
  int    Fer_GM;
  double K_GM_max, K_GM_min;
  int    K_GM_bvref;         /* reference N^2 of the Ferreira scaling: 0 surface, 1 below the mixed layer, 2 mean over it */
  double K_GM_rampmax, K_GM_rampmin, K_GM_resscalorder;
  int    scaling_Ferreira, scaling_Rossby /* unsupported */, scaling_resolution, scaling_FESOM14;
  int    Redi;               /* isoneutral (Redi) diffusion: rotated horizontal + explicit/implicit vertical parts (oce_ale_tracer.F90) */
  /* KPP (mix_scheme=1; namelist.oce: visc_sh_limit, diff_sh_limit, Ricr, concv; double_diffusion=.false.,
     use_sw_pene=.false., use_kpp_nonlclflx=.false. are the only supported settings of those switches) */
  double visc_sh_limit, diff_sh_limit, Ricr, concv;
  int    use_sw_pene;        /* short-wave penetration (namelist.config run_config): sw_3d of the forcing enters the temperature
                                equation (oce_ale_tracer.F90:785-791) and the KPP surface buoyancy forcing (oce_ale_mixing_kpp.F90:508-640) */
  int    tra_adv_ver;        /* high-order vertical tracer advection under FCT (namelist.oce tra_adv_ver, oce_adv_tra_driver.F90:162-177):
                                0 'QR4C' (default), 1 'CDIFF', 2 'UPW1', 3 'PPM'; tra_adv_lim='FCT' is fixed */
  int    tra_adv_hor;        /* high-order horizontal tracer advection under FCT (tra_adv_hor, oce_adv_tra_driver.F90:140-153):
                                0 'MFCT' (default), 1 'MUSCL' (nboundary_lay of oce_muscl_adv.F90:74-104 is formed inside the library), 2 'UPW1' */
  int    Kv0_const;          /* 1 (default): background vertical diffusivity K_ver; 0: latitude/depth dependent Kv0_background_qiang
                                (oce_ale_mixing_pp.F90:91-125) in the PP and KPP schemes */
  int    solver_precond;     /* SSH solver preconditioner, frozen at the operator of the first step like the reference's ILU factors
                                (psolve.c:117-150): 0 = Jacobi; 1 = explicit inverse of the row-scaled operator (fp32, applied as one
                                full-GPU matrix-vector product) where it fits: single partition, <= 4096 rows (pi), else Jacobi */
  int    tra_adv_lim;        /* limiter of the tracer advection (namelist.oce tra_adv_lim, oce_adv_tra_driver.F90:79-197): 0 'FCT' (default: low-order
                                solution + limited anti-diffusive fluxes), 1 'NON' (the high-order fluxes applied as they are, vertical part with the
                                explicit velocity; not together with w_split) */
  int    solver_xinv_its;    /* solver_precond=1: BiCGstab iterations enqueued per solve (no host read-back inside a step); a solve that
                                has not converged by then is finished by the Jacobi-preconditioned one-workgroup solver.  0 = default (1: after the spin-up the
                                extrapolated first guess makes one iteration enough; where a second is needed the continuation costs one 6.6 us
                                Jacobi iteration instead of five launches) */
  double Leith_c, Div_c;     /* visc_option 1-3: weights of the Leith and the modified (divergence) Leith viscosity (namelist.oce &oce_dyn; h_viscosity_leith,
                                src/oce_dyn.F90:461-561) */
  int    which_pgf;          /* namelist.oce which_pgf: 0 'shchepetkin' (default, oce_modules.F90:172): pressure_force_4_zxxxx_shchepetkin for zstar,
                                pressure_force_4_linfs_shchepetkin for linfs with partial cells; linfs with full cells always takes
                                pressure_force_4_linfs_fullcell (oce_ale_pressure_bv.F90:385-386); 1 'cubicspline': pressure_force_4_zxxxx_cubicspline (zstar),
                                pressure_force_4_linfs_cubicspline (linfs with partial cells); 2 'nemo': pressure_force_4_linfs_nemo (linfs with partial cells only);
                                3 'easypgf': pressure_force_4_zxxxx_easypgf (zstar), pressure_force_4_linfs_easypgf (linfs with partial cells).  Anything else (sergey): pass -1, fesom_gpu_init refuses it */
  int    use_momix;          /* Monin-Obukhov mixing of Timmermann & Beckmann 2004 inside mo_convect (oce_mo_conv.F90:22-55, :95; on in the shipped
                                config/namelist.oce:48; the reference allocates its arrays only with use_ice): needs u_ice, v_ice, a_ice with the forcing */
  double momix_lat, momix_kv;/* applied south of momix_lat [degrees] (-50), diffusivity / viscosity added within the mixing length (0.01) */
  int    use_kpp_nonlclflx;  /* KPP (mix_scheme = 1): non-local transport of heat and salt in the implicit vertical diffusion (oce_ale_tracer.F90:688-781) */
  int    ref_sss_local;      /* its reference salinity: the local surface salinity tr_arr(1,n,2) (namelist.oce: .true.) or ref_sss */
  double ref_sss;
  int    smooth_bh_tra;      /* biharmonic diffusion of the tracers applied as a filter at the end of diff_tracers_ale (diff_part_bh, oce_ale_tracer.F90:1081-1150;
                                it uses the momentum coefficients gamma0 / gamma1 / gamma2) */
  int    double_diffusion;   /* KPP: salt fingering / diffusive convection added to the interior diffusivities (ddmix, oce_ale_mixing_kpp.F90:857-934) */
  /* potentials beside g*eta_n in the surface pressure gradient of compute_vel_rhs (oce_ale_vel_rhs.F90:52-76); the nodal arrays come with the forcing */
  int    use_floatice;       /* ice + snow load g*min((m_ice*rhoice + m_snow*rhosno)/rhowat, max_ice_loading); the reference applies it unless which_ALE = 'linfs' */
  int    l_mslp;             /* atmospheric pressure press_air / 1000 */
  int    use_global_tides;   /* tidal potential ssh_gp (gen_modules_gpot.F90) */
  double max_ice_loading;    /* namelist.config &ale_def (5.0) */
  int    SPP;                /* salt plume parameterization at the head of solve_tracers_ale (cal_rejected_salt / app_rejected_salt, src/oce_spp.F90): the salt rejected by
                                growing ice (thdgr > 0, from S_oc_array and Sice) is taken from the surface layer and spread over the mixed layer (northern hemisphere) */
  double Sice;               /* ice salinity (i_therm_param, 4.0) */
  double clim_relax;         /* > 1e-8: relax_to_clim after diff_tracers_ale (oce_tracer_mod.F90:86-121): T, S += relax2clim(n) * dt * (clim - tracer); the static
                                arrays Tclim, Sclim (nl-1,N) and relax2clim (N) are handed over once with fesom_gpu_set_field */
  int    lzstar_lev;         /* which_ALE='zlevel' (namelist.config &ale_def, 4): number of surface layers the reference's local-zstar fallback works on */
  double min_hnode;          /* which_ALE='zlevel' (&ale_def, 0.5): smallest allowed fraction of the surface layer's resting thickness; a step that would go below it
                                needs the local-zstar fallback of vert_vel_ale (oce_ale.F90:1859-1942), which is not built: the library reports an error */
  /* visc_option = 8: kinematic backscatter with a sub-grid energy budget (Juricke et al.; backscatter_coef + visc_filt_dbcksc + uke_update,
     src/oce_dyn.F90:806-1152; namelist.oce &oce_dyn, defaults src/oce_modules.F90:34-41).  The unresolved kinetic energy `uke` is prognostic
     device state (fesom_gpu_get/set_field "uke", "uke_rhs", "uke_rhs_old").  Single partition; which_toy = 'soufflet' as in the shipped
     namelist.config (the hard-coded regional mask of uke_update, :1107-1121, is not built). */
  double c_back, K_back, uke_scaling_factor, rosb_dis, scale_area;
  int    uke_scaling, smooth_back, smooth_dis, smooth_back_tend;
  /* ice-shelf cavities (namelist.config &run_config use_cavity, use_cavity_partial_cell): the mesh carries upper levels ulevels > 1
     (src/oce_mesh.F90:897-1280); ocean_setup then switches the reference density profile on (src/oce_setup_step.F90:120-129: use_density_ref,
     init_ref_density src/oce_ale_pressure_bv.F90:3024-3070 from density_ref_T / _S, defaults src/oce_modules.F90:144-146): formed on the device from the
     Z_3d_n of the first uploaded state.  use_density_ref alone (without cavities) is accepted as well. */
  int    use_cavity, use_density_ref;
  double density_ref_T, density_ref_S;
  int    use_cavity_partial_cell;   /* partial cells at the shelf base (set-up: fesom_mesh_opts); in the step it only selects the pressure gradient of linfs:
                                       which_pgf 0 / 3 / 4 = 'sergey' (pressure_force_4_linfs_cavity, src/oce_ale_pressure_bv.F90:385-403,1451-1663) */
} fesom_params;

/* ---- prognostic state = restart set (io_restart.F90:99-155) + ALE thickness arrays -- */
typedef struct fesom_state_desc {
  double *tr_arr;        /* (nl-1,N,ntr) */
  double *tr_arr_old;    /* (nl-1,N,ntr) */
  double *UV;            /* (2,nl-1,E) */
  double *UV_rhsAB;      /* (2,nl-1,E) */
  double *eta_n, *d_eta, *ssh_rhs, *ssh_rhs_old, *hbar, *hbar_old;   /* (N) */
  double *dhe;           /* (myE) */
  double *hnode, *hnode_new;   /* (nl-1,N) */
  double *helem;         /* (nl-1,myE) */
  double *zbar_3d_n;     /* (nl,N) */
  double *Z_3d_n;        /* (nl-1,N) */
  double *Wvel, *Wvel_e, *Wvel_i;  /* (nl,N) */
  double *ssh_values;    /* (nza) current SSH operator */
} fesom_state_desc;

/* ---- per-step surface forcing (all optional: NULL = zero) --------------------------- */
typedef struct fesom_forcing_desc {
  const double *stress_surf;     /* (2,myE) */
  const double *heat_flux, *water_flux, *virtual_salt, *relax_salt, *real_salt_flux; /* (N) */
  const double *stress_atmoce_x, *stress_atmoce_y;   /* (N) wind stress at nodes (KPP friction velocity, oce_ale_mixing_kpp.F90:341) */
  const double *sw_3d;           /* (nl,N) penetrating short-wave flux / vcpw [K m/s], positive down (gen_modules_forcing.F90:76); use_sw_pene only */
  const double *m_ice, *m_snow;  /* (N) ice and snow thickness of i_ARRAYS; use_floatice only */
  const double *press_air;       /* (N) g_forcing_arrays; l_mslp only */
  const double *ssh_gp;          /* (N) o_ARRAYS; use_global_tides only */
  const double *thdgr, *S_oc_array;   /* (N) ice growth rate (g_forcing_arrays) and the ocean salinity seen by the ice (i_ARRAYS); SPP only */
  const double *u_ice, *v_ice, *a_ice;   /* (N) ice velocity and concentration of i_ARRAYS: the turbulent-kinetic-energy source of mo_length (oce_mo_conv.F90:36-39); use_momix only */
} fesom_forcing_desc;

/* Lifecycle.  fesom_gpu_init uploads the mesh and allocates every device mirror;
 * fesom_gpu_upload_state / _download_state move the prognostic set;
 * fesom_gpu_step(n) is `compute_vel_nodes` + `oce_timestep_ale(n, mesh)`
 * (fvom_main.F90:216,250) for the configured options. */
int  fesom_gpu_init(const fesom_mesh_desc *mesh, const fesom_part_desc *part, const fesom_params *par);
int  fesom_gpu_upload_state(const fesom_state_desc *st);
int  fesom_gpu_download_state(const fesom_state_desc *st);
int  fesom_gpu_set_forcing(const fesom_forcing_desc *f);
int  fesom_gpu_step(int n);                             /* asynchronous (stream-ordered); download_state / get_field / step_info / fesom_gpu_sync wait */
int  fesom_gpu_run_steps(int n_first, int nsteps);      /* nsteps back-to-back, no host sync in between */
int  fesom_gpu_finalize(void);

/* Introspection used by the parity tests and bench.py (no reference counterpart):
 * copy a named device field to the host; run one named routine of the step. */
int  fesom_gpu_get_field(const char *name, double *out, long long count);
int  fesom_gpu_set_field(const char *name, const double *in, long long count);
int  fesom_gpu_call(const char *routine, int arg);
/* ---- one step of a PARTITIONED run driven by the library, bytes moved by the host's transport ---------------------
 * The library runs the phases of the step in the reference's order (src/oce_ale.F90:2556-2767) with a halo exchange at
 * each of the reference's exchange points and the partitioned SSH solve (halo of the gathered vector before each SpMV,
 * global sums of the partial dot products after it); the HOST supplies how bytes move between ranks:
 *   exchange      : `send_dev` / `recv_dev` are the library's DEVICE buffers, already packed; a neighbour's block holds
 *                   count(p) * values_per_item doubles, blocks consecutive in the sPE / rPE order of fesom_gpu_halo_info(kind).
 *                   Fortran/MPI: fesom_gpu_copy to a host buffer, MPI_Isend/Irecv, copy back (or GPU-aware MPI on the device
 *                   pointers); Python: torch.distributed (RCCL) on the device buffers.
 *   allreduce_sum : global sum over the ranks of n doubles at the DEVICE address `buf_dev`, in place.
 * Both return 0 on success.  The library's kernels are stream-ordered on its stream (fesom_gpu_set_stream); a transport that
 * is not on that stream synchronises through fesom_gpu_copy / fesom_gpu_sync.  fvom_main equivalent: one call per step. */
typedef struct fesom_transport {
  void *ctx;
  int (*exchange)(void *ctx, int kind, void *send_dev, void *recv_dev, int values_per_item);
  int (*allreduce_sum)(void *ctx, void *buf_dev, int n);
} fesom_transport;
int  fesom_gpu_step_partitioned(int n, const fesom_transport *t);   /* t == NULL: the built-in RCCL transport below */
/* Soufflet channel on a partition: compute_zonal_mean (src/toy_channel_soufflet.F90:157-217: rank-local sums, global sums, division)
 * outside a step -- the set-up calls it once before the first step; inside fesom_gpu_step_partitioned it runs every 10th step. */
int  fesom_gpu_toy_zonal_mean(const fesom_transport *t);

/* ---- built-in transport: RCCL send/recv over xGMI issued by the library itself (replaces exchange_nod / exchange_elem,
 * src/gen_halo_exchange.F90:58-1035, and the MPI_Allreduce of pARMS' dot products, lib/parms/src/parms_comm.c:205-356).
 * Every halo exchange is ONE group of ncclSend/ncclRecv (a pair per neighbour of the com list) on the stream that runs the
 * pack / unpack kernels; the solver's global sums are ncclAllReduce on the same stream: no host callback, no host
 * synchronisation per exchange.  Set-up: rank 0 calls fesom_gpu_comm_unique_id (128 bytes), the host broadcasts them
 * (MPI_Bcast on the Fortran side, fesom_gpu_shim.F90), every rank calls fesom_gpu_comm_init(id, npes, mype) with the
 * partition's rank numbering; then fesom_gpu_step_partitioned(n, NULL).  librccl is loaded on first use (the copy already in
 * the process, e.g. PyTorch's; FESOM_GPU_RCCL_LIB=<path> overrides). */
int  fesom_gpu_comm_unique_id(void *id128);
int  fesom_gpu_comm_init(const void *id128, int nranks, int rank);
int  fesom_gpu_comm_finalize(void);
int  fesom_gpu_comm_selftest(int n);            /* ring shift of n doubles + a global sum through the transport; 0 = ok */
int  fesom_gpu_comm_timing(int on);             /* HIP-event timing of every exchange (pack .. unpack) from now on */
int  fesom_gpu_comm_stats(long long *exchanges, long long *allreduces, double *exchange_ms);   /* since the last call */
int  fesom_gpu_comm_counts(long long out[4]);   /* exchange points, message parts (node + element fields in one exchange point = 2), all-reduces, exchanges on the communication stream; since the last comm_stats call */

/* Device-side step monitor = write_step_info + check_blowup of the reference (src/write_step_info.F90:14-222, :225-447),
 * evaluated on the device over this rank's OWNED nodes, no per-step host synchronisation needed: call it at the logging
 * cadence.  The sums are the rank-local parts (sum over owned nodes of areasvol(ulevels,n)*x(n)); the host adds them over
 * the ranks and divides by the summed area (MPI_Allreduce in the reference), min/max likewise.  Field order = NAMES below.
 * Reference quirks kept: pgf_x/pgf_y/Av are scanned over the first myDim_nod2D ELEMENT columns (:154-163), T/S extrema
 * only where S /= 0, hnode extrema only where hnode /= 0. */
typedef struct fesom_step_info {
  double sum_eta, sum_hbar, sum_deta, sum_dhbar, sum_wflux, sum_area;
  double min_eta, min_hbar, min_wflux, min_hflux, min_temp, min_salt, min_wvel, min_wvel2, min_uvel, min_uvel2, min_vvel,
         min_vvel2, min_deta, min_hnode, min_hnode2;
  double max_eta, max_hbar, max_wflux, max_hflux, max_temp, max_salt, max_wvel, max_wvel2, max_uvel, max_uvel2, max_vvel,
         max_vvel2, max_deta, max_hnode, max_hnode2, max_cfl_z, max_pgfx, max_pgfy, max_av, max_kv;
  double blowup;             /* 1.0 if any owned node fails check_blowup's tests (NaN, |eta|>50, T outside -5..60, S outside 0..50, ...) */
} fesom_step_info;
int  fesom_gpu_step_info(fesom_step_info *out);
/* One step executed phase by phase on one stream with HIP events at the points where oce_timestep_ale reads MPI_Wtime
 * (src/oce_ale.F90:2546-2768): ms[0..6] = the device time of rtime_oce_mixpres, _dyn, _dynssh, _solvessh, _GMRedi, _solvetra and
 * rtime_oce (same sums of intervals as :2771-2777), so that the host can keep filling the reference's own phase statistics
 * ("BENCHMARK RUNTIME", src/fvom_main.F90:281-326).  A profiling aid: the phases do not overlap here, the step is slower than
 * fesom_gpu_step.  Single partition, no toy hooks. */
int  fesom_gpu_profile_step(int n, double ms[7]);
int  fesom_gpu_last_solver_iterations(void);
/* kernel shape chosen at init: 0 = one column per wave (pi class), > 0 = tiles for CORE2-class meshes (>= 20 000 node columns;
   FESOM_GPU_TILE overrides).  Informational: results do not depend on it. */
int  fesom_gpu_tile_shape(void);
/* SSH preconditioner in use: 0 = Jacobi, 1 = explicit (block) inverse -- see fesom_params.solver_precond */
int  fesom_gpu_solver_kind(void);
int  fesom_gpu_solver_safety_net_count(void);   /* solver_precond=1: solves the Jacobi safety net had to finish since fesom_gpu_init */
double fesom_gpu_last_solver_residual(void);
int  fesom_gpu_kernel_time_ms(const char *kernel_group, int nrep, double *ms_per_launch);
const char *fesom_gpu_last_error(void);

/* Halo exchange (multi-GPU, one rank per GPU; reference: exchange_nod / exchange_elem, src/gen_halo_exchange.F90:58-1035,
 * lists = com_struct of fesom_part_desc).  The library packs and unpacks on the device, the HOST moves the bytes between
 * ranks (MPI_Isend/Irecv on the device buffers in a Fortran/MPI host; torch.distributed in this repository's Python host).
 * kind: 0 com_nod2D, 1 com_elem2D, 2 com_elem2D_full.  Fields exchanged at the same point of the step share one message
 * per neighbour: block of neighbour p = count(p) * values_per_item doubles, blocks consecutive in sPE (send) / rPE (recv)
 * order.  With npes > 1 the step is driven phase by phase through fesom_gpu_call (kernel names "k_*", partitioned SSH solve
 * "ds_*"); the sequence with its exchange points is fesom2_amd/parallel.py. */
int  fesom_gpu_halo_info(int kind, int *npes, int *mype, int *nr, int *rPE, int *rcnt, int *ns, int *sPE, int *scnt);   /* rPE/rcnt/sPE/scnt: room for npes entries each */
int  fesom_gpu_halo_pack(int kind, int nfields, const char *const *names, void **send_dev, void **recv_dev, int *values_per_item);
int  fesom_gpu_halo_unpack(int kind, int nfields, const char *const *names);
int  fesom_gpu_copy(void *dst, const void *src, long long bytes, int dir);   /* 0: device->host, 1: host->device */
int  fesom_gpu_sync(void);
int  fesom_gpu_field_ptr(const char *name, void **dev, long long *count);   /* device address of a named field */
int  fesom_gpu_set_stream(void *hip_stream);   /* run the library's kernels on the host's stream (stream-ordered transport) */

/* ---- sea-ice mEVP rheology: EVPdynamics_m (src/ice_maEVP.F90:273-602), the subcycled momentum solve of the sea-ice model
 * (whichEVP = 1; Bouillon et al. 2013 / Kimmritz et al. 2015).  One call = evp_rheol_steps subcycles (default 120) of: strain rates
 * and viscous-plastic stresses on elements, stress divergence gathered to nodes, implicit Coriolis / ocean-drag velocity update,
 * coastal boundary condition.  Independent of the ocean core's context (may coexist with it); partitions: fesom_gpu_ice_evp_partitioned.
 * Arrays keep the reference's extents: node fields myDim_nod2D + eDim_nod2D, stresses myDim_elem2D.  Not built: cavities
 * (ulevels > 1), icepack.  The other two rheologies of the reference -- the classic EVP (whichEVP = 0, src/ice_EVP.F90) and the adaptive EVP (whichEVP = 2,
 * EVPdynamics_a :785-888) -- are selected by fesom_ice_params.whichEVP. */
typedef struct fesom_ice_params {
  double ice_dt;             /* ice_ave_steps * dt */
  double ellipse, alpha_evp, beta_evp, Pstar, c_pressure, delta_min, cd_oce_ice;   /* namelist.ice &ice_dyn (src/ice_modules.F90:7-27) */
  double max_ice_loading;    /* namelist.config &ale_def */
  int    evp_rheol_steps;
  int    use_floatice;       /* use_floatice .and. which_ALE /= 'linfs' (ice_maEVP.F90:159): ice + snow load in the sea-surface slope term */
  double ice_gamma_fct;      /* smoothing parameter of the FCT advection (namelist.ice &ice_dyn, src/ice_modules.F90:27: 0.25) */
  int    whichEVP;           /* as namelist.ice &ice_dyn (src/ice_modules.F90:42): 0 = the classic EVP, EVPdynamics (src/ice_EVP.F90:397-667; the reference's default),
                                1 = mEVP, EVPdynamics_m, 2 = adaptive EVP, EVPdynamics_a (src/ice_maEVP.F90:785-888) */
  double c_aevp;             /* aEVP: constant of the adaptive alpha (namelist.ice &ice_dyn, src/ice_modules.F90:36: 0.15) */
  double theta_io;           /* classic EVP: ice-ocean turning angle (src/ice_modules.F90:32: 0) */
  double Tevp_inv;           /* classic EVP: inverse relaxation time, 3 / ice_dt (ice_setup, src/ice_setup_step.F90:33) */
} fesom_ice_params;
typedef struct fesom_ice_state {
  double *u_ice, *v_ice;                                            /* in / out */
  double *a_ice, *m_ice, *m_snow;                                   /* in; in / out of the advection */
  double *elevation, *u_w, *v_w, *stress_atmice_x, *stress_atmice_y;   /* in */
  double *sigma11, *sigma12, *sigma22;                              /* in / out: the stresses are state across calls */
  double *alpha_evp_array, *beta_evp_array;                         /* aEVP, in / out: (myDim_elem2D), (nodes); state across calls, = alpha_evp at the start (src/ice_setup_step.F90:85-89) */
} fesom_ice_state;
int  fesom_gpu_ice_init(const fesom_mesh_desc *mesh, const fesom_part_desc *part, const fesom_ice_params *par);
int  fesom_gpu_ice_upload(const fesom_ice_state *st);     /* every non-NULL field host -> device */
int  fesom_gpu_ice_evp(int ncalls);                       /* ncalls x EVPdynamics_m (whichEVP = 2: EVPdynamics_a) on the device-resident state; asynchronous */
int  fesom_gpu_ice_evp_partitioned(int ncalls, const fesom_transport *t);   /* npes > 1: halo of (u_ice_aux, v_ice_aux) after every subcycle (ice_maEVP.F90:588-596); t == NULL: built-in RCCL transport */
/* FCT advection of m_ice, a_ice, m_snow with the current ice velocities = the "Advection part" of ice_timestep (src/ice_setup_step.F90:213-232):
 * ice_TG_rhs_div, ice_fct_solve (ice_solve_high_order, ice_solve_low_order, ice_fem_fct x 3), ice_update_for_div (src/ice_fct.F90), cut_off
 * (src/ice_thermo_oce.F90:2-63).  One ice step of the dynamics = fesom_gpu_ice_evp(1) + fesom_gpu_ice_advect(1). */
int  fesom_gpu_ice_advect(int ncalls);
int  fesom_gpu_ice_advect_partitioned(int ncalls, const fesom_transport *t);   /* npes > 1: the reference's exchange_nod calls, three tracers per message */
int  fesom_gpu_ice_download(const fesom_ice_state *st);   /* u_ice, v_ice, a_ice, m_ice, m_snow, sigma11/12/22 device -> host (synchronises) */
int  fesom_gpu_ice_time_ms(int ncalls, double *ms_per_call);   /* device time of a call (HIP events), state left as after the calls */
int  fesom_gpu_ice_finalize(void);
const char *fesom_gpu_ice_last_error(void);

/* SSH solver with the reference's own C signatures (src/psolve.c:16,117,152;
 * Fortran interface blocks src/oce_ale.F90:2272-2291).  All by reference,
 * 0-based CSR, part[0..npes] prefix of owned rows.  fcomm is ignored on one GPU. */
void psolver_init(int *id, int *stype, int *pctype, int *pcilutype, int *ilulevel, int *fillin,
                  double *droptol, int *maxits, int *restart, double *soltol,
                  int *part, int *rptr, int *cols, double *vals, int *reuse, int *fcomm);
void psolve(int *id, double *rhs, double *vals, double *sol, int *newvals);
void psolver_final(void);
/* The same three under library-prefixed names (what a host adapter that defines psolver_init / psolve / psolver_final itself forwards to on one rank). */
void fesom_gpu_psolver_init(int *id, int *stype, int *pctype, int *pcilutype, int *ilulevel, int *fillin,
                            double *droptol, int *maxits, int *restart, double *soltol,
                            int *part, int *rptr, int *cols, double *vals, int *reuse, int *fcomm);
void fesom_gpu_psolve(int *id, double *rhs, double *vals, double *sol, int *newvals);
void fesom_gpu_psolver_final(void);
/* Distributed SSH solve = psolver_init / psolve of the reference with npes > 1 (src/psolve.c:16-221: the rows of this rank's block, `part` the
 * prefix of the owned-row counts, `cols` in the global contiguous numbering of src/oce_ale.F90:1298-1344; pARMS BiCGstab + RAS/ILU,
 * lib/parms/src/bicgstab_ras.c:49-259).  The library holds no MPI: the host adapter fesom2_amd/fortran/fesom_gpu_psolve_mpi.c (compiled with
 * the application's mpi.h; it defines psolver_init / psolve / psolver_final with the reference's signatures) works out the halo of the row
 * block with MPI and hands it over: rglob[0 .. sum(rcnt)) = global rows of the halo columns in receive order (grouped by rPE), sloc[0 ..
 * sum(scnt)) = owned rows (0-based, local) to send, grouped by sPE.  t = the transport callbacks (the exchange moves kind-0 messages of the
 * halo just described; fesom_gpu_halo_info(0, ...) returns it), or NULL = the built-in RCCL transport after fesom_gpu_comm_init.  The
 * context then holds the solver alone (fesom_gpu_init replaces it).  Solve: BiCGstab over the owned rows, right-preconditioned with the
 * frozen RAS-Chebyshev operator of the rank's block, stop at ||scaled residual|| < soltol (1e-10 if <= 0) as the reference; a solve that does
 * not converge within maxits is an error.  Both return 0 on success, the message is in fesom_gpu_last_error(). */
int  fesom_gpu_psolver_init_dist(int npes, int mype, const int *part, const int *rptr, const int *cols, const double *vals, int maxits, double soltol,
                                 int nr, const int *rPE, const int *rcnt, const int *rglob, int ns, const int *sPE, const int *scnt, const int *sloc,
                                 const fesom_transport *t);
int  fesom_gpu_psolve_dist(const double *rhs, const double *vals, double *sol, int newvals);
int  fesom_gpu_psolver_iterations(void);          /* BiCGstab iterations of the last distributed solve (-1: none) */

/* ---- host mesh layer (setup, untimed): restates mesh_setup + ocean_setup geometry
 *      (src/oce_mesh.F90:108-143, src/oce_ale.F90:82-795,1088-1354, src/oce_muscl_adv.F90:124-281)
 *      for callers that have no Fortran host (tests, bench.py). --------------------------- */
typedef struct fesom_mesh_opts {
  int    force_rotation;     /* namelist geometry */
  double cyclic_length_deg;
  double alphaEuler_deg, betaEuler_deg, gammaEuler_deg;
  int    use_partial_cell;
  int    which_ale;          /* as fesom_params */
  double dt, alpha, theta, K_hor;
  int    npes, mype;         /* partition dist_<npes>/ ; npes==1 -> trivial partition, no files needed */
  int    use_cavity;         /* ice-shelf cavities: cavity_elvls.out, cavity_nlvls.out, cavity_depth.out of the mesh directory (src/oce_mesh.F90:897-1280) */
  int    use_cavity_partial_cell;       /* partial cells at the ice-shelf base: zbar_e_srf / zbar_n_srf from the draft (init_surface_elem_depth / init_surface_node_depth, src/oce_ale.F90:422-545) */
  double cavity_partial_cell_thresh;    /* only where the full surface cell is thicker than this (namelist.config &run_config, default 0) */
} fesom_mesh_opts;

void *fesom_mesh_load(const char *meshdir, const fesom_mesh_opts *opts);     /* NULL on error */
const fesom_mesh_desc *fesom_mesh_get_desc(void *h);
const fesom_part_desc *fesom_mesh_get_part(void *h);
const fesom_state_desc *fesom_mesh_get_initial_state(void *h, int num_tracers); /* zeros + ALE thickness init */
void  fesom_mesh_free(void *h);

#ifdef __cplusplus
}
#endif
#endif
