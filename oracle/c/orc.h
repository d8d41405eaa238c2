/* ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, scalar, reference loop order) of the FESOM2 hot path behind
 * oce_timestep_ale (src/oce_ale.F90:2521-2799).  It exists to CHECK the HIP product path
 * and to serve as bench.py's `cpu_baseline` leg; nothing in fesom2_amd/ may include, link
 * or call it.  Pinned against the real reference (oracle/_ref, built from the sources in
 * /root/reference by oracle/ref/build_ref.sh): see tests/test_oracle_vs_reference.py and
 * the digests in tests/golden/.  Single partition (global numbering): owned values of the
 * reference do not depend on the partition because owned lists are in global order.
 */
#ifndef ORC_H
#define ORC_H
#include "../../include/fesom_gpu.h"
#include <stddef.h>

typedef struct {
  fesom_mesh_desc m;
  fesom_params p;
  int N, E, D, nl, nlm1, ntr;
  int first_step_done;      /* `lfirst` of compute_vel_rhs (oce_ale_vel_rhs.F90:32,124-127) */
  /* node (nl-1) */
  double *tr_arr, *tr_arr_old, *density_m_rho0, *density_ref, *hnode, *hnode_new, *Z_3d_n, *sw_alpha, *sw_beta;
  double *del_ttf, *del_ttf_advhoriz, *del_ttf_advvert, *fct_LO, *fct_ttf_max, *fct_ttf_min, *fct_plus, *fct_minus;
  double *Ki, *Tclim, *Sclim;
  double *relax2clim;                            /* clim_relax > 0: nodal relaxation rate (N) */
  /* node (nl) */
  double *bvfreq, *hpressure, *zbar_3d_n, *Wvel, *Wvel_e, *Wvel_i, *CFL_z, *Kv, *tr_z, *adv_flux_ver, *dbsfc;
  /* node vectors */
  double *Unode, *Unode_rhs, *sigma_xy, *neutral_slope, *slope_tapered, *U_c;   /* (2|3,nl-1,N) */
  /* node 2D */
  double *eta_n, *d_eta, *ssh_rhs, *ssh_rhs_old, *hbar, *hbar_old, *MLD1, *MLD2;
  double *heat_flux, *water_flux, *virtual_salt, *relax_salt, *real_salt_flux;
  double *thdgr, *S_oc_array;                    /* SPP: ice growth rate, ocean salinity seen by the ice (N) */
  double *m_ice, *m_snow, *press_air, *ssh_gp;   /* use_floatice / l_mslp / use_global_tides: surface potentials of compute_vel_rhs (N) */
  double *u_ice, *v_ice, *a_ice, *mixlength;     /* use_momix: ice state (input) and the Monin-Obukhov mixing length (N), kept from step to step */
  /* elem */
  double *UV, *UV_rhs, *UV_rhsAB, *tr_xy, *U_b, *fct_ebnd;  /* (2,nl-1,E) */
  double *pgf_x, *pgf_y, *helem;                 /* (nl-1,E) */
  double *Av;                                    /* (nl,E) */
  double *dhe, *stress_surf;                     /* (E), (2,E) */
  double *KE_node;                               /* mom_adv = 3: kinetic energy at nodes (nl-1,N) */
  double *Visc, *vorticity, *leith_aux;          /* Leith viscosity (nl-1,E), relative vorticity (nl-1,N), smoothing work array (nl-1,N) */
  double *uke, *v_back, *uke_rhs, *uke_rhs_old, *uke_dif, *uke_dis, *uke_back;   /* visc_option = 8: sub-grid energy budget (nl-1,E) */
  double *UV_dis_tend, *UV_back_tend;            /* (2,nl-1,E) */
  /* edge */
  double *adv_flux_hor;                          /* (nl-1,D) */
  double *edge_up_dn_grad;                       /* (4,nl-1,D) */
  /* Gent-McWilliams bolus velocities (orc_gm.c) */
  double *fer_K, *fer_gamma, *fer_Wvel;      /* (nl,N), (2,nl,N), (nl,N) */
  double *fer_c, *fer_scal, *gm_scal_static;  /* (N) */
  double *fer_UV;                            /* (2,nl-1,E) */
  int *MLD1_ind;                             /* (N) */
  /* KPP (orc_kpp.c) */
  double *stress_atmoce_x, *stress_atmoce_y;                                  /* (N) wind stress at nodes */
  double *kpp_sw_node;                                                        /* (N) see orc_kpp.c, second pass of bldepth */
  double *sw_3d;                                                              /* (nl,N) penetrating short-wave flux / vcpw */
  double *kpp_Kv1, *kpp_Kv2, *kpp_viscA, *kpp_dVsq, *kpp_blmc[3];            /* (nl,N) */
  double *kpp_ghats;                                                          /* (nl-1,N) */
  double *kpp_hbl, *kpp_bfsfc, *kpp_caseA, *kpp_stable, *kpp_ustar, *kpp_Bo, *kpp_dkm1;   /* (N), dkm1 (3,N) */
  double *kpp_wmt, *kpp_wst, *kpp_work, *kpp_vol, kpp_deltaz, kpp_deltau, kpp_Vtc, kpp_cg;
  int *kpp_kbl;
  /* Soufflet toy channel (orc_toy.c) */
  double *Uclim, *toy_zvel, *toy_ztem, *toy_znum;  /* (nl-1,E), (nl-1,100) x3 */
  int *toy_bpos, *toy_owner, toy_nranks;
  /* ssh operator + solver */
  double *ssh_values;
  double *sv_h1, *sv_h2, *sv_h3; int sv_nhist;      /* previous SSH solutions for the extrapolated initial guess */
  int solver_iters; double solver_resid;
} orc_ctx;

extern orc_ctx C_;

#define NL   (C_.nl)
#define NLM1 (C_.nlm1)
/* Fortran-style 1-based accessors */
#define A2(a, nz, n)        (a)[(size_t)((n) - 1) * NLM1 + ((nz) - 1)]          /* (nl-1, X) */
#define A2L(a, nz, n)       (a)[(size_t)((n) - 1) * NL + ((nz) - 1)]            /* (nl,   X) */
#define V2(a, c, nz, e)     (a)[((size_t)((e) - 1) * NLM1 + ((nz) - 1)) * 2 + ((c) - 1)]
#define V3(a, c, nz, e)     (a)[((size_t)((e) - 1) * NLM1 + ((nz) - 1)) * 3 + ((c) - 1)]
#define V4(a, c, nz, e)     (a)[((size_t)((e) - 1) * NLM1 + ((nz) - 1)) * 4 + ((c) - 1)]
#define TR(nz, n, t)        C_.tr_arr[((size_t)((t) - 1) * C_.N + ((n) - 1)) * NLM1 + ((nz) - 1)]
#define TRO(nz, n, t)       C_.tr_arr_old[((size_t)((t) - 1) * C_.N + ((n) - 1)) * NLM1 + ((nz) - 1)]
#define EN(j, e)            C_.m.elem2D_nodes[3 * ((e) - 1) + (j) - 1]
#define EDG(j, d)           C_.m.edges[2 * ((d) - 1) + (j) - 1]
#define ETRI(j, d)          C_.m.edge_tri[2 * ((d) - 1) + (j) - 1]
#define NIE(k, n)           C_.m.nod_in_elem2D[(size_t)C_.m.max_nod_in_elem * ((n) - 1) + (k) - 1]
#define GS(j, e)            C_.m.gradient_sca[6 * ((e) - 1) + (j) - 1]
#define ECD(j, d)           C_.m.edge_cross_dxdy[4 * ((d) - 1) + (j) - 1]
#define EDXY(j, d)          C_.m.edge_dxdy[2 * ((d) - 1) + (j) - 1]
#define AREA(nz, n)         C_.m.area[(size_t)((n) - 1) * NL + ((nz) - 1)]
#define AREASVOL(nz, n)     C_.m.areasvol[(size_t)((n) - 1) * NL + ((nz) - 1)]
#define AREASVOL_INV(nz, n) C_.m.areasvol_inv[(size_t)((n) - 1) * NL + ((nz) - 1)]
#define NLEV(e)   C_.m.nlevels[(e) - 1]
#define ULEV(e)   C_.m.ulevels[(e) - 1]
#define NLEVN(n)  C_.m.nlevels_nod2D[(n) - 1]
#define ULEVN(n)  C_.m.ulevels_nod2D[(n) - 1]

#define G_ACC     9.81
#define DENSITY_0 1030.0
#define R_EARTH   6367500.0
#define VCPW      4.2e6

static inline double dmin(double a, double b) { return a < b ? a : b; }
static inline double dmax(double a, double b) { return a > b ? a : b; }

/* routines (one per reference subroutine) */
void orc_compute_vel_nodes(void);
void orc_pressure_bv(void);
void orc_init_ref_density(void);
void orc_pressure_force(void);
void orc_sw_alpha_beta(void);
void orc_compute_sigma_xy(void);
void orc_compute_neutral_slope(void);
void orc_mixing_pp(void);
void orc_mo_convect(void);
void orc_mixing_kpp(void);
void orc_kpp_tables(double *wmt, double *wst, double *deltaz, double *deltau);
void orc_compute_vel_rhs(void);
void orc_visc_filt_bcksct(void);
void orc_viscosity_filter(void);
double orc_kv0_background_qiang(int n, int nz);
void orc_impl_vert_visc_ale(void);
void orc_update_stiff_mat_ale(void);
void orc_compute_ssh_rhs_ale(void);
void orc_solve_ssh(void);
void orc_update_vel(void);
void orc_compute_hbar_ale(void);
void orc_eta_update(void);
void orc_vert_vel_ale(void);
void orc_init_tracers_AB(int tr);
void orc_adv_tracers_ale(int tr);
void orc_diff_tracers_ale(int tr);
void orc_salinity_clamp(void);
void orc_update_thickness_ale(void);
void orc_init_Redi_GM(void);
void orc_fer_solve_Gamma(void);
void orc_fer_gamma2vel(void);
void orc_fer_wvel(void);
void orc_bolus_add(void);
void orc_bolus_remove(void);
void orc_compute_zonal_mean_ini(void);
void orc_compute_zonal_mean(void);
void orc_relax_zonal_vel(void);
void orc_relax_zonal_temp(void);
void orc_relax_to_clim(int tr);
void orc_spp(void);
extern int orc_ale_flag;                        /* zlevel: a step needed the local-zstar fallback (not restated) */
int orc_get_ale_flag(void);
void orc_toy_set_partition(const int *owner, int nranks);
void orc_step(int n);
#endif
