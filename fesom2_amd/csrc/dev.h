// Device-side data model of the MI355X ocean core (gfx950 only).
//
// Layout in HBM = the reference's layout: column-major with the vertical
// index fastest, so one wet column (<= nl-1 values of 8 B, 376 B on the pi mesh) is a contiguous
// burst.  Kernels map ONE 64-lane wavefront to ONE column (lane = level), 4 columns per 256-thread
// workgroup: gathers of neighbouring columns (3 nodes of an element, 2 elements of an edge, the
// element/edge cluster of a node) are then whole-line reads.  Edge->node scatter-adds of the
// reference are turned into node-centred gathers over a CSR of incident edges kept in increasing
// edge order, which reproduces the reference's floating-point summation order bit for bit.
// No FMA contraction (-ffp-contract=off): results are compared bitwise with the CPU oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include "../../include/fesom_gpu.h"

#define WAVE 64
#define COLS_PER_BLOCK 4
#define BLOCK (WAVE * COLS_PER_BLOCK)

struct DM {   // device mesh + fields, passed BY VALUE to every kernel (kernarg segment, scalar loads)
  int N, E, EX, D, myN, myE, myD, nl, nlm1, ntr, maxk, nza, edge2D_in, ssh_maxnnz;   // EX = E + extended element halo
  int use_tile;               // CORE2-class shape of the kernels that end in a Thomas sweep (ThTile<.., TL_COLS>), see TL_MIN_COLUMNS
  // ---- connectivity, 0-based (-1 = none)
  const int *elem_nodes;      // (3,E)
  const int *edges;           // (2,D)
  const int *edge_tri;        // (2,D)
  const int *nie, *nie_num;   // (maxk,N), (N)
  const int *cl_nb, *cl_nbn, *cl_pos; int cl_maxu;   // distinct nodes of a node's element cluster (cl_maxu, myN), their number (myN), per cluster element the positions of its 3 nodes in that list, packed (maxk, myN): k_kpp_smooth_u
  const int *nlev, *ulev;     // (E)   1-based level counts as in the reference
  const int *nlev_n, *ulev_n, *nlev_n_min, *ulev_n_max;   // (N)
  const int *edge_glob;       // (D) global edge id (1-based) for the internal/boundary test
  const int *ne_ptr, *ne_idx, *ne_sgn;     // node -> incident owned edges (increasing), sign +1 if node==edges(1)
  const unsigned *ne_rng;                  // per entry: level range of the edge, lo | hi<<8 (min ulevels / max nlevels-1 of its triangles)
  const int *ee_idx, *ee_side;             // (3,E) element -> its edges sorted increasing; side 1/2 (=which edge_tri slot), 0 = skip
  const int *updn;            // (2,myD) up/down-wind triangles, 0-based, -1 none
  const int *rowptr, *colind; // SSH CSR, 0-based local
  const int *su_ptr, *su_elem; const double *su_coef;   // stiffness update lists per CSR entry
  // ---- geometry
  const double *elem_area, *area, *areasvol, *areasvol_inv, *gsca, *ecd, *edxy, *elem_cos, *coriolis;
  const double *zbar_e_bot, *zbar_n_bot, *zbar, *Z;
  const double *zbar_e_srf;                               // (E) upper face of the element column: 0, or the shelf base (pressure boundary term of the linfs cubic-spline gradient under a shelf)
  // ---- fields (same names as o_ARRAYS / o_MESH)
  double *tr_arr, *tr_arr_old, *density_m_rho0, *hnode, *hnode_new, *Z_3d_n, *sw_alpha, *sw_beta;
  double *del_ttf, *fct_LO, *fct_ttf_max, *fct_ttf_min, *fct_plus, *fct_minus, *Ki;
  double *bvfreq, *hpressure, *zbar_3d_n, *Wvel, *Wvel_e, *Wvel_i, *CFL_z, *Kv, *tr_z, *adv_flux_ver;
  double *Unode, *Unode_rhs, *sigma_xy, *neutral_slope, *slope_tapered, *U_c;
  double *eta_n, *d_eta, *ssh_rhs, *ssh_rhs_old, *hbar, *hbar_old, *MLD1, *MLD2;
  double *heat_flux, *water_flux, *virtual_salt, *relax_salt, *real_salt_flux;
  double *UV, *UV_rhs, *UV_rhsAB, *tr_xy, *tr_xy_ab, *U_b;
  int tru_nt2;                                            // one-column-per-wave k_tr_update with both tracers per wave (pi)
  int *ale_flag;                                          // zlevel: set when a column needs the local-zstar fallback (not built)
  const double *geo_lat;                                  // SPP: geographic latitude [rad] (N)
  const double *thdgr, *S_oc;                             // SPP: ice growth rate and ocean salinity seen by the ice (N), with the forcing
  const double *m_ice, *m_snow, *press_air, *ssh_gp;      // use_floatice / l_mslp / use_global_tides: potentials of the surface pressure gradient (N), with the forcing
  const double *u_ice, *v_ice, *a_ice;     // use_momix: ice state with the forcing (N)
  double *mixlength;                       // Monin-Obukhov mixing length (N), kept from step to step
  const int *momix_node, *momix_elem;      // 1 where mo_convect applies the Monin-Obukhov mixing (latitude / no cavity), per node and per owned element
  double *bh_tmp;                          // smooth_bh_tra: first stage of the biharmonic tracer filter (nl-1, N) per tracer
  double *KE_node;                         // mom_adv = 3: kinetic energy at nodes (nl-1, N)
  const unsigned char *wall_node;          // mom_adv = 3: 1 for both nodes of the owned boundary edges (KE_node = 0 at lateral walls)
  double *uke, *v_back, *uke_rhs, *uke_rhs_old, *uke_dif, *uke_dis, *uke_back, *UV_dis_tend, *UV_back_tend, *v8_work, *v8_rb;   // visc_option 8: sub-grid energy budget (nl-1, E), tendencies (2, nl-1, E), smoothing work array (2, nl-1, N), Rossby radius (N)
  double *Visc, *vorticity, *leith_aux;    // visc_option 1-3: Leith coefficient (nl-1, E), relative vorticity and smoothing work array (nl-1, N)
  double *pgf_x, *pgf_y, *helem, *Av, *dhe, *stress_surf;
  double *pgf_A, *pgf_B;                   // shchepetkin PGF: the two quotients of the density-Jacobian's vertical derivative that depend on the NODE column only (k_pressure_bv forms them once per node and level; every element around the node reads them)
  double *adv_flux_hor, *adv_flux_raw, *flux_lo_hor, *edge_up_dn_grad, *edge_c12, *diff_flux;
  double *density_ref;        // (nl-1, N) reference density profile of init_ref_density (use_density_ref / cavities); nullptr: density_0
  double *cl_grad; const unsigned char *cl_need;       // fused up/down-wind gradients (DM::use_tile): cluster mean of the tracer gradient per node (2, nl-1, N, tracer) and the nodes some edge needs it for
  double *ssh_values;
  // Gent-McWilliams bolus velocities (kernels_gm.hip)
  double *fer_K, *fer_gamma, *fer_Wvel, *fer_c, *fer_UV;   // (nl,N), (2,nl,N), (nl,N), (N), (2,nl-1,E)
  const double *lat_deg;                                  // (N) geographic latitude in degrees (geo_coord_nod2D(2,:)/rad), Kv0_const = .false. only
  const int *nb_lay;                                      // (N) nboundary_lay (oce_muscl_adv.F90:74-104), tra_adv_hor = MUSCL only
  int exp_batch;                                          // bit mask (FESOM_GPU_EXP_BATCH, default all on): kernels that issue their column gathers as ONE batch of independent loads and select afterwards (1 k_kpp_elem, 2 k_diff_flux, 4 k_vel_nodes, 8 k_visc_node, 16 k_fer_wvel, 64 k_flux_hor_nt: one load per up/down-wind value through a per-lane address select)
  int gm_nzl;                                             // max upper level over the elements of the LAST owned node: where init_Redi_GM copies fer_K into Ki (src/oce_fer_gm.F90:250)
  const double *redi_k0;                                  // (N) K_hor*(mesh_resolution/100km)^2: surface Ki of Redi without GM
  const double *gm_scal_A, *gm_scal_B, *mesh_resolution;    // (N) scaling_Rossby: the resolution scaling and the ramp as separate factors (the Rossby factor is applied first), mesh_resolution
  const double *gm_scal_static;                           // (N) mesh-only part of the horizontal GM scaling
  int *MLD1_ind;                                          // (N) level index of MLD1 (pressure_bv)
  // KPP (kernels_kpp.hip): interior values, boundary layer coefficients (3 slabs of (nl,N)) + two smoothing buffers, tables
  double *dbsfc, *stress_atmoce_x, *stress_atmoce_y, *sw_3d;
  const double *coriolis_node;
  double *kpp_viscA, *kpp_Kv1, *kpp_Kv2, *kpp_blmc, *kpp_sA, *kpp_sB, *kpp_ghats, *kpp_hbl, *kpp_caseA, *kpp_dkm1;
  int *kpp_kbl;
  const double *kpp_wmt, *kpp_wst;
  double kpp_deltaz, kpp_deltau, kpp_Vtc, kpp_cg;
  // Soufflet toy channel (kernels_toy.hip): relaxation targets, zonal means per (level, latitude bin), static bin tables
  double *Tclim, *Uclim, *toy_zvel, *toy_ztem;
  double *Sclim, *relax2clim;              // clim_relax > 0: salinity climatology (nl-1, N) and the relaxation rate (N); Tclim above
  const double *toy_znum, *toy_e_a, *toy_n_a;
  const int *toy_bptr, *toy_bidx, *toy_e_nn, *toy_n_nn;
  // solver workspace
  double *sv_vals, *sv_dinv, *sv_b, *sv_r, *sv_r0, *sv_p, *sv_v, *sv_s, *sv_t, *sv_ph, *sv_x0, *sv_snap, *sv_ph2, *sv_v2, *sv_rdinv;
  int *sv_info; double *sv_resid;   // sv_info[0] iterations of the last solve, [1] number of stored previous solutions
  double *sv_scale;
  double *sv_part, *sv_red, *sv_kry;    // partitioned solve: block partial sums, reduced sums, Krylov scalars + flags (solver.hip)
  double *sv_h1, *sv_h2, *sv_h3; int sv_extrap;   // previous SSH solutions (extrapolated initial guess), only on the step path
  unsigned short *sv_cols;    // static ELL column pattern [k][NP] of the SSH operator (padding -> own row)
  const int *sv_colsi;        // the same with 32-bit indices (multi-workgroup / partitioned phases)
  // one-workgroup solve: rows sorted by their number of entries (stable, descending) so that the 64 rows of a wavefront have the
  // same ELL width and the padding is skipped.  sv_perm[position] = row, sv_inv[row] = position, sv_wid[position / 64] = width.
  // Identity / full width wherever the natural order is needed (multi-workgroup and partitioned phases).
  const int *sv_perm, *sv_inv, *sv_wid;
  // explicit-inverse preconditioner of pi-class operators (csrc/precond_host.cpp, solver.hip "xinv"): fp32 inverse of the frozen
  // row-scaled operator with the entries below 1e-4 of their row's largest dropped, CSR (sv_mp row pointer, sv_mc columns, sv_minv
  // values); natural-order work vectors of the preconditioned BiCGstab; iterations per solve
  const float *sv_minv; const int *sv_mp; const unsigned short *sv_mc; int sv_xi_its;
  int sv_solves;       // host-side: SSH solves launched since init (the default iteration schedule of the explicit-inverse solve depends on it, solver.hip)
  double sv_tol; int sv_maxits;   // stop rule: ||scaled residual|| < sv_tol (0: the reference's 1e-10, bicgstab_ras.c:78), iteration cap (0: 2000)
  double *sv_bn, *sv_x, *sv_pd, *sv_sn, *sv_sh;
  // RAS-Chebyshev preconditioner of operators beyond the explicit inverse (csrc/ras_host.h, solver_ras.hip): patches of the row graph,
  // one workgroup each; all solver vectors then live in the patch order (rs_perm[position] = row, rs_inv[row] = position; halo rows of a
  // partition keep their place behind the owned ones).  rs_colsq = ELL column pattern [k][NP] as positions.
  const int *rs_pinfo, *rs_extq, *rs_perm, *rs_inv, *rs_colsq;
  const float *rs_lv; const unsigned short *rs_lc; const double *rs_dsc, *rs_cheb;   // rs_cheb: [0] 1/theta, [1+k] c1_k, [64+k] c2_k
  int rs_P, rs_NS, rs_rpt, rs_woff, rs_deg;
  // sub-launch of a column kernel over a LIST of columns (partitioned runs: interior columns while a halo exchange is in flight, then the
  // columns whose stencil reaches into the halo): column slot i works on sub_list[i], i < sub_n.  nullptr = all columns.
  const int *sub_list; int sub_n;
  fesom_params p;
};

// 1-based level index nz, 0-based horizontal index (n / e / d)
#define DA2(a, nz, n)      (a)[(size_t)(n) * m.nlm1 + ((nz) - 1)]
#define DA2L(a, nz, n)     (a)[(size_t)(n) * m.nl + ((nz) - 1)]
#define DV2(a, c, nz, e)   (a)[((size_t)(e) * m.nlm1 + ((nz) - 1)) * 2 + ((c) - 1)]
#define DV3(a, c, nz, e)   (a)[((size_t)(e) * m.nlm1 + ((nz) - 1)) * 3 + ((c) - 1)]
#define DV4(a, c, nz, e)   (a)[((size_t)(e) * m.nlm1 + ((nz) - 1)) * 4 + ((c) - 1)]
#define DTR(a, nz, n, t)   (a)[((size_t)(t) * m.N + (n)) * m.nlm1 + ((nz) - 1)]     /* t 0-based */
#define DGS(j, e)          m.gsca[6 * (size_t)(e) + (j) - 1]
#define DECD(j, d)         m.ecd[4 * (size_t)(d) + (j) - 1]

// The same accessors for a WAVE-UNIFORM horizontal index (the column a wavefront works on, or a neighbour column whose index
// came out of v_readlane / a scalar load): the column's base address is formed on the scalar unit and every lane adds the SAME
// 32-bit level offset, so the access is  global_load v, v_off, s[base:base+1]  -- no 64-bit vector address arithmetic per load.
// NEVER use them with an index that differs between the lanes of a wave.
template <class T> __device__ __forceinline__ __attribute__((address_space(1))) T *uni_ptr(T *p) {      // (global address space kept explicit)
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (__attribute__((address_space(1))) T *)(((unsigned long long)hi << 32) | lo);
}
// (the level offset is formed as a 32-bit BYTE offset: that is the form the address-mode selection folds into saddr + voffset)
template <class T> __device__ __forceinline__ __attribute__((address_space(1))) T &uni_at(T *colbase, unsigned byte_off) {
  return *(__attribute__((address_space(1))) T *)((__attribute__((address_space(1))) char *)uni_ptr(colbase) + byte_off);
}
#define UA2(a, nz, n)       uni_at((a) + (size_t)(n) * m.nlm1, ((unsigned)(nz) - 1u) * 8u)
#define UA2L(a, nz, n)      uni_at((a) + (size_t)(n) * m.nl, ((unsigned)(nz) - 1u) * 8u)
#define UV2(a, c, nz, e)    uni_at((a) + (size_t)(e) * m.nlm1 * 2, (((unsigned)(nz) - 1u) * 2u + ((unsigned)(c) - 1u)) * 8u)
#define UV3(a, c, nz, e)    uni_at((a) + (size_t)(e) * m.nlm1 * 3, (((unsigned)(nz) - 1u) * 3u + ((unsigned)(c) - 1u)) * 8u)
#define UV4(a, c, nz, e)    uni_at((a) + (size_t)(e) * m.nlm1 * 4, (((unsigned)(nz) - 1u) * 4u + ((unsigned)(c) - 1u)) * 8u)
#define UTR(a, nz, n, t)    uni_at((a) + ((size_t)(t) * m.N + (n)) * m.nlm1, ((unsigned)(nz) - 1u) * 8u)

#define D_G 9.81
#define D_RHO0 1030.0
#define D_REARTH 6367500.0
#define D_VCPW 4.2e6

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// one wavefront = one column: the column index is wave-uniform, say so to the compiler (scalar loads / scalar address math)
// Workgroups are dealt round-robin to the 8 XCDs (block b -> XCD b % 8), each with its own L2.  Consecutive columns are
// neighbours on the mesh (partition-sorted numbering) and share most of their gathers, so every XCD gets a CONTIGUOUS range
// of column blocks instead of every 8th one: bijective remap of blockIdx.x, affinity for speed only (nothing relies on it).
__device__ __forceinline__ int xcd_block() {
  const unsigned b = blockIdx.x, nb = gridDim.x, q = nb >> 3, r = nb & 7u, x = b & 7u, j = b >> 3;
  return (int)(x * q + (x < r ? x : r) + j);
}
// column slot -> column: the identity, or an entry of the launch's column list (DM::sub_list; slots past its end give a column no kernel owns)
__device__ __forceinline__ int sub_col(const DM &m, int slot) { return m.sub_list ? (slot < m.sub_n ? m.sub_list[slot] : 0x7fffffff) : slot; }
__device__ __forceinline__ int col_id(const DM &m) { return __builtin_amdgcn_readfirstlane(sub_col(m, xcd_block() * COLS_PER_BLOCK + (threadIdx.x >> 6))); }
// broadcast of lane `src`; src must be wave-uniform (it always is a level index of the wave's column): v_readlane, no LDS
__device__ __forceinline__ double bcast(double x, int src) {
  int lo = __builtin_amdgcn_readlane(__double2loint(x), src), hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int rdlane(int x, int lane) { return __builtin_amdgcn_readlane(x, lane); }   // lane must be wave-uniform
__device__ __forceinline__ double shup(double x) { return __shfl_up(x, 1, 64); }     // value of lane-1
__device__ __forceinline__ double shdn(double x) { return __shfl_down(x, 1, 64); }   // value of lane+1

// -(p_eta + p_ice + p_air) [- ssh_gp] at a node: the surface potentials of compute_vel_rhs (src/oce_ale_vel_rhs.F90:52-76)
__device__ __forceinline__ double surf_pre(const DM &m, int n) {
  double p_ice = 0.0, p_air = 0.0;
  if (m.p.use_floatice) { p_ice = (m.m_ice[n] * 910. + m.m_snow[n] * 290.) * (1. / 1025.); p_ice = 9.81 * (p_ice < m.p.max_ice_loading ? p_ice : m.p.max_ice_loading); }
  if (m.p.l_mslp) p_air = m.press_air[n] / 1000;
  double pre = -(9.81 * m.eta_n[n] + p_ice + p_air);
  if (m.p.use_global_tides) pre = pre - m.ssh_gp[n];
  return pre;
}
// mo(nz, node) of mo_convect (src/oce_mo_conv.F90:44-52): momix_kv inside the mixing length of a node the scheme applies to, else 0
__device__ __forceinline__ double momix_mo(const DM &m, int nz, int n) {
  if (!m.momix_node[n] || nz < m.ulev_n[n] + 1 || nz > m.nlev_n[n] - 1) return 0.0;
  return fabs(DA2L(m.zbar_3d_n, nz, n)) <= m.mixlength[n] ? m.p.momix_kv : 0.0;
}
// Many quotients with the SAME denominator (x * dt / areasvol for every edge of a node): the device's IEEE fp64 division is
// the sequence  ds = div_scale(d), r = rcp(ds) refined by two Newton steps, ns = div_scale(n), q0 = ns*r,
// rem = fma(-ds, q0, ns), q = div_fmas(rem, r, q0), div_fixup  -- and everything up to r depends on the denominator only.
// rcp_prepare() forms r once, div_by() finishes a quotient with the remaining three operations: the same instructions on the
// same values as `n / d`, hence the same bits, as long as div_scale leaves both operands unscaled and div_fixup passes q
// through: d and n normal with exponents in 2^-255 .. 2^256 (the quotient is then far from the exponent limits too), or
// n = 0.  A zero numerator gives +0 here where the division gives a zero with the sign of the quotient: use div_by only
// where the quotient is ADDED to an accumulator that is not -0 (x + (+-0) == x), as all call sites do.
// div_by() records in `bad` whether a lane left that range; the caller then redoes the column with the plain division
// (cold path; never observed on ocean data: fluxes are exactly zero or many orders of magnitude inside the range).
struct RcpD { double d, r; };
__device__ __forceinline__ RcpD rcp_prepare(double d, bool &bad) {
  RcpD k;
  k.d = d;
  double r0 = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, r0, 1.0);
  double r1 = __builtin_fma(r0, e, r0);
  e = __builtin_fma(-d, r1, 1.0);
  k.r = __builtin_fma(r1, e, r1);
  const unsigned ex = ((unsigned)__double2hiint(d) >> 20) & 0x7ffu;
  bad = bad || (ex - 0x300u) > 0x1ffu;
  return k;
}
__device__ __forceinline__ double div_by(double n, const RcpD &k, bool &bad) {
  const unsigned hn = (unsigned)__double2hiint(n) & 0x7fffffffu;
  bad = bad || (((hn >> 20) - 0x300u) > 0x1ffu && (hn | (unsigned)__double2loint(n)) != 0u);
  const double q0 = n * k.r;
  const double rem = __builtin_fma(-k.d, q0, n);
  return __builtin_fma(rem, k.r, q0);
}

// Sequential (reference-order) running sums across the lanes of one wavefront, bit-identical to the scalar loop
//   acc = init; for j = first..last: acc = acc + x[j]      (lane j keeps the partial sum after element j).
// Ripple form: every lane holds y (start: init); one step is  y[l] = y[l-1] + x[l]  for ALL lanes at once (DPP wavefront shift,
// lane 0 takes init; x = 0 outside [first, last]).  After k steps the lanes first .. first+k-1 hold their final value and keep
// it (their left neighbour no longer changes), so last-first+1 steps finish the column: 2 DPP moves + 1 add per step, no
// broadcast through SGPRs, no per-step selects.  (init + 0.0 == init for every init but -0.0, which no caller passes.)
__device__ __forceinline__ double dpp_wave_shr1(double y, double fill) {      // lane l <- y[l-1], lane 0 <- fill
  int lo = __builtin_amdgcn_update_dpp(__double2loint(fill), __double2loint(y), 0x138, 0xf, 0xf, false);
  int hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), __double2hiint(y), 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_wave_shl1(double y, double fill) {      // lane l <- y[l+1], lane 63 <- fill
  int lo = __builtin_amdgcn_update_dpp(__double2loint(fill), __double2loint(y), 0x130, 0xf, 0xf, false);
  int hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), __double2hiint(y), 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double seq_sum_up(double x, int first, int last, double init) {
  const int l = lane_id();
  first = __builtin_amdgcn_readfirstlane(first); last = __builtin_amdgcn_readfirstlane(last);      // level bounds of the wave's column: uniform
  const bool in = l >= first && l <= last;
  const double xi = in ? x : 0.0;
  double y = init;
  for (int k = last - first + 1; k > 0; --k) y = dpp_wave_shr1(y, init) + xi;
  return in ? y : init;
}
__device__ __forceinline__ double seq_sum_down(double x, int first, int last, double init) {   // j = first, first-1, ..., last
  const int l = lane_id();
  first = __builtin_amdgcn_readfirstlane(first); last = __builtin_amdgcn_readfirstlane(last);
  const bool in = l <= first && l >= last;
  const double xi = in ? x : 0.0;
  double y = init;
  for (int k = first - last + 1; k > 0; --k) y = dpp_wave_shl1(y, init) + xi;
  return in ? y : init;
}
__device__ __forceinline__ double wave_max(double x) {
  for (int s = 32; s >= 1; s >>= 1) x = fmax(x, __shfl_xor(x, s, 64));
  return x;
}
__device__ __forceinline__ double dmin_(double a, double b) { return a < b ? a : b; }
__device__ __forceinline__ double dmax_(double a, double b) { return a > b ? a : b; }

// ---- in-block Thomas sweep (impl_vert_visc_ale src/oce_ale.F90:2491-2510, diff_ver_part_impl_ale
// src/oce_ale_tracer.F90:838-852).  The sweep is sequential in z and latency-bound (a dependent fp64 divide per level),
// so its duration does not depend on how many columns a wavefront solves: the kernels that assemble the coefficients
// (wave per column, lane = level) hand them over in LDS, wave 0 of the block solves the block's TH_COLS columns (lane =
// column) and every wave picks its solution up again -- no extra launch, no round trip through global memory.
// Arithmetic order is the reference's.  Must be called by every thread of the block.
#define GATHER_MAXD 12                  // incident edges fetched in one batch by the node-gather kernels (more: remainder loop)
#define TH_COLS 8
#define TH_CP (TH_COLS + 1)
#define TH_BLOCK (WAVE * TH_COLS)
__device__ __forceinline__ int col_id_th(const DM &m) { return __builtin_amdgcn_readfirstlane(sub_col(m, xcd_block() * TH_COLS + (threadIdx.x >> 6))); }
static inline size_t thomas_lds_bytes(int nlm1, int nrhs) { return (size_t)(3 + nrhs) * nlm1 * TH_CP * sizeof(double) + 2 * TH_COLS * sizeof(int); }
static inline int nblocks_th(int ncol) { return (ncol + TH_COLS - 1) / TH_COLS; }
#define LAUNCH_TH(k, ncol, nrhs, ...) hipLaunchKernelGGL(k, dim3(nblocks_th(ncol)), dim3(TH_BLOCK), thomas_lds_bytes(m.nlm1, nrhs), s, __VA_ARGS__)

// Kv0_background_qiang (src/oce_ale_mixing_pp.F90:91-125): background vertical diffusivity by latitude [deg] and depth [m]
__device__ __forceinline__ double kv0_background_qiang(double lat, double dep) {
  const double aux = (0.6 + 1.0598 / 3.1415926 * atan(4.5e-3 * (dep - 2500.0))) * 1.0e-5;
  double ratio;
  if (fabs(lat) < 5.0) ratio = 1.0;
  else ratio = fmin(1.0 + 9.0 * (fabs(lat) - 5.0) / 10.0, 10.0);
  if (lat > 70.0) {
    if (dep <= 50.0) ratio = 4.0 + 6.0 * (50.0 - dep) / 50.0;
    else ratio = 4.0;
  }
  return aux * ratio;
}

// Tile of COLS columns whose tridiagonal systems (NRHS right-hand sides) are solved by ONE wavefront with lane = column.
// LDS image: [3 + NRHS arrays: a, b, c, r1 (, r2)][level][COLS + 1 (padding)] doubles, then the level range of every column.
//   put(ci, ...)  -- by the wave that assembled column ci (lane = level)          } a kernel calls put for all its columns,
//   sweep()       -- whole block: barrier, wave 0 solves all columns, barrier     } then sweep once, then get
//   get(ci, ...)  -- solution of column ci back to lane = level
// Two shapes are instantiated: COLS = 8, one column per wave (pi: latency-bound, the columns' values stay in registers across
// the sweep) and COLS = 32 / 64 with several columns per wave (CORE2-class meshes: the sweep is amortised over a full tile).
template <int NRHS, int COLS>
struct ThTile {
  double *sh; int nl1, astr; int *rng;
  static constexpr int CP = COLS + 1;
  __device__ __forceinline__ ThTile(double *s, int nl1_) : sh(s), nl1(nl1_), astr(nl1_ * CP) { rng = (int *)(sh + (size_t)(3 + NRHS) * astr); }
  static size_t lds_bytes(int nlm1) { return (size_t)(3 + NRHS) * nlm1 * CP * sizeof(double) + 2 * COLS * sizeof(int); }
  __device__ __forceinline__ void put(int ci, bool valid, int kmin, int kmax, double a, double b, double c, double r1, double r2) {
    const int l = threadIdx.x & 63, nz = l + 1;
    if (l == 0) { rng[2 * ci] = valid ? kmin : 1; rng[2 * ci + 1] = valid ? kmax : 0; }
    if (nz <= nl1) {
      double *p = sh + (nz - 1) * CP + ci;
      p[0] = a; p[astr] = b; p[2 * astr] = c; p[3 * astr] = r1;
      if (NRHS == 2) p[4 * astr] = r2;
    }
  }
  // the same in two parts (tiles with several right-hand sides assembled one after the other)
  __device__ __forceinline__ void put_abc(int ci, bool valid, int kmin, int kmax, double a, double b, double c) {
    const int l = threadIdx.x & 63, nz = l + 1;
    if (l == 0) { rng[2 * ci] = valid ? kmin : 1; rng[2 * ci + 1] = valid ? kmax : 0; }
    if (nz <= nl1) { double *p = sh + (nz - 1) * CP + ci; p[0] = a; p[astr] = b; p[2 * astr] = c; }
  }
  __device__ __forceinline__ void get_abc(int ci, double &a, double &b, double &c) const {      // what put_abc stored (same wave)
    const int nz = (threadIdx.x & 63) + 1;
    a = 0.0; b = 1.0; c = 0.0;
    if (nz <= nl1) { const double *p = sh + (nz - 1) * CP + ci; a = p[0]; b = p[astr]; c = p[2 * astr]; }
  }
  __device__ __forceinline__ void put_rhs(int ci, int which, double r) {
    const int nz = (threadIdx.x & 63) + 1;
    if (nz <= nl1) sh[(size_t)(3 + which) * astr + (nz - 1) * CP + ci] = r;
  }
  __device__ __forceinline__ void get(int ci, double &x1, double &x2) const {
    const int nz = (threadIdx.x & 63) + 1;
    x1 = 0.0; x2 = 0.0;
    if (nz <= nl1) {
      const double *p = sh + (nz - 1) * CP + ci;
      x1 = p[3 * astr];
      if (NRHS == 2) x2 = p[4 * astr];
    }
  }
  __device__ __forceinline__ void sweep() {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (w == 0) {
      // Branch-free and software-pipelined: the first active level uses a = 0 (b - cp*0 = b, r - x*0 = r exactly, i.e. the
      // reference's c/b, r/b start), inactive levels are discarded by selects, so the independent divide chains (cp, x1,
      // x2) interleave in straight-line code and  cp_j = c_j / (b_j - cp_{j-1} a_j)  is the only serial part.
      const int lc = l < COLS ? l : COLS - 1;
      int kmn = 1, kmxl = 0;
      if (l < COLS) { kmn = rng[2 * l]; kmxl = rng[2 * l + 1]; }
      int kmx = kmxl;
      for (int s = 32; s >= 1; s >>= 1) kmx = max(kmx, __shfl_xor(kmx, s, 64));
      kmx = __builtin_amdgcn_readfirstlane(kmx);
      double *p0 = sh + lc;
      double cpp = 0.0, x1p = 0.0, x2p = 0.0;
      double ca = p0[0], cb = p0[astr], cc = p0[2 * astr], c1 = p0[3 * astr], c2 = (NRHS == 2) ? p0[4 * astr] : 0.0;
      double *pj = p0;
      for (int j = 1; j <= kmx; j++) {
        double *pn = (j < kmx) ? pj + CP : pj;
        double na = pn[0], nb = pn[astr], nc = pn[2 * astr], n1 = pn[3 * astr], n2 = (NRHS == 2) ? pn[4 * astr] : 0.0;
        const bool act = (j >= kmn) && (j <= kmxl);
        const double am = (j == kmn) ? 0.0 : ca;
        double mm = cb - cpp * am;
        double ncp = cc / mm;
        double nx1 = (c1 - x1p * am) / mm;
        double nx2 = (NRHS == 2) ? (c2 - x2p * am) / mm : 0.0;
        if (act) {
          cpp = ncp; x1p = nx1; x2p = nx2;
          pj[2 * astr] = ncp; pj[3 * astr] = nx1;
          if (NRHS == 2) pj[4 * astr] = nx2;
        }
        ca = na; cb = nb; cc = nc; c1 = n1; c2 = n2;
        pj = pn;
      }
      double y1 = 0.0, y2 = 0.0;
      pj = p0 + (kmx > 0 ? kmx - 1 : 0) * CP;
      double cp = pj[2 * astr], u1 = pj[3 * astr], u2 = (NRHS == 2) ? pj[4 * astr] : 0.0;
      for (int j = kmx; j >= 1; j--) {
        double *pn = (j > 1) ? pj - CP : pj;
        double ncp = pn[2 * astr], nu1 = pn[3 * astr], nu2 = (NRHS == 2) ? pn[4 * astr] : 0.0;
        if (j >= kmn && j <= kmxl) {
          if (j == kmxl) { y1 = u1; if (NRHS == 2) y2 = u2; }
          else {
            y1 = u1 - cp * y1;
            if (NRHS == 2) y2 = u2 - cp * y2;
          }
          pj[3 * astr] = y1;
          if (NRHS == 2) pj[4 * astr] = y2;
        }
        cp = ncp; u1 = nu1; u2 = nu2;
        pj = pn;
      }
    }
    __syncthreads();
  }
};

// one column per wave, TH_COLS columns per block (the pi shape): must be called by every thread of the block
template <int NRHS>
__device__ __forceinline__ void thomas_inblock(double *sh, int nl1, bool valid, int kmin, int kmax, double a, double b, double c, double r1,
                                               double r2, double &x1, double &x2) {
  ThTile<NRHS, TH_COLS> t(sh, nl1);
  const int w = threadIdx.x >> 6;
  t.put(w, valid, kmin, kmax, a, b, c, r1, r2);
  t.sweep();
  t.get(w, x1, x2);
}

// CORE2-class meshes: tiles of TL_COLS columns per block of TL_WAVES wavefronts (TL_COLS / TL_WAVES columns per wave)
// DM::use_tile selects the shape: 1 = 32 columns x 8 waves (default), 2 = 32 x 4, 3 = 64 x 8, 4 = 64 x 4 (FESOM_GPU_TILE=<n>)
#define TL_COLS 32
#define TL_WAVES 8
#define TL_BLOCK (WAVE * TL_WAVES)
#define TILE_SHAPES(X) X(1, 32, 8) X(2, 32, 4) X(3, 64, 8) X(4, 64, 4)
static inline int nblocks_tl(int ncol) { return (ncol + TL_COLS - 1) / TL_COLS; }
// the tile kernels replace the one-column-per-wave kernels when the mesh has at least this many node columns (or when
// FESOM_GPU_TILE=1 / 0 forces / forbids them; decided once in fesom_gpu_init: DM::use_tile)
#define TL_MIN_COLUMNS 20000

static inline int nblocks(int ncol) { return (ncol + COLS_PER_BLOCK - 1) / COLS_PER_BLOCK; }
#define SUBN(m_, n_) ((m_).sub_list ? (m_).sub_n : (n_))       /* column slots of a launch: all n_ columns, or the launch's column list */

// launchers implemented in the kernel translation units
void launch_dynamics_pre(const DM &m, hipStream_t s, int first_step);
void launch_momix(const DM &m, hipStream_t s);      // mo_length of mo_convect (use_momix): before the fused mixing kernels
void launch_ssh_rhs(const DM &m, hipStream_t s);
int  launch_solver(const DM &m, hipStream_t s, int fuse_rhs = 0, int scale_done = 0);
void launch_row_scale(const DM &m, hipStream_t s);
void launch_dynamics_post(const DM &m, hipStream_t s);
void launch_tracer(const DM &m, hipStream_t s, int tr);
void launch_thickness(const DM &m, hipStream_t s, bool bolus_remove = false);
int  launch_named_toy(const DM &m, hipStream_t s, const char *name);
int  launch_named_gm(const DM &m, hipStream_t s, const char *name);
int  launch_named_kpp(const DM &m, hipStream_t s, const char *name);
void launch_step_info(const DM &m, hipStream_t s, double *col, double *out);
int  launch_named_dsolve(const DM &m, hipStream_t s, const char *name);
// one neighbour exchange through the library's RCCL communicator (csrc/api.hip) for a caller with its own lists and buffers (the sea-ice
// context): blocks of (ptr[p+1]-ptr[p]) * W doubles per neighbour, consecutive in sPE / rPE order; 1-based CSR pointers as in com_struct
#include <string>
int  fesom_internal_select_device(std::string &err);      // csrc/api.hip: the device of this rank (FESOM_GPU_DEVICE / LOCAL_RANK), shared by all contexts
int  fesom_internal_rccl_exchange(int ns, const int *sPE, const int *sptr, int nr, const int *rPE, const int *rptr, double *sd, double *rd, int W, hipStream_t s);
