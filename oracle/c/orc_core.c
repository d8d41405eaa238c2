/* ORACLE (test infrastructure): context, field registry, SSH solver restatement, step driver. */
#include "orc.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

orc_ctx C_;

typedef struct { const char *name; double **p; size_t cnt; } field_t;
#define ORC_MAXF 192
static field_t F_[ORC_MAXF];
static int nF_ = 0;

static void reg(const char *name, double **p, size_t cnt) {
  *p = (double *)calloc(cnt ? cnt : 1, sizeof(double));
  if (nF_ >= ORC_MAXF) { fprintf(stderr, "orc: field registry full\n"); abort(); }
  F_[nF_].name = name; F_[nF_].p = p; F_[nF_].cnt = cnt; nF_++;
}

int orc_ale_flag = 0;
int orc_get_ale_flag(void) { return orc_ale_flag; }
int orc_init(const fesom_mesh_desc *m, const fesom_params *p) {
  orc_ale_flag = 0;
  for (int i = 0; i < nF_; i++) free(*F_[i].p);
  nF_ = 0;
  free(C_.toy_bpos); free(C_.toy_owner); free(C_.MLD1_ind);
  free(C_.kpp_wmt); free(C_.kpp_wst); free(C_.kpp_work); free(C_.kpp_vol); free(C_.kpp_kbl);
  memset(&C_, 0, sizeof(C_));
  { extern void orc_solver_reset(void); orc_solver_reset(); }
  C_.m = *m; C_.p = *p;
  C_.N = m->myDim_nod2D + m->eDim_nod2D; C_.E = m->myDim_elem2D + m->eDim_elem2D; C_.D = m->myDim_edge2D + m->eDim_edge2D;
  C_.nl = m->nl; C_.nlm1 = m->nl - 1; C_.ntr = p->num_tracers;
  size_t N = C_.N, E = C_.E, D = C_.D, nl = C_.nl, n1 = C_.nlm1;
#define R(f, c) reg(#f, &C_.f, c)
  R(tr_arr, n1 * N * C_.ntr); R(tr_arr_old, n1 * N * C_.ntr);
  R(density_m_rho0, n1 * N); R(density_ref, n1 * N); R(hnode, n1 * N); R(hnode_new, n1 * N); R(Z_3d_n, n1 * N);
  R(sw_alpha, n1 * N); R(sw_beta, n1 * N); R(del_ttf, n1 * N); R(del_ttf_advhoriz, n1 * N); R(del_ttf_advvert, n1 * N);
  R(fct_LO, n1 * N); R(fct_ttf_max, n1 * N); R(fct_ttf_min, n1 * N); R(fct_plus, n1 * N); R(fct_minus, n1 * N);
  R(Ki, n1 * N); R(Tclim, n1 * N); R(Sclim, n1 * N); R(relax2clim, N);
  R(bvfreq, nl * N); R(hpressure, nl * N); R(zbar_3d_n, nl * N); R(Wvel, nl * N); R(Wvel_e, nl * N); R(Wvel_i, nl * N);
  R(CFL_z, nl * N); R(Kv, nl * N); R(tr_z, nl * N); R(adv_flux_ver, nl * N); R(dbsfc, nl * N);
  R(Unode, 2 * n1 * N); R(Unode_rhs, 2 * n1 * N); R(sigma_xy, 2 * n1 * N); R(neutral_slope, 3 * n1 * N); R(slope_tapered, 3 * n1 * N);
  R(U_c, 2 * n1 * N);
  R(eta_n, N); R(d_eta, N); R(ssh_rhs, N); R(ssh_rhs_old, N); R(hbar, N); R(hbar_old, N); R(MLD1, N); R(MLD2, N);
  R(heat_flux, N); R(water_flux, N); R(virtual_salt, N); R(relax_salt, N); R(real_salt_flux, N);
  R(u_ice, N); R(v_ice, N); R(a_ice, N); R(mixlength, N); R(m_ice, N); R(m_snow, N); R(press_air, N); R(ssh_gp, N); R(thdgr, N); R(S_oc_array, N);
  R(UV, 2 * n1 * E); R(UV_rhs, 2 * n1 * E); R(UV_rhsAB, 2 * n1 * E); R(tr_xy, 2 * n1 * E); R(U_b, 2 * n1 * E); R(fct_ebnd, 2 * n1 * E);
  R(pgf_x, n1 * E); R(pgf_y, n1 * E); R(helem, n1 * E); R(Av, nl * E); R(dhe, E); R(stress_surf, 2 * E);
  R(Visc, n1 * E); R(vorticity, n1 * N); R(leith_aux, n1 * N); R(KE_node, n1 * N);
  R(uke, n1 * E); R(v_back, n1 * E); R(uke_rhs, n1 * E); R(uke_rhs_old, n1 * E); R(uke_dif, n1 * E); R(uke_dis, n1 * E); R(uke_back, n1 * E);
  R(UV_dis_tend, 2 * n1 * E); R(UV_back_tend, 2 * n1 * E);
  R(adv_flux_hor, n1 * D); R(edge_up_dn_grad, 4 * n1 * D);
  R(fer_K, nl * N); R(fer_gamma, 2 * nl * N); R(fer_Wvel, nl * N); R(fer_c, N); R(fer_scal, N); R(gm_scal_static, N); R(fer_UV, 2 * n1 * E);
  R(stress_atmoce_x, N); R(stress_atmoce_y, N); R(sw_3d, nl * N); R(kpp_sw_node, N);
  R(kpp_Kv1, nl * N); R(kpp_Kv2, nl * N); R(kpp_viscA, nl * N); R(kpp_dVsq, nl * N); R(kpp_ghats, n1 * N);
  reg("kpp_blmc1", &C_.kpp_blmc[0], nl * N); reg("kpp_blmc2", &C_.kpp_blmc[1], nl * N); reg("kpp_blmc3", &C_.kpp_blmc[2], nl * N);
  R(kpp_hbl, N); R(kpp_bfsfc, N); R(kpp_caseA, N); R(kpp_stable, N); R(kpp_ustar, N); R(kpp_Bo, N); R(kpp_dkm1, 3 * N);
  R(Uclim, n1 * E); R(toy_zvel, n1 * 100); R(toy_ztem, n1 * 100); R(toy_znum, n1 * 100);
  R(ssh_values, m->ssh_nza); R(sv_h1, N); R(sv_h2, N); R(sv_h3, N);
#undef R
  C_.toy_bpos = calloc(E ? E : 1, sizeof(int));
  C_.MLD1_ind = calloc(N ? N : 1, sizeof(int));
  for (size_t i = 0; i < n1 * N; i++) C_.density_ref[i] = DENSITY_0;
  for (size_t i = 0; i < nl * N; i++) C_.fer_K[i] = 500.0;                              /* oce_setup_step.F90:359 (read back under an ice shelf: init_Redi_GM scales fer_k(ulevels_nod2D) where the template went to another level) */
  memcpy(C_.ssh_values, m->ssh_values, sizeof(double) * m->ssh_nza);
  /* Ki = K_hor*(mesh_resolution/100000)**2  (oce_setup_step.F90:328-331) */
  for (size_t n = 0; n < N; n++) {
    double r = m->mesh_resolution[n] / 100000.0;
    for (size_t k = 0; k < n1; k++) C_.Ki[n * n1 + k] = p->K_hor * (r * r);
  }
  for (size_t n = 0; n < N; n++) C_.kpp_sw_node[n] = (double)m->myDim_nod2D;
  if (p->mix_scheme == 0) {               /* no mixing scheme (an option of this build, not of the reference): constant A_ver / K_ver */
    for (size_t i = 0; i < nl * E; i++) C_.Av[i] = p->A_ver;
    for (size_t i = 0; i < nl * N; i++) C_.Kv[i] = p->K_ver;
  }
  { extern void orc_gm_static(void); if (p->Fer_GM) orc_gm_static(); }
  return 0;
}

static field_t *find(const char *name) {
  for (int i = 0; i < nF_; i++) if (!strcmp(F_[i].name, name)) return &F_[i];
  return NULL;
}
int orc_set_field(const char *name, const double *in, long long cnt) {
  field_t *f = find(name);
  if (!f || (size_t)cnt != f->cnt) { fprintf(stderr, "orc_set_field(%s): bad name or count %lld (want %zu)\n", name, cnt, f ? f->cnt : 0); return 1; }
  memcpy(*f->p, in, sizeof(double) * cnt);
  return 0;
}
int orc_get_field(const char *name, double *out, long long cnt) {
  field_t *f = find(name);
  if (!f || (size_t)cnt != f->cnt) { fprintf(stderr, "orc_get_field(%s): bad name or count %lld (want %zu)\n", name, cnt, f ? f->cnt : 0); return 1; }
  memcpy(out, *f->p, sizeof(double) * cnt);
  return 0;
}
long long orc_field_count(const char *name) { field_t *f = find(name); return f ? (long long)f->cnt : -1; }
void orc_set_first_step_done(int v) { C_.first_step_done = v; }
int orc_solver_iterations(void) { return C_.solver_iters; }
double orc_solver_residual(void) { return C_.solver_resid; }

/* ---- SSH solve.  Boundary: solve_ssh_ale src/oce_ale.F90:2210-2344 -> psolve src/psolve.c:152-221.
 * Row scaling scale[i]=1/sum_j|a_ij| and y=rhs*scale, warm start x=d_eta, and the stopping rule
 * ||r||^2 < tol^2 (tol=1e-10 absolute on the row-scaled residual, lib/parms/src/bicgstab_ras.c:78,146,220)
 * follow the reference.  The preconditioner is this build's GPU design (Jacobi, applied as the column scaling
 * B = A_s D^-1, y = D x, instead of pARMS' sequential RAS+ILU(2)): the solution agrees with the reference to
 * the solver tolerance, not bit for bit.  Dot products use the fixed reduction order of the HIP kernel
 * (SOLVER_T partial sums with stride SOLVER_T, a fixed tree inside each block of 64, then the same 16-wide tree over the 16 block sums) so that oracle == HIP bitwise. */
#define SOLVER_T 1024
/* Row order of the one-workgroup HIP solve: rows sorted by their number of entries (stable, descending), thread t owns the rows
 * at positions t, t+1024, ...; NULL = natural order.  Only the dot products depend on it. */
static const int *solver_perm = NULL;
static double row_tree16(double *x) {                    /* x[l] += x[l-s], s = 8,4,2,1 ; total in x[15] */
  for (int s = 8; s >= 1; s >>= 1)
    for (int l = 15; l >= 16 - s; l--) x[l] = x[l] + x[l - s];
  return x[15];
}
static double dot_fixed(const double *x, const double *y, int n) {
  static double part[SOLVER_T];
  double wave[16];
  for (int t = 0; t < SOLVER_T; t++) {
    double s = 0.0;
    for (int i = t; i < n; i += SOLVER_T) { const int r_ = solver_perm ? solver_perm[i] : i; s = s + x[r_] * y[r_]; }
    part[t] = s;
  }
  for (int w = 0; w < 16; w++) {                         /* one wave = 4 rows of 16 partial sums */
    double r0 = row_tree16(part + 64 * w), r1 = row_tree16(part + 64 * w + 16), r2 = row_tree16(part + 64 * w + 32),
           r3 = row_tree16(part + 64 * w + 48);
    wave[w] = (r3 + r2) + (r1 + r0);
  }
  return row_tree16(wave);
}
/* operators with more than 4096 rows take the multi-workgroup phases of the HIP path: one product per thread, halving tree
 * inside every block of 256 threads (strides 128..1); the block sums are then added 256-strided and tree-reduced the same way */
static double dot_blocks(const double *x, const double *y, int n) {
  double sh[256], col[256];
  int nb = 0;
  for (int t = 0; t < 256; t++) col[t] = 0.0;
  for (int b0 = 0; b0 < n; b0 += 256, nb++) {
    for (int t = 0; t < 256; t++) sh[t] = (b0 + t < n) ? x[b0 + t] * y[b0 + t] : 0.0;
    for (int s = 128; s >= 1; s >>= 1)
      for (int t = 0; t < s; t++) sh[t] = sh[t] + sh[t + s];
    col[nb % 256] = col[nb % 256] + sh[0];                 /* thread nb%256 of the summing block adds block nb's partial */
  }
  for (int s = 128; s >= 1; s >>= 1)
    for (int t = 0; t < s; t++) col[t] = col[t] + col[t + s];
  return col[0];
}
static double dot_sel(const double *x, const double *y, int n) { return n > 4 * SOLVER_T ? dot_blocks(x, y, n) : dot_fixed(x, y, n); }
/* explicit-inverse preconditioner of the HIP path for pi-class operators (orc_xinv.c): built once per context from the operator the
 * run starts with (the mesh's ssh_values, as fesom_gpu_init does), frozen afterwards */
int orc_xinv_build(int n, const int *rp, const int *ci, const double *vals, int ld, float *out);
void orc_xinv_sparsify(int n, int ld, const float *M, double tau, int *rowptr, unsigned short *cols, float *vals);
void orc_xinv_apply(int n, const int *mp, const unsigned short *mc, const float *mv, const double *x, double *z);
#define XINV_DROP 1.0e-4                      /* csrc/api.hip */
static float *xinv_M = NULL;                  /* sparsified inverse: CSR values / columns / row pointer */
static unsigned short *xinv_mc = NULL;
static int *xinv_mp = NULL;
static int xinv_n = 0, xinv_solves = 0;
static const void *xinv_key = NULL;
/* RAS-Chebyshev preconditioner of the HIP path for operators beyond the explicit inverse (orc_ras.c), frozen the same way */
#include "orc_ras.h"
static orc_ras_plan ras_plan;
static const void *ras_key = NULL;
static int ras_n = 0;
void orc_solver_reset(void) {
  free(xinv_M); free(xinv_mc); free(xinv_mp); xinv_M = NULL; xinv_mc = NULL; xinv_mp = NULL; xinv_n = 0; xinv_key = NULL; xinv_solves = 0;
  if (ras_key) orc_ras_free(&ras_plan);
  ras_key = NULL; ras_n = 0;
}

void orc_solve_ssh(void) {
  int n = C_.m.myDim_nod2D;
  const int *rp = C_.m.ssh_rowptr, *ci = C_.m.ssh_colind_loc;
  int off = rp[0];
  int *perm = NULL;
  int maxnnz = 0;
  for (int i = 0; i < n; i++) maxnnz = rp[i + 1] - rp[i] > maxnnz ? rp[i + 1] - rp[i] : maxnnz;
  if (n <= 4 * SOLVER_T && maxnnz <= 10) {               /* (csrc/api.hip: solver_row_order) */
    perm = malloc(sizeof(int) * n);
    int q = 0;
    for (int w = maxnnz; w >= 0; w--) for (int i = 0; i < n; i++) if (rp[i + 1] - rp[i] == w) perm[q++] = i;
  }
  double *B = malloc(sizeof(double) * C_.m.ssh_nza), *As = malloc(sizeof(double) * C_.m.ssh_nza), *diag = malloc(sizeof(double) * n * 14);
  double *dinv = diag + n, *b = dinv + n, *r = b + n, *r0 = r + n, *pv = r0 + n, *v = pv + n, *s = v + n, *t = s + n, *y = t + n;
  double *ph = y + n, *sh = ph + n, *xx = sh + n;
  double *x = C_.d_eta;
  /* initial guess: previous solution (reference) or quadratic / cubic extrapolation of the last solutions */
  for (int i = 0; i < n; i++) {
    double xi = x[i], x0 = xi;
    if (C_.p.solver_x0_order == 2 && C_.sv_nhist >= 2) x0 = (3.0 * xi - 3.0 * C_.sv_h1[i]) + C_.sv_h2[i];
    if (C_.p.solver_x0_order == 3 && C_.sv_nhist >= 3) x0 = ((4.0 * xi - 6.0 * C_.sv_h1[i]) + 4.0 * C_.sv_h2[i]) - C_.sv_h3[i];
    if (C_.p.solver_x0_order == 3 && C_.sv_nhist == 2) x0 = (3.0 * xi - 3.0 * C_.sv_h1[i]) + C_.sv_h2[i];
    C_.sv_h3[i] = C_.sv_h2[i]; C_.sv_h2[i] = C_.sv_h1[i]; C_.sv_h1[i] = xi;
    x[i] = x0;
  }
  if (C_.sv_nhist < 3) C_.sv_nhist++;
  for (int i = 0; i < n; i++) {
    double tmp = 0.;
    for (int j = rp[i] - off; j < rp[i + 1] - off; j++) tmp += fabs(C_.ssh_values[j]);
    double sc = 1. / tmp;
    for (int j = rp[i] - off; j < rp[i + 1] - off; j++) { As[j] = C_.ssh_values[j] * sc; B[j] = As[j]; }
    b[i] = C_.ssh_rhs[i] * sc;
    diag[i] = B[rp[i] - off];                   /* first entry of a row is the diagonal (oce_ale.F90:1128-1151) */
    dinv[i] = 1.0 / diag[i];
  }
  for (int i = 0; i < n; i++)
    for (int j = rp[i] - off; j < rp[i + 1] - off; j++) B[j] = B[j] * dinv[ci[j] - 1];
#define SPMV_(M_, out, in) for (int i = 0; i < n; i++) { double a = 0.0; for (int j = rp[i] - off; j < rp[i + 1] - off; j++) a = a + (M_)[j] * (in)[ci[j] - 1]; (out)[i] = a; }
#define SPMV(out, in) SPMV_(B, out, in)
  const double tol2 = 1e-10 * 1e-10;
  const int maxits = 2000;
  double rho = 1.0, alpha = 1.0, omega = 1.0, rr, rho_new;
  int it = 0, it0 = 0, converged = 0;
  const int use_xinv = C_.p.solver_precond == 1 && perm != NULL && n >= 64;
  if (use_xinv) {
    /* ---- BiCGstab on A_s, right-preconditioned with the frozen explicit inverse (csrc/solver.hip: launch_solver_xinv): the
     * multi-block summation order (dot_blocks), K iterations at most, then the Jacobi one-workgroup solver takes over */
    if (!xinv_M || xinv_n != n || xinv_key != (const void *)C_.m.ssh_values) {
      orc_solver_reset();
      const int ld = (n + 255) / 256 * 256;
      float *dense = malloc(sizeof(float) * (size_t)n * ld);
      int *rp0 = malloc(sizeof(int) * (n + 1)), *ci0 = malloc(sizeof(int) * C_.m.ssh_nza);
      for (int i = 0; i <= n; i++) rp0[i] = rp[i] - off;
      for (int j = 0; j < C_.m.ssh_nza; j++) ci0[j] = ci[j] - 1;
      if (orc_xinv_build(n, rp0, ci0, C_.m.ssh_values, ld, dense)) { fprintf(stderr, "orc: singular SSH operator\n"); abort(); }
      free(rp0); free(ci0);
      xinv_mp = malloc(sizeof(int) * (n + 1));
      orc_xinv_sparsify(n, ld, dense, XINV_DROP, xinv_mp, NULL, NULL);
      xinv_mc = malloc(sizeof(unsigned short) * (size_t)(xinv_mp[n] + 1)); xinv_M = malloc(sizeof(float) * (size_t)(xinv_mp[n] + 1));
      orc_xinv_sparsify(n, ld, dense, XINV_DROP, xinv_mp, xinv_mc, xinv_M);
      free(dense);
      xinv_n = n; xinv_key = (const void *)C_.m.ssh_values;
    }
    /* default schedule (csrc/solver.hip:launch_solver_xinv): two iterations for the first 300 solves after the set-up, one afterwards */
    const int K = C_.p.solver_xinv_its > 0 ? C_.p.solver_xinv_its : (xinv_solves < 300 ? 2 : 1);
    xinv_solves++;
    for (int i = 0; i < n; i++) xx[i] = x[i];
    SPMV_(As, r, xx);
    for (int i = 0; i < n; i++) { r[i] = b[i] - r[i]; r0[i] = r[i]; pv[i] = r[i]; }
    rr = dot_blocks(r, r, n);
    rho_new = rr;
    while (rr >= tol2 && it < maxits && it < K) {
      orc_xinv_apply(n, xinv_mp, xinv_mc, xinv_M, pv, ph);
      SPMV_(As, v, ph);
      alpha = rho_new / dot_blocks(r0, v, n);
      for (int i = 0; i < n; i++) s[i] = r[i] - alpha * v[i];
      orc_xinv_apply(n, xinv_mp, xinv_mc, xinv_M, s, sh);
      SPMV_(As, t, sh);
      double tt = dot_blocks(t, t, n), ts = dot_blocks(t, s, n), r0t = dot_blocks(r0, t, n), ss = dot_blocks(s, s, n);
      omega = (tt > 0.0) ? ts / tt : 0.0;
      rho = rho_new;
      rho_new = -omega * r0t;
      rr = ss - omega * (2.0 * ts - omega * tt);
      it++;
      const int more = (rr >= tol2 && it < maxits);
      const double beta = more ? (rho_new / rho) * (alpha / omega) : 0.0;
      for (int i = 0; i < n; i++) {
        r[i] = s[i] - omega * t[i];
        xx[i] = (xx[i] + alpha * ph[i]) + omega * sh[i];
        if (more) pv[i] = r[i] + beta * (pv[i] - omega * v[i]);
      }
    }
    converged = !(rr >= tol2 && it < maxits);
    for (int i = 0; i < n; i++) x[i] = xx[i];
    it0 = it;
  }
  int use_ras = C_.p.solver_precond == 1 && !use_xinv && maxnnz <= 16;
  if (use_ras && (ras_key != (const void *)C_.m.ssh_values || ras_n != n)) {
    orc_solver_reset();
    int *rp0 = malloc(sizeof(int) * (n + 1)), *ci0 = malloc(sizeof(int) * C_.m.ssh_nza);
    for (int i = 0; i <= n; i++) rp0[i] = rp[i] - off;
    for (int j = 0; j < C_.m.ssh_nza; j++) ci0[j] = ci[j] - 1;
    if (orc_ras_build(n, rp0, ci0, C_.m.ssh_values, ORC_RAS_PATCH_MAX, ORC_RAS_OVERLAP, ORC_RAS_DEG, ORC_RAS_KAPPA, &ras_plan)) use_ras = 0;      /* (does not qualify: Jacobi) */
    else { ras_key = (const void *)C_.m.ssh_values; ras_n = n; }
    free(rp0); free(ci0);
  }
  if (use_ras) {
    /* ---- BiCGstab on A_s, right-preconditioned with the frozen RAS-Chebyshev operator (csrc/solver_ras.hip: launch_solver_ras).  All
     * vectors in the patch order of the plan (perm[position] = row), block partial sums over positions (dot_blocks) */
    const int *perm_q = ras_plan.perm, *inv_q = ras_plan.inv;
    double *bq = malloc(sizeof(double) * n);
#define SPMVQ(out, in) for (int q = 0; q < n; q++) { const int i = perm_q[q]; double a = 0.0; for (int j = rp[i] - off; j < rp[i + 1] - off; j++) a = a + As[j] * (in)[inv_q[ci[j] - 1]]; (out)[q] = a; }
    for (int q = 0; q < n; q++) { xx[q] = x[perm_q[q]]; bq[q] = b[perm_q[q]]; }
    SPMVQ(r, xx);
    for (int q = 0; q < n; q++) { r[q] = bq[q] - r[q]; r0[q] = r[q]; pv[q] = r[q]; }
    rr = dot_blocks(r, r, n);
    rho_new = rr;
    while (rr >= tol2 && it < maxits) {
      orc_ras_apply(&ras_plan, pv, ph);
      SPMVQ(v, ph);
      alpha = rho_new / dot_blocks(r0, v, n);
      for (int q = 0; q < n; q++) s[q] = r[q] - alpha * v[q];
      orc_ras_apply(&ras_plan, s, sh);
      SPMVQ(t, sh);
      double tt = dot_blocks(t, t, n), ts = dot_blocks(t, s, n), r0t = dot_blocks(r0, t, n), ss = dot_blocks(s, s, n);
      omega = (tt > 0.0) ? ts / tt : 0.0;
      rho = rho_new;
      rho_new = -omega * r0t;
      rr = ss - omega * (2.0 * ts - omega * tt);
      it++;
      const int more = (rr >= tol2 && it < maxits);
      const double beta = more ? (rho_new / rho) * (alpha / omega) : 0.0;
      for (int q = 0; q < n; q++) {
        r[q] = s[q] - omega * t[q];
        xx[q] = (xx[q] + alpha * ph[q]) + omega * sh[q];
        if (more) pv[q] = r[q] + beta * (pv[q] - omega * v[q]);
      }
    }
#undef SPMVQ
    for (int q = 0; q < n; q++) x[perm_q[q]] = xx[q];
    free(bq);
    converged = 1;
  }
  if (!converged) {
    /* ---- Jacobi, applied as the column scaling B = A_s D^-1, y = D x: the one-workgroup / multi-workgroup HIP solve; after the
     * explicit-inverse iterations it continues from their iterate */
    solver_perm = perm;
    for (int i = 0; i < n; i++) y[i] = x[i] * diag[i];
    SPMV(r, y);
    for (int i = 0; i < n; i++) { r[i] = b[i] - r[i]; r0[i] = r[i]; pv[i] = 0.0; v[i] = 0.0; }
    rho = 1.0; alpha = 1.0; omega = 1.0;
    rr = dot_sel(r, r, n);
    rho_new = rr;
    it = 0;
    /* BiCGstab with two reduction points per iteration: rho and ||r||^2 come from recurrences
     * (r0.s = 0 by construction  =>  r0.r = -omega r0.t ;  r = s - omega t  =>  r.r = s.s - omega(2 t.s - omega t.t)) */
    while (rr >= tol2 && it < maxits) {
      double beta = (rho_new / rho) * (alpha / omega);
      for (int i = 0; i < n; i++) pv[i] = r[i] + beta * (pv[i] - omega * v[i]);
      SPMV(v, pv);
      alpha = rho_new / dot_sel(r0, v, n);
      for (int i = 0; i < n; i++) s[i] = r[i] - alpha * v[i];
      SPMV(t, s);
      double tt = dot_sel(t, t, n), ts = dot_sel(t, s, n), r0t = dot_sel(r0, t, n), ss = dot_sel(s, s, n);
      omega = (tt > 0.0) ? ts / tt : 0.0;
      for (int i = 0; i < n; i++) { y[i] = (y[i] + alpha * pv[i]) + omega * s[i]; r[i] = s[i] - omega * t[i]; }
      rho = rho_new;
      rho_new = -omega * r0t;
      rr = ss - omega * (2.0 * ts - omega * tt);
      it++;
    }
    for (int i = 0; i < n; i++) x[i] = y[i] * (1.0 / diag[i]);
    it += it0;
  }
  C_.solver_iters = it; C_.solver_resid = sqrt(rr > 0.0 ? rr : 0.0);
  free(B); free(As); free(diag); free(perm); solver_perm = NULL;
}

/* oce_timestep_ale sequence for the supported options: src/oce_ale.F90:2556-2767 (+ fvom_main.F90:216) */
void orc_step(int n) {
  orc_compute_vel_nodes();
  if (C_.p.toy_soufflet && n % 10 == 0) orc_compute_zonal_mean();      /* before_oce_step, oce_setup_step.F90:625-630 */
  orc_pressure_bv();
  orc_pressure_force();
  orc_sw_alpha_beta();
  orc_compute_sigma_xy();
  orc_compute_neutral_slope();
  if (C_.p.mix_scheme == 2) { orc_mixing_pp(); orc_mo_convect(); }
  if (C_.p.mix_scheme == 1) { orc_mixing_kpp(); orc_mo_convect(); }                  /* oce_ale.F90:2607-2611 */
  orc_compute_vel_rhs();
  orc_viscosity_filter();
  if (C_.p.i_vert_visc) orc_impl_vert_visc_ale();
  if (C_.p.which_ale != 0) orc_update_stiff_mat_ale();
  orc_compute_ssh_rhs_ale();
  orc_solve_ssh();
  if (C_.p.toy_soufflet) orc_relax_zonal_vel();                          /* oce_ale.F90:2696 */
  orc_update_vel();
  orc_compute_hbar_ale();
  orc_eta_update();
  if (C_.p.Fer_GM || C_.p.Redi) orc_init_Redi_GM();                                        /* oce_ale.F90:2729-2739 */
  if (C_.p.Fer_GM) { orc_fer_solve_Gamma(); orc_fer_gamma2vel(); }
  orc_vert_vel_ale();
  if (C_.p.SPP) orc_spp();                                                                  /* solve_tracers_ale :120-121 */
  if (C_.p.Fer_GM) { orc_fer_wvel(); orc_bolus_add(); }                                   /* oce_ale.F90:1720-1811, oce_ale_tracer.F90:127-131 */
  for (int tr = 1; tr <= C_.ntr; tr++) {
    orc_init_tracers_AB(tr);
    orc_adv_tracers_ale(tr);
    orc_diff_tracers_ale(tr);
    if (C_.p.toy_soufflet) orc_relax_zonal_temp();                       /* oce_ale_tracer.F90:150-151, once per tracer */
    else orc_relax_to_clim(tr);
  }
  if (C_.p.Fer_GM) orc_bolus_remove();                                  /* oce_ale_tracer.F90:165-169 */
  orc_salinity_clamp();
  orc_update_thickness_ale();
}

int orc_call(const char *name, int arg) {
#define CALL0(f) if (!strcmp(name, #f)) { orc_##f(); return 0; }
#define CALL1(f) if (!strcmp(name, #f)) { orc_##f(arg); return 0; }
  CALL0(init_ref_density) CALL0(compute_vel_nodes) CALL0(pressure_bv) CALL0(pressure_force) CALL0(sw_alpha_beta) CALL0(compute_sigma_xy)
  CALL0(compute_neutral_slope) CALL0(mixing_pp) CALL0(mixing_kpp) CALL0(mo_convect) CALL0(compute_vel_rhs) CALL0(visc_filt_bcksct) CALL0(viscosity_filter)
  CALL0(impl_vert_visc_ale) CALL0(update_stiff_mat_ale) CALL0(compute_ssh_rhs_ale) CALL0(solve_ssh) CALL0(update_vel)
  CALL0(compute_hbar_ale) CALL0(eta_update) CALL0(vert_vel_ale) CALL1(init_tracers_AB) CALL1(adv_tracers_ale)
  CALL1(diff_tracers_ale) CALL0(salinity_clamp) CALL0(update_thickness_ale) CALL1(step)
  CALL0(init_Redi_GM) CALL0(fer_solve_Gamma) CALL0(fer_gamma2vel) CALL0(fer_wvel) CALL0(bolus_add) CALL0(bolus_remove)
  CALL0(compute_zonal_mean_ini) CALL0(compute_zonal_mean) CALL0(relax_zonal_vel) CALL0(relax_zonal_temp) CALL1(relax_to_clim) CALL0(spp)
  fprintf(stderr, "orc_call: unknown routine %s\n", name);
  return 1;
}


/* write_step_info + check_blowup, src/write_step_info.F90:14-222, :225-447 (single partition: the owned nodes are all nodes).
 * out[42] in the field order of fesom_step_info (include/fesom_gpu.h). */
int orc_step_info(double *out) {
  const int myN = C_.m.myDim_nod2D;
  double s[6] = {0, 0, 0, 0, 0, 0}, mn[15], mx[20], blow = 0.0;
  for (int i = 0; i < 15; i++) mn[i] = 1.0e300;
  for (int i = 0; i < 20; i++) mx[i] = -1.0e300;
#define MN(i, v) do { double v_ = (v); if (v_ < mn[i]) mn[i] = v_; } while (0)
#define MX(i, v) do { double v_ = (v); if (v_ > mx[i]) mx[i] = v_; } while (0)
#define MM(i, v) do { MN(i, v); MX(i, v); } while (0)
  for (int n = 1; n <= myN; n++) {
    const double a = AREASVOL(ULEVN(n), n);
    s[0] = s[0] + a * C_.eta_n[n - 1]; s[1] = s[1] + a * C_.hbar[n - 1]; s[2] = s[2] + a * C_.d_eta[n - 1];
    s[3] = s[3] + a * (C_.hbar[n - 1] - C_.hbar_old[n - 1]); s[4] = s[4] + a * C_.water_flux[n - 1];
    s[5] = s[5] + AREA(ULEVN(n), n);
    MM(0, C_.eta_n[n - 1]); MM(1, C_.hbar[n - 1]); MM(2, C_.water_flux[n - 1]); MM(3, C_.heat_flux[n - 1]);
    for (int nz = 1; nz <= NLM1; nz++) if (TR(nz, n, 2) != 0.0) { MM(4, TR(nz, n, 1)); MM(5, TR(nz, n, 2)); }
    MM(6, A2L(C_.Wvel, 1, n)); MM(7, A2L(C_.Wvel, 2, n));
    MM(8, V2(C_.Unode, 1, 1, n)); MM(9, V2(C_.Unode, 1, 2, n)); MM(10, V2(C_.Unode, 2, 1, n)); MM(11, V2(C_.Unode, 2, 2, n));
    MM(12, C_.d_eta[n - 1]);
    if (A2(C_.hnode, 1, n) != 0.0) MM(13, A2(C_.hnode, 1, n));
    if (A2(C_.hnode, 2, n) != 0.0) MM(14, A2(C_.hnode, 2, n));
    for (int nz = 1; nz <= NL; nz++) { MX(15, A2L(C_.CFL_z, nz, n)); MX(19, fabs(A2L(C_.Kv, nz, n))); }
    if (n <= C_.E) {                                  /* element arrays scanned with the node count (reference quirk) */
      for (int nz = 1; nz <= NLM1; nz++) { MX(16, fabs(A2(C_.pgf_x, nz, n))); MX(17, fabs(A2(C_.pgf_y, nz, n))); }
      for (int nz = 1; nz <= NL; nz++) MX(18, fabs(A2L(C_.Av, nz, n)));
    }
    /* check_blowup */
    double e = C_.eta_n[n - 1], de = C_.d_eta[n - 1];
    if (e != e || e < -50.0 || e > 50.0 || de != de) blow = 1.0;
    if (C_.p.which_ale != 0) {
      double w = A2L(C_.Wvel, 1, n), h = A2(C_.hnode, 1, n);
      if (w != w) blow = 1.0;
      if (h != h || h < 0) blow = 1.0;
    }
    for (int nz = 1; nz <= NLEVN(n) - 1; nz++) {
      double t = TR(nz, n, 1), sa = TR(nz, n, 2);
      if (t != t || t < -5.0 || t > 60) blow = 1.0;
      if (sa != sa || sa < 0 || sa > 50) blow = 1.0;
    }
  }
  for (int i = 0; i < 6; i++) out[i] = s[i];
  for (int i = 0; i < 15; i++) out[6 + i] = mn[i];
  for (int i = 0; i < 20; i++) out[21 + i] = mx[i];
  out[41] = blow;
  return 0;
}
