// Gent-McWilliams bolus velocities after Ferrari et al. 2010 (src/oce_fer_gm.F90) and the places of the step where they
// enter: fer_Wvel of vert_vel_ale (src/oce_ale.F90:1720-1811) and the temporary addition of the bolus velocities around
// the tracer loop (solve_tracers_ale, src/oce_ale_tracer.F90:127-131,165-169).  gfx950; one wavefront = one column.
#include "dev.h"
#include <string.h>

#define DG3(a, c, nz, n) (a)[((size_t)(n) * m.nl + ((nz) - 1)) * 2 + ((c) - 1)]          // (2, nl, N)

// init_Redi_GM (:159-340), GM part.  The horizontal factor that only depends on the mesh (resolution scaling with a real
// exponent -> libm pow, resolution ramp) is prepared on the host at fesom_gpu_init (gm_scal_static).
__global__ void __launch_bounds__(BLOCK) k_gm_coef(DM m) {
  int n = col_id(m), l = lane_id(), nz = l + 1;
  if (n >= m.myN) return;
  const int nzmax1 = m.nlev_n_min[n], nzmin1 = m.ulev_n_max[n];
  const double c_min = 0.5, pi = 3.14159265358979;
  double bv = (nz <= m.nl) ? DA2L(m.bvfreq, nz, n) : 0.0;
  double bv_dn = shdn(bv);
  double term = 0.0;
  if (nz >= nzmin1 && nz <= nzmax1 - 1) term = DA2(m.hnode_new, nz, n) * (sqrt(fabs(dmax_(bv, 0.))) + sqrt(fabs(dmax_(bv_dn, 0.)))) / 2.;
  double run = seq_sum_up(term, nzmin1 - 1, nzmax1 - 2, 0.0);
  double c1 = (nzmax1 - 1 >= nzmin1) ? bcast(run, nzmax1 - 2) : 0.0;
  c1 = dmax_(c_min, c1 / pi);
  double scaling = m.gm_scal_static[n];
  if (m.p.scaling_Rossby) {      // :196-200: cut K_GM off where the mesh resolves the Rossby radius (Fermi function of resolution / radius; exp: device libm)
    const double f_min = 1.e-6, r_max = 200000., x0 = 1.5, sigma = .15;
    const double rosb = dmin_(c1 / dmax_(fabs(m.coriolis_node[n]), f_min), r_max);
    const double rr_ratio = dmin_(m.mesh_resolution[n] / rosb, 5.);
    scaling = 1. / (1. + exp(-(rr_ratio - x0) / sigma));
    scaling = scaling * m.gm_scal_A[n];
    scaling = scaling * m.gm_scal_B[n];
  }
  const double scal = dmin_(scaling, 1.0);
  double base = scal * m.p.K_GM_max;
  base = dmax_(base, m.p.K_GM_min);
  if (l == 0 && m.p.Fer_GM) m.fer_c[n] = c1 * c1;
  const int nzmax = m.nlev_n[n], nzmin = m.ulev_n[n];
  double zs = 1.0;
  if (m.p.scaling_Ferreira) {
    const int mi = m.MLD1_ind[n];
    double bvref;
    if (m.p.K_GM_bvref == 0) bvref = dmax_(bcast(bv, nzmin - 1), 1.e-6);
    else if (m.p.K_GM_bvref == 1) bvref = dmax_(bcast(bv, mi), 1.e-6);
    else {
      double s = seq_sum_up((nz >= nzmin && nz <= mi) ? bv : 0.0, nzmin - 1, mi - 1, 0.0);
      bvref = dmax_(bcast(s, mi - 1) / (double)mi, 1.e-6);
    }
    zs = dmax_(bv / bvref, 0.2);
    zs = dmin_(zs, 1.0);
  }
  if (m.p.scaling_FESOM14 && nz >= nzmin && nz <= nzmax) {
    int k = nz < m.nl - 1 ? nz : m.nl - 1;
    if (DV3(m.neutral_slope, 3, k, n) > 5.e-3) zs = 0.0;
  }
  double fb = base, kb = m.p.Fer_GM ? base : m.redi_k0[n];  // surface templates of fer_K and of Ki: K_hor*(reso/100km)^2, or the GM coefficient when both are on (:243-250)
  if (m.p.use_cavity) {
    // Under an ice shelf the reference's two loops disagree about the top level: the templates go to max(ulevels of the node's elements) (:181,232,243), the
    // vertical scaling starts from fer_k / Ki at ulevels_nod2D (:260,313-330).  Where the two differ (nodes at the rim of the draft) the scaling reads what the
    // previous step left there -- fer_K starts at 500 (oce_setup_step.F90:359), Ki at K_hor*(reso/100km)^2 (:330) -- and scales it again.  And "Redi equal GM",
    // Ki(nzmin,:)=fer_k(nzmin,:), stands after the first loop (:250): it acts on every node at the upper level of the LAST owned node (gm_nzl).
    const int nzl = m.gm_nzl;
    const double fk_old = m.p.Fer_GM ? DA2L(m.fer_K, min(nz, m.nl), n) : 0.0, ki_old = m.p.Redi ? DA2(m.Ki, min(nz, m.nlm1), n) : 0.0;
    if (nzmin < nzmin1) {
      fb = bcast(fk_old, nzmin - 1);
      kb = (m.p.Fer_GM && nzl == nzmin) ? fb : bcast(ki_old, nzmin - 1);
    } else if (m.p.Fer_GM && nzl != nzmin) kb = m.redi_k0[n];
    if (m.p.Fer_GM && m.p.Redi && nz == nzl && nz <= m.nlm1 && (nz < nzmin || nz > nzmax - 1)) DA2(m.Ki, nz, n) = fk_old;      // (a level outside the column: never read)
  }
  if (m.p.Fer_GM && nz >= nzmin && nz <= nzmax) DA2L(m.fer_K, nz, n) = fb * zs;
  if (m.p.Redi) {
    const double zs_dn = shdn(zs);
    if (nz >= nzmin && nz <= nzmax - 1) DA2(m.Ki, nz, n) = kb * 0.5 * (zs + zs_dn);
  }
}

// fer_solve_Gamma (:8-120): tridiagonal problem per node column with two right-hand sides; the sweep runs in the block.
// Shapes (dev.h:ThTile): <8, 8> one column per wave (pi: latency-bound); <TL_COLS, TL_WAVES> tiles with several columns per wave on CORE2-class
// meshes (DM::use_tile): ONE wavefront sweeps 32 / 64 columns with lane = column instead of 8, the dependent divide chain is amortised.
template <int COLS, int WAVES>
__global__ void __launch_bounds__(WAVE * WAVES) k_fer_gamma(DM m) {
  extern __shared__ double th_sh[];
  ThTile<2, COLS> tile(th_sh, m.nl);
  const int w = threadIdx.x >> 6, l = lane_id(), nz = l + 1;
  const int base = xcd_block() * COLS;
  for (int ci = w; ci < COLS; ci += WAVES) {
    int n = __builtin_amdgcn_readfirstlane(sub_col(m, base + ci));
    const bool valid = n < m.myN;
    if (!valid) n = m.myN - 1;
    int nzmax = m.nlev_n[n], nzmin = m.ulev_n[n];
    // zbar_n, Z_n of the column from hnode_new (bottom-up, reference order); lane nz-1 <-> level nz
    double hn = (nz >= nzmin && nz <= nzmax - 1) ? DA2(m.hnode_new, nz, n) : 0.0;
    double zb = seq_sum_down(hn, nzmax - 2, nzmin - 1, m.zbar_n_bot[n]);   // zbar_n(nz), nz = nzmin..nzmax-1
    if (nz == nzmax) zb = m.zbar_n_bot[n];
    double zb_dn = shdn(zb);                                               // zbar_n(nz+1)
    double Zn = zb_dn + hn / 2.0;                                          // Z_n(nz), nz <= nzmax-1
    double Zn_up = shup(Zn);
    nzmax = m.nlev_n_min[n]; nzmin = m.ulev_n_max[n];
    double a = 0.0, b = 1.0, c = 0.0, t1 = 0.0, t2 = 0.0;
    double zinv_own = 1.0 / (zb - zb_dn);                                  // 1/(zbar_n(nz)-zbar_n(nz+1))
    double zinv_up = shup(zinv_own);
    if (valid && nz >= nzmin + 1 && nz <= nzmax - 1) {
      double zinv = 1.0 / (Zn_up - Zn);
      const double fc = m.fer_c[n];
      a = fc * zinv_up * zinv;
      c = fc * zinv_own * zinv;
      b = -a - c - dmax_(DA2L(m.bvfreq, nz, n), 1.e-8);
      const double r = D_G / D_RHO0, fk = DA2L(m.fer_K, nz, n);
      t1 = r * 0.5 * (DV2(m.sigma_xy, 1, nz - 1, n) + DV2(m.sigma_xy, 1, nz, n)) * fk;
      t2 = r * 0.5 * (DV2(m.sigma_xy, 2, nz - 1, n) + DV2(m.sigma_xy, 2, nz, n)) * fk;
    }
    tile.put(ci, valid, nzmin, nzmax, a, b, c, t1, t2);
  }
  tile.sweep();
  for (int ci = w; ci < COLS; ci += WAVES) {
    const int n = __builtin_amdgcn_readfirstlane(sub_col(m, base + ci));
    if (n >= m.myN) continue;
    double g1, g2;
    tile.get(ci, g1, g2);
    if (nz >= m.ulev_n_max[n] && nz <= m.nlev_n_min[n]) { DG3(m.fer_gamma, 1, nz, n) = g1; DG3(m.fer_gamma, 2, nz, n) = g2; }
  }
}
#define FG_SHAPE(id, C_, W_) case id: hipLaunchKernelGGL((k_fer_gamma<C_, W_>), dim3((SUBN(m, m.myN) + C_ - 1) / C_), dim3(WAVE * W_), (ThTile<2, C_>::lds_bytes(m.nl)), s, m); break;
#define FG_ATTR(id, C_, W_) (void)hipFuncSetAttribute((const void *)k_fer_gamma<C_, W_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
static void launch_fer_gamma(const DM &m, hipStream_t s) {
  static bool attr = false;
  if (!attr) { attr = true; TILE_SHAPES(FG_ATTR) FG_ATTR(5, 16, 8) FG_ATTR(6, 16, 4) }
  static const int env = getenv("FESOM_GPU_EXP_FG_SHAPE") ? atoi(getenv("FESOM_GPU_EXP_FG_SHAPE")) : 0;
  // (default tile shape of THIS kernel: 16 columns x 8 waves -- 48 VGPRs, so the 31 KB image instead of 62 KB lets 5 workgroups share a CU: 357 -> 297 us on the basin)
  switch (env > 0 ? env : (m.use_tile == 1 ? 5 : m.use_tile)) {
    TILE_SHAPES(FG_SHAPE) FG_SHAPE(5, 16, 8) FG_SHAPE(6, 16, 4)
    default: hipLaunchKernelGGL((k_fer_gamma<TH_COLS, TH_COLS>), dim3(nblocks_th(SUBN(m, m.myN))), dim3(TH_BLOCK), (ThTile<2, TH_COLS>::lds_bytes(m.nl)), s, m); break;
  }
}

// fer_gamma2vel (:125-154)
__global__ void __launch_bounds__(BLOCK) k_fer_uv(DM m) {
  int el = col_id(m), nz = lane_id() + 1;
  if (el >= m.myE) return;
  if (nz < m.ulev[el] || nz > m.nlev[el] - 1) return;
  const double onethird = 1. / 3.;
  int n1 = m.elem_nodes[3 * el], n2 = m.elem_nodes[3 * el + 1], n3 = m.elem_nodes[3 * el + 2];
  double zinv = onethird / DA2(m.helem, nz, el);
#pragma unroll
  for (int k = 1; k <= 2; k++)
    DV2(m.fer_UV, k, nz, el) = (((DG3(m.fer_gamma, k, nz, n1) - DG3(m.fer_gamma, k, nz + 1, n1)) + (DG3(m.fer_gamma, k, nz, n2) - DG3(m.fer_gamma, k, nz + 1, n2))) +
                                (DG3(m.fer_gamma, k, nz, n3) - DG3(m.fer_gamma, k, nz + 1, n3))) * zinv;
}

// fer_Wvel of vert_vel_ale: divergence of the bolus transports gathered over the node's edges in edge order, summed
// bottom-up, divided by the area
__global__ void __launch_bounds__(BLOCK) k_fer_wvel(DM m) {
  int n = col_id(m), l = lane_id(), nz = l + 1;
  if (n >= m.myN) return;
  const int nzmin = m.ulev_n[n], nzmax = m.nlev_n[n] - 1;
  double w = 0.0;
  if (nz <= m.nlm1 && (m.exp_batch & 16)) {
    // the bolus velocities and thicknesses of both triangles of three edges at a time in one batch of independent loads (a triangle that does not reach
    // this level, or is not there, is read inside a column all the same and dropped in the select); the sum keeps the edge order
    constexpr int WB = 3;
    const int q0 = m.ne_ptr[n], q1 = m.ne_ptr[n + 1];
    for (int qb = q0; qb < q1; qb += WB) {
      double c1[WB], c2[WB]; bool on1[WB], on2[WB]; int sgn[WB];
#pragma unroll
      for (int j = 0; j < WB; j++) {
        const int q = qb + j < q1 ? qb + j : q0;
        const int ed = m.ne_idx[q], e1 = m.edge_tri[2 * ed], e2 = m.edge_tri[2 * ed + 1], e2c = e2 >= 0 ? e2 : e1;
        sgn[j] = m.ne_sgn[q];
        on1[j] = qb + j < q1 && nz >= m.ulev[e1] && nz <= m.nlev[e1] - 1;
        on2[j] = qb + j < q1 && e2 >= 0 && nz >= m.ulev[e2c] && nz <= m.nlev[e2c] - 1;
        c1[j] = (DV2(m.fer_UV, 2, nz, e1) * DECD(1, ed) - DV2(m.fer_UV, 1, nz, e1) * DECD(2, ed)) * DA2(m.helem, nz, e1);
        c2[j] = -(DV2(m.fer_UV, 2, nz, e2c) * DECD(3, ed) - DV2(m.fer_UV, 1, nz, e2c) * DECD(4, ed)) * DA2(m.helem, nz, e2c);
      }
#pragma unroll
      for (int j = 0; j < WB; j++) {
        const double a1 = (sgn[j] > 0) ? w + c1[j] : w - c1[j];
        w = on1[j] ? a1 : w;
        const double a2 = (sgn[j] > 0) ? w + c2[j] : w - c2[j];
        w = on2[j] ? a2 : w;
      }
    }
  } else if (nz <= m.nlm1) {
    for (int q = m.ne_ptr[n]; q < m.ne_ptr[n + 1]; q++) {
      int ed = m.ne_idx[q], sg = m.ne_sgn[q];
      int e1 = m.edge_tri[2 * ed], e2 = m.edge_tri[2 * ed + 1];
      if (nz >= m.ulev[e1] && nz <= m.nlev[e1] - 1) {
        double c1 = (DV2(m.fer_UV, 2, nz, e1) * DECD(1, ed) - DV2(m.fer_UV, 1, nz, e1) * DECD(2, ed)) * DA2(m.helem, nz, e1);
        w = (sg > 0) ? w + c1 : w - c1;
      }
      if (e2 >= 0 && nz >= m.ulev[e2] && nz <= m.nlev[e2] - 1) {
        double c2 = -(DV2(m.fer_UV, 2, nz, e2) * DECD(3, ed) - DV2(m.fer_UV, 1, nz, e2) * DECD(4, ed)) * DA2(m.helem, nz, e2);
        w = (sg > 0) ? w + c2 : w - c2;
      }
    }
  }
  double wc = seq_sum_down((nz >= nzmin && nz <= nzmax) ? w : 0.0, nzmax - 1, nzmin - 1, 0.0);
  if (nz <= m.nl) DA2L(m.fer_Wvel, nz, n) = (nz >= nzmin && nz <= nzmax) ? wc / DA2L(m.area, nz, n) : 0.0;
}

// UV, Wvel_e, Wvel +/- bolus velocities (whole arrays, as the reference's array statements)
__global__ void k_bolus(DM m, double sign) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nuv = (size_t)2 * m.nlm1 * m.E, nw = (size_t)m.nl * m.N;
  if (i < nuv) m.UV[i] = (sign > 0) ? m.UV[i] + m.fer_UV[i] : m.UV[i] - m.fer_UV[i];
  if (i < nw) {
    double f = m.fer_Wvel[i];
    m.Wvel_e[i] = (sign > 0) ? m.Wvel_e[i] + f : m.Wvel_e[i] - f;
    m.Wvel[i] = (sign > 0) ? m.Wvel[i] + f : m.Wvel[i] - f;
  }
}

#define LAUNCH_COL(k, ncol, ...) hipLaunchKernelGGL(k, dim3(nblocks(ncol)), dim3(BLOCK), 0, s, __VA_ARGS__)
int launch_named_gm(const DM &m, hipStream_t s, const char *name) {
  if (!m.p.Fer_GM && !m.p.Redi) return -1;
  if (!m.p.Fer_GM && strcmp(name, "init_Redi_GM") && strcmp(name, "k_gm_coef")) return -1;
  if (!strcmp(name, "init_Redi_GM") || !strcmp(name, "k_gm_coef")) { LAUNCH_COL(k_gm_coef, m.myN, m); return 0; }
  if (!strcmp(name, "fer_solve_Gamma") || !strcmp(name, "k_fer_gamma")) {
    launch_fer_gamma(m, s); return 0;
  }
  if (!strcmp(name, "fer_gamma2vel") || !strcmp(name, "k_fer_uv")) { LAUNCH_COL(k_fer_uv, m.myE, m); return 0; }
  if (!strcmp(name, "fer_wvel") || !strcmp(name, "k_fer_wvel")) { LAUNCH_COL(k_fer_wvel, m.myN, m); return 0; }
  if (!strcmp(name, "bolus_add") || !strcmp(name, "bolus_remove")) {
    size_t nmax = std::max((size_t)2 * m.nlm1 * m.E, (size_t)m.nl * m.N);
    hipLaunchKernelGGL(k_bolus, dim3((unsigned)((nmax + 255) / 256)), dim3(256), 0, s, m, !strcmp(name, "bolus_add") ? 1.0 : -1.0);
    return 0;
  }
  return -1;
}
