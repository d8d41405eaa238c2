// Tracer kernels of the ocean step (gfx950): AB2 + gradients, FCT advection (UPW1 low order, MFCT/QR4C
// high order, Zalesak limiter), flux -> tendency, horizontal + implicit vertical diffusion, T* update.
// One wavefront = one column (lane = level); edge->node scatter-adds are node gathers in reference order.
#include "dev.h"
#include <string.h>

// Per-tracer scratch: every tracer has its own slab of the FCT / gradient / Thomas work arrays, so the T and S chains
// are independent and can run concurrently on two streams (they only share read-only velocities and thicknesses).
struct TV {
  double *del_ttf, *fct_LO, *fct_ttf_max, *fct_ttf_min, *fct_plus, *fct_minus, *tr_z, *adv_flux_ver, *tr_xy_ab, *tr_xy,
         *adv_flux_hor, *adv_flux_raw, *flux_lo_hor, *diff_flux, *edge_up_dn_grad;
};
__device__ __forceinline__ TV tracer_view(const DM &m, int tr) {
  size_t n1N = (size_t)m.nlm1 * m.N, nlN = (size_t)m.nl * m.N, n1E = (size_t)m.nlm1 * m.EX, n1D = (size_t)m.nlm1 * m.D;
  TV t;
  t.del_ttf = m.del_ttf + tr * n1N; t.fct_LO = m.fct_LO + tr * n1N; t.fct_ttf_max = m.fct_ttf_max + tr * n1N;
  t.fct_ttf_min = m.fct_ttf_min + tr * n1N; t.fct_plus = m.fct_plus + tr * n1N; t.fct_minus = m.fct_minus + tr * n1N;
  t.tr_z = m.tr_z + tr * nlN; t.adv_flux_ver = m.adv_flux_ver + tr * nlN;
  t.tr_xy_ab = m.tr_xy_ab + tr * 2 * n1E; t.tr_xy = m.tr_xy + tr * 2 * n1E;
  t.adv_flux_hor = m.adv_flux_hor + tr * n1D; t.adv_flux_raw = m.adv_flux_raw + tr * n1D; t.flux_lo_hor = m.flux_lo_hor + tr * n1D; t.diff_flux = m.diff_flux + tr * n1D; t.edge_up_dn_grad = m.edge_up_dn_grad + tr * 4 * n1D;
  return t;
}

// init_tracers_AB head (src/oce_tracer_mod.F90:49-83): AB2 extrapolation (k_tr_ab) and tracer_gradient_z (:124-153, k_tr_z).
// Split in two kernels because the AB part only needs the tracers (it is hoisted to the start of the step and overlaps
// the SSH solve) while tr_z needs hnode_new of this step's vert_vel_ale.
__global__ void __launch_bounds__(BLOCK) k_tr_ab(DM m, int tr0) {
  const int tr = tr0 + blockIdx.y;                        // grid.y = tracers of this launch
  int n = col_id(m), nz = lane_id() + 1;
  if (n >= m.N || nz > m.nlm1) return;
  const double eps = m.p.epsilon;
  double cur = DTR(m.tr_arr, nz, n, tr);
  DTR(m.tr_arr_old, nz, n, tr) = -(0.5 + eps) * DTR(m.tr_arr_old, nz, n, tr) + (1.5 + eps) * cur;
}
__global__ void __launch_bounds__(BLOCK) k_tr_z(DM m, int tr0) {
  const int tr = tr0 + blockIdx.y;                        // grid.y = tracers of this launch
  const TV t = tracer_view(m, tr);
  int n = col_id(m), nz = lane_id() + 1;
  if (n >= m.N) return;
  int nzmax = m.nlev_n[n], nzmin = m.ulev_n[n];
  if (nz >= nzmin + 1 && nz <= nzmax - 1) {
    double dz = 0.5 * (DA2(m.hnode_new, nz - 1, n) + DA2(m.hnode_new, nz, n));
    DA2L(t.tr_z, nz, n) = (DTR(m.tr_arr, nz - 1, n, tr) - DTR(m.tr_arr, nz, n, tr)) / dz;
  }
  if (nz == nzmin || nz == nzmax) DA2L(t.tr_z, nz, n) = 0.0;
}

// tracer_gradient_elements (src/oce_tracer_mod.F90:19-45) for the AB field and the current field in one pass
__global__ void __launch_bounds__(BLOCK) k_tr_grad_elem(DM m, int tr0) {
  const int tr = tr0 + blockIdx.y;                        // grid.y = tracers of this launch
  const TV t = tracer_view(m, tr);
  int e = col_id(m), nz = lane_id() + 1;
  if (e >= m.myE) return;
  if (nz < m.ulev[e] || nz > m.nlev[e] - 1) return;
  int n1 = m.elem_nodes[3 * e], n2 = m.elem_nodes[3 * e + 1], n3 = m.elem_nodes[3 * e + 2];
  double a1 = DTR(m.tr_arr_old, nz, n1, tr), a2 = DTR(m.tr_arr_old, nz, n2, tr), a3 = DTR(m.tr_arr_old, nz, n3, tr);
  double c1 = DTR(m.tr_arr, nz, n1, tr), c2 = DTR(m.tr_arr, nz, n2, tr), c3 = DTR(m.tr_arr, nz, n3, tr);
  DV2(t.tr_xy_ab, 1, nz, e) = DGS(1, e) * a1 + DGS(2, e) * a2 + DGS(3, e) * a3;
  DV2(t.tr_xy_ab, 2, nz, e) = DGS(4, e) * a1 + DGS(5, e) * a2 + DGS(6, e) * a3;
  DV2(t.tr_xy, 1, nz, e) = DGS(1, e) * c1 + DGS(2, e) * c2 + DGS(3, e) * c3;
  DV2(t.tr_xy, 2, nz, e) = DGS(4, e) * c1 + DGS(5, e) * c2 + DGS(6, e) * c3;
}
// The same with EPW elements and NT tracers per wave.  The index chain (element -> its 3 nodes, level range) is fetched lane-parallel
// ONCE for the wave's elements, then all 6 * NT * EPW column loads are issued before the first use: the kernel is bound by the number of loads in
// flight, not by bandwidth (a column is only 376 B).  Same expressions per cell as k_tr_grad_elem.
template <int EPW, int NT>
__global__ void __launch_bounds__(BLOCK) k_tr_grad_elem_b(DM m, int tr0) {
  const int slot0 = (xcd_block() * COLS_PER_BLOCK + (threadIdx.x >> 6)) * EPW;
  const int l = lane_id(), nz = l + 1, trb = tr0 + blockIdx.y * NT;
  int e_l = 0x7fffffff, lo_l = 1, hi_l = 0, nd_l = 0;
  if (l < EPW) {
    e_l = sub_col(m, slot0 + l);
    if (e_l < m.myE) { lo_l = m.ulev[e_l]; hi_l = m.nlev[e_l] - 1; }
  }
  {
    const int eq = __shfl(e_l, l / 3, 64);
    if (l < 3 * EPW && eq < m.myE) nd_l = m.elem_nodes[3 * eq + l % 3];
  }
  double a[EPW][NT][3], c[EPW][NT][3];
  const int nzc = nz <= m.nlm1 ? nz : m.nlm1;
#pragma unroll
  for (int i = 0; i < EPW; i++) {
    const int n1 = rdlane(nd_l, 3 * i), n2 = rdlane(nd_l, 3 * i + 1), n3 = rdlane(nd_l, 3 * i + 2);
#pragma unroll
    for (int t = 0; t < NT; t++) {
      const int tr = trb + t < m.ntr ? trb + t : m.ntr - 1;
      a[i][t][0] = DTR(m.tr_arr_old, nzc, n1, tr); a[i][t][1] = DTR(m.tr_arr_old, nzc, n2, tr); a[i][t][2] = DTR(m.tr_arr_old, nzc, n3, tr);
      c[i][t][0] = DTR(m.tr_arr, nzc, n1, tr); c[i][t][1] = DTR(m.tr_arr, nzc, n2, tr); c[i][t][2] = DTR(m.tr_arr, nzc, n3, tr);
    }
  }
#pragma unroll
  for (int i = 0; i < EPW; i++) {
    const int e = rdlane(e_l, i);
    if (e >= m.myE) continue;
    if (nz < rdlane(lo_l, i) || nz > rdlane(hi_l, i)) continue;
    const double g1 = DGS(1, e), g2 = DGS(2, e), g3 = DGS(3, e), g4 = DGS(4, e), g5 = DGS(5, e), g6 = DGS(6, e);
#pragma unroll
    for (int t = 0; t < NT; t++) {
      if (trb + t >= m.ntr) continue;
      const TV tv = tracer_view(m, trb + t);
      DV2(tv.tr_xy_ab, 1, nz, e) = g1 * a[i][t][0] + g2 * a[i][t][1] + g3 * a[i][t][2];
      DV2(tv.tr_xy_ab, 2, nz, e) = g4 * a[i][t][0] + g5 * a[i][t][1] + g6 * a[i][t][2];
      DV2(tv.tr_xy, 1, nz, e) = g1 * c[i][t][0] + g2 * c[i][t][1] + g3 * c[i][t][2];
      DV2(tv.tr_xy, 2, nz, e) = g4 * c[i][t][0] + g5 * c[i][t][1] + g6 * c[i][t][2];
    }
  }
}
static void launch_tr_grad_elem(const DM &m, hipStream_t s, int tr) {
  static const int env = getenv("FESOM_GPU_EXP_GRAD") ? atoi(getenv("FESOM_GPU_EXP_GRAD")) : -1;
  const int epw = env >= 0 ? env : 1;       // one element, both tracers per wave: 590 -> 374 us on the channel, 8.7 -> 6.4 us on pi; more elements per wave do not gain
  const int ne = SUBN(m, m.myE);
  if (epw == 0 || tr >= 0) { hipLaunchKernelGGL(k_tr_grad_elem, dim3(nblocks(ne), tr < 0 ? m.ntr : 1), dim3(BLOCK), 0, s, m, tr < 0 ? 0 : tr); return; }
  const int gy = (m.ntr + 1) / 2;
  switch (epw) {
    case 1: hipLaunchKernelGGL((k_tr_grad_elem_b<1, 2>), dim3(nblocks(ne), gy), dim3(BLOCK), 0, s, m, 0); break;
    case 2: hipLaunchKernelGGL((k_tr_grad_elem_b<2, 2>), dim3(nblocks((ne + 1) / 2), gy), dim3(BLOCK), 0, s, m, 0); break;
    case 3: hipLaunchKernelGGL((k_tr_grad_elem_b<3, 2>), dim3(nblocks((ne + 2) / 3), gy), dim3(BLOCK), 0, s, m, 0); break;
    default: hipLaunchKernelGGL((k_tr_grad_elem_b<4, 2>), dim3(nblocks((ne + 3) / 4), gy), dim3(BLOCK), 0, s, m, 0); break;
  }
}

// fill_up_dn_grad (src/oce_muscl_adv.F90:285-447)
__device__ __forceinline__ void cluster_grad(const DM &m, const TV &t, int node, int nz, double &gx, double &gy) {
  double tvol = 0.0, tx = 0.0, ty = 0.0;
  int num = m.nie_num[node];
  for (int k = 0; k < num; k++) {
    int e = m.nie[(size_t)m.maxk * node + k];
    if (m.nlev[e] - 1 < nz || nz < m.ulev[e]) continue;
    double ar = m.elem_area[e];
    tvol = tvol + ar;
    tx = tx + DV2(t.tr_xy_ab, 1, nz, e) * ar;
    ty = ty + DV2(t.tr_xy_ab, 2, nz, e) * ar;
  }
  gx = tx / tvol; gy = ty / tvol;
}
// the four up/down-wind gradient values of edge `ed` at level nz (w1: values (1,3) are defined here, w2: (2,4))
struct UpdnIdx { int n1, n2, t1, t2; bool both, c1, c2; };      // what does not depend on the tracer: the two upwind triangles, per level: both values from them / cluster mean at node 1 / 2
__device__ __forceinline__ UpdnIdx updn_index(const DM &m, int ed, int nz) {
  UpdnIdx x;
  x.n1 = m.edges[2 * ed]; x.n2 = m.edges[2 * ed + 1];
  x.t1 = m.updn[2 * ed]; x.t2 = m.updn[2 * ed + 1];
  int ul1 = m.ulev_n[x.n1], ul2 = m.ulev_n[x.n2], nl1 = m.nlev_n[x.n1] - 1, nl2 = m.nlev_n[x.n2] - 1;
  x.both = x.c1 = x.c2 = false;
  if (x.t1 >= 0 && x.t2 >= 0) {
    int nzmin = max(m.ulev_n_max[x.n1], m.ulev_n_max[x.n2]), nzmax = min(m.nlev_n_min[x.n1], m.nlev_n_min[x.n2]);
    if (nz >= nzmin && nz <= nzmax - 1) x.both = true;
    else {
      x.c1 = (nz >= ul1 && nz <= nzmin - 1) || (nz >= nzmax && nz <= nl1);
      x.c2 = (nz >= ul2 && nz <= nzmin - 1) || (nz >= nzmax && nz <= nl2);
    }
  } else {
    x.c1 = nz >= ul1 && nz <= nl1;
    x.c2 = nz >= ul2 && nz <= nl2;
  }
  return x;
}
// PRE: the cluster means come from the node field of k_cluster_grad (CORE2-class meshes with a ragged bottom: every edge of a node would evaluate the
// same mean again, 6 elements x 2 components per node and tracer) instead of being formed on the fly; the same values either way
template <bool PRE = false>
__device__ __forceinline__ void updn_values(const DM &m, const TV &t, int tr, const UpdnIdx &x, int nz, double &g1, double &g2, double &g3, double &g4, bool &w1, bool &w2) {
  w1 = false; w2 = false;      // write (1,3) / (2,4)
  g1 = g2 = g3 = g4 = 0.0;
  if (x.both) {
    g1 = DV2(t.tr_xy_ab, 1, nz, x.t1); g2 = DV2(t.tr_xy_ab, 1, nz, x.t2);
    g3 = DV2(t.tr_xy_ab, 2, nz, x.t1); g4 = DV2(t.tr_xy_ab, 2, nz, x.t2);
    w1 = w2 = true;
  } else if (PRE) {
    const double *cg = m.cl_grad + (size_t)tr * 2 * m.nlm1 * m.N;
    if (x.c1) { g1 = DV2(cg, 1, nz, x.n1); g3 = DV2(cg, 2, nz, x.n1); w1 = true; }
    if (x.c2) { g2 = DV2(cg, 1, nz, x.n2); g4 = DV2(cg, 2, nz, x.n2); w2 = true; }
  } else {
    if (x.c1) { cluster_grad(m, t, x.n1, nz, g1, g3); w1 = true; }
    if (x.c2) { cluster_grad(m, t, x.n2, nz, g2, g4); w2 = true; }
  }
}
// fill_up_dn_grad's cluster mean of the horizontal tracer gradient (src/oce_muscl_adv.F90:285-447: area-weighted mean over the elements of the node that
// reach the level) as a node field, for the nodes some edge needs it for (DM::cl_need: where the two upwind triangles do not cover the node's column).
// Lane-parallel element list, one batch of loads, the sums in element order: the same additions as cluster_grad().
__global__ void __launch_bounds__(BLOCK) k_cluster_grad(DM m, int tr0) {
  const int tr = tr0 + blockIdx.y;
  const TV t = tracer_view(m, tr);
  const int n = col_id(m), l = lane_id(), nz = l + 1;
  if (n >= m.N) return;
  if (!m.cl_need[n]) return;
  const int num = m.nie_num[n], nzc = nz <= m.nlm1 ? nz : m.nlm1;
  int el_l = 0, rg_l = 1;                          // level range of an element packed lo | hi << 8; (1, 0) = empty
  double ar_l = 0.0;
  if (l < num) { el_l = m.nie[(size_t)m.maxk * n + l]; rg_l = m.ulev[el_l] | ((m.nlev[el_l] - 1) << 8); ar_l = m.elem_area[el_l]; }
  double tvol = 0.0, tx = 0.0, ty = 0.0;
  constexpr int CB = 6;
  for (int q0 = 0; q0 < num; q0 += CB) {
    double gx[CB], gy[CB];
#pragma unroll
    for (int j = 0; j < CB; j++) {
      const int el = rdlane(el_l, q0 + j < num ? q0 + j : 0);
      gx[j] = DV2(t.tr_xy_ab, 1, nzc, el); gy[j] = DV2(t.tr_xy_ab, 2, nzc, el);
    }
#pragma unroll
    for (int j = 0; j < CB; j++) {
      if (q0 + j < num) {
        const int rg = rdlane(rg_l, q0 + j);
        const double ar = bcast(ar_l, q0 + j);
        const bool on = nz >= (rg & 0xff) && nz <= (rg >> 8);
        const double nv = tvol + ar, nx = tx + gx[j] * ar, ny = ty + gy[j] * ar;
        tvol = on ? nv : tvol; tx = on ? nx : tx; ty = on ? ny : ty;
      }
    }
  }
  if (nz < m.ulev_n[n] || nz > m.nlev_n[n] - 1) return;
  double *cg = m.cl_grad + (size_t)tr * 2 * m.nlm1 * m.N;
  DV2(cg, 1, nz, n) = tx / tvol; DV2(cg, 2, nz, n) = ty / tvol;
}
__device__ __forceinline__ void updn_grad(const DM &m, const TV &t, int ed, int nz, double &g1, double &g2, double &g3, double &g4, bool &w1, bool &w2) {
  const UpdnIdx x = updn_index(m, ed, nz);
  updn_values<false>(m, t, 0, x, nz, g1, g2, g3, g4, w1, w2);
}
__global__ void __launch_bounds__(BLOCK) k_updn_grad(DM m, int tr0) {
  const int tr = tr0 + blockIdx.y;                        // grid.y = tracers of this launch
  const TV t = tracer_view(m, tr);
  int ed = col_id(m), nz = lane_id() + 1;
  if (ed >= m.myD) return;
  if (nz > m.nlm1) return;
  double g1, g2, g3, g4;
  bool w1, w2;
  updn_grad(m, t, ed, nz, g1, g2, g3, g4, w1, w2);
  if (w1) { DV4(t.edge_up_dn_grad, 1, nz, ed) = g1; DV4(t.edge_up_dn_grad, 3, nz, ed) = g3; }
  if (w2) { DV4(t.edge_up_dn_grad, 2, nz, ed) = g2; DV4(t.edge_up_dn_grad, 4, nz, ed) = g4; }
}

// adv_tra_hor_upw1 (src/oce_adv_tra_hor.F90:57-211) + adv_tra_hor_mfct (:485-733) in one edge pass:
// flux_lo_hor = low-order flux, adv_flux_raw = high-order minus low-order (the init_zero=.false. convention), not yet
// limited; k_fct_edge_limit turns it into adv_flux_hor.
// FUSED (CORE2-class meshes, running step): fill_up_dn_grad is evaluated on the fly instead of being written to / read from
// edge_up_dn_grad (4 values per edge cell and tracer, the largest array of the step); entries the reference leaves untouched
// (not defined at this level) are still taken from the array, as k_flux_hor<false> would.
template <bool FUSED>
__global__ void __launch_bounds__(BLOCK) k_flux_hor(DM m, int tr0) {
  const int tr = tr0 + blockIdx.y;                        // grid.y = tracers of this launch
  const TV t = tracer_view(m, tr);
  int ed = col_id(m), nz = lane_id() + 1;
  if (ed >= m.myD) return;
  if (nz > m.nlm1) return;
  int n1 = m.edges[2 * ed], n2 = m.edges[2 * ed + 1], e1 = m.edge_tri[2 * ed], e2 = m.edge_tri[2 * ed + 1];
  int nl1 = m.nlev[e1] - 1, nu1 = m.ulev[e1], nl2 = 0, nu2 = 0;
  double dX1 = DECD(1, ed), dY1 = DECD(2, ed), dX2 = 0, dY2 = 0;
  double a = D_REARTH * m.elem_cos[e1];
  if (e2 >= 0) {
    dX2 = DECD(3, ed); dY2 = DECD(4, ed);
    nl2 = m.nlev[e2] - 1; nu2 = m.ulev[e2];
    a = 0.5 * (a + D_REARTH * m.elem_cos[e2]);
  }
  int nl12 = min(nl1, nl2), nu12 = max(nu1, nu2);
  bool use1, use2;
  if (nz >= nu12 && nz <= nl12) { use1 = true; use2 = true; }
  else if ((nz >= nu1 && nz <= nu12 - 1) || (nz >= nl12 + 1 && nz <= nl1)) { use1 = true; use2 = false; }
  else if (nu2 > 0 && ((nz >= nu2 && nz <= nu12 - 1) || (nz >= nl12 + 1 && nz <= nl2))) { use1 = false; use2 = true; }
  else { DA2(t.flux_lo_hor, nz, ed) = 0.0; DA2(t.adv_flux_raw, nz, ed) = 0.0; return; }
  double vflux;
  if (use1 && use2)
    vflux = (-DV2(m.UV, 2, nz, e1) * dX1 + DV2(m.UV, 1, nz, e1) * dY1) * DA2(m.helem, nz, e1) +
            (DV2(m.UV, 2, nz, e2) * dX2 - DV2(m.UV, 1, nz, e2) * dY2) * DA2(m.helem, nz, e2);
  else if (use1) vflux = (-DV2(m.UV, 2, nz, e1) * dX1 + DV2(m.UV, 1, nz, e1) * dY1) * DA2(m.helem, nz, e1);
  else vflux = (DV2(m.UV, 2, nz, e2) * dX2 - DV2(m.UV, 1, nz, e2) * dY2) * DA2(m.helem, nz, e2);
  double av = fabs(vflux);
  double t1 = DTR(m.tr_arr, nz, n1, tr), t2 = DTR(m.tr_arr, nz, n2, tr);
  // tra_adv_lim = 'NON' (oce_adv_tra_driver.F90:137-153): no low-order flux, the high-order flux is formed with init_zero=.true. (flux - 0.0)
  double lo = m.p.tra_adv_lim ? 0.0 : -0.5 * (t1 * (vflux + av) + t2 * (vflux - av)) - 0.0;
  DA2(t.flux_lo_hor, nz, ed) = lo;
  double s1 = DTR(m.tr_arr_old, nz, n1, tr), s2 = DTR(m.tr_arr_old, nz, n2, tr);
  double num_ord = m.p.tra_adv_ph;
  double ex = m.edxy[2 * ed], ey = m.edxy[2 * ed + 1];
  const int hor = m.p.tra_adv_hor;                          // 0 MFCT, 1 MUSCL (:215-481), 2 UPW1 (:57-211) as the high-order scheme
  if (hor == 2) {
    DA2(t.adv_flux_raw, nz, ed) = -0.5 * (s1 * (vflux + av) + s2 * (vflux - av)) - lo;
    return;
  }
  double Tmean2 = s2, Tmean1 = s1;
  double g1, g2, g3, g4;
  if (FUSED) {
    bool w1, w2;
    if (m.cl_grad) { const UpdnIdx ux = updn_index(m, ed, nz); updn_values<true>(m, t, tr, ux, nz, g1, g2, g3, g4, w1, w2); }
    else updn_grad(m, t, ed, nz, g1, g2, g3, g4, w1, w2);
    if (!w1) { g1 = DV4(t.edge_up_dn_grad, 1, nz, ed); g3 = DV4(t.edge_up_dn_grad, 3, nz, ed); }
    if (!w2) { g2 = DV4(t.edge_up_dn_grad, 2, nz, ed); g4 = DV4(t.edge_up_dn_grad, 4, nz, ed); }
  } else {
    g1 = DV4(t.edge_up_dn_grad, 1, nz, ed); g2 = DV4(t.edge_up_dn_grad, 2, nz, ed);
    g3 = DV4(t.edge_up_dn_grad, 3, nz, ed); g4 = DV4(t.edge_up_dn_grad, 4, nz, ed);
  }
  if (hor == 1) {   // MUSCL: the gradient correction is switched off below nboundary_lay of the node (c_lo = 0 or 1)
    const double c1 = (m.nb_lay[n1] - nz >= 0) ? 1.0 : 0.0, c2 = (m.nb_lay[n2] - nz >= 0) ? 1.0 : 0.0;
    Tmean2 = s2 - (2.0 * (s2 - s1) + ex * a * g2 + ey * D_REARTH * g4) / 6.0 * c2;
    Tmean1 = s1 + (2.0 * (s2 - s1) + ex * a * g1 + ey * D_REARTH * g3) / 6.0 * c1;
  } else {
    Tmean2 = s2 - (2.0 * (s2 - s1) + ex * a * g2 + ey * D_REARTH * g4) / 6.0;
    Tmean1 = s1 + (2.0 * (s2 - s1) + ex * a * g1 + ey * D_REARTH * g3) / 6.0;
  }
  double cHO = (vflux + av) * Tmean1 + (vflux - av) * Tmean2;
  DA2(t.adv_flux_raw, nz, ed) = -0.5 * (1.0 - num_ord) * cHO - vflux * num_ord * (0.5 * (Tmean1 + Tmean2)) - lo;
}
// CORE2-class meshes: NT tracers of an edge column in one wave.  Everything that does not depend on the tracer (edge -> triangles / nodes, level ranges,
// the volume flux from UV and helem, the up/down-wind index decisions) is formed once, the loads of all tracers are issued together: these kernels
// are bound by the number of loads a wave has in flight, not by bandwidth.  Same expressions per cell as k_flux_hor<FUSED>.
template <bool FUSED, int NT>
__global__ void __launch_bounds__(BLOCK) k_flux_hor_nt(DM m, int tr0) {
  const int trb = tr0 + blockIdx.y * NT;
  int ed = col_id(m), nz = lane_id() + 1;
  if (ed >= m.myD) return;
  if (nz > m.nlm1) return;
  int n1 = m.edges[2 * ed], n2 = m.edges[2 * ed + 1], e1 = m.edge_tri[2 * ed], e2 = m.edge_tri[2 * ed + 1];
  int nl1 = m.nlev[e1] - 1, nu1 = m.ulev[e1], nl2 = 0, nu2 = 0;
  double dX1 = DECD(1, ed), dY1 = DECD(2, ed), dX2 = 0, dY2 = 0;
  double a = D_REARTH * m.elem_cos[e1];
  if (e2 >= 0) {
    dX2 = DECD(3, ed); dY2 = DECD(4, ed);
    nl2 = m.nlev[e2] - 1; nu2 = m.ulev[e2];
    a = 0.5 * (a + D_REARTH * m.elem_cos[e2]);
  }
  int nl12 = min(nl1, nl2), nu12 = max(nu1, nu2);
  bool use1 = false, use2 = false;
  if (nz >= nu12 && nz <= nl12) { use1 = true; use2 = true; }
  else if ((nz >= nu1 && nz <= nu12 - 1) || (nz >= nl12 + 1 && nz <= nl1)) { use1 = true; use2 = false; }
  else if (nu2 > 0 && ((nz >= nu2 && nz <= nu12 - 1) || (nz >= nl12 + 1 && nz <= nl2))) { use1 = false; use2 = true; }
  const bool dead = !use1 && !use2;
  const int hor = m.p.tra_adv_hor;                          // 0 MFCT, 1 MUSCL, 2 UPW1 as the high-order scheme
  UpdnIdx ux;
  if (FUSED && !dead && hor != 2) ux = updn_index(m, ed, nz);
  const bool sel_addr = FUSED && m.cl_grad != nullptr && hor != 2 && (m.exp_batch & 64);      // (wave-uniform)
  // velocities and thicknesses of the two triangles: unconditional loads (a missing second triangle re-reads the first), selected below
  const int e2c = e2 >= 0 ? e2 : e1;
  double u1x = 0, u1y = 0, h1 = 0, u2x = 0, u2y = 0, h2 = 0;
  if (sel_addr) {
    u1x = DV2(m.UV, 1, nz, e1); u1y = DV2(m.UV, 2, nz, e1); h1 = DA2(m.helem, nz, e1);
    u2x = DV2(m.UV, 1, nz, e2c); u2y = DV2(m.UV, 2, nz, e2c); h2 = DA2(m.helem, nz, e2c);
  }
  // loads of all tracers
  double t1[NT], t2[NT], s1[NT], s2[NT], g1[NT], g2[NT], g3[NT], g4[NT];
#pragma unroll
  for (int q = 0; q < NT; q++) {
    const int tr = trb + q < m.ntr ? trb + q : m.ntr - 1;
    const TV t = tracer_view(m, tr);
    t1[q] = t2[q] = s1[q] = s2[q] = g1[q] = g2[q] = g3[q] = g4[q] = 0.0;
    if (dead) continue;
    t1[q] = DTR(m.tr_arr, nz, n1, tr); t2[q] = DTR(m.tr_arr, nz, n2, tr);
    s1[q] = DTR(m.tr_arr_old, nz, n1, tr); s2[q] = DTR(m.tr_arr_old, nz, n2, tr);
    if (hor == 2) continue;
    if (sel_addr) {
      // each of the four up/down-wind values comes from ONE of three places -- the upwind triangle's gradient, the cluster mean of k_cluster_grad at the
      // edge's node, or the entry of edge_up_dn_grad (a level neither covers): the ADDRESS is selected per lane and one load per value is issued
      const double *cg = m.cl_grad + (size_t)tr * 2 * m.nlm1 * m.N;
      const double *p1 = ux.both ? &DV2(t.tr_xy_ab, 1, nz, ux.t1) : (ux.c1 ? &DV2(cg, 1, nz, ux.n1) : &DV4(t.edge_up_dn_grad, 1, nz, ed));
      const double *p3 = ux.both ? &DV2(t.tr_xy_ab, 2, nz, ux.t1) : (ux.c1 ? &DV2(cg, 2, nz, ux.n1) : &DV4(t.edge_up_dn_grad, 3, nz, ed));
      const double *p2 = ux.both ? &DV2(t.tr_xy_ab, 1, nz, ux.t2) : (ux.c2 ? &DV2(cg, 1, nz, ux.n2) : &DV4(t.edge_up_dn_grad, 2, nz, ed));
      const double *p4 = ux.both ? &DV2(t.tr_xy_ab, 2, nz, ux.t2) : (ux.c2 ? &DV2(cg, 2, nz, ux.n2) : &DV4(t.edge_up_dn_grad, 4, nz, ed));
      g1[q] = *p1; g2[q] = *p2; g3[q] = *p3; g4[q] = *p4;
    } else if (FUSED) {
      bool w1, w2;
      if (m.cl_grad) updn_values<true>(m, t, tr, ux, nz, g1[q], g2[q], g3[q], g4[q], w1, w2);       // (wave-uniform: the node field of k_cluster_grad exists)
      else updn_values<false>(m, t, tr, ux, nz, g1[q], g2[q], g3[q], g4[q], w1, w2);
      if (!w1) { g1[q] = DV4(t.edge_up_dn_grad, 1, nz, ed); g3[q] = DV4(t.edge_up_dn_grad, 3, nz, ed); }
      if (!w2) { g2[q] = DV4(t.edge_up_dn_grad, 2, nz, ed); g4[q] = DV4(t.edge_up_dn_grad, 4, nz, ed); }
    } else {
      g1[q] = DV4(t.edge_up_dn_grad, 1, nz, ed); g2[q] = DV4(t.edge_up_dn_grad, 2, nz, ed);
      g3[q] = DV4(t.edge_up_dn_grad, 3, nz, ed); g4[q] = DV4(t.edge_up_dn_grad, 4, nz, ed);
    }
  }
  double vflux = 0.0;
  if (sel_addr) {
    if (use1 && use2) vflux = (-u1y * dX1 + u1x * dY1) * h1 + (u2y * dX2 - u2x * dY2) * h2;
    else if (use1) vflux = (-u1y * dX1 + u1x * dY1) * h1;
    else if (use2) vflux = (u2y * dX2 - u2x * dY2) * h2;
  } else if (use1 && use2)
    vflux = (-DV2(m.UV, 2, nz, e1) * dX1 + DV2(m.UV, 1, nz, e1) * dY1) * DA2(m.helem, nz, e1) +
            (DV2(m.UV, 2, nz, e2) * dX2 - DV2(m.UV, 1, nz, e2) * dY2) * DA2(m.helem, nz, e2);
  else if (use1) vflux = (-DV2(m.UV, 2, nz, e1) * dX1 + DV2(m.UV, 1, nz, e1) * dY1) * DA2(m.helem, nz, e1);
  else if (use2) vflux = (DV2(m.UV, 2, nz, e2) * dX2 - DV2(m.UV, 1, nz, e2) * dY2) * DA2(m.helem, nz, e2);
  const double av = fabs(vflux);
  const double num_ord = m.p.tra_adv_ph;
  const double ex = m.edxy[2 * ed], ey = m.edxy[2 * ed + 1];
#pragma unroll
  for (int q = 0; q < NT; q++) {
    if (trb + q >= m.ntr) continue;
    const TV t = tracer_view(m, trb + q);
    if (dead) { DA2(t.flux_lo_hor, nz, ed) = 0.0; DA2(t.adv_flux_raw, nz, ed) = 0.0; continue; }
    // tra_adv_lim = 'NON' (oce_adv_tra_driver.F90:137-153): no low-order flux, the high-order flux is formed with init_zero=.true. (flux - 0.0)
    double lo = m.p.tra_adv_lim ? 0.0 : -0.5 * (t1[q] * (vflux + av) + t2[q] * (vflux - av)) - 0.0;
    DA2(t.flux_lo_hor, nz, ed) = lo;
    if (hor == 2) {
      DA2(t.adv_flux_raw, nz, ed) = -0.5 * (s1[q] * (vflux + av) + s2[q] * (vflux - av)) - lo;
      continue;
    }
    double Tmean2, Tmean1;
    if (hor == 1) {   // MUSCL: the gradient correction is switched off below nboundary_lay of the node (c_lo = 0 or 1)
      const double c1 = (m.nb_lay[n1] - nz >= 0) ? 1.0 : 0.0, c2 = (m.nb_lay[n2] - nz >= 0) ? 1.0 : 0.0;
      Tmean2 = s2[q] - (2.0 * (s2[q] - s1[q]) + ex * a * g2[q] + ey * D_REARTH * g4[q]) / 6.0 * c2;
      Tmean1 = s1[q] + (2.0 * (s2[q] - s1[q]) + ex * a * g1[q] + ey * D_REARTH * g3[q]) / 6.0 * c1;
    } else {
      Tmean2 = s2[q] - (2.0 * (s2[q] - s1[q]) + ex * a * g2[q] + ey * D_REARTH * g4[q]) / 6.0;
      Tmean1 = s1[q] + (2.0 * (s2[q] - s1[q]) + ex * a * g1[q] + ey * D_REARTH * g3[q]) / 6.0;
    }
    double cHO = (vflux + av) * Tmean1 + (vflux - av) * Tmean2;
    DA2(t.adv_flux_raw, nz, ed) = -0.5 * (1.0 - num_ord) * cHO - vflux * num_ord * (0.5 * (Tmean1 + Tmean2)) - lo;
  }
}
// all tracers of a launch (tr < 0): two per wave; a single tracer: one tracer per wave
template <bool FUSED>
static void launch_flux_hor(const DM &m, hipStream_t s, int tr) {
  static const int env = getenv("FESOM_GPU_EXP_NT") ? atoi(getenv("FESOM_GPU_EXP_NT")) : -1;
  const bool nt2 = tr < 0 && m.ntr > 1 && (env >= 0 ? (env & 1) != 0 : true);
  const int nb = nblocks(SUBN(m, m.myD));
  if (FUSED && m.cl_grad) hipLaunchKernelGGL(k_cluster_grad, dim3(nblocks(m.N), tr < 0 ? m.ntr : 1), dim3(BLOCK), 0, s, m, tr < 0 ? 0 : tr);
  if (nt2) hipLaunchKernelGGL((k_flux_hor_nt<FUSED, 2>), dim3(nb, (m.ntr + 1) / 2), dim3(BLOCK), 0, s, m, 0);
  else hipLaunchKernelGGL((k_flux_hor<FUSED>), dim3(nb, tr < 0 ? m.ntr : 1), dim3(BLOCK), 0, s, m, tr < 0 ? 0 : tr);
}

// low-order solution (src/oce_adv_tra_driver.F90:97-133) with adv_tra_ver_upw1 (src/oce_adv_tra_ver.F90:231-282)
// and adv_tra_ver_qr4c (:286-357) evaluated in registers; also the nodal bounds of oce_tra_adv_fct (:94-101).
__device__ __forceinline__ void k_fct_lo_node_col(const DM &m, const int tr) {
  const TV t = tracer_view(m, tr);
  int n = col_id(m), l = lane_id(), nz = l + 1;
  if (n >= m.myN) return;
  const int nzmin = m.ulev_n[n], nzmax = m.nlev_n[n];
  const double dt = m.p.dt, num_ord = m.p.tra_adv_pv;
  // vertical fluxes at interface nz (lane nz-1)
  // with w_split the low-order solution takes the explicit part of w (Wvel_e), the low-order part of the anti-diffusive flux
  // the full w (oce_adv_tra_driver.F90:111,124-131); without it the two are the same array values
  double fv = 0.0, adf = 0.0;
  const bool split = m.p.w_split != 0;
  // tra_adv_lim = 'NON' (oce_adv_tra_driver.F90:155-177): the high-order vertical flux alone, with the explicit velocity (pwvel => we) and
  // init_zero=.true. (flux - 0.0); no low-order solution
  const bool non = m.p.tra_adv_lim != 0;
  const double *Who = non ? m.Wvel_e : m.Wvel;
  const int ver = m.p.tra_adv_ver;                          // 0 QR4C, 1 CDIFF (adv_tra_ver_cdiff :542-590), 2 UPW1 (:231-282), 3 PPM (:361-538)
  double ppm = 0.0;
  if (ver == 3) {
    // adv_tra_vert_ppm: lane nz holds layer nz and interface nz; interface values tv and the two one-sided fluxes of a layer
    // travel between neighbouring lanes (the whole wave takes part: no divergent exit above)
    const bool lay = nz >= nzmin && nz <= nzmax - 1;
    const double T0 = lay ? DTR(m.tr_arr_old, nz, n, tr) : 0.0, h0 = lay ? DA2(m.hnode_new, nz, n) : 1.0;
    const double Tm1 = shup(T0), Tp1 = shdn(T0), Tp2 = shdn(Tp1);
    const double hm1 = shup(h0), hp1 = shdn(h0), hp2 = shdn(hp1);
    const double Wk = (nz >= nzmin && nz <= nzmax) ? DA2L(Who, nz, n) : 0.0, Wk1 = shdn(Wk);
    const double d0 = Tp1 - T0, dm = T0 - Tm1, dp = Tp2 - Tp1;
    double deltaj = h0 / (hm1 + h0 + hp1) * ((2. * hm1 + h0) / (hp1 + h0) * d0 + (h0 + 2. * hp1) / (hm1 + h0) * dm);
    double deltajp1 = hp1 / (h0 + hp1 + hp2) * ((2. * h0 + hp1) / (hp2 + hp1) * dp + (hp1 + 2. * hp2) / (h0 + hp1) * d0);
    if (d0 * dm > 0.) deltaj = fmin(fmin(fabs(deltaj), 2. * fabs(d0)), 2. * fabs(dm)) * copysign(1.0, deltaj);
    else deltaj = 0.0;
    if (dp * d0 > 0.) deltajp1 = fmin(fmin(fabs(deltajp1), 2. * fabs(dp)), 2. * fabs(d0)) * copysign(1.0, deltajp1);
    else deltajp1 = 0.0;
    const double tvn = T0 + h0 / (h0 + hp1) * d0 +
                       1. / (hm1 + h0 + hp1 + hp2) *
                           ((2. * hp1 * h0) / (h0 + hp1) * ((hm1 + h0) / (2. * h0 + hp1) - (hp2 + hp1) / (2. * hp1 + h0)) * d0 -
                            h0 * (hm1 + h0) / (2. * h0 + hp1) * deltajp1 + hp1 * (hp1 + hp2) / (h0 + 2. * hp1) * deltaj);
    double tvk = shup(tvn);                                   // tv(nz) was formed by layer nz-1 (nz = nzmin+2 .. nzmax-2)
    if (!(nz >= nzmin + 2 && nz <= nzmax - 2)) tvk = 0.0;
    if (nz == nzmin) tvk = T0;                                // the reference's assignment order: a later one wins on short columns
    if (nz == nzmin + 1) tvk = 0.5 * (Tm1 + T0);
    if (nz == nzmax - 1) { const double sg = copysign(1.0, Wk); tvk = -Tm1 * (sg < 0. ? sg : 0.) + T0 * (sg > 0. ? sg : 0.); }
    if (nz == nzmax) tvk = Tm1;
    double aL = tvk, aR = shdn(tvk);
    const double t = T0;
    if ((aR - t) * (t - aL) <= 0.) { aL = t; aR = t; }
    if ((aR - aL) * (t - 0.5 * (aL + aR)) > (aR - aL) * (aR - aL) / 6.) aL = 3. * t - 2. * aR;
    if ((aR - aL) * (t - 0.5 * (aR + aL)) < -((aR - aL) * (aR - aL)) / 6.) aR = 3. * t - 2. * aL;
    const double dz = lay ? DA2(m.hnode, nz, n) : 1.0;
    const double aj = 6.0 * (t - 0.5 * (aL + aR));
    double ftop = 0.0, fbot = 0.0;
    if (lay && Wk > 0.) {
      const double x = fmin(Wk * dt / dz, 1.);
      ftop = (-aL - 0.5 * x * (aR - aL + (1. - 2. / 3. * x) * aj));
      ftop = ftop * DA2L(m.area, nz, n) * Wk;
    }
    if (lay && Wk1 < 0.) {
      const double x = fmin(-Wk1 * dt / dz, 1.);
      fbot = (-aR + 0.5 * x * (aR - aL - (1. - 2. / 3. * x) * aj));
      fbot = fbot * DA2L(m.area, nz + 1, n) * Wk1;
    }
    const double fb_up = shup(fbot);                          // interface nz seen from layer nz-1
    if (lay && Wk > 0.) ppm = ftop;
    if (nz >= nzmin + 1 && nz <= nzmax && Wk < 0.) ppm = fb_up;
    if (nz == nzmin) ppm = -tvk * Wk * DA2L(m.area, nz, n);
    if (nz == nzmax) ppm = 0.0;
  }
  if (nz >= nzmin && nz <= nzmax) {
    double ar = DA2L(m.area, nz, n);
    if (nz == nzmin) {
      fv = -DA2L(m.Wvel_e, nz, n) * DTR(m.tr_arr, nz, n, tr) * ar - 0.0;
      const double fvw = non ? 0.0 : (split ? -DA2L(m.Wvel, nz, n) * DTR(m.tr_arr, nz, n, tr) * ar - 0.0 : fv);
      adf = -DTR(m.tr_arr_old, nz, n, tr) * DA2L(Who, nz, n) * ar - fvw;
      if (ver == 3) adf = ppm - fvw;
    } else if (nz == nzmax) {
      fv = 0.0 - 0.0;
      adf = 0.0 - (non ? 0.0 : fv);
    } else {
      double we = DA2L(m.Wvel_e, nz, n);
      fv = -0.5 * (DTR(m.tr_arr, nz, n, tr) * (we + fabs(we)) + DTR(m.tr_arr, nz - 1, n, tr) * (we - fabs(we))) * ar - 0.0;
      double w = DA2L(m.Wvel, nz, n);
      const double fvw = non ? 0.0 : (split ? -0.5 * (DTR(m.tr_arr, nz, n, tr) * (w + fabs(w)) + DTR(m.tr_arr, nz - 1, n, tr) * (w - fabs(w))) * ar - 0.0 : fv);
      w = DA2L(Who, nz, n);
      double s0 = DTR(m.tr_arr_old, nz, n, tr), sm1 = DTR(m.tr_arr_old, nz - 1, n, tr);
      if (ver == 3) {
        adf = ppm - fvw;
      } else if (ver == 2) {
        adf = -0.5 * (s0 * (w + fabs(w)) + sm1 * (w - fabs(w))) * ar - fvw;
      } else if (nz == nzmin + 1 || nz == nzmax - 1 || ver == 1) {
        adf = -0.5 * (sm1 + s0) * w * ar - fvw;
      } else {
        double sp1 = DTR(m.tr_arr_old, nz + 1, n, tr), sm2 = DTR(m.tr_arr_old, nz - 2, n, tr);
        double z0 = DA2(m.Z_3d_n, nz, n), zm1 = DA2(m.Z_3d_n, nz - 1, n), zp1 = DA2(m.Z_3d_n, nz + 1, n), zm2 = DA2(m.Z_3d_n, nz - 2, n);
        double zb = DA2L(m.zbar_3d_n, nz, n);
        double qc = (sm1 - s0) / (zm1 - z0), qu = (s0 - sp1) / (z0 - zp1), qd = (sm2 - sm1) / (zm2 - zm1);
        double Tmean1 = s0 + (2 * qc + qu) * (zb - z0) / 3.0;
        double Tmean2 = sm1 + (2 * qc + qd) * (zb - zm1) / 3.0;
        double Tmean = (w + fabs(w)) * Tmean1 + (w - fabs(w)) * Tmean2;
        adf = (-0.5 * (1.0 - num_ord) * Tmean - num_ord * (0.5 * (Tmean1 + Tmean2)) * w) * ar - fvw;
      }
    }
    DA2L(t.adv_flux_ver, nz, n) = adf;
  }
  double fv_dn = shdn(fv);
  // low-order horizontal fluxes of the incident edges: lane-parallel edge list, one batch of loads, ordered sum
  const int q0 = m.ne_ptr[n], deg = m.ne_ptr[n + 1] - q0;
  int ed_l = 0, sg_l = 0;
  unsigned rg_l = 1u;
  if (l < deg) { ed_l = m.ne_idx[q0 + l]; sg_l = m.ne_sgn[q0 + l]; rg_l = m.ne_rng[q0 + l]; }
  const int nzc = min(nz, m.nlm1);
  double fl[GATHER_MAXD];
#pragma unroll
  for (int q = 0; q < GATHER_MAXD; q++) fl[q] = q < deg ? DA2(t.flux_lo_hor, nzc, rdlane(ed_l, q)) : 0.0;      // (q < deg: wave-uniform, a scalar branch)
  double lo = 0.0;
#pragma unroll
  for (int q = 0; q < GATHER_MAXD; q++) {
    if (q >= deg) continue;
    unsigned rg = (unsigned)rdlane((int)rg_l, q);
    bool on = nz >= (int)(rg & 0xffu) && nz <= (int)((rg >> 8) & 0xffu);
    double nlo = (rdlane(sg_l, q) > 0) ? lo + fl[q] : lo - fl[q];
    lo = on ? nlo : lo;
  }
  for (int q = GATHER_MAXD; q < deg; q++) {                  // nodes with more incident edges than the batch (rare)
    unsigned rg = m.ne_rng[q0 + q];
    if (nz < (int)(rg & 0xffu) || nz > (int)((rg >> 8) & 0xffu)) continue;
    double f = DA2(t.flux_lo_hor, nz, m.ne_idx[q0 + q]);
    lo = (m.ne_sgn[q0 + q] > 0) ? lo + f : lo - f;
  }
  if (!non && nz >= nzmin && nz <= nzmax - 1) {
    double ttf = DTR(m.tr_arr, nz, n, tr);
    lo = (ttf * DA2(m.hnode, nz, n) + (lo + (fv - fv_dn)) * dt / DA2L(m.areasvol, nz, n)) / DA2(m.hnode_new, nz, n);
    DA2(t.fct_LO, nz, n) = lo;            // the nodal bounds max/min(LO, ttf) (oce_adv_tra_fct.F90:94-101) are formed by their consumer
  }
}
__global__ void __launch_bounds__(BLOCK) k_fct_lo_node(DM m, int tr0) { k_fct_lo_node_col(m, tr0 + blockIdx.y); }      // grid.y = tracers of this launch
template <int NT>
__global__ void __launch_bounds__(BLOCK) k_fct_lo_node_nt(DM m, int tr0) {      // CORE2-class meshes: NT tracers of a column in one wave (shared index chain, half the waves)
#pragma unroll
  for (int q = 0; q < NT; q++)
    if (tr0 + (int)blockIdx.y * NT + q < m.ntr) k_fct_lo_node_col(m, tr0 + (int)blockIdx.y * NT + q);
}

// adv_tra_vert_impl (src/oce_adv_tra_ver.F90:83-227), w_split only: implicit vertical advection by Wvel_i applied to the low-order
// solution (oce_adv_tra_driver.F90:124-126); tridiagonal problem per node column through the in-block Thomas sweep.
__global__ void __launch_bounds__(TH_BLOCK) k_fct_lo_wimpl(DM m, int tr0) {
  extern __shared__ double th_sh[];
  const int tr = tr0 + blockIdx.y;
  const TV t = tracer_view(m, tr);
  int n = col_id_th(m), l = lane_id(), nz = l + 1;
  const bool valid = n < m.myN;
  if (!valid) n = m.myN - 1;
  const int nzmin = m.ulev_n[n], nzmax = m.nlev_n[n];
  const bool wet = valid && nz >= nzmin && nz <= nzmax - 1;
  const double dt = m.p.dt;
  double W = (nz >= nzmin && nz <= nzmax) ? DA2L(m.Wvel_i, nz, n) : 0.0, ar = (nz >= nzmin && nz <= nzmax) ? DA2L(m.area, nz, n) : 0.0;
  const double W_dn = shdn(W), ar_dn = shdn(ar);
  double lo = wet ? DA2(t.fct_LO, nz, n) : 0.0;
  const double lo_up = shup(lo), lo_dn = shdn(lo);
  double a = 0.0, b = 1.0, c = 0.0, rhs = 0.0;
  if (wet) {
    const double zinv = 1.0 * dt, asv = DA2L(m.areasvol, nz, n), hn = DA2(m.hnode_new, nz, n);
    const double v1 = zinv * ar / asv, v2 = zinv * ar_dn / asv;
    if (nz == nzmin) {
      a = 0.0;
      b = hn + W * v1;
      b = b - dmin_(0., W_dn) * v2;
      c = -dmax_(0., W_dn) * v2;
    } else {
      a = dmin_(0., W) * v1;
      b = hn + dmax_(0., W) * v1;
      b = b - dmin_(0., W_dn) * v2;
      c = -dmax_(0., W_dn) * v2;
    }
    if (nz == nzmax - 1) {
      a = dmin_(0., W) * v1;
      b = hn + dmax_(0., W) * v1;
      c = 0.0;
    }
    const double dz = hn;
    if (nz == nzmax - 1) rhs = -a * lo_up - (b - dz) * lo;
    else if (nz == nzmin) rhs = -(b - dz) * lo - c * lo_dn;
    else rhs = -a * lo_up - (b - dz) * lo - c * lo_dn;
  }
  double x, unused;
  thomas_inblock<1>(th_sh, m.nlm1, valid, nzmin, nzmax - 1, a, b, c, rhs, 0.0, x, unused);
  if (wet) DA2(t.fct_LO, nz, n) = lo + x;
}

// nodal bounds max/min(LO, ttf) (src/oce_adv_tra_fct.F90:94-101), element bounds (:108-121, max/min over the 3 nodes;
// the reference parks them in UV_rhs), cluster bounds, sums of positive/negative antidiffusive fluxes, limiting factors
// and the limiting of the vertical antidiffusive flux (:127-311, vlimit=1).  max/min are exact and order-free, so the
// bound "max over the elements around the node that are wet at this level of the max over their 3 nodes" is formed as
// max over the node itself and the far-end nodes of its incident edges that are wet at this level (an edge is wet
// where one of its triangles is): no element array, one launch less on the critical chain, the edge list is shared
// with the flux sums, and fct_ttf_max/min can take their final value (bound - LO) without a race.
__device__ __forceinline__ void k_fct_node_col(const DM &m, const int tr) {
  const TV t = tracer_view(m, tr);
  int n = col_id(m), l = lane_id(), nz = l + 1;
  if (n >= m.myN) return;
  const int nu1 = m.ulev_n[n], nl1 = m.nlev_n[n];
  const double dt = m.p.dt, flux_eps = 1e-16;
  const bool wet = (nz >= nu1 && nz <= nl1 - 1);
  const int nzc = min(nz, m.nlm1);
  // lane-parallel edge list of the node: edge, sign, level range, far-end node
  const int q0 = m.ne_ptr[n], deg = m.ne_ptr[n + 1] - q0;
  int ed_l = 0, sg_l = 0, fn_l = 0;
  unsigned rg_l = 1u;
  if (l < deg) {
    ed_l = m.ne_idx[q0 + l]; sg_l = m.ne_sgn[q0 + l]; rg_l = m.ne_rng[q0 + l];
    fn_l = (sg_l > 0) ? m.edges[2 * ed_l + 1] : m.edges[2 * ed_l];
  }
  const double *LOp = t.fct_LO, *Tp = m.tr_arr + (size_t)tr * m.N * m.nlm1;
  double lo_own = DA2(LOp, nzc, n), t_own = DA2(Tp, nzc, n);
  double adv = (nz >= nu1 && nz <= nl1) ? DA2L(t.adv_flux_ver, nz, n) : 0.0;
  double tvmax = wet ? dmax_(lo_own, t_own) : -1e3, tvmin = wet ? dmin_(lo_own, t_own) : 1e3;   // dry: the reference's -1e3 / 1e3
  double adv_dn = shdn(adv);
  double plus = 0.0 + (dmax_(0.0, adv) + dmax_(0.0, -adv_dn));
  double minus = 0.0 + (dmin_(0.0, adv) + dmin_(0.0, -adv_dn));
  // the incident edges in groups of FN_B: the 3 * FN_B column loads of a group are issued back to back (a slot beyond the node's degree re-reads slot 0 and is
  // dropped in the selects), then folded -- a guard around every single slot (round 3, first form) made the loads of one slot wait for the previous slot's:
  // 6+ dependent round trips per wave where this has 1 or 2.  The flux sums keep the edge order; max / min do not depend on it.
  constexpr int FN_B = 6;
#pragma unroll
  for (int g0 = 0; g0 < GATHER_MAXD; g0 += FN_B) {
    if (g0 > 0 && deg <= g0) break;                           // (wave-uniform)
    double lk[FN_B], tk[FN_B], fh[FN_B];
#pragma unroll
    for (int j = 0; j < FN_B; j++) {
      const int q = g0 + j < deg ? g0 + j : 0;
      const int k = rdlane(fn_l, q);
      lk[j] = DA2(LOp, nzc, k); tk[j] = DA2(Tp, nzc, k);
      fh[j] = DA2(t.adv_flux_raw, nzc, rdlane(ed_l, q));
    }
#pragma unroll
    for (int j = 0; j < FN_B; j++) {
      const int q = g0 + j < deg ? g0 + j : 0;
      const unsigned rg = (unsigned)rdlane((int)rg_l, q);
      const bool on = g0 + j < deg && nz >= (int)(rg & 0xffu) && nz <= (int)((rg >> 8) & 0xffu);
      tvmax = on ? dmax_(tvmax, dmax_(lk[j], tk[j])) : tvmax;
      tvmin = on ? dmin_(tvmin, dmin_(lk[j], tk[j])) : tvmin;
      const double f = (rdlane(sg_l, q) < 0) ? -fh[j] : fh[j];
      const double np = plus + dmax_(0.0, f), nm = minus + dmin_(0.0, f);
      plus = on ? np : plus; minus = on ? nm : minus;
    }
  }
  for (int q = GATHER_MAXD; q < deg; q++) {                  // nodes with more incident edges than the batch (rare)
    unsigned rg = m.ne_rng[q0 + q];
    if (nz < (int)(rg & 0xffu) || nz > (int)((rg >> 8) & 0xffu)) continue;
    int ed = m.ne_idx[q0 + q];
    int k = (m.ne_sgn[q0 + q] > 0) ? m.edges[2 * ed + 1] : m.edges[2 * ed];
    double lk = DA2(LOp, nzc, k), tk = DA2(Tp, nzc, k);
    tvmax = dmax_(tvmax, dmax_(lk, tk)); tvmin = dmin_(tvmin, dmin_(lk, tk));
    double f = DA2(t.adv_flux_raw, nz, ed);
    if (m.ne_sgn[q0 + q] < 0) f = -f;
    plus = plus + dmax_(0.0, f);
    minus = minus + dmin_(0.0, f);
  }
  // under an ice shelf: an element around the node that starts below this level (nz < ulevels(elem)) contributes the untouched entries of the reference's
  // scratch array -- UV_rhs, which no routine writes above an element's upper level, i.e. 0 -- to both bounds (src/oce_adv_tra_fct.F90:110-121,141-142)
  if (wet && nz < m.ulev_n_max[n]) { tvmax = dmax_(tvmax, 0.0); tvmin = dmin_(tvmin, 0.0); }
  double mx_u = shup(tvmax), mx_d = shdn(tvmax), mn_u = shup(tvmin), mn_d = shdn(tvmin);
  if (wet) {
    double lo = DA2(t.fct_LO, nz, n);
    double bmax, bmin;
    if (nz >= nu1 + 1 && nz <= nl1 - 2) {
      bmax = dmax_(dmax_(mx_u, tvmax), mx_d) - lo;
      bmin = dmin_(dmin_(mn_u, tvmin), mn_d) - lo;
    } else { bmax = tvmax - lo; bmin = tvmin - lo; }
    DA2(t.fct_ttf_max, nz, n) = bmax;
    DA2(t.fct_ttf_min, nz, n) = bmin;
    double asv = DA2L(m.areasvol, nz, n);
    double flux = plus * dt / asv + flux_eps;
    plus = dmin_(1.0, bmax / flux);
    flux = minus * dt / asv - flux_eps;
    minus = dmin_(1.0, bmin / flux);
    DA2(t.fct_plus, nz, n) = plus;
    DA2(t.fct_minus, nz, n) = minus;
  } else { plus = 0.0; minus = 0.0; }
  double plus_u = shup(plus), minus_u = shup(minus);
  if (wet) {
    double ae = 1.0;
    if (nz == nu1) {
      if (adv >= 0.0) ae = dmin_(ae, plus); else ae = dmin_(ae, minus);
    } else {
      if (adv >= 0.) { ae = dmin_(ae, minus_u); ae = dmin_(ae, plus); }
      else { ae = dmin_(ae, plus_u); ae = dmin_(ae, minus); }
    }
    DA2L(t.adv_flux_ver, nz, n) = ae * adv;
  }
}
__global__ void __launch_bounds__(BLOCK) k_fct_node(DM m, int tr0) { k_fct_node_col(m, tr0 + blockIdx.y); }      // grid.y = tracers of this launch
template <int NT>
__global__ void __launch_bounds__(BLOCK) k_fct_node_nt(DM m, int tr0) {      // CORE2-class meshes: NT tracers of a column in one wave (shared index chain, half the waves)
#pragma unroll
  for (int q = 0; q < NT; q++)
    if (tr0 + (int)blockIdx.y * NT + q < m.ntr) k_fct_node_col(m, tr0 + (int)blockIdx.y * NT + q);
}

// limiting of the horizontal antidiffusive flux (src/oce_adv_tra_fct.F90:318-347): adv_flux_hor = ae * adv_flux_raw.
// Off the critical chain: k_tr_update applies the same factors on the fly; this kernel only materialises the field.
__device__ __forceinline__ void k_fct_edge_limit_col(const DM &m, const int tr) {
  const TV t = tracer_view(m, tr);
  int ed = col_id(m), nz = lane_id() + 1;
  if (ed >= m.myD) return;
  int n1 = m.edges[2 * ed], n2 = m.edges[2 * ed + 1], e1 = m.edge_tri[2 * ed], e2 = m.edge_tri[2 * ed + 1];
  int nl12 = m.nlev[e1] - 1, nu12 = m.ulev[e1];
  if (e2 >= 0) { nl12 = max(nl12, m.nlev[e2] - 1); nu12 = min(nu12, m.ulev[e2]); }
  if (nz < nu12 || nz > nl12) return;
  double ae = 1.0, flux = DA2(t.adv_flux_raw, nz, ed);
  if (m.p.tra_adv_lim) ae = 1.0;                           // 'NON': the high-order flux itself
  else if (flux >= 0.) { ae = dmin_(ae, DA2(t.fct_plus, nz, n1)); ae = dmin_(ae, DA2(t.fct_minus, nz, n2)); }
  else { ae = dmin_(ae, DA2(t.fct_minus, nz, n1)); ae = dmin_(ae, DA2(t.fct_plus, nz, n2)); }
  DA2(t.adv_flux_hor, nz, ed) = ae * flux;
}
__global__ void __launch_bounds__(BLOCK) k_fct_edge_limit(DM m, int tr0) { k_fct_edge_limit_col(m, tr0 + blockIdx.y); }      // grid.y = tracers of this launch
template <int NT>
__global__ void __launch_bounds__(BLOCK) k_fct_edge_limit_nt(DM m, int tr0) {      // CORE2-class meshes: NT tracers of a column in one wave (shared index chain, half the waves)
#pragma unroll
  for (int q = 0; q < NT; q++)
    if (tr0 + (int)blockIdx.y * NT + q < m.ntr) k_fct_edge_limit_col(m, tr0 + (int)blockIdx.y * NT + q);
}

// Horizontal diffusive flux through every edge (diff_part_hor_redi src/oce_ale_tracer.F90:929-1077, Redi off): the value
// the reference adds to / subtracts from the two end nodes.  It only needs T^n gradients, Ki and helem of the current
// step, so it is computed edge-parallel during tracer preparation (hidden under the SSH solve) and k_tr_update just
// gathers it in reference order.
template <bool REDI>
__device__ __forceinline__ void k_diff_flux_col(const DM &m, const int tr) {
  const TV t = tracer_view(m, tr);
  int ed = col_id(m), nz = lane_id() + 1;
  if (ed >= m.myD || nz > m.nlm1) return;
  int n1 = m.edges[2 * ed], n2 = m.edges[2 * ed + 1], e1 = m.edge_tri[2 * ed], e2 = m.edge_tri[2 * ed + 1];
  int nl1 = m.nlev[e1] - 1, ul1 = m.ulev[e1], nl2 = 0, ul2 = 0;
  double dX1 = DECD(1, ed), dY1 = DECD(2, ed), dX2 = 0, dY2 = 0;
  if (e2 >= 0) { nl2 = m.nlev[e2] - 1; ul2 = m.ulev[e2]; dX2 = DECD(3, ed); dY2 = DECD(4, ed); }
  int nl12 = min(nl1, nl2), ul12 = max(ul1, ul2);
  int hi = max(nl1, nl2), lo = ul1;
  if (ul2 > 0) lo = min(ul1, ul2);
  if (nz < lo || nz > hi) return;
  double Kh = (DA2(m.Ki, nz, n1) + DA2(m.Ki, nz, n2)) / 2.0, c;
  double ax = 0.0, ay = 0.0;                              // Redi: slope * vertical gradient at the two edge nodes (:990-993)
  if (REDI) {
    double Tz1 = 0.5 * (DA2L(t.tr_z, nz, n1) + DA2L(t.tr_z, nz + 1, n1)), Tz2 = 0.5 * (DA2L(t.tr_z, nz, n2) + DA2L(t.tr_z, nz + 1, n2));
    ax = (Tz1 * DV3(m.slope_tapered, 1, nz, n1) + Tz2 * DV3(m.slope_tapered, 1, nz, n2)) / 2.0;
    ay = (Tz1 * DV3(m.slope_tapered, 2, nz, n1) + Tz2 * DV3(m.slope_tapered, 2, nz, n2)) / 2.0;
    ax = ax * 1.0; ay = ay * 1.0;
  }
  if (m.exp_batch & 2) {
    // the element values of both triangles in the same batch of loads as the node values above (a level only one triangle reaches reads the other one's
    // entry too -- inside its column, e2 < 0 falls back to e1 -- and drops it in the selects): no divergent second round of loads
    const int e2c = e2 >= 0 ? e2 : e1;
    const double h1 = DA2(m.helem, nz, e1), h2 = DA2(m.helem, nz, e2c);
    const double x1 = DV2(t.tr_xy, 1, nz, e1), y1 = DV2(t.tr_xy, 2, nz, e1), x2 = DV2(t.tr_xy, 1, nz, e2c), y2 = DV2(t.tr_xy, 2, nz, e2c);
    if (nz >= ul12 && nz <= nl12) {
      double dz = (h1 + h2) / 2.0;
      double Tx = 0.5 * (x1 + x2);
      double Ty = 0.5 * (y1 + y2);
      double Fx = Kh * (Tx + ax), Fy = Kh * (Ty + ay);
      c = ((dX2 - dX1) * Fy - (dY2 - dY1) * Fx) * dz;
    } else if ((nz >= ul1 && nz <= ul12 - 1) || (nz >= nl12 + 1 && nz <= nl1)) {
      double Fx = Kh * (x1 + ax), Fy = Kh * (y1 + ay);
      c = (-dX1 * Fy + dY1 * Fx) * h1;
    } else {
      double Fx = Kh * (x2 + ax), Fy = Kh * (y2 + ay);
      c = (dX2 * Fy - dY2 * Fx) * h2;
    }
  } else if (nz >= ul12 && nz <= nl12) {
    double dz = (DA2(m.helem, nz, e1) + DA2(m.helem, nz, e2)) / 2.0;
    double Tx = 0.5 * (DV2(t.tr_xy, 1, nz, e1) + DV2(t.tr_xy, 1, nz, e2));
    double Ty = 0.5 * (DV2(t.tr_xy, 2, nz, e1) + DV2(t.tr_xy, 2, nz, e2));
    double Fx = Kh * (Tx + ax), Fy = Kh * (Ty + ay);
    c = ((dX2 - dX1) * Fy - (dY2 - dY1) * Fx) * dz;
  } else if ((nz >= ul1 && nz <= ul12 - 1) || (nz >= nl12 + 1 && nz <= nl1)) {
    double dz = DA2(m.helem, nz, e1);
    double Fx = Kh * (DV2(t.tr_xy, 1, nz, e1) + ax), Fy = Kh * (DV2(t.tr_xy, 2, nz, e1) + ay);
    c = (-dX1 * Fy + dY1 * Fx) * dz;
  } else {
    double dz = DA2(m.helem, nz, e2);
    double Fx = Kh * (DV2(t.tr_xy, 1, nz, e2) + ax), Fy = Kh * (DV2(t.tr_xy, 2, nz, e2) + ay);
    c = (dX2 * Fy - dY2 * Fx) * dz;
  }
  DA2(t.diff_flux, nz, ed) = c;
}
template <bool REDI>
__global__ void __launch_bounds__(BLOCK) k_diff_flux(DM m, int tr0) { k_diff_flux_col<REDI>(m, tr0 + blockIdx.y); }      // grid.y = tracers of this launch
template <bool REDI, int NT>
__global__ void __launch_bounds__(BLOCK) k_diff_flux_nt(DM m, int tr0) {      // CORE2-class meshes: NT tracers of a column in one wave (shared index chain, half the waves)
#pragma unroll
  for (int q = 0; q < NT; q++)
    if (tr0 + (int)blockIdx.y * NT + q < m.ntr) k_diff_flux_col<REDI>(m, tr0 + (int)blockIdx.y * NT + q);
}

// oce_tra_adv_flux2dtracer (src/oce_adv_tra_driver.F90:201-269) + adv_tracers_ale tail (src/oce_ale_tracer.F90:241)
// + diff_tracers_ale (:253-325): horizontal diffusion (k_diff_flux) gathered over edges, T* update, coefficients of the
// implicit vertical diffusion (diff_ver_part_impl_ale :398-856) with its in-block Thomas sweep, salinity clamp (:176-198).
// The kernel is latency-bound, not bandwidth-bound: the edge list of the node is read lane-parallel (lane q = q-th
// incident edge), broadcast with v_readlane, and all edge values are fetched in one batch before the ordered sums.
#define TRU_MAXD 10                     // batch of this kernel (4 loads per edge): keeps it at <= 128 VGPRs, 2 blocks per CU
#define TRU_MAXD_TILE 6                 // tile shapes: two tracers per block, 6 edges per batch (the typical degree; nodes of higher degree take the remainder loop)
// Shapes (dev.h:ThTile): <REDI, 1, 8, 8> = one column per wave, one tracer per block row (pi); <REDI, 2, TL_COLS, TL_WAVES> =
// tiles of TL_COLS columns, several columns per wave, BOTH tracers of a column in the same block (CORE2-class meshes): the
// tracer-independent part of the column (thicknesses, interface depths, the coefficients a, b, c of the implicit operator, the
// edge list) is formed once and the sweep solves the two right-hand sides together (3 divides per level instead of 2 x 2).
struct TruCol {                          // tracer-independent part of one node column (lane = level)
  int n, nzmin, nzmax, q0, deg, ed_l, sg_l, fn_l;
  unsigned rg_l;
  bool valid, wet;
  double hn, hnn, asv;                                   // tru_head
  double zb_top, zb_bot, Zn, Zn_up, Zn_dn, ki, s3sq, s1, s2;   // tru_zcol (ki .. s2: Redi only)
  double a, b, c, ar_dn;                                 // tru_coeffs
};
// The parts are called in the order  head, [hor, (zcol once), fin] per tracer, coeffs, rhs per tracer : what the horizontal
// gathers keep live (two batches of edge values) never overlaps with the interface depths and the coefficients.
__device__ __forceinline__ void tru_head(const DM &m, int n_in, TruCol &k) {
  const int l = lane_id(), nz = l + 1;
  int n = n_in;
  k.valid = n < m.myN;
  if (!k.valid) n = m.myN - 1;
  k.n = n;
  k.nzmin = m.ulev_n[n]; k.nzmax = m.nlev_n[n];
  k.wet = k.valid && (nz >= k.nzmin && nz <= k.nzmax - 1);
  k.q0 = m.ne_ptr[n]; k.deg = m.ne_ptr[n + 1] - k.q0;
  k.ed_l = 0; k.sg_l = 0; k.fn_l = 0; k.rg_l = 1u;          // lo = 1, hi = 0: empty range
  if (l < k.deg) {
    k.ed_l = m.ne_idx[k.q0 + l]; k.sg_l = m.ne_sgn[k.q0 + l]; k.rg_l = m.ne_rng[k.q0 + l];
    k.fn_l = (k.sg_l > 0) ? m.edges[2 * k.ed_l + 1] : m.edges[2 * k.ed_l];       // far-end node of the edge
  }
  k.hn = 0.0; k.hnn = 1.0; k.asv = 1.0;
  if (k.wet) { k.hn = DA2(m.hnode, nz, n); k.hnn = DA2(m.hnode_new, nz, n); k.asv = DA2L(m.areasvol, nz, n); }
}
// flux -> tendency (oce_tra_adv_flux2dtracer) and horizontal diffusion of tracer `tr`: T^n and del (lane = level)
// FAST: the quotients by areasvol(nz,n) share one reciprocal (dev.h: div_by, same bits as '/'); returns true if a lane left the
// range in which that holds -- the caller then repeats the call with FAST = false (plain divisions)
template <bool FAST, int MAXD, bool NON = false>        // NON: tra_adv_lim = 'NON' (no limiter, no low-order solution); a compile-time path so that the default costs nothing
__device__ __forceinline__ bool tru_hor(const DM &m, const TruCol &k, int tr, double &T, double &del) {
  const TV t = tracer_view(m, tr);
  const int l = lane_id(), nz = l + 1, n = k.n, nzmin = k.nzmin, nzmax = k.nzmax;
  const double dt = m.p.dt, asv = k.asv, hn = k.hn, hnn = k.hnn;
  const bool wet = k.wet;
  const int nzc = min(nz, m.nlm1);
  const bool dif = m.p.with_diffusion != 0;
  constexpr bool non = NON;                                // 'NON': no limiter (factor 1), no low-order solution in the vertical update
  constexpr bool GUARD = MAXD > 6;
  const double p_own = non ? 1.0 : UA2(t.fct_plus, nzc, n), m_own = non ? 1.0 : UA2(t.fct_minus, nzc, n);
  // Per edge of the batch: the limited antidiffusive flux and the diffusive flux, already masked with the level range of the
  // edge (+0.0 outside: x + 0.0 == x for the running sums below, which start at +0.0 and therefore never are -0.0) and carrying
  // the sign of this node's end of the edge (x - f == x + (-f) bit for bit), so that the ordered sums are plain additions.
  double fa[MAXD], fd[MAXD];
#pragma unroll
  for (int q = 0; q < MAXD; q++) {                       // one batch of independent loads
    fa[q] = 0.0; fd[q] = 0.0;
    // wave-uniform: slots beyond the node's degree cost a scalar branch instead of four loads.  Only where the batch is wider than the typical degree
    // (pi: 10 slots, 51 -> 46 us); with the 6 slots of the tile shapes the branches break up the batch of loads and cost more than they save (1093 -> 1281 us)
    // (one guarded GROUP for the slots beyond the sixth instead of a branch per slot, as in k_fct_node: measured, no gain on pi -- 46.1 us either way -- and 126 instead of 96 VGPRs)
    if (GUARD && q >= k.deg) continue;
    int ed = rdlane(k.ed_l, q), kk = rdlane(k.fn_l, q);
    const bool first = rdlane(k.sg_l, q) > 0;               // this node is edges(1,ed)  (wave-uniform)
    const unsigned rg = (unsigned)rdlane((int)k.rg_l, q);
    const bool on = nz >= (int)(rg & 0xffu) && nz <= (int)((rg >> 8) & 0xffu);
    const int flip = first ? 0 : (int)0x80000000;
    // limited antidiffusive flux ae * flux with the factors of oce_adv_tra_fct.F90:318-347 applied on the fly:
    // flux >= 0: min(1, plus(n1), minus(n2)); flux < 0: min(1, minus(n1), plus(n2)).  With s = (flux >= 0) == (this node is n1)
    // the factor of this node is s ? plus : minus, that of the far node s ? minus : plus.
    double fr = UA2(t.adv_flux_raw, nzc, ed), p_far = non ? 1.0 : UA2(t.fct_plus, nzc, kk), m_far = non ? 1.0 : UA2(t.fct_minus, nzc, kk);
    const bool sel = (fr >= 0.) == first;
    double ae = non ? 1.0 : dmin_(dmin_(1.0, sel ? p_own : m_own), sel ? m_far : p_far);
    double f = ae * fr;
    f = __hiloint2double(__double2hiint(f) ^ flip, __double2loint(f));
    fa[q] = on ? f : 0.0;
    double d = dif ? UA2(t.diff_flux, nzc, ed) : 0.0;
    d = 0.0 + __hiloint2double(__double2hiint(d) ^ flip, __double2loint(d));       // (0.0 + fd) resp. (0.0 - fd) of the reference
    fd[q] = on ? d : 0.0;
  }
  double adv = (nz >= nzmin && nz <= nzmax) ? UA2L(t.adv_flux_ver, nz, n) : 0.0;
  double adv_dn = shdn(adv);
  T = 0.0;
  if (wet) T = UTR(m.tr_arr, nz, n, tr);
  bool bad = false;
  const RcpD rasv = rcp_prepare(asv, bad);                 // ~21 quotients by areasvol(nz,n) follow
#define QDIV(x) (FAST ? div_by((x), rasv, bad) : (x) / asv)
  double dv = non ? 0.0 : 0.0 - T * hn + (wet ? UA2(t.fct_LO, nz, n) : 0.0) * hnn;      // flux2dtracer use_lo (oce_adv_tra_driver.F90:222-233)
  dv = dv + QDIV((adv - adv_dn) * dt);
  double dh = 0.0;
#pragma unroll
  for (int q = 0; q < MAXD; q++) if (!GUARD || q < k.deg) dh = dh + QDIV(fa[q] * dt);          // (a skipped slot would add + 0.0: the same bits)
  for (int q = MAXD; q < k.deg; q++) {                  // nodes with more incident edges than the batch (rare)
    int ed = m.ne_idx[k.q0 + q];
    unsigned rg = m.ne_rng[k.q0 + q];
    if (nz < (int)(rg & 0xffu) || nz > (int)((rg >> 8) & 0xffu)) continue;
    int n1 = m.edges[2 * ed], n2 = m.edges[2 * ed + 1];
    double fr = UA2(t.adv_flux_raw, nz, ed);
    double ae = non ? 1.0 : dmin_(dmin_(1.0, (fr >= 0.) ? UA2(t.fct_plus, nz, n1) : UA2(t.fct_minus, nz, n1)), (fr >= 0.) ? UA2(t.fct_minus, nz, n2) : UA2(t.fct_plus, nz, n2));
    double f = (ae * fr) * dt / asv;
    dh = (m.ne_sgn[k.q0 + q] > 0) ? dh + f : dh - f;
  }
  del = 0.0 + dh + dv;
  if (dif) {
#pragma unroll
    for (int q = 0; q < MAXD; q++) if (!GUARD || q < k.deg) del = del + QDIV(fd[q] * dt);
    for (int q = MAXD; q < k.deg; q++) {
      int ed = m.ne_idx[k.q0 + q];
      unsigned rg = m.ne_rng[k.q0 + q];
      if (nz < (int)(rg & 0xffu) || nz > (int)((rg >> 8) & 0xffu)) continue;
      double c = UA2(t.diff_flux, nz, ed);
      double r_ = (m.ne_sgn[k.q0 + q] > 0) ? 0.0 + c : 0.0 - c;
      del = del + r_ * dt / asv;
    }
  }
#undef QDIV
  return FAST && __any(bad);
}
// zbar_n / Z_n of the node column from hnode_new (bottom-up, reference order); Redi: slopes and Ki of the column
template <bool REDI>
__device__ __forceinline__ void tru_zcol(const DM &m, TruCol &k) {
  const int nz = lane_id() + 1, n = k.n;
  k.zb_top = seq_sum_down(k.wet ? k.hnn : 0.0, k.nzmax - 2, k.nzmin - 1, m.zbar_n_bot[n]);
  k.zb_bot = shdn(k.zb_top);
  if (nz == k.nzmax - 1) k.zb_bot = m.zbar_n_bot[n];
  k.Zn = k.zb_bot + k.hnn / 2.0;
  k.Zn_up = shup(k.Zn); k.Zn_dn = shdn(k.Zn);
  k.ki = 0.0; k.s3sq = 0.0; k.s1 = 0.0; k.s2 = 0.0;
  if (REDI && k.wet) {
    k.s1 = DV3(m.slope_tapered, 1, nz, n); k.s2 = DV3(m.slope_tapered, 2, nz, n);
    double s3 = DV3(m.slope_tapered, 3, nz, n);
    k.s3sq = s3 * s3; k.ki = DA2(m.Ki, nz, n);
  }
}
// Redi's explicit vertical flux, T* update, copies: returns T* in T
template <bool REDI>
__device__ __forceinline__ void tru_fin(const DM &m, const TruCol &k, int tr, double &T, double del) {
  const TV t = tracer_view(m, tr);
  const int nz = lane_id() + 1, n = k.n, nzmin = k.nzmin, nzmax = k.nzmax;
  const double dt = m.p.dt, asv = k.asv, hn = k.hn, hnn = k.hnn;
  const bool wet = k.wet;
  if (REDI) {            // diff_ver_part_redi_expl (:860-927): vertical flux of the isoneutral tensor's off-diagonal part
    double Tx = 0.0, Ty = 0.0, G = 0.0;
    {
      // the elements around the node: their index chain (element, level range, area) once per lane, the gradients in batches of independent loads;
      // the sums stay in the reference's element order
      const int l = lane_id(), num = m.nie_num[n], nzc = nz <= m.nlm1 ? nz : m.nlm1;
      int el_l = 0, rg_l = 1;                          // range packed lo | hi << 8 ; (1, 0) = empty
      double ar_l = 0.0;
      if (l < num) { el_l = m.nie[(size_t)m.maxk * n + l]; rg_l = m.ulev[el_l] | ((m.nlev[el_l] - 1) << 8); ar_l = m.elem_area[el_l]; }
      constexpr int RB = 6;
      for (int q0 = 0; q0 < num; q0 += RB) {
        double tx[RB], ty[RB];
#pragma unroll
        for (int j = 0; j < RB; j++) {
          const int el = rdlane(el_l, q0 + j < num ? q0 + j : 0);
          tx[j] = DV2(t.tr_xy, 1, nzc, el); ty[j] = DV2(t.tr_xy, 2, nzc, el);
        }
#pragma unroll
        for (int j = 0; j < RB; j++) {
          if (q0 + j < num) {
            const int rg = rdlane(rg_l, q0 + j);
            const double ar = bcast(ar_l, q0 + j);
            const bool on = wet && nz >= (rg & 0xff) && nz <= (rg >> 8);
            const double ax = Tx + tx[j] * ar, ay = Ty + ty[j] * ar;
            Tx = on ? ax : Tx; Ty = on ? ay : Ty;
          }
        }
      }
    }
    if (wet) {
      Tx = Tx / 3.0 / asv; Ty = Ty / 3.0 / asv;
      G = k.s1 * Tx + k.s2 * Ty;
    }
    double G_up = shup(G), ki_up = shup(k.ki);
    double vd = 0.0;
    if (nz >= nzmin + 1 && nz <= nzmax - 1) {
      vd = (k.Zn_up - k.zb_top) * G_up * ki_up;
      vd = vd + (k.zb_top - k.Zn) * G * k.ki;
      vd = vd / (k.Zn_up - k.Zn) * DA2L(m.area, nz, n);
    }
    double vd_dn = shdn(vd);
    if (nz == nzmax - 1) vd_dn = 0.0;
    if (wet) del = del + (vd - vd_dn) * dt / asv;
  }
  if (wet) {
    DTR(m.tr_arr_old, nz, n, tr) = T;          // tr_arr_old(:,:,tr) = tr_arr(:,:,tr)  (oce_ale_tracer.F90:274)
    del = del + T * (hn - hnn);
    DA2(t.del_ttf, nz, n) = del;
    T = T + del / hnn;
  }
  if (k.valid && !wet && nz <= m.nlm1) DTR(m.tr_arr_old, nz, n, tr) = DTR(m.tr_arr, nz, n, tr);   // whole-array copy incl. dry cells
}
// coefficients of the implicit vertical diffusion (diff_ver_part_impl_ale :398-856): the same for every tracer
template <bool REDI>
__device__ __forceinline__ void tru_coeffs(const DM &m, TruCol &k) {
  const int nz = lane_id() + 1, n = k.n;
  k.a = 0.0; k.b = 1.0; k.c = 0.0; k.ar_dn = 0.0;
  const double ki_up = shup(k.ki), ki_dn = shdn(k.ki), sq_up = shup(k.s3sq), sq_dn = shdn(k.s3sq);
  if (k.wet) {
    const double dt = m.p.dt, hnn = k.hnn, asv = k.asv;
    double zinv = 1.0 * dt;
    double zinv1 = 1.0 / (k.Zn_up - k.Zn), zinv2 = 1.0 / (k.Zn - k.Zn_dn);
    double Ty = 0.0, Ty1 = 0.0;              // K33 = slope^2 * Ki of the isoneutral tensor (:528-599), isredi = 1
    if (REDI) {
      if (nz > k.nzmin) Ty = (k.Zn_up - k.zb_top) * zinv1 * sq_up * ki_up + (k.zb_top - k.Zn) * zinv1 * k.s3sq * k.ki;
      if (nz <= k.nzmax - 2) Ty1 = (k.Zn - k.zb_bot) * zinv2 * k.s3sq * k.ki + (k.zb_bot - k.Zn_dn) * zinv2 * sq_dn * ki_dn;
      Ty = Ty * 1.0; Ty1 = Ty1 * 1.0;
    }
    double ar = DA2L(m.area, nz, n), ar_dn = DA2L(m.area, nz + 1, n);
    double kv = DA2L(m.Kv, nz, n), kv_dn = DA2L(m.Kv, nz + 1, n);
    k.ar_dn = ar_dn;
    if (nz == k.nzmin) {
      k.a = 0.0;
      k.c = -(kv_dn + Ty1) * zinv2 * zinv * ar_dn / asv;
      k.b = -k.c + hnn;
    } else if (nz <= k.nzmax - 2) {
      k.a = -(kv + Ty) * zinv1 * zinv * (ar / asv);
      k.c = -(kv_dn + Ty1) * zinv2 * zinv * ar_dn / asv;
      k.b = -k.a - k.c + hnn;
    } else {
      k.a = -(kv + Ty) * zinv1 * zinv * (ar / asv);
      k.c = 0.0;
      k.b = -k.a + hnn;
    }
    if (m.p.w_split && m.p.tra_adv_lim) {      // do_wimpl (:424, :560-572, :604-617, :641-649): without the FCT low-order solution the implicit part of the vertical velocity enters the solve, upwind
      const double wi = DA2L(m.Wvel_i, nz, n), wi_dn = DA2L(m.Wvel_i, nz + 1, n);
      double v_adv = zinv * (ar / asv);
      if (nz == k.nzmin) k.b = k.b + wi * v_adv;
      else { k.a = k.a + dmin_(0.0, wi) * v_adv; k.b = k.b + dmax_(0.0, wi) * v_adv; }
      if (nz <= k.nzmax - 2) {
        v_adv = zinv * ar_dn / asv;
        k.b = k.b - dmin_(0.0, wi_dn) * v_adv;
        k.c = k.c - dmax_(0.0, wi_dn) * v_adv;
      }
    }
  }
}
// right-hand side of the implicit vertical diffusion for T* (surface boundary condition, short-wave penetration)
__device__ __forceinline__ double tru_rhs(const DM &m, const TruCol &k, int tr, double T) {
  const int nz = lane_id() + 1, n = k.n, nzmin = k.nzmin, nzmax = k.nzmax;
  const double dt = m.p.dt, asv = k.asv, hnn = k.hnn;
  double rhs = 0.0;
  double T_up = shup(T), T_dn = shdn(T);
  if (k.wet) {
    const double a = k.a, b = k.b, c = k.c;
    double zinv = 1.0 * dt;
    if (nz == nzmin) rhs = -(b - hnn) * T - c * T_dn;
    else if (nz <= nzmax - 2) rhs = -a * T_up - (b - hnn) * T - c * T_dn;
    else rhs = -a * T_up - (b - hnn) * T;
    if (m.p.use_kpp_nonlclflx && m.p.mix_scheme == 1 && tr < 2) {      // KPP non-local transport of heat / salt (oce_ale_tracer.F90:688-724)
      const double *bl = m.kpp_blmc + (size_t)(tr + 1) * m.nl * m.N + (size_t)n * m.nl;           // blmc(:, n, 2) for heat, (:, n, 3) for salt
      const double g0 = dmin_(DA2(m.kpp_ghats, nz, n) * bl[nz - 1], 1.0) * (DA2L(m.area, nz, n) / asv);
      const int nzp = nz + 1 <= m.nlm1 ? nz + 1 : m.nlm1;
      const double g1 = dmin_(DA2(m.kpp_ghats, nzp, n) * bl[nzp - 1], 1.0) * (k.ar_dn / asv);
      const double X = (nz == nzmin) ? -g1 : ((nz <= nzmax - 2) ? g0 - g1 : g0);
      if (tr == 0) rhs = rhs + X * m.heat_flux[n] / D_VCPW * dt;
      else {
        const double s1 = bcast(T, 0);                 // tr_arr(1, n, 2) at this point of the reference = T* of the first level
        const double rsss = m.p.ref_sss_local ? s1 : m.p.ref_sss;
        rhs = rhs - X * rsss * m.water_flux[n] * dt;
      }
    }
    if (m.p.use_sw_pene && tr == 0)            // short-wave penetration (oce_ale_tracer.F90:785-791)
      rhs = rhs + (DA2L(m.sw_3d, nz, n) - DA2L(m.sw_3d, nz + 1, n) * k.ar_dn / asv) * zinv;
    if (nz == nzmin) {
      double nonlin = (m.p.which_ale == 0) ? 0.0 : 1.0, bc;
      if (tr == 0) bc = -dt * (m.heat_flux[n] / D_VCPW + T * m.water_flux[n] * nonlin);
      else if (tr == 1) bc = dt * (m.virtual_salt[n] + m.relax_salt[n] - m.real_salt_flux[n] * nonlin);
      else bc = 0.0;
      rhs = rhs + bc;
    }
  }
  return rhs;
}
// the reference clamps the salinity after the tracer loop (oce_ale_tracer.F90:176-198): when a filter or a relaxation follows the implicit solve, the clamp moves behind it
__device__ __forceinline__ bool defer_clamp(const DM &m) { return m.p.smooth_bh_tra || (m.p.clim_relax > 1.0e-8 && !m.p.toy_soufflet); }
__device__ __forceinline__ double tru_clamp(double T, int tr) {           // salinity clamp (oce_ale_tracer.F90:176-198)
  if (tr == 1) { if (T > 45.0) T = 45.0; if (T < 3.0) T = 3.0; }
  return T;
}
// diff_part_bh (smooth_bh_tra, oce_ale_tracer.F90:1081-1150): biharmonic diffusion of the tracer as a filter at the end of diff_tracers_ale, with the
// flow-dependent coefficient of the momentum filters.  Two node gathers over the incident internal edges in edge order (= the reference's scatter order);
// the salinity clamp, which the reference applies after the tracer loop, moves from k_tr_update to the second stage.  grid.y = tracer.
__device__ __forceinline__ bool bh_edge(const DM &m, int ed, int nz, double &vi) {
  if (m.edge_glob[ed] > m.edge2D_in) return false;
  const int e1 = m.edge_tri[2 * ed], e2 = m.edge_tri[2 * ed + 1], n1 = m.edges[2 * ed], n2 = m.edges[2 * ed + 1];
  const int ul1 = min(m.ulev_n_max[n1], m.ulev_n_max[n2]), nl1 = max(m.nlev_n_min[n1], m.nlev_n_min[n2]) - 1;
  if (nz < ul1 || nz > nl1) return false;
  const double len = sqrt(m.elem_area[e1] + m.elem_area[e2]);
  const double u1 = DV2(m.UV, 1, nz, e1) - DV2(m.UV, 1, nz, e2), v1 = DV2(m.UV, 2, nz, e1) - DV2(m.UV, 2, nz, e2);
  vi = u1 * u1 + v1 * v1;
  vi = sqrt(dmax_(m.p.gamma0, dmax_(m.p.gamma1 * sqrt(vi), m.p.gamma2 * vi)) * len);
  return true;
}
// Salt plume parameterization (SPP): cal_rejected_salt + app_rejected_salt (src/oce_spp.F90) at the head of solve_tracers_ale.  One thread per node (owned and
// halo, as the reference): the salt rejected by growing ice leaves the surface layer and is spread over the mixed layer (northern hemisphere, levels above
// the first one with drho/dz >= 0.01 kg/m^4 or below 50 m) with weights area * h * (Z_1 - Z_k)^5 (the integer power as flang's runtime forms it: x * (x^2)^2).
__global__ void k_spp(DM m) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= m.N) return;
  const int nzmin = m.ulev_n[n], nzmax = m.nlev_n[n];
  if (nzmin > 1) return;
  const double th = m.thdgr[n];
  if (!(th > 0.0)) return;
  const double rej = (m.S_oc[n] - m.p.Sice) * th * (910. / 1025. * m.p.dt) * DA2L(m.area, 1, n);
  if (rej <= 0.0) return;
  if (DTR(m.tr_arr, nzmin, n, 1) < 10.0) return;
  if (!(m.geo_lat[n] > 0.0)) return;
  int kml = 1;
  double spar[64], ssum = 0.0;
  spar[nzmin] = 0.0;
  for (int k = nzmin; k <= nzmax && k + 1 <= m.nlm1 && k + 1 < 64; k++) {          // (the search ends at 50 m depth: far above the bottom)
    const double drhodz = DA2L(m.bvfreq, k, n) * D_RHO0 / D_G;
    if (drhodz >= 0.01 || DA2(m.Z_3d_n, k, n) < -50.0) break;
    kml = kml + 1;
    const double x = DA2(m.Z_3d_n, 1, n) - DA2(m.Z_3d_n, k + 1, n), x2 = x * x;
    spar[k + 1] = DA2L(m.area, k + 1, n) * DA2(m.hnode, k + 1, n) * (x * (x2 * x2));
  }
  if (kml > nzmin) {
    DTR(m.tr_arr, nzmin, n, 1) = DTR(m.tr_arr, nzmin, n, 1) - rej / DA2L(m.areasvol, 1, n) / DA2(m.hnode, 1, n);
    for (int k = nzmin + 1; k <= kml; k++) ssum = ssum + spar[k];
    for (int k = nzmin + 1; k <= kml; k++) {
      const double w = spar[k] / ssum;
      DTR(m.tr_arr, k, n, 1) = DTR(m.tr_arr, k, n, 1) + rej * w / DA2L(m.areasvol, k, n) / DA2(m.hnode, k, n);
    }
  }
}
// relax_to_clim (clim_relax > 0, src/oce_tracer_mod.F90:86-121) after diff_tracers_ale: T and S of the owned nodes towards the climatology at the nodal rate
// relax2clim; the salinity clamp follows it as in the reference.  grid.y = tracer (0, 1).
__global__ void __launch_bounds__(BLOCK) k_relax_clim(DM m, int tr0) {
  const int tr = tr0 + blockIdx.y, n = col_id(m), nz = lane_id() + 1;
  if (n >= m.myN || nz < m.ulev_n[n] || nz > m.nlev_n[n] - 1) return;
  const double *cl = tr == 0 ? m.Tclim : m.Sclim;
  double T = DTR(m.tr_arr, nz, n, tr);
  T = T + m.relax2clim[n] * m.p.dt * (DA2(cl, nz, n) - T);
  DTR(m.tr_arr, nz, n, tr) = tru_clamp(T, tr);
}
__global__ void __launch_bounds__(BLOCK) k_bh1(DM m, int tr0) {
  const int tr = tr0 + blockIdx.y, n = col_id(m), nz = lane_id() + 1;
  if (n >= m.myN || nz > m.nlm1) return;
  double tmp = 0.0;
  for (int q = m.ne_ptr[n]; q < m.ne_ptr[n + 1]; q++) {
    const int ed = m.ne_idx[q];
    double vi;
    if (!bh_edge(m, ed, nz, vi)) continue;
    const double tt = (DTR(m.tr_arr, nz, m.edges[2 * ed], tr) - DTR(m.tr_arr, nz, m.edges[2 * ed + 1], tr)) * vi;
    tmp = m.ne_sgn[q] > 0 ? tmp - tt : tmp + tt;
  }
  DTR(m.bh_tmp, nz, n, tr) = tmp;
}
__global__ void __launch_bounds__(BLOCK) k_bh2(DM m, int tr0) {
  const int tr = tr0 + blockIdx.y, n = col_id(m), nz = lane_id() + 1;
  if (n >= m.myN || nz > m.nlm1) return;
  double T = DTR(m.tr_arr, nz, n, tr);
  const double ar = DA2L(m.area, nz, n);
  for (int q = m.ne_ptr[n]; q < m.ne_ptr[n + 1]; q++) {
    const int ed = m.ne_idx[q];
    double vi;
    if (!bh_edge(m, ed, nz, vi)) continue;
    const double tt = -(DTR(m.bh_tmp, nz, m.edges[2 * ed], tr) - DTR(m.bh_tmp, nz, m.edges[2 * ed + 1], tr)) * vi * m.p.dt;
    T = m.ne_sgn[q] > 0 ? T - tt / ar : T + tt / ar;
  }
  if (nz >= m.ulev_n[n] && nz <= m.nlev_n[n] - 1 && !(m.p.clim_relax > 1.0e-8 && !m.p.toy_soufflet)) T = tru_clamp(T, tr);     // (with relax_to_clim the clamp follows that)
  DTR(m.tr_arr, nz, n, tr) = T;
}
template <bool REDI, int NT, int COLS, int WAVES>
__global__ void __launch_bounds__(WAVE * WAVES, (WAVES == 8) ? 4 : 2) k_tr_update(DM m, int tr0) {
  extern __shared__ double th_sh[];
  ThTile<NT, COLS> tile(th_sh, m.nlm1);
  const int trA = tr0 + blockIdx.y * NT;                  // first tracer of this block (grid.y = tracer groups of the launch)
  const int w = threadIdx.x >> 6, nz = lane_id() + 1;
  const int base = xcd_block() * COLS;
  const bool impl = m.p.with_diffusion && m.p.i_vert_diff;
  constexpr bool SINGLE = (COLS == WAVES);                // one column per wave: everything stays in registers across the sweep
  TruCol k;
  double Ts[NT] = {}, rhs[NT] = {};
  for (int ci = w; ci < COLS; ci += WAVES) {
    const int n = __builtin_amdgcn_readfirstlane(sub_col(m, base + ci));
    if (n >= m.myN && n < m.N && nz <= m.nlm1) {           // halo columns: only tr_arr_old(:,:,tr) = tr_arr(:,:,tr) (whole-array copy, :274)
#pragma unroll
      for (int t = 0; t < NT; t++) if (trA + t < m.ntr) DTR(m.tr_arr_old, nz, n, trA + t) = DTR(m.tr_arr, nz, n, trA + t);
    }
    tru_head(m, n, k);
    if (SINGLE) {
      // the horizontal gathers of all tracers first, then the interface depths / slopes of the column, then the tracers' T*: what the gathers keep
      // live (a batch of edge values) never overlaps with the nine column values of tru_zcol (register budget of the two-tracer shape)
      double del[NT];
#pragma unroll
      for (int t = 0; t < NT; t++) {
        Ts[t] = 0.0; rhs[t] = 0.0; del[t] = 0.0;
        if (trA + t < m.ntr) {                                                          // (latency-bound shape: plain divisions keep it at its register budget)
          if (m.p.tra_adv_lim) tru_hor<false, TRU_MAXD, true>(m, k, trA + t, Ts[t], del[t]); else tru_hor<false, TRU_MAXD>(m, k, trA + t, Ts[t], del[t]);
        }
      }
      tru_zcol<REDI>(m, k);
#pragma unroll
      for (int t = 0; t < NT; t++)
        if (trA + t < m.ntr) tru_fin<REDI>(m, k, trA + t, Ts[t], del[t]);
      if (impl) {
        tru_coeffs<REDI>(m, k);
#pragma unroll
        for (int t = 0; t < NT; t++) if (trA + t < m.ntr) rhs[t] = tru_rhs(m, k, trA + t, Ts[t]);
        tile.put(ci, k.valid, k.nzmin, k.nzmax - 1, k.a, k.b, k.c, rhs[0], NT == 2 ? rhs[NT - 1] : 0.0);
      } else {
#pragma unroll
        for (int t = 0; t < NT; t++) if (k.wet && trA + t < m.ntr) DTR(m.tr_arr, nz, k.n, trA + t) = defer_clamp(m) ? Ts[t] : tru_clamp(Ts[t], trA + t);
      }
    } else {
      // several columns per wave: the horizontal gathers of all tracers first (their batches of edge values are the widest live set), then the column's
      // interface depths and coefficients, then tracer after tracer straight into the tile / memory
      double Tt[NT], dl[NT];
#pragma unroll
      for (int t = 0; t < NT; t++) {
        Tt[t] = 0.0; dl[t] = 0.0;
        if (trA + t < m.ntr) {
          if (m.p.tra_adv_lim) tru_hor<false, TRU_MAXD_TILE, true>(m, k, trA + t, Tt[t], dl[t]);
          else if (tru_hor<true, TRU_MAXD_TILE>(m, k, trA + t, Tt[t], dl[t])) tru_hor<false, TRU_MAXD_TILE>(m, k, trA + t, Tt[t], dl[t]);
        }
      }
      tru_zcol<REDI>(m, k);
      if (impl) { tru_coeffs<REDI>(m, k); tile.put_abc(ci, k.valid, k.nzmin, k.nzmax - 1, k.a, k.b, k.c); }
#pragma unroll
      for (int t = 0; t < NT; t++) {
        double T = Tt[t], r = 0.0;
        if (trA + t < m.ntr) {
          tru_fin<REDI>(m, k, trA + t, T, dl[t]);
          if (k.wet) DTR(m.tr_arr, nz, k.n, trA + t) = (impl || defer_clamp(m)) ? T : tru_clamp(T, trA + t);    // T*: picked up again after the sweep
          if (impl) r = tru_rhs(m, k, trA + t, T);
        }
        if (impl) tile.put_rhs(ci, t, r);
      }
    }
  }
  if (!impl) return;
  tile.sweep();
  for (int ci = w; ci < COLS; ci += WAVES) {
    double dT[2];
    tile.get(ci, dT[0], dT[1]);
    int n = k.n; bool wet = k.wet;
    if (!SINGLE) {
      n = __builtin_amdgcn_readfirstlane(sub_col(m, base + ci));
      wet = n < m.myN;
      if (wet) wet = nz >= m.ulev_n[n] && nz <= m.nlev_n[n] - 1;
    }
#pragma unroll
    for (int t = 0; t < NT; t++)
      if (wet && trA + t < m.ntr) {                        // tr_arr = T* + dT ; salinity clamp
        double T = SINGLE ? Ts[t] : DTR(m.tr_arr, nz, n, trA + t);
        DTR(m.tr_arr, nz, n, trA + t) = defer_clamp(m) ? T + dT[t] : tru_clamp(T + dT[t], trA + t);
      }
  }
}

// tr >= 0: that tracer only; tr < 0: all tracers in one launch (grid.y), their chains are independent
#define LAUNCH_COL(k, ncol, m_, tr_) hipLaunchKernelGGL(k, dim3(nblocks(SUBN(m_, ncol)), (tr_) < 0 ? m.ntr : 1), dim3(BLOCK), 0, s, m_, (tr_) < 0 ? 0 : (tr_))
// one column per wave, one tracer per block row (pi) -- or tiles with both tracers per block (DM::use_tile; all tracers of the launch)
#define LAUNCH_TRU1(R, m_, tr_) hipLaunchKernelGGL((k_tr_update<R, 1, TH_COLS, TH_COLS>), dim3(nblocks_th(SUBN(m_, m.N)), (tr_) < 0 ? m.ntr : 1), dim3(TH_BLOCK), (ThTile<1, TH_COLS>::lds_bytes(m.nlm1)), s, m_, (tr_) < 0 ? 0 : (tr_))
#define LAUNCH_TRU2(R, m_) hipLaunchKernelGGL((k_tr_update<R, 2, TH_COLS, TH_COLS>), dim3(nblocks_th(SUBN(m_, m.N)), (m.ntr + 1) / 2), dim3(TH_BLOCK), (ThTile<2, TH_COLS>::lds_bytes(m.nlm1)), s, m_, 0)
#define TRU_SHAPE(id, C_, W_) case id: hipLaunchKernelGGL((k_tr_update<R_, NT_, C_, W_>), dim3((SUBN(m, m.N) + C_ - 1) / C_, gy), dim3(WAVE * W_), (ThTile<NT_, C_>::lds_bytes(m.nlm1)), s, m, tr0); break;
template <bool R_, int NT_> static void launch_tru_tile(const DM &m, hipStream_t s, int gy, int tr0) { switch (m.use_tile) { TILE_SHAPES(TRU_SHAPE) default: break; } }
#define LAUNCH_TRU(m_, tr_) do { if (m.use_tile && (tr_) < 0) { if (m.p.Redi) launch_tru_tile<true, 2>(m_, s, (m.ntr + 1) / 2, 0); else launch_tru_tile<false, 2>(m_, s, (m.ntr + 1) / 2, 0); } \
  else if (m.use_tile) { if (m.p.Redi) launch_tru_tile<true, 1>(m_, s, 1, tr_); else launch_tru_tile<false, 1>(m_, s, 1, tr_); }   /* one named tracer (routine-level tests) */ \
  else if (m.tru_nt2 && (tr_) < 0) { if (m.p.Redi) LAUNCH_TRU2(true, m_); else LAUNCH_TRU2(false, m_); }   /* both tracers of a column in one wave */ \
  else if (m.p.Redi) LAUNCH_TRU1(true, m_, tr_); else LAUNCH_TRU1(false, m_, tr_); } while (0)
#define LAUNCH_WIMPL(m_, tr_) do { if (m.p.w_split) hipLaunchKernelGGL(k_fct_lo_wimpl, dim3(nblocks_th(m.myN), (tr_) < 0 ? m.ntr : 1), dim3(TH_BLOCK), thomas_lds_bytes(m.nlm1, 1), s, m_, (tr_) < 0 ? 0 : (tr_)); } while (0)
// all tracers of a launch (tr_ < 0): two tracers per wave (k##_nt<2>) where that pays -- measured on MI355X (profiles/r03_*): k_flux_hor 14.2 -> 11.9 us on pi,
// 1033 -> 863 us on the channel; k_diff_flux 505 -> 411 us on the channel, unchanged on pi; k_fct_lo_node / k_fct_node do not gain (their waves are bound by
// VALU issue, not by the index chain) and keep one tracer per wave.  FESOM_GPU_EXP_NT = bit mask that overrides the choice (experiments).
static inline bool nt2_on(const DM &m, int tr, int bit) {
  static const int env = getenv("FESOM_GPU_EXP_NT") ? atoi(getenv("FESOM_GPU_EXP_NT")) : -1;
  const bool dflt = bit == 4 ? m.use_tile != 0 : true;      // (bit 4 = k_diff_flux)
  return tr < 0 && m.ntr > 1 && (env >= 0 ? ((env >> bit) & 1) != 0 : dflt);
}
#define LAUNCH_COL_NT(bit, k, ncol, m_, tr_) do { if (nt2_on(m, tr_, bit)) hipLaunchKernelGGL((k##_nt<2>), dim3(nblocks(SUBN(m_, ncol)), (m.ntr + 1) / 2), dim3(BLOCK), 0, s, m_, 0); \
                                                   else LAUNCH_COL(k, ncol, m_, tr_); } while (0)
#define LAUNCH_DFX(m_, tr_) do { if (nt2_on(m, tr_, 4)) { if (m.p.Redi) hipLaunchKernelGGL((k_diff_flux_nt<true, 2>), dim3(nblocks(SUBN(m_, m.myD)), (m.ntr + 1) / 2), dim3(BLOCK), 0, s, m_, 0); \
                                                           else hipLaunchKernelGGL((k_diff_flux_nt<false, 2>), dim3(nblocks(SUBN(m_, m.myD)), (m.ntr + 1) / 2), dim3(BLOCK), 0, s, m_, 0); } \
                                  else if (m.p.Redi) LAUNCH_COL(k_diff_flux<true>, m.myD, m_, tr_); else LAUNCH_COL(k_diff_flux<false>, m.myD, m_, tr_); } while (0)

#define TRU_ATTR(id, C_, W_) (void)hipFuncSetAttribute((const void *)k_tr_update<false, 2, C_, W_>, hipFuncAttributeMaxDynamicSharedMemorySize, big); \
  (void)hipFuncSetAttribute((const void *)k_tr_update<true, 2, C_, W_>, hipFuncAttributeMaxDynamicSharedMemorySize, big); \
  (void)hipFuncSetAttribute((const void *)k_tr_update<false, 1, C_, W_>, hipFuncAttributeMaxDynamicSharedMemorySize, big); \
  (void)hipFuncSetAttribute((const void *)k_tr_update<true, 1, C_, W_>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
void tile_prepare_tra() {      // tiles of meshes with many levels / 64 columns need more than the default 64 KB of dynamic LDS
  const int big = 160 * 1024;
  TILE_SHAPES(TRU_ATTR)
}
void launch_tracer(const DM &m, hipStream_t s, int tr) {   // tr 0-based
  LAUNCH_COL(k_tr_ab, m.N, m, tr);
  LAUNCH_COL(k_tr_z, m.N, m, tr);
  launch_tr_grad_elem(m, s, tr);
  const bool fuse_updn = m.use_tile && tr < 0;             // as the step DAG does (api.hip)
  if (!fuse_updn) LAUNCH_COL(k_updn_grad, m.myD, m, tr);
  if (fuse_updn) launch_flux_hor<true>(m, s, tr); else launch_flux_hor<false>(m, s, tr);
  LAUNCH_COL(k_fct_lo_node, m.myN, m, tr); LAUNCH_WIMPL(m, tr);
  if (!m.p.tra_adv_lim) LAUNCH_COL(k_fct_node, m.myN, m, tr);
  LAUNCH_COL_NT(3, k_fct_edge_limit, m.myD, m, tr);
  if (m.p.with_diffusion) LAUNCH_DFX(m, tr);
  LAUNCH_TRU(m, tr);
  if (m.p.smooth_bh_tra) {
    hipLaunchKernelGGL(k_bh1, dim3(nblocks(m.myN), tr < 0 ? m.ntr : 1), dim3(BLOCK), 0, s, m, tr < 0 ? 0 : tr);
    hipLaunchKernelGGL(k_bh2, dim3(nblocks(m.myN), tr < 0 ? m.ntr : 1), dim3(BLOCK), 0, s, m, tr < 0 ? 0 : tr);
  }
}

int launch_named_tra(const DM &m, hipStream_t s, const char *name, int arg) {
  int tr = arg - 1;
  if (!strncmp(name, "k_", 2)) {
    if (!strcmp(name, "k_tr_ab")) { LAUNCH_COL(k_tr_ab, m.N, m, tr); return 0; }
    if (!strcmp(name, "k_tr_z")) { LAUNCH_COL(k_tr_z, m.N, m, tr); return 0; }
    if (!strcmp(name, "k_tr_grad_elem")) { launch_tr_grad_elem(m, s, tr); return 0; }
    if (!strcmp(name, "k_updn_grad")) { LAUNCH_COL(k_updn_grad, m.myD, m, tr); return 0; }
    if (!strcmp(name, "k_flux_hor")) { launch_flux_hor<false>(m, s, tr); return 0; }
    if (!strcmp(name, "k_flux_hor_fused")) { launch_flux_hor<true>(m, s, tr); return 0; }     // fill_up_dn_grad on the fly
    if (!strcmp(name, "k_fct_lo_node")) { LAUNCH_COL(k_fct_lo_node, m.myN, m, tr); LAUNCH_WIMPL(m, tr); return 0; }   // (+ implicit part with w_split)
    if (!strcmp(name, "k_fct_node")) { if (!m.p.tra_adv_lim) LAUNCH_COL(k_fct_node, m.myN, m, tr); return 0; }      // (no limiter with tra_adv_lim='NON')
    if (!strcmp(name, "k_fct_edge_limit")) { LAUNCH_COL_NT(3, k_fct_edge_limit, m.myD, m, tr); return 0; }
    if (!strcmp(name, "k_diff_flux")) { LAUNCH_DFX(m, tr); return 0; }
    if (!strcmp(name, "k_tr_update")) { LAUNCH_TRU(m, tr); return 0; }
    if (!strcmp(name, "k_spp")) { if (m.p.SPP) hipLaunchKernelGGL(k_spp, dim3((m.N + 127) / 128), dim3(128), 0, s, m); return 0; }
    if (!strcmp(name, "k_bh1")) { hipLaunchKernelGGL(k_bh1, dim3(nblocks(m.myN), tr < 0 ? m.ntr : 1), dim3(BLOCK), 0, s, m, tr < 0 ? 0 : tr); return 0; }
    if (!strcmp(name, "k_bh2")) { hipLaunchKernelGGL(k_bh2, dim3(nblocks(m.myN), tr < 0 ? m.ntr : 1), dim3(BLOCK), 0, s, m, tr < 0 ? 0 : tr); return 0; }
    return -1;
  }
  if (!strcmp(name, "init_tracers_AB")) {
    LAUNCH_COL(k_tr_ab, m.N, m, tr); LAUNCH_COL(k_tr_z, m.N, m, tr); launch_tr_grad_elem(m, s, tr);
    LAUNCH_COL(k_updn_grad, m.myD, m, tr); return 0;
  }
  if (!strcmp(name, "adv_tracers_ale")) {
    launch_flux_hor<false>(m, s, tr); LAUNCH_COL(k_fct_lo_node, m.myN, m, tr); LAUNCH_WIMPL(m, tr);
    if (!m.p.tra_adv_lim) LAUNCH_COL(k_fct_node, m.myN, m, tr);
    LAUNCH_COL_NT(3, k_fct_edge_limit, m.myD, m, tr);
    return 0;
  }
  if (!strcmp(name, "diff_tracers_ale")) {                                                    // incl. flux2dtracer + clamp
    if (m.p.with_diffusion) LAUNCH_DFX(m, tr);
    LAUNCH_TRU(m, tr);
    if (m.p.smooth_bh_tra) {
      hipLaunchKernelGGL(k_bh1, dim3(nblocks(m.myN), tr < 0 ? m.ntr : 1), dim3(BLOCK), 0, s, m, tr < 0 ? 0 : tr);
      hipLaunchKernelGGL(k_bh2, dim3(nblocks(m.myN), tr < 0 ? m.ntr : 1), dim3(BLOCK), 0, s, m, tr < 0 ? 0 : tr);
    }
    return 0;
  }
  if (!strcmp(name, "relax_to_clim")) {
    if (m.p.clim_relax > 1.0e-8 && !m.p.toy_soufflet && tr < 2)           // tr < 0: T and S in one launch
      hipLaunchKernelGGL(k_relax_clim, dim3(nblocks(m.myN), tr < 0 ? (m.ntr < 2 ? m.ntr : 2) : 1), dim3(BLOCK), 0, s, m, tr < 0 ? 0 : tr);
    return 0;
  }
  if (!strcmp(name, "spp")) { if (m.p.SPP) hipLaunchKernelGGL(k_spp, dim3((m.N + 127) / 128), dim3(128), 0, s, m); return 0; }
  if (!strcmp(name, "salinity_clamp")) return 0;
  return -1;
}
