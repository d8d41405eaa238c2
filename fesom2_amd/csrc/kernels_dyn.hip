// Dynamics kernels of the ocean step (gfx950).  One wavefront = one vertical column, lane = level.
// Every kernel cites the reference routine it replaces; arithmetic order follows the reference
// (checked bitwise against the CPU oracle in tests/).
#include "dev.h"
#include <string.h>

// ------------------------------------------------------------------------------------------------
// compute_vel_nodes (src/oce_dyn.F90:133-169): node <- area-weighted mean of surrounding elements.
// HBM-bound gather; algorithmic traffic 2 N3 + 2 E3 values.
__global__ void __launch_bounds__(BLOCK) k_vel_nodes(DM m) {
  int n = col_id(m), nz = lane_id() + 1;
  if (n >= m.myN) return;
  if (nz < m.ulev_n[n] || nz > m.nlev_n[n] - 1) return;
  double tvol = 0.0, tx = 0.0, ty = 0.0;
  int num = m.nie_num[n];
  if (m.exp_batch & 4) {
    // the velocities of the elements around the node in batches of independent loads (an element that does not reach this level is read inside its
    // column all the same and dropped in the select), the sums in the reference's element order
    constexpr int VB = 6;
    for (int k0 = 0; k0 < num; k0 += VB) {
      double u[VB], v[VB], a[VB]; bool on[VB];
#pragma unroll
      for (int j = 0; j < VB; j++) {
        const int e = m.nie[(size_t)m.maxk * n + (k0 + j < num ? k0 + j : 0)];
        on[j] = k0 + j < num && !(m.nlev[e] - 1 < nz || nz < m.ulev[e]);
        a[j] = m.elem_area[e];
        u[j] = DV2(m.UV, 1, nz, e); v[j] = DV2(m.UV, 2, nz, e);
      }
#pragma unroll
      for (int j = 0; j < VB; j++) {
        const double nv = tvol + a[j], nx = tx + u[j] * a[j], ny = ty + v[j] * a[j];
        tvol = on[j] ? nv : tvol; tx = on[j] ? nx : tx; ty = on[j] ? ny : ty;
      }
    }
  } else
  for (int k = 0; k < num; k++) {
    int e = m.nie[(size_t)m.maxk * n + k];
    if (m.nlev[e] - 1 < nz || nz < m.ulev[e]) continue;
    double a = m.elem_area[e];
    tvol = tvol + a;
    tx = tx + DV2(m.UV, 1, nz, e) * a;
    ty = ty + DV2(m.UV, 2, nz, e) * a;
  }
  DV2(m.Unode, 1, nz, n) = tx / tvol;
  DV2(m.Unode, 2, nz, n) = ty / tvol;
}

// ------------------------------------------------------------------------------------------------
// Equation of state: densityJM_components (src/oce_ale_pressure_bv.F90:2586-2654), density_linear (:2992-3019)
__device__ __forceinline__ void eos(const fesom_params &p, double t, double s, double &bulk_0, double &bulk_pz, double &bulk_pz2,
                                    double &rhopot) {
  if (p.state_equation == 0) {
    bulk_0 = 1; bulk_pz = 0; bulk_pz2 = 0;
    if (p.toy_soufflet) rhopot = D_RHO0 - 0.00025 * (t - 10.0) * D_RHO0;
    else rhopot = D_RHO0 + 0.8 * (s - 34.0) - 0.2 * (t - 20.0);
    return;
  }
  const double a0 = 19092.56, at = 209.8925, at2 = -3.041638, at3 = -1.852732e-3, at4 = -1.361629e-5;
  const double as = 104.4077, ast = -6.500517, ast2 = .1553190, ast3 = 2.326469e-4;
  const double ass = -5.587545, asst = 0.7390729, asst2 = -1.909078e-2;
  const double ap = -4.721788e-1, apt = -1.028859e-2, apt2 = 2.512549e-4, apt3 = 5.939910e-7;
  const double aps = 1.571896e-2, apst = 2.598241e-4, apst2 = -7.267926e-6, apss = -2.042967e-3;
  const double ap2 = 1.045941e-5, ap2t = -5.782165e-10, ap2t2 = 1.296821e-7;
  const double ap2s = -2.595994e-7, ap2st = -1.248266e-9, ap2st2 = -3.508914e-9;
  const double b0 = 999.842594, bt = 6.793952e-2, bt2 = -9.095290e-3, bt3 = 1.001685e-4, bt4 = -1.120083e-6, bt5 = 6.536332e-9;
  const double bs = 0.824493, bst = -4.08990e-3, bst2 = 7.64380e-5, bst3 = -8.24670e-7, bst4 = 5.38750e-9;
  const double bss = -5.72466e-3, bsst = 1.02270e-4, bsst2 = -1.65460e-6, bss2 = 4.8314e-4;
  double s_sqrt = sqrt(s);
  bulk_0 = a0 + t * (at + t * (at2 + t * (at3 + t * at4))) +
           s * (as + t * (ast + t * (ast2 + t * ast3)) + s_sqrt * (ass + t * (asst + t * asst2)));
  bulk_pz = ap + t * (apt + t * (apt2 + t * apt3)) + s * (aps + t * (apst + t * apst2) + s_sqrt * apss);
  bulk_pz2 = ap2 + t * (ap2t + t * ap2t2) + s * (ap2s + t * (ap2st + t * ap2st2));
  rhopot = b0 + t * (bt + t * (bt2 + t * (bt3 + t * (bt4 + t * bt5)))) +
           s * (bs + t * (bst + t * (bst2 + t * (bst3 + t * bst4))) + s_sqrt * (bss + t * (bsst + t * bsst2)) + s * bss2);
}

// pressure_bv (src/oce_ale_pressure_bv.F90:106-365) fused with sw_alpha_beta (:2736-2821): both are
// pointwise in (T,S,Z) per node column.  Vertical neighbours come from wave shuffles; MLD searches are
// ballots.  ~150 flops/cell, 7+5 values/cell -> HBM-bound.
// init_ref_density (src/oce_ale_pressure_bv.F90:3024-3070; ocean_setup, once): the reference density profile from (density_ref_T, density_ref_S) at the
// depths of the initial Z_3d_n; from level 1 also under an ice shelf, as the reference has it (nzmin = 1)
__global__ void __launch_bounds__(BLOCK) k_init_density_ref(DM m) {
  const int n = col_id(m), nz = lane_id() + 1;
  if (n >= m.N || nz > m.nlm1) return;
  double b0, bpz, bpz2, rp;
  eos(m.p, m.p.density_ref_T, m.p.density_ref_S, b0, bpz, bpz2, rp);
  double out = 0.0;
  if (nz <= m.nlev_n[n] - 1) {
    double auxz = DA2(m.Z_3d_n, nz, n);
    if (nz == 1) auxz = auxz < 0.0 ? auxz : 0.0;
    const double rho = b0 + auxz * bpz + auxz * bpz2;
    out = rho * rp / (rho + 0.1 * auxz);
  }
  DA2(m.density_ref, nz, n) = out;
}
void launch_init_density_ref(const DM &m, hipStream_t s) { hipLaunchKernelGGL(k_init_density_ref, dim3(nblocks(m.N)), dim3(BLOCK), 0, s, m); }
__global__ void __launch_bounds__(BLOCK) k_pressure_bv(DM m) {
  int n = col_id(m), l = lane_id(), nz = l + 1;
  if (n >= m.N) return;
  const int nzmin = m.ulev_n[n], nzmax = m.nlev_n[n];
  const bool wet = (nz >= nzmin && nz <= nzmax - 1);
  const double seq = (double)m.p.state_equation;
  double t = 0.0, s = 0.0, z = 0.0, b0 = 0.0, bpz = 0.0, bpz2 = 0.0, rpot = 0.0, rho = 0.0;
  if (wet) { t = DTR(m.tr_arr, nz, n, 0); s = DTR(m.tr_arr, nz, n, 1); z = DA2(m.Z_3d_n, nz, n); eos(m.p, t, s, b0, bpz, bpz2, rpot); }
  const double b0s = bcast(b0, nzmin - 1), bpzs = bcast(bpz, nzmin - 1), bpz2s = bcast(bpz2, nzmin - 1), rpots = bcast(rpot, nzmin - 1);
  const double zmin = bcast(z, nzmin - 1);
  double dbq = 0.0;
  const double z2 = bcast(z, nzmin);          // Z_3d_n(nzmin+1)
  // density_ref: density_0, or the profile of init_ref_density (use_density_ref; with cavities always, src/oce_setup_step.F90:120-129)
  const double dref = (m.density_ref && nz <= m.nlm1) ? DA2(m.density_ref, nz, n) : D_RHO0;
  if (nzmin > 1 && nz < nzmin) {      // :235-258 the levels the ice shelf occupies take the density of the water mass at the cavity-ocean interface
    z = DA2(m.Z_3d_n, nz, n);
    b0 = b0s; bpz = bpzs; bpz2 = bpz2s; rpot = rpots;
    rho = b0 + z * (bpz + z * bpz2);
    rho = rho * rpot / (rho + 0.1 * z * seq) - dref;
    DA2(m.density_m_rho0, nz, n) = rho;
  }
  if (wet) {
    rho = b0 + z * (bpz + z * bpz2);
    rho = rho * rpot / (rho + 0.1 * z * seq) - dref;
    DA2(m.density_m_rho0, nz, n) = rho;
    double rho_surf = b0s + z * (bpzs + z * bpz2s);
    rho_surf = rho_surf * rpots / (rho_surf + 0.1 * z * seq);
    double rr = rho + dref;
    double dbsfc1 = -D_G * (rho_surf - rr) / rr;
    double zk = (nz > nzmin + 1) ? z : z2;                      // Z_3d_n(max(nz,nzmin+1))
    dbq = dbsfc1 / fabs(zmin - zk);
    if (m.dbsfc) { DA2L(m.dbsfc, nz, n) = dbsfc1; if (nz == nzmax - 1) DA2L(m.dbsfc, nzmax, n) = dbsfc1; }   // KPP: buoyancy difference to the surface
  }
  if (m.pgf_A) {
    // pressure_force_4_zxxxx_shchepetkin (src/oce_ale_pressure_bv.F90:1878-2104): drho/dz at a node of an element is
    //   df10/dx10 + (dx10*df21 - dx21*df10)/(dx20*dx21*dx10) * ((Z_n - zc) + (Z_n - zm))
    // around a level k0 of the NODE column; the two quotients do not depend on the element.  They are formed here, once per node and level, with the
    // operands and operations of the reference (the same bits as when every element forms them again: six fp64 divisions per element and level saved)
    const double zm = shup(z), zp = shdn(z), rm = shup(rho), rp = shdn(rho);
    double qa = 0.0, qb = 0.0;
    if (wet && nz - 1 >= nzmin && nz + 1 <= nzmax - 1) {
      const double dx10 = z - zm, dx21 = zp - z, dx20 = zp - zm, df10 = rho - rm, df21 = rp - rho;
      qa = df10 / dx10;
      qb = (dx10 * df21 - dx21 * df10) / (dx20 * dx21 * dx10);
    }
    if (nz <= m.nlm1) { DA2(m.pgf_A, nz, n) = qa; DA2(m.pgf_B, nz, n) = qb; }
  }
  double db_max = wave_max(wet ? dmax_(dbq, 0.0) : 0.0);
  // linfs: hydrostatic pressure (sequential running sum, reference order)
  if (m.p.which_ale == 0 || m.p.use_cavity) {      // :262
    double hn = wet ? DA2(m.hnode, nz, n) : 0.0;
    double rh = rho * hn;                       // rho(nz)*hnode(nz)
    double rh_up = shup(rh);
    double a = (wet && nz > nzmin) ? 0.5 * D_G * (rh_up + rh) : 0.0;
    double h0 = -zmin * bcast(rho, nzmin - 1) * D_G;
    if (nzmin > 1) {      // :268-275 pressure at the cavity-ocean interface: the column of interface water above it, top-down
      const double zb = (nz <= nzmin + 1) ? DA2L(m.zbar_3d_n, nz, n) : 0.0, zb_dn = shdn(zb), zb_up = shup(zb), rho_up = shup(rho);
      const double ac = (nz >= 2 && nz <= nzmin) ? 0.5 * D_G * (rho_up * (zb_up - zb) + rho * (zb - zb_dn)) : 0.0;
      const double hs = 0.5 * (bcast(zb, 0) - bcast(zb, 1)) * bcast(rho, 0) * D_G;
      h0 = bcast(seq_sum_up(ac, 1, nzmin - 1, hs), nzmin - 1);
    }
    double hp = seq_sum_up(a, nzmin, nzmax - 2, h0);            // lanes nzmin..nzmax-2 <-> levels nzmin+1..nzmax-1
    if (wet) DA2L(m.hpressure, nz, n) = (nz == nzmin) ? h0 : hp;
  }
  // N^2 at interfaces nz = nzmin+1..nzmax-1 (needs layer nz-1 -> shuffle up)
  double b0u = shup(b0), bpzu = shup(bpz), bpz2u = shup(bpz2), rpotu = shup(rpot), zu = shup(z);
  double bv = 0.0;
  const bool inner = (nz >= nzmin + 1 && nz <= nzmax - 1);
  if (inner) {
    double zb = DA2L(m.zbar_3d_n, nz, n);
    double bulk_up = b0u + zb * (bpzu + zb * bpz2u);
    double bulk_dn = b0 + zb * (bpz + zb * bpz2);
    double rho_up = bulk_up * rpotu / (bulk_up + 0.1 * zb * seq);
    double rho_dn = bulk_dn * rpot / (bulk_dn + 0.1 * zb * seq);
    double dz_inv = 1.0 / (zu - z);
    bv = -D_G * dz_inv * (rho_up - rho_dn) / D_RHO0;
    DA2L(m.bvfreq, nz, n) = bv;
  }
  double bv_first = bcast(bv, nzmin), bv_last = bcast(bv, nzmax - 2);
  if (nz == nzmin) DA2L(m.bvfreq, nzmin, n) = bv_first;
  if (nz == nzmax) DA2L(m.bvfreq, nzmax, n) = bv_last;
  // mixed layer depths
  unsigned long long b1 = __ballot(inner && bv > db_max);
  double zsel = z;
  int i1 = b1 ? (__ffsll((long long)b1) - 1) : nzmin;          // default: Z_3d_n(nzmin+1) = lane nzmin
  double mld1 = bcast(zsel, i1);
  unsigned long long b2 = __ballot(inner && (rpot - rpots > 0.125));
  double rp1 = bcast(rpot, 0);                                  // reference quirk: rhopot(1), not rhopot(nzmin)
  double mld2;
  if (b2) {
    int i2 = __ffsll((long long)b2) - 1;                        // lane of first level beyond the threshold
    double prev = (i2 == nzmin) ? bcast(z, nzmin) : bcast(z, i2 - 1);   // MLD2 before the hit
    double zi = bcast(z, i2), ri = bcast(rpot, i2), rim = bcast(rpot, i2 - 1);
    mld2 = prev + (zi - prev) / (ri - rim + 1.e-20) * (rp1 + 0.125 - rim);
  } else {
    mld2 = bcast(z, (nzmax - 2 > nzmin) ? nzmax - 2 : nzmin);   // last Z visited (or the initial value)
  }
  if (l == 0) { m.MLD1[n] = mld1; m.MLD2[n] = mld2; if (m.MLD1_ind) m.MLD1_ind[n] = i1 + 1; }
  // sw_alpha_beta (owned nodes)
  if (wet && n < m.myN) {
    double t1 = t * 1.00024, s1 = s, p1 = fabs(z);
    double t1_2 = t1 * t1, t1_3 = t1_2 * t1, t1_4 = t1_3 * t1, p1_2 = p1 * p1, p1_3 = p1_2 * p1;
    double s35 = s1 - 35.0, s35_2 = s35 * s35;
    double beta = 0.785567e-3 - 0.301985e-5 * t1 + 0.555579e-7 * t1_2 - 0.415613e-9 * t1_3 +
                  s35 * (-0.356603e-6 + 0.788212e-8 * t1 + 0.408195e-10 * p1 - 0.602281e-15 * p1_2) + s35_2 * (0.515032e-8) +
                  p1 * (-0.121555e-7 + 0.192867e-9 * t1 - 0.213127e-11 * t1_2) + p1_2 * (0.176621e-12 - 0.175379e-14 * t1) +
                  p1_3 * (0.121551e-17);
    double a_over_b = 0.665157e-1 + 0.170907e-1 * t1 - 0.203814e-3 * t1_2 + 0.298357e-5 * t1_3 - 0.255019e-7 * t1_4 +
                      s35 * (0.378110e-2 - 0.846960e-4 * t1 - 0.164759e-6 * p1 - 0.251520e-11 * p1_2) + s35_2 * (-0.678662e-5) +
                      p1 * (0.380374e-4 - 0.933746e-6 * t1 + 0.791325e-8 * t1_2) + p1_2 * t1_2 * (0.512857e-12) -
                      p1_3 * (0.302285e-13);
    DA2(m.sw_beta, nz, n) = beta;
    DA2(m.sw_alpha, nz, n) = a_over_b * beta;
  }
}

// pressure_force_4_zxxxx_cubicspline (src/oce_ale_pressure_bv.F90:1697-1866): density of one node interpolated to the depth Zn with the
// monotonised cubic spline of the four levels around it (surface / bottom / bulk cases :1757-1803)
template <bool LINFS_BOTTOM>          // pressure_force_4_linfs_cubicspline (:1365-1413) always takes the bottom-case stencil
__device__ __forceinline__ double pgf_cubic_rho(const DM &m, int node, double Zn) {
  const int nln = m.nlev_n[node] - 1, uln = m.ulev_n[node];
  int nlc = nln - 1;
  for (int dd = uln; dd <= nln; dd++)
    if (DA2(m.Z_3d_n, dd, node) <= Zn) { nlc = dd - 1; if (dd == 1) nlc = 1; break; }
  int i0 = nlc - 1, i3 = nlc + 2;
  const bool surf = !LINFS_BOTTOM && nlc == uln, bot = LINFS_BOTTOM || (!surf && nlc == nln - 1);
  if (surf) i0 = uln;
  if (bot) i3 = nlc + 1;
  const double z0 = DA2(m.Z_3d_n, i0, node), z1 = DA2(m.Z_3d_n, nlc, node), z2 = DA2(m.Z_3d_n, nlc + 1, node), z3 = DA2(m.Z_3d_n, i3, node);
  const double d0 = DA2(m.density_m_rho0, i0, node), d1 = DA2(m.density_m_rho0, nlc, node), d2 = DA2(m.density_m_rho0, nlc + 1, node),
               d3 = DA2(m.density_m_rho0, i3, node);
  const double s_H = z2 - z1, aux1 = (d2 - d1) / s_H;
  double s_dup, s_dlo, aux2;
  if (surf) {
    aux2 = (d3 - d2) / (z3 - z2);
    s_dlo = 0.0;
    if (aux1 * aux2 > 0.) s_dlo = 2.0 * aux1 * aux2 / (aux1 + aux2);
    s_dup = 1.5 * aux1 - 0.5 * s_dlo;
  } else if (bot) {
    aux2 = (d1 - d0) / (z1 - z0);
    s_dup = 0.0;
    if (aux1 * aux2 > 0.) s_dup = 2.0 * aux1 * aux2 / (aux1 + aux2);
    s_dlo = 1.5 * aux1 - 0.5 * s_dup;
  } else {
    aux2 = (d1 - d0) / (z1 - z0);
    s_dup = 0.0;
    if (aux1 * aux2 > 0.) s_dup = 2.0 * aux1 * aux2 / (aux1 + aux2);
    aux2 = (d3 - d2) / (z3 - z2);
    s_dlo = 0.0;
    if (aux1 * aux2 > 0.) s_dlo = 2.0 * aux1 * aux2 / (aux1 + aux2);
  }
  const double c = -(2.0 * s_dup + s_dlo) / s_H + 3.0 * (d2 - d1) / (s_H * s_H);
  const double d = (s_dup + s_dlo) / (s_H * s_H) - 2.0 * (d2 - d1) / ((s_H * s_H) * s_H);
  const double dz = Zn - z1;
  return d1 + s_dup * dz + c * (dz * dz) + d * ((dz * dz) * dz);
}

// ------------------------------------------------------------------------------------------------
// pressure_force_4_zxxxx_shchepetkin (src/oce_ale_pressure_bv.F90:1878-2104) / _zxxxx_cubicspline (:1697-1866) / _linfs_fullcell (:432-466) /
// _linfs_shchepetkin (:647-891).
// Per element column: density-Jacobian terms per level in parallel, the two vertical integrals as
// reference-order running sums.  Reads 3 node columns x (rho, Z) -> HBM-bound gather, 2 N3 + 3 E3 values.
__global__ void __launch_bounds__(BLOCK) k_pgf(DM m) {
  int e = col_id(m), l = lane_id(), nlz = l + 1;
  if (e >= m.myE) return;
  const int nle = m.nlev[e] - 1, ule = m.ulev[e];
  const bool wet = (nlz >= ule && nlz <= nle);
  const int n0 = m.elem_nodes[3 * e], n1 = m.elem_nodes[3 * e + 1], n2 = m.elem_nodes[3 * e + 2];
  const bool cav_pc = m.p.use_cavity && m.p.use_cavity_partial_cell;      // linfs: the full-cell gradient only without partial cells at the bottom AND at the shelf base (:385)
  if (m.p.which_ale == 0 && !m.p.use_partial_cell && !cav_pc) {
    if (wet) {
      DA2(m.pgf_x, nlz, e) = DGS(1, e) * DA2L(m.hpressure, nlz, n0) / D_RHO0 + DGS(2, e) * DA2L(m.hpressure, nlz, n1) / D_RHO0 +
                             DGS(3, e) * DA2L(m.hpressure, nlz, n2) / D_RHO0;
      DA2(m.pgf_y, nlz, e) = DGS(4, e) * DA2L(m.hpressure, nlz, n0) / D_RHO0 + DGS(5, e) * DA2L(m.hpressure, nlz, n1) / D_RHO0 +
                             DGS(6, e) * DA2L(m.hpressure, nlz, n2) / D_RHO0;
    }
    return;
  }
  if (m.p.which_ale == 0 && m.p.which_pgf == 2) {         // 'nemo' (pressure_force_4_linfs_nemo :479-635): hydrostatic pressure above the bottom layer; in the bottom
    if (!wet) return;                                     // layer T, S interpolated to the shallowest bottom mid-depth of the three nodes, density there, thinnest layer
    if (nlz < nle) {
      DA2(m.pgf_x, nlz, e) = (DGS(1, e) * DA2L(m.hpressure, nlz, n0) / D_RHO0 + DGS(2, e) * DA2L(m.hpressure, nlz, n1) / D_RHO0) + DGS(3, e) * DA2L(m.hpressure, nlz, n2) / D_RHO0;
      DA2(m.pgf_y, nlz, e) = (DGS(4, e) * DA2L(m.hpressure, nlz, n0) / D_RHO0 + DGS(5, e) * DA2L(m.hpressure, nlz, n1) / D_RHO0) + DGS(6, e) * DA2L(m.hpressure, nlz, n2) / D_RHO0;
      return;
    }
    const int en[3] = {n0, n1, n2};
    const double Zn = m.zbar_e_bot[e] + DA2(m.helem, nle, e) / 2.0, seq = (double)m.p.state_equation;
    double zmax = DA2(m.Z_3d_n, nle, n0), dh = DA2(m.hnode, nle, n0);
    for (int k = 1; k < 3; k++) { zmax = dmax_(zmax, DA2(m.Z_3d_n, nle, en[k])); dh = dmin_(dh, DA2(m.hnode, nle, en[k])); }
    double hpb[3];
    for (int ni = 0; ni < 3; ni++) {
      const int n = en[ni], nln = m.nlev_n[n] - 1, uln = m.ulev_n[n];
      int pos = 0; double best = 0.0;                     // minloc of the positive differences (first minimum)
      for (int k = uln; k <= nln; k++) {
        const double dd = DA2(m.Z_3d_n, k, n) - zmax;
        if (dd > 0.0 && (pos == 0 || dd < best)) { pos = k - uln + 1; best = dd; }
      }
      int nlc = pos + 1;
      if (nlc > nln) nlc = nln;
      const double dZn = DA2(m.Z_3d_n, nlc, n) - DA2(m.Z_3d_n, nlc - 1, n), dZn_i = zmax - DA2(m.Z_3d_n, nlc - 1, n);
      double dval = DTR(m.tr_arr, nlc, n, 0) - DTR(m.tr_arr, nlc - 1, n, 0);
      const double ti = DTR(m.tr_arr, nlc - 1, n, 0) + (dval / dZn * dZn_i);
      dval = DTR(m.tr_arr, nlc, n, 1) - DTR(m.tr_arr, nlc - 1, n, 1);
      const double si = DTR(m.tr_arr, nlc - 1, n, 1) + (dval / dZn * dZn_i);
      double b0, bpz, bpz2, rp;
      eos(m.p, ti, si, b0, bpz, bpz2, rp);
      double dens = b0 + Zn * (bpz + Zn * bpz2);
      dens = dens * rp / (dens + 0.1 * Zn * seq) - (m.density_ref ? DA2(m.density_ref, nle, n) : D_RHO0);          // density_ref(nle, node), :611
      const int nlce = nlc < nle ? nlc : nle;
      hpb[ni] = DA2L(m.hpressure, nlce - 1, n) + 0.5 * D_G * (DA2(m.density_m_rho0, nlce - 1, n) * DA2(m.hnode, nlce - 1, n) + dens * dh);
    }
    DA2(m.pgf_x, nle, e) = ((DGS(1, e) * hpb[0] + DGS(2, e) * hpb[1]) + DGS(3, e) * hpb[2]) / D_RHO0;
    DA2(m.pgf_y, nle, e) = ((DGS(4, e) * hpb[0] + DGS(5, e) * hpb[1]) + DGS(6, e) * hpb[2]) / D_RHO0;
    return;
  }
  double he = wet ? DA2(m.helem, nlz, e) : 0.0;
  // zbar_n(nlz) = zbar_e_bot + sum_{k=nle..nlz} helem(k)  (bottom-up, reference order); lane l <-> level l+1
  double zb_top = seq_sum_down(he, nle - 1, ule - 1, m.zbar_e_bot[e]);   // zbar_n(nlz)
  double zb_bot = shdn(zb_top);                                          // zbar_n(nlz+1)
  if (nlz == nle) zb_bot = m.zbar_e_bot[e];
  double Zn = zb_bot + he * 0.5;                                         // Z_n(nlz)
  double auxx = 0.0, auxy = 0.0, p0x = 0.0, p0y = 0.0;
  if (wet && m.p.which_pgf == 1 && m.p.which_ale == 0) {   // 'cubicspline', linfs with partial cells (:1252-1444): flat layers, spline in the bottom layer
    double r0, r1, r2;
    if (nlz == nle && nle > ule) { r0 = pgf_cubic_rho<true>(m, n0, Zn); r1 = pgf_cubic_rho<true>(m, n1, Zn); r2 = pgf_cubic_rho<true>(m, n2, Zn); }
    else { r0 = DA2(m.density_m_rho0, nlz, n0); r1 = DA2(m.density_m_rho0, nlz, n1); r2 = DA2(m.density_m_rho0, nlz, n2); }
    const double gx = (DGS(1, e) * r0 + DGS(2, e) * r1) + DGS(3, e) * r2, gy = (DGS(4, e) * r0 + DGS(5, e) * r1) + DGS(6, e) * r2;
    auxx = gx * he * D_G / D_RHO0; auxy = gy * he * D_G / D_RHO0;
    if (nlz == ule && ule > 1) { p0x = gx * (-m.zbar_e_srf[e]) * D_G / D_RHO0; p0y = gy * (-m.zbar_e_srf[e]) * D_G / D_RHO0; }      // pressure boundary condition at the shelf base (:1316-1335)
  } else if (wet && m.p.which_pgf == 1) {                  // 'cubicspline', zstar (:1697-1866)
    const double r0 = pgf_cubic_rho<false>(m, n0, Zn), r1 = pgf_cubic_rho<false>(m, n1, Zn), r2 = pgf_cubic_rho<false>(m, n2, Zn);
    const double gx = (DGS(1, e) * r0 + DGS(2, e) * r1) + DGS(3, e) * r2, gy = (DGS(4, e) * r0 + DGS(5, e) * r1) + DGS(6, e) * r2;
    auxx = D_G * he * gx / D_RHO0; auxy = D_G * he * gy / D_RHO0;
  } else if (wet && m.p.which_pgf == 3) {                  // 'easypgf', zstar (:2116-2546) and linfs with partial cells (:898-1245): T, S interpolated to Z_n with the Newton polynomial of three levels, density there
    const int en[3] = {n0, n1, n2};
    const double seq = (double)m.p.state_equation;
    double r3[3];
#pragma unroll
    for (int ni = 0; ni < 3; ni++) {
      int n = en[ni], k0;
      if (m.p.which_ale == 0 && nlz != nle && !(nlz == ule && ule > 1)) { r3[ni] = DA2(m.density_m_rho0, nlz, n); continue; }      // pressure_force_4_linfs_easypgf (:898-1245): flat above the bottom layer, except directly under a shelf
      if (nlz == ule && (nlz - m.ulev_n[n]) == 0) k0 = nlz + 1;
      else if (nlz == nle && nlz != ule && (m.nlev_n[n] - 1 - nlz) == 0) k0 = nlz - 1;
      else k0 = nlz;
      const double zm = DA2(m.Z_3d_n, k0 - 1, n), zc = DA2(m.Z_3d_n, k0, n), zp = DA2(m.Z_3d_n, k0 + 1, n);
      const double dx10 = zc - zm, dx21 = zp - zc, dx20 = zp - zm;
      double ts[2];
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const double x0 = DTR(m.tr_arr, k0 - 1, n, t), d10 = DTR(m.tr_arr, k0, n, t) - x0, d21 = DTR(m.tr_arr, k0 + 1, n, t) - DTR(m.tr_arr, k0, n, t);
        ts[t] = x0 + d10 / dx10 * (Zn - zm) + (dx10 * d21 - dx21 * d10) / (dx20 * dx21 * dx10) * (Zn - zc) * (Zn - zm);
      }
      double b0, bpz, bpz2, rp;
      eos(m.p, ts[0], ts[1], b0, bpz, bpz2, rp);
      const double rho = b0 + Zn * (bpz + Zn * bpz2);
      r3[ni] = rho * rp / (rho + 0.1 * Zn * seq) - D_RHO0;
    }
    const double gx = (DGS(1, e) * r3[0] + DGS(2, e) * r3[1]) + DGS(3, e) * r3[2], gy = (DGS(4, e) * r3[0] + DGS(5, e) * r3[1]) + DGS(6, e) * r3[2];
    auxx = gx * he * D_G / D_RHO0; auxy = gy * he * D_G / D_RHO0;
  } else if (wet) {
    const int en[3] = {n0, n1, n2};
    double drho_dz[3], rho_c[3], z_c[3];
#pragma unroll
    for (int ni = 0; ni < 3; ni++) {
      int n = en[ni], k0;
      if (nlz == ule && (nlz - m.ulev_n[n]) == 0) k0 = nlz + 1;
      else if (nlz == nle && nlz != ule && (m.nlev_n[n] - 1 - nlz) == 0) k0 = nlz - 1;
      else k0 = nlz;
      const double zm = DA2(m.Z_3d_n, k0 - 1, n), zc = DA2(m.Z_3d_n, k0, n);
      drho_dz[ni] = DA2(m.pgf_A, k0, n) + DA2(m.pgf_B, k0, n) * ((Zn - zc) + (Zn - zm));      // (the quotients: k_pressure_bv)
      rho_c[ni] = DA2(m.density_m_rho0, nlz, n);
      z_c[ni] = (k0 == nlz) ? zc : DA2(m.Z_3d_n, nlz, n);
    }
    double s3 = (drho_dz[0] + drho_dz[1] + drho_dz[2]) / 3.0;
    double drho_dx = DGS(1, e) * rho_c[0] + DGS(2, e) * rho_c[1] + DGS(3, e) * rho_c[2];
    double dz_dx = DGS(1, e) * z_c[0] + DGS(2, e) * z_c[1] + DGS(3, e) * z_c[2];
    // pressure_force_4_linfs_shchepetkin (:647-891, linfs with partial cells): the Jacobian correction only in the bottom layer
    const bool flat = m.p.which_ale == 0 && nlz != nle && !(nlz == ule && ule > 1);      // (directly under a shelf the correction stays, :703-776)
    auxx = flat ? drho_dx * he * D_G / D_RHO0 : (drho_dx - s3 * dz_dx) * he * D_G / D_RHO0;
    double drho_dy = DGS(4, e) * rho_c[0] + DGS(5, e) * rho_c[1] + DGS(6, e) * rho_c[2];
    double dz_dy = DGS(4, e) * z_c[0] + DGS(5, e) * z_c[1] + DGS(6, e) * z_c[2];
    auxy = flat ? drho_dy * he * D_G / D_RHO0 : (drho_dy - s3 * dz_dy) * he * D_G / D_RHO0;
  }
  if (m.p.which_pgf == 4) {
    // 'sergey' = pressure_force_4_linfs_cavity (:1451-1663; linfs with partial cells at the shelf base): the gradient of the hydrostatic pressure, except
    // directly under a shelf (half the density-Jacobian term of that layer) and, with partial cells, in the bottom layer (pressure at its upper face + half the term)
    if (!wet) return;
    const bool top = nlz == ule && ule > 1, bot = nlz == nle && m.p.use_partial_cell;
    double px, py;
    if (top) { px = auxx * 0.5; py = auxy * 0.5; }
    else if (bot) {
      const double h0 = (DA2L(m.hpressure, nlz - 1, n0) + 0.5 * D_G * (DA2(m.density_m_rho0, nlz - 1, n0) * DA2(m.hnode, nlz - 1, n0))),
                   h1 = (DA2L(m.hpressure, nlz - 1, n1) + 0.5 * D_G * (DA2(m.density_m_rho0, nlz - 1, n1) * DA2(m.hnode, nlz - 1, n1))),
                   h2 = (DA2L(m.hpressure, nlz - 1, n2) + 0.5 * D_G * (DA2(m.density_m_rho0, nlz - 1, n2) * DA2(m.hnode, nlz - 1, n2)));
      px = (DGS(1, e) * h0 / D_RHO0 + DGS(2, e) * h1 / D_RHO0 + DGS(3, e) * h2 / D_RHO0) + auxx * 0.5;
      py = (DGS(4, e) * h0 / D_RHO0 + DGS(5, e) * h1 / D_RHO0 + DGS(6, e) * h2 / D_RHO0) + auxy * 0.5;
    } else {
      px = DGS(1, e) * DA2L(m.hpressure, nlz, n0) / D_RHO0 + DGS(2, e) * DA2L(m.hpressure, nlz, n1) / D_RHO0 + DGS(3, e) * DA2L(m.hpressure, nlz, n2) / D_RHO0;
      py = DGS(4, e) * DA2L(m.hpressure, nlz, n0) / D_RHO0 + DGS(5, e) * DA2L(m.hpressure, nlz, n1) / D_RHO0 + DGS(6, e) * DA2L(m.hpressure, nlz, n2) / D_RHO0;
    }
    DA2(m.pgf_x, nlz, e) = px; DA2(m.pgf_y, nlz, e) = py;
    return;
  }
  // int_dp_dx after level nlz: first level assigns aux, later levels add (reference order)
  double ax0 = bcast(auxx, ule - 1), ay0 = bcast(auxy, ule - 1);
  const bool has_p0 = m.p.which_pgf == 1 && m.p.which_ale == 0 && ule > 1;                     // (linfs cubic spline under a shelf: the integral starts from the boundary term)
  if (has_p0) { p0x = bcast(p0x, ule - 1); p0y = bcast(p0y, ule - 1); ax0 = p0x + ax0; ay0 = p0y + ay0; }
  double ix = seq_sum_up(auxx, ule, nle - 1, ax0), iy = seq_sum_up(auxy, ule, nle - 1, ay0);   // inclusive sums
  double ixp = shup(ix), iyp = shup(iy);                                                       // sum before this level
  if (nlz == ule + 1) { ixp = ax0; iyp = ay0; }
  if (wet) {
    DA2(m.pgf_x, nlz, e) = (nlz == ule) ? (has_p0 ? p0x + auxx * 0.5 : auxx * 0.5) : ixp + auxx * 0.5;
    DA2(m.pgf_y, nlz, e) = (nlz == ule) ? (has_p0 ? p0y + auxy * 0.5 : auxy * 0.5) : iyp + auxy * 0.5;
  }
}

// The same (Shchepetkin variants) for CORE2-class meshes (DM::use_tile).  The three vertical sums of an element column are sequential in the
// reference; in k_pgf all 64 lanes of a wave step through them for ONE element (2 DPP moves + 1 add per level and sum: what that kernel's time is
// made of).  Here a wave takes PG_ELEMS elements: per-level terms go through a wave-private LDS image [level][element], lane = element runs the
// chains of all its elements at once, the results go back through the image (in place).
#define PG_ELEMS 8                      // (measured on the 182 600-node meshes: 16 elements per wave 484 us, 8: 400 us, 4: 424 us -- the LDS images of a wave bound the occupancy)
#define PG_CP (PG_ELEMS + 1)
__global__ void __launch_bounds__(BLOCK) k_pgf_tile(DM m) {
  extern __shared__ double pg_sh[];
  const int w = threadIdx.x >> 6, l = lane_id(), nlz = l + 1, nl1 = m.nlm1;
  double *imA = pg_sh + (size_t)w * 2 * nl1 * PG_CP, *imB = imA + (size_t)nl1 * PG_CP;
  const int base = (xcd_block() * COLS_PER_BLOCK + w) * PG_ELEMS;
  // the index chain of all PG_ELEMS elements in one round (lane k = k-th element)
  int nle_l = 0, ule_l = 1, n0_l = 0, n1_l = 0, n2_l = 0, lv0_l = 0, lv1_l = 0, lv2_l = 0;      // lv: ulev_n | (nlev_n - 1) << 8 of the three nodes
  double bot_l = 0.0;
  if (l < PG_ELEMS && base + l < m.myE) {
    const int e = base + l;
    nle_l = m.nlev[e] - 1; ule_l = m.ulev[e]; bot_l = m.zbar_e_bot[e];
    n0_l = m.elem_nodes[3 * e]; n1_l = m.elem_nodes[3 * e + 1]; n2_l = m.elem_nodes[3 * e + 2];
    lv0_l = m.ulev_n[n0_l] | ((m.nlev_n[n0_l] - 1) << 8); lv1_l = m.ulev_n[n1_l] | ((m.nlev_n[n1_l] - 1) << 8); lv2_l = m.ulev_n[n2_l] | ((m.nlev_n[n2_l] - 1) << 8);
  }
  // A: layer thicknesses into the image
#pragma unroll 4
  for (int k = 0; k < PG_ELEMS; k++) {
    const int e = __builtin_amdgcn_readfirstlane(base + k);
    double he = 0.0;
    if (nlz >= rdlane(ule_l, k) && nlz <= rdlane(nle_l, k)) he = UA2(m.helem, nlz, e);      // (elements beyond myE: empty range)
    if (nlz <= nl1) imA[l * PG_CP + k] = he;
  }
  __builtin_amdgcn_wave_barrier();
  // B: lane = element: zbar_n(nlz) = zbar_e_bot + sum_{k = nle .. nlz} helem(k), bottom-up (seq_sum_down of k_pgf)
  {
    const int kk = l < PG_ELEMS ? l : PG_ELEMS - 1;
    double y = bot_l;
    for (int j = nl1 - 1; j >= 0; j--) {
      const bool in = (j <= nle_l - 1 && j >= ule_l - 1);
      const double x = imA[j * PG_CP + kk];
      y = in ? y + x : y;
      if (l < PG_ELEMS) imA[j * PG_CP + kk] = in ? y : bot_l;
    }
  }
  __builtin_amdgcn_wave_barrier();
  // C: lane = level: density-Jacobian terms of every element of the wave
#pragma unroll 2
  for (int k = 0; k < PG_ELEMS; k++) {
    const int e = __builtin_amdgcn_readfirstlane(base + k);
    const int nle = rdlane(nle_l, k), ule = rdlane(ule_l, k);
    const bool wet = (nlz >= ule && nlz <= nle);
    double auxx = 0.0, auxy = 0.0;
    const int lc = l < nl1 ? l : nl1 - 1, ln = l + 1 < nl1 ? l + 1 : nl1 - 1;
    const double zb_top = imA[lc * PG_CP + k];
    double zb_bot = imA[ln * PG_CP + k];
    (void)zb_top;
    if (wet) {
      const double bot = bcast(bot_l, k);
      if (nlz == nle) zb_bot = bot;
      const double he = UA2(m.helem, nlz, e);
      const double Zn = zb_bot + he * 0.5;
      const int en[3] = {rdlane(n0_l, k), rdlane(n1_l, k), rdlane(n2_l, k)};
      const int lv[3] = {rdlane(lv0_l, k), rdlane(lv1_l, k), rdlane(lv2_l, k)};
      double drho_dz[3], rho_c[3], z_c[3];
#pragma unroll
      for (int ni = 0; ni < 3; ni++) {
        const int n = en[ni];
        int k0;
        if (nlz == ule && (nlz - (lv[ni] & 0xff)) == 0) k0 = nlz + 1;
        else if (nlz == nle && nlz != ule && ((lv[ni] >> 8) - nlz) == 0) k0 = nlz - 1;
        else k0 = nlz;
        const double zm = UA2(m.Z_3d_n, k0 - 1, n), zc = UA2(m.Z_3d_n, k0, n);      // (node index wave-uniform: scalar column base)
        drho_dz[ni] = UA2(m.pgf_A, k0, n) + UA2(m.pgf_B, k0, n) * ((Zn - zc) + (Zn - zm));      // (the node column's two quotients: k_pressure_bv)
        rho_c[ni] = UA2(m.density_m_rho0, nlz, n);
        z_c[ni] = (k0 == nlz) ? zc : UA2(m.Z_3d_n, nlz, n);
      }
      double s3 = (drho_dz[0] + drho_dz[1] + drho_dz[2]) / 3.0;
      double drho_dx = DGS(1, e) * rho_c[0] + DGS(2, e) * rho_c[1] + DGS(3, e) * rho_c[2];
      double dz_dx = DGS(1, e) * z_c[0] + DGS(2, e) * z_c[1] + DGS(3, e) * z_c[2];
      const bool flat = m.p.which_ale == 0 && nlz != nle && !(nlz == ule && ule > 1);
      auxx = flat ? drho_dx * he * D_G / D_RHO0 : (drho_dx - s3 * dz_dx) * he * D_G / D_RHO0;
      double drho_dy = DGS(4, e) * rho_c[0] + DGS(5, e) * rho_c[1] + DGS(6, e) * rho_c[2];
      double dz_dy = DGS(4, e) * z_c[0] + DGS(5, e) * z_c[1] + DGS(6, e) * z_c[2];
      auxy = flat ? drho_dy * he * D_G / D_RHO0 : (drho_dy - s3 * dz_dy) * he * D_G / D_RHO0;
    }
    if (nlz <= nl1) { imA[l * PG_CP + k] = auxx; imB[l * PG_CP + k] = auxy; }
  }
  __builtin_amdgcn_wave_barrier();
  // D: lane = element: int_dp_dx top-down; the first level assigns aux, later levels add (reference order); pgf = sum before the level + aux / 2
  {
    const int kk = l < PG_ELEMS ? l : PG_ELEMS - 1;
    double sx = 0.0, sy = 0.0;
    for (int j = 0; j < nl1; j++) {
      const int lev = j + 1;
      const double ax = imA[j * PG_CP + kk], ay = imB[j * PG_CP + kk];
      const bool first = lev == ule_l, in = (lev >= ule_l && lev <= nle_l);
      const double ox = first ? ax * 0.5 : sx + ax * 0.5, oy = first ? ay * 0.5 : sy + ay * 0.5;
      if (in) { sx = first ? ax : sx + ax; sy = first ? ay : sy + ay; }
      if (l < PG_ELEMS) { imA[j * PG_CP + kk] = ox; imB[j * PG_CP + kk] = oy; }
    }
  }
  __builtin_amdgcn_wave_barrier();
  // E: results out
#pragma unroll 4
  for (int k = 0; k < PG_ELEMS; k++) {
    const int e = __builtin_amdgcn_readfirstlane(base + k);
    if (nlz >= rdlane(ule_l, k) && nlz <= rdlane(nle_l, k)) { UA2(m.pgf_x, nlz, e) = imA[l * PG_CP + k]; UA2(m.pgf_y, nlz, e) = imB[l * PG_CP + k]; }
  }
}
static void launch_pgf(const DM &m, hipStream_t s) {
  const bool shch = m.p.which_pgf == 0 && !(m.p.which_ale == 0 && !m.p.use_partial_cell && !(m.p.use_cavity && m.p.use_cavity_partial_cell));      // (cubicspline / nemo / sergey / full cells: k_pgf)
  if (m.use_tile && shch) {                    // (on pi the tile shape is slower: 21.7 against 11.6 us)
    const int per_block = COLS_PER_BLOCK * PG_ELEMS;
    hipLaunchKernelGGL(k_pgf_tile, dim3((m.myE + per_block - 1) / per_block), dim3(BLOCK), (size_t)COLS_PER_BLOCK * 2 * m.nlm1 * PG_CP * sizeof(double), s, m);
  } else hipLaunchKernelGGL(k_pgf, dim3(nblocks(m.myE)), dim3(BLOCK), 0, s, m);
}

// ------------------------------------------------------------------------------------------------
// compute_sigma_xy (:2826-2900) fused with compute_neutral_slope (:2905-2946): node gathers T,S at the
// 3 nodes of each surrounding element.  20 N3 values.
__global__ void __launch_bounds__(BLOCK) k_sigma_slope(DM m) {
  int n = col_id(m), l = lane_id(), nz = l + 1;
  if (n >= m.myN) return;
  const int nln = m.nlev_n[n] - 1, uln = m.ulev_n[n];
  const bool wet = (nz >= uln && nz <= nln);
  double vol = 0.0, tx = 0.0, ty = 0.0, sx = 0.0, sy = 0.0;
  int num = m.nie_num[n];
  for (int k = 0; k < num; k++) {
    int el = m.nie[(size_t)m.maxk * n + k];
    if (!(nz >= m.ulev[el] && nz <= m.nlev[el] - 1)) continue;
    double ar = m.elem_area[el];
    int e1 = m.elem_nodes[3 * el], e2 = m.elem_nodes[3 * el + 1], e3 = m.elem_nodes[3 * el + 2];
    double T1 = DTR(m.tr_arr, nz, e1, 0), T2 = DTR(m.tr_arr, nz, e2, 0), T3 = DTR(m.tr_arr, nz, e3, 0);
    double S1 = DTR(m.tr_arr, nz, e1, 1), S2 = DTR(m.tr_arr, nz, e2, 1), S3 = DTR(m.tr_arr, nz, e3, 1);
    vol = vol + ar;
    tx = tx + (DGS(1, el) * T1 + DGS(2, el) * T2 + DGS(3, el) * T3) * ar;
    ty = ty + (DGS(4, el) * T1 + DGS(5, el) * T2 + DGS(6, el) * T3) * ar;
    sx = sx + (DGS(1, el) * S1 + DGS(2, el) * S2 + DGS(3, el) * S3) * ar;
    sy = sy + (DGS(4, el) * S1 + DGS(5, el) * S2 + DGS(6, el) * S3) * ar;
  }
  double sg1 = 0.0, sg2 = 0.0;
  if (wet) {
    double al = DA2(m.sw_alpha, nz, n), be = DA2(m.sw_beta, nz, n);
    sg1 = (-al * tx + be * sx) / vol * D_RHO0;
    sg2 = (-al * ty + be * sy) / vol * D_RHO0;
    DV2(m.sigma_xy, 1, nz, n) = sg1;
    DV2(m.sigma_xy, 2, nz, n) = sg2;
  }
  if (nz <= m.nlm1) {
    double s1 = 0.0, s2 = 0.0, s3 = 0.0, c = 0.0;
    bool in = (nz >= uln + 1 && nz <= nln);
    if (in) {
      const double eps = 5.0e-6, S_cr = 1.0e-2, S_d = 1.0e-3;
      double bv0 = DA2L(m.bvfreq, nz, n), bv1 = DA2L(m.bvfreq, nz + 1, n);
      double ro_z_inv = 2.0 * D_G / D_RHO0 / dmax_(bv0 + bv1, eps * eps);
      s1 = sg1 * ro_z_inv; s2 = sg2 * ro_z_inv;
      s3 = sqrt(s1 * s1 + s2 * s2);
      c = 0.5 * (1.0 + tanh((S_cr - s3) / S_d));
      if ((bv0 <= 0.0) || (bv1 <= 0.0)) c = 0.0;
      DV3(m.neutral_slope, 1, nz, n) = s1; DV3(m.neutral_slope, 2, nz, n) = s2; DV3(m.neutral_slope, 3, nz, n) = s3;
    }
    DV3(m.slope_tapered, 1, nz, n) = s1 * c; DV3(m.slope_tapered, 2, nz, n) = s2 * c; DV3(m.slope_tapered, 3, nz, n) = s3 * c;
  }
}

// ------------------------------------------------------------------------------------------------
// oce_mixing_PP (src/oce_ale_mixing_pp.F90:2-83) + mo_convect (src/oce_mo_conv.F90:4-103, use_momix=.false.)
// Ri-dependent factor of oce_mixing_PP's first node loop (:27-43) at interface nz of node n: shear^2 / (shear^2 + 5 max(N^2,0) + 1e-14)
__device__ __forceinline__ double pp_raw(const DM &m, int nz, int n) {
  double dz_inv = 1.0 / (DA2(m.Z_3d_n, nz - 1, n) - DA2(m.Z_3d_n, nz, n));
  double du = DV2(m.Unode, 1, nz - 1, n) - DV2(m.Unode, 1, nz, n), dv = DV2(m.Unode, 2, nz - 1, n) - DV2(m.Unode, 2, nz, n);
  double shear = du * du + dv * dv;
  shear = shear * dz_inv * dz_inv;
  return shear / (shear + 5. * dmax_(DA2L(m.bvfreq, nz, n), 0.0) + 1.0e-14);
}
// Monin-Obukhov mixing of mo_convect (src/oce_mo_conv.F90:22-55): mo_length / pmlktmo (:107-182) for every node south of momix_lat,
// one thread per node (owned + halo, as in the reference); the mixing length is state.  (The reference's unsuffixed literals are doubles:
// it is built with -fdefault-real-8.)  exp is the device's: the one place where the last bit may differ from the host's libm.
__global__ void k_momix(DM m) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= m.N || !m.momix_node[n]) return;
  const double cosgam = 0.913632, qhw = 1 / 7.0, betas = 0.0008, betat = 0.00004;
  const double qfm = m.water_flux[n] * 34.0, qtm = -2.38e-7 * m.heat_flux[n];
  const double sx = m.stress_atmoce_x[n], sy = m.stress_atmoce_y[n], ui = m.u_ice[n], vi = m.v_ice[n], ai = m.a_ice[n];
  const double tau = sqrt(sx * sx + sy * sy), ustar = sqrt(tau / 1030.0), uabs = sqrt(ui * ui + vi * vi);
  const double qw = 1.25 * (ustar * ustar * ustar) * (1.0 - ai) + 0.005 * (uabs * uabs * uabs) * cosgam * ai;
  const double qrho = betas * qfm - betat * qtm;
  double ttmp = 60.0;
  if (qrho > 0.) ttmp = 0.0;
  else
    for (int iter = 1; iter <= 5; iter++) {
      double a1 = exp(-ttmp * qhw);
      double f0 = 2.0 * qw * a1 + 9.81 * qrho * ttmp;
      double f1 = -(2.0 * qw * a1 * qhw) + 9.81 * qrho;
      ttmp = ttmp - f0 / f1;
      ttmp = dmax_(ttmp, 10.0);
    }
  const double obuk = dmax_(ttmp, 10.0), rtc = m.p.dt / (10.0 * 86400.0);
  double ml = m.mixlength[n];
  if (obuk < ml) { double ret = (obuk - ml) * rtc; ml = ml + ret; }
  else ml = obuk;
  m.mixlength[n] = ml;
}

// Av incl. mo_convect element part.  The factor of the three nodes is evaluated here again instead of read from Kv, which makes the
// element and the node part independent of each other: both run in ONE launch (k_pp).
__device__ __forceinline__ void pp_elem_body(const DM &m, int e) {
  int nz = lane_id() + 1;
  if (e >= m.myE) return;
  int nzmin = m.ulev[e];
  if (nz < nzmin + 1 || nz > m.nlev[e] - 1) return;
  int n1 = m.elem_nodes[3 * e], n2 = m.elem_nodes[3 * e + 1], n3 = m.elem_nodes[3 * e + 2];
  double k1 = pp_raw(m, nz, n1), k2 = pp_raw(m, nz, n2), k3 = pp_raw(m, nz, n3);
  double av = 0.01 * (k1 * k1 + k2 * k2 + k3 * k3) / 3.0 + m.p.A_ver;
  if (m.p.use_instabmix && (DA2L(m.bvfreq, nz, n1) < 0. || DA2L(m.bvfreq, nz, n2) < 0. || DA2L(m.bvfreq, nz, n3) < 0.))
    av = dmax_(av, m.p.instabmix_kv);
  if (m.p.use_momix && m.momix_elem[e]) av = av + ((momix_mo(m, nz, n1) + momix_mo(m, nz, n2)) + momix_mo(m, nz, n3)) / 3.0;
  if (nzmin <= 1 && m.p.use_windmix && nz <= m.p.windmix_nl + 1) av = dmax_(av, m.p.windmix_kv);
  DA2L(m.Av, nz, e) = av;
}
__device__ __forceinline__ void pp_node_body(const DM &m, int n) {     // Kv: Ri factor, cubic + mo_convect node part
  int nz = lane_id() + 1;
  if (n >= m.N) return;
  int nzmin = m.ulev_n[n];
  if (nz < nzmin + 1 || nz > m.nlev_n[n] - 1) return;
  double k = pp_raw(m, nz, n);
  double kv = 0.01 * (k * k * k) + (m.p.Kv0_const ? m.p.K_ver : kv0_background_qiang(m.lat_deg[n], fabs(DA2L(m.zbar_3d_n, nz, n))));
  if (m.p.use_momix) kv = kv + momix_mo(m, nz, n);
  if (m.p.use_instabmix && DA2L(m.bvfreq, nz, n) < 0.) kv = dmax_(kv, m.p.instabmix_kv);
  if (nzmin <= 1 && m.p.use_windmix && nz <= m.p.windmix_nl + 1) kv = dmax_(kv, m.p.windmix_kv);
  DA2L(m.Kv, nz, n) = kv;
}
__global__ void __launch_bounds__(BLOCK) k_pp_elem(DM m) { pp_elem_body(m, col_id(m)); }
__global__ void __launch_bounds__(BLOCK) k_pp_node_final(DM m) { pp_node_body(m, col_id(m)); }
// oce_mixing_PP + mo_convect in one launch: the first ncolE column slots are element columns, the rest node columns
__global__ void __launch_bounds__(BLOCK) k_pp(DM m, int ncolE) {
  const int c = col_id(m);
  if (c < ncolE) pp_elem_body(m, c); else pp_node_body(m, c - ncolE);
}

// ------------------------------------------------------------------------------------------------
// momentum_adv_scalar, node part (src/oce_ale_vel_rhs.F90:154-331): vertical advection from the element
// cluster + horizontal flux-form advection gathered over the incident edges (reference edge order).
// The index chains (element cluster; incident edges with their triangles, level ranges and cross-edge coefficients) are read lane-parallel
// (lane k = k-th element / edge) and broadcast with v_readlane; the field loads of MA_B elements / edges are issued as one batch before the ordered
// sums -- no chain of dependent loads per element or edge (same arithmetic and order as the plain loops).
__global__ void __launch_bounds__(BLOCK) k_momadv_node(DM m) {
  int n = col_id(m), l = lane_id(), nz = l + 1;
  if (n >= m.myN) return;
  const int nl1 = m.nlev_n[n] - 1, ul1 = m.ulev_n[n];
  const int nzc = nz <= m.nlm1 ? nz : m.nlm1, nzm = nzc > 1 ? nzc - 1 : 1;
  constexpr int MA_B = 3;
  double wu = 0.0, wv = 0.0;                 // wu(nz), nz = 1..nl1+1
  {
    const int num = m.nie_num[n];
    int el_l = 0, r_l = 1;                   // range packed ule | nle << 8 ; (1, 0) = empty
    double ar_l = 0.0;
    if (l < num) {
      el_l = m.nie[(size_t)m.maxk * n + l];
      r_l = m.ulev[el_l] | ((m.nlev[el_l] - 1) << 8);
      ar_l = m.elem_area[el_l];
    }
    for (int k0 = 0; k0 < num; k0 += MA_B) {
      double u0[MA_B], v0[MA_B], um[MA_B], vm[MA_B];
#pragma unroll
      for (int k = 0; k < MA_B; k++) {
        const int el = rdlane(el_l, (k0 + k < num) ? k0 + k : 0);
        u0[k] = DV2(m.UV, 1, nzc, el); v0[k] = DV2(m.UV, 2, nzc, el);
        um[k] = DV2(m.UV, 1, nzm, el); vm[k] = DV2(m.UV, 2, nzm, el);
      }
#pragma unroll
      for (int k = 0; k < MA_B; k++) {
        const int kk = k0 + k;
        if (kk < num) {
          const int r = rdlane(r_l, kk), ule = r & 0xff, nle = r >> 8;
          const double ar = bcast(ar_l, kk);
          if (ule == 1 && nz == ule) { wu = wu + u0[k] * ar; wv = wv + v0[k] * ar; }
          if (nz >= ule + 1 && nz <= nle) { wu = wu + 0.5 * (u0[k] + um[k]) * ar; wv = wv + 0.5 * (v0[k] + vm[k]) * ar; }
        }
      }
    }
  }
  const bool wet = (nz >= ul1 && nz <= nl1);
  if (wet) { double we = DA2L(m.Wvel_e, nz, n); wu = wu * we; wv = wv * we; }
  double wu_dn = shdn(wu), wv_dn = shdn(wv);   // wu(nz+1); wu(nl1+1)=0 by construction
  if (nz == nl1) { wu_dn = 0.0; wv_dn = 0.0; }
  double ur = 0.0, vr = 0.0;
  if (wet) {
    double h3 = 3.0 * DA2(m.hnode, nz, n);
    ur = -(wu - wu_dn) / h3;
    vr = -(wv - wv_dn) / h3;
  }
  {
    const int q0 = m.ne_ptr[n], deg = m.ne_ptr[n + 1] - q0;
    int sg_l = 0, e1_l = 0, e2_l = 0, r1_l = 1, r2_l = 1, has2_l = 0;
    double x1_l = 0.0, x2_l = 0.0, x3_l = 0.0, x4_l = 0.0;
    if (l < deg) {
      const int ed = m.ne_idx[q0 + l];
      sg_l = m.ne_sgn[q0 + l];
      e1_l = m.edge_tri[2 * ed];
      const int e2 = m.edge_tri[2 * ed + 1];
      r1_l = m.ulev[e1_l] | ((m.nlev[e1_l] - 1) << 8);
      has2_l = e2 >= 0;
      e2_l = e2 >= 0 ? e2 : e1_l;
      r2_l = e2 >= 0 ? (m.ulev[e2] | ((m.nlev[e2] - 1) << 8)) : 1;
      x1_l = DECD(1, ed); x2_l = DECD(2, ed); x3_l = DECD(3, ed); x4_l = DECD(4, ed);
    }
    for (int k0 = 0; k0 < deg; k0 += MA_B) {
      double A1[MA_B], B1[MA_B], A2[MA_B], B2[MA_B];
#pragma unroll
      for (int k = 0; k < MA_B; k++) {
        const int kk = (k0 + k < deg) ? k0 + k : 0;
        const int e1 = rdlane(e1_l, kk), e2 = rdlane(e2_l, kk);
        A1[k] = DV2(m.UV, 1, nzc, e1); B1[k] = DV2(m.UV, 2, nzc, e1);
        A2[k] = DV2(m.UV, 1, nzc, e2); B2[k] = DV2(m.UV, 2, nzc, e2);
      }
#pragma unroll
      for (int k = 0; k < MA_B; k++) {
        const int kk = k0 + k;
        if (kk < deg) {
          const int r1 = rdlane(r1_l, kk), r2 = rdlane(r2_l, kk), u1 = r1 & 0xff, l1 = r1 >> 8;
          const bool pos = rdlane(sg_l, kk) > 0, has2 = rdlane(has2_l, kk) != 0;
          const double a1 = A1[k], b1 = B1[k], a2 = A2[k], b2 = B2[k];
          double un1 = 0.0, un2 = 0.0;
          if (nz >= u1 && nz <= l1) un1 = b1 * bcast(x1_l, kk) - a1 * bcast(x2_l, kk);
          if (has2) {
            const int u2 = r2 & 0xff, l2 = r2 >> 8;
            if (nz >= u2 && nz <= l2) un2 = -b2 * bcast(x3_l, kk) + a2 * bcast(x4_l, kk);
            const bool act = (nz >= (u1 < u2 ? u1 : u2) && nz <= (l1 > l2 ? l1 : l2));
            if (act) {
              if (pos) { ur = ur + un1 * a1 + un2 * a2; vr = vr + un1 * b1 + un2 * b2; }
              else     { ur = ur - un1 * a1 - un2 * a2; vr = vr - un1 * b1 - un2 * b2; }
            }
          } else if (nz >= u1 && nz <= l1) {
            if (pos) { ur = ur + un1 * a1; vr = vr + un1 * b1; }
            else     { ur = ur - un1 * a1; vr = vr - un1 * b1; }
          }
        }
      }
    }
  }
  if (nz <= m.nlm1) {
    if (wet) { double ai = DA2L(m.areasvol_inv, nz, n); ur = ur * ai; vr = vr * ai; }
    DV2(m.Unode_rhs, 1, nz, n) = ur;
    DV2(m.Unode_rhs, 2, nz, n) = vr;
  }
}

// mom_adv = 3, compute_vel_rhs_vinv (src/oce_vel_rhs_vinv.F90:104-322; linear free surface only: hpressure): kinetic energy at nodes (:128-166),
// then (after relative_vorticity = k_leith_vort) the element part: old AB term, gradient of g eta + p/rho0, (f + zeta) k x u, gradient of the
// kinetic energy, AB2 update.  The reference's vertical term is multiplied by w = 0 (:122): nothing to add.
__global__ void __launch_bounds__(BLOCK) k_vinv_ke(DM m) {
  const int n = col_id(m), nz = lane_id() + 1;
  if (n >= m.myN || nz > m.nlm1) return;
  double ke = 0.0;
  const int num = m.nie_num[n];
  for (int k = 0; k < num; k++) {
    const int el = m.nie[(size_t)m.maxk * n + k];
    if (nz < m.ulev[el] || nz > m.nlev[el] - 1) continue;
    const double u = DV2(m.UV, 1, nz, el), v = DV2(m.UV, 2, nz, el);
    ke = ke + (u * u + v * v) * m.elem_area[el];
  }
  if (nz >= m.ulev_n[n] && nz <= m.nlev_n[n] - 1) ke = ke / (6. * DA2L(m.areasvol, nz, n));
  if (m.wall_node[n]) ke = 0.0;
  DA2(m.KE_node, nz, n) = ke;
}
__global__ void __launch_bounds__(BLOCK) k_vinv_elem(DM m, int first_step) {
  const int e = col_id(m), nz = lane_id() + 1;
  if (e >= m.myE) return;
  if (nz < m.ulev[e] || nz > m.nlev[e] - 1) return;
  const int n0 = m.elem_nodes[3 * e], n1 = m.elem_nodes[3 * e + 1], n2 = m.elem_nodes[3 * e + 2];
  const double eps = m.p.epsilon, d0inv = 1. / D_RHO0, gg = m.elem_area[e];
  double r1 = -(0.5 + eps) * DV2(m.UV_rhsAB, 1, nz, e), r2 = -(0.5 + eps) * DV2(m.UV_rhsAB, 2, nz, e);
  double p0 = -(D_G * m.eta_n[n0] + DA2L(m.hpressure, nz, n0) * d0inv), p1 = -(D_G * m.eta_n[n1] + DA2L(m.hpressure, nz, n1) * d0inv),
         p2 = -(D_G * m.eta_n[n2] + DA2L(m.hpressure, nz, n2) * d0inv);
  double Fx = (DGS(1, e) * p0 + DGS(2, e) * p1) + DGS(3, e) * p2, Fy = (DGS(4, e) * p0 + DGS(5, e) * p1) + DGS(6, e) * p2;
  r1 = r1 + Fx * gg; r2 = r2 + Fy * gg;
  p0 = -DA2(m.KE_node, nz, n0); p1 = -DA2(m.KE_node, nz, n1); p2 = -DA2(m.KE_node, nz, n2);
  Fx = (DGS(1, e) * p0 + DGS(2, e) * p1) + DGS(3, e) * p2; Fy = (DGS(4, e) * p0 + DGS(5, e) * p1) + DGS(6, e) * p2;
  const double sfv = ((m.coriolis_node[n0] + DA2(m.vorticity, nz, n0)) + (m.coriolis_node[n1] + DA2(m.vorticity, nz, n1))) + (m.coriolis_node[n2] + DA2(m.vorticity, nz, n2));
  const double da = DV2(m.UV, 2, nz, e) * sfv / 3.0, db = -DV2(m.UV, 1, nz, e) * sfv / 3.0;
  const double ab1 = (da + Fx) * gg, ab2 = (db + Fy) * gg;
  DV2(m.UV_rhsAB, 1, nz, e) = ab1; DV2(m.UV_rhsAB, 2, nz, e) = ab2;
  const double g2 = first_step ? 1.0 : (1.5 + eps), ai = m.p.dt / gg;
  DV2(m.UV_rhs, 1, nz, e) = (r1 + ab1 * g2) * ai;
  DV2(m.UV_rhs, 2, nz, e) = (r2 + ab2 * g2) * ai;
}

// compute_vel_rhs (src/oce_ale_vel_rhs.F90:13-148) incl. the element part of momentum_adv_scalar (:333-343):
// one streaming pass per element column.  16 E3 values.
__global__ void __launch_bounds__(BLOCK) k_vel_rhs(DM m, int first_step) {
  int e = col_id(m), nz = lane_id() + 1;
  if (e >= m.myE) return;
  if (nz < m.ulev[e] || nz > m.nlev[e] - 1) return;
  const double eps = m.p.epsilon, dt = m.p.dt;
  int n1 = m.elem_nodes[3 * e], n2 = m.elem_nodes[3 * e + 1], n3 = m.elem_nodes[3 * e + 2];
  double ar = m.elem_area[e];
  double r1 = -(0.5 + eps) * DV2(m.UV_rhsAB, 1, nz, e), r2 = -(0.5 + eps) * DV2(m.UV_rhsAB, 2, nz, e);
  double p0 = surf_pre(m, n1), p1 = surf_pre(m, n2), p2 = surf_pre(m, n3);
  double ff = m.coriolis[e] * ar;
  double Fx = DGS(1, e) * p0 + DGS(2, e) * p1 + DGS(3, e) * p2;
  double Fy = DGS(4, e) * p0 + DGS(5, e) * p1 + DGS(6, e) * p2;
  r1 = r1 + (Fx - DA2(m.pgf_x, nz, e)) * ar;
  r2 = r2 + (Fy - DA2(m.pgf_y, nz, e)) * ar;
  double ab1 = DV2(m.UV, 2, nz, e) * ff, ab2 = -DV2(m.UV, 1, nz, e) * ff;
  ab1 = ab1 + ar * (DV2(m.Unode_rhs, 1, nz, n1) + DV2(m.Unode_rhs, 1, nz, n2) + DV2(m.Unode_rhs, 1, nz, n3)) / 3.0;
  ab2 = ab2 + ar * (DV2(m.Unode_rhs, 2, nz, n1) + DV2(m.Unode_rhs, 2, nz, n2) + DV2(m.Unode_rhs, 2, nz, n3)) / 3.0;
  DV2(m.UV_rhsAB, 1, nz, e) = ab1;
  DV2(m.UV_rhsAB, 2, nz, e) = ab2;
  double f2 = first_step ? 1.0 : (1.5 + eps);
  DV2(m.UV_rhs, 1, nz, e) = dt * (r1 + ab1 * f2) / ar;
  DV2(m.UV_rhs, 2, nz, e) = dt * (r2 + ab2 * f2) / ar;
}

// ------------------------------------------------------------------------------------------------
// visc_option 1 / 2 / 3: the Leith coefficient of h_viscosity_leith (src/oce_dyn.F90:461-561).
// relative_vorticity (src/oce_vel_rhs_vinv.F90:14-102): circulation around the scalar control volume, gathered over the node's
// incident owned edges in the reference's edge order, / areasvol.  Owned nodes; halo nodes arrive by exchange.
__global__ void __launch_bounds__(BLOCK) k_leith_vort(DM m) {
  int n = col_id(m), nz = lane_id() + 1;
  if (n >= m.myN || nz > m.nlm1) return;
  double vo = 0.0;
  for (int q = m.ne_ptr[n]; q < m.ne_ptr[n + 1]; q++) {
    int ed = m.ne_idx[q], sg = m.ne_sgn[q];
    int el1 = m.edge_tri[2 * ed], el2 = m.edge_tri[2 * ed + 1];
    bool in1 = nz >= m.ulev[el1] && nz <= m.nlev[el1] - 1, in2 = false;
    if (el2 >= 0) in2 = nz >= m.ulev[el2] && nz <= m.nlev[el2] - 1;
    if (!in1 && !in2) continue;
    double c1;
    if (in1 && in2) c1 = DECD(1, ed) * DV2(m.UV, 1, nz, el1) + DECD(2, ed) * DV2(m.UV, 2, nz, el1) - DECD(3, ed) * DV2(m.UV, 1, nz, el2) - DECD(4, ed) * DV2(m.UV, 2, nz, el2);
    else if (in1)   c1 = DECD(1, ed) * DV2(m.UV, 1, nz, el1) + DECD(2, ed) * DV2(m.UV, 2, nz, el1);
    else            c1 = -DECD(3, ed) * DV2(m.UV, 1, nz, el2) - DECD(4, ed) * DV2(m.UV, 2, nz, el2);
    vo = sg > 0 ? vo + c1 : vo - c1;
  }
  if (nz >= m.ulev_n[n] && nz <= m.nlev_n[n] - 1) vo = vo / DA2L(m.areasvol, nz, n);
  DA2(m.vorticity, nz, n) = vo;
}
// Leith + modified Leith coefficient on the owned elements (:483-523); halo elements hold 0 through the smoothing rounds, as in
// the reference (Visc = 0 at :482, exchange_elem only after the rounds :558)
__global__ void __launch_bounds__(BLOCK) k_leith_elem(DM m) {
  int e = col_id(m), l = lane_id(), nz = l + 1;
  if (e >= m.E) return;
  double vi = 0.0;
  if (e < m.myE) {
    const int nl1 = m.nlev[e] - 1, ul1 = m.ulev[e];
    const bool wet = nz >= ul1 && nz <= nl1;
    double he = wet ? DA2(m.helem, nz, e) : 0.0;
    double zt = seq_sum_down(he, nl1 - 1, ul1 - 1, m.zbar_e_bot[e]);          // zbar_n(nz), summed from the bottom up
    double zb = shdn(zt);
    if (nz == nl1) zb = m.zbar_e_bot[e];
    if (wet) {
      const double dz = zt - zb, ar = m.elem_area[e];
      const int n1 = m.elem_nodes[3 * e], n2 = m.elem_nodes[3 * e + 1], n3 = m.elem_nodes[3 * e + 2];
      double d1 = (DA2L(m.Wvel, nz, n1) - DA2L(m.Wvel, nz + 1, n1)) / dz, d2 = (DA2L(m.Wvel, nz, n2) - DA2L(m.Wvel, nz + 1, n2)) / dz,
             d3 = (DA2L(m.Wvel, nz, n3) - DA2L(m.Wvel, nz + 1, n3)) / dz;
      double v1 = DA2(m.vorticity, nz, n1), v2 = DA2(m.vorticity, nz, n2), v3 = DA2(m.vorticity, nz, n3);
      double xe = (DGS(1, e) * d1 + DGS(2, e) * d2) + DGS(3, e) * d3, ye = (DGS(4, e) * d1 + DGS(5, e) * d2) + DGS(6, e) * d3;
      double lx = (DGS(1, e) * v1 + DGS(2, e) * v2) + DGS(3, e) * v3, ly = (DGS(4, e) * v1 + DGS(5, e) * v2) + DGS(6, e) * v3;
      vi = dmin_(m.p.gamma1 * ar * sqrt((m.p.Div_c * (xe * xe + ye * ye) + m.p.Leith_c * (lx * lx + ly * ly)) * ar), ar / m.p.dt);
    }
  }
  if (nz <= m.nlm1) DA2(m.Visc, nz, e) = vi;
}
// the two smoothing rounds (:527-557): area-weighted node average over the element cluster, then the mean of the three nodes
__global__ void __launch_bounds__(BLOCK) k_leith_node(DM m) {
  int n = col_id(m), nz = lane_id() + 1;
  if (n >= m.myN) return;
  if (nz < m.ulev_n[n] || nz > m.nlev_n[n] - 1) return;
  double dz = 0.0, vi = 0.0;
  const int num = m.nie_num[n];
  for (int k = 0; k < num; k++) {
    int el = m.nie[(size_t)m.maxk * n + k];
    double ar = m.elem_area[el];
    dz = dz + ar;
    vi = vi + DA2(m.Visc, nz, el) * ar;
  }
  DA2(m.leith_aux, nz, n) = vi / dz;
}
__global__ void __launch_bounds__(BLOCK) k_leith_avg(DM m) {
  int e = col_id(m), nz = lane_id() + 1;
  if (e >= m.myE || nz > m.nlm1) return;
  double vi = 0.0;
  if (nz >= m.ulev[e] && nz <= m.nlev[e] - 1) {
    const int n1 = m.elem_nodes[3 * e], n2 = m.elem_nodes[3 * e + 1], n3 = m.elem_nodes[3 * e + 2];
    vi = ((DA2(m.leith_aux, nz, n1) + DA2(m.leith_aux, nz, n2)) + DA2(m.leith_aux, nz, n3)) / 3.0;
  }
  DA2(m.Visc, nz, e) = vi;
}

// ------------------------------------------------------------------------------------------------
// visc_filt_bcksct (src/oce_dyn.F90:563-649): (1) element gather over its <=3 internal edges,
// (2) node average, (3) apply (fused into k_impl_visc).  4 N3 + 12 E3 values.
// visc_option 4 / 6 / 7 (visc_filt_biharm(1) :275-372, visc_filt_bilapl :658-726, visc_filt_bidiff :734-801): the same gather is the first stage of the
// biharmonic operator (the result lives in U_b, the reference's U_c/V_c), the second stage is k_visc_apply.
// (measured, round 3: the element's own velocity once + its three neighbours in one batch of loads instead of both triangles edge by edge: 321 -> 318 us on the
//  basin -- the kernel is bound by its square roots and divisions, 58 % VALU issue, not by its gathers)
__global__ void __launch_bounds__(BLOCK) k_visc_elem(DM m) {
  int e = col_id(m), nz = lane_id() + 1;
  if (e >= m.E) return;
  if (nz > m.nlm1) return;
  double ub = 0.0, vb = 0.0;
  const double dt = m.p.dt, g0 = m.p.gamma0, g1 = m.p.gamma1, g2 = m.p.gamma2;
  const int opt = m.p.visc_option;
  for (int q = 0; q < 3; q++) {
    int side = m.ee_side[3 * e + q];
    if (side == 0) continue;
    int ed = m.ee_idx[3 * e + q];
    int e1 = m.edge_tri[2 * ed], e2 = m.edge_tri[2 * ed + 1];
    int nzmax = min(m.nlev[e1], m.nlev[e2]), nzmin = max(m.ulev[e1], m.ulev[e2]);
    if (nz < nzmin || nz > nzmax - 1) continue;
    double a1 = m.elem_area[e1], a2 = m.elem_area[e2];
    double len = sqrt(a1 + a2);
    double u1 = DV2(m.UV, 1, nz, e1) - DV2(m.UV, 1, nz, e2);
    double v1 = DV2(m.UV, 2, nz, e1) - DV2(m.UV, 2, nz, e2);
    if (opt == 5) {
      double vi = dt * dmax_(g0, dmax_(g1 * sqrt(u1 * u1 + v1 * v1), g2 * (u1 * u1 + v1 * v1))) * len;
      u1 = u1 * vi; v1 = v1 * vi;
      if (side == 1) { ub = ub - u1 / a1; vb = vb - v1 / a1; }
      else           { ub = ub + u1 / a2; vb = vb + v1 / a2; }
    } else {
      if (opt == 7) {
        double vi = u1 * u1 + v1 * v1;
        vi = sqrt(dmax_(g0, dmax_(g1 * sqrt(vi), g2 * vi)) * len);
        u1 = u1 * vi; v1 = v1 * vi;
      }
      if (side == 1) { ub = ub - u1; vb = vb - v1; }
      else           { ub = ub + u1; vb = vb + v1; }
    }
  }
  if (opt == 6 && nz >= m.ulev[e] && nz <= m.nlev[e] - 1) {    // :694-706 (owned elements; halo values arrive by exchange)
    double len = sqrt(m.elem_area[e]);
    double u1 = ub * ub + vb * vb;
    double vi = dmax_(g0, dmax_(g1 * sqrt(u1), g2 * u1)) * len * dt;
    ub = -ub * vi; vb = -vb * vi;
  }
  if ((opt == 2 || opt == 3) && nz >= m.ulev[e] && nz <= m.nlev[e] - 1) {    // visc_filt_hbhmix :427-436 (background), visc_filt_biharm(2) :332-343 (Leith)
    double len = sqrt(m.elem_area[e]);
    double vi = (opt == 2) ? dt * g0 * len : dmax_(DA2(m.Visc, nz, e), g0 * len) * dt;
    ub = -ub * vi; vb = -vb * vi;
  }
  if (opt == 4 && nz >= m.ulev[e] && nz <= m.nlev[e] - 1) {    // visc_filt_biharm(1) :314-331: "an analog to the third-order upwind", vi = gamma1 |u| l
    double len = sqrt(m.elem_area[e]);
    double uu = DV2(m.UV, 1, nz, e), vv = DV2(m.UV, 2, nz, e);
    double vi = dmax_(g0, g1 * sqrt(uu * uu + vv * vv)) * len * dt;
    ub = -ub * vi; vb = -vb * vi;
  }
  DV2(m.U_b, 1, nz, e) = ub;
  DV2(m.U_b, 2, nz, e) = vb;
}
// second stage of visc_option 6 / 7 (:709-724, :781-799): UV_rhs += differences of the first-stage field over the internal
// edges of the element, in the reference's edge order
__global__ void __launch_bounds__(BLOCK) k_visc_apply(DM m) {
  int e = col_id(m), nz = lane_id() + 1;
  if (e >= m.myE) return;
  if (nz > m.nlm1) return;
  const double dt = m.p.dt, g0 = m.p.gamma0, g1 = m.p.gamma1, g2 = m.p.gamma2;
  const int opt = m.p.visc_option;
  double ur = DV2(m.UV_rhs, 1, nz, e), vr = DV2(m.UV_rhs, 2, nz, e);
  if (opt == 1 || opt == 2)           // harmonic Leith viscosity: visc_filt_harmon :236-273, first edge loop of visc_filt_hbhmix :398-425
    for (int q = 0; q < 3; q++) {
      int side = m.ee_side[3 * e + q];
      if (side == 0) continue;
      int ed = m.ee_idx[3 * e + q];
      int e1 = m.edge_tri[2 * ed], e2 = m.edge_tri[2 * ed + 1];
      int nzmax = min(m.nlev[e1], m.nlev[e2]), nzmin = max(m.ulev[e1], m.ulev[e2]);
      if (nz < nzmin || nz > nzmax - 1) continue;
      double a1 = m.elem_area[e1], a2 = m.elem_area[e2];
      double u1 = DV2(m.UV, 1, nz, e1) - DV2(m.UV, 1, nz, e2);
      double v1 = DV2(m.UV, 2, nz, e1) - DV2(m.UV, 2, nz, e2);
      double vi;
      if (opt == 1) { vi = 0.5 * (DA2(m.Visc, nz, e1) + DA2(m.Visc, nz, e2)); vi = dmax_(vi, g0 * sqrt(a1 + a2)) * dt; }
      else vi = dt * 0.5 * (DA2(m.Visc, nz, e1) + DA2(m.Visc, nz, e2));
      u1 = u1 * vi; v1 = v1 * vi;
      if (side == 1) { ur = ur - u1 / a1; vr = vr - v1 / a1; }
      else           { ur = ur + u1 / a2; vr = vr + v1 / a2; }
    }
  for (int q = 0; q < 3 && opt != 1; q++) {
    int side = m.ee_side[3 * e + q];
    if (side == 0) continue;
    int ed = m.ee_idx[3 * e + q];
    int e1 = m.edge_tri[2 * ed], e2 = m.edge_tri[2 * ed + 1];
    int nzmax = min(m.nlev[e1], m.nlev[e2]), nzmin = max(m.ulev[e1], m.ulev[e2]);
    if (nz < nzmin || nz > nzmax - 1) continue;
    double a1 = m.elem_area[e1], a2 = m.elem_area[e2];
    double u1 = DV2(m.U_b, 1, nz, e1) - DV2(m.U_b, 1, nz, e2);
    double v1 = DV2(m.U_b, 2, nz, e1) - DV2(m.U_b, 2, nz, e2);
    if (opt == 7) {
      double len = sqrt(a1 + a2);
      double du = DV2(m.UV, 1, nz, e1) - DV2(m.UV, 1, nz, e2);
      double dv = DV2(m.UV, 2, nz, e1) - DV2(m.UV, 2, nz, e2);
      double vi = du * du + dv * dv;
      vi = -dt * sqrt(dmax_(g0, dmax_(g1 * sqrt(vi), g2 * vi)) * len);
      u1 = vi * u1; v1 = vi * v1;
    }
    if (side == 1) { ur = ur - u1 / a1; vr = vr - v1 / a1; }
    else           { ur = ur + u1 / a2; vr = vr + v1 / a2; }
  }
  DV2(m.UV_rhs, 1, nz, e) = ur;
  DV2(m.UV_rhs, 2, nz, e) = vr;
}
__global__ void __launch_bounds__(BLOCK) k_visc_node(DM m) {
  int n = col_id(m), nz = lane_id() + 1;
  if (n >= m.myN) return;
  if (nz < m.ulev_n[n] || nz > m.nlev_n[n] - 1) return;
  double vi = 0.0, u1 = 0.0, v1 = 0.0;
  int num = m.nie_num[n];
  if (m.exp_batch & 8) {                       // batches of independent loads, sums in the reference's element order (as k_vel_nodes)
    constexpr int VB = 6;
    for (int k0 = 0; k0 < num; k0 += VB) {
      double u[VB], v[VB], a[VB];
#pragma unroll
      for (int j = 0; j < VB; j++) {
        const int e = m.nie[(size_t)m.maxk * n + (k0 + j < num ? k0 + j : 0)];
        a[j] = m.elem_area[e];
        u[j] = DV2(m.U_b, 1, nz, e); v[j] = DV2(m.U_b, 2, nz, e);
      }
#pragma unroll
      for (int j = 0; j < VB; j++) {
        if (k0 + j < num) { vi = vi + a[j]; u1 = u1 + u[j] * a[j]; v1 = v1 + v[j] * a[j]; }      // (wave-uniform)
      }
    }
  } else
  for (int k = 0; k < num; k++) {
    int e = m.nie[(size_t)m.maxk * n + k];
    double ar = m.elem_area[e];
    vi = vi + ar;
    u1 = u1 + DV2(m.U_b, 1, nz, e) * ar;
    v1 = v1 + DV2(m.U_b, 2, nz, e) * ar;
  }
  DV2(m.U_c, 1, nz, n) = u1 / vi;
  DV2(m.U_c, 2, nz, n) = v1 / vi;
}

// ------------------------------------------------------------------------------------------------
// visc_option = 8: backscatter_coef (src/oce_dyn.F90:967-996), visc_filt_dbcksc (:806-964), uke_update (:999-1152) -- kinematic backscatter with
// the prognostic sub-grid energy `uke`.  Single partition, which_toy = 'soufflet' (the branch without the regional mask, :1122-1125).  The
// reference's edge loops become gathers over the <=3 internal edges of an element in edge order (ee_idx), as in k_visc_elem.
// (1) v_back and the first stage U_c (kept in U_b)
__global__ void __launch_bounds__(BLOCK) k_v8_first(DM m) {
  int e = col_id(m), nz = lane_id() + 1;
  if (e >= m.myE) return;
  if (nz > m.nlm1) return;
  const double dt = m.p.dt, ar = m.elem_area[e];
  const bool wet = nz <= m.nlev[e] - 1;
  DA2(m.v_back, nz, e) = wet ? dmin_(-m.p.c_back * sqrt(ar) * sqrt(dmax_(2.0 * DA2(m.uke, nz, e), 0.0)), 0.2 * ar / dt) : 0.0;
  double ub = 0.0, vb = 0.0;
  for (int q = 0; q < 3; q++) {
    int side = m.ee_side[3 * e + q];
    if (side == 0) continue;
    int ed = m.ee_idx[3 * e + q];
    int e1 = m.edge_tri[2 * ed], e2 = m.edge_tri[2 * ed + 1];
    if (nz > min(m.nlev[e1], m.nlev[e2]) - 1) continue;
    double u1 = DV2(m.UV, 1, nz, e1) - DV2(m.UV, 1, nz, e2);
    double v1 = DV2(m.UV, 2, nz, e1) - DV2(m.UV, 2, nz, e2);
    if (side == 1) { ub = ub - u1; vb = vb - v1; }
    else           { ub = ub + u1; vb = vb + v1; }
  }
  if (wet) {      // :861-870
    double len = sqrt(ar);
    len = dt * len / 30.0;
    double uu = DV2(m.UV, 1, nz, e), vv = DV2(m.UV, 2, nz, e);
    double vi = dmax_(0.2, sqrt(uu * uu + vv * vv)) * len;
    ub = -ub * vi; vb = -vb * vi;
  }
  DV2(m.U_b, 1, nz, e) = ub;
  DV2(m.U_b, 2, nz, e) = vb;
}
// (2) the tendencies of the second edge loop (:875-917): backscatter (UV_back_tend), dissipation (UV_dis_tend), diffusion of uke (uke_dif)
__global__ void __launch_bounds__(BLOCK) k_v8_tend(DM m) {
  int e = col_id(m), nz = lane_id() + 1;
  if (e >= m.myE) return;
  if (nz > m.nlm1) return;
  const double dt = m.p.dt;
  double bu = 0.0, bv = 0.0, du = 0.0, dv = 0.0, kd = 0.0;
  for (int q = 0; q < 3; q++) {
    int side = m.ee_side[3 * e + q];
    if (side == 0) continue;
    int ed = m.ee_idx[3 * e + q];
    int e1 = m.edge_tri[2 * ed], e2 = m.edge_tri[2 * ed + 1];
    if (nz > min(m.nlev[e1], m.nlev[e2]) - 1) continue;
    double a1 = m.elem_area[e1], a2 = m.elem_area[e2];
    double le1 = m.edxy[2 * ed] * (m.elem_cos[e1] + m.elem_cos[e2]) * 0.25, le2 = m.edxy[2 * ed + 1];
    double len = sqrt(le1 * le1 + le2 * le2) * D_REARTH;
    le1 = m.ecd[4 * ed] - m.ecd[4 * ed + 2]; le2 = m.ecd[4 * ed + 1] - m.ecd[4 * ed + 3];
    double crosslen = sqrt(le1 * le1 + le2 * le2);
    double vi = dt * len * (DA2(m.v_back, nz, e1) + DA2(m.v_back, nz, e2)) / crosslen;
    double u1 = (DV2(m.UV, 1, nz, e1) - DV2(m.UV, 1, nz, e2)) * vi, v1 = (DV2(m.UV, 2, nz, e1) - DV2(m.UV, 2, nz, e2)) * vi;
    vi = dt * len * (m.p.K_back * sqrt(a1 / m.p.scale_area) + m.p.K_back * sqrt(a2 / m.p.scale_area)) / crosslen;
    double uke1 = (DA2(m.uke, nz, e1) - DA2(m.uke, nz, e2)) * vi;
    double uc = DV2(m.U_b, 1, nz, e1) - DV2(m.U_b, 1, nz, e2), vc = DV2(m.U_b, 2, nz, e1) - DV2(m.U_b, 2, nz, e2);
    if (side == 1) { bu = bu - u1 / a1; bv = bv - v1 / a1; kd = kd - uke1 / a1; du = du - uc / a1; dv = dv - vc / a1; }
    else           { bu = bu + u1 / a2; bv = bv + v1 / a2; kd = kd + uke1 / a2; du = du + uc / a2; dv = dv + vc / a2; }
  }
  DV2(m.UV_back_tend, 1, nz, e) = bu; DV2(m.UV_back_tend, 2, nz, e) = bv;
  DV2(m.UV_dis_tend, 1, nz, e) = du; DV2(m.UV_dis_tend, 2, nz, e) = dv;
  DA2(m.uke_dif, nz, e) = kd;
}
// smooth_elem2D (src/gen_support.F90:183-212), one round = element -> node (area-weighted mean over the whole cluster, dry cells included)
// -> element (mean of the three nodes); every level of the array, nc interleaved components
__global__ void __launch_bounds__(BLOCK) k_v8_smooth_node(DM m, const double *arr, int nc, double *work) {
  int n = col_id(m), nz = lane_id() + 1;
  if (n >= m.myN) return;
  if (nz > m.nlm1) return;
  double vol = 0.0, w0 = 0.0, w1 = 0.0;
  int num = m.nie_num[n];
  for (int k = 0; k < num; k++) {
    int e = m.nie[(size_t)m.maxk * n + k];
    double ar = m.elem_area[e];
    size_t i = ((size_t)e * m.nlm1 + (nz - 1)) * nc;
    w0 = w0 + arr[i] * ar;
    if (nc == 2) w1 = w1 + arr[i + 1] * ar;
    vol = vol + ar;
  }
  size_t o = ((size_t)n * m.nlm1 + (nz - 1)) * nc;
  work[o] = w0 / vol;
  if (nc == 2) work[o + 1] = w1 / vol;
}
__global__ void __launch_bounds__(BLOCK) k_v8_smooth_elem(DM m, double *arr, int nc, const double *work) {
  int e = col_id(m), nz = lane_id() + 1;
  if (e >= m.myE) return;
  if (nz > m.nlm1) return;
  const int n1 = m.elem_nodes[3 * e], n2 = m.elem_nodes[3 * e + 1], n3 = m.elem_nodes[3 * e + 2];
  for (int c = 0; c < nc; c++) {
    double a = work[((size_t)n1 * m.nlm1 + (nz - 1)) * nc + c], b = work[((size_t)n2 * m.nlm1 + (nz - 1)) * nc + c], d = work[((size_t)n3 * m.nlm1 + (nz - 1)) * nc + c];
    arr[((size_t)e * m.nlm1 + (nz - 1)) * nc + c] = ((a + b) + d) / 3.0;
  }
}
// (3) :945-952 UV_rhs += dissipation + smoothed backscatter; uke_update :1031-1043: the work of both tendencies
__global__ void __launch_bounds__(BLOCK) k_v8_apply(DM m) {
  int e = col_id(m), nz = lane_id() + 1;
  if (e >= m.myE) return;
  if (nz > m.nlm1) return;
  double kdis = 0.0, kback = 0.0;
  if (nz <= m.nlev[e] - 1) {
    double du = DV2(m.UV_dis_tend, 1, nz, e), dv = DV2(m.UV_dis_tend, 2, nz, e), bu = DV2(m.UV_back_tend, 1, nz, e), bv = DV2(m.UV_back_tend, 2, nz, e);
    double u = DV2(m.UV, 1, nz, e), v = DV2(m.UV, 2, nz, e);
    DV2(m.UV_rhs, 1, nz, e) = DV2(m.UV_rhs, 1, nz, e) + du + bu;
    DV2(m.UV_rhs, 2, nz, e) = DV2(m.UV_rhs, 2, nz, e) + dv + bv;
    kdis = (u * du + v * dv);
    kback = (u * bu + v * bv);
  }
  DA2(m.uke_dis, nz, e) = kdis; DA2(m.uke_back, nz, e) = kback;
}
// uke_update :1053-1070: U_work = area-weighted node mean of u over the whole cluster, V_work = U_work / vol as the reference has it (:1066); kept in U_c.
// One extra lane forms the baroclinic Rossby radius of the node column (:1090-1101) when uke_scaling is on.
__global__ void __launch_bounds__(BLOCK) k_v8_unode(DM m) {
  int n = col_id(m), nz = lane_id() + 1;
  if (n >= m.myN) return;
  if (m.p.uke_scaling && nz == WAVE) {
    const double c_min = 0.5, f_min = 1.e-6, r_max = 200000., pi = 3.14159265358979;
    double c1 = 0.0;
    int nzmax = m.nlev_n_min[n];
    for (int k = 1; k <= nzmax - 1; k++)
      c1 = c1 + DA2(m.hnode_new, k, n) * (sqrt(dmax_(DA2L(m.bvfreq, k, n), 0.0)) + sqrt(dmax_(DA2L(m.bvfreq, k + 1, n), 0.0))) / 2.;
    c1 = dmax_(c_min, c1 / pi);
    m.v8_rb[n] = dmin_(c1 / dmax_(fabs(m.coriolis_node[n]), f_min), r_max);
  }
  if (nz > m.nlm1) return;
  double vol = 0.0, u = 0.0;
  int num = m.nie_num[n];
  for (int k = 0; k < num; k++) {
    int e = m.nie[(size_t)m.maxk * n + k];
    double ar = m.elem_area[e];
    u = u + DV2(m.UV, 1, nz, e) * ar;
    vol = vol + ar;
  }
  u = u / vol;
  DV2(m.U_c, 1, nz, n) = u; DV2(m.U_c, 2, nz, n) = u / vol;
}
// :1072-1132 Rossby number of the node-mean flow, resolution scaling, damping of the dissipated energy
__global__ void __launch_bounds__(BLOCK) k_v8_rosb(DM m) {
  int e = col_id(m), nz = lane_id() + 1;
  if (e >= m.myE) return;
  if (nz > m.nlev[e] - 1) return;
  const double f_min = 1.e-6;
  const int n1 = m.elem_nodes[3 * e], n2 = m.elem_nodes[3 * e + 1], n3 = m.elem_nodes[3 * e + 2];
  const double *gs = m.gsca + 6 * (size_t)e;
  double u1 = DV2(m.U_c, 1, nz, n1), u2 = DV2(m.U_c, 1, nz, n2), u3 = DV2(m.U_c, 1, nz, n3);
  double v1 = DV2(m.U_c, 2, nz, n1), v2 = DV2(m.U_c, 2, nz, n2), v3 = DV2(m.U_c, 2, nz, n3);
  double gu = (gs[0] * u1 + gs[1] * u2) + gs[2] * u3, hv = (gs[3] * v1 + gs[4] * v2) + gs[5] * v3;
  double hu = (gs[3] * u1 + gs[4] * u2) + gs[5] * u3, gv = (gs[0] * v1 + gs[1] * v2) + gs[2] * v3;
  double rosb = sqrt((gu - hv) * (gu - hv) + (hu + gv) * (hu + gv));
  double scaling = 1.0;
  if (m.p.uke_scaling) {
    double reso = sqrt(m.elem_area[e] * 4.0 / sqrt(3.0));
    double rb = ((m.v8_rb[n1] + m.v8_rb[n2]) + m.v8_rb[n3]) / 3.0;
    scaling = 1.0 / (1.0 + (m.p.uke_scaling_factor * reso / rb));
  }
  double fsum = (m.coriolis_node[n1] + m.coriolis_node[n2]) + m.coriolis_node[n3];
  rosb = rosb / dmax_(fabs(fsum), f_min);
  DA2(m.uke_dis, nz, e) = scaling * 1.0 / (1.0 + rosb / m.p.rosb_dis) * DA2(m.uke_dis, nz, e);
}
// :1139-1149 second-order Adams-Bashforth step of uke
__global__ void __launch_bounds__(BLOCK) k_v8_uke(DM m) {
  int e = col_id(m), nz = lane_id() + 1;
  if (e >= m.myE) return;
  if (nz > m.nlev[e] - 1) return;
  double old = DA2(m.uke_rhs, nz, e);
  double rhs = -DA2(m.uke_dis, nz, e) - DA2(m.uke_back, nz, e) + DA2(m.uke_dif, nz, e);
  DA2(m.uke_rhs_old, nz, e) = old;
  DA2(m.uke_rhs, nz, e) = rhs;
  DA2(m.uke, nz, e) = DA2(m.uke, nz, e) + 1.5 * rhs - 0.5 * old;
}

// impl_vert_visc_ale (src/oce_ale.F90:2348-2517) with the last loop of visc_filt_bcksct (oce_dyn.F90:638-648)
// fused in front.  Coefficients per level in parallel (this kernel); the Thomas sweep runs one lane per column in
// k_thomas<2>.  1 N3 + 8 E3 values (+ 5 E3 scratch written, 5 read).
// Shapes (dev.h:ThTile): <8, 8> one element column per wave (pi), <TL_COLS, TL_WAVES> tiles with several columns per wave.
template <int COLS, int WAVES>
__global__ void __launch_bounds__(WAVE * WAVES) k_impl_visc(DM m, int apply_visc, int do_impl) {
  extern __shared__ double th_sh[];
  ThTile<2, COLS> tile(th_sh, m.nlm1);
  const int w = threadIdx.x >> 6, l = lane_id(), nz = l + 1;
  const int base = xcd_block() * COLS;
  constexpr bool SINGLE = (COLS == WAVES);
  int e = 0; bool wet = false;
  for (int ci = w; ci < COLS; ci += WAVES) {
    e = __builtin_amdgcn_readfirstlane(base + ci);
    const bool valid = e < m.myE;                            // no early exit: the block meets at the barriers of the sweep
    if (!valid) e = m.myE - 1;
    const int nzmin = m.ulev[e], nzmax = m.nlev[e];
    wet = valid && (nz >= nzmin && nz <= nzmax - 1);
    const int n1 = m.elem_nodes[3 * e], n2 = m.elem_nodes[3 * e + 1], n3 = m.elem_nodes[3 * e + 2];
    const double dt = m.p.dt;
    double ur = 0.0, vr = 0.0, u = 0.0, v = 0.0, he = 0.0;
    if (wet) {
      ur = DV2(m.UV_rhs, 1, nz, e); vr = DV2(m.UV_rhs, 2, nz, e);
      u = DV2(m.UV, 1, nz, e); v = DV2(m.UV, 2, nz, e); he = DA2(m.helem, nz, e);
      if (apply_visc) {
        double bs = m.p.easy_bs_return;
        ur = ur + DV2(m.U_b, 1, nz, e) - bs * (DV2(m.U_c, 1, nz, n1) + DV2(m.U_c, 1, nz, n2) + DV2(m.U_c, 1, nz, n3)) / 3.0;
        vr = vr + DV2(m.U_b, 2, nz, e) - bs * (DV2(m.U_c, 2, nz, n1) + DV2(m.U_c, 2, nz, n2) + DV2(m.U_c, 2, nz, n3)) / 3.0;
      }
    }
    if (!do_impl) {
      if (wet) { DV2(m.UV_rhs, 1, nz, e) = ur; DV2(m.UV_rhs, 2, nz, e) = vr; }
      continue;
    }
    // zbar_n, Z_n of the element column
    double zb_top = seq_sum_down(he, nzmax - 2, nzmin - 1, m.zbar_e_bot[e]);   // zbar_n(nz)
    double zb_bot = shdn(zb_top);
    if (nz == nzmax - 1) zb_bot = m.zbar_e_bot[e];
    double Zn = zb_bot + he / 2.0;                                             // Z_n(nz)
    double Zn_up = shup(Zn), Zn_dn = shdn(Zn);
    double a = 0.0, b = 1.0, c = 0.0;
    double wi_top = 0.0, wi_bot = 0.0, av_top = 0.0, av_bot = 0.0;
    if (wet) {
      wi_top = (DA2L(m.Wvel_i, nz, n1) + DA2L(m.Wvel_i, nz, n2) + DA2L(m.Wvel_i, nz, n3)) / 3.;
      wi_bot = (DA2L(m.Wvel_i, nz + 1, n1) + DA2L(m.Wvel_i, nz + 1, n2) + DA2L(m.Wvel_i, nz + 1, n3)) / 3.;
      av_top = DA2L(m.Av, nz, e); av_bot = DA2L(m.Av, nz + 1, e);
      double zinv = 1.0 * dt / (zb_top - zb_bot);
      if (nz > nzmin && nz < nzmax - 1) {
        a = -av_top / (Zn_up - Zn) * zinv;
        c = -av_bot / (Zn - Zn_dn) * zinv;
        b = -a - c + 1.0;
        a = a + dmin_(0., wi_top) * zinv;
        b = b + dmax_(0., wi_top) * zinv;
        b = b - dmin_(0., wi_bot) * zinv;
        c = c - dmax_(0., wi_bot) * zinv;
      } else if (nz == nzmax - 1 && nz != nzmin) {
        a = -av_top / (Zn_up - Zn) * zinv;
        b = -a + 1.0;
        c = 0.0;
        a = a + dmin_(0., wi_top) * zinv;
        b = b + dmax_(0., wi_top) * zinv;
      } else if (nz == nzmin) {
        c = -av_bot / (Zn - Zn_dn) * zinv;
        a = 0.0;
        b = -c + 1.0;
        b = b + wi_top * zinv;
        b = b - dmin_(0., wi_bot) * zinv;
        c = c - dmax_(0., wi_bot) * zinv;
      }
      if (nz == nzmin) {
        ur = ur + zinv * m.stress_surf[2 * e] / D_RHO0;
        vr = vr + zinv * m.stress_surf[2 * e + 1] / D_RHO0;
      }
      if (nz == nzmax - 1) {
        double friction = -m.p.C_d * sqrt(u * u + v * v);
        ur = ur + zinv * friction * u;
        vr = vr + zinv * friction * v;
      }
    }
    double u_up = shup(u), v_up = shup(v), u_dn = shdn(u), v_dn = shdn(v);
    if (wet) {
      if (nz > nzmin && nz < nzmax - 1) {
        ur = ur - a * u_up - (b - 1.0) * u - c * u_dn;
        vr = vr - a * v_up - (b - 1.0) * v - c * v_dn;
      } else if (nz == nzmin) {
        ur = ur - (b - 1.0) * u - c * u_dn;
        vr = vr - (b - 1.0) * v - c * v_dn;
      } else {
        ur = ur - a * u_up - (b - 1.0) * u;
        vr = vr - a * v_up - (b - 1.0) * v;
      }
    }
    tile.put(ci, valid, nzmin, nzmax - 1, a, b, c, ur, vr);
  }
  if (!do_impl) return;
  tile.sweep();
  for (int ci = w; ci < COLS; ci += WAVES) {
    double du, dv;
    tile.get(ci, du, dv);
    if (!SINGLE) {
      e = __builtin_amdgcn_readfirstlane(base + ci);
      wet = e < m.myE;
      if (wet) wet = nz >= m.ulev[e] && nz <= m.nlev[e] - 1;
    }
    if (wet) { DV2(m.UV_rhs, 1, nz, e) = du; DV2(m.UV_rhs, 2, nz, e) = dv; }     // UV_rhs = (du, dv), oce_ale.F90:2505-2510
  }
}
#define IV_SHAPE(id, C_, W_) case id: hipLaunchKernelGGL((k_impl_visc<C_, W_>), dim3((m.myE + C_ - 1) / C_), dim3(WAVE * W_), (ThTile<2, C_>::lds_bytes(m.nlm1)), s, m, av, di); break;
static void launch_impl_visc(const DM &m, hipStream_t s, int av, int di) {
  static const int env = getenv("FESOM_GPU_EXP_IV_SHAPE") ? atoi(getenv("FESOM_GPU_EXP_IV_SHAPE")) : 0;      // (experiments: 5 = 16 columns x 8 waves, 6 = 16 x 4)
  switch (env > 0 ? env : m.use_tile) {
    TILE_SHAPES(IV_SHAPE) IV_SHAPE(5, 16, 8) IV_SHAPE(6, 16, 4)
    default: hipLaunchKernelGGL((k_impl_visc<TH_COLS, TH_COLS>), dim3(nblocks_th(m.myE)), dim3(TH_BLOCK), (ThTile<2, TH_COLS>::lds_bytes(m.nlm1)), s, m, av, di);
  }
}
#define LAUNCH_IMPL_VISC(av, di) launch_impl_visc(m, s, av, di)

// viscosity_filter for visc_option = 8 (:196-228 -> backscatter_coef, visc_filt_dbcksc, uke_update), 10 + 2 x (smoothing rounds) launches
static void launch_visc8(const DM &m, hipStream_t s) {
  auto smooth = [&](double *arr, int nc, int rounds) {
    for (int q = 0; q < rounds; q++) {
      hipLaunchKernelGGL(k_v8_smooth_node, dim3(nblocks(m.myN)), dim3(BLOCK), 0, s, m, arr, nc, m.v8_work);
      hipLaunchKernelGGL(k_v8_smooth_elem, dim3(nblocks(m.myE)), dim3(BLOCK), 0, s, m, arr, nc, m.v8_work);
    }
  };
  hipLaunchKernelGGL(k_v8_first, dim3(nblocks(m.myE)), dim3(BLOCK), 0, s, m);
  hipLaunchKernelGGL(k_v8_tend, dim3(nblocks(m.myE)), dim3(BLOCK), 0, s, m);
  smooth(m.UV_back_tend, 2, m.p.smooth_back_tend);
  hipLaunchKernelGGL(k_v8_apply, dim3(nblocks(m.myE)), dim3(BLOCK), 0, s, m);
  smooth(m.uke_back, 1, m.p.smooth_back);
  hipLaunchKernelGGL(k_v8_unode, dim3(nblocks(m.myN)), dim3(BLOCK), 0, s, m);
  hipLaunchKernelGGL(k_v8_rosb, dim3(nblocks(m.myE)), dim3(BLOCK), 0, s, m);
  smooth(m.uke_dis, 1, m.p.smooth_dis);
  hipLaunchKernelGGL(k_v8_uke, dim3(nblocks(m.myE)), dim3(BLOCK), 0, s, m);
}

// ------------------------------------------------------------------------------------------------
// update_stiff_mat_ale (src/oce_ale.F90:1371-1470) as a gather per CSR entry: the contribution list of every
// entry (element, geometric coefficient incl. the i/j sign flips) is precomputed on the host in reference order.
__global__ void k_stiff_update(DM m) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= m.nza) return;
  double factor = D_G * m.p.dt * m.p.alpha * m.p.theta;
  double v = m.ssh_values[p];
  for (int q = m.su_ptr[p]; q < m.su_ptr[p + 1]; q++) {
    int code = m.su_elem[q];               // +-(element+1): sign carries the reference's fy=-fy flips (i==2, j==2)
    int el = (code > 0 ? code : -code) - 1;
    double fy = -m.dhe[el] * m.su_coef[q]; // coef = GS(k)*ECD(2i) - GS(3+k)*ECD(2i-1), computed on the host
    if (code < 0) fy = -fy;
    v = v + fy * factor;
  }
  m.ssh_values[p] = v;
}

// compute_ssh_rhs_ale (:1478-1572) and compute_hbar_ale (:1585-1676): per-edge vertical integrals
// (reference-order running sums), then a node gather.  mode 0: with UV_rhs and alpha; mode 1: UV only.
__global__ void __launch_bounds__(BLOCK) k_edge_transport(DM m, int mode) {
  int ed = col_id(m), l = lane_id(), nz = l + 1;
  if (ed >= m.myD) return;
  int e1 = m.edge_tri[2 * ed], e2 = m.edge_tri[2 * ed + 1];
  const double alpha = m.p.alpha;
  double t1 = 0.0, t2 = 0.0;
  int u1 = m.ulev[e1], l1 = m.nlev[e1] - 1, u2 = 1, l2 = 0;
  if (nz >= u1 && nz <= l1) {
    double uu = DV2(m.UV, 1, nz, e1), vv = DV2(m.UV, 2, nz, e1), h = DA2(m.helem, nz, e1);
    if (mode == 0) t1 = alpha * ((vv + DV2(m.UV_rhs, 2, nz, e1)) * DECD(1, ed) - (uu + DV2(m.UV_rhs, 1, nz, e1)) * DECD(2, ed)) * h;
    else t1 = (vv * DECD(1, ed) - uu * DECD(2, ed)) * h;
  }
  if (e2 >= 0) {
    u2 = m.ulev[e2]; l2 = m.nlev[e2] - 1;
    if (nz >= u2 && nz <= l2) {
      double uu = DV2(m.UV, 1, nz, e2), vv = DV2(m.UV, 2, nz, e2), h = DA2(m.helem, nz, e2);
      if (mode == 0) t2 = alpha * ((vv + DV2(m.UV_rhs, 2, nz, e2)) * DECD(3, ed) - (uu + DV2(m.UV_rhs, 1, nz, e2)) * DECD(4, ed)) * h;
      else t2 = (vv * DECD(3, ed) - uu * DECD(4, ed)) * h;
    }
  }
  // c1 = sum_{nz} t1 (top-down), c2 = -sum t2 (reference: c2 = c2 - term); one pass over the levels for both
  // (wave-uniform trip counts, one readlane pair + one add per level: the serial sums are what this kernel's time is made of)
  double c1 = 0.0, c2 = 0.0;
  u1 = __builtin_amdgcn_readfirstlane(u1); l1 = __builtin_amdgcn_readfirstlane(l1);
  u2 = __builtin_amdgcn_readfirstlane(u2); l2 = __builtin_amdgcn_readfirstlane(l2);
  for (int j = u1 - 1; j <= l1 - 1; ++j) c1 = c1 + bcast(t1, j);
  for (int j = u2 - 1; j <= l2 - 1; ++j) c2 = c2 - bcast(t2, j);
  if (l == 0) m.edge_c12[ed] = c1 + c2;
}
// The same for CORE2-class meshes (DM::use_tile).  The two vertical sums are sequential in the reference, so in the kernel above
// all 64 lanes of a wave run the same 2 x (nl-1)-step chain for ONE edge.  Here a wave takes ET_EDGES edges: their per-level
// terms go through a wave-private LDS image [t1 | t2][level][edge], then lane = edge runs the two chains for all its edges at
// once (levels outside an element's range hold +0.0: c + 0.0 == c, c - 0.0 == c for the running sums, which are never -0).
#define ET_EDGES 4                      // (measured on the 182 600-node meshes, k_edge_transport(0) / (1): 16 edges per wave 286 / 229 us, 8: 225 / 174, 4: 198 / 151, 2: 240 / 206)
#define ET_CP (ET_EDGES + 1)
__global__ void __launch_bounds__(BLOCK) k_edge_transport_tile(DM m, int mode) {
  extern __shared__ double et_sh[];
  const int w = threadIdx.x >> 6, l = lane_id(), nz = l + 1, nl1 = m.nlm1;
  double *img = et_sh + (size_t)w * 2 * nl1 * ET_CP;          // this wave's image
  const int base = (xcd_block() * COLS_PER_BLOCK + w) * ET_EDGES;
  const double alpha = m.p.alpha;
  // the index chain of all ET_EDGES edges in one round (lane k = k-th edge), then four edges' field loads in flight at a time
  int e1_l = 0, e2_l = -1, u1_l = 1, l1_l = 0, u2_l = 1, l2_l = 0;
  if (l < ET_EDGES && base + l < m.myD) {
    e1_l = m.edge_tri[2 * (base + l)]; e2_l = m.edge_tri[2 * (base + l) + 1];
    u1_l = m.ulev[e1_l]; l1_l = m.nlev[e1_l] - 1;
    if (e2_l >= 0) { u2_l = m.ulev[e2_l]; l2_l = m.nlev[e2_l] - 1; }
  }
#pragma unroll 4
  for (int k = 0; k < ET_EDGES; k++) {
    const int ed = __builtin_amdgcn_readfirstlane(base + k);
    const int e1 = rdlane(e1_l, k), e2 = rdlane(e2_l, k);
    double t1 = 0.0, t2 = 0.0;
    if (nz >= rdlane(u1_l, k) && nz <= rdlane(l1_l, k)) {            // (edges beyond myD: empty ranges)
      double uu = UV2(m.UV, 1, nz, e1), vv = UV2(m.UV, 2, nz, e1), h = UA2(m.helem, nz, e1);
      if (mode == 0) t1 = alpha * ((vv + UV2(m.UV_rhs, 2, nz, e1)) * DECD(1, ed) - (uu + UV2(m.UV_rhs, 1, nz, e1)) * DECD(2, ed)) * h;
      else t1 = (vv * DECD(1, ed) - uu * DECD(2, ed)) * h;
    }
    if (nz >= rdlane(u2_l, k) && nz <= rdlane(l2_l, k)) {
      double uu = UV2(m.UV, 1, nz, e2), vv = UV2(m.UV, 2, nz, e2), h = UA2(m.helem, nz, e2);
      if (mode == 0) t2 = alpha * ((vv + UV2(m.UV_rhs, 2, nz, e2)) * DECD(3, ed) - (uu + UV2(m.UV_rhs, 1, nz, e2)) * DECD(4, ed)) * h;
      else t2 = (vv * DECD(3, ed) - uu * DECD(4, ed)) * h;
    }
    if (nz <= nl1) { img[l * ET_CP + k] = t1; img[(nl1 + l) * ET_CP + k] = t2; }
  }
  __builtin_amdgcn_wave_barrier();
  // lane = edge: c1 = sum t1 top-down, c2 = c2 - t2 (reference: c2 = c2 - term), then c1 + c2
  const int kk = l < ET_EDGES ? l : ET_EDGES - 1;
  double c1 = 0.0, c2 = 0.0;
  for (int j = 0; j < nl1; j++) {
    c1 = c1 + img[j * ET_CP + kk];
    c2 = c2 - img[(nl1 + j) * ET_CP + kk];
  }
  if (l < ET_EDGES && base + l < m.myD) m.edge_c12[base + l] = c1 + c2;
}
static void launch_edge_transport(const DM &m, hipStream_t s, int mode) {
  // the several-edges-per-wave shape on every mesh (round 3: with 4 edges per wave it wins on pi as well, 10.6 -> 6.4 us and 10.2 -> 5.8 us for the two calls of a
  // step; FESOM_GPU_EXP_ET_TILE=0 keeps one edge per wave)
  static const int env = getenv("FESOM_GPU_EXP_ET_TILE") ? atoi(getenv("FESOM_GPU_EXP_ET_TILE")) : -1;
  if (env >= 0 ? env != 0 : true) {
    const int per_block = COLS_PER_BLOCK * ET_EDGES;
    hipLaunchKernelGGL(k_edge_transport_tile, dim3((m.myD + per_block - 1) / per_block), dim3(BLOCK), (size_t)COLS_PER_BLOCK * 2 * m.nlm1 * ET_CP * sizeof(double), s, m, mode);
  } else hipLaunchKernelGGL(k_edge_transport, dim3(nblocks(m.myD)), dim3(BLOCK), 0, s, m, mode);
}
__global__ void k_ssh_rhs_node(DM m) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= m.myN) return;
  double s = 0.0;
  for (int q = m.ne_ptr[n]; q < m.ne_ptr[n + 1]; q++) {
    double c = m.edge_c12[m.ne_idx[q]];
    s = (m.ne_sgn[q] > 0) ? s + c : s - c;
  }
  const double alpha = m.p.alpha;
  int uln = m.ulev_n[n];
  if (m.p.which_ale != 0) s = s - alpha * m.water_flux[n] * DA2L(m.areasvol, uln, n) + (1.0 - alpha) * m.ssh_rhs_old[n];
  else s = s + (1.0 - alpha) * m.ssh_rhs_old[n];
  m.ssh_rhs[n] = s;
}
// compute_hbar_ale node part + eta_n update (oce_ale.F90:2722)
__global__ void k_hbar_node(DM m) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= m.myN) return;
  double s = 0.0;
  for (int q = m.ne_ptr[n]; q < m.ne_ptr[n + 1]; q++) {
    double c = m.edge_c12[m.ne_idx[q]];
    s = (m.ne_sgn[q] > 0) ? s + c : s - c;
  }
  int uln = m.ulev_n[n];
  double asv = DA2L(m.areasvol, uln, n);
  if (m.p.which_ale != 0) s = s - m.water_flux[n] * asv;
  m.ssh_rhs_old[n] = s;
  double hb_old = m.hbar[n];
  m.hbar_old[n] = hb_old;
  double hb = hb_old + s * m.p.dt / asv;
  m.hbar[n] = hb;
  if (uln == 1) m.eta_n[n] = m.p.alpha * hb + (1.0 - m.p.alpha) * hb_old;
}
__global__ void k_dhe(DM m) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m.myE) return;
  int n1 = m.elem_nodes[3 * e], n2 = m.elem_nodes[3 * e + 1], n3 = m.elem_nodes[3 * e + 2];
  if (m.ulev[e] > 1) m.dhe[e] = 0.0;
  else m.dhe[e] = ((m.hbar[n1] - m.hbar_old[n1]) + (m.hbar[n2] - m.hbar_old[n2]) + (m.hbar[n3] - m.hbar_old[n3])) / 3.0;
}

// update_vel (src/oce_dyn.F90:101-131); the 2-D eta_n += d_eta rides along on extra threads.
__global__ void __launch_bounds__(BLOCK) k_update_vel(DM m) {
  int e = col_id(m), nz = lane_id() + 1;
  int gid = blockIdx.x * BLOCK + threadIdx.x;
  if (gid < m.N) m.eta_n[gid] = m.eta_n[gid] + m.d_eta[gid];
  if (e >= m.myE) return;
  if (nz < m.ulev[e] || nz > m.nlev[e] - 1) return;
  double fac = -D_G * m.p.theta * m.p.dt;
  double e0 = fac * m.d_eta[m.elem_nodes[3 * e]], e1 = fac * m.d_eta[m.elem_nodes[3 * e + 1]], e2 = fac * m.d_eta[m.elem_nodes[3 * e + 2]];
  double Fx = DGS(1, e) * e0 + DGS(2, e) * e1 + DGS(3, e) * e2;
  double Fy = DGS(4, e) * e0 + DGS(5, e) * e1 + DGS(6, e) * e2;
  DV2(m.UV, 1, nz, e) = DV2(m.UV, 1, nz, e) + DV2(m.UV_rhs, 1, nz, e) + Fx;
  DV2(m.UV, 2, nz, e) = DV2(m.UV, 2, nz, e) + DV2(m.UV_rhs, 2, nz, e) + Fy;
}

// ------------------------------------------------------------------------------------------------
// vert_vel_ale (src/oce_ale.F90:1692-2204; linfs + zstar branches): divergence gathered over incident
// edges, bottom-up running sum, /area, zstar distribution of d(hbar), CFL_z, explicit/implicit split.
// 20 N3 + 3 E3 values.
__global__ void __launch_bounds__(BLOCK) k_vert_vel(DM m, int fuse_hbar) {
  int n = col_id(m), l = lane_id(), nz = l + 1;
  if (n >= m.myN) return;
  const int nzmin = m.ulev_n[n], nzmax = m.nlev_n[n] - 1;
  const double dt = m.p.dt;
  double hb_new = 0.0, hb_old = 0.0;
  if (fuse_hbar) {       // node part of compute_hbar_ale + eta_n update (k_hbar_node) fused in; every lane computes the same scalars
    double sacc = 0.0;
    for (int q = m.ne_ptr[n]; q < m.ne_ptr[n + 1]; q++) {
      double c = m.edge_c12[m.ne_idx[q]];
      sacc = (m.ne_sgn[q] > 0) ? sacc + c : sacc - c;
    }
    double asv = DA2L(m.areasvol, nzmin, n);
    if (m.p.which_ale != 0) sacc = sacc - m.water_flux[n] * asv;
    hb_old = m.hbar[n];
    hb_new = hb_old + sacc * dt / asv;
    if (l == 0) {
      m.ssh_rhs_old[n] = sacc; m.hbar_old[n] = hb_old; m.hbar[n] = hb_new;
      if (nzmin == 1) m.eta_n[n] = m.p.alpha * hb_new + (1.0 - m.p.alpha) * hb_old;
    }
  } else { hb_new = m.hbar[n]; hb_old = m.hbar_old[n]; }
  // divergence of the edge transports in the layer: incident edges in list order (= reference's edge loop order).  The edge list
  // of the node is read lane-parallel (lane q = q-th incident edge: edge, sign, its triangles with their level ranges, the four
  // cross-edge coefficients), broadcast with v_readlane, and the field loads of VW_B edges are issued as one batch before the
  // ordered sums: no chain of dependent loads per edge.
  double w = 0.0;
  {
    const int q0 = m.ne_ptr[n], deg = m.ne_ptr[n + 1] - q0;
    int sg_l = 0, e1_l = 0, e2_l = 0, r1_l = 1, r2_l = 1;            // ranges packed lo | hi << 8 ; (1, 0) = empty
    double x1_l = 0.0, x2_l = 0.0, x3_l = 0.0, x4_l = 0.0;
    if (l < deg) {
      const int ed = m.ne_idx[q0 + l];
      sg_l = m.ne_sgn[q0 + l];
      e1_l = m.edge_tri[2 * ed];
      const int e2 = m.edge_tri[2 * ed + 1];
      r1_l = m.ulev[e1_l] | ((m.nlev[e1_l] - 1) << 8);
      e2_l = e2 >= 0 ? e2 : e1_l;
      r2_l = e2 >= 0 ? (m.ulev[e2] | ((m.nlev[e2] - 1) << 8)) : 1;
      x1_l = DECD(1, ed); x2_l = DECD(2, ed); x3_l = DECD(3, ed); x4_l = DECD(4, ed);
    }
    const int nzc = nz <= m.nlm1 ? nz : m.nlm1;
    constexpr int VW_B = 3;
    for (int k0 = 0; k0 < deg; k0 += VW_B) {
      double u1[VW_B], v1[VW_B], h1[VW_B], u2[VW_B], v2[VW_B], h2[VW_B];
#pragma unroll
      for (int k = 0; k < VW_B; k++) {
        const int kk = (k0 + k < deg) ? k0 + k : 0;
        const int e1 = rdlane(e1_l, kk), e2 = rdlane(e2_l, kk);
        u1[k] = DV2(m.UV, 1, nzc, e1); v1[k] = DV2(m.UV, 2, nzc, e1); h1[k] = DA2(m.helem, nzc, e1);
        u2[k] = DV2(m.UV, 1, nzc, e2); v2[k] = DV2(m.UV, 2, nzc, e2); h2[k] = DA2(m.helem, nzc, e2);
      }
#pragma unroll
      for (int k = 0; k < VW_B; k++) {
        const int kk = k0 + k;
        if (kk < deg) {
          const int r1 = rdlane(r1_l, kk), r2 = rdlane(r2_l, kk);
          const bool pos = rdlane(sg_l, kk) > 0;
          const bool on1 = nz >= (r1 & 0xff) && nz <= (r1 >> 8), on2 = nz >= (r2 & 0xff) && nz <= (r2 >> 8);
          const double c1 = (v1[k] * bcast(x1_l, kk) - u1[k] * bcast(x2_l, kk)) * h1[k];
          const double w1 = pos ? w + c1 : w - c1;
          w = on1 ? w1 : w;
          const double c2 = -(v2[k] * bcast(x3_l, kk) - u2[k] * bcast(x4_l, kk)) * h2[k];
          const double w2 = pos ? w + c2 : w - c2;
          w = on2 ? w2 : w;
        }
      }
    }
  }
  // Wvel(nz) = Wvel(nz) + Wvel(nz+1), nz = nzmax..nzmin ; Wvel(nzmax+1) = 0
  double wc = seq_sum_down((nz >= nzmin && nz <= nzmax) ? w : 0.0, nzmax - 1, nzmin - 1, 0.0);
  const bool wet = (nz >= nzmin && nz <= nzmax);
  double W = 0.0, hn_new = 0.0;
  if (wet) { W = wc / DA2L(m.area, nz, n); hn_new = DA2(m.hnode_new, nz, n); }
  if (m.p.which_ale == 2) {
    int nzm = m.nlev_n_min[n] - 1;
    if (nzmin == 1) {
      double dd1 = DA2L(m.zbar_3d_n, nzm, n);
      double dd = DA2L(m.zbar_3d_n, nzmin, n) - dd1;
      dd = (hb_new - hb_old) / dd;
      double dddt = dd / dt;
      if (nz >= nzmin && nz <= nzm - 1) {
        double zb = DA2L(m.zbar_3d_n, nz, n), zb1 = DA2L(m.zbar_3d_n, nz + 1, n);
        W = W - (zb - dd1) * dddt;
        hn_new = DA2(m.hnode, nz, n) + (zb - zb1) * dd;
        DA2(m.hnode_new, nz, n) = hn_new;
      }
    }
    if (nz == nzmin) W = W - m.water_flux[n];
  }
  if (m.p.which_ale == 1) {                    // zlevel (oce_ale.F90:1830-2023): the ssh change goes into the surface layer
    if (nzmin == 1) {
      const double dh = hb_new - hb_old;
      const int lz = m.p.lzstar_lev, k = nz - nzmin + 1;
      const bool inl = k >= 1 && k <= lz && nz <= m.nlm1;
      const double h_old = inl ? DA2(m.hnode, nz, n) : 0.0, zd = inl ? (m.zbar[nz - 1] - m.zbar[nz]) : 0.0;
      const unsigned long long ne = __ballot(inl && h_old != zd);                      // layers off their resting thickness
      if (dh < 0.0 && bcast(h_old, nzmin - 1) + dh <= bcast(zd, nzmin - 1) * m.p.min_hnode) *m.ale_flag = 1;     // the local-zstar fallback (:1859-1942) is not built
      if (dh > 0.0 && (ne & ~(1ull << (nzmin - 1))) != 0ull) {           // return to zlevel (:1950-2003): refill the sub-surface layers first
        const int nzr = (63 - __clzll((long long)ne)) - (nzmin - 1) + 1, nlm = m.nlev_n_min[n] - 2;
        const int top = nzr < nlm ? nzr : nlm;
        const double md = k == 1 ? 1000.0 : zd - h_old;
        double rest = dh, integ = 0.0;
        for (int kk = top; kk >= 1; kk--) {
          const double d = dmin_(rest, bcast(md, nzmin - 1 + kk - 1));
          rest = rest - d;
          rest = dmax_(0.0, rest);
          integ = integ + d;
          if (k == kk) { W = W - integ / dt; hn_new = h_old + d; DA2(m.hnode_new, nz, n) = hn_new; }
        }
      } else if (nz == nzmin) {
        W = W - dh / dt;
        hn_new = h_old + dh;
        DA2(m.hnode_new, nz, n) = hn_new;
      }
    }
    if (nz == nzmin) W = W - m.water_flux[n];
  }
  // CFL_z(nz) = |W(nz-1 .. )| pieces: c2 of the layer above + c1 of this layer
  double W_dn = shdn(W);                       // W(nz+1) ; W(nzmax+1) = 0
  if (nz == nzmax) W_dn = 0.0;
  double c1 = 0.0, c2 = 0.0;
  if (wet) { c1 = fabs(W * dt / hn_new); c2 = fabs(W_dn * dt / hn_new); }
  double c2_up = shup(c2);
  double cfl = 0.0;
  if (nz >= nzmin && nz <= nzmax + 1) {
    if (nz == nzmin) cfl = (nzmin > 1 ? DA2L(m.CFL_z, nz, n) : 0.0) + c1;   // only CFL_z(1,:) is reset (src/oce_ale.F90:2141): under an ice shelf the top entry accumulates over the steps, as in the reference
    else if (nz == nzmax + 1) cfl = c2_up;
    else cfl = c2_up + c1;
    double Wl = (nz == nzmax + 1) ? 0.0 : W;
    double e1 = 1.0, e2 = 0.0;
    if (m.p.w_split && (cfl > m.p.w_max_cfl)) {
      double dd = dmax_((cfl - m.p.w_max_cfl), 0.0) / dmax_(m.p.w_max_cfl, 1.e-12);
      e1 = 1.0 / (1.0 + dd);
      e2 = dd / (1.0 + dd);
    }
    DA2L(m.Wvel, nz, n) = Wl;
    DA2L(m.CFL_z, nz, n) = cfl;
    DA2L(m.Wvel_e, nz, n) = e1 * Wl;
    DA2L(m.Wvel_i, nz, n) = e2 * Wl;
  }
}

// update_thickness_ale (src/oce_ale.F90:800-993, zlevel and zstar branches)
__device__ __forceinline__ void thick_node_body(const DM &m, int n) {
  int l = lane_id(), nz = l + 1;
  if (n >= m.N) return;
  int nzmin = m.ulev_n[n], nzmax = m.nlev_n_min[n] - 2;
  if (m.p.which_ale == 1) {                   // zlevel (oce_ale.F90:883-943): the surface layer, or the layers the return to zlevel changed
    nzmin = m.ulev_n_max[n];
    if (nzmin > 1) return;
    const int lz = m.p.lzstar_lev, k = nz - nzmin + 1;
    const bool inl = k >= 1 && k <= lz && nz <= m.nlm1;
    const unsigned long long ch = __ballot(inl && (DA2(m.hnode_new, inl ? nz : 1, n) - DA2(m.hnode, inl ? nz : 1, n) != 0.0));
    int top = nzmin;
    if ((ch & ~(1ull << (nzmin - 1))) != 0ull) { top = 63 - __clzll((long long)ch) + 1; top = top < nzmax ? top : nzmax; }     // (nzmax = nlevels_nod2D_min - 2)
    nzmax = top;
  }
  if (nzmin > 1) return;
  bool in = (nz >= nzmin && nz <= nzmax);
  double hn = in ? DA2(m.hnode_new, nz, n) : 0.0;
  double zb0 = DA2L(m.zbar_3d_n, nzmax + 1, n);
  double zb = seq_sum_down(hn, nzmax - 1, nzmin - 1, zb0);   // zbar_3d_n(nz) = zbar_3d_n(nz+1) + hnode_new(nz)
  double zb_below = shdn(zb);
  if (nz == nzmax) zb_below = zb0;
  if (in) {
    DA2(m.hnode, nz, n) = hn;
    DA2L(m.zbar_3d_n, nz, n) = zb;
    DA2(m.Z_3d_n, nz, n) = zb_below + hn / 2.0;
  }
}
__device__ __forceinline__ void thick_elem_body(const DM &m, int e) {
  int nz = lane_id() + 1;
  if (e >= m.myE) return;
  int nzmin = m.ulev[e], nzmax = m.nlev[e] - 1;
  if (nzmin > 1) return;
  if (m.p.which_ale == 1) nzmax = nzmin + 1;  // zlevel (oce_ale.F90:881-884): the surface layer only
  if (nz < nzmin || nz > nzmax - 1) return;
  int n1 = m.elem_nodes[3 * e], n2 = m.elem_nodes[3 * e + 1], n3 = m.elem_nodes[3 * e + 2];
  // hnode after the update = hnode_new on every level the update touches, and the two are equal elsewhere: reading hnode_new
  // makes the element part independent of the node part, so both run in ONE launch (k_thick)
  DA2(m.helem, nz, e) = (DA2(m.hnode_new, nz, n1) + DA2(m.hnode_new, nz, n2) + DA2(m.hnode_new, nz, n3)) / 3.0;
}
__global__ void __launch_bounds__(BLOCK) k_thick_node(DM m) { thick_node_body(m, col_id(m)); }
__global__ void __launch_bounds__(BLOCK) k_thick_elem(DM m) { thick_elem_body(m, col_id(m)); }
// update_thickness_ale in one launch: the first ncolN column slots are node columns, the rest element columns
// bolus_slots > 0: the removal of the bolus velocities at the end of solve_tracers_ale (k_bolus, :165-169) rides in this launch (column slots from bolus_slots on;
// update_thickness_ale reads neither UV nor the vertical velocities)
__global__ void __launch_bounds__(BLOCK) k_thick(DM m, int ncolN, int bolus_slots) {
  const int c = col_id(m);
  if (bolus_slots > 0 && c >= bolus_slots) {
    const size_t i = (size_t)(c - bolus_slots) * WAVE + lane_id();
    if (i < (size_t)2 * m.nlm1 * m.E) m.UV[i] = m.UV[i] - m.fer_UV[i];
    if (i < (size_t)m.nl * m.N) { const double f = m.fer_Wvel[i]; m.Wvel_e[i] = m.Wvel_e[i] - f; m.Wvel[i] = m.Wvel[i] - f; }
    return;
  }
  if (c < ncolN) thick_node_body(m, c); else thick_elem_body(m, c - ncolN);
}

// ------------------------------------------------------------------------------------------------
#define LAUNCH_COL(k, ncol, ...) hipLaunchKernelGGL(k, dim3(nblocks(ncol)), dim3(BLOCK), 0, s, __VA_ARGS__)
// (measured, round 3: staging the distinct nodes of the cluster in LDS as k_kpp_smooth_u does changes nothing here, 399 -> 407 us on the basin, and neither does
//  issuing the T / S gathers of three elements as one batch, 423 -> 412 us: the kernel is bound by its arithmetic -- tanh, sqrt, the divisions of the slope)
static void launch_sigma_slope(const DM &m, hipStream_t s) { LAUNCH_COL(k_sigma_slope, m.myN, m); }
#define LAUNCH_FLAT(k, n, ...) hipLaunchKernelGGL(k, dim3(((n) + 255) / 256), dim3(256), 0, s, __VA_ARGS__)

#define IV_ATTR(id, C_, W_) (void)hipFuncSetAttribute((const void *)k_impl_visc<C_, W_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
void tile_prepare_dyn() { TILE_SHAPES(IV_ATTR) (void)hipFuncSetAttribute((const void *)k_edge_transport_tile, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute((const void *)k_pgf_tile, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
// compute_vel_rhs (mom_adv = 2) / compute_vel_rhs_vinv (mom_adv = 3) on one partition
static void launch_vel_rhs(const DM &m, hipStream_t s, int first_step) {
  if (m.p.mom_adv == 3) { LAUNCH_COL(k_vinv_ke, m.myN, m); LAUNCH_COL(k_leith_vort, m.myN, m); LAUNCH_COL(k_vinv_elem, m.myE, m, first_step); }
  else { LAUNCH_COL(k_momadv_node, m.myN, m); LAUNCH_COL(k_vel_rhs, m.myE, m, first_step); }
}
void launch_momix(const DM &m, hipStream_t s) { if (m.p.use_momix) LAUNCH_FLAT(k_momix, m.N, m); }
// h_viscosity_leith on one partition: vorticity, coefficient, two smoothing rounds
void launch_leith(const DM &m, hipStream_t s) {
  LAUNCH_COL(k_leith_vort, m.myN, m); LAUNCH_COL(k_leith_elem, m.E, m);
  for (int nt = 0; nt < 2; nt++) { LAUNCH_COL(k_leith_node, m.myN, m); LAUNCH_COL(k_leith_avg, m.myE, m); }
}
void launch_dynamics_pre(const DM &m, hipStream_t s, int first_step) {
  LAUNCH_COL(k_vel_nodes, m.myN, m);
  LAUNCH_COL(k_pressure_bv, m.N, m);
  launch_pgf(m, s);
  launch_sigma_slope(m, s);
  if (m.p.mix_scheme == 2) {
    launch_momix(m, s);
    hipLaunchKernelGGL(k_pp, dim3(nblocks(m.myE) + nblocks(m.N)), dim3(BLOCK), 0, s, m, nblocks(m.myE) * COLS_PER_BLOCK);
  }
  if (m.p.mix_scheme == 1) launch_named_kpp(m, s, "mixing_kpp");      // (launches k_momix itself)
  launch_vel_rhs(m, s, first_step);
  if (m.p.visc_option == 8) launch_visc8(m, s);
  else {
    if (m.p.visc_option <= 3) launch_leith(m, s);
    if (m.p.visc_option != 1) LAUNCH_COL(k_visc_elem, m.E, m);
    if (m.p.visc_option == 5) LAUNCH_COL(k_visc_node, m.myN, m);
    else LAUNCH_COL(k_visc_apply, m.myE, m);
  }
  LAUNCH_IMPL_VISC(m.p.visc_option == 5, m.p.i_vert_visc);
}
void launch_ssh_rhs(const DM &m, hipStream_t s) {
  if (m.p.which_ale != 0) LAUNCH_FLAT(k_stiff_update, m.nza, m);
  launch_edge_transport(m, s, 0);
  LAUNCH_FLAT(k_ssh_rhs_node, m.myN, m);
}
void launch_dynamics_post(const DM &m, hipStream_t s) {
  int ncol = m.myE > (m.N + BLOCK - 1) / BLOCK * COLS_PER_BLOCK ? m.myE : (m.N + BLOCK - 1) / BLOCK * COLS_PER_BLOCK;
  LAUNCH_COL(k_update_vel, ncol, m);
  launch_edge_transport(m, s, 1);
  LAUNCH_FLAT(k_hbar_node, m.myN, m);
  LAUNCH_FLAT(k_dhe, m.myE, m);
  LAUNCH_COL(k_vert_vel, m.myN, m, 0);
}
void launch_thickness(const DM &m, hipStream_t s, bool bolus_remove) {
  if (m.p.which_ale == 0) return;
  const int ncolN = nblocks(m.N) * COLS_PER_BLOCK;
  if (bolus_remove) {
    const int slots = (nblocks(m.N) + nblocks(m.myE)) * COLS_PER_BLOCK;
    const size_t nmax = std::max((size_t)2 * m.nlm1 * m.E, (size_t)m.nl * m.N);
    hipLaunchKernelGGL(k_thick, dim3(nblocks(m.N) + nblocks(m.myE) + nblocks((int)((nmax + WAVE - 1) / WAVE))), dim3(BLOCK), 0, s, m, ncolN, slots);
  } else hipLaunchKernelGGL(k_thick, dim3(nblocks(m.N) + nblocks(m.myE)), dim3(BLOCK), 0, s, m, ncolN, 0);
}

int launch_named_dyn(const DM &m, hipStream_t s, const char *name, int arg, int first_step) {
  (void)arg;
  if (!strcmp(name, "h_viscosity_leith")) { launch_leith(m, s); return 0; }
  // single kernels (bench.py times each one for the roofline object)
  if (!strncmp(name, "k_", 2)) {
    int ncol_uv = m.myE > (m.N + BLOCK - 1) / BLOCK * COLS_PER_BLOCK ? m.myE : (m.N + BLOCK - 1) / BLOCK * COLS_PER_BLOCK;
    if (!strcmp(name, "k_vel_nodes")) { LAUNCH_COL(k_vel_nodes, m.myN, m); return 0; }
    if (!strcmp(name, "k_pressure_bv")) { LAUNCH_COL(k_pressure_bv, m.N, m); return 0; }
    if (!strcmp(name, "k_pgf")) { launch_pgf(m, s); return 0; }
    if (!strcmp(name, "k_sigma_slope")) { launch_sigma_slope(m, s); return 0; }
    if (!strcmp(name, "k_pp")) { hipLaunchKernelGGL(k_pp, dim3(nblocks(m.myE) + nblocks(m.N)), dim3(BLOCK), 0, s, m, nblocks(m.myE) * COLS_PER_BLOCK); return 0; }
    if (!strcmp(name, "k_pp_elem")) { LAUNCH_COL(k_pp_elem, m.myE, m); return 0; }
    if (!strcmp(name, "k_pp_node_final")) { LAUNCH_COL(k_pp_node_final, m.N, m); return 0; }
    if (!strcmp(name, "k_momadv_node")) { LAUNCH_COL(k_momadv_node, m.myN, m); return 0; }
    if (!strcmp(name, "k_vel_rhs")) { LAUNCH_COL(k_vel_rhs, m.myE, m, first_step); return 0; }
    if (!strcmp(name, "k_vinv_ke")) { LAUNCH_COL(k_vinv_ke, m.myN, m); return 0; }
    if (!strcmp(name, "k_vinv_elem")) { LAUNCH_COL(k_vinv_elem, m.myE, m, first_step); return 0; }
    if (!strcmp(name, "k_vel_rhs_step")) { launch_vel_rhs(m, s, first_step); return 0; }
    if (!strcmp(name, "k_momix")) { if (m.p.use_momix) LAUNCH_FLAT(k_momix, m.N, m); return 0; }
    if (!strcmp(name, "k_visc_elem")) { LAUNCH_COL(k_visc_elem, m.E, m); return 0; }
    if (!strcmp(name, "k_leith_vort")) { LAUNCH_COL(k_leith_vort, m.myN, m); return 0; }
    if (!strcmp(name, "k_leith_elem")) { LAUNCH_COL(k_leith_elem, m.E, m); return 0; }
    if (!strcmp(name, "k_leith_node")) { LAUNCH_COL(k_leith_node, m.myN, m); return 0; }
    if (!strcmp(name, "k_leith_avg")) { LAUNCH_COL(k_leith_avg, m.myE, m); return 0; }
    if (!strcmp(name, "k_visc_node")) { LAUNCH_COL(k_visc_node, m.myN, m); return 0; }
    if (!strcmp(name, "k_visc_apply")) { LAUNCH_COL(k_visc_apply, m.myE, m); return 0; }
    if (!strcmp(name, "k_impl_visc")) { LAUNCH_IMPL_VISC(m.p.visc_option == 5, m.p.i_vert_visc); return 0; }
    if (!strcmp(name, "k_stiff_update")) { LAUNCH_FLAT(k_stiff_update, m.nza, m); return 0; }
    if (!strcmp(name, "k_edge_transport")) { launch_edge_transport(m, s, 0); return 0; }
    if (!strcmp(name, "k_edge_transport1")) { launch_edge_transport(m, s, 1); return 0; }
    if (!strcmp(name, "k_ssh_rhs_node")) { LAUNCH_FLAT(k_ssh_rhs_node, m.myN, m); return 0; }
    if (!strcmp(name, "k_update_vel")) { LAUNCH_COL(k_update_vel, ncol_uv, m); return 0; }
    if (!strcmp(name, "k_hbar_node")) { LAUNCH_FLAT(k_hbar_node, m.myN, m); return 0; }
    if (!strcmp(name, "k_dhe")) { LAUNCH_FLAT(k_dhe, m.myE, m); return 0; }
    if (!strcmp(name, "k_vert_vel")) { LAUNCH_COL(k_vert_vel, m.myN, m, 0); return 0; }
    if (!strcmp(name, "k_vert_vel_hbar")) { LAUNCH_COL(k_vert_vel, m.myN, m, 1); return 0; }
    if (!strcmp(name, "k_thick_node")) { if (m.p.which_ale != 0) LAUNCH_COL(k_thick_node, m.N, m); return 0; }          // (linfs: update_thickness_ale does nothing)
    if (!strcmp(name, "k_thick_elem")) { if (m.p.which_ale != 0) LAUNCH_COL(k_thick_elem, m.myE, m); return 0; }
    return -1;
  }
  if (!strcmp(name, "compute_vel_nodes")) { LAUNCH_COL(k_vel_nodes, m.myN, m); return 0; }
  if (!strcmp(name, "pressure_bv")) { LAUNCH_COL(k_pressure_bv, m.N, m); return 0; }       // includes sw_alpha_beta
  if (!strcmp(name, "sw_alpha_beta")) return 0;
  if (!strcmp(name, "pressure_force")) { launch_pgf(m, s); return 0; }
  if (!strcmp(name, "compute_sigma_xy")) { launch_sigma_slope(m, s); return 0; } // includes neutral slope
  if (!strcmp(name, "compute_neutral_slope")) return 0;
  if (!strcmp(name, "mixing_pp")) {
    if (m.p.use_momix) LAUNCH_FLAT(k_momix, m.N, m);                                        // mo_length of mo_convect, which is fused into k_pp
    hipLaunchKernelGGL(k_pp, dim3(nblocks(m.myE) + nblocks(m.N)), dim3(BLOCK), 0, s, m, nblocks(m.myE) * COLS_PER_BLOCK); return 0;
  }
  if (!strcmp(name, "mo_convect")) return 0;                                                  // fused into mixing_pp
  if (!strcmp(name, "compute_vel_rhs")) { launch_vel_rhs(m, s, first_step); return 0; }
  if (!strcmp(name, "visc_filt_bcksct") || !strcmp(name, "viscosity_filter")) {      // viscosity_filter(visc_option): 1 .. 8
    if (m.p.visc_option == 8) { launch_visc8(m, s); return 0; }
    if (m.p.visc_option <= 3) launch_leith(m, s);
    if (m.p.visc_option != 1) LAUNCH_COL(k_visc_elem, m.E, m);
    if (m.p.visc_option != 5) { LAUNCH_COL(k_visc_apply, m.myE, m); return 0; }
    LAUNCH_COL(k_visc_node, m.myN, m); LAUNCH_IMPL_VISC(1, 0); return 0;
  }
  if (!strcmp(name, "impl_vert_visc_ale")) { LAUNCH_IMPL_VISC(0, 1); return 0; }
  if (!strcmp(name, "update_stiff_mat_ale")) { LAUNCH_FLAT(k_stiff_update, m.nza, m); return 0; }
  if (!strcmp(name, "compute_ssh_rhs_ale")) { launch_edge_transport(m, s, 0); LAUNCH_FLAT(k_ssh_rhs_node, m.myN, m); return 0; }
  if (!strcmp(name, "update_vel")) {
    int ncol = m.myE > (m.N + BLOCK - 1) / BLOCK * COLS_PER_BLOCK ? m.myE : (m.N + BLOCK - 1) / BLOCK * COLS_PER_BLOCK;
    LAUNCH_COL(k_update_vel, ncol, m); return 0;
  }
  if (!strcmp(name, "compute_hbar_ale")) {                                                    // includes the eta_n update
    launch_edge_transport(m, s, 1); LAUNCH_FLAT(k_hbar_node, m.myN, m); LAUNCH_FLAT(k_dhe, m.myE, m); return 0;
  }
  if (!strcmp(name, "eta_update")) return 0;
  if (!strcmp(name, "vert_vel_ale")) { LAUNCH_COL(k_vert_vel, m.myN, m, 0); return 0; }
  if (!strcmp(name, "update_thickness_ale")) { launch_thickness(m, s); return 0; }
  return -1;
}
