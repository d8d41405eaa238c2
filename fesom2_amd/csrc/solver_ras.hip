// SSH solve for operators beyond the explicit inverse (CORE2-class meshes, partitions): BiCGstab on the row-scaled operator,
// right-preconditioned with a restricted additive Schwarz method whose subdomain solves are Chebyshev polynomials (gfx950).
//
// Reference: solve_ssh_ale (src/oce_ale.F90:2210-2344) -> psolve (src/psolve.c:152-221) -> pARMS BiCGstab with the RAS
// preconditioner, one subdomain per MPI rank, ILU(k) subdomain solves, factors frozen at the first matrix
// (lib/parms/src/bicgstab_ras.c:49-259, parms_ilu_vcsr.c:651-1128, psolve.c:117-150).  Kept: row scaling 1/sum|a_ij|
// (psolve.c:58-65), BiCGstab, the stop rule ||r||^2 < tol^2 on the row-scaled residual (bicgstab_ras.c:78,146,220), the frozen
// preconditioner.  Rebuilt for the GPU: triangular solves are sequential, so the subdomains are small PATCHES of the row graph
// (<= 768 owned rows + 4 rings of overlap, csrc/precond_host.cpp), one 512-thread workgroup each, and the patch solve is a
// degree-16 Chebyshev polynomial of the Jacobi-scaled frozen patch operator: the patch operator sits in registers (fp32), the
// iterate in LDS, 15 products cost one launch instead of 15 -- on a 182 600-row operator BiCGstab needs ~15 iterations of
// 7 launches instead of ~105 Jacobi iterations of 2.  No dot product inside the preconditioner: nothing to all-reduce on a
// partition either (patches never cross the rank's owned rows).
// All solver vectors live in the patch order (DM::rs_perm / rs_inv), so a patch's owned rows are one contiguous run.
// Summation orders are fixed and restated by the CPU checker of the tests: HIP == checker bitwise.
#include "dev.h"
#include "solver_dev.h"
#include "ras_host.h"
#include <string.h>

void launch_row_scale(const DM &m, hipStream_t s);

// sv_kry: 0 alpha, 1 omega, 2 beta, 3 rho, 4 rho_new, 5 ||r||^2, 6 iterations, 7 done (latched: the update of the last iteration has
// run), 8 finished (no further iteration wanted).  The update kernel tests [7], everything else [8]; [7] follows [8] in the next
// one-workgroup kernel behind the update, so no workgroup can see the flag change under its feet.
#define RAS_FINISHED(m) ((m).sv_kry[8] != 0.0)
#define RAS_DONE(m) ((m).sv_kry[7] != 0.0)

// Set-up of one solve, thread per row (natural order in, patch order out): optional node part of compute_ssh_rhs_ale
// (oce_ale.F90:1548-1570, as k_solver_setup), b = rhs * scale, extrapolated first guess, A_s in ELL [k][NP] at the row's position.
template <int W>
__global__ void k_ras_setup(DM m, int NP, int fuse_rhs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NP) return;
  const int n = m.myN;
  if (i >= n) {
#pragma unroll
    for (int k = 0; k < W; k++) m.sv_x0[(size_t)k * NP + i] = 0.0;
    return;
  }
  const int q = m.rs_inv[i];
  const int j0 = m.rowptr[i], j1 = m.rowptr[i + 1];
  const double sc = m.sv_scale[i];
  double rhs;
  if (fuse_rhs) {
    double sacc = 0.0;
    const int q0 = m.ne_ptr[i], deg = m.ne_ptr[i + 1] - q0;
    constexpr int EB = 8;
    for (int b0 = 0; b0 < deg; b0 += EB) {
      double c[EB]; int sg[EB];
#pragma unroll
      for (int k = 0; k < EB; k++) { const int qq = q0 + (b0 + k < deg ? b0 + k : 0); c[k] = m.edge_c12[m.ne_idx[qq]]; sg[k] = m.ne_sgn[qq]; }
#pragma unroll
      for (int k = 0; k < EB; k++) if (b0 + k < deg) sacc = (sg[k] > 0) ? sacc + c[k] : sacc - c[k];
    }
    const double al = m.p.alpha;
    if (m.p.which_ale != 0) sacc = sacc - al * m.water_flux[i] * m.areasvol[(size_t)i * m.nl + m.ulev_n[i] - 1] + (1.0 - al) * m.ssh_rhs_old[i];
    else sacc = sacc + (1.0 - al) * m.ssh_rhs_old[i];
    m.ssh_rhs[i] = sacc;
    rhs = sacc;
  } else rhs = m.ssh_rhs[i];
  m.sv_bn[q] = rhs * sc;
  const double xi = m.d_eta[i];
  double x0 = xi;
  if (m.sv_extrap) {
    const int nh = m.sv_info[1];
    if (m.p.solver_x0_order == 2 && nh >= 2) x0 = (3.0 * xi - 3.0 * m.sv_h1[i]) + m.sv_h2[i];
    if (m.p.solver_x0_order == 3 && nh >= 3) x0 = ((4.0 * xi - 6.0 * m.sv_h1[i]) + 4.0 * m.sv_h2[i]) - m.sv_h3[i];
    if (m.p.solver_x0_order == 3 && nh == 2) x0 = (3.0 * xi - 3.0 * m.sv_h1[i]) + m.sv_h2[i];
    m.sv_h3[i] = m.sv_h2[i]; m.sv_h2[i] = m.sv_h1[i]; m.sv_h1[i] = xi;
  }
  m.sv_x[q] = x0;
#pragma unroll
  for (int k = 0; k < W; k++) m.sv_x0[(size_t)k * NP + q] = (j0 + k < j1) ? m.ssh_values[j0 + k] * sc : 0.0;
}

template <int W>
__device__ __forceinline__ double ras_row(const DM &m, int NP, int q, const double *x) {
  double a = 0.0;
#pragma unroll
  for (int k = 0; k < W; k++) a = a + m.sv_x0[(size_t)k * NP + q] * x[m.rs_colsq[(size_t)k * NP + q]];
  return a;
}
template <int W>
__global__ void __launch_bounds__(DSB) k_ras_init(DM m, int NP, int nblk) {     // r = b - A_s x0 ; r0 = p = r ; partial ||r||^2
  const int q = blockIdx.x * DSB + threadIdx.x;
  double s[1] = {0.0};
  if (q < m.myN) {
    const double ri = m.sv_bn[q] - ras_row<W>(m, NP, q, m.sv_x);
    m.sv_r[q] = ri; m.sv_r0[q] = ri; m.sv_pd[q] = ri;
    s[0] = ri * ri;
  }
  ds_block_partials<1>(s, m.sv_part, nblk);
}
template <int W>
__global__ void __launch_bounds__(DSB) k_ras_spmv1(DM m, int NP, int nblk, int bank) {    // v = A_s p^ ; partial r0.v
  if (m.sv_kry[bank + 8] != 0.0) return;
  const int q = blockIdx.x * DSB + threadIdx.x;
  double s[1] = {0.0};
  if (q < m.myN) { const double a = ras_row<W>(m, NP, q, m.sv_ph); m.sv_v[q] = a; s[0] = m.sv_r0[q] * a; }
  ds_block_partials<1>(s, m.sv_part, nblk);
}
template <int W>
__global__ void __launch_bounds__(DSB) k_ras_spmv2(DM m, int NP, int nblk, int bank) {    // t = A_s s^ ; partial t.t, t.s, r0.t, s.s
  if (m.sv_kry[bank + 8] != 0.0) return;
  const int q = blockIdx.x * DSB + threadIdx.x;
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  if (q < m.myN) {
    const double a = ras_row<W>(m, NP, q, m.sv_sh), si = m.sv_sn[q];
    m.sv_t[q] = a;
    s[0] = a * a; s[1] = a * si; s[2] = m.sv_r0[q] * a; s[3] = si * si;
  }
  ds_block_partials<4>(s, m.sv_part, nblk);
}

// Krylov scalars from the (global) sums in sv_red: PHASE 0 after ||r0||^2, 1 after r0.v, 2 after (t.t, t.s, r0.t, s.s)
template <int PHASE>
__device__ __forceinline__ void ras_scalars(const DM &m, double tol2, int maxits) {
  double *k = m.sv_kry;
  if (PHASE == 0) {
    const double rr = m.sv_red[0];
    const bool go = (rr >= tol2 && 0 < maxits);
    k[0] = 1.0; k[1] = 1.0; k[2] = 0.0; k[3] = 1.0; k[4] = rr; k[5] = rr; k[6] = 0.0; k[7] = go ? 0.0 : 1.0; k[8] = go ? 0.0 : 1.0;
  } else if (PHASE == 1) {
    if (k[8] != 0.0) { k[7] = 1.0; return; }                  // the latch (see RAS_DONE)
    k[0] = k[4] / m.sv_red[0];
  } else {
    if (k[8] != 0.0) return;
    const double tt = m.sv_red[0], ts = m.sv_red[1], r0t = m.sv_red[2], ss = m.sv_red[3];
    const double alpha = k[0];
    const double omega = (tt > 0.0) ? ts / tt : 0.0;
    const double rho = k[4], rho_new = -omega * r0t;
    const double rr = ss - omega * (2.0 * ts - omega * tt);
    const double it = k[6] + 1.0;
    const bool more = (rr >= tol2 && it < (double)maxits);
    k[1] = omega; k[3] = rho; k[4] = rho_new; k[5] = rr; k[6] = it;
    k[2] = more ? (rho_new / rho) * (alpha / omega) : 0.0;
    k[8] = more ? 0.0 : 1.0;
  }
}
// one workgroup: block partials -> sv_red in the order of dm_sum_blocks; SCAL: the scalars too (single partition: nothing to all-reduce)
template <int NQ, int PHASE, bool SCAL>
__global__ void __launch_bounds__(DSB) k_ras_red(DM m, int nblk, double tol2, int maxits) {
  __shared__ double sh[DSB];
  if (PHASE != 0 && RAS_FINISHED(m)) {
    if (SCAL && PHASE == 1 && threadIdx.x == 0) m.sv_kry[7] = 1.0;
    return;
  }
  double tot[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) tot[q] = dm_sum_blocks(m.sv_part + (size_t)q * nblk, nblk, sh);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < NQ; q++) m.sv_red[q] = tot[q];
    if (SCAL) ras_scalars<PHASE>(m, tol2, maxits);
  }
}
template <int PHASE>
__global__ void k_ras_scal(DM m, double tol2, int maxits) { if (threadIdx.x == 0) ras_scalars<PHASE>(m, tol2, maxits); }   // partition: after the all-reduce

__global__ void __launch_bounds__(DSB) k_ras_update(DM m) {      // x += alpha p^ + omega s^ ; r = s - omega t ; p = r + beta (p - omega v)
  if (RAS_DONE(m)) return;
  const int q = blockIdx.x * DSB + threadIdx.x;
  if (q >= m.myN) return;
  const double alpha = m.sv_kry[0], omega = m.sv_kry[1], beta = m.sv_kry[2];
  const bool more = m.sv_kry[8] == 0.0;
  const double si = m.sv_sn[q], ri = si - omega * m.sv_t[q];
  m.sv_r[q] = ri;
  m.sv_x[q] = (m.sv_x[q] + alpha * m.sv_ph[q]) + omega * m.sv_sh[q];
  if (more) m.sv_pd[q] = ri + beta * (m.sv_pd[q] - omega * m.sv_v[q]);
}
// Single partition, fused form of one iteration (5 launches instead of 7): no one-workgroup kernels between the products.  Every workgroup of a consumer
// sums the block partials of its producer itself, in the order of dm_sum_blocks, and forms the Krylov scalars locally (the same operations on the same
// values in every workgroup: the same bits); workgroup 0 also stores them.  The scalar state ping-pongs between two banks of sv_kry (bank = 16 * (iteration
// & 1)): a kernel reads the bank of its iteration and writes only the other one (k_ras_update_f) or a slot nobody reads in that kernel (alpha, k_ras_apply
// MODE 1), so no workgroup sees a value change under its feet without any fence -- kernel boundaries order everything.  Slot 8 (no further iteration
// wanted) travels with the bank: iterations enqueued behind the convergence only copy the bank forward.
__global__ void __launch_bounds__(DSB) k_ras_update_f(DM m, int nblk, int bank, double tol2, int maxits) {
  __shared__ double sh[4][DSB];
  const double *kc = m.sv_kry + bank;
  double *kn = m.sv_kry + (16 - bank);
  if (kc[8] != 0.0) {                                           // converged earlier: the state moves on unchanged
    if (blockIdx.x == 0 && threadIdx.x < 16) kn[threadIdx.x] = kc[threadIdx.x];
    return;
  }
  double sums[4];
  dm_sum_blocks4(m.sv_part, nblk, sh, sums);
  const double tt = sums[0], ts = sums[1], r0t = sums[2], ss = sums[3];
  const double alpha = kc[0];
  const double omega = (tt > 0.0) ? ts / tt : 0.0;
  const double rho = kc[4], rho_new = -omega * r0t;
  const double rr = ss - omega * (2.0 * ts - omega * tt);
  const double it = kc[6] + 1.0;
  const bool more = (rr >= tol2 && it < (double)maxits);
  const double beta = more ? (rho_new / rho) * (alpha / omega) : 0.0;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    kn[0] = alpha; kn[1] = omega; kn[2] = beta; kn[3] = rho; kn[4] = rho_new; kn[5] = rr; kn[6] = it; kn[7] = more ? 0.0 : 1.0; kn[8] = more ? 0.0 : 1.0;
    m.sv_red[0] = tt; m.sv_red[1] = ts; m.sv_red[2] = r0t; m.sv_red[3] = ss;
  }
  const int q = blockIdx.x * DSB + threadIdx.x;
  if (q >= m.myN) return;
  const double si = m.sv_sn[q], ri = si - omega * m.sv_t[q];
  m.sv_r[q] = ri;
  m.sv_x[q] = (m.sv_x[q] + alpha * m.sv_ph[q]) + omega * m.sv_sh[q];
  if (more) m.sv_pd[q] = ri + beta * (m.sv_pd[q] - omega * m.sv_v[q]);
}
__global__ void __launch_bounds__(DSB) k_ras_finish(DM m, int bank) {      // back to the natural order
  const int q = blockIdx.x * DSB + threadIdx.x;
  if (q < m.myN) m.d_eta[m.rs_perm[q]] = m.sv_x[q];
  if (q == 0) {
    const double *kc = m.sv_kry + bank;
    m.sv_info[0] = (int)kc[6]; m.sv_resid[0] = sqrt(kc[5] > 0.0 ? kc[5] : 0.0);
    if (m.sv_extrap && m.sv_info[1] < 3) m.sv_info[1] = m.sv_info[1] + 1;
  }
}

// The preconditioner: z = M x, one workgroup per patch.  MODE 0: x = p -> p^ ; MODE 1: x = s = r - alpha v (stored for the owned rows) -> s^.
// Patch system (rows of the patch incl. overlap, zero outside): (I + N) z = dsc * x with N = off-diagonal a_ij / a_ii of the frozen
// operator; deg - 1 Chebyshev steps from z_1 = rhs / theta; the owned rows of z are the result (restricted additive Schwarz).
// Thread t owns the rows t, t + 512, ...: their entries (fp32) and packed LDS byte offsets stay in registers, z ping-pongs between two
// LDS images, one barrier per step.
template <int WOFF, int RPT, int MODE>
__global__ void __launch_bounds__(RAS_THREADS) k_ras_apply(DM m, int bank, int fused_nblk) {
  if (m.sv_kry[bank + 8] != 0.0) return;
  constexpr int NS = RAS_THREADS * RPT;
  __shared__ double zb0[NS], zb1[NS];                      // two images of z: a step reads one and writes the other, one barrier per step
  const int p = blockIdx.x, t = threadIdx.x;
  const int own0 = m.rs_pinfo[4 * p], no = m.rs_pinfo[4 * p + 1], eoff = m.rs_pinfo[4 * p + 2], ne = m.rs_pinfo[4 * p + 3];
  double alpha = MODE ? m.sv_kry[bank] : 0.0;
  if (MODE == 1 && fused_nblk > 0) {                       // fused form: alpha = rho / (r0 . v) from the block partials of k_ras_spmv1, summed in the order of dm_sum_blocks
    double a = 0.0;
    if (t < DSB) { for (int b = t; b < fused_nblk; b += DSB) a = a + m.sv_part[b]; zb0[t] = a; }
    __syncthreads();
    for (int s2 = DSB / 2; s2 >= 1; s2 >>= 1) {
      if (t < s2) zb0[t] = zb0[t] + zb0[t + s2];
      __syncthreads();
    }
    const double tot = zb0[0];
    __syncthreads();
    alpha = m.sv_kry[bank + 4] / tot;
    if (p == 0 && t == 0) { m.sv_kry[bank] = alpha; m.sv_red[0] = tot; }       // (slot 0 is read by the update kernel only)
  }
  const double inv_theta = m.rs_cheb[0];
  double lv[RPT][WOFF];                                    // (fp32 in memory; widened once: a conversion per use would cost more than the multiply)
  unsigned lo[RPT][WOFF];                                  // LDS byte offsets of the columns
  double rb[RPT], d[RPT], z[RPT];
#pragma unroll
  for (int j = 0; j < RPT; j++) {
    const int slot = t + j * RAS_THREADS;
    double val = 0.0;
    if (slot < ne) {
      const int q = slot < no ? own0 + slot : m.rs_extq[eoff + slot];
      val = MODE == 0 ? m.sv_pd[q] : m.sv_r[q] - alpha * m.sv_v[q];
      if (MODE == 1 && slot < no) m.sv_sn[q] = val;
    }
    rb[j] = val * m.rs_dsc[(size_t)p * NS + slot];
#pragma unroll
    for (int k = 0; k < WOFF; k++) {
      lv[j][k] = (double)m.rs_lv[((size_t)p * WOFF + k) * NS + slot];
      lo[j][k] = (unsigned)m.rs_lc[((size_t)p * WOFF + k) * NS + slot] << 3;
    }
    d[j] = rb[j] * inv_theta; z[j] = d[j];
    zb0[slot] = z[j];
  }
  __syncthreads();
  const int deg = m.rs_deg;
#define RAS_STEP(SRC, DST, K)                                                                            \
  {                                                                                                      \
    const double c1 = m.rs_cheb[1 + (K)], c2 = m.rs_cheb[64 + (K)];                                      \
    _Pragma("unroll") for (int j = 0; j < RPT; j++) {                                                    \
      double acc = 0.0;                                                                                  \
      _Pragma("unroll") for (int kk = 0; kk < WOFF; kk++) acc = acc + lv[j][kk] * *(const double *)((const char *)(SRC) + lo[j][kk]); \
      const double res = (rb[j] - z[j]) - acc;                                                           \
      d[j] = c1 * d[j] + c2 * res;                                                                       \
      z[j] = z[j] + d[j];                                                                                \
      (DST)[t + j * RAS_THREADS] = z[j];                                                                 \
    }                                                                                                    \
    __syncthreads();                                                                                     \
  }
  int k = 1;
  for (; k + 1 < deg; k += 2) {
    RAS_STEP(zb0, zb1, k)
    RAS_STEP(zb1, zb0, k + 1)
  }
  if (k < deg) RAS_STEP(zb0, zb1, k)
#undef RAS_STEP
  double *out = MODE == 0 ? m.sv_ph : m.sv_sh;
#pragma unroll
  for (int j = 0; j < RPT; j++) {
    const int slot = t + j * RAS_THREADS;
    if (slot < no) out[own0 + slot] = z[j];
  }
}

namespace {
template <int MODE>
void launch_apply(const DM &m, hipStream_t s, int bank = 0, int fused_nblk = 0) {
#define RA(WO, RP) hipLaunchKernelGGL((k_ras_apply<WO, RP, MODE>), dim3(m.rs_P), dim3(RAS_THREADS), 0, s, m, bank, fused_nblk)
#define RW(RP) do { if (m.rs_woff == 6) RA(6, RP); else if (m.rs_woff == 9) RA(9, RP); else RA(15, RP); } while (0)
  if (m.rs_rpt == 2) RW(2); else if (m.rs_rpt == 3) RW(3); else RW(4);
#undef RW
#undef RA
}
struct Shape { int W, NP, nblk; double tol2; int maxits; };
Shape shape_of(const DM &m) {
  Shape sh;
  sh.W = m.ssh_maxnnz <= 8 ? 8 : m.ssh_maxnnz <= 10 ? 10 : 16; sh.NP = (m.myN + 63) / 64 * 64; sh.nblk = (m.myN + DSB - 1) / DSB;
  sh.tol2 = m.sv_tol > 0.0 ? m.sv_tol * m.sv_tol : 1e-10 * 1e-10; sh.maxits = m.sv_maxits > 0 ? m.sv_maxits : 2000;   // bicgstab_ras.c:78,146,220 / solve_ssh_ale
  return sh;
}
}  // namespace
#define RASW(k, grid, blk, ...) do { if (sh.W == 8) hipLaunchKernelGGL(k<8>, grid, blk, 0, s, __VA_ARGS__); else if (sh.W == 10) hipLaunchKernelGGL(k<10>, grid, blk, 0, s, __VA_ARGS__); else hipLaunchKernelGGL(k<16>, grid, blk, 0, s, __VA_ARGS__); } while (0)

// one BiCGstab iteration on a single partition: 7 launches, nothing read back
static void ras_iteration(const DM &m, hipStream_t s, const Shape &sh, int it_idx, bool fused) {
  if (fused) {       // 5 launches, the scalar state in bank 16 * (it_idx & 1)
    const int bank = 16 * (it_idx & 1);
    launch_apply<0>(m, s, bank, 0);
    RASW(k_ras_spmv1, dim3(sh.nblk), dim3(DSB), m, sh.NP, sh.nblk, bank);
    launch_apply<1>(m, s, bank, sh.nblk);
    RASW(k_ras_spmv2, dim3(sh.nblk), dim3(DSB), m, sh.NP, sh.nblk, bank);
    hipLaunchKernelGGL(k_ras_update_f, dim3(sh.nblk), dim3(DSB), 0, s, m, sh.nblk, bank, sh.tol2, sh.maxits);
    return;
  }
  launch_apply<0>(m, s);
  RASW(k_ras_spmv1, dim3(sh.nblk), dim3(DSB), m, sh.NP, sh.nblk, 0);
  hipLaunchKernelGGL((k_ras_red<1, 1, true>), dim3(1), dim3(DSB), 0, s, m, sh.nblk, sh.tol2, sh.maxits);
  launch_apply<1>(m, s);
  RASW(k_ras_spmv2, dim3(sh.nblk), dim3(DSB), m, sh.NP, sh.nblk, 0);
  hipLaunchKernelGGL((k_ras_red<4, 2, true>), dim3(1), dim3(DSB), 0, s, m, sh.nblk, sh.tol2, sh.maxits);
  hipLaunchKernelGGL(k_ras_update, dim3(sh.nblk), dim3(DSB), 0, s, m);
}

// Single partition.  Iterations are enqueued in chunks; the convergence flag is read back once per solve as a rule (two iterations more
// than the last solve needed are enqueued first; launches behind the convergence are no-ops) -- results do not depend on the chunking.
int launch_solver_ras(const DM &m, hipStream_t s, int fuse_rhs, int scale_done) {
  if (!m.rs_pinfo || m.ssh_maxnnz > 16) return 1;
  const Shape sh = shape_of(m);
  if (!scale_done) launch_row_scale(m, s);
  RASW(k_ras_setup, dim3((sh.NP + 255) / 256), dim3(256), m, sh.NP, fuse_rhs);
  RASW(k_ras_init, dim3(sh.nblk), dim3(DSB), m, sh.NP, sh.nblk);
  hipLaunchKernelGGL((k_ras_red<1, 0, true>), dim3(1), dim3(DSB), 0, s, m, sh.nblk, sh.tol2, sh.maxits);
  static double *hk = nullptr;                                   // pinned copy of the scalar state
  static int last_its = 14;
  if (!hk && hipHostMalloc((void **)&hk, 32 * sizeof(double)) != hipSuccess) return 1;
  static const bool fused = !getenv("FESOM_GPU_RAS_UNFUSED");     // (the 7-launch iteration with its two one-workgroup reduction kernels, for comparison)
  int total = 0, chunk = last_its + 2;
  const double *cur = hk;
  for (;;) {
    for (int k = 0; k < chunk; k++) ras_iteration(m, s, sh, total + k, fused);
    total += chunk;
    if (hipMemcpyAsync(hk, m.sv_kry, 32 * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess) return 1;
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    cur = hk + (fused ? 16 * (total & 1) : 0);                    // the bank the last enqueued iteration wrote
    if (cur[8] != 0.0 || total >= sh.maxits) break;
    chunk = 3;
  }
  last_its = (int)cur[6];
  hipLaunchKernelGGL(k_ras_finish, dim3(sh.nblk), dim3(DSB), 0, s, m, fused ? 16 * (total & 1) : 0);
  return 0;
}

// Named phases: partitioned solve ("dsr_*", driven by fesom_gpu_step_partitioned / fesom2_amd/parallel.py with a halo exchange of the
// gathered vector before each product and an all-reduce of sv_red after it) and single kernels for the timing table ("ras_*").
int launch_named_ras(const DM &m, hipStream_t s, const char *name) {
  if (strncmp(name, "dsr_", 4) && strncmp(name, "ras_", 4)) return -1;
  if (!m.rs_pinfo) return 1;
  const Shape sh = shape_of(m);
  if (!strcmp(name, "dsr_setup")) { RASW(k_ras_setup, dim3((sh.NP + 255) / 256), dim3(256), m, sh.NP, 0); return 0; }
  if (!strcmp(name, "dsr_init")) {
    RASW(k_ras_init, dim3(sh.nblk), dim3(DSB), m, sh.NP, sh.nblk);
    hipLaunchKernelGGL((k_ras_red<1, 0, false>), dim3(1), dim3(DSB), 0, s, m, sh.nblk, sh.tol2, sh.maxits); return 0;
  }
  if (!strcmp(name, "dsr_scal_init")) { hipLaunchKernelGGL(k_ras_scal<0>, dim3(1), dim3(64), 0, s, m, sh.tol2, sh.maxits); return 0; }
  if (!strcmp(name, "dsr_scal_alpha")) { hipLaunchKernelGGL(k_ras_scal<1>, dim3(1), dim3(64), 0, s, m, sh.tol2, sh.maxits); return 0; }
  if (!strcmp(name, "dsr_scal_omega")) { hipLaunchKernelGGL(k_ras_scal<2>, dim3(1), dim3(64), 0, s, m, sh.tol2, sh.maxits); return 0; }
  if (!strcmp(name, "dsr_prec0") || !strcmp(name, "ras_apply0")) { launch_apply<0>(m, s); return 0; }
  if (!strcmp(name, "dsr_prec1") || !strcmp(name, "ras_apply1")) { launch_apply<1>(m, s); return 0; }
  if (!strcmp(name, "dsr_spmv1")) {
    RASW(k_ras_spmv1, dim3(sh.nblk), dim3(DSB), m, sh.NP, sh.nblk, 0);
    hipLaunchKernelGGL((k_ras_red<1, 1, false>), dim3(1), dim3(DSB), 0, s, m, sh.nblk, sh.tol2, sh.maxits); return 0;
  }
  if (!strcmp(name, "dsr_spmv2")) {
    RASW(k_ras_spmv2, dim3(sh.nblk), dim3(DSB), m, sh.NP, sh.nblk, 0);
    hipLaunchKernelGGL((k_ras_red<4, 2, false>), dim3(1), dim3(DSB), 0, s, m, sh.nblk, sh.tol2, sh.maxits); return 0;
  }
  if (!strcmp(name, "dsr_update")) { hipLaunchKernelGGL(k_ras_update, dim3(sh.nblk), dim3(DSB), 0, s, m); return 0; }
  if (!strcmp(name, "dsr_finish")) { hipLaunchKernelGGL(k_ras_finish, dim3(sh.nblk), dim3(DSB), 0, s, m, 0); return 0; }
  if (!strcmp(name, "ras_spmv1")) { RASW(k_ras_spmv1, dim3(sh.nblk), dim3(DSB), m, sh.NP, sh.nblk, 0); return 0; }
  if (!strcmp(name, "ras_spmv2")) { RASW(k_ras_spmv2, dim3(sh.nblk), dim3(DSB), m, sh.NP, sh.nblk, 0); return 0; }
  if (!strcmp(name, "ras_arm")) { hipMemsetAsync(m.sv_kry + 7, 0, 2 * sizeof(double), s); return 0; }      // timing: clear the flags a finished solve leaves
  return -1;
}
