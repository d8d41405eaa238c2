// SSH solver (gfx950).  Boundary: solve_ssh_ale (src/oce_ale.F90:2210-2344) -> psolve (src/psolve.c:152-221).
//
// The reference solves the row-scaled system with pARMS BiCGstab + RAS/ILU(2) (lib/parms/src/bicgstab_ras.c:49-259);
// ILU triangular solves are sequential, so this build keeps what defines the answer -- row scaling
// scale[i]=1/sum|a_ij| (psolve.c:58-65,180-188), warm start x=d_eta (psolve.c:206-212), stop when
// ||r||^2 < tol^2 with tol=1e-10 absolute on the scaled residual (bicgstab_ras.c:78,146,220), maxits 2000 --
// and replaces the preconditioner by Jacobi, applied as a column scaling: BiCGstab runs on B = A_s D^-1,
// y = D x (D = diag of the row-scaled operator A_s), so the residual b - B y is the reference's scaled residual.
// The solution agrees with the reference to solver tolerance (tests), not bit for bit.
//
// N2 is tiny (3140 rows, 21112 nnz on pi), far too small for a multi-kernel Krylov loop (8 launches per
// iteration): ONE launch, ONE 1024-thread workgroup runs the whole loop without host round trips or grid
// barriers.  The operator is stored ELL-transposed ([k][row]) so that every SpMV load is a coalesced wave read
// from L2; column indices (uint16) and the two gathered vectors p, s live in LDS when they fit (they do up to
// ~5k rows), own-row vectors are coalesced L2 traffic.  Dot products use a FIXED reduction order (1024 strided
// partial sums, halving tree) that the CPU oracle reproduces: oracle and HIP agree bitwise.
#include "dev.h"
#include "solver_dev.h"
#include <string.h>

#define ST 1024

// Block reduction of NQ quantities in a FIXED order that the CPU checker used by the tests reproduces: thread t holds the
// partial sum of rows t, t+1024, ...
//   inside a wave (DPP, no LDS traffic): in every 16-lane row  x[l] += x[l-s]  for s = 8,4,2,1 (row total in lane 15),
//   then rows: (R0+R1) and (R2+R3) (row_bcast:15), then their sum (row_bcast:31) -> wave total in lane 63;
//   the 16 wave totals go through LDS and the same 16-lane row tree, evaluated redundantly by every wave:
//   ONE barrier per reduction.
// `buf` (NQ*16 doubles) must not be reused by the next reduction (callers alternate two buffers).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  int l2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);   // lanes without a source add +0.0
  int h2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
  return x + __hiloint2double(h2, l2);
}
template <int NQ>
__device__ __forceinline__ void block_reduce(double (&v)[NQ], double *buf) {
  const int t = threadIdx.x;
#pragma unroll
  for (int q = 0; q < NQ; q++) v[q] = dpp_add<0x118, 0xf>(v[q]);   // row_shr:8
#pragma unroll
  for (int q = 0; q < NQ; q++) v[q] = dpp_add<0x114, 0xf>(v[q]);   // row_shr:4
#pragma unroll
  for (int q = 0; q < NQ; q++) v[q] = dpp_add<0x112, 0xf>(v[q]);   // row_shr:2
#pragma unroll
  for (int q = 0; q < NQ; q++) v[q] = dpp_add<0x111, 0xf>(v[q]);   // row_shr:1
#pragma unroll
  for (int q = 0; q < NQ; q++) v[q] = dpp_add<0x142, 0xa>(v[q]);   // row_bcast:15 into rows 1 and 3
#pragma unroll
  for (int q = 0; q < NQ; q++) v[q] = dpp_add<0x143, 0xc>(v[q]);   // row_bcast:31 into rows 2 and 3
  if ((t & 63) == 63) {
#pragma unroll
    for (int q = 0; q < NQ; q++) buf[q * 16 + (t >> 6)] = v[q];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NQ; q++) v[q] = buf[q * 16 + (t & 15)];      // every 16-lane row holds the 16 wave totals
#pragma unroll
  for (int q = 0; q < NQ; q++) v[q] = dpp_add<0x118, 0xf>(v[q]);
#pragma unroll
  for (int q = 0; q < NQ; q++) v[q] = dpp_add<0x114, 0xf>(v[q]);
#pragma unroll
  for (int q = 0; q < NQ; q++) v[q] = dpp_add<0x112, 0xf>(v[q]);
#pragma unroll
  for (int q = 0; q < NQ; q++) v[q] = dpp_add<0x111, 0xf>(v[q]);
#pragma unroll
  for (int q = 0; q < NQ; q++)                                       // lane 15 -> uniform
    v[q] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v[q]), 15), __builtin_amdgcn_readlane(__double2loint(v[q]), 15));
}

// Set-up of one solve on the whole GPU (thread per row, coalesced ELL writes): row scaling (psolve.c:58-65), Jacobi
// diagonal, B = A_s D^-1 in ELL [k][row], b = rhs*scale, y0 = D x0.  The column pattern (ELL, uint16) is static and
// built once at init (m.sv_cols).
// Row scale 1/sum|a_ij| and scaled diagonal of every row (psolve.c:58-65).  Depends only on the operator, so it runs
// right after k_stiff_update on a side stream, off the critical path.
__global__ void k_row_scale(DM m) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m.myN) return;
  int j0 = m.rowptr[i], j1 = m.rowptr[i + 1];
  double tmp = 0.;
  for (int j = j0; j < j1; j++) tmp += fabs(m.ssh_values[j]);
  double sc = 1. / tmp;
  m.sv_scale[i] = sc;
  const double dg = m.ssh_values[j0] * sc;
  m.sv_dinv[i] = dg;                                     // D (first entry of a row is the diagonal, oce_ale.F90:1128-1151)
  if (m.sv_rdinv) m.sv_rdinv[i] = 1.0 / dg;              // (k_solver_setup, on the critical chain, then has no divisions left)
}
void launch_row_scale(const DM &m, hipStream_t s) { hipLaunchKernelGGL(k_row_scale, dim3((m.myN + 255) / 256), dim3(256), 0, s, m); }

template <int W>
__global__ void k_solver_setup(DM m, int NP, int fuse_rhs, int sorted) {      // sorted: write in the one-workgroup solver's row order
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NP) return;
  const int n = m.myN;
  const int *rp = m.rowptr, *ci = m.colind;
  double *Bg = m.sv_vals;
  if (i >= n) {                                            // padding rows keep their position
#pragma unroll
    for (int k = 0; k < W; k++) { Bg[k * NP + i] = 0.0; if (m.sv_minv) m.sv_x0[k * NP + i] = 0.0; }
    m.sv_s[i] = 0.0;
    return;
  }
  const int q = sorted ? m.sv_inv[i] : i;                  // position of row i in the solver's row order
  int j0 = rp[i], j1 = rp[i + 1];
  double sc = m.sv_scale[i];
  double rhs;
  if (fuse_rhs) {          // node part of compute_ssh_rhs_ale (oce_ale.F90:1548-1570) fused in: gather of the edge transports
    double sacc = 0.0;
    {                                                     // batches of independent loads, the sum in the edge order of the list
      const int q0 = m.ne_ptr[i], deg = m.ne_ptr[i + 1] - q0;
      constexpr int EB = 8;
      for (int b0 = 0; b0 < deg; b0 += EB) {
        double c[EB]; int sg[EB];
#pragma unroll
        for (int k = 0; k < EB; k++) { const int qq = q0 + (b0 + k < deg ? b0 + k : 0); c[k] = m.edge_c12[m.ne_idx[qq]]; sg[k] = m.ne_sgn[qq]; }
#pragma unroll
        for (int k = 0; k < EB; k++) if (b0 + k < deg) sacc = (sg[k] > 0) ? sacc + c[k] : sacc - c[k];
      }
    }
    const double al = m.p.alpha;
    if (m.p.which_ale != 0) sacc = sacc - al * m.water_flux[i] * m.areasvol[(size_t)i * m.nl + m.ulev_n[i] - 1] + (1.0 - al) * m.ssh_rhs_old[i];
    else sacc = sacc + (1.0 - al) * m.ssh_rhs_old[i];
    m.ssh_rhs[i] = sacc;
    rhs = sacc;
  } else rhs = m.ssh_rhs[i];
  m.sv_b[q] = rhs * sc;
  double diag = m.sv_dinv[i];
  double xi = m.d_eta[i], x0 = xi;                       // initial guess: previous solution or quadratic extrapolation
  const int nh = m.sv_info[1];                          // solutions in the history (the final thread of the solve counts them up)
  if (m.sv_extrap) {
    if (m.p.solver_x0_order == 2 && nh >= 2) x0 = (3.0 * xi - 3.0 * m.sv_h1[i]) + m.sv_h2[i];
    if (m.p.solver_x0_order == 3 && nh >= 3) x0 = ((4.0 * xi - 6.0 * m.sv_h1[i]) + 4.0 * m.sv_h2[i]) - m.sv_h3[i];
    if (m.p.solver_x0_order == 3 && nh == 2) x0 = (3.0 * xi - 3.0 * m.sv_h1[i]) + m.sv_h2[i];
    m.sv_h3[i] = m.sv_h2[i]; m.sv_h2[i] = m.sv_h1[i]; m.sv_h1[i] = xi;
  }
  m.sv_s[q] = x0 * diag;                                  // y0 = D x0
  const bool xinv = m.sv_minv != nullptr;                 // explicit-inverse path: A_s, b, x0 in natural row order as well
  if (xinv) { m.sv_bn[i] = rhs * sc; m.sv_x[i] = x0; m.d_eta[i] = x0; }   // (d_eta always holds the best iterate: no finishing launch)
#pragma unroll
  for (int k = 0; k < W; k++) {
    double bk = 0.0, ak = 0.0;
    if (j0 + k < j1) {
      int c = ci[j0 + k];
      double dinv_c = m.sv_rdinv ? m.sv_rdinv[c] : 1.0 / m.sv_dinv[c];
      ak = m.ssh_values[j0 + k] * sc;                      // A_s
      bk = ak * dinv_c;                                    // B = A_s D^-1
    }
    Bg[k * NP + q] = bk;
    if (xinv) m.sv_x0[k * NP + i] = ak;
  }
}

// Register-resident variant for n <= 4*ST rows: thread t owns rows t, t+1024, t+2048, t+3072.  Own-row vectors r, v, t
// and the (pre-shifted, packed) column indices stay in registers for the whole solve; p, s, y, r~ live in LDS with a fixed
// row stride (own-row accesses are immediate-offset, conflict-free); the operator values are the only L2 traffic in
// the loop (coalesced, scalar base + lane offset).  Same arithmetic and reduction order as solver_body.
template <int W, int NP4, int WL, int WR>  // NP4: LDS stride of the vectors (>= rows); entries [0,WL) of every row in LDS, [WL,WL+WR) in registers
__global__ void __launch_bounds__(ST) k_solver_reg(DM m, int maxits, double tol2, int NP, const double *cont) {
  // cont != nullptr: safety net behind the explicit-inverse solve -- cont = its final scalar state (5 ||r||^2, 6 iterations, 7 done):
  // nothing to do if it converged; else continue from its iterate m.sv_x with the Jacobi preconditioner
  if (cont && cont[7] != 0.0) {
    if (threadIdx.x == 0 && m.sv_extrap && m.sv_info[1] < 3) m.sv_info[1] = m.sv_info[1] + 1;
    return;
  }
  extern __shared__ double lds[];
  constexpr int R = 4;
  double *bufA = lds, *bufB = lds + 64;
  double *pl = lds + 128, *sl = pl + NP4, *bl = sl + NP4;                 // gathered vectors p, s in LDS; r~ and y are only read by their owner: registers
  const unsigned t = threadIdx.x;
  const unsigned n = (unsigned)m.myN;
  const double *Bg = m.sv_vals;
  const unsigned short *cg = m.sv_cols;
  unsigned cpk[R][W / 2];                                  // two byte offsets (col*8 < 65536) per register
  bool ok[R];
  int wk[R];                                               // ELL width of this wavefront's rows in slab k (rows are sorted by width)
  double r[R], v[R], tv[R], r0[R], y[R];
  double breg[R][WR > 0 ? WR : 1];
#pragma unroll
  for (int k = 0; k < R; k++) {
    const unsigned i = t + k * ST;
    ok[k] = i < n;
    wk[k] = __builtin_amdgcn_readfirstlane(m.sv_wid[i >> 6]);
    r[k] = v[k] = tv[k] = r0[k] = y[k] = 0.0;
#pragma unroll
    for (int w2 = 0; w2 < W / 2; w2++) {
      unsigned c0 = 0, c1 = 0;
      if (ok[k]) { c0 = cg[(unsigned)(2 * w2) * NP + i]; c1 = cg[(unsigned)(2 * w2 + 1) * NP + i]; }
      cpk[k][w2] = (c0 << 3) | (c1 << 19);
    }
    if (i < (unsigned)NP4) {
      sl[i] = ok[k] ? (cont ? m.sv_x[m.sv_perm[i]] * m.sv_dinv[m.sv_perm[i]] : m.sv_s[i]) : 0.0;     // y0 = D x0 from the set-up kernel
#pragma unroll
      for (int w = 0; w < WL; w++) bl[w * NP4 + i] = ok[k] ? Bg[(size_t)w * (unsigned)NP + i] : 0.0;
      pl[i] = 0.0;
    }
#pragma unroll
    for (int w = 0; w < WR; w++) breg[k][w] = ok[k] ? Bg[(size_t)(WL + w) * (unsigned)NP + i] : 0.0;
  }
  __syncthreads();
#define SPMV_ROW(acc, vec, k)                                                                    \
  {                                                                                              \
    acc = 0.0;                                                                                   \
    unsigned o_ = t + k * ST;                                                                    \
    asm volatile("" : "+v"(o_)); /* keeps the 40 operand addresses from being hoisted out of the loop and spilled */ \
    _Pragma("unroll") for (int w = 0; w < W; w++) {                                              \
      const double *Bw = Bg + (size_t)w * (unsigned)NP;                                          \
      const unsigned c8 = (w & 1) ? (cpk[k][w >> 1] >> 16) : (cpk[k][w >> 1] & 0xffffu);         \
      if (w < wk[k]) acc = acc + (w < WL ? bl[w * NP4 + o_] : (w < WL + WR ? breg[k][w - WL < 0 ? 0 : (w - WL < WR ? w - WL : 0)] : Bw[o_])) * *(const double *)((const char *)(vec) + c8);   /* beyond the width: + 0.0 * x, skipped */ \
    }                                                                                            \
  }
  double prr = 0.0;
#pragma unroll
  for (int k = 0; k < R; k++)
    if (ok[k]) {
      const unsigned i = t + k * ST;
      double a;
      SPMV_ROW(a, sl, k);
      double ri = m.sv_b[i] - a;
      r[k] = ri; r0[k] = ri; y[k] = sl[i];
      prr = prr + ri * ri;
    }
  double rr, rho_new;
  { double q1[1] = {prr}; block_reduce<1>(q1, bufB); rr = q1[0]; rho_new = q1[0]; }
  double rho = 1.0, alpha = 1.0, omega = 1.0;
  int it = 0;
  if (rr >= tol2 && it < maxits) {
    double beta = (rho_new / rho) * (alpha / omega);
#pragma unroll
    for (int k = 0; k < R; k++)
      if (ok[k]) pl[t + k * ST] = r[k] + beta * (0.0 - omega * v[k]);
  }
  while (rr >= tol2 && it < maxits) {
    __syncthreads();                                       // p complete in LDS
    double p1 = 0.0;
#pragma unroll
    for (int k = 0; k < R; k++)
      if (ok[k]) {
        double a;
        SPMV_ROW(a, pl, k);
        v[k] = a;
        p1 = p1 + r0[k] * a;
      }
    double r0v;
    { double q1[1] = {p1}; block_reduce<1>(q1, bufA); r0v = q1[0]; }
    alpha = rho_new / r0v;
#pragma unroll
    for (int k = 0; k < R; k++)
      if (ok[k]) sl[t + k * ST] = r[k] - alpha * v[k];
    __syncthreads();                                       // s complete in LDS
    double ptt = 0.0, pts = 0.0, pr0t = 0.0, pss = 0.0;
#pragma unroll
    for (int k = 0; k < R; k++)
      if (ok[k]) {
        double a;
        SPMV_ROW(a, sl, k);
        tv[k] = a;
        double si = sl[t + k * ST];
        ptt = ptt + a * a; pts = pts + a * si; pr0t = pr0t + r0[k] * a; pss = pss + si * si;
      }
    double tt, ts, r0t, ss;
    { double q4[4] = {ptt, pts, pr0t, pss}; block_reduce<4>(q4, bufB); tt = q4[0]; ts = q4[1]; r0t = q4[2]; ss = q4[3]; }
    omega = (tt > 0.0) ? ts / tt : 0.0;
    rho = rho_new;
    rho_new = -omega * r0t;
    rr = ss - omega * (2.0 * ts - omega * tt);
    it++;
    bool more = (rr >= tol2 && it < maxits);
    double beta = more ? (rho_new / rho) * (alpha / omega) : 0.0;
#pragma unroll
    for (int k = 0; k < R; k++)
      if (ok[k]) {
        const unsigned i = t + k * ST;
        double si = sl[i], pi = pl[i];
        double ri = si - omega * tv[k];
        r[k] = ri;
        y[k] = (y[k] + alpha * pi) + omega * si;
        if (more) pl[i] = ri + beta * (pi - omega * v[k]);
      }
  }
#undef SPMV_ROW
#pragma unroll
  for (int k = 0; k < R; k++)
    if (ok[k]) { const unsigned i = t + k * ST; const int row = m.sv_perm[i]; m.d_eta[row] = y[k] * (1.0 / m.sv_dinv[row]); }
  if (t == 0) {
    m.sv_info[0] = it + (cont ? (int)cont[6] : 0); m.sv_resid[0] = sqrt(rr > 0.0 ? rr : 0.0);
    if (cont) m.sv_info[2] = m.sv_info[2] + 1;             // solves the safety net had to finish
    if (m.sv_extrap && m.sv_info[1] < 3) m.sv_info[1] = m.sv_info[1] + 1;
  }
}

void solver_prepare() {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)k_solver_reg<10, 4096, 2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_solver_reg<10, 3200, 4, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
}
int launch_solver_multi(const DM &m, hipStream_t s, int fuse_rhs, int scale_done);
int launch_solver_xinv(const DM &m, hipStream_t s, int fuse_rhs, int scale_done);
int launch_solver_ras(const DM &m, hipStream_t s, int fuse_rhs, int scale_done);
// Returns non-zero if the operator is wider than the widest instantiated ELL kernel.
int launch_solver(const DM &m, hipStream_t s, int fuse_rhs, int scale_done) {
  if (m.ssh_maxnnz > 16) return 1;
  if (m.sv_minv) return launch_solver_xinv(m, s, fuse_rhs, scale_done);
  if (m.rs_pinfo) return launch_solver_ras(m, s, fuse_rhs, scale_done);     // operators beyond the explicit inverse (solver_ras.hip)
  // one workgroup holds up to 4096 rows of <= 10 entries in registers/LDS; larger (or wider) operators take the
  // multi-workgroup phases
  if (m.myN > 4 * ST || m.ssh_maxnnz > 10) return launch_solver_multi(m, s, fuse_rhs, scale_done);
  const int W = 10, NP = (m.myN + 63) / 64 * 64;
  if (!scale_done) launch_row_scale(m, s);
  hipLaunchKernelGGL(k_solver_setup<10>, dim3((NP + 255) / 256), dim3(256), 0, s, m, NP, fuse_rhs, 1);
  const double tol2 = m.sv_tol > 0.0 ? m.sv_tol * m.sv_tol : 1e-10 * 1e-10;                     // bicgstab_ras.c:78,146,220
  const int maxits = m.sv_maxits > 0 ? m.sv_maxits : 2000;
  (void)W;
  // LDS: reduction scratch + p, s + the leading entries of every row; up to 3200 rows (pi: 3140) four of them fit, else two
  if (m.myN <= 3200) hipLaunchKernelGGL((k_solver_reg<10, 3200, 4, 3>), dim3(1), dim3(ST), (size_t)(128 + (2 + 4) * 3200) * sizeof(double), s, m, maxits, tol2, NP, (const double *)nullptr);
  else hipLaunchKernelGGL((k_solver_reg<10, 4096, 2, 3>), dim3(1), dim3(ST), (size_t)(128 + (2 + 2) * 4096) * sizeof(double), s, m, maxits, tol2, NP, (const double *)nullptr);
  return 0;
}

// =====================================================================================================================
// Partitioned SSH solve (npes > 1): the same Jacobi-scaled BiCGstab, one short kernel per phase over the rank's owned
// rows.  The host (fesom2_amd/parallel.py, or an MPI host behind the C ABI) exchanges the halo of the gathered vector
// before each SpMV and all-reduces the partial sums after it; the Krylov scalars come back through sv_scal.
// Reference: the row partition of psolve (src/psolve.c:16-115, part[]) and the halo exchange + MPI_Allreduce inside
// pARMS' bicgstab_ras (lib/parms/src/bicgstab_ras.c:49-259).  Partial sums: per block, then over blocks in block order.
// =====================================================================================================================
__global__ void k_ds_reduce(DM m, int nq, int nblk) {
  int q = threadIdx.x;
  if (q >= nq) return;
  double a = 0.0;
  for (int b = 0; b < nblk; b++) a = a + m.sv_part[(size_t)q * nblk + b];
  m.sv_red[q] = a;                                       // (rank-local partial; the ranks' sums are combined by the host all-reduce)
}
template <int W>
__device__ __forceinline__ double ds_row(const DM &m, int NP, int i, const double *x) {
  double a = 0.0;
#pragma unroll
  for (int k = 0; k < W; k++) a = a + m.sv_vals[(size_t)k * NP + i] * x[m.sv_colsi[(size_t)k * NP + i]];
  return a;
}
template <int W>
__global__ void __launch_bounds__(DSB) k_ds_init(DM m, int NP, int nblk) {
  int i = blockIdx.x * DSB + threadIdx.x;
  double q[1] = {0.0};
  if (i < m.myN) {
    double ri = m.sv_b[i] - ds_row<W>(m, NP, i, m.sv_s);
    m.sv_r[i] = ri; m.sv_r0[i] = ri; m.sv_v[i] = 0.0; m.sv_p[i] = m.sv_s[i]; m.sv_ph[i] = 0.0;
    q[0] = ri * ri;
  }
  ds_block_partials<1>(q, m.sv_part, nblk);
}
// Krylov scalars live on the device (sv_kry: 0 alpha, 1 omega, 2 beta, 3 rho, 4 rho_new, 5 ||r||^2, 6 iterations, 7 done) so that
// a stream-ordered host (RCCL on the same stream) never has to read them back inside the loop; once `done` is set every
// phase is a no-op, so polling the flag only every few iterations leaves the result unchanged.
#define KRY_DONE(m) ((m).sv_kry[7] != 0.0)
__global__ void k_ds_scal_init(DM m, double tol2, int maxits) {   // after the all-reduce of ||r0||^2
  if (threadIdx.x) return;
  double rr = m.sv_red[0];
  m.sv_kry[0] = 1.0; m.sv_kry[1] = 1.0; m.sv_kry[3] = 1.0; m.sv_kry[4] = rr; m.sv_kry[5] = rr; m.sv_kry[6] = 0.0;
  m.sv_kry[7] = (rr >= tol2 && 0 < maxits) ? 0.0 : 1.0;
  m.sv_kry[2] = (rr / 1.0) * (1.0 / 1.0);                        // beta = (rho_new/rho)*(alpha/omega)
}
__global__ void k_ds_scal_alpha(DM m) {                           // after the all-reduce of r0.v
  if (threadIdx.x || KRY_DONE(m)) return;
  m.sv_kry[0] = m.sv_kry[4] / m.sv_red[0];
}
__global__ void k_ds_scal_omega(DM m, double tol2, int maxits) {  // after the all-reduce of (t.t, t.s, r0.t, s.s)
  if (threadIdx.x || KRY_DONE(m)) return;
  const double tt = m.sv_red[0], ts = m.sv_red[1], r0t = m.sv_red[2], ss = m.sv_red[3];
  const double alpha = m.sv_kry[0];
  const double omega = (tt > 0.0) ? ts / tt : 0.0;
  const double rho = m.sv_kry[4], rho_new = -omega * r0t;
  const double rr = ss - omega * (2.0 * ts - omega * tt);
  const double it = m.sv_kry[6] + 1.0;
  m.sv_kry[1] = omega; m.sv_kry[3] = rho; m.sv_kry[4] = rho_new; m.sv_kry[5] = rr; m.sv_kry[6] = it;
  const bool more = (rr >= tol2 && it < (double)maxits);
  m.sv_kry[2] = more ? (rho_new / rho) * (alpha / omega) : 0.0;
  m.sv_kry[8] = more ? 0.0 : 1.0;                                // becomes `done` after this iteration's update (k_ds_update)
}
__global__ void __launch_bounds__(DSB) k_ds_p(DM m) {             // p = r + beta (p - omega v)
  int i = blockIdx.x * DSB + threadIdx.x;
  if (i >= m.myN || KRY_DONE(m)) return;
  const double beta = m.sv_kry[2], omega = m.sv_kry[1];
  m.sv_ph[i] = m.sv_r[i] + beta * (m.sv_ph[i] - omega * m.sv_v[i]);
}
template <int W>
__global__ void __launch_bounds__(DSB) k_ds_spmv1(DM m, int NP, int nblk) {   // v = B p ; r0.v
  int i = blockIdx.x * DSB + threadIdx.x;
  double q[1] = {0.0};
  if (i < m.myN && !KRY_DONE(m)) { double a = ds_row<W>(m, NP, i, m.sv_ph); m.sv_v[i] = a; q[0] = m.sv_r0[i] * a; }
  ds_block_partials<1>(q, m.sv_part, nblk);
}
__global__ void __launch_bounds__(DSB) k_ds_s(DM m) {             // s = r - alpha v
  int i = blockIdx.x * DSB + threadIdx.x;
  if (i >= m.myN || KRY_DONE(m)) return;
  m.sv_s[i] = m.sv_r[i] - m.sv_kry[0] * m.sv_v[i];
}
template <int W>
__global__ void __launch_bounds__(DSB) k_ds_spmv2(DM m, int NP, int nblk) {   // t = B s ; t.t, t.s, r0.t, s.s
  int i = blockIdx.x * DSB + threadIdx.x;
  double q[4] = {0.0, 0.0, 0.0, 0.0};
  if (i < m.myN && !KRY_DONE(m)) {
    double a = ds_row<W>(m, NP, i, m.sv_s), si = m.sv_s[i];
    m.sv_t[i] = a;
    q[0] = a * a; q[1] = a * si; q[2] = m.sv_r0[i] * a; q[3] = si * si;
  }
  ds_block_partials<4>(q, m.sv_part, nblk);
}
__global__ void __launch_bounds__(DSB) k_ds_update(DM m) {        // y += alpha p + omega s ; r = s - omega t
  int i = blockIdx.x * DSB + threadIdx.x;
  if (KRY_DONE(m)) return;
  if (i < m.myN) {
    const double alpha = m.sv_kry[0], omega = m.sv_kry[1];
    double si = m.sv_s[i];
    m.sv_r[i] = si - omega * m.sv_t[i];
    m.sv_p[i] = (m.sv_p[i] + alpha * m.sv_ph[i]) + omega * si;
  }
}
__global__ void k_ds_latch(DM m) { if (!threadIdx.x && m.sv_kry[8] != 0.0) m.sv_kry[7] = 1.0; }   // after k_ds_update of the last iteration
__global__ void __launch_bounds__(DSB) k_ds_finish(DM m) {        // x = D^-1 y
  int i = blockIdx.x * DSB + threadIdx.x;
  if (i < m.myN) m.d_eta[i] = m.sv_p[i] * (1.0 / m.sv_dinv[i]);
  if (i == 0) {
    m.sv_info[0] = (int)m.sv_kry[6]; m.sv_resid[0] = sqrt(m.sv_kry[5] > 0.0 ? m.sv_kry[5] : 0.0);
    if (m.sv_extrap && m.sv_info[1] < 3) m.sv_info[1] = m.sv_info[1] + 1;
  }
}

// named phases of the Jacobi-preconditioned partitioned solve (fesom_gpu_call; solver_precond = 0 -- the default on a partition is the
// RAS-Chebyshev preconditioner, solver_ras.hip "dsr_*"): ds_scale, ds_setup, ds_init, ds_scal_init, ds_p, ds_spmv1,
// ds_scal_alpha, ds_s, ds_spmv2, ds_scal_omega, ds_update, ds_finish
int launch_named_xi(const DM &m, hipStream_t s, const char *name);
int launch_named_dsolve(const DM &m, hipStream_t s, const char *name) {
  if (!strncmp(name, "xi_", 3)) return launch_named_xi(m, s, name);
  if (strncmp(name, "ds_", 3)) return -1;
  // ELL width of the phases: 8 / 10 / 16 slabs (the column pattern sv_colsi holds >= that many; narrower kernels skip padding slabs)
  const int W = m.ssh_maxnnz <= 8 ? 8 : m.ssh_maxnnz <= 10 ? 10 : 16, NP = (m.myN + 63) / 64 * 64, nblk = (m.myN + DSB - 1) / DSB;
  if (m.ssh_maxnnz > 16) return 1;
#define DSW(k, ...) do { if (W == 8) hipLaunchKernelGGL(k<8>, dim3(nblk), dim3(DSB), 0, s, __VA_ARGS__); else if (W == 10) hipLaunchKernelGGL(k<10>, dim3(nblk), dim3(DSB), 0, s, __VA_ARGS__); else hipLaunchKernelGGL(k<16>, dim3(nblk), dim3(DSB), 0, s, __VA_ARGS__); } while (0)
  if (!strcmp(name, "ds_scale")) { launch_row_scale(m, s); return 0; }
  if (!strcmp(name, "ds_setup")) {
    if (W == 8) hipLaunchKernelGGL(k_solver_setup<8>, dim3((NP + 255) / 256), dim3(256), 0, s, m, NP, 0, 0);
    else if (W == 10) hipLaunchKernelGGL(k_solver_setup<10>, dim3((NP + 255) / 256), dim3(256), 0, s, m, NP, 0, 0);
    else hipLaunchKernelGGL(k_solver_setup<16>, dim3((NP + 255) / 256), dim3(256), 0, s, m, NP, 0, 0);
    return 0;
  }
  const double tol2 = m.sv_tol > 0.0 ? m.sv_tol * m.sv_tol : 1e-10 * 1e-10; const int maxits = m.sv_maxits > 0 ? m.sv_maxits : 2000;      // bicgstab_ras.c:78,146,220 / solve_ssh_ale
  if (!strcmp(name, "ds_init")) {
    hipMemsetAsync(m.sv_kry, 0, 16 * sizeof(double), s);
    DSW(k_ds_init, m, NP, nblk); hipLaunchKernelGGL(k_ds_reduce, dim3(1), dim3(64), 0, s, m, 1, nblk); return 0;
  }
  if (!strcmp(name, "ds_scal_init")) { hipLaunchKernelGGL(k_ds_scal_init, dim3(1), dim3(64), 0, s, m, tol2, maxits); return 0; }
  if (!strcmp(name, "ds_scal_alpha")) { hipLaunchKernelGGL(k_ds_scal_alpha, dim3(1), dim3(64), 0, s, m); return 0; }
  if (!strcmp(name, "ds_scal_omega")) { hipLaunchKernelGGL(k_ds_scal_omega, dim3(1), dim3(64), 0, s, m, tol2, maxits); return 0; }
  if (!strcmp(name, "ds_p")) { hipLaunchKernelGGL(k_ds_p, dim3(nblk), dim3(DSB), 0, s, m); return 0; }
  if (!strcmp(name, "ds_spmv1")) { DSW(k_ds_spmv1, m, NP, nblk); hipLaunchKernelGGL(k_ds_reduce, dim3(1), dim3(64), 0, s, m, 1, nblk); return 0; }
  if (!strcmp(name, "ds_s")) { hipLaunchKernelGGL(k_ds_s, dim3(nblk), dim3(DSB), 0, s, m); return 0; }
  if (!strcmp(name, "ds_spmv2")) { DSW(k_ds_spmv2, m, NP, nblk); hipLaunchKernelGGL(k_ds_reduce, dim3(1), dim3(64), 0, s, m, 4, nblk); return 0; }
  if (!strcmp(name, "ds_update")) {
    hipLaunchKernelGGL(k_ds_update, dim3(nblk), dim3(DSB), 0, s, m);
    hipLaunchKernelGGL(k_ds_latch, dim3(1), dim3(64), 0, s, m); return 0;
  }
  if (!strcmp(name, "ds_finish")) { hipLaunchKernelGGL(k_ds_finish, dim3(nblk), dim3(DSB), 0, s, m); return 0; }
  return -1;
}

// Single GPU, operator too large for one workgroup (> 4096 rows): the same recurrences and the same summation order as the
// phases above, 2 launches per iteration (k_dm_upd_spmv1, k_dm_spmv2) instead of 9 -- there is no host all-reduce between the phases here, so every block
// sums the block partials itself (in block order) and evaluates the Krylov scalars redundantly.  The scalar state ping-pongs
// between two slots of sv_kry (a block must not read what block 0 of the same launch is about to write); the convergence
// flag is read back between chunks of iterations (launches after convergence are no-ops).
//   state slot (16 doubles): 0 alpha, 1 omega, 2 beta, 3 rho, 4 rho_new, 5 ||r||^2, 6 iterations, 7 done
// sum of the nblk block partials, evaluated by every block in the same fixed order: thread t adds part[t], part[t+256], ...
// in that order, then the halving tree over the 256 threads (strides 128..1).  Must be called by the whole block.
__global__ void __launch_bounds__(DSB) k_dm_start(DM m, int nblk, double tol2, int maxits) {   // after k_ds_init: state + first p
  __shared__ double sh[DSB];
  const double rr = dm_sum_blocks(m.sv_part, nblk, sh);
  const bool go = (rr >= tol2 && 0 < maxits);
  int i = blockIdx.x * DSB + threadIdx.x;
  if (go && i < m.myN) m.sv_ph[i] = m.sv_r[i] + ((rr / 1.0) * (1.0 / 1.0)) * (m.sv_ph[i] - 1.0 * m.sv_v[i]);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double *st = m.sv_kry;                                 // slot 0
    st[0] = 1.0; st[1] = 1.0; st[2] = rr; st[3] = 1.0; st[4] = rr; st[5] = rr; st[6] = 0.0; st[7] = go ? 0.0 : 1.0;
  }
}
template <int W>
__global__ void __launch_bounds__(DSB) k_dm_spmv1(DM m, int NP, int nblk, int slot) {          // v = B p^ ; r0.v
  const double *st = m.sv_kry + 16 * slot;
  const bool done = st[7] != 0.0;
  int i = blockIdx.x * DSB + threadIdx.x;
  double q[1] = {0.0};
  if (i < m.myN && !done) { double a = ds_row<W>(m, NP, i, m.sv_ph); m.sv_v[i] = a; q[0] = m.sv_r0[i] * a; }
  ds_block_partials<1>(q, m.sv_part, nblk);
}
// alpha ; s = r - alpha v ; t = B s ; t.t, t.s, r0.t, s.s.  s at the neighbour columns is evaluated on the fly from r and v (the
// same expression, hence the same bits, as the stored s_i), which saves the launch that used to sit between the two products.
// The partials go to the second half of sv_part: other blocks may still be summing the r0.v partials of the first half.
template <int W>
__global__ void __launch_bounds__(DSB) k_dm_spmv2(DM m, int NP, int nblk, int slot) {
  const double *st = m.sv_kry + 16 * slot;
  const bool done = st[7] != 0.0;
  __shared__ double sh[DSB];
  double *part2 = m.sv_part + 4 * (size_t)nblk;
  int i = blockIdx.x * DSB + threadIdx.x;
  double q[4] = {0.0, 0.0, 0.0, 0.0};
  if (!done) {
    // the row's operands do not depend on alpha: their loads are issued before the block sum (whose barriers the compiler will not move loads across)
    double av[W], rj[W], vj[W], r_i = 0.0, v_i = 0.0, r0_i = 0.0;
    if (i < m.myN) {
#pragma unroll
      for (int k = 0; k < W; k++) {
        const int j = m.sv_colsi[(size_t)k * NP + i];
        av[k] = m.sv_vals[(size_t)k * NP + i]; rj[k] = m.sv_r[j]; vj[k] = m.sv_v[j];
      }
      r_i = m.sv_r[i]; v_i = m.sv_v[i]; r0_i = m.sv_r0[i];
    }
    const double alpha = st[4] / dm_sum_blocks(m.sv_part, nblk, sh);
    if (i < m.myN) {
      double a = 0.0;
#pragma unroll
      for (int k = 0; k < W; k++) a = a + av[k] * (rj[k] - alpha * vj[k]);
      double si = r_i - alpha * v_i;
      m.sv_s[i] = si; m.sv_t[i] = a;
      q[0] = a * a; q[1] = a * si; q[2] = r0_i * a; q[3] = si * si;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) m.sv_kry[32] = alpha;
  }
  ds_block_partials<4>(q, part2, nblk);
}
__global__ void __launch_bounds__(DSB) k_dm_upd(DM m, int nblk, int slot, double tol2, int maxits) {  // scalars ; y, r ; next p
  const double *st = m.sv_kry + 16 * slot;
  double *so = m.sv_kry + 16 * (1 - slot);
  if (st[7] != 0.0) {
    if (blockIdx.x == 0 && threadIdx.x < 16) so[threadIdx.x] = st[threadIdx.x];
    return;
  }
  __shared__ double sh[4][DSB];
  const double *part2 = m.sv_part + 4 * (size_t)nblk;
  double tot[4];
  dm_sum_blocks4(part2, nblk, sh, tot);
  const double tt = tot[0], ts = tot[1], r0t = tot[2], ss = tot[3];
  const double alpha = m.sv_kry[32];
  const double omega = (tt > 0.0) ? ts / tt : 0.0;
  const double rho = st[4], rho_new = -omega * r0t;
  const double rr = ss - omega * (2.0 * ts - omega * tt);
  const double it = st[6] + 1.0;
  const bool more = (rr >= tol2 && it < (double)maxits);
  const double beta = more ? (rho_new / rho) * (alpha / omega) : 0.0;
  int i = blockIdx.x * DSB + threadIdx.x;
  if (i < m.myN) {
    double si = m.sv_s[i], pi = m.sv_ph[i];
    double ri = si - omega * m.sv_t[i];
    m.sv_r[i] = ri;
    m.sv_p[i] = (m.sv_p[i] + alpha * pi) + omega * si;
    if (more) m.sv_ph[i] = ri + beta * (pi - omega * m.sv_v[i]);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    so[0] = alpha; so[1] = omega; so[2] = beta; so[3] = rho; so[4] = rho_new; so[5] = rr; so[6] = it; so[7] = more ? 0.0 : 1.0;
  }
}
// k_dm_upd of iteration k and k_dm_spmv1 of iteration k+1 in ONE launch: the new p^ at the neighbour columns is evaluated on the fly from s, t, the old p^ and
// the old v (the same expressions, hence the same bits, as the stored p^_i), so the product does not have to wait for a launch boundary behind the update.
// p^ and v are double-buffered (other workgroups still read the old ones): 2 launches per iteration instead of 3.
template <int W>
__global__ void __launch_bounds__(DSB) k_dm_upd_spmv1(DM m, int NP, int nblk, int slot, double tol2, int maxits, const double *ph_old, const double *v_old,
                                                      double *ph_new, double *v_new) {
  const double *st = m.sv_kry + 16 * slot;
  double *so = m.sv_kry + 16 * (1 - slot);
  if (st[7] != 0.0) {
    if (blockIdx.x == 0 && threadIdx.x < 16) so[threadIdx.x] = st[threadIdx.x];
    return;
  }
  __shared__ double sh[4][DSB];
  const double *part2 = m.sv_part + 4 * (size_t)nblk;
  const int i = blockIdx.x * DSB + threadIdx.x;
  // operands first (independent of the Krylov scalars), then the block sums
  double av[W], sj[W], tj[W], pj[W], vj[W], s_i = 0.0, t_i = 0.0, p_i = 0.0, v_i = 0.0, y_i = 0.0, r0_i = 0.0;
  if (i < m.myN) {
#pragma unroll
    for (int k = 0; k < W; k++) {
      const int j = m.sv_colsi[(size_t)k * NP + i];
      av[k] = m.sv_vals[(size_t)k * NP + i]; sj[k] = m.sv_s[j]; tj[k] = m.sv_t[j]; pj[k] = ph_old[j]; vj[k] = v_old[j];
    }
    s_i = m.sv_s[i]; t_i = m.sv_t[i]; p_i = ph_old[i]; v_i = v_old[i]; y_i = m.sv_p[i]; r0_i = m.sv_r0[i];
  }
  double tot[4];
  dm_sum_blocks4(part2, nblk, sh, tot);
  const double tt = tot[0], ts = tot[1], r0t = tot[2], ss = tot[3];
  const double alpha = m.sv_kry[32];
  const double omega = (tt > 0.0) ? ts / tt : 0.0;
  const double rho = st[4], rho_new = -omega * r0t;
  const double rr = ss - omega * (2.0 * ts - omega * tt);
  const double it = st[6] + 1.0;
  const bool more = (rr >= tol2 && it < (double)maxits);
  const double beta = more ? (rho_new / rho) * (alpha / omega) : 0.0;
  double q[1] = {0.0};
  if (i < m.myN) {
    const double si = s_i, pi = p_i;
    const double ri = si - omega * t_i;
    m.sv_r[i] = ri;
    m.sv_p[i] = (y_i + alpha * pi) + omega * si;
    if (more) {
      ph_new[i] = ri + beta * (pi - omega * v_i);
      double a = 0.0;
#pragma unroll
      for (int k = 0; k < W; k++) {
        const double rj = sj[k] - omega * tj[k];
        a = a + av[k] * (rj + beta * (pj[k] - omega * vj[k]));
      }
      v_new[i] = a; q[0] = r0_i * a;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    so[0] = alpha; so[1] = omega; so[2] = beta; so[3] = rho; so[4] = rho_new; so[5] = rr; so[6] = it; so[7] = more ? 0.0 : 1.0;
  }
  if (more) ds_block_partials<1>(q, m.sv_part, nblk);             // (`more` is the same in every workgroup)
}
__global__ void __launch_bounds__(DSB) k_dm_finish(DM m, int slot) {
  const double *st = m.sv_kry + 16 * slot;
  int i = blockIdx.x * DSB + threadIdx.x;
  if (i < m.myN) m.d_eta[i] = m.sv_p[i] * (1.0 / m.sv_dinv[i]);
  if (i == 0) {
    m.sv_info[0] = (int)st[6]; m.sv_resid[0] = sqrt(st[5] > 0.0 ? st[5] : 0.0);
    if (m.sv_extrap && m.sv_info[1] < 3) m.sv_info[1] = m.sv_info[1] + 1;
  }
}

int launch_solver_multi(const DM &m, hipStream_t s, int fuse_rhs, int scale_done) {
  if (m.ssh_maxnnz > 16) return 1;
  // ELL width of the phases: 8 / 10 / 16 slabs (the column pattern sv_colsi holds >= that many; narrower kernels skip padding slabs)
  const int W = m.ssh_maxnnz <= 8 ? 8 : m.ssh_maxnnz <= 10 ? 10 : 16, NP = (m.myN + 63) / 64 * 64, nblk = (m.myN + DSB - 1) / DSB;
  const double tol2 = m.sv_tol > 0.0 ? m.sv_tol * m.sv_tol : 1e-10 * 1e-10; const int maxits = m.sv_maxits > 0 ? m.sv_maxits : 2000;
  if (!scale_done) launch_row_scale(m, s);
  if (W == 8) hipLaunchKernelGGL(k_solver_setup<8>, dim3((NP + 255) / 256), dim3(256), 0, s, m, NP, fuse_rhs, 0);
  else if (W == 10) hipLaunchKernelGGL(k_solver_setup<10>, dim3((NP + 255) / 256), dim3(256), 0, s, m, NP, fuse_rhs, 0);
  else hipLaunchKernelGGL(k_solver_setup<16>, dim3((NP + 255) / 256), dim3(256), 0, s, m, NP, fuse_rhs, 0);
  static double *hk = nullptr;                                   // pinned copy of the scalar state
  static int last_its = 24;
  if (!hk && hipHostMalloc((void **)&hk, 16 * sizeof(double)) != hipSuccess) return 1;
#define DMW(k, ...) do { if (W == 8) hipLaunchKernelGGL(k<8>, dim3(nblk), dim3(DSB), 0, s, __VA_ARGS__); else if (W == 10) hipLaunchKernelGGL(k<10>, dim3(nblk), dim3(DSB), 0, s, __VA_ARGS__); else hipLaunchKernelGGL(k<16>, dim3(nblk), dim3(DSB), 0, s, __VA_ARGS__); } while (0)
  DMW(k_ds_init, m, NP, nblk);
  hipLaunchKernelGGL(k_dm_start, dim3(nblk), dim3(DSB), 0, s, m, nblk, tol2, maxits);
  // one read-back of the convergence flag per solve as a rule: a few iterations more than the last solve needed (launches after
  // convergence are no-ops of ~2 us; a second read-back costs a host round trip of 30-50 us)
  int slot = 0, total = 0, chunk = last_its + 6;
  const char *e3 = getenv("FESOM_GPU_SOLVER_3LAUNCH");           // the three-launch iteration (same bits; kept for the comparison test)
  const bool three = e3 && atoi(e3) != 0;
  DM mm[2] = {m, m};                                              // the two (p^, v) buffer pairs
  mm[1].sv_ph = m.sv_ph2; mm[1].sv_v = m.sv_v2; mm[1].sv_ph2 = m.sv_ph; mm[1].sv_v2 = m.sv_v;
  int cur = 0;
  for (;;) {
    if (three) {
      for (int k = 0; k < chunk; k++) {
        DMW(k_dm_spmv1, m, NP, nblk, slot);
        DMW(k_dm_spmv2, m, NP, nblk, slot);
        hipLaunchKernelGGL(k_dm_upd, dim3(nblk), dim3(DSB), 0, s, m, nblk, slot, tol2, maxits);
        slot = 1 - slot;
      }
    } else {
      // spmv1 spmv2 { upd+spmv1 spmv2 } x (chunk-1) upd
      DMW(k_dm_spmv1, mm[cur], NP, nblk, slot);
      DMW(k_dm_spmv2, mm[cur], NP, nblk, slot);
      for (int k = 1; k < chunk; k++) {
        DMW(k_dm_upd_spmv1, mm[cur], NP, nblk, slot, tol2, maxits, (const double *)mm[cur].sv_ph, (const double *)mm[cur].sv_v, mm[1 - cur].sv_ph, mm[1 - cur].sv_v);
        cur = 1 - cur; slot = 1 - slot;
        DMW(k_dm_spmv2, mm[cur], NP, nblk, slot);
      }
      hipLaunchKernelGGL(k_dm_upd, dim3(nblk), dim3(DSB), 0, s, mm[cur], nblk, slot, tol2, maxits);
      slot = 1 - slot;
    }
    total += chunk;
    if (hipMemcpyAsync(hk, m.sv_kry + 16 * slot, 16 * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess) return 1;
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    if (hk[7] != 0.0 || total >= maxits) break;
    chunk = 8;
  }
  last_its = (int)hk[6];
  hipLaunchKernelGGL(k_dm_finish, dim3(nblk), dim3(DSB), 0, s, m, slot);
  return 0;
}


// =====================================================================================================================
// pi-class operators (single partition, <= 4096 rows): BiCGstab preconditioned with the EXPLICIT INVERSE of the frozen
// row-scaled operator (csrc/precond_host.cpp; the reference freezes its ILU(2) factors the same way, psolve.c:117-150).
// One workgroup is LDS-gather-bound at ~6.6 us per Jacobi iteration and needs 20-30 of them.  The inverse of this Helmholtz-type
// operator decays exponentially away from the diagonal, so it is kept as a SPARSE fp32 matrix (entries above 1e-4 of their row's
// largest: 82 per row on pi, 1.5 MB, L2-resident) that all CUs apply in one short launch (k_xi_gemv: wave per row, vector in LDS,
// fp64 accumulation); BiCGstab converges in 1-2 iterations, and the whole solve is a fixed number of stream-ordered launches
// with no host read-back:
//     setup | init | K x { gemv(p^ = M p) , v = A p^ , gemv(s^ = M s) , t = A s^ , update } | finish | safety net
// Right preconditioning: the residual b - A_s x is the reference's row-scaled residual and the stop rule is unchanged
// (||r||^2 < 1e-20, bicgstab_ras.c:78,146,220).  Launches after convergence are no-ops; if K iterations do not suffice the
// Jacobi one-workgroup solver above continues from the current iterate (k_solver_reg, cont), so the result is always converged.
// Summation orders are fixed (per-thread sequential, DPP wave tree, wave partials in order; block partials as in the
// multi-workgroup phases) and restated by the CPU checker of the tests.
// =====================================================================================================================
#define XI_SPINUP 300                   // solves after init that enqueue two explicit-inverse iterations under the default schedule (solver_xinv_its = 0)
#define XI_ROWS 16                       // rows per 256-thread block of the preconditioner kernel (4 per wavefront)
__device__ __forceinline__ double xi_wave_total(double x) {           // same lane tree as block_reduce: total in lane 63
  x = dpp_add<0x118, 0xf>(x); x = dpp_add<0x114, 0xf>(x); x = dpp_add<0x112, 0xf>(x); x = dpp_add<0x111, 0xf>(x);
  x = dpp_add<0x142, 0xa>(x); x = dpp_add<0x143, 0xc>(x);
  return x;
}
// sum of nblk <= 16 block partials in the order of dm_sum_blocks (thread b holds part[b], halving tree: the strides 128..16 only add
// zeros), evaluated by every thread in registers: no barrier
__device__ __forceinline__ double xi_sum16(const double *part, int nblk) {
  double p[16];
#pragma unroll
  for (int b = 0; b < 16; b++) p[b] = b < nblk ? part[b] : 0.0;
#pragma unroll
  for (int b = 0; b < 8; b++) p[b] = p[b] + p[b + 8];
#pragma unroll
  for (int b = 0; b < 4; b++) p[b] = p[b] + p[b + 4];
  p[0] = p[0] + p[2]; p[1] = p[1] + p[3];
  return p[0] + p[1];
}
template <int W>
__global__ void __launch_bounds__(DSB) k_xi_init(DM m, int NP, int nblk) {    // r = b - A_s x0 ; r0 = p = r ; partial ||r||^2
  int i = blockIdx.x * DSB + threadIdx.x;
  double q[1] = {0.0};
  if (i < m.myN) {
    double a = 0.0;
#pragma unroll
    for (int k = 0; k < W; k++) a = a + m.sv_x0[(size_t)k * NP + i] * m.sv_x[m.sv_colsi[(size_t)k * NP + i]];
    double ri = m.sv_bn[i] - a;
    m.sv_r[i] = ri; m.sv_r0[i] = ri; m.sv_pd[i] = ri;
    q[0] = ri * ri;
  }
  ds_block_partials<1>(q, m.sv_part, nblk);
}
// z = M x with the sparsified inverse (CSR: sv_mp, sv_mc, sv_minv).  The vector (<= 4096 doubles) is staged in LDS by every block,
// a wavefront takes one row at a time: lane l adds the entries l, l+64, ... of the row in that order, the 64 partial sums go through
// the wave tree.  MODE 0: x = p (first: evaluates ||r0||^2 and the start state);  MODE 1: alpha ; x = s = r - alpha v (block 0 stores s).
template <int MODE>
__global__ void __launch_bounds__(256) k_xi_gemv(DM m, int nblk, int slot, int first, double tol2) {
  __shared__ double xs[4096];
  const int t = threadIdx.x;
  const double *st = m.sv_kry + 16 * slot;
  double alpha = 0.0;
  if (MODE == 0) {
    if (first) {
      const double rr = xi_sum16(m.sv_part, nblk);
      const bool go = rr >= tol2;
      if (blockIdx.x == 0 && t == 0) {
        double *s0 = m.sv_kry;
        s0[0] = 1.0; s0[1] = 1.0; s0[2] = rr; s0[3] = 1.0; s0[4] = rr; s0[5] = rr; s0[6] = 0.0; s0[7] = go ? 0.0 : 1.0;
        m.sv_info[0] = 0; m.sv_resid[0] = sqrt(rr > 0.0 ? rr : 0.0);
      }
      if (!go) return;
    } else if (st[7] != 0.0) return;
  } else {
    if (st[7] != 0.0) return;
    alpha = st[4] / xi_sum16(m.sv_part, nblk);
    if (blockIdx.x == 0 && t == 0) m.sv_kry[32] = alpha;
  }
  const int n = m.myN;
  for (int i = t; i < n; i += 256) {
    const double val = (MODE == 0) ? m.sv_pd[i] : m.sv_r[i] - alpha * m.sv_v[i];
    xs[i] = val;
    if (MODE == 1 && blockIdx.x == 0) m.sv_sn[i] = val;
  }
  __syncthreads();
  const int lane = t & 63, w = t >> 6;
  double *out = MODE == 0 ? m.sv_ph : m.sv_sh;
#pragma unroll
  for (int q = 0; q < XI_ROWS / 4; q++) {
    const int row = blockIdx.x * XI_ROWS + w * (XI_ROWS / 4) + q;
    if (row >= n) break;
    const int beg = m.sv_mp[row], end = m.sv_mp[row + 1];
    double a = 0.0;
    for (int e = beg + lane; e < end; e += 64) a = a + (double)m.sv_minv[e] * xs[m.sv_mc[e]];
    a = xi_wave_total(a);
    if (lane == 63) out[row] = a;
  }
}
template <int W>
__global__ void __launch_bounds__(DSB) k_xi_spmv1(DM m, int NP, int nblk, int slot) {          // v = A_s p^ ; r0.v
  const double *st = m.sv_kry + 16 * slot;
  if (st[7] != 0.0) return;
  int i = blockIdx.x * DSB + threadIdx.x;
  double q[1] = {0.0};
  if (i < m.myN) {
    double a = 0.0;
#pragma unroll
    for (int k = 0; k < W; k++) a = a + m.sv_x0[(size_t)k * NP + i] * m.sv_ph[m.sv_colsi[(size_t)k * NP + i]];
    m.sv_v[i] = a; q[0] = m.sv_r0[i] * a;
  }
  ds_block_partials<1>(q, m.sv_part, nblk);
}
template <int W>
__global__ void __launch_bounds__(DSB) k_xi_spmv2(DM m, int NP, int nblk, int slot) {          // t = A_s s^ ; t.t, t.s, r0.t, s.s
  const double *st = m.sv_kry + 16 * slot;
  if (st[7] != 0.0) return;
  int i = blockIdx.x * DSB + threadIdx.x;
  double q[4] = {0.0, 0.0, 0.0, 0.0};
  if (i < m.myN) {
    double a = 0.0;
#pragma unroll
    for (int k = 0; k < W; k++) a = a + m.sv_x0[(size_t)k * NP + i] * m.sv_sh[m.sv_colsi[(size_t)k * NP + i]];
    const double si = m.sv_sn[i];
    m.sv_t[i] = a;
    q[0] = a * a; q[1] = a * si; q[2] = m.sv_r0[i] * a; q[3] = si * si;
  }
  ds_block_partials<4>(q, m.sv_part + 4 * (size_t)nblk, nblk);
}
__global__ void __launch_bounds__(DSB) k_xi_upd(DM m, int nblk, int slot, double tol2, int maxits) {   // scalars ; x, r ; next p
  const double *st = m.sv_kry + 16 * slot;
  double *so = m.sv_kry + 16 * (1 - slot);
  if (st[7] != 0.0) {
    if (blockIdx.x == 0 && threadIdx.x < 16) so[threadIdx.x] = st[threadIdx.x];
    return;
  }
  __shared__ double sh[4][DSB];
  double tot[4];
  dm_sum_blocks4(m.sv_part + 4 * (size_t)nblk, nblk, sh, tot);
  const double tt = tot[0], ts = tot[1], r0t = tot[2], ss = tot[3];
  const double alpha = m.sv_kry[32];
  const double omega = (tt > 0.0) ? ts / tt : 0.0;
  const double rho = st[4], rho_new = -omega * r0t;
  const double rr = ss - omega * (2.0 * ts - omega * tt);
  const double it = st[6] + 1.0;
  const bool more = (rr >= tol2 && it < (double)maxits);
  const double beta = more ? (rho_new / rho) * (alpha / omega) : 0.0;
  int i = blockIdx.x * DSB + threadIdx.x;
  if (i < m.myN) {
    const double si = m.sv_sn[i];
    const double ri = si - omega * m.sv_t[i];
    m.sv_r[i] = ri;
    const double xn = (m.sv_x[i] + alpha * m.sv_ph[i]) + omega * m.sv_sh[i];
    m.sv_x[i] = xn;
    if (more) m.sv_pd[i] = ri + beta * (m.sv_pd[i] - omega * m.sv_v[i]);
    else m.d_eta[i] = xn;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    so[0] = alpha; so[1] = omega; so[2] = beta; so[3] = rho; so[4] = rho_new; so[5] = rr; so[6] = it; so[7] = more ? 0.0 : 1.0;
    m.sv_info[0] = (int)it; m.sv_resid[0] = sqrt(rr > 0.0 ? rr : 0.0);
  }
}
int launch_solver_xinv(const DM &m, hipStream_t s, int fuse_rhs, int scale_done) {
  const int NP = (m.myN + 63) / 64 * 64, nblk = (m.myN + DSB - 1) / DSB;
  const double tol2 = m.sv_tol > 0.0 ? m.sv_tol * m.sv_tol : 1e-10 * 1e-10; const int maxits = m.sv_maxits > 0 ? m.sv_maxits : 2000;
  if (!scale_done) launch_row_scale(m, s);
  hipLaunchKernelGGL(k_solver_setup<10>, dim3((NP + 255) / 256), dim3(256), 0, s, m, NP, fuse_rhs, 1);   // Jacobi copies for the safety net + natural-order A_s, b, x0
  hipLaunchKernelGGL(k_xi_init<10>, dim3(nblk), dim3(DSB), 0, s, m, NP, nblk);
  // enqueued iterations: solver_xinv_its, or the default schedule -- 2 during the spin-up (the first XI_SPINUP solves after init: from rest the SSH
  // tendency is large, a solve needs a second iteration, and a second explicit-inverse iteration, 23 us, is cheaper than the Jacobi continuation,
  // 5 iterations / 36 us), 1 afterwards (the extrapolated first guess: one iteration suffices, a second would be 5 no-op launches)
  const int K = m.sv_xi_its > 0 ? m.sv_xi_its : (m.sv_solves < XI_SPINUP ? 2 : 1), gblk = (m.myN + XI_ROWS - 1) / XI_ROWS;
  const_cast<DM &>(m).sv_solves = m.sv_solves + 1;           // (host-side counter of the context's DM)
  int slot = 0;
  for (int k = 0; k < K; k++) {
    hipLaunchKernelGGL(k_xi_gemv<0>, dim3(gblk), dim3(256), 0, s, m, nblk, slot, k == 0 ? 1 : 0, tol2);
    hipLaunchKernelGGL(k_xi_spmv1<10>, dim3(nblk), dim3(DSB), 0, s, m, NP, nblk, slot);
    hipLaunchKernelGGL(k_xi_gemv<1>, dim3(gblk), dim3(256), 0, s, m, nblk, slot, 0, tol2);
    hipLaunchKernelGGL(k_xi_spmv2<10>, dim3(nblk), dim3(DSB), 0, s, m, NP, nblk, slot);
    hipLaunchKernelGGL(k_xi_upd, dim3(nblk), dim3(DSB), 0, s, m, nblk, slot, tol2, maxits);
    slot = 1 - slot;
  }
  const double *cont = m.sv_kry + 16 * slot;
  if (m.myN <= 3200) hipLaunchKernelGGL((k_solver_reg<10, 3200, 4, 3>), dim3(1), dim3(ST), (size_t)(128 + (2 + 4) * 3200) * sizeof(double), s, m, maxits, tol2, NP, cont);
  else hipLaunchKernelGGL((k_solver_reg<10, 4096, 2, 3>), dim3(1), dim3(ST), (size_t)(128 + (2 + 2) * 4096) * sizeof(double), s, m, maxits, tol2, NP, cont);
  return 0;
}

// single phases of the explicit-inverse solve for per-kernel timing (fesom_gpu_kernel_time_ms): they run on whatever the last
// solve left in the work vectors, with the `done` flag of slot 1 cleared first by "xi_arm"
__global__ void k_xi_arm(DM m) { if (!threadIdx.x) { m.sv_kry[7] = 0.0; m.sv_kry[23] = 0.0; m.sv_kry[4] = 1.0; m.sv_kry[20] = 1.0; } }
int launch_named_xi(const DM &m, hipStream_t s, const char *name) {
  if (!m.sv_minv) return 1;
  const int NP = (m.myN + 63) / 64 * 64, nblk = (m.myN + DSB - 1) / DSB, gblk = (m.myN + XI_ROWS - 1) / XI_ROWS;
  const double tol2 = 1e-10 * 1e-10;
  if (!strcmp(name, "xi_arm")) { hipLaunchKernelGGL(k_xi_arm, dim3(1), dim3(64), 0, s, m); return 0; }
  if (!strcmp(name, "xi_gemv0")) { hipLaunchKernelGGL(k_xi_gemv<0>, dim3(gblk), dim3(256), 0, s, m, nblk, 1, 0, tol2); return 0; }
  if (!strcmp(name, "xi_gemv1")) { hipLaunchKernelGGL(k_xi_gemv<1>, dim3(gblk), dim3(256), 0, s, m, nblk, 1, 0, tol2); return 0; }
  if (!strcmp(name, "xi_spmv1")) { hipLaunchKernelGGL(k_xi_spmv1<10>, dim3(nblk), dim3(DSB), 0, s, m, NP, nblk, 1); return 0; }
  if (!strcmp(name, "xi_spmv2")) { hipLaunchKernelGGL(k_xi_spmv2<10>, dim3(nblk), dim3(DSB), 0, s, m, NP, nblk, 1); return 0; }
  if (!strcmp(name, "xi_init")) { hipLaunchKernelGGL(k_xi_init<10>, dim3(nblk), dim3(DSB), 0, s, m, NP, nblk); return 0; }
  if (!strcmp(name, "xi_setup")) { hipLaunchKernelGGL(k_solver_setup<10>, dim3((NP + 255) / 256), dim3(256), 0, s, m, NP, 0, 1); return 0; }
  return -1;
}
