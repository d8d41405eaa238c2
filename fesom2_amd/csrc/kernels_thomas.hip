// Batched tridiagonal (Thomas) solve, one LANE per column (gfx950).
//
// impl_vert_visc_ale (src/oce_ale.F90:2491-2510) and diff_ver_part_impl_ale (src/oce_ale_tracer.F90:838-852) end in a
// Thomas sweep that is sequential in z and dominated by dependent fp64 divides.  Executing it once per wavefront with
// all 64 lanes in lock-step (round-1 first version) is issue-bound: 64x redundant divides.  Here the coefficient
// kernels (wave per column, coalesced) leave a, b, c, rhs in column-major scratch; this kernel stages 64 columns
// through LDS (transposing: coalesced 376-B column reads -> [level][column] rows, padded to 65 against bank conflicts),
// then ONE wave solves 64 columns at once (lane = column), and all waves write the result back coalesced.
// Arithmetic order is the reference's, so results stay bit-identical.
#include "dev.h"

#define TCOLS 64
#define TPAD 65
#define TWAVES 4

template <int NRHS>
__global__ void __launch_bounds__(64 * TWAVES) k_thomas(DM m, int ncol, const int *__restrict__ lev_hi /* kmax = lev_hi[c]-1 */,
                                                         const int *__restrict__ lev_lo /* kmin = lev_lo[c] */, int mode, int tr) {
  extern __shared__ double sh[];
  const int nl1 = m.nlm1;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int col0 = blockIdx.x * TCOLS;
  const size_t toff = (size_t)tr * m.nlm1 * m.N;          // per-tracer scratch slab (mode 1)
  const double *G[5] = {mode == 0 ? m.th_a : m.tt_a + toff, mode == 0 ? m.th_b : m.tt_b + toff, mode == 0 ? m.th_c : m.tt_c + toff,
                        mode == 0 ? m.th_r1 : m.tt_r + toff, m.th_r2};
  const int NARR = 3 + NRHS;
#define SH(arr, j, c) sh[((arr) * nl1 + ((j) - 1)) * TPAD + (c)]
  // ---- stage in (lane = level); loads of different columns are independent -> unrolled, many in flight
#pragma unroll 4
  for (int cc = 0; cc < TCOLS / TWAVES; cc++) {
    int cl = w * (TCOLS / TWAVES) + cc, c = col0 + cl;
    int nz = l + 1;
    if (c < ncol && nz >= lev_lo[c] && nz <= lev_hi[c] - 1) {
#pragma unroll
      for (int a = 0; a < NARR; a++) SH(a, nz, cl) = G[a][(size_t)c * nl1 + l];
    }
  }
  __syncthreads();
  // ---- solve (lane = column), wave 0.  Software-pipelined: the coefficients of level j+1 are read from LDS before the
  // results of level j are stored, so the dependent divide chain is the only serial part.
  if (w == 0) {
    int c = col0 + l;
    int kmin = 1, kmax = 0;
    if (c < ncol) { kmin = lev_lo[c]; kmax = lev_hi[c] - 1; }
    int kmx = kmax;
    for (int s = 32; s >= 1; s >>= 1) kmx = max(kmx, __shfl_xor(kmx, s, 64));
    double cpp = 0.0, x1p = 0.0, x2p = 0.0;
    double a = SH(0, 1, l), b = SH(1, 1, l), cc = SH(2, 1, l), r1 = SH(3, 1, l), r2 = (NRHS == 2) ? SH(4, 1, l) : 0.0;
    for (int j = 1; j <= kmx; j++) {
      int jn = (j < kmx) ? j + 1 : j;
      double na = SH(0, jn, l), nb = SH(1, jn, l), nc = SH(2, jn, l), nr1 = SH(3, jn, l), nr2 = (NRHS == 2) ? SH(4, jn, l) : 0.0;
      if (j >= kmin && j <= kmax) {
        if (j == kmin) {
          cpp = cc / b; x1p = r1 / b;
          if (NRHS == 2) x2p = r2 / b;
        } else {
          double mm = b - cpp * a;
          cpp = cc / mm;
          x1p = (r1 - x1p * a) / mm;
          if (NRHS == 2) x2p = (r2 - x2p * a) / mm;
        }
        SH(2, j, l) = cpp; SH(3, j, l) = x1p;
        if (NRHS == 2) SH(4, j, l) = x2p;
      }
      a = na; b = nb; cc = nc; r1 = nr1; r2 = nr2;
    }
    double x1 = 0.0, x2 = 0.0;
    double cp = SH(2, kmx, l), u1 = SH(3, kmx, l), u2 = (NRHS == 2) ? SH(4, kmx, l) : 0.0;
    for (int j = kmx; j >= 1; j--) {
      int jn = (j > 1) ? j - 1 : j;
      double ncp = SH(2, jn, l), nu1 = SH(3, jn, l), nu2 = (NRHS == 2) ? SH(4, jn, l) : 0.0;
      if (j >= kmin && j <= kmax) {
        if (j == kmax) { x1 = u1; if (NRHS == 2) x2 = u2; }
        else {
          x1 = u1 - cp * x1;
          if (NRHS == 2) x2 = u2 - cp * x2;
        }
        SH(3, j, l) = x1;
        if (NRHS == 2) SH(4, j, l) = x2;
      }
      cp = ncp; u1 = nu1; u2 = nu2;
    }
  }
  __syncthreads();
  // ---- stage out (lane = level)
#pragma unroll 4
  for (int cc = 0; cc < TCOLS / TWAVES; cc++) {
    int cl = w * (TCOLS / TWAVES) + cc, c = col0 + cl;
    int nz = l + 1;
    if (c < ncol && nz >= lev_lo[c] && nz <= lev_hi[c] - 1) {
      if (mode == 0) {                       // impl_vert_visc_ale: UV_rhs = (du, dv)
        DV2(m.UV_rhs, 1, nz, c) = SH(3, nz, cl);
        DV2(m.UV_rhs, 2, nz, c) = SH(4, nz, cl);
      } else {                               // diff_ver_part_impl_ale: tr_arr = T* + dT ; salinity clamp (oce_ale_tracer.F90:176-198)
        double T = DTR(m.tr_arr, nz, c, tr) + SH(3, nz, cl);
        if (tr == 1) { if (T > 45.0) T = 45.0; if (T < 3.0) T = 3.0; }
        DTR(m.tr_arr, nz, c, tr) = T;
      }
    }
  }
}

static size_t thomas_lds(const DM &m, int nrhs) { return (size_t)(3 + nrhs) * m.nlm1 * TPAD * sizeof(double); }
void thomas_prepare() {
  static bool done = false;
  if (done) return;
  (void)hipFuncSetAttribute((const void *)k_thomas<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute((const void *)k_thomas<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  done = true;
}
void launch_thomas_visc(const DM &m, hipStream_t s) {
  hipLaunchKernelGGL(k_thomas<2>, dim3((m.myE + TCOLS - 1) / TCOLS), dim3(64 * TWAVES), thomas_lds(m, 2), s, m, m.myE, m.nlev, m.ulev, 0, 0);
}
void launch_thomas_tracer(const DM &m, hipStream_t s, int tr) {
  hipLaunchKernelGGL(k_thomas<1>, dim3((m.myN + TCOLS - 1) / TCOLS), dim3(64 * TWAVES), thomas_lds(m, 1), s, m, m.myN, m.nlev_n, m.ulev_n, 1, tr);
}
