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
                                                         const int *__restrict__ lev_lo /* kmin = lev_lo[c] */, int mode, int tr, int dbg) {
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
  // ---- solve (lane = column), wave 0.  Branch-free and software-pipelined: the first active level uses a = 0
  // (b - cp*0 = b and r - x*0 = r exactly, i.e. the reference's c/b, r/b start), inactive levels compute on garbage and
  // are discarded by selects, so the independent divide chains (cp, x1, x2) interleave in straight-line code and the
  // dependent chain  cp_j = c_j / (b_j - cp_{j-1} a_j)  is the only serial part.
  if (w == 0 && !(dbg & 1)) {
    int c = col0 + l;
    int kmin = 1, kmax = 0;
    if (c < ncol) { kmin = lev_lo[c]; kmax = lev_hi[c] - 1; }
    int kmx = kmax;
    for (int s = 32; s >= 1; s >>= 1) kmx = max(kmx, __shfl_xor(kmx, s, 64));
    const int astr = nl1 * TPAD;                              // array stride in doubles
    double *p0 = sh + l;                                      // level 1 of array 0, this lane's column
    double cpp = 0.0, x1p = 0.0, x2p = 0.0;
    double a = p0[0], b = p0[astr], cc = p0[2 * astr], r1 = p0[3 * astr], r2 = (NRHS == 2) ? p0[4 * astr] : 0.0;
    double *pj = p0;
    for (int j = 1; j <= kmx; j++) {
      double *pn = (j < kmx) ? pj + TPAD : pj;
      double na = pn[0], nb = pn[astr], nc = pn[2 * astr], nr1 = pn[3 * astr], nr2 = (NRHS == 2) ? pn[4 * astr] : 0.0;
      const bool act = (j >= kmin) && (j <= kmax);
      const double am = (j == kmin) ? 0.0 : a;
      double mm = b - cpp * am;
      double ncp = cc / mm;
      double nx1 = (r1 - x1p * am) / mm;
      double nx2 = (NRHS == 2) ? (r2 - x2p * am) / mm : 0.0;
      if (act) {
        cpp = ncp; x1p = nx1; x2p = nx2;
        pj[2 * astr] = ncp; pj[3 * astr] = nx1;
        if (NRHS == 2) pj[4 * astr] = nx2;
      }
      a = na; b = nb; cc = nc; r1 = nr1; r2 = nr2;
      pj = pn;
    }
    double x1 = 0.0, x2 = 0.0;
    pj = p0 + (size_t)(kmx > 0 ? kmx - 1 : 0) * TPAD;
    double cp = pj[2 * astr], u1 = pj[3 * astr], u2 = (NRHS == 2) ? pj[4 * astr] : 0.0;
    for (int j = kmx; j >= 1; j--) {
      double *pn = (j > 1) ? pj - TPAD : pj;
      double ncp = pn[2 * astr], nu1 = pn[3 * astr], nu2 = (NRHS == 2) ? pn[4 * astr] : 0.0;
      if (j >= kmin && j <= kmax) {
        if (j == kmax) { x1 = u1; if (NRHS == 2) x2 = u2; }
        else {
          x1 = u1 - cp * x1;
          if (NRHS == 2) x2 = u2 - cp * x2;
        }
        pj[3 * astr] = x1;
        if (NRHS == 2) pj[4 * astr] = x2;
      }
      cp = ncp; u1 = nu1; u2 = nu2;
      pj = pn;
    }
  }
  __syncthreads();
  if (dbg & 2) return;
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
static int thomas_dbg() { static int d = getenv("FESOM_THOMAS_DBG") ? atoi(getenv("FESOM_THOMAS_DBG")) : 0; return d; }
void thomas_prepare() {
  static bool done = false;
  if (done) return;
  (void)hipFuncSetAttribute((const void *)k_thomas<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute((const void *)k_thomas<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  done = true;
}
void launch_thomas_visc(const DM &m, hipStream_t s) {
  hipLaunchKernelGGL(k_thomas<2>, dim3((m.myE + TCOLS - 1) / TCOLS), dim3(64 * TWAVES), thomas_lds(m, 2), s, m, m.myE, m.nlev, m.ulev, 0, 0, thomas_dbg());
}
void launch_thomas_tracer(const DM &m, hipStream_t s, int tr) {
  hipLaunchKernelGGL(k_thomas<1>, dim3((m.myN + TCOLS - 1) / TCOLS), dim3(64 * TWAVES), thomas_lds(m, 1), s, m, m.myN, m.nlev_n, m.ulev_n, 1, tr, thomas_dbg());
}
