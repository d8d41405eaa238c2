// Device-side step monitor: write_step_info (src/write_step_info.F90:14-222) + check_blowup (:225-447) over the owned nodes,
// so that logging and the blow-up test need no per-step download of fields.  Two launches: k_mon_col (one wavefront per
// column: extrema of the 3-D fields of the column, blow-up tests) and k_mon_reduce (ONE workgroup: node loop strided over
// the threads, butterfly across the wave, the 16 wave results combined in wave order -> a fixed summation order).
#include "dev.h"

#define MON_NCOL 10          // per-column results: tmin tmax smin smax cflmax kvmax pgfxmax pgfymax avmax blowup
#define MON_NSUM 6
#define MON_NMIN 15
#define MON_NMAX 20
#define MON_NOUT 42

__device__ __forceinline__ double wmin(double x) { for (int s = 32; s >= 1; s >>= 1) x = fmin(x, __shfl_xor(x, s, 64)); return x; }
__device__ __forceinline__ double wmax(double x) { for (int s = 32; s >= 1; s >>= 1) x = fmax(x, __shfl_xor(x, s, 64)); return x; }
__device__ __forceinline__ double wsum(double x) { for (int s = 32; s >= 1; s >>= 1) x = x + __shfl_xor(x, s, 64); return x; }

__global__ void __launch_bounds__(BLOCK) k_mon_col(DM m, double *col) {
  const int n = col_id(m), l = lane_id(), nz = l + 1;
  if (n >= m.myN) return;
  const double BIG = 1.0e300;
  double tmin = BIG, tmax = -BIG, smin = BIG, smax = -BIG, cfl = -BIG, kv = -BIG, px = -BIG, py = -BIG, av = -BIG, blow = 0.0;
  if (nz <= m.nlm1) {
    const double t = DTR(m.tr_arr, nz, n, 0), s = DTR(m.tr_arr, nz, n, 1);
    if (s != 0.0) { tmin = tmax = t; smin = smax = s; }
    if (nz <= m.nlev_n[n] - 1) {
      if (t != t || t < -5.0 || t > 60) blow = 1.0;
      if (s != s || s < 0 || s > 50) blow = 1.0;
    }
    if (n < m.E) { px = fabs(DA2(m.pgf_x, nz, n)); py = fabs(DA2(m.pgf_y, nz, n)); }      // element arrays, node count (reference quirk)
  }
  if (nz <= m.nl) {
    cfl = DA2L(m.CFL_z, nz, n); kv = fabs(DA2L(m.Kv, nz, n));
    if (n < m.E) av = fabs(DA2L(m.Av, nz, n));
  }
  if (l == 0) {
    const double e = m.eta_n[n], de = m.d_eta[n];
    if (e != e || e < -50.0 || e > 50.0 || de != de) blow = 1.0;
    if (m.p.which_ale != 0) {
      const double w = DA2L(m.Wvel, 1, n), h = DA2(m.hnode, 1, n);
      if (w != w) blow = 1.0;
      if (h != h || h < 0) blow = 1.0;
    }
  }
  tmin = wmin(tmin); tmax = wmax(tmax); smin = wmin(smin); smax = wmax(smax); cfl = wmax(cfl); kv = wmax(kv);
  px = wmax(px); py = wmax(py); av = wmax(av); blow = wmax(blow);
  if (l == 0) {
    double *c = col + (size_t)n * MON_NCOL;
    c[0] = tmin; c[1] = tmax; c[2] = smin; c[3] = smax; c[4] = cfl; c[5] = kv; c[6] = px; c[7] = py; c[8] = av; c[9] = blow;
  }
}

__global__ void __launch_bounds__(1024) k_mon_reduce(DM m, const double *col, double *out) {
  __shared__ double part[16][MON_NOUT];
  const int t = threadIdx.x, w = t >> 6, l = t & 63;
  const double BIG = 1.0e300;
  double v[MON_NOUT];
  for (int i = 0; i < MON_NSUM; i++) v[i] = 0.0;
  for (int i = 0; i < MON_NMIN; i++) v[MON_NSUM + i] = BIG;
  for (int i = 0; i < MON_NMAX + 1; i++) v[MON_NSUM + MON_NMIN + i] = -BIG;
  double *mn = v + MON_NSUM, *mx = v + MON_NSUM + MON_NMIN;
#define MM(i, x) do { const double x_ = (x); mn[i] = fmin(mn[i], x_); mx[i] = fmax(mx[i], x_); } while (0)
  for (int n = t; n < m.myN; n += 1024) {
    const int ul = m.ulev_n[n];
    const double a = DA2L(m.areasvol, ul, n), e = m.eta_n[n], hb = m.hbar[n], de = m.d_eta[n], wf = m.water_flux[n];
    v[0] = v[0] + a * e; v[1] = v[1] + a * hb; v[2] = v[2] + a * de; v[3] = v[3] + a * (hb - m.hbar_old[n]); v[4] = v[4] + a * wf;
    v[5] = v[5] + DA2L(m.area, ul, n);
    MM(0, e); MM(1, hb); MM(2, wf); MM(3, m.heat_flux[n]);
    const double *c = col + (size_t)n * MON_NCOL;
    mn[4] = fmin(mn[4], c[0]); mx[4] = fmax(mx[4], c[1]); mn[5] = fmin(mn[5], c[2]); mx[5] = fmax(mx[5], c[3]);
    MM(6, DA2L(m.Wvel, 1, n)); MM(7, DA2L(m.Wvel, 2, n));
    MM(8, DV2(m.Unode, 1, 1, n)); MM(9, DV2(m.Unode, 1, 2, n)); MM(10, DV2(m.Unode, 2, 1, n)); MM(11, DV2(m.Unode, 2, 2, n));
    MM(12, de);
    const double h1 = DA2(m.hnode, 1, n), h2 = DA2(m.hnode, 2, n);
    if (h1 != 0.0) MM(13, h1);
    if (h2 != 0.0) MM(14, h2);
    mx[15] = fmax(mx[15], c[4]); mx[16] = fmax(mx[16], c[6]); mx[17] = fmax(mx[17], c[7]); mx[18] = fmax(mx[18], c[8]);
    mx[19] = fmax(mx[19], c[5]); mx[20] = fmax(mx[20], c[9]);
  }
  for (int i = 0; i < MON_NSUM; i++) v[i] = wsum(v[i]);
  for (int i = 0; i < MON_NMIN; i++) mn[i] = wmin(mn[i]);
  for (int i = 0; i < MON_NMAX + 1; i++) mx[i] = wmax(mx[i]);
  if (l == 0) for (int i = 0; i < MON_NOUT; i++) part[w][i] = v[i];
  __syncthreads();
  if (t < MON_NOUT) {
    double r = part[0][t];
    for (int k = 1; k < 16; k++) r = t < MON_NSUM ? r + part[k][t] : (t < MON_NSUM + MON_NMIN ? fmin(r, part[k][t]) : fmax(r, part[k][t]));
    out[t] = r;
  }
}

void launch_step_info(const DM &m, hipStream_t s, double *col, double *out) {
  hipLaunchKernelGGL(k_mon_col, dim3(nblocks(m.myN)), dim3(BLOCK), 0, s, m, col);
  hipLaunchKernelGGL(k_mon_reduce, dim3(1), dim3(1024), 0, s, m, (const double *)col, out);
}
