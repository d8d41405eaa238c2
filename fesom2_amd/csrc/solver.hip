// SSH solver (gfx950).  Boundary: solve_ssh_ale (src/oce_ale.F90:2210-2344) -> psolve (src/psolve.c:152-221).
//
// The reference solves the row-scaled system with pARMS BiCGstab + RAS/ILU(2) (lib/parms/src/bicgstab_ras.c:49-259);
// ILU triangular solves are sequential, so this build keeps what defines the answer -- row scaling
// scale[i]=1/sum|a_ij| (psolve.c:58-65,180-188), warm start x=d_eta (psolve.c:206-212), stop when
// ||r||^2 < tol^2 with tol=1e-10 absolute on the scaled residual (bicgstab_ras.c:78,146,220), maxits 2000 --
// and replaces the preconditioner by Jacobi on the scaled operator.  The solution agrees with the
// reference to solver tolerance (tests), not bit for bit.
//
// N2 is tiny (3140 rows, 21112 nnz on pi): one launch, ONE 1024-thread workgroup runs the whole Krylov loop
// (no host round trips, no grid barrier).  The search vector lives in LDS (SpMV gathers hit LDS), the other
// vectors are own-row and stay in L2.  Dot products use a FIXED reduction order (1024 strided partial sums,
// then a halving tree) that the CPU oracle reproduces, so oracle and HIP agree bitwise.
#include "dev.h"

#define ST 1024

// tree of  part[t] += part[t+s], s = 512..1  evaluated by wave 0; lane l combines part[l+64k], k=0..15
__device__ __forceinline__ double tree16(const double *p, int l) {
  double q[16];
#pragma unroll
  for (int k = 0; k < 16; k++) q[k] = p[l + 64 * k];
#pragma unroll
  for (int k = 0; k < 8; k++) q[k] = q[k] + q[k + 8];      // stride 512
#pragma unroll
  for (int k = 0; k < 4; k++) q[k] = q[k] + q[k + 4];      // stride 256
#pragma unroll
  for (int k = 0; k < 2; k++) q[k] = q[k] + q[k + 2];      // stride 128
  double x = q[0] + q[1];                                   // stride 64
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) x = x + __shfl_down(x, s, 64);
  return x;                                                 // valid in lane 0
}

// reduce up to two partial sums per thread; result broadcast through LDS
__device__ __forceinline__ void reduce2(double a, double b, double *red, double *out, double &ra, double &rb) {
  int t = threadIdx.x;
  red[t] = a; red[ST + t] = b;
  __syncthreads();
  if (t < 64) {
    double xa = tree16(red, t), xb = tree16(red + ST, t);
    if (t == 0) { out[0] = xa; out[1] = xb; }
  }
  __syncthreads();
  ra = out[0]; rb = out[1];
}

__global__ void __launch_bounds__(ST) k_solver(DM m, int maxits, double tol2, int ph_in_lds) {
  extern __shared__ double lds[];
  double *red = lds;                 // 2*ST
  double *out = lds + 2 * ST;        // 2 (+pad)
  double *ph = ph_in_lds ? (lds + 2 * ST + 8) : m.sv_ph;
  const int t = threadIdx.x, n = m.myN;
  const int *rp = m.rowptr, *ci = m.colind;
  double *vals = m.sv_vals, *dinv = m.sv_dinv, *b = m.sv_b, *r = m.sv_r, *r0 = m.sv_r0, *pv = m.sv_p, *v = m.sv_v, *s = m.sv_s,
         *tv = m.sv_t, *x = m.d_eta;
  // row scaling + Jacobi diagonal
  for (int i = t; i < n; i += ST) {
    double tmp = 0.;
    for (int j = rp[i]; j < rp[i + 1]; j++) tmp += fabs(m.ssh_values[j]);
    double sc = 1. / tmp;
    for (int j = rp[i]; j < rp[i + 1]; j++) vals[j] = m.ssh_values[j] * sc;
    b[i] = m.ssh_rhs[i] * sc;
    dinv[i] = 1.0 / vals[rp[i]];
    ph[i] = x[i];
  }
  __syncthreads();
  double prr = 0.0, prho = 0.0;
  for (int i = t; i < n; i += ST) {
    double a = 0.0;
    for (int j = rp[i]; j < rp[i + 1]; j++) a = a + vals[j] * ph[ci[j]];
    double ri = b[i] - a;
    r[i] = ri; r0[i] = ri; pv[i] = 0.0; v[i] = 0.0;
    prr = prr + ri * ri; prho = prho + ri * ri;
  }
  double rr, rho_new;
  reduce2(prr, prho, red, out, rr, rho_new);
  double rho = 1.0, alpha = 1.0, omega = 1.0;
  int it = 0;
  while (rr >= tol2 && it < maxits) {
    double beta = (rho_new / rho) * (alpha / omega);
    for (int i = t; i < n; i += ST) {
      double pi = r[i] + beta * (pv[i] - omega * v[i]);
      pv[i] = pi;
      ph[i] = pi * dinv[i];
    }
    __syncthreads();
    double p1 = 0.0;
    for (int i = t; i < n; i += ST) {
      double a = 0.0;
      for (int j = rp[i]; j < rp[i + 1]; j++) a = a + vals[j] * ph[ci[j]];
      v[i] = a;
      p1 = p1 + r0[i] * a;
    }
    double r0v, dummy;
    reduce2(p1, 0.0, red, out, r0v, dummy);
    alpha = rho_new / r0v;
    for (int i = t; i < n; i += ST) {
      double si = r[i] - alpha * v[i];
      s[i] = si;
      x[i] = x[i] + alpha * ph[i];
    }
    // own rows only: every SpMV read of ph finished before reduce2's barriers
    for (int i = t; i < n; i += ST) ph[i] = s[i] * dinv[i];
    __syncthreads();
    double ptt = 0.0, pts = 0.0;
    for (int i = t; i < n; i += ST) {
      double a = 0.0;
      for (int j = rp[i]; j < rp[i + 1]; j++) a = a + vals[j] * ph[ci[j]];
      tv[i] = a;
      ptt = ptt + a * a; pts = pts + a * s[i];
    }
    double tt, ts;
    reduce2(ptt, pts, red, out, tt, ts);
    omega = (tt > 0.0) ? ts / tt : 0.0;
    prr = 0.0; prho = 0.0;
    for (int i = t; i < n; i += ST) {
      x[i] = x[i] + omega * ph[i];
      double ri = s[i] - omega * tv[i];
      r[i] = ri;
      prr = prr + ri * ri; prho = prho + r0[i] * ri;
    }
    rho = rho_new;
    reduce2(prr, prho, red, out, rr, rho_new);
    it++;
  }
  if (t == 0) { m.sv_info[0] = it; m.sv_resid[0] = sqrt(rr); }
}

void solver_prepare() {
  static bool attr_set = false;
  if (!attr_set) { hipFuncSetAttribute((const void *)k_solver, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
}
void launch_solver(const DM &m, hipStream_t s) {
  size_t need = (size_t)(2 * ST + 8 + m.myN) * sizeof(double);
  int in_lds = need <= 150 * 1024;
  size_t shm = in_lds ? need : (size_t)(2 * ST + 8) * sizeof(double);
  hipLaunchKernelGGL(k_solver, dim3(1), dim3(ST), shm, s, m, 2000, 1e-10 * 1e-10, in_lds);
}
