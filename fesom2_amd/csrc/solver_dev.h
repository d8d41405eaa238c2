// Block-level reductions shared by the multi-workgroup SSH solver phases (solver.hip, solver_ras.hip).  Fixed summation orders that the
// CPU checker of the tests reproduces: halving tree inside a 256-thread block, block partials added 256-strided, then the same tree.
#pragma once
#include "dev.h"
#define DSB 256
template <int NQ>
__device__ __forceinline__ void ds_block_partials(double (&v)[NQ], double *part, int nblk) {
  __shared__ double sh[NQ][DSB];
  const int t = threadIdx.x;
#pragma unroll
  for (int q = 0; q < NQ; q++) sh[q][t] = v[q];
  __syncthreads();
  for (int s = DSB / 2; s >= 1; s >>= 1) {
    if (t < s) {
#pragma unroll
      for (int q = 0; q < NQ; q++) sh[q][t] = sh[q][t] + sh[q][t + s];
    }
    __syncthreads();
  }
  if (t == 0) {
#pragma unroll
    for (int q = 0; q < NQ; q++) part[(size_t)q * nblk + blockIdx.x] = sh[q][0];
  }
}
// sum of the nblk block partials, evaluated by every block in the same fixed order: thread t adds part[t], part[t+256], ...
// in that order, then the halving tree over the 256 threads (strides 128..1).  Must be called by the whole block.
__device__ __forceinline__ double dm_sum_blocks(const double *part, int nblk, double *sh /* DSB doubles */) {
  const int t = threadIdx.x;
  double a = 0.0;
  for (int b = t; b < nblk; b += DSB) a = a + part[b];
  sh[t] = a;
  __syncthreads();
  for (int s2 = DSB / 2; s2 >= 1; s2 >>= 1) {
    if (t < s2) sh[t] = sh[t] + sh[t + s2];
    __syncthreads();
  }
  double r = sh[0];
  __syncthreads();
  return r;
}
// four sums at once (same order per quantity as dm_sum_blocks, one set of barriers for all four)
__device__ __forceinline__ void dm_sum_blocks4(const double *part, int nblk, double (*sh)[DSB], double (&out)[4]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    double a = 0.0;
    for (int b = t; b < nblk; b += DSB) a = a + part[(size_t)q * nblk + b];
    sh[q][t] = a;
  }
  __syncthreads();
  for (int s2 = DSB / 2; s2 >= 1; s2 >>= 1) {
    if (t < s2) {
#pragma unroll
      for (int q = 0; q < 4; q++) sh[q][t] = sh[q][t] + sh[q][t + s2];
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < 4; q++) out[q] = sh[q][0];
  __syncthreads();
}
