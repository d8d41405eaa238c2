// Soufflet channel hooks of the reference's toy set-up (src/toy_channel_soufflet.F90), gfx950.  They sit on the step path
// when toy_ocean / which_toy='soufflet' (the reference's CI known-answer case setups/test_souf):
//   compute_zonal_mean   before_oce_step every 10 steps (forcing-update period of the toy module) (oce_setup_step.F90:625-630)
//   relax_zonal_vel      after solve_ssh_ale (oce_ale.F90:2696)
//   relax_zonal_temp     after diff_tracers_ale of EVERY tracer of the loop, always on tracer 1 (oce_ale_tracer.F90:150-151)
// The element -> latitude-bin map, the per-bin element counts and the interpolation weights are static and prepared on the
// host at fesom_gpu_init (compute_zonal_mean_ini :104-155 and the interpolation headers of :57-70, :89-100).
#include "dev.h"
#include <string.h>

// compute_zonal_mean (:157-217): one wavefront per latitude bin, lane = level; the elements of the bin are summed in
// element order (the reference's loop order on one partition), then divided by (count + 0.001).
// The bin's elements come in batches of ZB: their index chains (element, bottom level, the three nodes) are read lane-parallel, the 4 * ZB column
// loads of a batch are issued together, the sums stay in element order (a bin of the 182 600-node channel holds 3648 elements: 2.7 ms per call with one
// dependent load chain per element, a quarter of that batched).
template <bool DIVIDE>
__device__ __forceinline__ void toy_zonal_bin(const DM &m) {
  const int b = col_id(m), l = lane_id(), nz = l + 1;
  if (b >= 100) return;
  const int nzc = nz <= m.nlm1 ? nz : m.nlm1;
  double zt = 0.0, zv = 0.0;
  constexpr int ZB = 8;
  const int q0 = m.toy_bptr[b], q1 = m.toy_bptr[b + 1];
  for (int qb = q0; qb < q1; qb += ZB) {
    int e_l = 0, hi_l = 0, n1_l = 0, n2_l = 0, n3_l = 0;
    if (l < ZB && qb + l < q1) {
      e_l = m.toy_bidx[qb + l]; hi_l = m.nlev[e_l] - 1;
      n1_l = m.elem_nodes[3 * e_l]; n2_l = m.elem_nodes[3 * e_l + 1]; n3_l = m.elem_nodes[3 * e_l + 2];
    }
    double t1[ZB], t2[ZB], t3[ZB], u[ZB];
#pragma unroll
    for (int k = 0; k < ZB; k++) {
      t1[k] = DTR(m.tr_arr, nzc, rdlane(n1_l, k), 0); t2[k] = DTR(m.tr_arr, nzc, rdlane(n2_l, k), 0); t3[k] = DTR(m.tr_arr, nzc, rdlane(n3_l, k), 0);
      u[k] = DV2(m.UV, 1, nzc, rdlane(e_l, k));
    }
#pragma unroll
    for (int k = 0; k < ZB; k++) {
      const bool on = qb + k < q1 && nz <= rdlane(hi_l, k);
      const double nt = zt + ((t1[k] + t2[k]) + t3[k]) / 3.0, nv = zv + u[k];
      zt = on ? nt : zt; zv = on ? nv : zv;
    }
  }
  if (nz > m.nlm1) return;
  if (DIVIDE) {
    const double cnt = m.toy_znum[b];
    zv = zv / (cnt + 0.001); zt = zt / (cnt + 0.001);
  }
  m.toy_zvel[(size_t)b * m.nlm1 + nz - 1] = zv;
  m.toy_ztem[(size_t)b * m.nlm1 + nz - 1] = zt;
}
__global__ void __launch_bounds__(BLOCK) k_toy_zonal_mean(DM m) { toy_zonal_bin<true>(m); }

// Partitioned runs: the rank-local sums (every element once: where its first node is owned, :167) first, then the host's / the
// library's all-reduce over the ranks (the two MPI_AllREDUCE of :182-203), then the division by the global count.
__global__ void __launch_bounds__(BLOCK) k_toy_zonal_sum(DM m) { toy_zonal_bin<false>(m); }
__global__ void __launch_bounds__(BLOCK) k_toy_zonal_div(DM m) {
  int b = col_id(m), nz = lane_id() + 1;
  if (b >= 100 || nz > m.nlm1) return;
  double cnt = m.toy_znum[b];
  m.toy_zvel[(size_t)b * m.nlm1 + nz - 1] = m.toy_zvel[(size_t)b * m.nlm1 + nz - 1] / (cnt + 0.001);
  m.toy_ztem[(size_t)b * m.nlm1 + nz - 1] = m.toy_ztem[(size_t)b * m.nlm1 + nz - 1] / (cnt + 0.001);
}

// relax_zonal_vel (:46-79)
__global__ void __launch_bounds__(BLOCK) k_toy_relax_vel(DM m) {
  int e = col_id(m), nz = lane_id() + 1;
  if (e >= m.myE || nz > m.nlev[e] - 1) return;
  const double tau_inv = 1.0 / 50.0 / 24.0 / 3600.0;
  int nn = m.toy_e_nn[2 * e], nn1 = m.toy_e_nn[2 * e + 1];
  double a = m.toy_e_a[e];
  double Uzon = (1.0 - a) * m.toy_zvel[(size_t)(nn - 1) * m.nlm1 + nz - 1] + a * m.toy_zvel[(size_t)(nn1 - 1) * m.nlm1 + nz - 1];
  DV2(m.UV_rhs, 1, nz, e) = DV2(m.UV_rhs, 1, nz, e) + m.p.dt * tau_inv * (DA2(m.Uclim, nz, e) - Uzon);
}

// relax_zonal_temp (:81-103), owned + halo nodes
__global__ void __launch_bounds__(BLOCK) k_toy_relax_temp(DM m) {
  int n = col_id(m), nz = lane_id() + 1;
  if (n >= m.N || nz > m.nlev_n[n] - 1) return;
  const double tau_inv = 1.0 / 50.0 / 24.0 / 3600.0;
  int nn = m.toy_n_nn[2 * n], nn1 = m.toy_n_nn[2 * n + 1];
  double a = m.toy_n_a[n];
  double Tzon = (1.0 - a) * m.toy_ztem[(size_t)(nn - 1) * m.nlm1 + nz - 1] + a * m.toy_ztem[(size_t)(nn1 - 1) * m.nlm1 + nz - 1];
  DTR(m.tr_arr, nz, n, 0) = DTR(m.tr_arr, nz, n, 0) + m.p.dt * tau_inv * (DA2(m.Tclim, nz, n) - Tzon);
}

#define LAUNCH_COL(k, ncol, ...) hipLaunchKernelGGL(k, dim3(nblocks(ncol)), dim3(BLOCK), 0, s, __VA_ARGS__)
int launch_named_toy(const DM &m, hipStream_t s, const char *name) {
  if (!m.p.toy_soufflet) return -1;
  if (!strcmp(name, "compute_zonal_mean") || !strcmp(name, "k_toy_zonal_mean")) { LAUNCH_COL(k_toy_zonal_mean, 100, m); return 0; }
  if (!strcmp(name, "toy_zonal_sum")) { LAUNCH_COL(k_toy_zonal_sum, 100, m); return 0; }
  if (!strcmp(name, "toy_zonal_div")) { LAUNCH_COL(k_toy_zonal_div, 100, m); return 0; }
  if (!strcmp(name, "relax_zonal_vel") || !strcmp(name, "k_toy_relax_vel")) { LAUNCH_COL(k_toy_relax_vel, m.myE, m); return 0; }
  if (!strcmp(name, "relax_zonal_temp") || !strcmp(name, "k_toy_relax_temp")) { LAUNCH_COL(k_toy_relax_temp, m.N, m); return 0; }
  return -1;
}
