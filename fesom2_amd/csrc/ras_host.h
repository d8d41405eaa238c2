// Host-side plan of the RAS-Chebyshev SSH preconditioner (csrc/precond_host.cpp builds it, csrc/api.hip uploads it,
// csrc/solver_ras.hip applies it).  Reference counterpart: the restricted additive Schwarz preconditioner of pARMS with one
// subdomain per MPI rank and ILU(k) subdomain solves (lib/parms/src/bicgstab_ras.c:49-259, parms_ilu_vcsr.c:651-1128),
// frozen at the first matrix (src/psolve.c:117-150).  Here: one subdomain ("patch") per workgroup, subdomain solve = a
// fixed number of Chebyshev steps on the Jacobi-scaled frozen operator, run entirely out of LDS and registers.
#pragma once
#include <vector>

// defaults (the CPU checker of the tests uses the same numbers; FESOM_GPU_RAS_* override them for experiments only)
#define RAS_PATCH_MAX 768      /* owned rows per patch */
#define RAS_OVERLAP 4          /* rings of overlap rows around a patch */
#define RAS_DEG 16             /* Chebyshev degree of the patch solve (first step is free: deg-1 products) */
#define RAS_KAPPA 200.0        /* the polynomial is optimal on [lmax / kappa, lmax] */
#define RAS_THREADS 512        /* threads per patch workgroup */
#define RAS_MAX_RPT 4          /* rows per thread at most -> 2048 rows per patch incl. overlap */
#define RAS_MAX_DEG 48

struct RasPlan {
  int n = 0, P = 0, NS = 0, rpt = 0, woff = 0, deg = 0, ovl = 0;
  double lmax = 0.0, kappa = 0.0, inv_theta = 0.0;
  std::vector<double> c1, c2;            // Chebyshev recurrence coefficients of step k = 1 .. deg-1 (index k)
  std::vector<int> perm, inv;            // perm[q] = row at position q of the patch order, inv[row] = q
  std::vector<int> pinfo;                // (4,P): first owned position, owned rows, offset into extq, rows incl. overlap
  std::vector<int> extq;                 // per patch: positions of its rows, owned first (consecutive), then the overlap rings
  std::vector<float> lv;                 // [P][woff][NS] off-diagonal entries a_ij / a_ii of the frozen operator inside the patch
  std::vector<unsigned short> lc;        // [P][woff][NS] their local columns (padding: the row itself with a zero entry)
  std::vector<double> dsc;               // [P][NS] 1 / (scale_i a_ii): row-scaled residual -> right-hand side of the Jacobi-scaled patch system
};

// n rows, CSR (rp, ci; first entry of a row = its diagonal; columns >= n are halo columns of a partition and ignored), frozen
// values, row scales (NULL: 1 / sum_j |a_ij| over the whole row).  Returns non-zero if the operator does not qualify.
int fesom_ras_build(int n, const int *rp, const int *ci, const double *vals, const double *scale, int patch_max, int overlap, int deg, double kappa, RasPlan &out);
