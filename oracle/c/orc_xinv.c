/* ORACLE (test infrastructure): CPU restatement of the explicit-inverse SSH preconditioner of the HIP path
 * (fesom2_amd/csrc/precond_host.cpp builds it, fesom2_amd/csrc/solver.hip "xinv" applies it).  The reference has no
 * counterpart of this arithmetic -- its preconditioner is pARMS' RAS + ILU(2) (lib/parms/src/parms_ilu_vcsr.c:651-1128,
 * frozen after the first psolve call, src/psolve.c:117-150); the solution is compared with the reference's to the solver
 * tolerance (tests/test_gpu_parity.py, tests/test_gpu_dropin.py).  This file exists so that HIP == oracle stays a BITWISE
 * statement for whole steps: the matrix is formed by the same sequence of fp64 operations (row scaling, reverse
 * Cuthill-McKee, banded LU without pivoting, substitution with ascending column index), rounded to fp32, sparsified with the
 * same drop rule, and applied with the summation order of k_xi_gemv (64 lane-strided partial sums per row, fixed lane tree). */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static int *g_deg;
static int cmp_deg(const void *a, const void *b) {
  const int x = *(const int *)a, y = *(const int *)b;
  if (g_deg[x] != g_deg[y]) return g_deg[x] < g_deg[y] ? -1 : 1;
  return x < y ? -1 : (x > y);
}
static int cmp_int(const void *a, const void *b) { const int x = *(const int *)a, y = *(const int *)b; return x < y ? -1 : (x > y); }

/* breadth-first search over the unnumbered part, neighbours by (degree, index); returns the depth, queue in `out` (count *nout) */
static int bfs(int n, int start, int *const *adj, const int *deg, const char *seen, char *vis, int *level, int *out, int *nout) {
  memset(vis, 0, (size_t)n);
  int cnt = 0;
  out[cnt++] = start; vis[start] = 1; level[start] = 0;
  for (int h = 0; h < cnt; h++) {
    const int u = out[h];
    for (int q = 0; q < deg[u]; q++) {
      const int nb = adj[u][q];
      if (!vis[nb] && !seen[nb]) { vis[nb] = 1; level[nb] = level[u] + 1; out[cnt++] = nb; }
    }
  }
  *nout = cnt;
  return level[out[cnt - 1]];
}

/* n rows, 0-based CSR; out = n rows of ld floats.  Returns 0 or 1 (zero pivot). */
int orc_xinv_build(int n, const int *rp, const int *ci, const double *vals, int ld, float *out) {
  /* symmetrised adjacency without the diagonal, duplicates removed */
  int *cnt = calloc((size_t)n + 1, sizeof(int));
  for (int i = 0; i < n; i++)
    for (int q = rp[i]; q < rp[i + 1]; q++) { const int j = ci[q]; if (j != i) { cnt[i]++; cnt[j]++; } }
  int **adj = malloc(sizeof(int *) * (size_t)n), *deg = calloc((size_t)n, sizeof(int));
  for (int i = 0; i < n; i++) adj[i] = malloc(sizeof(int) * (size_t)(cnt[i] ? cnt[i] : 1));
  for (int i = 0; i < n; i++)
    for (int q = rp[i]; q < rp[i + 1]; q++) { const int j = ci[q]; if (j != i) { adj[i][deg[i]++] = j; adj[j][deg[j]++] = i; } }
  for (int i = 0; i < n; i++) {
    qsort(adj[i], (size_t)deg[i], sizeof(int), cmp_int);
    int u = 0;
    for (int q = 0; q < deg[i]; q++) if (q == 0 || adj[i][q] != adj[i][q - 1]) adj[i][u++] = adj[i][q];
    deg[i] = u;
  }
  g_deg = deg;
  for (int i = 0; i < n; i++) qsort(adj[i], (size_t)deg[i], sizeof(int), cmp_deg);
  char *seen = calloc((size_t)n, 1), *vis = malloc((size_t)n);
  int *level = malloc(sizeof(int) * (size_t)n), *queue = malloc(sizeof(int) * (size_t)n), *q2 = malloc(sizeof(int) * (size_t)n);
  int *order = malloc(sizeof(int) * (size_t)n), *pos = malloc(sizeof(int) * (size_t)n), nord = 0;
  for (;;) {
    int start = -1;
    for (int i = 0; i < n; i++) if (!seen[i] && (start < 0 || deg[i] < deg[start])) start = i;
    if (start < 0) break;
    int nq = 0, nq2 = 0;
    int depth = bfs(n, start, adj, deg, seen, vis, level, queue, &nq);
    for (int sweep = 0; sweep < 4; sweep++) {
      int cand = queue[nq - 1];
      for (int k = nq - 1; k >= 0 && level[queue[k]] == depth; k--)
        if (deg[queue[k]] < deg[cand] || (deg[queue[k]] == deg[cand] && queue[k] < cand)) cand = queue[k];
      const int d2 = bfs(n, cand, adj, deg, seen, vis, level, q2, &nq2);
      if (d2 <= depth) break;
      depth = d2; start = cand;
      memcpy(queue, q2, sizeof(int) * (size_t)nq2); nq = nq2;
    }
    bfs(n, start, adj, deg, seen, vis, level, queue, &nq);
    for (int k = 0; k < nq; k++) { seen[queue[k]] = 1; order[nord++] = queue[k]; }
  }
  for (int k = 0; k < n / 2; k++) { const int t = order[k]; order[k] = order[n - 1 - k]; order[n - 1 - k] = t; }
  for (int k = 0; k < n; k++) pos[order[k]] = k;
  int bw = 0;
  for (int i = 0; i < n; i++)
    for (int q = rp[i]; q < rp[i + 1]; q++) { const int d = abs(pos[i] - pos[ci[q]]); if (d > bw) bw = d; }
  const size_t W = 2 * (size_t)bw + 1;
  double *ab = calloc((size_t)n * W, sizeof(double));
#define AB(i, j) ab[(size_t)(i) * W + (size_t)((j) - (i) + bw)]
  for (int i = 0; i < n; i++) {
    double tmp = 0.;
    for (int q = rp[i]; q < rp[i + 1]; q++) tmp += fabs(vals[q]);
    const double sc = 1. / tmp;                            /* psolve.c:58-65 */
    for (int q = rp[i]; q < rp[i + 1]; q++) AB(pos[i], pos[ci[q]]) = vals[q] * sc;
  }
  int bad = 0;
  for (int k = 0; k < n && !bad; k++) {
    const double piv = AB(k, k);
    if (piv == 0.0) { bad = 1; break; }
    const int hi = k + bw < n - 1 ? k + bw : n - 1;
    for (int i = k + 1; i <= hi; i++) {
      if (AB(i, k) == 0.0) continue;
      AB(i, k) = AB(i, k) / piv;
      const double l = AB(i, k);
      for (int j = k + 1; j <= hi; j++) AB(i, j) = AB(i, j) - l * AB(k, j);
    }
  }
  if (!bad) {
    double *y = malloc(sizeof(double) * (size_t)n);
    for (int i = 0; i < n; i++) memset(out + (size_t)i * ld, 0, sizeof(float) * (size_t)ld);
    for (int c = 0; c < n; c++) {
      for (int i = 0; i < n; i++) y[i] = 0.0;
      y[c] = 1.0;
      for (int i = c + 1; i < n; i++) {
        const int lo = i - bw > c ? i - bw : c;
        for (int j = lo; j < i; j++) { const double l = AB(i, j); if (l != 0.0) y[i] = y[i] - l * y[j]; }
      }
      for (int i = n - 1; i >= 0; i--) {
        const int hi = i + bw < n - 1 ? i + bw : n - 1;
        for (int j = i + 1; j <= hi; j++) { const double u = AB(i, j); if (u != 0.0) y[i] = y[i] - u * y[j]; }
        y[i] = y[i] / AB(i, i);
      }
      for (int i = 0; i < n; i++) out[(size_t)order[i] * ld + order[c]] = (float)y[i];
    }
    free(y);
  }
#undef AB
  for (int i = 0; i < n; i++) free(adj[i]);
  free(adj); free(deg); free(cnt); free(seen); free(vis); free(level); free(queue); free(q2); free(order); free(pos); free(ab);
  return bad;
}

/* sparsification of the inverse (fesom_xinv_sparsify): CSR of the entries with |M_ij| >= tau * max_j|M_ij|, columns ascending.
 * Two passes: cols == NULL counts. */
void orc_xinv_sparsify(int n, int ld, const float *M, double tau, int *rowptr, unsigned short *cols, float *vals) {
  rowptr[0] = 0;
  for (int i = 0; i < n; i++) {
    const float *mr = M + (size_t)i * ld;
    double big = 0.0;
    for (int j = 0; j < n; j++) { const double a = fabs((double)mr[j]); if (a > big) big = a; }
    const double cut = tau * big;
    int q = rowptr[i];
    for (int j = 0; j < n; j++)
      if (fabs((double)mr[j]) >= cut && mr[j] != 0.0f) {
        if (cols) { cols[q] = (unsigned short)j; vals[q] = mr[j]; }
        q++;
      }
    rowptr[i + 1] = q;
  }
}

/* z = M x with the summation order of k_xi_gemv: lane l of 64 adds the entries l, l+64, ... of the row in that order, the 64 partial
 * sums go through the lane tree of the HIP reduction (16-lane rows: x[l] += x[l-s], s = 8,4,2,1; rows (R3+R2)+(R1+R0)) */
static double tree16(double *x) {
  for (int s = 8; s >= 1; s >>= 1)
    for (int l = 15; l >= 16 - s; l--) x[l] = x[l] + x[l - s];
  return x[15];
}
void orc_xinv_apply(int n, const int *mp, const unsigned short *mc, const float *mv, const double *x, double *z) {
  double part[64];
  for (int row = 0; row < n; row++) {
    for (int l = 0; l < 64; l++) {
      double a = 0.0;
      for (int e = mp[row] + l; e < mp[row + 1]; e += 64) a = a + (double)mv[e] * x[mc[e]];
      part[l] = a;
    }
    const double r0 = tree16(part), r1 = tree16(part + 16), r2 = tree16(part + 32), r3 = tree16(part + 48);
    z[row] = (r3 + r2) + (r1 + r0);
  }
}
