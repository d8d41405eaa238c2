/* ORACLE (test infrastructure): restatement of this build's RAS-Chebyshev SSH preconditioner -- plan construction
 * (fesom2_amd/csrc/precond_host.cpp: fesom_ras_build) and application (fesom2_amd/csrc/solver_ras.hip: k_ras_apply).
 * Not a restatement of pARMS: the reference's preconditioner is RAS with one subdomain per MPI rank and ILU(k) subdomain
 * solves (lib/parms/src/bicgstab_ras.c:49-259); sequential triangular solves have no GPU form, so the build replaces it
 * (patches of the row graph, Chebyshev patch solves) and keeps what defines the answer: row scaling, BiCGstab, stop rule
 * (orc_core.c).  The solution agrees with the reference to the solver tolerance; this file exists so that HIP == oracle
 * can be checked bit for bit.
 *
 * Rules (csrc/precond_host.cpp states them too):
 *   patches : recursive bisection of the row graph into L = ceil(n / patch_max) leaves; one bisection of a set S (kept sorted):
 *             breadth-first order from S's smallest row, then again from the row that order ends with, unreached rows by further
 *             searches from the smallest unvisited row; neighbours in CSR order; the first |S| * (L/2) / L rows of the second
 *             order are the left part (L/2 leaves), the rest the right part.
 *   layout  : patches in emission order (left before right), rows of a patch in breadth-first order from its smallest row.
 *   overlap : up to `overlap` rings, each ring = the not yet included neighbours of the previous ring in the order they are met; a
 *             ring that would take the patch beyond 2048 rows ends the growth.
 *   patch operator: off-diagonal a_ij / a_ii (fp32) for columns inside the patch, sorted by position in the patch, ELL-padded with (0, own row);
 *             right-hand side scale 1 / (a_ii / sum_j |a_ij|).
 *   Chebyshev: interval [lmax / kappa, lmax], lmax = max_i sum_j |a_ij / a_ii|; z_1 = rhs / theta, deg - 1 further steps. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "orc_ras.h"

#define CAP (ORC_RAS_THREADS * ORC_RAS_MAX_RPT)

typedef struct {
  int n; const int *rp, *ci;
  int *mark, *inset, stamp, setid;
  int **leaf; int *leaf_n; int nleaf, leaf_cap;
} graph_t;

static int cmp_int(const void *a, const void *b) { int x = *(const int *)a, y = *(const int *)b; return (x > y) - (x < y); }

static int bfs_from(graph_t *g, int start, int sid, int *out, int cnt) {
  int head = cnt;
  g->mark[start] = g->stamp; out[cnt++] = start;
  while (head < cnt) {
    const int u = out[head++];
    for (int q = g->rp[u]; q < g->rp[u + 1]; q++) {
      const int v = g->ci[q];
      if (v < 0 || v >= g->n || v == u) continue;
      if (g->inset[v] != sid || g->mark[v] == g->stamp) continue;
      g->mark[v] = g->stamp; out[cnt++] = v;
    }
  }
  return cnt;
}
static void set_order(graph_t *g, const int *S, int ns, int first, int sid, int *out) {
  g->stamp++;
  int cnt = bfs_from(g, first, sid, out, 0);
  for (int k = 0; k < ns && cnt < ns; k++) if (g->mark[S[k]] != g->stamp) cnt = bfs_from(g, S[k], sid, out, cnt);
}
static void split(graph_t *g, int *S, int ns, int L) {          /* takes ownership of S */
  if (L <= 1) {
    if (g->nleaf == g->leaf_cap) {
      g->leaf_cap = g->leaf_cap ? 2 * g->leaf_cap : 64;
      g->leaf = realloc(g->leaf, sizeof(int *) * g->leaf_cap); g->leaf_n = realloc(g->leaf_n, sizeof(int) * g->leaf_cap);
    }
    g->leaf[g->nleaf] = S; g->leaf_n[g->nleaf] = ns; g->nleaf++;
    return;
  }
  const int sid = ++g->setid;
  for (int k = 0; k < ns; k++) g->inset[S[k]] = sid;
  int *o1 = malloc(sizeof(int) * ns), *o2 = malloc(sizeof(int) * ns);
  set_order(g, S, ns, S[0], sid, o1);
  set_order(g, S, ns, o1[ns - 1], sid, o2);
  free(o1); free(S);
  const int nl = L / 2;
  const int cut = (int)((long long)ns * nl / L);
  int *a = malloc(sizeof(int) * (cut ? cut : 1)), *b = malloc(sizeof(int) * (ns - cut ? ns - cut : 1));
  memcpy(a, o2, sizeof(int) * cut); memcpy(b, o2 + cut, sizeof(int) * (ns - cut));
  free(o2);
  qsort(a, cut, sizeof(int), cmp_int); qsort(b, ns - cut, sizeof(int), cmp_int);
  split(g, a, cut, nl);
  split(g, b, ns - cut, L - nl);
}

void orc_ras_free(orc_ras_plan *pl) {
  free(pl->perm); free(pl->inv); free(pl->pinfo); free(pl->extq); free(pl->lv); free(pl->lc); free(pl->dsc);
  memset(pl, 0, sizeof(*pl));
}

int orc_ras_build(int n, const int *rp, const int *ci, const double *vals, int patch_max, int overlap, int deg, double kappa, orc_ras_plan *pl) {
  memset(pl, 0, sizeof(*pl));
  if (n < 1 || deg < 1 || deg >= 64) return 1;
  int *dpos = malloc(sizeof(int) * n);
  for (int i = 0; i < n; i++) {
    dpos[i] = -1;
    for (int q = rp[i]; q < rp[i + 1]; q++) if (ci[q] == i) { dpos[i] = q; break; }
    if (dpos[i] < 0 || vals[dpos[i]] == 0.0) { free(dpos); return 1; }
  }
  graph_t g;
  memset(&g, 0, sizeof(g));
  g.n = n; g.rp = rp; g.ci = ci; g.mark = calloc(n, sizeof(int)); g.inset = calloc(n, sizeof(int));
  int *all = malloc(sizeof(int) * n);
  for (int i = 0; i < n; i++) all[i] = i;
  split(&g, all, n, (n + patch_max - 1) / patch_max);
  const int P = g.nleaf;
  pl->n = n; pl->P = P; pl->deg = deg;
  pl->perm = malloc(sizeof(int) * n); pl->inv = malloc(sizeof(int) * n); pl->pinfo = calloc(4 * (size_t)P, sizeof(int));
  for (int p = 0, q = 0; p < P; p++) {
    {                                                       /* rows of a patch in breadth-first order from its smallest row */
      const int sid = ++g.setid, ns = g.leaf_n[p];
      int *o = malloc(sizeof(int) * ns);
      for (int k = 0; k < ns; k++) g.inset[g.leaf[p][k]] = sid;
      set_order(&g, g.leaf[p], ns, g.leaf[p][0], sid, o);
      free(g.leaf[p]); g.leaf[p] = o;
    }
    pl->pinfo[4 * p] = q; pl->pinfo[4 * p + 1] = g.leaf_n[p];
    for (int k = 0; k < g.leaf_n[p]; k++, q++) { pl->perm[q] = g.leaf[p][k]; pl->inv[g.leaf[p][k]] = q; }
  }
  /* rings of overlap */
  int **ext = malloc(sizeof(int *) * P), *ext_n = malloc(sizeof(int) * P), *member = malloc(sizeof(int) * n), *cand = malloc(sizeof(int) * (size_t)CAP * 17);
  for (int i = 0; i < n; i++) member[i] = -1;
  int ne_max = 0, tot = 0;
  for (int p = 0; p < P; p++) {
    int *e = malloc(sizeof(int) * CAP), ne = g.leaf_n[p];
    if (ne > CAP) return 1;
    memcpy(e, g.leaf[p], sizeof(int) * ne);
    for (int k = 0; k < ne; k++) member[e[k]] = p;
    int ring0 = 0;
    for (int r = 0; r < overlap; r++) {
      int nu = 0;                                           /* the next ring, rows in the order they are met */
      for (int h = ring0; h < ne; h++)
        for (int q = rp[e[h]]; q < rp[e[h] + 1]; q++) {
          const int v = ci[q];
          if (v < 0 || v >= n || member[v] == p || member[v] == -2 - p) continue;
          member[v] = -2 - p;
          if (nu < CAP) cand[nu] = v;
          nu++;
        }
      if (nu == 0 || ne + nu > CAP) {
        for (int k = 0; k < nu && k < CAP; k++) member[cand[k]] = -1;
        if (nu > CAP) for (int i = 0; i < n; i++) if (member[i] == -2 - p) member[i] = -1;
        break;
      }
      ring0 = ne;
      for (int k = 0; k < nu; k++) { member[cand[k]] = p; e[ne++] = cand[k]; }
    }
    for (int k = 0; k < ne; k++) member[e[k]] = -1;
    ext[p] = e; ext_n[p] = ne; pl->pinfo[4 * p + 3] = ne; pl->pinfo[4 * p + 2] = tot; tot += ne;
    if (ne > ne_max) ne_max = ne;
  }
  free(cand);
  pl->rpt = (ne_max + ORC_RAS_THREADS - 1) / ORC_RAS_THREADS;
  if (pl->rpt < 2) pl->rpt = 2;
  pl->NS = ORC_RAS_THREADS * pl->rpt;
  const int NS = pl->NS;
  /* widest patch row, then the patch operators */
  int *lidx = member;                                     /* all -1 again */
  int maxoff = 0;
  for (int p = 0; p < P; p++) {
    for (int s = 0; s < ext_n[p]; s++) lidx[ext[p][s]] = s;
    for (int s = 0; s < ext_n[p]; s++) {
      const int i = ext[p][s];
      int k = 0;
      for (int q = rp[i]; q < rp[i + 1]; q++) if (q != dpos[i] && ci[q] >= 0 && ci[q] < n && lidx[ci[q]] >= 0) k++;
      if (k > maxoff) maxoff = k;
    }
    for (int s = 0; s < ext_n[p]; s++) lidx[ext[p][s]] = -1;
  }
  if (maxoff > 15) return 1;
  pl->woff = maxoff <= 6 ? 6 : maxoff <= 9 ? 9 : 15;
  const int WO = pl->woff;
  pl->extq = malloc(sizeof(int) * (tot ? tot : 1));
  pl->lv = calloc((size_t)P * WO * NS, sizeof(float)); pl->lc = malloc(sizeof(unsigned short) * (size_t)P * WO * NS); pl->dsc = calloc((size_t)P * NS, sizeof(double));
  for (int p = 0; p < P; p++) {
    for (int k = 0; k < WO; k++) for (int s = 0; s < NS; s++) pl->lc[((size_t)p * WO + k) * NS + s] = (unsigned short)s;
    for (int s = 0; s < ext_n[p]; s++) { lidx[ext[p][s]] = s; pl->extq[pl->pinfo[4 * p + 2] + s] = pl->inv[ext[p][s]]; }
    for (int s = 0; s < ext_n[p]; s++) {
      const int i = ext[p][s];
      const double aii = vals[dpos[i]];
      int k = 0, col[16], pos[16];
      double tmp = 0.;
      for (int q = rp[i]; q < rp[i + 1]; q++) {
        tmp += fabs(vals[q]);
        if (q == dpos[i] || ci[q] < 0 || ci[q] >= n || lidx[ci[q]] < 0) continue;
        int h = k++;                                        /* insertion by increasing local column */
        while (h > 0 && col[h - 1] > lidx[ci[q]]) { col[h] = col[h - 1]; pos[h] = pos[h - 1]; h--; }
        col[h] = lidx[ci[q]]; pos[h] = q;
      }
      for (int kk = 0; kk < k; kk++) {
        pl->lv[((size_t)p * WO + kk) * NS + s] = (float)(vals[pos[kk]] / aii);
        pl->lc[((size_t)p * WO + kk) * NS + s] = (unsigned short)col[kk];
      }
      const double sc = 1. / tmp, dg = aii * sc;
      pl->dsc[(size_t)p * NS + s] = 1.0 / dg;
    }
    for (int s = 0; s < ext_n[p]; s++) lidx[ext[p][s]] = -1;
  }
  double lmax = 0.0;
  for (int i = 0; i < n; i++) {
    const double aii = vals[dpos[i]];
    double s = 0.0;
    for (int q = rp[i]; q < rp[i + 1]; q++) if (ci[q] >= 0 && ci[q] < n) s += fabs(vals[q] / aii);
    if (s > lmax) lmax = s;
  }
  pl->lmax = lmax;
  const double lmin = lmax / kappa, theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma = theta / delta;
  pl->inv_theta = 1.0 / theta;
  double rho = 1.0 / sigma;
  for (int k = 1; k < deg; k++) {
    const double rn = 1.0 / (2.0 * sigma - rho);
    pl->c1[k] = rn * rho; pl->c2[k] = 2.0 * rn / delta;
    rho = rn;
  }
  for (int p = 0; p < P; p++) { free(ext[p]); free(g.leaf[p]); }
  free(ext); free(ext_n); free(member); free(dpos); free(g.leaf); free(g.leaf_n); free(g.mark); free(g.inset);
  return 0;
}

/* z = M x in the patch order (x, z: n values); every patch on its own, owned rows written */
void orc_ras_apply(const orc_ras_plan *pl, const double *x, double *z) {
  const int NS = pl->NS, WO = pl->woff;
  double *rb = malloc(sizeof(double) * NS * 5), *d = rb + NS, *zc = d + NS, *za = zc + NS, *zn = za + NS;
  for (int p = 0; p < pl->P; p++) {
    const int own0 = pl->pinfo[4 * p], no = pl->pinfo[4 * p + 1], eoff = pl->pinfo[4 * p + 2], ne = pl->pinfo[4 * p + 3];
    for (int s = 0; s < NS; s++) {
      double val = 0.0;
      if (s < ne) val = x[s < no ? own0 + s : pl->extq[eoff + s]];
      rb[s] = val * pl->dsc[(size_t)p * NS + s];
      d[s] = rb[s] * pl->inv_theta; zc[s] = d[s]; za[s] = zc[s];
    }
    for (int k = 1; k < pl->deg; k++) {
      for (int s = 0; s < NS; s++) {
        double acc = 0.0;
        for (int kk = 0; kk < WO; kk++) acc = acc + (double)pl->lv[((size_t)p * WO + kk) * NS + s] * za[pl->lc[((size_t)p * WO + kk) * NS + s]];
        const double res = (rb[s] - zc[s]) - acc;
        d[s] = pl->c1[k] * d[s] + pl->c2[k] * res;
        zc[s] = zc[s] + d[s];
        zn[s] = zc[s];
      }
      double *t = za; za = zn; zn = t;
    }
    for (int s = 0; s < no; s++) z[own0 + s] = zc[s];
  }
  free(rb);
}

/* the plan of the defaults as plain arrays, same layout as fesom_ras_plan_export of the product (tests compare the two) */
int orc_ras_plan_export(int n, const int *rp, const int *ci, const double *vals, int *dims, int *perm, int *pinfo, int *extq, float *lv,
                        unsigned short *lc, double *dsc, double *cheb) {
  orc_ras_plan pl;
  if (orc_ras_build(n, rp, ci, vals, ORC_RAS_PATCH_MAX, ORC_RAS_OVERLAP, ORC_RAS_DEG, ORC_RAS_KAPPA, &pl)) return 1;
  const int next = pl.pinfo[4 * (pl.P - 1) + 2] + pl.pinfo[4 * (pl.P - 1) + 3];
  dims[0] = pl.P; dims[1] = pl.NS; dims[2] = pl.rpt; dims[3] = pl.woff; dims[4] = pl.deg; dims[5] = next;
  if (perm) {
    memcpy(perm, pl.perm, sizeof(int) * n); memcpy(pinfo, pl.pinfo, sizeof(int) * 4 * pl.P); memcpy(extq, pl.extq, sizeof(int) * next);
    memcpy(lv, pl.lv, sizeof(float) * (size_t)pl.P * pl.woff * pl.NS); memcpy(lc, pl.lc, sizeof(unsigned short) * (size_t)pl.P * pl.woff * pl.NS);
    memcpy(dsc, pl.dsc, sizeof(double) * (size_t)pl.P * pl.NS);
    for (int k = 0; k < 128; k++) cheb[k] = 0.0;
    cheb[0] = pl.inv_theta; cheb[127] = pl.lmax;
    for (int k = 1; k < pl.deg; k++) { cheb[1 + k] = pl.c1[k]; cheb[64 + k] = pl.c2[k]; }
  }
  orc_ras_free(&pl);
  return 0;
}
