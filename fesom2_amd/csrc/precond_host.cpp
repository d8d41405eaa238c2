// Host set-up of the SSH preconditioner for pi-class operators: the EXPLICIT INVERSE of the row-scaled operator, frozen at
// the operator the run starts with -- the role pARMS' ILU factors play in the reference, which are computed in the first
// psolve call and reused for the whole run (src/psolve.c:117-150, lib/parms/src/parms_ilu_vcsr.c:651-1128).
// Why an explicit inverse: a 2-D operator with a few thousand rows (pi: 3140) is far too small for the GPU to be busy with
// sparse triangular solves (sequential) and its dense inverse is only n*n*4 bytes (pi: 42 MB in fp32), which 256 CUs stream
// in a few microseconds at HBM speed, so applying the preconditioner is ONE full-GPU matrix-vector product and BiCGstab
// needs 1-2 iterations instead of 20-30 (csrc/solver_xinv.hip).
//
// Algorithm (deterministic, no pivoting: the operator is a diagonally dominant M-matrix after row scaling):
//   1. A_s = diag(1/sum_j|a_ij|) A         (psolve.c:58-65)
//   2. reverse Cuthill-McKee ordering     (small bandwidth: pi 136)
//   3. banded LU of P A_s P^T
//   4. for every unit vector: forward / backward substitution, 16 right-hand sides at a time, threads over the blocks
//   5. Minv[i][j] (row-major, leading dimension ld, fp32) = (A_s^-1)_ij, un-permuted
//   6. sparsification: the inverse of this Helmholtz-type operator decays exponentially with distance (pi: 5 % of the entries
//      exceed 1e-6 of the largest); entries below tau * (largest entry of their row) are dropped (fesom_xinv_sparsify).  With
//      tau = 1e-4 pi keeps 82 entries per row (1.5 MB instead of 42 MB) and BiCGstab still needs 2 iterations.
// Every entry is formed by the same sequence of fp64 operations whatever the thread count or the block size, so the
// CPU checker of the tests (its own restatement of these steps) reproduces the matrix bit for bit.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {
// reverse Cuthill-McKee on the symmetrised pattern; ties by index; components in order of their lowest-degree node
void rcm_order(int n, const int *rp, const int *ci, std::vector<int> &order) {
  std::vector<std::vector<int>> adj(n);
  for (int i = 0; i < n; i++)
    for (int q = rp[i]; q < rp[i + 1]; q++) {
      const int j = ci[q];
      if (j == i || j < 0 || j >= n) continue;
      adj[i].push_back(j); adj[j].push_back(i);
    }
  std::vector<int> deg(n);
  for (int i = 0; i < n; i++) {
    std::sort(adj[i].begin(), adj[i].end());
    adj[i].erase(std::unique(adj[i].begin(), adj[i].end()), adj[i].end());
    deg[i] = (int)adj[i].size();
  }
  for (int i = 0; i < n; i++) std::sort(adj[i].begin(), adj[i].end(), [&](int a, int b) { return deg[a] != deg[b] ? deg[a] < deg[b] : a < b; });
  std::vector<char> seen(n, 0);
  std::vector<int> level(n), queue;
  order.clear(); order.reserve(n);
  auto bfs = [&](int start, std::vector<int> &out) {          // breadth first, neighbours by (degree, index); returns the depth
    out.clear(); out.push_back(start);
    level[start] = 0;
    std::vector<char> vis(n, 0);
    vis[start] = 1;
    for (size_t h = 0; h < out.size(); h++)
      for (int nb : adj[out[h]])
        if (!vis[nb] && !seen[nb]) { vis[nb] = 1; level[nb] = level[out[h]] + 1; out.push_back(nb); }
    return level[out.back()];
  };
  for (;;) {
    int start = -1;
    for (int i = 0; i < n; i++) if (!seen[i] && (start < 0 || deg[i] < deg[start])) start = i;
    if (start < 0) break;
    int depth = bfs(start, queue);
    for (int sweep = 0; sweep < 4; sweep++) {                  // pseudo-peripheral start node
      int cand = queue.back();
      for (int k = (int)queue.size() - 1; k >= 0 && level[queue[k]] == depth; k--)
        if (deg[queue[k]] < deg[cand] || (deg[queue[k]] == deg[cand] && queue[k] < cand)) cand = queue[k];
      std::vector<int> q2;
      int d2 = bfs(cand, q2);
      if (d2 <= depth) break;
      depth = d2; start = cand; queue.swap(q2);
    }
    bfs(start, queue);
    for (int v : queue) { seen[v] = 1; order.push_back(v); }
  }
  std::reverse(order.begin(), order.end());
}
}  // namespace

// n rows, 0-based CSR (rp, ci, vals), out: n rows of ld floats (ld >= n; the padding columns are set to 0).
// `scale` (optional): the row scales to use instead of 1/sum|a_ij| over the given entries -- a rank of a partitioned run inverts the
// owned-owned block of its rows but scales them like the solver does, with the sums over the WHOLE rows (halo columns included).
// Returns 0, or 1 if the factorisation meets a zero pivot.  *bandwidth (optional) receives the half bandwidth after RCM.
extern "C" int fesom_xinv_build(int n, const int *rp, const int *ci, const double *vals, const double *scale, int ld, float *out, int *bandwidth) {
  std::vector<int> order, pos(n);
  rcm_order(n, rp, ci, order);
  for (int k = 0; k < n; k++) pos[order[k]] = k;
  int bw = 0;
  for (int i = 0; i < n; i++)
    for (int q = rp[i]; q < rp[i + 1]; q++) bw = std::max(bw, std::abs(pos[i] - pos[ci[q]]));
  if (bandwidth) *bandwidth = bw;
  const size_t W = 2 * (size_t)bw + 1;
  std::vector<double> ab((size_t)n * W, 0.0);                   // band storage: ab[i*W + (j - i + bw)]
  for (int i = 0; i < n; i++) {
    double tmp = 0.;
    for (int q = rp[i]; q < rp[i + 1]; q++) tmp += fabs(vals[q]);
    const double sc = scale ? scale[i] : 1. / tmp;
    for (int q = rp[i]; q < rp[i + 1]; q++) ab[(size_t)pos[i] * W + (size_t)(pos[ci[q]] - pos[i] + bw)] = vals[q] * sc;
  }
  for (int k = 0; k < n; k++) {                                 // LU, L below the diagonal (unit), U on and above
    const double piv = ab[(size_t)k * W + bw];
    if (piv == 0.0) return 1;
    const int hi = std::min(n - 1, k + bw);
    for (int i = k + 1; i <= hi; i++) {
      double &lik = ab[(size_t)i * W + (size_t)(k - i + bw)];
      if (lik == 0.0) continue;
      lik = lik / piv;
      const double l = lik;
      double *ri = &ab[(size_t)i * W + (size_t)(bw - i)];        // ri[j] = a(i,j)
      const double *rk = &ab[(size_t)k * W + (size_t)(bw - k)];
      for (int j = k + 1; j <= hi; j++) ri[j] = ri[j] - l * rk[j];
    }
  }
  for (int i = 0; i < n; i++) memset(out + (size_t)i * ld, 0, sizeof(float) * (size_t)ld);
  constexpr int NB = 16;
  const int nblocks = (n + NB - 1) / NB;
  unsigned hc = std::thread::hardware_concurrency();
  int nthreads = (int)std::max(1u, std::min(hc ? hc : 1u, 16u));
  if (const char *e = getenv("FESOM_GPU_HOST_THREADS")) nthreads = std::max(1, atoi(e));
  nthreads = std::min(nthreads, nblocks);
  auto work = [&](int tid) {
    std::vector<double> y((size_t)n * NB);
    for (int b = tid; b < nblocks; b += nthreads) {
      const int c0 = b * NB, nc = std::min(NB, n - c0);
      std::fill(y.begin(), y.end(), 0.0);
      for (int r = 0; r < nc; r++) y[(size_t)(c0 + r) * NB + r] = 1.0;
      for (int i = c0 + 1; i < n; i++) {                        // L y = e : y_i -= l_ij y_j, j ascending
        double *yi = &y[(size_t)i * NB];
        const double *ri = &ab[(size_t)i * W + (size_t)(bw - i)];
        for (int j = std::max(c0, i - bw); j < i; j++) {
          const double l = ri[j];
          if (l == 0.0) continue;
          const double *yj = &y[(size_t)j * NB];
          for (int r = 0; r < NB; r++) yi[r] = yi[r] - l * yj[r];
        }
      }
      for (int i = n - 1; i >= 0; i--) {                        // U x = y : x_i = (y_i - sum_{j>i, ascending} u_ij x_j) / u_ii
        double *yi = &y[(size_t)i * NB];
        const double *ri = &ab[(size_t)i * W + (size_t)(bw - i)];
        const int hi = std::min(n - 1, i + bw);
        for (int j = i + 1; j <= hi; j++) {
          const double u = ri[j];
          if (u == 0.0) continue;
          const double *yj = &y[(size_t)j * NB];
          for (int r = 0; r < NB; r++) yi[r] = yi[r] - u * yj[r];
        }
        const double d = ri[i];
        for (int r = 0; r < NB; r++) yi[r] = yi[r] / d;
      }
      for (int i = 0; i < n; i++)
        for (int r = 0; r < nc; r++) out[(size_t)order[i] * ld + order[c0 + r]] = (float)y[(size_t)i * NB + r];
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nthreads; t++) th.emplace_back(work, t);
  work(0);
  for (auto &t : th) t.join();
  return 0;
}

// CSR of the entries with |M_ij| >= tau * max_j |M_ij| (columns ascending).  Call with cols == vals == nullptr to get the
// row pointer (n + 1 entries) only, then again with arrays of rowptr[n] entries.
extern "C" void fesom_xinv_sparsify(int n, int ld, const float *M, double tau, int *rowptr, unsigned short *cols, float *vals) {
  rowptr[0] = 0;
  for (int i = 0; i < n; i++) {
    const float *mr = M + (size_t)i * ld;
    double big = 0.0;
    for (int j = 0; j < n; j++) big = std::max(big, fabs((double)mr[j]));
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

// =====================================================================================================================
// RAS-Chebyshev preconditioner for operators that are too large for the explicit inverse (CORE2-class meshes, partitions):
// plan construction.  See ras_host.h for the reference counterpart.  Everything here is integer / graph work plus a few
// fp64 divisions per entry, formed in a fixed order, so that the CPU checker of the tests (its own restatement of the same
// rules) reproduces the plan bit for bit.
//   patches : recursive bisection of the row graph into L = ceil(n / patch_max) leaves.  One bisection of a set S: breadth-first
//             order from the smallest row of S, again from the row that order ends with (a far end of the set), rows the search
//             does not reach are appended by further searches from the smallest unvisited row; the first |S| * (L/2) / L rows of
//             the second order form the left part.  Neighbours are visited in CSR order.
//   order   : patches in the order the bisection emits them, rows of a patch in breadth-first order from its smallest row (= layout of
//             all solver vectors)
//   overlap : `overlap` rings of neighbouring rows (each ring in the order its rows are met from the previous one), as long as the
//             patch stays within 2048 rows
//   patch operator: a_ij / a_ii of the frozen operator for the columns inside the patch (Dirichlet condition outside), fp32, the
//             entries of a row sorted by their position in the patch
// =====================================================================================================================
#include "ras_host.h"
namespace {
struct RasGraph {
  int n; const int *rp, *ci;
  std::vector<int> mark;            // visit stamps of the searches
  std::vector<int> inset;           // id of the set a row currently belongs to
  int stamp = 0, setid = 0;
  void bfs(int start, int sid, std::vector<int> &out) {
    mark[start] = stamp; out.push_back(start);
    for (size_t h = out.size() - 1; h < out.size(); h++) {
      const int u = out[h];
      for (int q = rp[u]; q < rp[u + 1]; q++) {
        const int v = ci[q];
        if (v < 0 || v >= n || v == u || inset[v] != sid || mark[v] == stamp) continue;
        mark[v] = stamp; out.push_back(v);
      }
    }
  }
  void order(const std::vector<int> &S, int first, int sid, std::vector<int> &out) {    // S sorted by index
    stamp++; out.clear(); out.reserve(S.size());
    bfs(first, sid, out);
    if (out.size() < S.size())
      for (int u : S) if (mark[u] != stamp) bfs(u, sid, out);
  }
  void bisect(std::vector<int> &S, int L, std::vector<std::vector<int>> &patches) {
    if (L <= 1) { patches.push_back(S); return; }
    const int sid = ++setid;
    for (int u : S) inset[u] = sid;
    std::vector<int> o1, o2;
    order(S, S[0], sid, o1);
    order(S, o1.back(), sid, o2);
    const int nl = L / 2;
    const size_t cut = (size_t)((long long)S.size() * nl / L);
    std::vector<int> a(o2.begin(), o2.begin() + cut), b(o2.begin() + cut, o2.end());
    std::vector<int>().swap(o1); std::vector<int>().swap(o2); std::vector<int>().swap(S);
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    bisect(a, nl, patches); bisect(b, L - nl, patches);
  }
};
}  // namespace

int fesom_ras_build(int n, const int *rp, const int *ci, const double *vals, const double *scale, int patch_max, int overlap, int deg, double kappa, RasPlan &out) {
  if (n < 1 || patch_max < 1 || deg < 1 || deg > RAS_MAX_DEG || overlap < 0 || !(kappa > 1.0)) return 1;
  std::vector<int> dpos(n);
  for (int i = 0; i < n; i++) {                                  // position of the diagonal (first entry as a rule, oce_ale.F90:1128-1151)
    int d = -1;
    for (int q = rp[i]; q < rp[i + 1]; q++) if (ci[q] == i) { d = q; break; }
    if (d < 0 || vals[d] == 0.0) return 1;
    dpos[i] = d;
  }
  RasGraph g{n, rp, ci, std::vector<int>(n, 0), std::vector<int>(n, 0)};
  std::vector<std::vector<int>> patches;
  {
    std::vector<int> all(n);
    for (int i = 0; i < n; i++) all[i] = i;
    g.bisect(all, (n + patch_max - 1) / patch_max, patches);
  }
  const int P = (int)patches.size(), cap = RAS_THREADS * RAS_MAX_RPT;
  out = RasPlan();
  out.n = n; out.P = P; out.deg = deg; out.ovl = overlap; out.kappa = kappa;
  out.perm.resize(n); out.inv.resize(n); out.pinfo.assign(4 * (size_t)P, 0);
  {
    int q = 0;
    std::vector<int> bo;
    for (int p = 0; p < P; p++) {
      if ((int)patches[p].size() > cap) return 1;
      // rows of a patch in breadth-first order from its smallest row (neighbouring rows get neighbouring positions: few LDS bank
      // conflicts in the patch solve, coalesced gathers in the products with A_s)
      const int sid = ++g.setid;
      for (int r : patches[p]) g.inset[r] = sid;
      g.order(patches[p], patches[p][0], sid, bo);
      patches[p] = bo;
      out.pinfo[4 * p] = q; out.pinfo[4 * p + 1] = (int)patches[p].size();
      for (int r : patches[p]) { out.perm[q] = r; out.inv[r] = q; q++; }
    }
  }
  // overlap rings
  std::vector<std::vector<int>> ext(P);
  std::vector<int> member(n, -1);
  int ne_max = 0;
  for (int p = 0; p < P; p++) {
    std::vector<int> &e = ext[p];
    e = patches[p];
    for (int r : e) member[r] = p;
    size_t ring0 = 0;
    for (int k = 0; k < overlap; k++) {
      std::vector<int> cand;                                   // the next ring in the order its rows are met
      for (size_t h = ring0; h < e.size(); h++)
        for (int q = rp[e[h]]; q < rp[e[h] + 1]; q++) {
          const int v = ci[q];
          if (v >= 0 && v < n && member[v] != p && member[v] != -2 - p) { member[v] = -2 - p; cand.push_back(v); }
        }
      if (cand.empty() || e.size() + cand.size() > (size_t)cap) { for (int v : cand) member[v] = -1; break; }
      ring0 = e.size();
      for (int v : cand) { member[v] = p; e.push_back(v); }
    }
    for (int r : e) member[r] = -1;
    ne_max = std::max(ne_max, (int)e.size());
    out.pinfo[4 * p + 3] = (int)e.size();
  }
  out.rpt = std::max(2, (ne_max + RAS_THREADS - 1) / RAS_THREADS);
  out.NS = RAS_THREADS * out.rpt;
  // patch operators
  int maxoff = 0;
  std::vector<int> lidx(n, -1);
  for (int pass = 0; pass < 2; pass++) {
    if (pass == 1) {
      out.woff = maxoff <= 6 ? 6 : maxoff <= 9 ? 9 : 15;
      if (maxoff > 15) return 1;
      out.lv.assign((size_t)P * out.woff * out.NS, 0.0f);
      out.lc.assign((size_t)P * out.woff * out.NS, 0);
      out.dsc.assign((size_t)P * out.NS, 0.0);
    }
    int off = 0;
    for (int p = 0; p < P; p++) {
      const std::vector<int> &e = ext[p];
      for (size_t s = 0; s < e.size(); s++) lidx[e[s]] = (int)s;
      if (pass == 1) {
        out.pinfo[4 * p + 2] = off;
        for (int r : e) out.extq.push_back(out.inv[r]);
        for (int k = 0; k < out.woff; k++)
          for (int s = 0; s < out.NS; s++) out.lc[((size_t)p * out.woff + k) * out.NS + s] = (unsigned short)s;
      }
      for (size_t s = 0; s < e.size(); s++) {
        const int i = e[s];
        const double aii = vals[dpos[i]];
        int k = 0;
        std::pair<int, int> ent[64];                          // (local column, CSR position), sorted by local column: lanes of a wave read neighbouring LDS words
        for (int q = rp[i]; q < rp[i + 1]; q++) {
          const int c = ci[q];
          if (q == dpos[i] || c < 0 || c >= n || lidx[c] < 0) continue;
          if (k < 64) ent[k] = {lidx[c], q};
          k++;
        }
        if (k > 15) return 1;
        std::sort(ent, ent + k);
        if (pass == 1)
          for (int kk = 0; kk < k; kk++) {
            out.lv[((size_t)p * out.woff + kk) * out.NS + s] = (float)(vals[ent[kk].second] / aii);
            out.lc[((size_t)p * out.woff + kk) * out.NS + s] = (unsigned short)ent[kk].first;
          }
        maxoff = std::max(maxoff, k);
        if (pass == 1) {
          double sc;
          if (scale) sc = scale[i];
          else { double tmp = 0.; for (int q = rp[i]; q < rp[i + 1]; q++) tmp += fabs(vals[q]); sc = 1. / tmp; }
          const double dg = aii * sc;
          out.dsc[(size_t)p * out.NS + s] = 1.0 / dg;
        }
      }
      for (int r : e) lidx[r] = -1;
      off += (int)e.size();
    }
  }
  // Chebyshev coefficients: Gershgorin bound of the Jacobi-scaled operator, interval [lmax / kappa, lmax]
  double lmax = 0.0;
  for (int i = 0; i < n; i++) {
    const double aii = vals[dpos[i]];
    double s = 0.0;
    for (int q = rp[i]; q < rp[i + 1]; q++) if (ci[q] >= 0 && ci[q] < n) s += fabs(vals[q] / aii);
    lmax = std::max(lmax, s);
  }
  out.lmax = lmax;
  const double lmin = lmax / kappa, theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma = theta / delta;
  out.inv_theta = 1.0 / theta;
  out.c1.assign(deg + 1, 0.0); out.c2.assign(deg + 1, 0.0);
  double rho = 1.0 / sigma;
  for (int k = 1; k < deg; k++) {
    const double rn = 1.0 / (2.0 * sigma - rho);
    out.c1[k] = rn * rho; out.c2[k] = 2.0 * rn / delta;
    rho = rn;
  }
  return 0;
}

// Plan of the defaults as plain arrays (tests: the plan equals the CPU checker's, entry for entry).  dims = P, NS, rpt, woff, deg, rows in extq;
// with perm == NULL only dims is filled.  cheb: [0] 1/theta, [1+k] c1_k, [64+k] c2_k, [127] lmax.
extern "C" int fesom_ras_plan_export(int n, const int *rp, const int *ci, const double *vals, int *dims, int *perm, int *pinfo, int *extq, float *lv,
                                     unsigned short *lc, double *dsc, double *cheb) {
  RasPlan pl;
  if (fesom_ras_build(n, rp, ci, vals, nullptr, RAS_PATCH_MAX, RAS_OVERLAP, RAS_DEG, RAS_KAPPA, pl)) return 1;
  dims[0] = pl.P; dims[1] = pl.NS; dims[2] = pl.rpt; dims[3] = pl.woff; dims[4] = pl.deg; dims[5] = (int)pl.extq.size();
  if (!perm) return 0;
  memcpy(perm, pl.perm.data(), sizeof(int) * n); memcpy(pinfo, pl.pinfo.data(), sizeof(int) * pl.pinfo.size());
  memcpy(extq, pl.extq.data(), sizeof(int) * pl.extq.size()); memcpy(lv, pl.lv.data(), sizeof(float) * pl.lv.size());
  memcpy(lc, pl.lc.data(), sizeof(unsigned short) * pl.lc.size()); memcpy(dsc, pl.dsc.data(), sizeof(double) * pl.dsc.size());
  for (int k = 0; k < 128; k++) cheb[k] = 0.0;
  cheb[0] = pl.inv_theta; cheb[127] = pl.lmax;
  for (int k = 1; k < pl.deg; k++) { cheb[1 + k] = pl.c1[k]; cheb[64 + k] = pl.c2[k]; }
  return 0;
}
