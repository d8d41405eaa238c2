/* MPI host adapter of the SSH solver: psolver_init / psolve / psolver_final with the signatures of the reference (src/psolve.c:16,117,152;
 * Fortran interface blocks src/oce_ale.F90:2272-2291) for runs with npes >= 1 MPI ranks, one GPU per rank.  Link it INSTEAD of src/psolve.c
 * and pARMS, together with libfesom_gpu.so.  Compiled by the integrator with the application's own mpi.h -- the
 * library itself holds no MPI.
 *
 * One rank: forwards to the library's single-partition solver (fesom_gpu_psolver_init / fesom_gpu_psolve).
 * Several ranks: psolver_init receives this rank's row block (part = prefix of the owned-row counts, cols = global contiguous numbering,
 * src/oce_ale.F90:1298-1344).  The adapter works out the halo of the block the way pARMS does from the same arguments (parms_map / parms_mat
 * set-up, lib/parms/src/parms_map.c, parms_comm.c): columns outside the own range, their owners from `part`, and -- one MPI_Alltoall of counts
 * plus one round of index messages -- which of its own rows every neighbour needs.  It hands that to fesom_gpu_psolver_init_dist with two
 * callbacks: the halo exchange of a solver vector (MPI_Isend / MPI_Irecv per neighbour, host-staged through fesom_gpu_copy; a GPU-aware MPI
 * may pass the device pointers straight to MPI instead, FESOM_GPU_MPI_DEVICE_BUFFERS=1) and the global sum of the partial dot products
 * (MPI_Allreduce, as lib/parms/src/parms_comm.c:205-356).  Errors: one line on stderr + MPI_Abort, as pARMS. */
#include <mpi.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fesom_gpu.h"

static struct {
  int dist, npes, mype, nr, ns, nrecv, nsend, device_buffers;
  MPI_Comm comm;
  int *rPE, *rcnt, *sPE, *scnt;
  double *hs, *hr;
  size_t cap;
  MPI_Request *req;
} A;

static void die(const char *what) {
  fprintf(stderr, "fesom_gpu_psolve_mpi (rank %d): %s: %s\n", A.mype, what, fesom_gpu_last_error());
  fflush(stderr);
  MPI_Abort(A.comm ? A.comm : MPI_COMM_WORLD, 3);
}

/* halo exchange of the packed node messages: block p of the send buffer (scnt[p] * W doubles) goes to sPE[p], block p of the receive buffer
 * comes from rPE[p]; blocks are consecutive in list order (include/fesom_gpu.h: fesom_transport) */
static int mpi_exchange(void *ctx, int kind, void *send_dev, void *recv_dev, int W) {
  (void)ctx;
  if (kind != 0) return 1;
  const size_t ns = (size_t)A.nsend * W, nr = (size_t)A.nrecv * W;
  double *sb = (double *)send_dev, *rb = (double *)recv_dev;
  if (!A.device_buffers) {
    if (ns + nr > A.cap) {
      free(A.hs);
      A.cap = 2 * (ns + nr);
      A.hs = (double *)malloc(A.cap * sizeof(double));
      if (!A.hs) return 1;
    }
    A.hr = A.hs + ns;
    if (fesom_gpu_copy(A.hs, send_dev, (long long)(ns * sizeof(double)), 0)) return 1;       /* (waits for the pack kernels) */
    sb = A.hs; rb = A.hr;
  } else if (fesom_gpu_sync()) return 1;
  int q = 0;
  size_t off = 0;
  for (int p = 0; p < A.nr; p++) { MPI_Irecv(rb + off, A.rcnt[p] * W, MPI_DOUBLE, A.rPE[p], 4711, A.comm, &A.req[q++]); off += (size_t)A.rcnt[p] * W; }
  off = 0;
  for (int p = 0; p < A.ns; p++) { MPI_Isend(sb + off, A.scnt[p] * W, MPI_DOUBLE, A.sPE[p], 4711, A.comm, &A.req[q++]); off += (size_t)A.scnt[p] * W; }
  MPI_Waitall(q, A.req, MPI_STATUSES_IGNORE);
  if (!A.device_buffers && fesom_gpu_copy(recv_dev, A.hr, (long long)(nr * sizeof(double)), 1)) return 1;
  return 0;
}
static int mpi_allreduce(void *ctx, void *buf_dev, int n) {
  (void)ctx;
  double tmp[16];
  if (n > 16) return 1;
  if (fesom_gpu_copy(tmp, buf_dev, (long long)(n * sizeof(double)), 0)) return 1;
  MPI_Allreduce(MPI_IN_PLACE, tmp, n, MPI_DOUBLE, MPI_SUM, A.comm);
  return fesom_gpu_copy(buf_dev, tmp, (long long)(n * sizeof(double)), 1);
}

static int cmp_int(const void *a, const void *b) { const int x = *(const int *)a, y = *(const int *)b; return (x > y) - (x < y); }
static int owner_of(const int *part, int npes, int g) {        /* part[r] <= g < part[r + 1] */
  int lo = 0, hi = npes;
  while (hi - lo > 1) { const int mid = (lo + hi) / 2; if (part[mid] <= g) lo = mid; else hi = mid; }
  return lo;
}

void psolver_init(int *id, int *stype, int *pctype, int *pcilutype, int *ilulevel, int *fillin, double *droptol, int *maxits, int *restart,
                  double *soltol, int *part, int *rptr, int *cols, double *vals, int *reuse, MPI_Fint *fcomm) {
  memset(&A, 0, sizeof(A));
  A.comm = MPI_Comm_f2c(*fcomm);
  MPI_Comm_size(A.comm, &A.npes);
  MPI_Comm_rank(A.comm, &A.mype);
  if (A.npes == 1) {
    int fc = (int)*fcomm;
    fesom_gpu_psolver_init(id, stype, pctype, pcilutype, ilulevel, fillin, droptol, maxits, restart, soltol, part, rptr, cols, vals, reuse, &fc);
    return;
  }
  A.dist = 1;
  { const char *e = getenv("FESOM_GPU_MPI_DEVICE_BUFFERS"); A.device_buffers = e && atoi(e) != 0; }
  const int npes = A.npes, me = A.mype, n = part[me + 1] - part[me], g0 = part[me], nza = rptr[n];
  /* halo = the distinct columns outside the own range, ascending (hence grouped by owner: the ranges of `part` ascend) */
  int *h = (int *)malloc(sizeof(int) * (size_t)(nza > 0 ? nza : 1)), nh = 0;
  for (int j = 0; j < nza; j++) if (cols[j] < g0 || cols[j] >= g0 + n) h[nh++] = cols[j];
  qsort(h, (size_t)nh, sizeof(int), cmp_int);
  int nu = 0;
  for (int k = 0; k < nh; k++) if (k == 0 || h[k] != h[k - 1]) h[nu++] = h[k];
  int *need = (int *)calloc((size_t)npes, sizeof(int)), *give = (int *)calloc((size_t)npes, sizeof(int));
  for (int k = 0; k < nu; k++) {
    if (h[k] < part[0] || h[k] >= part[npes]) { fprintf(stderr, "fesom_gpu_psolve_mpi (rank %d): column %d outside the global row range\n", me, h[k]); MPI_Abort(A.comm, 3); }
    need[owner_of(part, npes, h[k])]++;
  }
  MPI_Alltoall(need, 1, MPI_INT, give, 1, MPI_INT, A.comm);
  A.rPE = (int *)malloc(sizeof(int) * (size_t)npes); A.rcnt = (int *)malloc(sizeof(int) * (size_t)npes);
  A.sPE = (int *)malloc(sizeof(int) * (size_t)npes); A.scnt = (int *)malloc(sizeof(int) * (size_t)npes);
  for (int r = 0; r < npes; r++) {
    if (need[r]) { A.rPE[A.nr] = r; A.rcnt[A.nr++] = need[r]; A.nrecv += need[r]; }
    if (give[r]) { A.sPE[A.ns] = r; A.scnt[A.ns++] = give[r]; A.nsend += give[r]; }
  }
  A.req = (MPI_Request *)malloc(sizeof(MPI_Request) * (size_t)(A.nr + A.ns + 1));
  /* every neighbour learns which of its rows are wanted here (global ids; the send list becomes local row indices) */
  int *sl = (int *)malloc(sizeof(int) * (size_t)(A.nsend > 0 ? A.nsend : 1));
  int q = 0, off = 0;
  for (int p = 0; p < A.ns; p++) { MPI_Irecv(sl + off, A.scnt[p], MPI_INT, A.sPE[p], 4712, A.comm, &A.req[q++]); off += A.scnt[p]; }
  off = 0;
  for (int p = 0; p < A.nr; p++) { MPI_Isend(h + off, A.rcnt[p], MPI_INT, A.rPE[p], 4712, A.comm, &A.req[q++]); off += A.rcnt[p]; }
  MPI_Waitall(q, A.req, MPI_STATUSES_IGNORE);
  for (int k = 0; k < A.nsend; k++) sl[k] -= g0;
  static fesom_transport T;
  T.ctx = NULL; T.exchange = mpi_exchange; T.allreduce_sum = mpi_allreduce;
  if (fesom_gpu_psolver_init_dist(npes, me, part, rptr, cols, vals, *maxits, *soltol, A.nr, A.rPE, A.rcnt, h, A.ns, A.sPE, A.scnt, sl, &T)) die("psolver_init");
  free(h); free(sl); free(need); free(give);
}

void psolve(int *id, double *rhs, double *vals, double *sol, int *newvals) {
  if (!A.dist) { fesom_gpu_psolve(id, rhs, vals, sol, newvals); return; }
  if (fesom_gpu_psolve_dist(rhs, vals, sol, *newvals)) die("psolve");
}

void psolver_final(void) {
  fesom_gpu_psolver_final();
  free(A.rPE); free(A.rcnt); free(A.sPE); free(A.scnt); free(A.hs); free(A.req);
  memset(&A, 0, sizeof(A));
}
