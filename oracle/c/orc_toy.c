/* ORACLE (test infrastructure): Soufflet channel hooks of the reference's toy set-up
 * (src/toy_channel_soufflet.F90).  They sit on the step path when toy_ocean/which_toy='soufflet':
 *   compute_zonal_mean      before_oce_step every soufflet_forc_update=10 steps (oce_setup_step.F90:625-630)
 *   relax_zonal_vel         after solve_ssh_ale (oce_ale.F90:2696)
 *   relax_zonal_temp        after diff_tracers_ale of EVERY tracer (oce_ale_tracer.F90:150-151) -- it always relaxes
 *                           tracer 1, so temperature is relaxed once per tracer of the loop. */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NYBINS 100
static const double lat0 = 0.0, ysize = 2000000.0;

static double toy_dy(void) { double Ly = ysize / R_EARTH; return Ly / (double)NYBINS; }      /* :111-112 */
static double toy_tau_inv(void) { return 1.0 / 50.0 / 24.0 / 3600.0; }                      /* :19 */
#define COORD(j, n) C_.m.coord_nod2D[2 * ((n) - 1) + (j) - 1]
#define ZB(a, nz, b) (a)[(size_t)((b) - 1) * NLM1 + ((nz) - 1)]

/* compute_zonal_mean_ini (:104-155): bin of every element, number of elements per bin */
void orc_compute_zonal_mean_ini(void) {
  double dy = toy_dy();
  int myN = C_.m.myDim_nod2D, myE = C_.m.myDim_elem2D;
  memset(C_.toy_znum, 0, sizeof(double) * NLM1 * NYBINS);
  for (int e = 1; e <= myE; e++) {
    double ymean = ((COORD(2, EN(1, e)) + COORD(2, EN(2, e))) + COORD(2, EN(3, e))) / 3.0;
    C_.toy_bpos[e - 1] = (int)floor((ymean - lat0) / dy) + 1;
  }
  for (int r = 0; r < (C_.toy_nranks > 0 ? C_.toy_nranks : 1); r++)          /* per-rank partial sums, then MPI_SUM */
    for (int e = 1; e <= myE; e++) {
      if (EN(1, e) > myN) continue;
      if (C_.toy_nranks > 0 && C_.toy_owner[e - 1] != r) continue;
      for (int nz = 1; nz <= NLM1; nz++) ZB(C_.toy_znum, nz, C_.toy_bpos[e - 1]) += 1.0;
    }
}

/* compute_zonal_mean (:157-217).  With toy_nranks > 0 (tests against a multi-rank reference run) the sums are formed
 * per rank over the rank's elements and the partial sums added in rank order, as MPI_Allreduce(MPI_SUM) does. */
void orc_compute_zonal_mean(void) {
  int myN = C_.m.myDim_nod2D, myE = C_.m.myDim_elem2D;
  size_t nb = (size_t)NLM1 * NYBINS;
  int nr = C_.toy_nranks > 0 ? C_.toy_nranks : 1;
  double *pt = calloc(nb * nr, sizeof(double)), *pv = calloc(nb * nr, sizeof(double));
  for (int e = 1; e <= myE; e++) {
    if (EN(1, e) > myN) continue;
    int r = C_.toy_nranks > 0 ? C_.toy_owner[e - 1] : 0, b = C_.toy_bpos[e - 1];
    double *zt = pt + nb * r, *zv = pv + nb * r;
    for (int nz = 1; nz <= NLEV(e) - 1; nz++) {
      ZB(zt, nz, b) = ZB(zt, nz, b) + ((TR(nz, EN(1, e), 1) + TR(nz, EN(2, e), 1)) + TR(nz, EN(3, e), 1)) / 3.0;
      ZB(zv, nz, b) = ZB(zv, nz, b) + V2(C_.UV, 1, nz, e);
    }
  }
  for (size_t i = 0; i < nb; i++) {
    double st = pt[i], sv = pv[i];
    for (int r = 1; r < nr; r++) { st = st + pt[nb * r + i]; sv = sv + pv[nb * r + i]; }
    C_.toy_zvel[i] = sv / (C_.toy_znum[i] + 0.001);
    C_.toy_ztem[i] = st / (C_.toy_znum[i] + 0.001);
  }
  free(pt); free(pv);
}

/* linear interpolation between bin centres (:57-70, :89-100) */
static void toy_interp(double yy, double dy, int *nn, int *nn1, double *a) {
  *a = 0;
  if (yy < dy / 2) { *nn = 1; *nn1 = 1; }
  else {
    *nn = (int)floor(yy / dy - 0.5) + 1;
    *nn1 = *nn + 1;
    if (*nn1 > 100) *nn1 = *nn;
    *a = yy / dy + 0.5 - (double)(*nn);
  }
}

/* relax_zonal_vel (:46-79) */
void orc_relax_zonal_vel(void) {
  double dy = toy_dy(), dt = C_.p.dt, ti = toy_tau_inv();
  for (int e = 1; e <= C_.m.myDim_elem2D; e++) {
    double yy = ((COORD(2, EN(1, e)) + COORD(2, EN(2, e))) + COORD(2, EN(3, e))) / 3.0 - lat0, a;
    int nn, nn1;
    toy_interp(yy, dy, &nn, &nn1, &a);
    for (int nz = 1; nz <= NLEV(e) - 1; nz++) {
      double Uzon = (1.0 - a) * ZB(C_.toy_zvel, nz, nn) + a * ZB(C_.toy_zvel, nz, nn1);
      V2(C_.UV_rhs, 1, nz, e) = V2(C_.UV_rhs, 1, nz, e) + dt * ti * (A2(C_.Uclim, nz, e) - Uzon);
    }
  }
}

/* relax_zonal_temp (:81-103) */
void orc_relax_zonal_temp(void) {
  double dy = toy_dy(), dt = C_.p.dt, ti = toy_tau_inv();
  for (int n = 1; n <= C_.N; n++) {
    double yy = COORD(2, n) - lat0, a;
    int nn, nn1;
    toy_interp(yy, dy, &nn, &nn1, &a);
    for (int nz = 1; nz <= NLEVN(n) - 1; nz++) {
      double Tzon = (1.0 - a) * ZB(C_.toy_ztem, nz, nn) + a * ZB(C_.toy_ztem, nz, nn1);
      TR(nz, n, 1) = TR(nz, n, 1) + dt * ti * (A2(C_.Tclim, nz, n) - Tzon);
    }
  }
}

/* test hook: emulate the reduction order of an nranks-rank reference run (owner = rank of the element's first node) */
void orc_toy_set_partition(const int *owner, int nranks) {
  free(C_.toy_owner); C_.toy_owner = NULL; C_.toy_nranks = 0;
  if (owner && nranks > 0) {
    C_.toy_owner = malloc(sizeof(int) * C_.E);
    memcpy(C_.toy_owner, owner, sizeof(int) * C_.E);
    C_.toy_nranks = nranks;
  }
}
