/* ORACLE (test infrastructure): plan and application of this build's RAS-Chebyshev SSH preconditioner, see orc_ras.c */
#ifndef ORC_RAS_H
#define ORC_RAS_H
/* the defaults of fesom2_amd/csrc/ras_host.h */
#define ORC_RAS_PATCH_MAX 768
#define ORC_RAS_OVERLAP 4
#define ORC_RAS_DEG 16
#define ORC_RAS_KAPPA 200.0
#define ORC_RAS_THREADS 512
#define ORC_RAS_MAX_RPT 4
typedef struct {
  int n, P, NS, rpt, woff, deg;
  double lmax, inv_theta, c1[64], c2[64];
  int *perm, *inv, *pinfo, *extq;
  float *lv; unsigned short *lc; double *dsc;
} orc_ras_plan;
int orc_ras_build(int n, const int *rp, const int *ci, const double *vals, int patch_max, int overlap, int deg, double kappa, orc_ras_plan *pl);
void orc_ras_apply(const orc_ras_plan *pl, const double *x, double *z);
void orc_ras_free(orc_ras_plan *pl);
#endif
