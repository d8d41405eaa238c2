/* ORACLE (test infrastructure): Gent-McWilliams bolus velocities after Ferrari et al. 2010, src/oce_fer_gm.F90, and the
 * places of the step where they enter: fer_Wvel in vert_vel_ale (src/oce_ale.F90:1720-1811) and the temporary addition
 * of the bolus velocities around the tracer loop (solve_tracers_ale, src/oce_ale_tracer.F90:127-131,165-169). */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define G3(a, c, nz, n) (a)[((size_t)((n) - 1) * NL + ((nz) - 1)) * 2 + ((c) - 1)]      /* (2, nl, N) */

/* static horizontal part of the GM scaling (init_Redi_GM :204-232 for scaling_Rossby = .false.) */
void orc_gm_static(void) {
  for (int n = 1; n <= C_.m.myDim_nod2D; n++) {
    double reso = C_.m.mesh_resolution[n - 1], scaling = 1.;
    if (C_.p.scaling_resolution) scaling = scaling * pow(reso / 100000., C_.p.K_GM_resscalorder);
    if (reso / 1000.0 < C_.p.K_GM_rampmax) scaling = scaling * dmax((reso / 1000.0 - C_.p.K_GM_rampmin) / (C_.p.K_GM_rampmax - C_.p.K_GM_rampmin), 0.);
    C_.gm_scal_static[n - 1] = scaling;
  }
}

/* init_Redi_GM (:159-340) */
void orc_init_Redi_GM(void) {
  const double c_min = 0.5, pi = 3.14159265358979;
  double zscaling[80];
  const int gm = C_.p.Fer_GM, redi = C_.p.Redi;
  for (int n = 1; n <= C_.m.myDim_nod2D; n++) {
    int nzmax = C_.m.nlevels_nod2D_min[n - 1], nzmin = C_.m.ulevels_nod2D_max[n - 1];   /* min / max over the node's elements */
    double reso = C_.m.mesh_resolution[n - 1];
    if (gm) {
      double c1 = 0.;
      for (int nz = nzmin; nz <= nzmax - 1; nz++)
        c1 = c1 + A2(C_.hnode_new, nz, n) * (sqrt(fabs(dmax(A2L(C_.bvfreq, nz, n), 0.))) + sqrt(fabs(dmax(A2L(C_.bvfreq, nz + 1, n), 0.)))) / 2.;
      c1 = dmax(c_min, c1 / pi);
      double scaling = C_.gm_scal_static[n - 1];
      if (C_.p.scaling_Rossby) {          /* :196-200: cut K_GM off where the mesh resolves the Rossby radius (Fermi function of resolution / radius) */
        const double f_min = 1.e-6, r_max = 200000., x0 = 1.5, sigma = .15;
        double rosb = dmin(c1 / dmax(fabs(C_.m.coriolis_node[n - 1]), f_min), r_max);
        double rr_ratio = dmin(reso / rosb, 5.);
        scaling = 1. / (1. + exp(-(rr_ratio - x0) / sigma));
        if (C_.p.scaling_resolution) scaling = scaling * pow(reso / 100000., C_.p.K_GM_resscalorder);
        if (reso / 1000.0 < C_.p.K_GM_rampmax) scaling = scaling * dmax((reso / 1000.0 - C_.p.K_GM_rampmin) / (C_.p.K_GM_rampmax - C_.p.K_GM_rampmin), 0.);
      }
      C_.fer_scal[n - 1] = dmin(scaling, 1.0);
      A2L(C_.fer_K, nzmin, n) = C_.fer_scal[n - 1] * C_.p.K_GM_max;
      A2L(C_.fer_K, nzmin, n) = dmax(A2L(C_.fer_K, nzmin, n), C_.p.K_GM_min);
      C_.fer_c[n - 1] = c1 * c1;
    }
    if (redi) { double q = reso / 100000.0; A2(C_.Ki, nzmin, n) = C_.p.K_hor * (q * q); }
  }
  if (redi && gm) {                                        /* "like in FESOM 1.4 we make Redi equal GM": Ki(nzmin,:)=fer_k(nzmin,:) OUTSIDE the node loop (:249-250), i.e.
                                                            * at the level the loop left in nzmin -- the upper level of the LAST owned node -- for every node */
    const int nzl = C_.m.ulevels_nod2D_max[C_.m.myDim_nod2D - 1];
    for (int n = 1; n <= C_.N; n++) A2(C_.Ki, nzl, n) = A2L(C_.fer_K, nzl, n);
  }
  for (int n = 1; n <= C_.m.myDim_nod2D; n++) {
    int nzmax = NLEVN(n), nzmin = ULEVN(n);
    if (C_.p.scaling_Ferreira) {
      double bvref;
      int mi = C_.MLD1_ind[n - 1];
      if (C_.p.K_GM_bvref == 0) bvref = dmax(A2L(C_.bvfreq, nzmin, n), 1.e-6);
      else if (C_.p.K_GM_bvref == 1) bvref = dmax(A2L(C_.bvfreq, mi + 1, n), 1.e-6);
      else {
        double sm = 0.;
        for (int nz = nzmin; nz <= mi; nz++) sm = sm + A2L(C_.bvfreq, nz, n);
        bvref = dmax(sm / (double)mi, 1.e-6);
      }
      for (int nz = nzmin; nz <= nzmax; nz++) { zscaling[nz] = dmax(A2L(C_.bvfreq, nz, n) / bvref, 0.2); zscaling[nz] = dmin(zscaling[nz], 1.0); }
    } else for (int nz = 0; nz < 80; nz++) zscaling[nz] = 1.0;
    if (C_.p.scaling_FESOM14)
      for (int nz = nzmin; nz <= nzmax; nz++) { int k = nz < NL - 1 ? nz : NL - 1; if (V3(C_.neutral_slope, 3, k, n) > 5.e-3) zscaling[nz] = 0.0; }
    if (gm) {
      for (int nz = nzmin + 1; nz <= nzmax; nz++) A2L(C_.fer_K, nz, n) = A2L(C_.fer_K, nzmin, n) * zscaling[nz];
      A2L(C_.fer_K, nzmin, n) = A2L(C_.fer_K, nzmin, n) * zscaling[nzmin];
    }
    if (redi) {
      for (int nz = nzmin + 1; nz <= nzmax - 1; nz++) A2(C_.Ki, nz, n) = A2(C_.Ki, nzmin, n) * 0.5 * (zscaling[nz] + zscaling[nz + 1]);
      A2(C_.Ki, nzmin, n) = A2(C_.Ki, nzmin, n) * 0.5 * (zscaling[nzmin] + zscaling[nzmin + 1]);
    }
  }
}

/* fer_solve_Gamma (:8-120) */
void orc_fer_solve_Gamma(void) {
  double zbar_n[80], Z_n[80], a[80], b[80], c[80], cp[80], tp1[80], tp2[80], t1[80], t2[80];
  for (int n = 1; n <= C_.m.myDim_nod2D; n++) {
    int nzmax = NLEVN(n), nzmin = ULEVN(n);
    for (int k = 0; k < 80; k++) { zbar_n[k] = 0.; Z_n[k] = 0.; }
    zbar_n[nzmax] = C_.m.zbar_n_bot[n - 1];
    Z_n[nzmax - 1] = zbar_n[nzmax] + A2(C_.hnode_new, nzmax - 1, n) / 2.0;
    for (int nz = nzmax - 1; nz >= nzmin + 1; nz--) {
      zbar_n[nz] = zbar_n[nz + 1] + A2(C_.hnode_new, nz, n);
      Z_n[nz - 1] = zbar_n[nz] + A2(C_.hnode_new, nz - 1, n) / 2.0;
    }
    zbar_n[nzmin] = zbar_n[nzmin + 1] + A2(C_.hnode_new, nzmin, n);
    nzmax = C_.m.nlevels_nod2D_min[n - 1]; nzmin = C_.m.ulevels_nod2D_max[n - 1];
    c[nzmin] = 0.; a[nzmin] = 0.; b[nzmin] = 1.;
    double zinv2 = 1.0 / (zbar_n[nzmin] - zbar_n[nzmin + 1]);
    for (int nz = nzmin + 1; nz <= nzmax - 1; nz++) {
      double zinv1 = zinv2;
      zinv2 = 1.0 / (zbar_n[nz] - zbar_n[nz + 1]);
      double zinv = 1.0 / (Z_n[nz - 1] - Z_n[nz]);
      a[nz] = C_.fer_c[n - 1] * zinv1 * zinv;
      c[nz] = C_.fer_c[n - 1] * zinv2 * zinv;
      b[nz] = -a[nz] - c[nz] - dmax(A2L(C_.bvfreq, nz, n), 1.e-8);
    }
    c[nzmax] = 0.; a[nzmax] = 0.; b[nzmax] = 1.;
    t1[nzmin] = 0.; t2[nzmin] = 0.; t1[nzmax] = 0.; t2[nzmax] = 0.;
    for (int nz = nzmin + 1; nz <= nzmax - 1; nz++) {
      double r = G_ACC / DENSITY_0;
      t1[nz] = r * 0.5 * (V2(C_.sigma_xy, 1, nz - 1, n) + V2(C_.sigma_xy, 1, nz, n)) * A2L(C_.fer_K, nz, n);
      t2[nz] = r * 0.5 * (V2(C_.sigma_xy, 2, nz - 1, n) + V2(C_.sigma_xy, 2, nz, n)) * A2L(C_.fer_K, nz, n);
    }
    cp[nzmin] = c[nzmin] / b[nzmin]; tp1[nzmin] = t1[nzmin] / b[nzmin]; tp2[nzmin] = t2[nzmin] / b[nzmin];
    for (int nz = nzmin + 1; nz <= nzmax; nz++) {
      double mm = b[nz] - cp[nz - 1] * a[nz];
      cp[nz] = c[nz] / mm;
      tp1[nz] = (t1[nz] - tp1[nz - 1] * a[nz]) / mm;
      tp2[nz] = (t2[nz] - tp2[nz - 1] * a[nz]) / mm;
    }
    t1[nzmax] = tp1[nzmax]; t2[nzmax] = tp2[nzmax];
    for (int nz = nzmax - 1; nz >= nzmin; nz--) { t1[nz] = tp1[nz] - cp[nz] * t1[nz + 1]; t2[nz] = tp2[nz] - cp[nz] * t2[nz + 1]; }
    for (int nz = nzmin; nz <= nzmax; nz++) { G3(C_.fer_gamma, 1, nz, n) = t1[nz]; G3(C_.fer_gamma, 2, nz, n) = t2[nz]; }
  }
}

/* fer_gamma2vel (:125-154) */
void orc_fer_gamma2vel(void) {
  const double onethird = 1. / 3.;
  for (int el = 1; el <= C_.m.myDim_elem2D; el++) {
    int n1 = EN(1, el), n2 = EN(2, el), n3 = EN(3, el);
    for (int nz = ULEV(el); nz <= NLEV(el) - 1; nz++) {
      double zinv = onethird / A2(C_.helem, nz, el);
      for (int k = 1; k <= 2; k++)
        V2(C_.fer_UV, k, nz, el) = (((G3(C_.fer_gamma, k, nz, n1) - G3(C_.fer_gamma, k, nz + 1, n1)) + (G3(C_.fer_gamma, k, nz, n2) - G3(C_.fer_gamma, k, nz + 1, n2))) +
                                    (G3(C_.fer_gamma, k, nz, n3) - G3(C_.fer_gamma, k, nz + 1, n3))) * zinv;
    }
  }
}

/* fer_Wvel of vert_vel_ale (src/oce_ale.F90:1720-1811) */
void orc_fer_wvel(void) {
  memset(C_.fer_Wvel, 0, sizeof(double) * NL * C_.N);
  for (int ed = 1; ed <= C_.m.myDim_edge2D; ed++) {
    int n1 = EDG(1, ed), n2 = EDG(2, ed), e1 = ETRI(1, ed), e2 = ETRI(2, ed);
    for (int nz = NLEV(e1) - 1; nz >= ULEV(e1); nz--) {
      double c1 = (V2(C_.fer_UV, 2, nz, e1) * ECD(1, ed) - V2(C_.fer_UV, 1, nz, e1) * ECD(2, ed)) * A2(C_.helem, nz, e1);
      A2L(C_.fer_Wvel, nz, n1) = A2L(C_.fer_Wvel, nz, n1) + c1;
      A2L(C_.fer_Wvel, nz, n2) = A2L(C_.fer_Wvel, nz, n2) - c1;
    }
    if (e2 > 0)
      for (int nz = NLEV(e2) - 1; nz >= ULEV(e2); nz--) {
        double c2 = -(V2(C_.fer_UV, 2, nz, e2) * ECD(3, ed) - V2(C_.fer_UV, 1, nz, e2) * ECD(4, ed)) * A2(C_.helem, nz, e2);
        A2L(C_.fer_Wvel, nz, n1) = A2L(C_.fer_Wvel, nz, n1) + c2;
        A2L(C_.fer_Wvel, nz, n2) = A2L(C_.fer_Wvel, nz, n2) - c2;
      }
  }
  for (int n = 1; n <= C_.m.myDim_nod2D; n++) {
    for (int nz = NLEVN(n) - 1; nz >= ULEVN(n); nz--) A2L(C_.fer_Wvel, nz, n) = A2L(C_.fer_Wvel, nz, n) + A2L(C_.fer_Wvel, nz + 1, n);
    for (int nz = ULEVN(n); nz <= NLEVN(n) - 1; nz++) A2L(C_.fer_Wvel, nz, n) = A2L(C_.fer_Wvel, nz, n) / AREA(nz, n);
  }
}

/* solve_tracers_ale :127-131 / :165-169 (whole arrays) */
void orc_bolus_add(void) {
  for (size_t i = 0; i < (size_t)2 * NLM1 * C_.E; i++) C_.UV[i] = C_.UV[i] + C_.fer_UV[i];
  for (size_t i = 0; i < (size_t)NL * C_.N; i++) { C_.Wvel_e[i] = C_.Wvel_e[i] + C_.fer_Wvel[i]; C_.Wvel[i] = C_.Wvel[i] + C_.fer_Wvel[i]; }
}
void orc_bolus_remove(void) {
  for (size_t i = 0; i < (size_t)2 * NLM1 * C_.E; i++) C_.UV[i] = C_.UV[i] - C_.fer_UV[i];
  for (size_t i = 0; i < (size_t)NL * C_.N; i++) { C_.Wvel_e[i] = C_.Wvel_e[i] - C_.fer_Wvel[i]; C_.Wvel[i] = C_.Wvel[i] - C_.fer_Wvel[i]; }
}
