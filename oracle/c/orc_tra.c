/* ORACLE (test infrastructure): tracer part of the step (AB2, gradients, FCT advection,
 * T* update, diffusion closure, PP mixing).  Reference loop/expression order throughout. */
#include "orc.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>

/* tracer_gradient_elements: src/oce_tracer_mod.F90:19-45 */
static void tracer_gradient_elements(const double *ttf) {
  for (int e = 1; e <= C_.m.myDim_elem2D; e++) {
    int n1 = EN(1, e), n2 = EN(2, e), n3 = EN(3, e);
    for (int nz = ULEV(e); nz <= NLEV(e) - 1; nz++) {
      V2(C_.tr_xy, 1, nz, e) = GS(1, e) * A2(ttf, nz, n1) + GS(2, e) * A2(ttf, nz, n2) + GS(3, e) * A2(ttf, nz, n3);
      V2(C_.tr_xy, 2, nz, e) = GS(4, e) * A2(ttf, nz, n1) + GS(5, e) * A2(ttf, nz, n2) + GS(6, e) * A2(ttf, nz, n3);
    }
  }
}

/* tracer_gradient_z: src/oce_tracer_mod.F90:124-153 */
static void tracer_gradient_z(const double *ttf) {
  for (int n = 1; n <= C_.N; n++) {
    int nzmax = NLEVN(n), nzmin = ULEVN(n);
    for (int nz = nzmin + 1; nz <= nzmax - 1; nz++) {
      double dz = 0.5 * (A2(C_.hnode_new, nz - 1, n) + A2(C_.hnode_new, nz, n));
      A2L(C_.tr_z, nz, n) = (A2(ttf, nz - 1, n) - A2(ttf, nz, n)) / dz;
    }
    A2L(C_.tr_z, nzmin, n) = 0.0;
    A2L(C_.tr_z, nzmax, n) = 0.0;
  }
}

/* fill_up_dn_grad: src/oce_muscl_adv.F90:285-447 */
static void cluster_grad(int node, int nz, double *gx, double *gy) {
  double tvol = 0.0, tx = 0.0, ty = 0.0;
  for (int k = 1; k <= C_.m.nod_in_elem2D_num[node - 1]; k++) {
    int e = NIE(k, node);
    if (NLEV(e) - 1 < nz || nz < ULEV(e)) continue;
    double ar = C_.m.elem_area[e - 1];
    tvol = tvol + ar;
    tx = tx + V2(C_.tr_xy, 1, nz, e) * ar;
    ty = ty + V2(C_.tr_xy, 2, nz, e) * ar;
  }
  *gx = tx / tvol; *gy = ty / tvol;
}
static void fill_up_dn_grad(void) {
  double *G = C_.edge_up_dn_grad;
  for (int ed = 1; ed <= C_.m.myDim_edge2D; ed++) {
    int n1 = EDG(1, ed), n2 = EDG(2, ed);
    int t1 = C_.m.edge_up_dn_tri[2 * (ed - 1)], t2 = C_.m.edge_up_dn_tri[2 * (ed - 1) + 1];
    double gx, gy;
    if (t1 != 0 && t2 != 0) {
      int u1 = C_.m.ulevels_nod2D_max[n1 - 1], u2 = C_.m.ulevels_nod2D_max[n2 - 1];
      int l1 = C_.m.nlevels_nod2D_min[n1 - 1], l2 = C_.m.nlevels_nod2D_min[n2 - 1];
      int nzmin = u1 > u2 ? u1 : u2, nzmax = l1 < l2 ? l1 : l2;
      for (int nz = ULEVN(n1); nz <= nzmin - 1; nz++) { cluster_grad(n1, nz, &gx, &gy); V4(G, 1, nz, ed) = gx; V4(G, 3, nz, ed) = gy; }
      for (int nz = ULEVN(n2); nz <= nzmin - 1; nz++) { cluster_grad(n2, nz, &gx, &gy); V4(G, 2, nz, ed) = gx; V4(G, 4, nz, ed) = gy; }
      for (int nz = nzmin; nz <= nzmax - 1; nz++) {
        V4(G, 1, nz, ed) = V2(C_.tr_xy, 1, nz, t1); V4(G, 2, nz, ed) = V2(C_.tr_xy, 1, nz, t2);
        V4(G, 3, nz, ed) = V2(C_.tr_xy, 2, nz, t1); V4(G, 4, nz, ed) = V2(C_.tr_xy, 2, nz, t2);
      }
      for (int nz = nzmax; nz <= NLEVN(n1) - 1; nz++) { cluster_grad(n1, nz, &gx, &gy); V4(G, 1, nz, ed) = gx; V4(G, 3, nz, ed) = gy; }
      for (int nz = nzmax; nz <= NLEVN(n2) - 1; nz++) { cluster_grad(n2, nz, &gx, &gy); V4(G, 2, nz, ed) = gx; V4(G, 4, nz, ed) = gy; }
    } else {
      for (int nz = ULEVN(n1); nz <= NLEVN(n1) - 1; nz++) { cluster_grad(n1, nz, &gx, &gy); V4(G, 1, nz, ed) = gx; V4(G, 3, nz, ed) = gy; }
      for (int nz = ULEVN(n2); nz <= NLEVN(n2) - 1; nz++) { cluster_grad(n2, nz, &gx, &gy); V4(G, 2, nz, ed) = gx; V4(G, 4, nz, ed) = gy; }
    }
  }
}

/* init_tracers_AB: src/oce_tracer_mod.F90:49-83 */
void orc_init_tracers_AB(int tr) {
  size_t cnt = (size_t)NLM1 * C_.N;
  memset(C_.del_ttf, 0, sizeof(double) * cnt);
  double *old = &TRO(1, 1, tr), *cur = &TR(1, 1, tr);
  double eps = C_.p.epsilon;
  for (size_t i = 0; i < cnt; i++) old[i] = -(0.5 + eps) * old[i] + (1.5 + eps) * cur[i];
  tracer_gradient_elements(old);
  tracer_gradient_z(cur);
  fill_up_dn_grad();
  tracer_gradient_elements(cur);
}

/* adv_tra_hor_upw1: src/oce_adv_tra_hor.F90:57-211 ; adv_tra_hor_mfct: :485-733.
 * mode 0: upwind with init_zero=.true. ; mode 1: MFCT with init_zero=.false. (flux = new - flux) */
static const int *muscl_nboundary_lay(void) {    /* oce_muscl_adv.F90:74-104 (owned edges, no exchange: as the reference) */
  static int *nb = NULL;
  if (nb) return nb;
  nb = malloc(sizeof(int) * C_.N);
  for (int n = 0; n < C_.N; n++) nb[n] = NL - 1;
  for (int ed = 1; ed <= C_.m.myDim_edge2D; ed++) {
    int n1 = EDG(1, ed), n2 = EDG(2, ed), e1 = ETRI(1, ed), e2 = ETRI(2, ed);
    if (e1 <= 0 || e2 <= 0) { nb[n1 - 1] = 0; nb[n2 - 1] = 0; }
    else {
      int lv = (NLEV(e1) < NLEV(e2) ? NLEV(e1) : NLEV(e2)) - 1;
      if (lv < nb[n1 - 1]) nb[n1 - 1] = lv;
      if (lv < nb[n2 - 1]) nb[n2 - 1] = lv;
    }
  }
  return nb;
}
static void adv_tra_hor(const double *ttf, int mode, double num_ord) {
  double *flux = C_.adv_flux_hor;
  const int *nb_lay = (mode == 3) ? muscl_nboundary_lay() : NULL;
  if (mode == 0) memset(flux, 0, sizeof(double) * (size_t)NLM1 * C_.m.myDim_edge2D);
  for (int ed = 1; ed <= C_.m.myDim_edge2D; ed++) {
    int n1 = EDG(1, ed), n2 = EDG(2, ed), e1 = ETRI(1, ed), e2 = ETRI(2, ed);
    int nl1 = NLEV(e1) - 1, nu1 = ULEV(e1), nl2 = 0, nu2 = 0;
    double dX1 = ECD(1, ed), dY1 = ECD(2, ed), dX2 = 0, dY2 = 0;
    double a = R_EARTH * C_.m.elem_cos[e1 - 1];
    if (e2 > 0) {
      dX2 = ECD(3, ed); dY2 = ECD(4, ed);
      nl2 = NLEV(e2) - 1; nu2 = ULEV(e2);
      a = 0.5 * (a + R_EARTH * C_.m.elem_cos[e2 - 1]);
    }
    int nl12 = nl1 < nl2 ? nl1 : nl2, nu12 = nu1 > nu2 ? nu1 : nu2;
    int lo = nu1, hi = nl1 > nl2 ? nl1 : nl2;
    if (nu2 > 0 && nu2 < lo) lo = nu2;
    for (int nz = lo; nz <= hi; nz++) {
      /* which of the five sub-ranges (A..E) this level is in */
      int use1, use2;
      if (nz >= nu12 && nz <= nl12) { use1 = 1; use2 = 1; }
      else if ((nz >= nu1 && nz <= nu12 - 1) || (nz >= nl12 + 1 && nz <= nl1)) { use1 = 1; use2 = 0; }
      else if (nu2 > 0 && ((nz >= nu2 && nz <= nu12 - 1) || (nz >= nl12 + 1 && nz <= nl2))) { use1 = 0; use2 = 1; }
      else continue;
      double vflux;
      if (use1 && use2)
        vflux = (-V2(C_.UV, 2, nz, e1) * dX1 + V2(C_.UV, 1, nz, e1) * dY1) * A2(C_.helem, nz, e1) +
                (V2(C_.UV, 2, nz, e2) * dX2 - V2(C_.UV, 1, nz, e2) * dY2) * A2(C_.helem, nz, e2);
      else if (use1) vflux = (-V2(C_.UV, 2, nz, e1) * dX1 + V2(C_.UV, 1, nz, e1) * dY1) * A2(C_.helem, nz, e1);
      else vflux = (V2(C_.UV, 2, nz, e2) * dX2 - V2(C_.UV, 1, nz, e2) * dY2) * A2(C_.helem, nz, e2);
      double t1 = A2(ttf, nz, n1), t2 = A2(ttf, nz, n2);
      if (mode == 0) {
        A2(flux, nz, ed) = -0.5 * (t1 * (vflux + fabs(vflux)) + t2 * (vflux - fabs(vflux))) - A2(flux, nz, ed);
      } else if (mode == 2) {                      /* adv_tra_hor_upw1 as the high-order scheme, init_zero=.false. */
        A2(flux, nz, ed) = -0.5 * (t1 * (vflux + fabs(vflux)) + t2 * (vflux - fabs(vflux))) - A2(flux, nz, ed);
      } else {
        const double *G = C_.edge_up_dn_grad;
        double Tmean2 = t2 - (2.0 * (t2 - t1) + EDXY(1, ed) * a * V4(G, 2, nz, ed) + EDXY(2, ed) * R_EARTH * V4(G, 4, nz, ed)) / 6.0;
        double Tmean1 = t1 + (2.0 * (t2 - t1) + EDXY(1, ed) * a * V4(G, 1, nz, ed) + EDXY(2, ed) * R_EARTH * V4(G, 3, nz, ed)) / 6.0;
        if (mode == 3) {                           /* adv_tra_hor_muscl :215-481: c_lo = real(max(sign(1, nboundary_lay - nz), 0)) */
          double c1 = (nb_lay[n1 - 1] - nz >= 0) ? 1.0 : 0.0, c2 = (nb_lay[n2 - 1] - nz >= 0) ? 1.0 : 0.0;
          Tmean2 = t2 - (2.0 * (t2 - t1) + EDXY(1, ed) * a * V4(G, 2, nz, ed) + EDXY(2, ed) * R_EARTH * V4(G, 4, nz, ed)) / 6.0 * c2;
          Tmean1 = t1 + (2.0 * (t2 - t1) + EDXY(1, ed) * a * V4(G, 1, nz, ed) + EDXY(2, ed) * R_EARTH * V4(G, 3, nz, ed)) / 6.0 * c1;
        }
        double cHO = (vflux + fabs(vflux)) * Tmean1 + (vflux - fabs(vflux)) * Tmean2;
        A2(flux, nz, ed) = -0.5 * (1.0 - num_ord) * cHO - vflux * num_ord * (0.5 * (Tmean1 + Tmean2)) - A2(flux, nz, ed);
      }
    }
  }
}

/* adv_tra_ver_upw1: src/oce_adv_tra_ver.F90:231-282 (init_zero=.true.) */
static void adv_tra_ver_upw1(const double *ttf, const double *W) {
  double *flux = C_.adv_flux_ver;
  memset(flux, 0, sizeof(double) * (size_t)NL * C_.m.myDim_nod2D);
  for (int n = 1; n <= C_.m.myDim_nod2D; n++) {
    int nzmax = NLEVN(n), nzmin = ULEVN(n);
    A2L(flux, nzmin, n) = -A2L(W, nzmin, n) * A2(ttf, nzmin, n) * AREA(nzmin, n) - A2L(flux, nzmin, n);
    A2L(flux, nzmax, n) = 0.0 - A2L(flux, nzmax, n);
    for (int nz = nzmin + 1; nz <= nzmax - 1; nz++) {
      double w = A2L(W, nz, n);
      A2L(flux, nz, n) = -0.5 * (A2(ttf, nz, n) * (w + fabs(w)) + A2(ttf, nz - 1, n) * (w - fabs(w))) * AREA(nz, n) - A2L(flux, nz, n);
    }
  }
}

/* adv_tra_ver_cdiff: src/oce_adv_tra_ver.F90:542-590 and adv_tra_ver_upw1 :231-282 as the high-order scheme (init_zero=.false.) */
static void adv_tra_ver_cdiff(const double *ttf, const double *W) {
  double *flux = C_.adv_flux_ver;
  for (int n = 1; n <= C_.m.myDim_nod2D; n++) {
    int nzmax = NLEVN(n) - 1, nzmin = ULEVN(n);
    A2L(flux, nzmin, n) = -A2L(W, nzmin, n) * A2(ttf, nzmin, n) * AREA(nzmin, n) - A2L(flux, nzmin, n);
    for (int nz = nzmin + 1; nz <= nzmax; nz++) {
      double tv = 0.5 * (A2(ttf, nz - 1, n) + A2(ttf, nz, n));
      A2L(flux, nz, n) = -tv * A2L(W, nz, n) * AREA(nz, n) - A2L(flux, nz, n);
    }
  }
}
static void adv_tra_ver_upw1_ho(const double *ttf, const double *W) {
  double *flux = C_.adv_flux_ver;
  for (int n = 1; n <= C_.m.myDim_nod2D; n++) {
    int nzmax = NLEVN(n), nzmin = ULEVN(n);
    A2L(flux, nzmin, n) = -A2L(W, nzmin, n) * A2(ttf, nzmin, n) * AREA(nzmin, n) - A2L(flux, nzmin, n);
    A2L(flux, nzmax, n) = 0.0 - A2L(flux, nzmax, n);
    for (int nz = nzmin + 1; nz <= nzmax - 1; nz++) {
      double w = A2L(W, nz, n);
      A2L(flux, nz, n) = -0.5 * (A2(ttf, nz, n) * (w + fabs(w)) + A2(ttf, nz - 1, n) * (w - fabs(w))) * AREA(nz, n) - A2L(flux, nz, n);
    }
  }
}

/* adv_tra_vert_ppm: src/oce_adv_tra_ver.F90:361-538 (Colella & Woodward 1984), init_zero=.false. */
static void adv_tra_vert_ppm(const double *ttf, const double *W) {
  double *flux = C_.adv_flux_ver;
  const double dt = C_.p.dt;
  double *tv = calloc((size_t)2 * (NL + 3), sizeof(double)), *tvert = tv + NL + 3;
  for (int n = 1; n <= C_.m.myDim_nod2D; n++) {
    int nzmax = NLEVN(n), nzmin = ULEVN(n);
#define T_(k) A2(ttf, k, n)
    tv[nzmin] = T_(nzmin);
    tv[nzmin + 1] = 0.5 * (T_(nzmin) + T_(nzmin + 1));
    {
      double sg = copysign(1.0, A2L(W, nzmax - 1, n));
      tv[nzmax - 1] = -T_(nzmax - 2) * (sg < 0. ? sg : 0.) + T_(nzmax - 1) * (sg > 0. ? sg : 0.);
    }
    tv[nzmax] = T_(nzmax - 1);
    for (int nz = nzmin + 1; nz <= nzmax - 3; nz++) {
      double dzjm1 = A2(C_.hnode_new, nz - 1, n), dzj = A2(C_.hnode_new, nz, n), dzjp1 = A2(C_.hnode_new, nz + 1, n), dzjp2 = A2(C_.hnode_new, nz + 2, n);
      double d0 = T_(nz + 1) - T_(nz), dm = T_(nz) - T_(nz - 1), dp = T_(nz + 2) - T_(nz + 1);
      double deltaj = dzj / (dzjm1 + dzj + dzjp1) * ((2. * dzjm1 + dzj) / (dzjp1 + dzj) * d0 + (dzj + 2. * dzjp1) / (dzjm1 + dzj) * dm);
      double deltajp1 = dzjp1 / (dzj + dzjp1 + dzjp2) * ((2. * dzj + dzjp1) / (dzjp2 + dzjp1) * dp + (dzjp1 + 2. * dzjp2) / (dzj + dzjp1) * d0);
      if (d0 * dm > 0.) deltaj = fmin(fmin(fabs(deltaj), 2. * fabs(d0)), 2. * fabs(dm)) * copysign(1.0, deltaj);
      else deltaj = 0.0;
      if (dp * d0 > 0.) deltajp1 = fmin(fmin(fabs(deltajp1), 2. * fabs(dp)), 2. * fabs(d0)) * copysign(1.0, deltajp1);
      else deltajp1 = 0.0;
      tv[nz + 1] = T_(nz) + dzj / (dzj + dzjp1) * d0 +
                   1. / (dzjm1 + dzj + dzjp1 + dzjp2) *
                       ((2. * dzjp1 * dzj) / (dzj + dzjp1) * ((dzjm1 + dzj) / (2. * dzj + dzjp1) - (dzjp2 + dzjp1) / (2. * dzjp1 + dzj)) * d0 -
                        dzj * (dzjm1 + dzj) / (2. * dzj + dzjp1) * deltajp1 + dzjp1 * (dzjp1 + dzjp2) / (dzj + 2. * dzjp1) * deltaj);
    }
    for (int k = 1; k <= nzmax; k++) tvert[k] = 0.;
    for (int nz = nzmin; nz <= nzmax - 1; nz++) {
      double w0 = A2L(W, nz, n), w1 = A2L(W, nz + 1, n);
      if (w0 <= 0. && w1 >= 0.) continue;
      double aL = tv[nz], aR = tv[nz + 1], t = T_(nz);
      if ((aR - t) * (t - aL) <= 0.) { aL = t; aR = t; }
      if ((aR - aL) * (t - 0.5 * (aL + aR)) > (aR - aL) * (aR - aL) / 6.) aL = 3. * t - 2. * aR;
      if ((aR - aL) * (t - 0.5 * (aR + aL)) < -((aR - aL) * (aR - aL)) / 6.) aR = 3. * t - 2. * aL;
      double dzj = A2(C_.hnode, nz, n);
      double aj = 6.0 * (t - 0.5 * (aL + aR));
      if (w0 > 0.) {
        double x = fmin(w0 * dt / dzj, 1.);
        tvert[nz] = (-aL - 0.5 * x * (aR - aL + (1. - 2. / 3. * x) * aj));
        tvert[nz] = tvert[nz] * AREA(nz, n) * w0;
      }
      if (w1 < 0.) {
        double x = fmin(-w1 * dt / dzj, 1.);
        tvert[nz + 1] = (-aR + 0.5 * x * (aR - aL - (1. - 2. / 3. * x) * aj));
        tvert[nz + 1] = tvert[nz + 1] * AREA(nz + 1, n) * w1;
      }
    }
    tvert[nzmin] = -tv[nzmin] * A2L(W, nzmin, n) * AREA(nzmin, n);
    tvert[nzmax] = 0.0;
    for (int nz = nzmin; nz <= nzmax; nz++) A2L(flux, nz, n) = tvert[nz] - A2L(flux, nz, n);
#undef T_
  }
  free(tv);
}

/* adv_tra_ver_qr4c: src/oce_adv_tra_ver.F90:286-357 (init_zero=.false.) */
static void adv_tra_ver_qr4c(const double *ttf, const double *W, double num_ord) {
  double *flux = C_.adv_flux_ver;
  for (int n = 1; n <= C_.m.myDim_nod2D; n++) {
    int nzmax = NLEVN(n), nzmin = ULEVN(n), nz;
    nz = nzmin;
    A2L(flux, nz, n) = -A2(ttf, nz, n) * A2L(W, nz, n) * AREA(nz, n) - A2L(flux, nz, n);
    nz = nzmin + 1;
    A2L(flux, nz, n) = -0.5 * (A2(ttf, nz - 1, n) + A2(ttf, nz, n)) * A2L(W, nz, n) * AREA(nz, n) - A2L(flux, nz, n);
    nz = nzmax - 1;
    A2L(flux, nz, n) = -0.5 * (A2(ttf, nz - 1, n) + A2(ttf, nz, n)) * A2L(W, nz, n) * AREA(nz, n) - A2L(flux, nz, n);
    nz = nzmax;
    A2L(flux, nz, n) = 0.0 - A2L(flux, nz, n);
    for (nz = nzmin + 2; nz <= nzmax - 2; nz++) {
      double qc = (A2(ttf, nz - 1, n) - A2(ttf, nz, n)) / (A2(C_.Z_3d_n, nz - 1, n) - A2(C_.Z_3d_n, nz, n));
      double qu = (A2(ttf, nz, n) - A2(ttf, nz + 1, n)) / (A2(C_.Z_3d_n, nz, n) - A2(C_.Z_3d_n, nz + 1, n));
      double qd = (A2(ttf, nz - 2, n) - A2(ttf, nz - 1, n)) / (A2(C_.Z_3d_n, nz - 2, n) - A2(C_.Z_3d_n, nz - 1, n));
      double Tmean1 = A2(ttf, nz, n) + (2 * qc + qu) * (A2L(C_.zbar_3d_n, nz, n) - A2(C_.Z_3d_n, nz, n)) / 3.0;
      double Tmean2 = A2(ttf, nz - 1, n) + (2 * qc + qd) * (A2L(C_.zbar_3d_n, nz, n) - A2(C_.Z_3d_n, nz - 1, n)) / 3.0;
      double w = A2L(W, nz, n);
      double Tmean = (w + fabs(w)) * Tmean1 + (w - fabs(w)) * Tmean2;
      A2L(flux, nz, n) = (-0.5 * (1.0 - num_ord) * Tmean - num_ord * (0.5 * (Tmean1 + Tmean2)) * w) * AREA(nz, n) - A2L(flux, nz, n);
    }
  }
}

static void edge_range(int ed, int *nu12, int *nl12) {
  int e1 = ETRI(1, ed), e2 = ETRI(2, ed);
  int nl1 = NLEV(e1) - 1, nu1 = ULEV(e1), nl2 = 0, nu2 = 0;
  if (e2 > 0) { nl2 = NLEV(e2) - 1; nu2 = ULEV(e2); }
  *nl12 = nl1 > nl2 ? nl1 : nl2;
  *nu12 = nu1;
  if (nu2 > 0) *nu12 = nu1 < nu2 ? nu1 : nu2;
}

/* oce_tra_adv_fct: src/oce_adv_tra_fct.F90:58-349 (vlimit=1).  The reference uses UV_rhs as
 * scratch for the element bounds (:108-121); the oracle uses fct_ebnd instead. */
static void oce_tra_adv_fct(const double *ttf) {
  const double flux_eps = 1e-16, bignumber = 1e3;
  double *LO = C_.fct_LO, *adf_h = C_.adv_flux_hor, *adf_v = C_.adv_flux_ver, *EB = C_.fct_ebnd;
  double dt = C_.p.dt;
  int nl = NL;
  for (int n = 1; n <= C_.N; n++)
    for (int nz = ULEVN(n); nz <= NLEVN(n) - 1; nz++) {
      A2(C_.fct_ttf_max, nz, n) = dmax(A2(LO, nz, n), A2(ttf, nz, n));
      A2(C_.fct_ttf_min, nz, n) = dmin(A2(LO, nz, n), A2(ttf, nz, n));
    }
  for (int e = 1; e <= C_.m.myDim_elem2D; e++) {
    int n1 = EN(1, e), n2 = EN(2, e), n3 = EN(3, e), nl1 = NLEV(e);
    for (int nz = ULEV(e); nz <= nl1 - 1; nz++) {
      V2(EB, 1, nz, e) = dmax(dmax(A2(C_.fct_ttf_max, nz, n1), A2(C_.fct_ttf_max, nz, n2)), A2(C_.fct_ttf_max, nz, n3));
      V2(EB, 2, nz, e) = dmin(dmin(A2(C_.fct_ttf_min, nz, n1), A2(C_.fct_ttf_min, nz, n2)), A2(C_.fct_ttf_min, nz, n3));
    }
    if (nl1 <= nl - 1)
      for (int nz = nl1; nz <= nl - 1; nz++) { V2(EB, 1, nz, e) = -bignumber; V2(EB, 2, nz, e) = bignumber; }
  }
  double *tvmax = malloc(sizeof(double) * 2 * (nl + 1)), *tvmin = tvmax + nl + 1;
  for (int n = 1; n <= C_.m.myDim_nod2D; n++) {
    int nu1 = ULEVN(n), nl1 = NLEVN(n);
    for (int nz = nu1; nz <= nl1 - 1; nz++) {
      double mx = V2(EB, 1, nz, NIE(1, n)), mn = V2(EB, 2, nz, NIE(1, n));
      for (int k = 2; k <= C_.m.nod_in_elem2D_num[n - 1]; k++) { mx = dmax(mx, V2(EB, 1, nz, NIE(k, n))); mn = dmin(mn, V2(EB, 2, nz, NIE(k, n))); }
      tvmax[nz] = mx; tvmin[nz] = mn;
    }
    A2(C_.fct_ttf_max, nu1, n) = tvmax[nu1] - A2(LO, nu1, n);
    A2(C_.fct_ttf_min, nu1, n) = tvmin[nu1] - A2(LO, nu1, n);
    for (int nz = nu1 + 1; nz <= nl1 - 2; nz++) {
      A2(C_.fct_ttf_max, nz, n) = dmax(dmax(tvmax[nz - 1], tvmax[nz]), tvmax[nz + 1]) - A2(LO, nz, n);
      A2(C_.fct_ttf_min, nz, n) = dmin(dmin(tvmin[nz - 1], tvmin[nz]), tvmin[nz + 1]) - A2(LO, nz, n);
    }
    int nz = nl1 - 1;
    A2(C_.fct_ttf_max, nz, n) = tvmax[nz] - A2(LO, nz, n);
    A2(C_.fct_ttf_min, nz, n) = tvmin[nz] - A2(LO, nz, n);
  }
  free(tvmax);
  for (int n = 1; n <= C_.m.myDim_nod2D; n++)
    for (int nz = ULEVN(n); nz <= NLEVN(n) - 1; nz++) {
      A2(C_.fct_plus, nz, n) = 0.0; A2(C_.fct_minus, nz, n) = 0.0;
      A2(C_.fct_plus, nz, n) = A2(C_.fct_plus, nz, n) + (dmax(0.0, A2L(adf_v, nz, n)) + dmax(0.0, -A2L(adf_v, nz + 1, n)));
      A2(C_.fct_minus, nz, n) = A2(C_.fct_minus, nz, n) + (dmin(0.0, A2L(adf_v, nz, n)) + dmin(0.0, -A2L(adf_v, nz + 1, n)));
    }
  for (int ed = 1; ed <= C_.m.myDim_edge2D; ed++) {
    int n1 = EDG(1, ed), n2 = EDG(2, ed), nu12, nl12;
    edge_range(ed, &nu12, &nl12);
    for (int nz = nu12; nz <= nl12; nz++) {
      double f = A2(adf_h, nz, ed);
      A2(C_.fct_plus, nz, n1) = A2(C_.fct_plus, nz, n1) + dmax(0.0, f);
      A2(C_.fct_minus, nz, n1) = A2(C_.fct_minus, nz, n1) + dmin(0.0, f);
      A2(C_.fct_plus, nz, n2) = A2(C_.fct_plus, nz, n2) + dmax(0.0, -f);
      A2(C_.fct_minus, nz, n2) = A2(C_.fct_minus, nz, n2) + dmin(0.0, -f);
    }
  }
  for (int n = 1; n <= C_.m.myDim_nod2D; n++)
    for (int nz = ULEVN(n); nz <= NLEVN(n) - 1; nz++) {
      double flux = A2(C_.fct_plus, nz, n) * dt / AREASVOL(nz, n) + flux_eps;
      A2(C_.fct_plus, nz, n) = dmin(1.0, A2(C_.fct_ttf_max, nz, n) / flux);
      flux = A2(C_.fct_minus, nz, n) * dt / AREASVOL(nz, n) - flux_eps;
      A2(C_.fct_minus, nz, n) = dmin(1.0, A2(C_.fct_ttf_min, nz, n) / flux);
    }
  for (int n = 1; n <= C_.m.myDim_nod2D; n++) {
    int nu1 = ULEVN(n), nl1 = NLEVN(n);
    int nz = nu1;
    double ae = 1.0, flux = A2L(adf_v, nz, n);
    if (flux >= 0.0) ae = dmin(ae, A2(C_.fct_plus, nz, n)); else ae = dmin(ae, A2(C_.fct_minus, nz, n));
    A2L(adf_v, nz, n) = ae * A2L(adf_v, nz, n);
    for (nz = nu1 + 1; nz <= nl1 - 1; nz++) {
      ae = 1.0; flux = A2L(adf_v, nz, n);
      if (flux >= 0.) { ae = dmin(ae, A2(C_.fct_minus, nz - 1, n)); ae = dmin(ae, A2(C_.fct_plus, nz, n)); }
      else { ae = dmin(ae, A2(C_.fct_plus, nz - 1, n)); ae = dmin(ae, A2(C_.fct_minus, nz, n)); }
      A2L(adf_v, nz, n) = ae * A2L(adf_v, nz, n);
    }
  }
  for (int ed = 1; ed <= C_.m.myDim_edge2D; ed++) {
    int n1 = EDG(1, ed), n2 = EDG(2, ed), nu12, nl12;
    edge_range(ed, &nu12, &nl12);
    for (int nz = nu12; nz <= nl12; nz++) {
      double ae = 1.0, flux = A2(adf_h, nz, ed);
      if (flux >= 0.) { ae = dmin(ae, A2(C_.fct_plus, nz, n1)); ae = dmin(ae, A2(C_.fct_minus, nz, n2)); }
      else { ae = dmin(ae, A2(C_.fct_minus, nz, n1)); ae = dmin(ae, A2(C_.fct_plus, nz, n2)); }
      A2(adf_h, nz, ed) = ae * A2(adf_h, nz, ed);
    }
  }
}

/* adv_tra_vert_impl: src/oce_adv_tra_ver.F90:83-227 -- implicit part of the vertical advection (w_split), applied to the
 * low-order solution (oce_adv_tra_driver.F90:124-127) */
static void adv_tra_vert_impl(double *ttf, const double *W) {
  int nl = NL;
  double *a = malloc(sizeof(double) * (nl + 2) * 6), *b = a + nl + 2, *c = b + nl + 2, *tr = c + nl + 2, *cp = tr + nl + 2, *tp = cp + nl + 2;
  const double dt = C_.p.dt;
  for (int n = 1; n <= C_.m.myDim_nod2D; n++) {
    int nzmax = NLEVN(n), nzmin = ULEVN(n), nz;
    for (int k = 0; k < (nl + 2) * 6; k++) a[k] = 0.0;
    double zinv = 1.0 * dt, v_adv;
    nz = nzmin;
    a[nz] = 0.0;
    v_adv = zinv * AREA(nz, n) / AREASVOL(nz, n);
    b[nz] = A2(C_.hnode_new, nz, n) + A2L(W, nz, n) * v_adv;
    v_adv = zinv * AREA(nz + 1, n) / AREASVOL(nz, n);
    b[nz] = b[nz] - dmin(0., A2L(W, nz + 1, n)) * v_adv;
    c[nz] = -dmax(0., A2L(W, nz + 1, n)) * v_adv;
    for (nz = nzmin + 1; nz <= nzmax - 2; nz++) {
      v_adv = zinv * AREA(nz, n) / AREASVOL(nz, n);
      a[nz] = dmin(0., A2L(W, nz, n)) * v_adv;
      b[nz] = A2(C_.hnode_new, nz, n) + dmax(0., A2L(W, nz, n)) * v_adv;
      v_adv = zinv * AREA(nz + 1, n) / AREASVOL(nz, n);
      b[nz] = b[nz] - dmin(0., A2L(W, nz + 1, n)) * v_adv;
      c[nz] = -dmax(0., A2L(W, nz + 1, n)) * v_adv;
    }
    nz = nzmax - 1;
    v_adv = zinv * AREA(nz, n) / AREASVOL(nz, n);
    a[nz] = dmin(0., A2L(W, nz, n)) * v_adv;
    b[nz] = A2(C_.hnode_new, nz, n) + dmax(0., A2L(W, nz, n)) * v_adv;
    c[nz] = 0.0;
    nz = nzmin;
    double dz = A2(C_.hnode_new, nz, n);
    tr[nz] = -(b[nz] - dz) * A2(ttf, nz, n) - c[nz] * A2(ttf, nz + 1, n);
    for (nz = nzmin + 1; nz <= nzmax - 2; nz++) {
      dz = A2(C_.hnode_new, nz, n);
      tr[nz] = -a[nz] * A2(ttf, nz - 1, n) - (b[nz] - dz) * A2(ttf, nz, n) - c[nz] * A2(ttf, nz + 1, n);
    }
    nz = nzmax - 1;
    dz = A2(C_.hnode_new, nz, n);
    tr[nz] = -a[nz] * A2(ttf, nz - 1, n) - (b[nz] - dz) * A2(ttf, nz, n);
    nz = nzmin;
    cp[nz] = c[nz] / b[nz];
    tp[nz] = tr[nz] / b[nz];
    for (nz = nzmin + 1; nz <= nzmax - 1; nz++) {
      double m = b[nz] - cp[nz - 1] * a[nz];
      cp[nz] = c[nz] / m;
      tp[nz] = (tr[nz] - tp[nz - 1] * a[nz]) / m;
    }
    tr[nzmax - 1] = tp[nzmax - 1];
    for (nz = nzmax - 2; nz >= nzmin; nz--) tr[nz] = tp[nz] - cp[nz] * tr[nz + 1];
    for (nz = nzmin; nz <= nzmax - 1; nz++) A2(ttf, nz, n) = A2(ttf, nz, n) + tr[nz];
  }
  free(a);
}

/* do_oce_adv_tra (FCT + MFCT + QR4C): src/oce_adv_tra_driver.F90:41-197 ;
 * oce_tra_adv_flux2dtracer: :201-269 ; adv_tracers_ale: src/oce_ale_tracer.F90:203-249 */
void orc_adv_tracers_ale(int tr) {
  size_t cnt = (size_t)NLM1 * C_.N;
  double dt = C_.p.dt;
  const double *ttf = &TR(1, 1, tr), *ttfAB = &TRO(1, 1, tr);
  double *dh = C_.del_ttf_advhoriz, *dv = C_.del_ttf_advvert, *LO = C_.fct_LO;
  memset(dh, 0, sizeof(double) * cnt);
  memset(dv, 0, sizeof(double) * cnt);
  const int fct = C_.p.tra_adv_lim == 0;           /* tra_adv_lim = 'FCT' | 'NON' (oce_adv_tra_driver.F90:79,137,155,189) */
  const double *Who = fct ? C_.Wvel : C_.Wvel_e;   /* pwvel => w with FCT, => we without (:155-159) */
  if (fct) {
  adv_tra_hor(ttf, 0, 0.0);
  memset(LO, 0, sizeof(double) * cnt);
  for (int ed = 1; ed <= C_.m.myDim_edge2D; ed++) {
    int n1 = EDG(1, ed), n2 = EDG(2, ed), nu12, nl12;
    edge_range(ed, &nu12, &nl12);
    for (int nz = nu12; nz <= nl12; nz++) {
      A2(LO, nz, n1) = A2(LO, nz, n1) + A2(C_.adv_flux_hor, nz, ed);
      A2(LO, nz, n2) = A2(LO, nz, n2) - A2(C_.adv_flux_hor, nz, ed);
    }
  }
  adv_tra_ver_upw1(ttf, C_.Wvel_e);
  for (int n = 1; n <= C_.m.myDim_nod2D; n++)
    for (int nz = ULEVN(n); nz <= NLEVN(n) - 1; nz++)
      A2(LO, nz, n) = (A2(ttf, nz, n) * A2(C_.hnode, nz, n) +
                       (A2(LO, nz, n) + (A2L(C_.adv_flux_ver, nz, n) - A2L(C_.adv_flux_ver, nz + 1, n))) * dt / AREASVOL(nz, n)) /
                      A2(C_.hnode_new, nz, n);
  if (C_.p.w_split) {                      /* oce_adv_tra_driver.F90:124-132 */
    adv_tra_vert_impl(LO, C_.Wvel_i);
    adv_tra_ver_upw1(ttf, C_.Wvel);        /* low-order part of the anti-diffusive vertical fluxes: on the full w */
  }
  } else {                                 /* do_zero_flux = .true.: the high-order routines zero their flux arrays first */
    memset(C_.adv_flux_hor, 0, sizeof(double) * (size_t)NLM1 * C_.m.myDim_edge2D);
    memset(C_.adv_flux_ver, 0, sizeof(double) * (size_t)NL * C_.m.myDim_nod2D);
  }
  adv_tra_hor(ttfAB, C_.p.tra_adv_hor == 1 ? 3 : C_.p.tra_adv_hor == 2 ? 2 : 1, C_.p.tra_adv_ph);
  if (C_.p.tra_adv_ver == 1) adv_tra_ver_cdiff(ttfAB, Who);
  else if (C_.p.tra_adv_ver == 2) adv_tra_ver_upw1_ho(ttfAB, Who);
  else if (C_.p.tra_adv_ver == 3) adv_tra_vert_ppm(ttfAB, Who);
  else adv_tra_ver_qr4c(ttfAB, Who, C_.p.tra_adv_pv);
  if (fct) {
    oce_tra_adv_fct(ttf);
    /* flux2dtracer with use_lo */
    for (int n = 1; n <= C_.m.myDim_nod2D; n++)
      for (int nz = ULEVN(n); nz <= NLEVN(n) - 1; nz++)
        A2(dv, nz, n) = A2(dv, nz, n) - A2(ttf, nz, n) * A2(C_.hnode, nz, n) + A2(LO, nz, n) * A2(C_.hnode_new, nz, n);
  }
  for (int n = 1; n <= C_.m.myDim_nod2D; n++)
    for (int nz = ULEVN(n); nz <= NLEVN(n) - 1; nz++)
      A2(dv, nz, n) = A2(dv, nz, n) + (A2L(C_.adv_flux_ver, nz, n) - A2L(C_.adv_flux_ver, nz + 1, n)) * dt / AREASVOL(nz, n);
  for (int ed = 1; ed <= C_.m.myDim_edge2D; ed++) {
    int n1 = EDG(1, ed), n2 = EDG(2, ed), nu12, nl12;
    edge_range(ed, &nu12, &nl12);
    for (int nz = nu12; nz <= nl12; nz++) {
      A2(dh, nz, n1) = A2(dh, nz, n1) + A2(C_.adv_flux_hor, nz, ed) * dt / AREASVOL(nz, n1);
      A2(dh, nz, n2) = A2(dh, nz, n2) - A2(C_.adv_flux_hor, nz, ed) * dt / AREASVOL(nz, n2);
    }
  }
  for (size_t i = 0; i < cnt; i++) C_.del_ttf[i] = C_.del_ttf[i] + dh[i] + dv[i];
}

/* diff_part_hor_redi with Redi=.false.: src/oce_ale_tracer.F90:929-1077 */
static void diff_part_hor(void) {
  double dt = C_.p.dt;
  for (int ed = 1; ed <= C_.m.myDim_edge2D; ed++) {
    int n1 = EDG(1, ed), n2 = EDG(2, ed), e1 = ETRI(1, ed), e2 = ETRI(2, ed);
    double dX1 = ECD(1, ed), dY1 = ECD(2, ed), dX2 = 0, dY2 = 0;
    int nl1 = NLEV(e1) - 1, ul1 = ULEV(e1), nl2 = 0, ul2 = 0;
    if (e2 > 0) { nl2 = NLEV(e2) - 1; ul2 = ULEV(e2); dX2 = ECD(3, ed); dY2 = ECD(4, ed); }
    int nl12 = nl1 < nl2 ? nl1 : nl2, ul12 = ul1 > ul2 ? ul1 : ul2;
    int hi = nl1 > nl2 ? nl1 : nl2, lo = ul1;
    if (ul2 > 0) lo = ul1 < ul2 ? ul1 : ul2;
    const double isredi = C_.p.Redi ? 1.0 : 0.0;
    for (int nz = lo; nz <= hi; nz++) {
      double Kh = (A2(C_.Ki, nz, n1) + A2(C_.Ki, nz, n2)) / 2.0, c;
      /* Redi: slope * vertical gradient at the two edge nodes (:990-993); with Redi off the reference still forms S*Tz*0 */
      double sxtz = 0.0, sytz = 0.0;
      if (C_.p.Redi) {
        double Tz1 = 0.5 * (A2L(C_.tr_z, nz, n1) + A2L(C_.tr_z, nz + 1, n1)), Tz2 = 0.5 * (A2L(C_.tr_z, nz, n2) + A2L(C_.tr_z, nz + 1, n2));
        sxtz = (Tz1 * V3(C_.slope_tapered, 1, nz, n1) + Tz2 * V3(C_.slope_tapered, 1, nz, n2)) / 2.0;
        sytz = (Tz1 * V3(C_.slope_tapered, 2, nz, n1) + Tz2 * V3(C_.slope_tapered, 2, nz, n2)) / 2.0;
      }
      const double ax = sxtz * isredi, ay = sytz * isredi;
      if (nz >= ul12 && nz <= nl12) {
        double dz = (A2(C_.helem, nz, e1) + A2(C_.helem, nz, e2)) / 2.0;
        double Tx = 0.5 * (V2(C_.tr_xy, 1, nz, e1) + V2(C_.tr_xy, 1, nz, e2));
        double Ty = 0.5 * (V2(C_.tr_xy, 2, nz, e1) + V2(C_.tr_xy, 2, nz, e2));
        double Fx = Kh * (Tx + ax), Fy = Kh * (Ty + ay);
        c = ((dX2 - dX1) * Fy - (dY2 - dY1) * Fx) * dz;
      } else if ((nz >= ul1 && nz <= ul12 - 1) || (nz >= nl12 + 1 && nz <= nl1)) {
        double dz = A2(C_.helem, nz, e1);
        double Fx = Kh * (V2(C_.tr_xy, 1, nz, e1) + ax), Fy = Kh * (V2(C_.tr_xy, 2, nz, e1) + ay);
        c = (-dX1 * Fy + dY1 * Fx) * dz;
      } else {
        double dz = A2(C_.helem, nz, e2);
        double Fx = Kh * (V2(C_.tr_xy, 1, nz, e2) + ax), Fy = Kh * (V2(C_.tr_xy, 2, nz, e2) + ay);
        c = (dX2 * Fy - dY2 * Fx) * dz;
      }
      double rhs1 = 0.0 + c, rhs2 = 0.0 - c;
      A2(C_.del_ttf, nz, n1) = A2(C_.del_ttf, nz, n1) + rhs1 * dt / AREASVOL(nz, n1);
      A2(C_.del_ttf, nz, n2) = A2(C_.del_ttf, nz, n2) + rhs2 * dt / AREASVOL(nz, n2);
    }
  }
}

/* diff_ver_part_redi_expl: src/oce_ale_tracer.F90:860-927 */
static void diff_ver_part_redi_expl(void) {
  int nl = NL;
  double dt = C_.p.dt;
  double *txn = calloc((size_t)2 * NLM1 * C_.N, sizeof(double));
  double *buf = calloc((size_t)3 * (nl + 2), sizeof(double)), *zbar_n = buf, *Z_n = buf + nl + 2, *vd = Z_n + nl + 2;
  for (int n = 1; n <= C_.m.myDim_nod2D; n++)
    for (int nz = ULEVN(n); nz <= NLEVN(n) - 1; nz++) {
      double Tx = 0.0, Ty = 0.0;
      for (int k = 1; k <= C_.m.nod_in_elem2D_num[n - 1]; k++) {
        int el = NIE(k, n);
        if (nz <= NLEV(el) - 1 && nz >= ULEV(el)) { Tx = Tx + V2(C_.tr_xy, 1, nz, el) * C_.m.elem_area[el - 1]; Ty = Ty + V2(C_.tr_xy, 2, nz, el) * C_.m.elem_area[el - 1]; }
      }
      V2(txn, 1, nz, n) = Tx / 3.0 / AREASVOL(nz, n);
      V2(txn, 2, nz, n) = Ty / 3.0 / AREASVOL(nz, n);
    }
  for (int n = 1; n <= C_.m.myDim_nod2D; n++) {
    int nl1 = NLEVN(n) - 1, ul1 = ULEVN(n);
    for (int k = 0; k < nl + 2; k++) { vd[k] = 0.; zbar_n[k] = 0.; Z_n[k] = 0.; }
    zbar_n[nl1 + 1] = C_.m.zbar_n_bot[n - 1];
    Z_n[nl1] = zbar_n[nl1 + 1] + A2(C_.hnode_new, nl1, n) / 2.0;
    for (int nz = nl1; nz >= ul1 + 1; nz--) {
      zbar_n[nz] = zbar_n[nz + 1] + A2(C_.hnode_new, nz, n);
      Z_n[nz - 1] = zbar_n[nz] + A2(C_.hnode_new, nz - 1, n) / 2.0;
    }
    zbar_n[ul1] = zbar_n[ul1 + 1] + A2(C_.hnode_new, ul1, n);
    for (int nz = ul1 + 1; nz <= nl1; nz++) {
      vd[nz] = (Z_n[nz - 1] - zbar_n[nz]) * (V3(C_.slope_tapered, 1, nz - 1, n) * V2(txn, 1, nz - 1, n) + V3(C_.slope_tapered, 2, nz - 1, n) * V2(txn, 2, nz - 1, n)) * A2(C_.Ki, nz - 1, n);
      vd[nz] = vd[nz] + (zbar_n[nz] - Z_n[nz]) * (V3(C_.slope_tapered, 1, nz, n) * V2(txn, 1, nz, n) + V3(C_.slope_tapered, 2, nz, n) * V2(txn, 2, nz, n)) * A2(C_.Ki, nz, n);
      vd[nz] = vd[nz] / (Z_n[nz - 1] - Z_n[nz]) * AREA(nz, n);
    }
    for (int nz = ul1; nz <= nl1; nz++) A2(C_.del_ttf, nz, n) = A2(C_.del_ttf, nz, n) + (vd[nz] - vd[nz + 1]) * dt / AREASVOL(nz, n);
  }
  free(txn); free(buf);
}

/* bc_surface: src/oce_ale_tracer.F90:1154-1195 */
static double bc_surface(int n, int id) {
  double dt = C_.p.dt, nonlin = (C_.p.which_ale == 0) ? 0.0 : 1.0;
  if (id == 0) return -dt * (C_.heat_flux[n - 1] / VCPW + TR(ULEVN(n), n, 1) * C_.water_flux[n - 1] * nonlin);
  if (id == 1) return dt * (C_.virtual_salt[n - 1] + C_.relax_salt[n - 1] - C_.real_salt_flux[n - 1] * nonlin);
  return 0.0;
}

/* diff_ver_part_impl_ale (Redi optional), no w_split, no KPP non-local, no SW penetration:
 * src/oce_ale_tracer.F90:398-856 */
static void diff_ver_part_impl_ale(int tr) {
  int nl = NL;
  double *buf = calloc((size_t)8 * (nl + 2), sizeof(double));
  double *a = buf, *b = a + nl + 2, *c = b + nl + 2, *trv = c + nl + 2, *cp = trv + nl + 2, *tp = cp + nl + 2, *zbar_n = tp + nl + 2,
         *Z_n = zbar_n + nl + 2;
  double dt = C_.p.dt;
  for (int n = 1; n <= C_.m.myDim_nod2D; n++) {
    int nzmax = NLEVN(n), nzmin = ULEVN(n), nz;
    zbar_n[nzmax] = C_.m.zbar_n_bot[n - 1];
    Z_n[nzmax - 1] = zbar_n[nzmax] + A2(C_.hnode_new, nzmax - 1, n) / 2.0;
    for (nz = nzmax - 1; nz >= nzmin + 1; nz--) {
      zbar_n[nz] = zbar_n[nz + 1] + A2(C_.hnode_new, nz, n);
      Z_n[nz - 1] = zbar_n[nz] + A2(C_.hnode_new, nz - 1, n) / 2.0;
    }
    zbar_n[nzmin] = zbar_n[nzmin + 1] + A2(C_.hnode_new, nzmin, n);
    nz = nzmin;
    double zinv2 = 1.0 / (Z_n[nz] - Z_n[nz + 1]), zinv = 1.0 * dt, zinv1;
    const double isredi = C_.p.Redi ? 1.0 : 0.0;
#define K33(k) * (V3(C_.slope_tapered, 3, k, n) * V3(C_.slope_tapered, 3, k, n)) * A2(C_.Ki, k, n)    /* a*zinv*S**2*Ki, left to right */
    double Ty = 0.0, Ty1 = 0.0;
    if (C_.p.Redi) Ty1 = (Z_n[nz] - zbar_n[nz + 1]) * zinv2 K33(nz) + (zbar_n[nz + 1] - Z_n[nz + 1]) * zinv2 K33(nz + 1);
    Ty1 = Ty1 * isredi;
    a[nz] = 0.0;
    c[nz] = -(A2L(C_.Kv, nz + 1, n) + Ty1) * zinv2 * zinv * AREA(nz + 1, n) / AREASVOL(nz, n);
    b[nz] = -c[nz] + A2(C_.hnode_new, nz, n);
    const int do_wimpl = C_.p.w_split && C_.p.tra_adv_lim != 0;      /* :424: the implicit part of the vertical velocity in the solve, only without the FCT low-order solution */
    if (do_wimpl) {                                                    /* :560-572, upwind */
      double v_adv = zinv * (AREA(nz, n) / AREASVOL(nz, n));
      b[nz] = b[nz] + A2L(C_.Wvel_i, nz, n) * v_adv;
      v_adv = zinv * AREA(nz + 1, n) / AREASVOL(nz, n);
      b[nz] = b[nz] - dmin(0.0, A2L(C_.Wvel_i, nz + 1, n)) * v_adv;
      c[nz] = c[nz] - dmax(0.0, A2L(C_.Wvel_i, nz + 1, n)) * v_adv;
    }
    zinv1 = zinv2;
    for (nz = nzmin + 1; nz <= nzmax - 2; nz++) {
      zinv2 = 1.0 / (Z_n[nz] - Z_n[nz + 1]);
      if (C_.p.Redi) {
        Ty = (Z_n[nz - 1] - zbar_n[nz]) * zinv1 K33(nz - 1) + (zbar_n[nz] - Z_n[nz]) * zinv1 K33(nz);
        Ty1 = (Z_n[nz] - zbar_n[nz + 1]) * zinv2 K33(nz) + (zbar_n[nz + 1] - Z_n[nz + 1]) * zinv2 K33(nz + 1);
      }
      Ty = Ty * isredi; Ty1 = Ty1 * isredi;
      a[nz] = -(A2L(C_.Kv, nz, n) + Ty) * zinv1 * zinv * (AREA(nz, n) / AREASVOL(nz, n));
      c[nz] = -(A2L(C_.Kv, nz + 1, n) + Ty1) * zinv2 * zinv * AREA(nz + 1, n) / AREASVOL(nz, n);
      b[nz] = -a[nz] - c[nz] + A2(C_.hnode_new, nz, n);
      zinv1 = zinv2;
      if (do_wimpl) {                                                  /* :604-617 */
        double v_adv = zinv * (AREA(nz, n) / AREASVOL(nz, n));
        a[nz] = a[nz] + dmin(0.0, A2L(C_.Wvel_i, nz, n)) * v_adv;
        b[nz] = b[nz] + dmax(0.0, A2L(C_.Wvel_i, nz, n)) * v_adv;
        v_adv = zinv * AREA(nz + 1, n) / AREASVOL(nz, n);
        b[nz] = b[nz] - dmin(0.0, A2L(C_.Wvel_i, nz + 1, n)) * v_adv;
        c[nz] = c[nz] - dmax(0.0, A2L(C_.Wvel_i, nz + 1, n)) * v_adv;
      }
    }
    nz = nzmax - 1;
    zinv = 1.0 * dt;
    if (C_.p.Redi) Ty = (Z_n[nz - 1] - zbar_n[nz]) * zinv1 K33(nz - 1) + (zbar_n[nz] - Z_n[nz]) * zinv1 K33(nz);
    Ty = Ty * isredi;
#undef K33
    a[nz] = -(A2L(C_.Kv, nz, n) + Ty) * zinv1 * zinv * (AREA(nz, n) / AREASVOL(nz, n));
    c[nz] = 0.0;
    b[nz] = -a[nz] + A2(C_.hnode_new, nz, n);
    if (do_wimpl) {                                                    /* :641-649 */
      double v_adv = zinv * (AREA(nz, n) / AREASVOL(nz, n));
      a[nz] = a[nz] + dmin(0.0, A2L(C_.Wvel_i, nz, n)) * v_adv;
      b[nz] = b[nz] + dmax(0.0, A2L(C_.Wvel_i, nz, n)) * v_adv;
    }
    nz = nzmin;
    double dz = A2(C_.hnode_new, nz, n);
    trv[nz] = -(b[nz] - dz) * TR(nz, n, tr) - c[nz] * TR(nz + 1, n, tr);
    for (nz = nzmin + 1; nz <= nzmax - 2; nz++) {
      dz = A2(C_.hnode_new, nz, n);
      trv[nz] = -a[nz] * TR(nz - 1, n, tr) - (b[nz] - dz) * TR(nz, n, tr) - c[nz] * TR(nz + 1, n, tr);
    }
    nz = nzmax - 1;
    dz = A2(C_.hnode_new, nz, n);
    trv[nz] = -a[nz] * TR(nz - 1, n, tr) - (b[nz] - dz) * TR(nz, n, tr);
    if (C_.p.use_kpp_nonlclflx && C_.p.mix_scheme == 1 && (tr == 1 || tr == 2)) {      /* KPP non-local transport, oce_ale_tracer.F90:688-724 */
      const double *bl = C_.kpp_blmc[tr];                /* blmc(:,:,2) for heat, (:,:,3) for salt */
      const double rsss = C_.p.ref_sss_local ? TR(1, n, 2) : C_.p.ref_sss;
#define GH(z) (dmin(A2(C_.kpp_ghats, z, n) * A2L(bl, z, n), 1.0) * (AREA(z, n) / AREASVOL(nz, n)))
      for (nz = nzmin; nz <= nzmax - 1; nz++) {
        double X;
        if (nz == nzmin) X = -GH(nz + 1);
        else if (nz <= nzmax - 2) X = GH(nz) - GH(nz + 1);
        else X = GH(nz);
        if (tr == 1) trv[nz] = trv[nz] + X * C_.heat_flux[n - 1] / VCPW * dt;
        else trv[nz] = trv[nz] - X * rsss * C_.water_flux[n - 1] * dt;
      }
#undef GH
    }
    if (C_.p.use_sw_pene && tr == 1)                   /* short-wave penetration, oce_ale_tracer.F90:785-791 */
      for (nz = nzmin; nz <= nzmax - 1; nz++)
        trv[nz] = trv[nz] + (A2L(C_.sw_3d, nz, n) - A2L(C_.sw_3d, nz + 1, n) * AREA(nz + 1, n) / AREASVOL(nz, n)) * (1.0 * dt);
    trv[nzmin] = trv[nzmin] + bc_surface(n, tr - 1);
    cp[nzmin] = c[nzmin] / b[nzmin];
    tp[nzmin] = trv[nzmin] / b[nzmin];
    for (nz = nzmin + 1; nz <= nzmax - 1; nz++) {
      double m = b[nz] - cp[nz - 1] * a[nz];
      cp[nz] = c[nz] / m;
      tp[nz] = (trv[nz] - tp[nz - 1] * a[nz]) / m;
    }
    trv[nzmax - 1] = tp[nzmax - 1];
    for (nz = nzmax - 2; nz >= nzmin; nz--) trv[nz] = tp[nz] - cp[nz] * trv[nz + 1];
    for (nz = nzmin; nz <= nzmax - 1; nz++) TR(nz, n, tr) = TR(nz, n, tr) + trv[nz];
  }
  free(buf);
}

/* diff_part_bh (smooth_bh_tra): src/oce_ale_tracer.F90:1081-1150, biharmonic diffusion of the tracer as a filter with the flow-dependent
 * coefficient of the momentum filters (gamma0, gamma1, gamma2) */
static void diff_part_bh(int tr) {
  double *tmp = calloc((size_t)NLM1 * C_.N, sizeof(double));
  const double g0 = C_.p.gamma0, g1 = C_.p.gamma1, g2 = C_.p.gamma2, dt = C_.p.dt;
  for (int stage = 0; stage < 2; stage++) {
    for (int ed = 1; ed <= C_.D; ed++) {
      if (C_.m.myList_edge2D[ed - 1] > C_.m.edge2D_in) continue;
      int e1 = ETRI(1, ed), e2 = ETRI(2, ed), n1 = EDG(1, ed), n2 = EDG(2, ed);
      double len = sqrt(C_.m.elem_area[e1 - 1] + C_.m.elem_area[e2 - 1]);
      int ul1 = C_.m.ulevels_nod2D_max[n1 - 1] < C_.m.ulevels_nod2D_max[n2 - 1] ? C_.m.ulevels_nod2D_max[n1 - 1] : C_.m.ulevels_nod2D_max[n2 - 1];
      int nl1 = (C_.m.nlevels_nod2D_min[n1 - 1] > C_.m.nlevels_nod2D_min[n2 - 1] ? C_.m.nlevels_nod2D_min[n1 - 1] : C_.m.nlevels_nod2D_min[n2 - 1]) - 1;
      for (int nz = ul1; nz <= nl1; nz++) {
        double u1 = V2(C_.UV, 1, nz, e1) - V2(C_.UV, 1, nz, e2), v1 = V2(C_.UV, 2, nz, e1) - V2(C_.UV, 2, nz, e2);
        double vi = u1 * u1 + v1 * v1;
        vi = sqrt(dmax(g0, dmax(g1 * sqrt(vi), g2 * vi)) * len);
        if (stage == 0) {
          double tt = (TR(nz, n1, tr) - TR(nz, n2, tr)) * vi;
          A2(tmp, nz, n1) = A2(tmp, nz, n1) - tt; A2(tmp, nz, n2) = A2(tmp, nz, n2) + tt;
        } else {
          double tt = -(A2(tmp, nz, n1) - A2(tmp, nz, n2)) * vi * dt;
          TR(nz, n1, tr) = TR(nz, n1, tr) - tt / AREA(nz, n1); TR(nz, n2, tr) = TR(nz, n2, tr) + tt / AREA(nz, n2);
        }
      }
    }
    /* (exchange_nod(temporary_ttf): single partition) */
  }
  free(tmp);
}

/* diff_tracers_ale: src/oce_ale_tracer.F90:253-325 */
void orc_diff_tracers_ale(int tr) {
  size_t cnt = (size_t)NLM1 * C_.N;
  memcpy(&TRO(1, 1, tr), &TR(1, 1, tr), sizeof(double) * cnt);
  if (C_.p.with_diffusion) diff_part_hor();
  if (C_.p.with_diffusion && C_.p.Redi) diff_ver_part_redi_expl();
  for (int n = 1; n <= C_.m.myDim_nod2D; n++)
    for (int nz = ULEVN(n); nz <= NLEVN(n) - 1; nz++) {
      A2(C_.del_ttf, nz, n) = A2(C_.del_ttf, nz, n) + TR(nz, n, tr) * (A2(C_.hnode, nz, n) - A2(C_.hnode_new, nz, n));
      TR(nz, n, tr) = TR(nz, n, tr) + A2(C_.del_ttf, nz, n) / A2(C_.hnode_new, nz, n);
    }
  if (C_.p.with_diffusion && C_.p.i_vert_diff) diff_ver_part_impl_ale(tr);
  if (C_.p.smooth_bh_tra) diff_part_bh(tr);
}

/* cal_rejected_salt + app_rejected_salt (SPP): src/oce_spp.F90.  spar(k)**n_distr with the integer variable n_distr = 5 is what the compiler's runtime makes of
 * it (__powidf2: repeated squaring, x * (x^2)^2). */
void orc_spp(void) {
  const double aux = 910. / 1025. * C_.p.dt;
  double spar[128];
  for (int n = 1; n <= C_.N; n++) {
    const int nzmin = ULEVN(n), nzmax = NLEVN(n);
    double rej = 0.0;
    if (C_.thdgr[n - 1] > 0.0 && nzmin == 1) rej = (C_.S_oc_array[n - 1] - C_.p.Sice) * C_.thdgr[n - 1] * aux * AREA(1, n);
    if (nzmin > 1) continue;
    if (rej <= 0.0) continue;
    if (TR(nzmin, n, 2) < 10.0) continue;
    if (!(C_.m.geo_coord_nod2D[2 * (n - 1) + 1] > 0.0)) continue;
    int kml = 1;
    spar[nzmin] = 0.0;
    for (int k = nzmin; k <= nzmax; k++) {
      const double drhodz = A2L(C_.bvfreq, k, n) * DENSITY_0 / G_ACC;
      if (drhodz >= 0.01 || A2(C_.Z_3d_n, k, n) < -50.0) break;
      kml = kml + 1;
      const double x = A2(C_.Z_3d_n, 1, n) - A2(C_.Z_3d_n, k + 1, n), x2 = x * x;
      spar[k + 1] = AREA(k + 1, n) * A2(C_.hnode, k + 1, n) * (x * (x2 * x2));
    }
    if (kml > nzmin) {
      TR(nzmin, n, 2) = TR(nzmin, n, 2) - rej / AREASVOL(1, n) / A2(C_.hnode, 1, n);
      double ssum = 0.0;
      for (int k = nzmin + 1; k <= kml; k++) ssum = ssum + spar[k];
      for (int k = nzmin + 1; k <= kml; k++) TR(k, n, 2) = TR(k, n, 2) + rej * (spar[k] / ssum) / AREASVOL(k, n) / A2(C_.hnode, k, n);
    }
  }
}

/* relax_to_clim: src/oce_tracer_mod.F90:86-121 (clim_relax > 0; T towards Tclim, S towards Sclim at the nodal rate relax2clim) */
void orc_relax_to_clim(int tr) {
  if (!(C_.p.clim_relax > 1.0e-8) || tr > 2) return;
  const double *cl = tr == 1 ? C_.Tclim : C_.Sclim;
  for (int n = 1; n <= C_.m.myDim_nod2D; n++)
    for (int nz = ULEVN(n); nz <= NLEVN(n) - 1; nz++)
      TR(nz, n, tr) = TR(nz, n, tr) + C_.relax2clim[n - 1] * C_.p.dt * (A2(cl, nz, n) - TR(nz, n, tr));
}

/* solve_tracers_ale tail: salinity clamp, src/oce_ale_tracer.F90:176-198 */
void orc_salinity_clamp(void) {
  for (int n = 1; n <= C_.N; n++)
    for (int nz = ULEVN(n); nz <= NLEVN(n) - 1; nz++) {
      if (TR(nz, n, 2) > 45.0) TR(nz, n, 2) = 45.0;
      if (TR(nz, n, 2) < 3.0) TR(nz, n, 2) = 3.0;
    }
}

/* oce_mixing_PP: src/oce_ale_mixing_pp.F90:2-83 (Kv0_const) */
void orc_mixing_pp(void) {
  const double mix_coeff_PP = 0.01;
  for (int n = 1; n <= C_.N; n++)
    for (int nz = ULEVN(n) + 1; nz <= NLEVN(n) - 1; nz++) {
      double dz_inv = 1.0 / (A2(C_.Z_3d_n, nz - 1, n) - A2(C_.Z_3d_n, nz, n));
      double du = V2(C_.Unode, 1, nz - 1, n) - V2(C_.Unode, 1, nz, n), dv = V2(C_.Unode, 2, nz - 1, n) - V2(C_.Unode, 2, nz, n);
      double shear = du * du + dv * dv;
      shear = shear * dz_inv * dz_inv;
      A2L(C_.Kv, nz, n) = shear / (shear + 5. * dmax(A2L(C_.bvfreq, nz, n), 0.0) + 1.0e-14);
    }
  for (int e = 1; e <= C_.m.myDim_elem2D; e++) {
    int n1 = EN(1, e), n2 = EN(2, e), n3 = EN(3, e);
    for (int nz = ULEV(e) + 1; nz <= NLEV(e) - 1; nz++) {
      double k1 = A2L(C_.Kv, nz, n1), k2 = A2L(C_.Kv, nz, n2), k3 = A2L(C_.Kv, nz, n3);
      A2L(C_.Av, nz, e) = mix_coeff_PP * (k1 * k1 + k2 * k2 + k3 * k3) / 3.0 + C_.p.A_ver;
    }
  }
  for (int n = 1; n <= C_.N; n++)
    for (int nz = ULEVN(n) + 1; nz <= NLEVN(n) - 1; nz++) {
      double k = A2L(C_.Kv, nz, n);
      A2L(C_.Kv, nz, n) = mix_coeff_PP * (k * k * k) + (C_.p.Kv0_const ? C_.p.K_ver : orc_kv0_background_qiang(n, nz));
    }
}

/* pmlktmo (Monin-Obukhov length, src/oce_mo_conv.F90:148-182) and mo_length (:107-145).  The reference is built with
 * -fdefault-real-8 (src/CMakeLists.txt:73): its unsuffixed literals (cosgam, qhw = 1/7.0, betas, betat) are doubles. */
static double mo_pmlktmo(double qfm, double qtm, double qw) {
  const double qhw = 1 / 7.0, betas = 0.0008, betat = 0.00004;
  double qrho = betas * qfm - betat * qtm, ttmp = 60.0;
  if (qrho > 0.) ttmp = 0.0;
  else
    for (int iter = 1; iter <= 5; iter++) {
      double a1 = exp(-ttmp * qhw);
      double f0 = 2.0 * qw * a1 + 9.81 * qrho * ttmp;
      double f1 = -(2.0 * qw * a1 * qhw) + 9.81 * qrho;
      ttmp = ttmp - f0 / f1;
      ttmp = dmax(ttmp, 10.0);
    }
  return dmax(ttmp, 10.0);
}
static void mo_length(double water_flux, double heat_flux, double sx, double sy, double ui, double vi, double ai, double dt, double *mixlength) {
  const double cosgam = 0.913632;
  double qfm = water_flux * 34.0, qtm = -2.38e-7 * heat_flux;
  double tau = sqrt(sx * sx + sy * sy), ustar = sqrt(tau / 1030.0), uabs = sqrt(ui * ui + vi * vi);
  double qw = 1.25 * (ustar * ustar * ustar) * (1.0 - ai) + 0.005 * (uabs * uabs * uabs) * cosgam * ai;
  double obuk = mo_pmlktmo(qfm, qtm, qw);
  double rtc = dt / (10.0 * 86400.0);
  if (obuk < *mixlength) { double ret = (obuk - *mixlength) * rtc; *mixlength = *mixlength + ret; }
  else *mixlength = obuk;
}

/* mo_convect: src/oce_mo_conv.F90:4-103 (use_momix: Timmermann & Beckmann 2004 south of momix_lat; instabmix; windmix) */
void orc_mo_convect(void) {
  const double rad = 3.14159265358979 / 180.0;        /* oce_modules.F90:11-12 */
  double *mo = NULL;
  if (C_.p.use_momix) {
    mo = calloc((size_t)NL * C_.N, sizeof(double));
    for (int n = 1; n <= C_.N; n++) {
      int nzmax = NLEVN(n), nzmin = ULEVN(n);
      if (C_.m.geo_coord_nod2D[2 * (n - 1) + 1] > C_.p.momix_lat * rad) continue;
      if (nzmin > 1) continue;
      mo_length(C_.water_flux[n - 1], C_.heat_flux[n - 1], C_.stress_atmoce_x[n - 1], C_.stress_atmoce_y[n - 1], C_.u_ice[n - 1], C_.v_ice[n - 1],
                C_.a_ice[n - 1], C_.p.dt, &C_.mixlength[n - 1]);
      for (int nz = nzmin + 1; nz <= nzmax - 1; nz++)
        if (fabs(A2L(C_.zbar_3d_n, nz, n)) <= C_.mixlength[n - 1]) {
          A2L(mo, nz, n) = C_.p.momix_kv;
          A2L(C_.Kv, nz, n) = A2L(C_.Kv, nz, n) + A2L(mo, nz, n);
        }
    }
  }
  for (int n = 1; n <= C_.N; n++) {
    int nzmin = ULEVN(n);
    for (int nz = nzmin + 1; nz <= NLEVN(n) - 1; nz++) {
      if (C_.p.use_instabmix && A2L(C_.bvfreq, nz, n) < 0.) A2L(C_.Kv, nz, n) = dmax(A2L(C_.Kv, nz, n), C_.p.instabmix_kv);
      if (nzmin > 1) continue;
      if (C_.p.use_windmix && nz <= C_.p.windmix_nl + 1) A2L(C_.Kv, nz, n) = dmax(A2L(C_.Kv, nz, n), C_.p.windmix_kv);
    }
  }
  for (int e = 1; e <= C_.m.myDim_elem2D; e++) {
    int n1 = EN(1, e), n2 = EN(2, e), n3 = EN(3, e), nzmin = ULEV(e);
    for (int nz = nzmin + 1; nz <= NLEV(e) - 1; nz++) {
      if (C_.p.use_instabmix && (A2L(C_.bvfreq, nz, n1) < 0. || A2L(C_.bvfreq, nz, n2) < 0. || A2L(C_.bvfreq, nz, n3) < 0.))
        A2L(C_.Av, nz, e) = dmax(A2L(C_.Av, nz, e), C_.p.instabmix_kv);
      if (nzmin > 1) continue;
      if (C_.p.use_momix && ((C_.m.geo_coord_nod2D[2 * (n1 - 1) + 1] + C_.m.geo_coord_nod2D[2 * (n2 - 1) + 1]) + C_.m.geo_coord_nod2D[2 * (n3 - 1) + 1]) / 3.0 <= C_.p.momix_lat * rad)
        A2L(C_.Av, nz, e) = A2L(C_.Av, nz, e) + ((A2L(mo, nz, n1) + A2L(mo, nz, n2)) + A2L(mo, nz, n3)) / 3.0;
      if (C_.p.use_windmix && nz <= C_.p.windmix_nl + 1) A2L(C_.Av, nz, e) = dmax(A2L(C_.Av, nz, e), C_.p.windmix_kv);
    }
  }
  free(mo);
}
