/* ORACLE (test infrastructure): dynamics part of the step.  Each function restates one
 * reference subroutine in its loop and expression order; citations at each function. */
#include "orc.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>

/* compute_vel_nodes: src/oce_dyn.F90:133-169 */
void orc_compute_vel_nodes(void) {
  for (int n = 1; n <= C_.m.myDim_nod2D; n++)
    for (int nz = ULEVN(n); nz <= NLEVN(n) - 1; nz++) {
      double tvol = 0.0, tx = 0.0, ty = 0.0;
      for (int k = 1; k <= C_.m.nod_in_elem2D_num[n - 1]; k++) {
        int e = NIE(k, n);
        if (NLEV(e) - 1 < nz || nz < ULEV(e)) continue;
        double a = C_.m.elem_area[e - 1];
        tvol = tvol + a;
        tx = tx + V2(C_.UV, 1, nz, e) * a;
        ty = ty + V2(C_.UV, 2, nz, e) * a;
      }
      V2(C_.Unode, 1, nz, n) = tx / tvol;
      V2(C_.Unode, 2, nz, n) = ty / tvol;
    }
}

/* densityJM_components: src/oce_ale_pressure_bv.F90:2586-2654 ; density_linear: :2992-3019 */
static void eos(double t, double s, double *bulk_0, double *bulk_pz, double *bulk_pz2, double *rhopot) {
  if (C_.p.state_equation == 0) {
    *bulk_0 = 1; *bulk_pz = 0; *bulk_pz2 = 0;
    if (C_.p.toy_soufflet) *rhopot = DENSITY_0 - 0.00025 * (t - 10.0) * DENSITY_0;
    else *rhopot = DENSITY_0 + 0.8 * (s - 34.0) - 0.2 * (t - 20.0);
    return;
  }
  const double a0 = 19092.56, at = 209.8925, at2 = -3.041638, at3 = -1.852732e-3, at4 = -1.361629e-5;
  const double as = 104.4077, ast = -6.500517, ast2 = .1553190, ast3 = 2.326469e-4;
  const double ass = -5.587545, asst = 0.7390729, asst2 = -1.909078e-2;
  const double ap = -4.721788e-1, apt = -1.028859e-2, apt2 = 2.512549e-4, apt3 = 5.939910e-7;
  const double aps = 1.571896e-2, apst = 2.598241e-4, apst2 = -7.267926e-6, apss = -2.042967e-3;
  const double ap2 = 1.045941e-5, ap2t = -5.782165e-10, ap2t2 = 1.296821e-7;
  const double ap2s = -2.595994e-7, ap2st = -1.248266e-9, ap2st2 = -3.508914e-9;
  const double b0 = 999.842594, bt = 6.793952e-2, bt2 = -9.095290e-3, bt3 = 1.001685e-4, bt4 = -1.120083e-6, bt5 = 6.536332e-9;
  const double bs = 0.824493, bst = -4.08990e-3, bst2 = 7.64380e-5, bst3 = -8.24670e-7, bst4 = 5.38750e-9;
  const double bss = -5.72466e-3, bsst = 1.02270e-4, bsst2 = -1.65460e-6, bss2 = 4.8314e-4;
  double s_sqrt = sqrt(s);
  *bulk_0 = a0 + t * (at + t * (at2 + t * (at3 + t * at4)))
          + s * (as + t * (ast + t * (ast2 + t * ast3)) + s_sqrt * (ass + t * (asst + t * asst2)));
  *bulk_pz = ap + t * (apt + t * (apt2 + t * apt3)) + s * (aps + t * (apst + t * apst2) + s_sqrt * apss);
  *bulk_pz2 = ap2 + t * (ap2t + t * ap2t2) + s * (ap2s + t * (ap2st + t * ap2st2));
  *rhopot = b0 + t * (bt + t * (bt2 + t * (bt3 + t * (bt4 + t * bt5))))
          + s * (bs + t * (bst + t * (bst2 + t * (bst3 + t * bst4))) + s_sqrt * (bss + t * (bsst + t * bsst2)) + s * bss2);
}

/* init_ref_density: src/oce_ale_pressure_bv.F90:3024-3070 (ocean_setup, once, from the initial Z_3d_n; nzmin = 1 as the reference has it) */
void orc_init_ref_density(void) {
  double b0, bpz, bpz2, rp;
  eos(C_.p.density_ref_T, C_.p.density_ref_S, &b0, &bpz, &bpz2, &rp);
  for (size_t i = 0; i < (size_t)NLM1 * C_.N; i++) C_.density_ref[i] = 0.0;
  for (int n = 1; n <= C_.N; n++) {
    const int nzmin = 1, nzmax = NLEVN(n) - 1;
    double auxz = A2(C_.Z_3d_n, nzmin, n) < 0.0 ? A2(C_.Z_3d_n, nzmin, n) : 0.0;
    double rho = b0 + auxz * bpz + auxz * bpz2;
    A2(C_.density_ref, nzmin, n) = rho * rp / (rho + 0.1 * auxz);
    for (int nz = nzmin + 1; nz <= nzmax; nz++) {
      auxz = A2(C_.Z_3d_n, nz, n);
      rho = b0 + auxz * bpz + auxz * bpz2;
      A2(C_.density_ref, nz, n) = rho * rp / (rho + 0.1 * auxz);
    }
  }
}

/* pressure_bv: src/oce_ale_pressure_bv.F90:106-365 */
void orc_pressure_bv(void) {
  int nl = NL;
  double *rhopot = malloc(sizeof(double) * (nl + 1) * 6);
  double *bulk_0 = rhopot + (nl + 1), *bulk_pz = bulk_0 + (nl + 1), *bulk_pz2 = bulk_pz + (nl + 1), *rho = bulk_pz2 + (nl + 1),
         *dbsfc1 = rho + (nl + 1);
  const double sigma_theta_crit = 0.125;
  double seq = (double)C_.p.state_equation;
  for (int n = 1; n <= C_.N; n++) {
    int nzmin = ULEVN(n), nzmax = NLEVN(n);
    for (int k = 0; k <= nl; k++) rho[k] = bulk_0[k] = bulk_pz[k] = bulk_pz2[k] = rhopot[k] = dbsfc1[k] = 0.0;
    double db_max = 0.0;
    for (int nz = nzmin; nz <= nzmax - 1; nz++)
      eos(TR(nz, n, 1), TR(nz, n, 2), &bulk_0[nz], &bulk_pz[nz], &bulk_pz2[nz], &rhopot[nz]);
    for (int nz = nzmin; nz <= nzmax - 1; nz++) {
      double z = A2(C_.Z_3d_n, nz, n);
      rho[nz] = bulk_0[nz] + z * (bulk_pz[nz] + z * bulk_pz2[nz]);
      rho[nz] = rho[nz] * rhopot[nz] / (rho[nz] + 0.1 * z * seq) - A2(C_.density_ref, nz, n);
      A2(C_.density_m_rho0, nz, n) = rho[nz];
      double rho_surf = bulk_0[nzmin] + z * (bulk_pz[nzmin] + z * bulk_pz2[nzmin]);
      rho_surf = rho_surf * rhopot[nzmin] / (rho_surf + 0.1 * z * seq);
      double rr = rho[nz] + A2(C_.density_ref, nz, n);
      dbsfc1[nz] = -G_ACC * (rho_surf - rr) / rr;
      int kk = nz > nzmin + 1 ? nz : nzmin + 1;
      db_max = dmax(dbsfc1[nz] / fabs(A2(C_.Z_3d_n, nzmin, n) - A2(C_.Z_3d_n, kk, n)), db_max);
    }
    dbsfc1[nzmax] = dbsfc1[nzmax - 1];
    for (int nz = nzmin; nz <= nzmax; nz++) A2L(C_.dbsfc, nz, n) = dbsfc1[nz];
    if (nzmin > 1) {      /* :235-258 the levels the ice shelf occupies take the density of the water mass at the cavity-ocean interface */
      double t = TR(nzmin, n, 1), sal = TR(nzmin, n, 2);
      for (int nz = 1; nz <= nzmin - 1; nz++) {
        eos(t, sal, &bulk_0[nz], &bulk_pz[nz], &bulk_pz2[nz], &rhopot[nz]);
        double z = A2(C_.Z_3d_n, nz, n);
        rho[nz] = bulk_0[nz] + z * (bulk_pz[nz] + z * bulk_pz2[nz]);
        rho[nz] = rho[nz] * rhopot[nz] / (rho[nz] + 0.1 * z * seq) - A2(C_.density_ref, nz, n);
        A2(C_.density_m_rho0, nz, n) = rho[nz];
      }
    }
    if (C_.p.which_ale == 0 || C_.p.use_cavity) {      /* :262 */
      if (nzmin > 1) {      /* :268-275 pressure at the cavity-ocean interface */
        A2L(C_.hpressure, nzmin, n) = 0.5 * (A2L(C_.zbar_3d_n, 1, n) - A2L(C_.zbar_3d_n, 2, n)) * rho[1] * G_ACC;
        for (int nz = 2; nz <= nzmin; nz++) {
          double a = 0.5 * G_ACC * (rho[nz - 1] * (A2L(C_.zbar_3d_n, nz - 1, n) - A2L(C_.zbar_3d_n, nz, n)) + rho[nz] * (A2L(C_.zbar_3d_n, nz, n) - A2L(C_.zbar_3d_n, nz + 1, n)));
          A2L(C_.hpressure, nzmin, n) = A2L(C_.hpressure, nzmin, n) + a;
        }
      } else
      A2L(C_.hpressure, nzmin, n) = -A2(C_.Z_3d_n, nzmin, n) * rho[nzmin] * G_ACC;
      for (int nz = nzmin + 1; nz <= nzmax - 1; nz++) {
        double a = 0.5 * G_ACC * (rho[nz - 1] * A2(C_.hnode, nz - 1, n) + rho[nz] * A2(C_.hnode, nz, n));
        A2L(C_.hpressure, nz, n) = A2L(C_.hpressure, nz - 1, n) + a;
      }
    }
    C_.MLD1[n - 1] = A2(C_.Z_3d_n, nzmin + 1, n);
    C_.MLD2[n - 1] = A2(C_.Z_3d_n, nzmin + 1, n);
    C_.MLD1_ind[n - 1] = nzmin + 1;
    int flag1 = 1, flag2 = 1;
    for (int nz = nzmin + 1; nz <= nzmax - 1; nz++) {
      double zb = A2L(C_.zbar_3d_n, nz, n);
      double bulk_up = bulk_0[nz - 1] + zb * (bulk_pz[nz - 1] + zb * bulk_pz2[nz - 1]);
      double bulk_dn = bulk_0[nz] + zb * (bulk_pz[nz] + zb * bulk_pz2[nz]);
      double rho_up = bulk_up * rhopot[nz - 1] / (bulk_up + 0.1 * zb * seq);
      double rho_dn = bulk_dn * rhopot[nz] / (bulk_dn + 0.1 * zb * seq);
      double dz_inv = 1.0 / (A2(C_.Z_3d_n, nz - 1, n) - A2(C_.Z_3d_n, nz, n));
      A2L(C_.bvfreq, nz, n) = -G_ACC * dz_inv * (rho_up - rho_dn) / DENSITY_0;
      if (A2L(C_.bvfreq, nz, n) > db_max && flag1) { C_.MLD1[n - 1] = A2(C_.Z_3d_n, nz, n); C_.MLD1_ind[n - 1] = nz; flag1 = 0; }
      if ((rhopot[nz] - rhopot[nzmin] > sigma_theta_crit) && flag2) {
        C_.MLD2[n - 1] = C_.MLD2[n - 1] + (A2(C_.Z_3d_n, nz, n) - C_.MLD2[n - 1]) / (rhopot[nz] - rhopot[nz - 1] + 1.e-20) *
                                              (rhopot[1] + sigma_theta_crit - rhopot[nz - 1]);
        flag2 = 0;
      } else if (flag2) C_.MLD2[n - 1] = A2(C_.Z_3d_n, nz, n);
    }
    A2L(C_.bvfreq, nzmin, n) = A2L(C_.bvfreq, nzmin + 1, n);
    A2L(C_.bvfreq, nzmax, n) = A2L(C_.bvfreq, nzmax - 1, n);
  }
  free(rhopot);
}

/* pressure_force_4_zxxxx_shchepetkin: src/oce_ale_pressure_bv.F90:1878-2104 ;
 * pressure_force_4_linfs_fullcell: :432-466 ;
 * pressure_force_4_linfs_shchepetkin (linfs with partial cells): :647-891 -- the same integral, the density-Jacobian correction only
 * in the bottom layer (the levels above are flat: "in case linfs: dz_dx == 0.0", :799) */
static void pgf_linfs_fullcell(void);
/* pressure_force_4_zxxxx_cubicspline: src/oce_ale_pressure_bv.F90:1697-1866 (which_pgf = 'cubicspline': the density of the three nodes is
 * interpolated to the mid-depth of the element layer with a monotonised cubic spline, "like in FESOM1.4") */
static void pgf_zxxxx_cubicspline(void) {
  int nl = NL;
  double *zbar_n = calloc(nl + 2, sizeof(double)), *Z_n = calloc(nl + 2, sizeof(double));
  for (int e = 1; e <= C_.m.myDim_elem2D; e++) {
    int nle = NLEV(e) - 1, ule = ULEV(e);
    int en[3] = {EN(1, e), EN(2, e), EN(3, e)};
    for (int k = 0; k <= nl; k++) { zbar_n[k] = 0.0; Z_n[k] = 0.0; }
    zbar_n[nle + 1] = C_.m.zbar_e_bot[e - 1];
    Z_n[nle] = zbar_n[nle + 1] + A2(C_.helem, nle, e) / 2.0;
    for (int nlz = nle; nlz >= ule + 1; nlz--) {
      zbar_n[nlz] = zbar_n[nlz + 1] + A2(C_.helem, nlz, e);
      Z_n[nlz - 1] = zbar_n[nlz] + A2(C_.helem, nlz - 1, e) / 2.0;
    }
    zbar_n[ule] = zbar_n[ule + 1] + A2(C_.helem, ule, e);
    double p_grad[2] = {0.0, 0.0};
    for (int nlz = ule; nlz <= nle; nlz++) {
      double rho_n[3];
      for (int ni = 0; ni < 3; ni++) {
        int node = en[ni], nln = NLEVN(node) - 1, uln = ULEVN(node);
        int nlc = nln - 1;
        for (int dd = uln; dd <= nln; dd++)
          if (A2(C_.Z_3d_n, dd, node) <= Z_n[nlz]) { nlc = dd - 1; if (dd == 1) nlc = 1; break; }
        int si[4] = {nlc - 1, nlc, nlc + 1, nlc + 2};
        double s_z[4], s_d[4], s_H, aux1, aux2, s_dup, s_dlo;
        if (nlc == uln) si[0] = uln;
        else if (nlc == nln - 1) si[3] = nlc + 1;
        for (int k = 0; k < 4; k++) { s_z[k] = A2(C_.Z_3d_n, si[k], node); s_d[k] = A2(C_.density_m_rho0, si[k], node); }
        s_H = s_z[2] - s_z[1];
        aux1 = (s_d[2] - s_d[1]) / s_H;
        if (nlc == uln) {                      /* surface case */
          aux2 = (s_d[3] - s_d[2]) / (s_z[3] - s_z[2]);
          s_dlo = 0.0;
          if (aux1 * aux2 > 0.) s_dlo = 2.0 * aux1 * aux2 / (aux1 + aux2);
          s_dup = 1.5 * aux1 - 0.5 * s_dlo;
        } else if (nlc == nln - 1) {           /* bottom case */
          aux2 = (s_d[1] - s_d[0]) / (s_z[1] - s_z[0]);
          s_dup = 0.0;
          if (aux1 * aux2 > 0.) s_dup = 2.0 * aux1 * aux2 / (aux1 + aux2);
          s_dlo = 1.5 * aux1 - 0.5 * s_dup;
        } else {
          aux2 = (s_d[1] - s_d[0]) / (s_z[1] - s_z[0]);
          s_dup = 0.0;
          if (aux1 * aux2 > 0.) s_dup = 2.0 * aux1 * aux2 / (aux1 + aux2);
          aux2 = (s_d[3] - s_d[2]) / (s_z[3] - s_z[2]);
          s_dlo = 0.0;
          if (aux1 * aux2 > 0.) s_dlo = 2.0 * aux1 * aux2 / (aux1 + aux2);
        }
        double a = s_d[1], b = s_dup;
        double c = -(2.0 * s_dup + s_dlo) / s_H + 3.0 * (s_d[2] - s_d[1]) / (s_H * s_H);
        double d = (s_dup + s_dlo) / (s_H * s_H) - 2.0 * (s_d[2] - s_d[1]) / ((s_H * s_H) * s_H);
        double dz = Z_n[nlz] - s_z[1];
        rho_n[ni] = a + b * dz + c * (dz * dz) + d * ((dz * dz) * dz);
      }
      double gx = (GS(1, e) * rho_n[0] + GS(2, e) * rho_n[1]) + GS(3, e) * rho_n[2], gy = (GS(4, e) * rho_n[0] + GS(5, e) * rho_n[1]) + GS(6, e) * rho_n[2];
      double ax = G_ACC * A2(C_.helem, nlz, e) * gx / DENSITY_0, ay = G_ACC * A2(C_.helem, nlz, e) * gy / DENSITY_0;
      A2(C_.pgf_x, nlz, e) = p_grad[0] + ax * 0.5;
      A2(C_.pgf_y, nlz, e) = p_grad[1] + ay * 0.5;
      p_grad[0] = p_grad[0] + ax; p_grad[1] = p_grad[1] + ay;
    }
  }
  free(zbar_n); free(Z_n);
}

/* pressure_force_4_linfs_cubicspline: src/oce_ale_pressure_bv.F90:1252-1444 (linfs with partial cells, which_pgf = 'cubicspline': flat layers, the
 * densities of the bottom layer interpolated to the element's mid-depth with the bottom-case spline) */
static void pgf_linfs_cubicspline(void) {
  int nl = NL;
  double *zbar_n = calloc(nl + 2, sizeof(double)), *Z_n = calloc(nl + 2, sizeof(double));
  for (int e = 1; e <= C_.m.myDim_elem2D; e++) {
    int nle = NLEV(e) - 1, ule = ULEV(e);
    int en[3] = {EN(1, e), EN(2, e), EN(3, e)};
    for (int k = 0; k <= nl; k++) { zbar_n[k] = 0.0; Z_n[k] = 0.0; }
    zbar_n[nle + 1] = C_.m.zbar_e_bot[e - 1];
    Z_n[nle] = zbar_n[nle + 1] + A2(C_.helem, nle, e) / 2.0;
    for (int nlz = nle; nlz >= 2; nlz--) {
      zbar_n[nlz] = zbar_n[nlz + 1] + A2(C_.helem, nlz, e);
      Z_n[nlz - 1] = zbar_n[nlz] + A2(C_.helem, nlz - 1, e) / 2.0;
    }
    zbar_n[1] = zbar_n[2] + A2(C_.helem, 1, e);
    double ip[2] = {0.0, 0.0};
    for (int nlz = ule; nlz <= nle; nlz++) {
      double r3[3];
      if (nlz < nle || nle == ule) for (int ni = 0; ni < 3; ni++) r3[ni] = A2(C_.density_m_rho0, nlz, en[ni]);
      if (nlz == nle && nle > ule)
        for (int ni = 0; ni < 3; ni++) {
          int node = en[ni], nln = NLEVN(node) - 1, uln = ULEVN(node);
          int nlc = nln - 1;
          for (int dd = uln; dd <= nln; dd++)
            if (A2(C_.Z_3d_n, dd, node) <= Z_n[nlz]) { nlc = dd - 1; if (dd == 1) nlc = 1; break; }
          const int si[4] = {nlc - 1, nlc, nlc + 1, nlc + 1};
          double s_z[4], s_d[4];
          for (int k = 0; k < 4; k++) { s_z[k] = A2(C_.Z_3d_n, si[k], node); s_d[k] = A2(C_.density_m_rho0, si[k], node); }
          double s_H = s_z[2] - s_z[1], aux1 = (s_d[2] - s_d[1]) / s_H, aux2 = (s_d[1] - s_d[0]) / (s_z[1] - s_z[0]);
          double s_dup = 0.0;
          if (aux1 * aux2 > 0.) s_dup = 2.0 * aux1 * aux2 / (aux1 + aux2);
          double s_dlo = 1.5 * aux1 - 0.5 * s_dup;
          double c = -(2.0 * s_dup + s_dlo) / s_H + 3.0 * (s_d[2] - s_d[1]) / (s_H * s_H);
          double d = (s_dup + s_dlo) / (s_H * s_H) - 2.0 * (s_d[2] - s_d[1]) / ((s_H * s_H) * s_H);
          double dz = Z_n[nlz] - s_z[1];
          r3[ni] = s_d[1] + s_dup * dz + c * (dz * dz) + d * ((dz * dz) * dz);
        }
      double gx = (GS(1, e) * r3[0] + GS(2, e) * r3[1]) + GS(3, e) * r3[2], gy = (GS(4, e) * r3[0] + GS(5, e) * r3[1]) + GS(6, e) * r3[2];
      double ax = gx * A2(C_.helem, nlz, e) * G_ACC / DENSITY_0, ay = gy * A2(C_.helem, nlz, e) * G_ACC / DENSITY_0;
      if (nlz == ule && ule > 1) {           /* pressure boundary condition at the shelf base (:1316-1335) */
        ip[0] = ip[0] + gx * (-C_.m.zbar_e_srf[e - 1]) * G_ACC / DENSITY_0;
        ip[1] = ip[1] + gy * (-C_.m.zbar_e_srf[e - 1]) * G_ACC / DENSITY_0;
      }
      A2(C_.pgf_x, nlz, e) = ip[0] + ax * 0.5; ip[0] = ip[0] + ax;
      A2(C_.pgf_y, nlz, e) = ip[1] + ay * 0.5; ip[1] = ip[1] + ay;
    }
  }
  free(zbar_n); free(Z_n);
}

/* pressure_force_4_linfs_nemo: src/oce_ale_pressure_bv.F90:479-635 (linfs with partial cells, which_pgf = 'nemo': hydrostatic pressure on the levels above
 * the bottom; in the bottom layer T and S of every node are interpolated to the shallowest of the three bottom mid-depths, the density is formed there and the
 * pressure is integrated over the thinnest bottom layer) */
static void pgf_linfs_nemo(void) {
  const double seq = (double)C_.p.state_equation;
  for (int e = 1; e <= C_.m.myDim_elem2D; e++) {
    const int nle = NLEV(e) - 1, ule = ULEV(e);
    const int en[3] = {EN(1, e), EN(2, e), EN(3, e)};
    for (int nlz = ule; nlz <= nle - 1; nlz++) {
      A2(C_.pgf_x, nlz, e) = (GS(1, e) * A2L(C_.hpressure, nlz, en[0]) / DENSITY_0 + GS(2, e) * A2L(C_.hpressure, nlz, en[1]) / DENSITY_0) + GS(3, e) * A2L(C_.hpressure, nlz, en[2]) / DENSITY_0;
      A2(C_.pgf_y, nlz, e) = (GS(4, e) * A2L(C_.hpressure, nlz, en[0]) / DENSITY_0 + GS(5, e) * A2L(C_.hpressure, nlz, en[1]) / DENSITY_0) + GS(6, e) * A2L(C_.hpressure, nlz, en[2]) / DENSITY_0;
    }
    const double Zn = C_.m.zbar_e_bot[e - 1] + A2(C_.helem, nle, e) / 2.0;                 /* Z_n(nle) */
    double zmax = A2(C_.Z_3d_n, nle, en[0]), dh = A2(C_.hnode, nle, en[0]);
    for (int k = 1; k < 3; k++) { zmax = dmax(zmax, A2(C_.Z_3d_n, nle, en[k])); dh = dmin(dh, A2(C_.hnode, nle, en[k])); }
    double hpb[3];
    for (int ni = 0; ni < 3; ni++) {
      const int n = en[ni], nln = NLEVN(n) - 1, uln = ULEVN(n);
      int pos = 0; double best = 0.0;                               /* minloc of the positive differences, first minimum */
      for (int k = uln; k <= nln; k++) {
        const double dd = A2(C_.Z_3d_n, k, n) - zmax;
        if (dd > 0.0 && (pos == 0 || dd < best)) { pos = k - uln + 1; best = dd; }
      }
      int nlc = pos + 1;
      if (nlc > nln) nlc = nln;
      const double dZn = A2(C_.Z_3d_n, nlc, n) - A2(C_.Z_3d_n, nlc - 1, n), dZn_i = zmax - A2(C_.Z_3d_n, nlc - 1, n);
      double dval = TR(nlc, n, 1) - TR(nlc - 1, n, 1);
      const double ti = TR(nlc - 1, n, 1) + (dval / dZn * dZn_i);
      dval = TR(nlc, n, 2) - TR(nlc - 1, n, 2);
      const double si = TR(nlc - 1, n, 2) + (dval / dZn * dZn_i);
      double b0, bpz, bpz2, rp;
      eos(ti, si, &b0, &bpz, &bpz2, &rp);
      double dens = b0 + Zn * (bpz + Zn * bpz2);
      dens = dens * rp / (dens + 0.1 * Zn * seq) - A2(C_.density_ref, nle, n);
      const int nlce = nlc < nle ? nlc : nle;
      hpb[ni] = A2L(C_.hpressure, nlce - 1, n) + 0.5 * G_ACC * (A2(C_.density_m_rho0, nlce - 1, n) * A2(C_.hnode, nlce - 1, n) + dens * dh);
    }
    A2(C_.pgf_x, nle, e) = ((GS(1, e) * hpb[0] + GS(2, e) * hpb[1]) + GS(3, e) * hpb[2]) / DENSITY_0;
    A2(C_.pgf_y, nle, e) = ((GS(4, e) * hpb[0] + GS(5, e) * hpb[1]) + GS(6, e) * hpb[2]) / DENSITY_0;
  }
}

/* pressure_force_4_zxxxx_easypgf: src/oce_ale_pressure_bv.F90:2116-2546 (which_pgf = 'easypgf': T and S of the three nodes interpolated to the mid-depth of the
 * element layer with the second-order Newton polynomial of three levels, the density formed there; the three blocks of the source -- surface, bulk, bottom -- differ
 * only in the centre level k0 of the stencil) */
static void pgf_zxxxx_easypgf(void) {         /* also pressure_force_4_linfs_easypgf (:898-1245, linfs with partial cells): the interpolation only in the bottom layer */
  const int lin = C_.p.which_ale == 0;
  int nl = NL;
  const double seq = (double)C_.p.state_equation;
  double *zbar_n = calloc(nl + 2, sizeof(double)), *Z_n = calloc(nl + 2, sizeof(double));
  for (int e = 1; e <= C_.m.myDim_elem2D; e++) {
    int nle = NLEV(e) - 1, ule = ULEV(e);
    int en[3] = {EN(1, e), EN(2, e), EN(3, e)};
    for (int k = 0; k <= nl; k++) { zbar_n[k] = 0.0; Z_n[k] = 0.0; }
    zbar_n[nle + 1] = C_.m.zbar_e_bot[e - 1];
    Z_n[nle] = zbar_n[nle + 1] + A2(C_.helem, nle, e) * 0.5;
    for (int nlz = nle; nlz >= ule + 1; nlz--) {
      zbar_n[nlz] = zbar_n[nlz + 1] + A2(C_.helem, nlz, e);
      Z_n[nlz - 1] = zbar_n[nlz] + A2(C_.helem, nlz - 1, e) * 0.5;
    }
    double ip[2] = {0.0, 0.0};
    for (int nlz = ule; nlz <= nle; nlz++) {
      double r3[3];
      for (int ni = 0; ni < 3; ni++) {
        int n = en[ni], k0;
        if (lin && nlz != nle && !(nlz == ule && ule > 1)) { r3[ni] = A2(C_.density_m_rho0, nlz, n); continue; }      /* (directly under a shelf the interpolation stays, :957-1040) */
        if (nlz == ule && (nlz - ULEVN(n)) == 0) k0 = nlz + 1;
        else if (nlz == nle && nlz != ule && (NLEVN(n) - 1 - nlz) == 0) k0 = nlz - 1;
        else k0 = nlz;
        const double zm = A2(C_.Z_3d_n, k0 - 1, n), zc = A2(C_.Z_3d_n, k0, n), zp = A2(C_.Z_3d_n, k0 + 1, n);
        const double dx10 = zc - zm, dx21 = zp - zc, dx20 = zp - zm, Zn = Z_n[nlz];
        double ts[2];
        for (int t = 1; t <= 2; t++) {
          const double x0 = TR(k0 - 1, n, t), d10 = TR(k0, n, t) - TR(k0 - 1, n, t), d21 = TR(k0 + 1, n, t) - TR(k0, n, t);
          ts[t - 1] = x0 + d10 / dx10 * (Zn - zm) + (dx10 * d21 - dx21 * d10) / (dx20 * dx21 * dx10) * (Zn - zc) * (Zn - zm);
        }
        double b0, bpz, bpz2, rp;
        eos(ts[0], ts[1], &b0, &bpz, &bpz2, &rp);
        double rho = b0 + Zn * (bpz + Zn * bpz2);
        r3[ni] = rho * rp / (rho + 0.1 * Zn * seq) - DENSITY_0;
      }
      double gx = (GS(1, e) * r3[0] + GS(2, e) * r3[1]) + GS(3, e) * r3[2], gy = (GS(4, e) * r3[0] + GS(5, e) * r3[1]) + GS(6, e) * r3[2];
      double ax = gx * A2(C_.helem, nlz, e) * G_ACC / DENSITY_0, ay = gy * A2(C_.helem, nlz, e) * G_ACC / DENSITY_0;
      A2(C_.pgf_x, nlz, e) = (nlz == ule) ? ax * 0.5 : ip[0] + ax * 0.5; ip[0] = (nlz == ule) ? ax : ip[0] + ax;
      A2(C_.pgf_y, nlz, e) = (nlz == ule) ? ay * 0.5 : ip[1] + ay * 0.5; ip[1] = (nlz == ule) ? ay : ip[1] + ay;
    }
  }
  free(zbar_n); free(Z_n);
}

static double drho_dy_of(int e, int nlz, const int *en) {
  return GS(4, e) * A2(C_.density_m_rho0, nlz, en[0]) + GS(5, e) * A2(C_.density_m_rho0, nlz, en[1]) + GS(6, e) * A2(C_.density_m_rho0, nlz, en[2]);
}
static double dz_dy_of(int e, int nlz, const int *en) {
  return GS(4, e) * A2(C_.Z_3d_n, nlz, en[0]) + GS(5, e) * A2(C_.Z_3d_n, nlz, en[1]) + GS(6, e) * A2(C_.Z_3d_n, nlz, en[2]);
}
void orc_pressure_force(void) {
  const int cav_pc = C_.p.use_cavity && C_.p.use_cavity_partial_cell;      /* :385-403 */
  if (C_.p.which_ale == 0 && !C_.p.use_partial_cell && !cav_pc) { pgf_linfs_fullcell(); return; }
  if (C_.p.which_ale != 0 && C_.p.which_pgf == 1) { pgf_zxxxx_cubicspline(); return; }
  if (C_.p.which_pgf == 3) { pgf_zxxxx_easypgf(); return; }        /* zstar, and linfs with partial cells (full cells returned above) */
  if (C_.p.which_ale == 0 && C_.p.which_pgf == 1) { pgf_linfs_cubicspline(); return; }
  if (C_.p.which_ale == 0 && C_.p.which_pgf == 2) { pgf_linfs_nemo(); return; }
  const int lin = C_.p.which_ale == 0;
  int nl = NL;
  double *zbar_n = calloc(nl + 2, sizeof(double)), *Z_n = calloc(nl + 2, sizeof(double));
  for (int e = 1; e <= C_.m.myDim_elem2D; e++) {
    int nle = NLEV(e) - 1, ule = ULEV(e);
    int en[3] = {EN(1, e), EN(2, e), EN(3, e)};
    for (int k = 0; k <= nl; k++) { zbar_n[k] = 0.0; Z_n[k] = 0.0; }
    zbar_n[nle + 1] = C_.m.zbar_e_bot[e - 1];
    Z_n[nle] = zbar_n[nle + 1] + A2(C_.helem, nle, e) * 0.5;
    for (int nlz = nle; nlz >= ule + 1; nlz--) {
      zbar_n[nlz] = zbar_n[nlz + 1] + A2(C_.helem, nlz, e);
      Z_n[nlz - 1] = zbar_n[nlz] + A2(C_.helem, nlz - 1, e) * 0.5;
    }
    zbar_n[ule] = zbar_n[ule + 1] + A2(C_.helem, ule, e);
    double int_dp_dx[2] = {0, 0};
    for (int nlz = ule; nlz <= nle; nlz++) {
      double drho_dz[3];
      for (int ni = 0; ni < 3; ni++) {
        int n = en[ni];
        int k0;  /* centre index of the 3-point stencil (k0-1,k0,k0+1) */
        if (nlz == ule && (nlz - ULEVN(n)) == 0) k0 = nlz + 1;
        else if (nlz == nle && nlz != ule && (NLEVN(n) - 1 - nlz) == 0) k0 = nlz - 1;
        else k0 = nlz;
        /* NB for nlz==ule==nle the reference runs the surface block only for ule, then the bottom block: handled below */
        double dx10 = A2(C_.Z_3d_n, k0, n) - A2(C_.Z_3d_n, k0 - 1, n);
        double dx21 = A2(C_.Z_3d_n, k0 + 1, n) - A2(C_.Z_3d_n, k0, n);
        double dx20 = A2(C_.Z_3d_n, k0 + 1, n) - A2(C_.Z_3d_n, k0 - 1, n);
        double df10 = A2(C_.density_m_rho0, k0, n) - A2(C_.density_m_rho0, k0 - 1, n);
        double df21 = A2(C_.density_m_rho0, k0 + 1, n) - A2(C_.density_m_rho0, k0, n);
        drho_dz[ni] = df10 / dx10 + (dx10 * df21 - dx21 * df10) / (dx20 * dx21 * dx10) *
                                        ((Z_n[nlz] - A2(C_.Z_3d_n, k0, n)) + (Z_n[nlz] - A2(C_.Z_3d_n, k0 - 1, n)));
      }
      double s3 = (drho_dz[0] + drho_dz[1] + drho_dz[2]) / 3.0;
      double drho_dx = GS(1, e) * A2(C_.density_m_rho0, nlz, en[0]) + GS(2, e) * A2(C_.density_m_rho0, nlz, en[1]) +
                       GS(3, e) * A2(C_.density_m_rho0, nlz, en[2]);
      double dz_dx = GS(1, e) * A2(C_.Z_3d_n, nlz, en[0]) + GS(2, e) * A2(C_.Z_3d_n, nlz, en[1]) + GS(3, e) * A2(C_.Z_3d_n, nlz, en[2]);
      const int flat = lin && nlz != nle && !(nlz == ule && ule > 1);      /* linfs: the correction only in the bottom layer and directly under a shelf (:703-776) */
      double aux = flat ? drho_dx * A2(C_.helem, nlz, e) * G_ACC / DENSITY_0 : (drho_dx - s3 * dz_dx) * A2(C_.helem, nlz, e) * G_ACC / DENSITY_0;
      if (C_.p.which_pgf == 4) {
        /* 'sergey' = pressure_force_4_linfs_cavity (:1451-1663): hydrostatic pressure gradient; half the density-Jacobian term directly under a shelf;
         * with partial cells the bottom layer from the pressure at its upper face + half the term */
        double auy = (drho_dy_of(e, nlz, en) - s3 * dz_dy_of(e, nlz, en)) * A2(C_.helem, nlz, e) * G_ACC / DENSITY_0;
        if (nlz == ule && ule > 1) { A2(C_.pgf_x, nlz, e) = aux * 0.5; A2(C_.pgf_y, nlz, e) = auy * 0.5; }
        else if (nlz == nle && C_.p.use_partial_cell) {
          double h[3];
          for (int ni = 0; ni < 3; ni++) h[ni] = (A2L(C_.hpressure, nlz - 1, en[ni]) + 0.5 * G_ACC * (A2(C_.density_m_rho0, nlz - 1, en[ni]) * A2(C_.hnode, nlz - 1, en[ni])));
          A2(C_.pgf_x, nlz, e) = (GS(1, e) * h[0] / DENSITY_0 + GS(2, e) * h[1] / DENSITY_0 + GS(3, e) * h[2] / DENSITY_0) + aux * 0.5;
          A2(C_.pgf_y, nlz, e) = (GS(4, e) * h[0] / DENSITY_0 + GS(5, e) * h[1] / DENSITY_0 + GS(6, e) * h[2] / DENSITY_0) + auy * 0.5;
        } else {
          A2(C_.pgf_x, nlz, e) = GS(1, e) * A2L(C_.hpressure, nlz, en[0]) / DENSITY_0 + GS(2, e) * A2L(C_.hpressure, nlz, en[1]) / DENSITY_0 + GS(3, e) * A2L(C_.hpressure, nlz, en[2]) / DENSITY_0;
          A2(C_.pgf_y, nlz, e) = GS(4, e) * A2L(C_.hpressure, nlz, en[0]) / DENSITY_0 + GS(5, e) * A2L(C_.hpressure, nlz, en[1]) / DENSITY_0 + GS(6, e) * A2L(C_.hpressure, nlz, en[2]) / DENSITY_0;
        }
        continue;
      }
      A2(C_.pgf_x, nlz, e) = (nlz == ule) ? aux * 0.5 : int_dp_dx[0] + aux * 0.5;
      int_dp_dx[0] = (nlz == ule) ? aux : int_dp_dx[0] + aux;
      double drho_dy = GS(4, e) * A2(C_.density_m_rho0, nlz, en[0]) + GS(5, e) * A2(C_.density_m_rho0, nlz, en[1]) +
                       GS(6, e) * A2(C_.density_m_rho0, nlz, en[2]);
      double dz_dy = GS(4, e) * A2(C_.Z_3d_n, nlz, en[0]) + GS(5, e) * A2(C_.Z_3d_n, nlz, en[1]) + GS(6, e) * A2(C_.Z_3d_n, nlz, en[2]);
      aux = flat ? drho_dy * A2(C_.helem, nlz, e) * G_ACC / DENSITY_0 : (drho_dy - s3 * dz_dy) * A2(C_.helem, nlz, e) * G_ACC / DENSITY_0;
      A2(C_.pgf_y, nlz, e) = (nlz == ule) ? aux * 0.5 : int_dp_dx[1] + aux * 0.5;
      int_dp_dx[1] = (nlz == ule) ? aux : int_dp_dx[1] + aux;
    }
  }
  free(zbar_n); free(Z_n);
}

static void pgf_linfs_fullcell(void) {
  for (int e = 1; e <= C_.m.myDim_elem2D; e++) {
    int en[3] = {EN(1, e), EN(2, e), EN(3, e)};
    for (int nz = ULEV(e); nz <= NLEV(e) - 1; nz++) {
      A2(C_.pgf_x, nz, e) = GS(1, e) * A2L(C_.hpressure, nz, en[0]) / DENSITY_0 + GS(2, e) * A2L(C_.hpressure, nz, en[1]) / DENSITY_0 +
                            GS(3, e) * A2L(C_.hpressure, nz, en[2]) / DENSITY_0;
      A2(C_.pgf_y, nz, e) = GS(4, e) * A2L(C_.hpressure, nz, en[0]) / DENSITY_0 + GS(5, e) * A2L(C_.hpressure, nz, en[1]) / DENSITY_0 +
                            GS(6, e) * A2L(C_.hpressure, nz, en[2]) / DENSITY_0;
    }
  }
}

/* sw_alpha_beta: src/oce_ale_pressure_bv.F90:2736-2821 */
void orc_sw_alpha_beta(void) {
  for (int n = 1; n <= C_.m.myDim_nod2D; n++)
    for (int nz = ULEVN(n); nz <= NLEVN(n) - 1; nz++) {
      double t1 = TR(nz, n, 1) * 1.00024, s1 = TR(nz, n, 2), p1 = fabs(A2(C_.Z_3d_n, nz, n));
      double t1_2 = t1 * t1, t1_3 = t1_2 * t1, t1_4 = t1_3 * t1, p1_2 = p1 * p1, p1_3 = p1_2 * p1;
      double s35 = s1 - 35.0, s35_2 = s35 * s35;
      double beta = 0.785567e-3 - 0.301985e-5 * t1 + 0.555579e-7 * t1_2 - 0.415613e-9 * t1_3 +
                    s35 * (-0.356603e-6 + 0.788212e-8 * t1 + 0.408195e-10 * p1 - 0.602281e-15 * p1_2) + s35_2 * (0.515032e-8) +
                    p1 * (-0.121555e-7 + 0.192867e-9 * t1 - 0.213127e-11 * t1_2) + p1_2 * (0.176621e-12 - 0.175379e-14 * t1) +
                    p1_3 * (0.121551e-17);
      double a_over_b = 0.665157e-1 + 0.170907e-1 * t1 - 0.203814e-3 * t1_2 + 0.298357e-5 * t1_3 - 0.255019e-7 * t1_4 +
                        s35 * (0.378110e-2 - 0.846960e-4 * t1 - 0.164759e-6 * p1 - 0.251520e-11 * p1_2) + s35_2 * (-0.678662e-5) +
                        p1 * (0.380374e-4 - 0.933746e-6 * t1 + 0.791325e-8 * t1_2) + p1_2 * t1_2 * (0.512857e-12) -
                        p1_3 * (0.302285e-13);
      A2(C_.sw_beta, nz, n) = beta;
      A2(C_.sw_alpha, nz, n) = a_over_b * beta;
    }
}

/* compute_sigma_xy: src/oce_ale_pressure_bv.F90:2826-2900 */
void orc_compute_sigma_xy(void) {
  int nl = NL;
  double *tx = malloc(sizeof(double) * 5 * (nl + 1)), *ty = tx + nl + 1, *sx = ty + nl + 1, *sy = sx + nl + 1, *vol = sy + nl + 1;
  for (int n = 1; n <= C_.m.myDim_nod2D; n++) {
    int nln = NLEVN(n) - 1, uln = ULEVN(n);
    for (int k = uln; k <= nln; k++) vol[k] = tx[k] = ty[k] = sx[k] = sy[k] = 0.0;
    for (int k = 1; k <= C_.m.nod_in_elem2D_num[n - 1]; k++) {
      int el = NIE(k, n);
      double ar = C_.m.elem_area[el - 1];
      int e1 = EN(1, el), e2 = EN(2, el), e3 = EN(3, el);
      for (int nz = ULEV(el); nz <= NLEV(el) - 1; nz++) {
        vol[nz] = vol[nz] + ar;
        tx[nz] = tx[nz] + (GS(1, el) * TR(nz, e1, 1) + GS(2, el) * TR(nz, e2, 1) + GS(3, el) * TR(nz, e3, 1)) * ar;
        ty[nz] = ty[nz] + (GS(4, el) * TR(nz, e1, 1) + GS(5, el) * TR(nz, e2, 1) + GS(6, el) * TR(nz, e3, 1)) * ar;
        sx[nz] = sx[nz] + (GS(1, el) * TR(nz, e1, 2) + GS(2, el) * TR(nz, e2, 2) + GS(3, el) * TR(nz, e3, 2)) * ar;
        sy[nz] = sy[nz] + (GS(4, el) * TR(nz, e1, 2) + GS(5, el) * TR(nz, e2, 2) + GS(6, el) * TR(nz, e3, 2)) * ar;
      }
    }
    for (int nz = uln; nz <= nln; nz++) {
      V2(C_.sigma_xy, 1, nz, n) = (-A2(C_.sw_alpha, nz, n) * tx[nz] + A2(C_.sw_beta, nz, n) * sx[nz]) / vol[nz] * DENSITY_0;
      V2(C_.sigma_xy, 2, nz, n) = (-A2(C_.sw_alpha, nz, n) * ty[nz] + A2(C_.sw_beta, nz, n) * sy[nz]) / vol[nz] * DENSITY_0;
    }
  }
  free(tx);
}

/* compute_neutral_slope: src/oce_ale_pressure_bv.F90:2905-2946 */
void orc_compute_neutral_slope(void) {
  const double eps = 5.0e-6, S_cr = 1.0e-2, S_d = 1.0e-3;
  memset(C_.slope_tapered, 0, sizeof(double) * 3 * NLM1 * C_.N);
  for (int n = 1; n <= C_.m.myDim_nod2D; n++)
    for (int nz = ULEVN(n) + 1; nz <= NLEVN(n) - 1; nz++) {
      double ro_z_inv = 2.0 * G_ACC / DENSITY_0 / dmax(A2L(C_.bvfreq, nz, n) + A2L(C_.bvfreq, nz + 1, n), eps * eps);
      double s1 = V2(C_.sigma_xy, 1, nz, n) * ro_z_inv, s2 = V2(C_.sigma_xy, 2, nz, n) * ro_z_inv;
      double s3 = sqrt(s1 * s1 + s2 * s2);
      V3(C_.neutral_slope, 1, nz, n) = s1; V3(C_.neutral_slope, 2, nz, n) = s2; V3(C_.neutral_slope, 3, nz, n) = s3;
      double c = 0.5 * (1.0 + tanh((S_cr - s3) / S_d));
      if ((A2L(C_.bvfreq, nz, n) <= 0.0) || (A2L(C_.bvfreq, nz + 1, n) <= 0.0)) c = 0.0;
      V3(C_.slope_tapered, 1, nz, n) = s1 * c; V3(C_.slope_tapered, 2, nz, n) = s2 * c; V3(C_.slope_tapered, 3, nz, n) = s3 * c;
    }
}

/* momentum_adv_scalar: src/oce_ale_vel_rhs.F90:154-343 */
static void momentum_adv_scalar(void) {
  int nl = NL;
  double *wu = malloc(sizeof(double) * 4 * (nl + 2)), *wv = wu + nl + 2, *un1 = wv + nl + 2, *un2 = un1 + nl + 2;
  double *Ur = C_.Unode_rhs;
  for (int n = 1; n <= C_.m.myDim_nod2D; n++) {
    int nl1 = NLEVN(n) - 1, ul1 = ULEVN(n);
    for (int k = 1; k <= nl1 + 1; k++) { wu[k] = 0.0; wv[k] = 0.0; }
    for (int k = 1; k <= C_.m.nod_in_elem2D_num[n - 1]; k++) {
      int el = NIE(k, n);
      int nle = NLEV(el) - 1, ule = ULEV(el);
      double ar = C_.m.elem_area[el - 1];
      if (ule == 1) {
        wu[ule] = wu[ule] + V2(C_.UV, 1, ule, el) * ar;
        wv[ule] = wv[ule] + V2(C_.UV, 2, ule, el) * ar;
      }
      for (int nz = ule + 1; nz <= nle; nz++) {
        wu[nz] = wu[nz] + 0.5 * (V2(C_.UV, 1, nz, el) + V2(C_.UV, 1, nz - 1, el)) * ar;
        wv[nz] = wv[nz] + 0.5 * (V2(C_.UV, 2, nz, el) + V2(C_.UV, 2, nz - 1, el)) * ar;
      }
    }
    for (int nz = ul1; nz <= nl1; nz++) { wu[nz] = wu[nz] * A2L(C_.Wvel_e, nz, n); wv[nz] = wv[nz] * A2L(C_.Wvel_e, nz, n); }
    for (int nz = ul1; nz <= nl1; nz++) {
      V2(Ur, 1, nz, n) = -(wu[nz] - wu[nz + 1]) / (3.0 * A2(C_.hnode, nz, n));
      V2(Ur, 2, nz, n) = -(wv[nz] - wv[nz + 1]) / (3.0 * A2(C_.hnode, nz, n));
    }
    for (int nz = nl1 + 1; nz <= nl - 1; nz++) { V2(Ur, 1, nz, n) = 0.0; V2(Ur, 2, nz, n) = 0.0; }
    for (int nz = 1; nz <= ul1 - 1; nz++) { V2(Ur, 1, nz, n) = 0.0; V2(Ur, 2, nz, n) = 0.0; }
  }
  for (int ed = 1; ed <= C_.m.myDim_edge2D; ed++) {
    int nod[2] = {EDG(1, ed), EDG(2, ed)};
    int el1 = ETRI(1, ed), el2 = ETRI(2, ed);
    int nl1 = NLEV(el1) - 1, ul1 = ULEV(el1);
    for (int nz = ul1; nz <= nl1; nz++) un1[nz] = V2(C_.UV, 2, nz, el1) * ECD(1, ed) - V2(C_.UV, 1, nz, el1) * ECD(2, ed);
    if (el2 > 0) {
      int nl2 = NLEV(el2) - 1, ul2 = ULEV(el2);
      for (int nz = ul2; nz <= nl2; nz++) un2[nz] = -V2(C_.UV, 2, nz, el2) * ECD(3, ed) + V2(C_.UV, 1, nz, el2) * ECD(4, ed);
      int mx = nl1 > nl2 ? nl1 : nl2, mn = ul1 < ul2 ? ul1 : ul2;
      for (int nz = nl1 + 1; nz <= mx; nz++) un1[nz] = 0.0;
      for (int nz = nl2 + 1; nz <= mx; nz++) un2[nz] = 0.0;
      for (int nz = 1; nz <= ul1 - 1; nz++) un1[nz] = 0.0;
      for (int nz = 1; nz <= ul2 - 1; nz++) un2[nz] = 0.0;
      if (nod[0] <= C_.m.myDim_nod2D)
        for (int nz = mn; nz <= mx; nz++) {
          V2(Ur, 1, nz, nod[0]) = V2(Ur, 1, nz, nod[0]) + un1[nz] * V2(C_.UV, 1, nz, el1) + un2[nz] * V2(C_.UV, 1, nz, el2);
          V2(Ur, 2, nz, nod[0]) = V2(Ur, 2, nz, nod[0]) + un1[nz] * V2(C_.UV, 2, nz, el1) + un2[nz] * V2(C_.UV, 2, nz, el2);
        }
      if (nod[1] <= C_.m.myDim_nod2D)
        for (int nz = mn; nz <= mx; nz++) {
          V2(Ur, 1, nz, nod[1]) = V2(Ur, 1, nz, nod[1]) - un1[nz] * V2(C_.UV, 1, nz, el1) - un2[nz] * V2(C_.UV, 1, nz, el2);
          V2(Ur, 2, nz, nod[1]) = V2(Ur, 2, nz, nod[1]) - un1[nz] * V2(C_.UV, 2, nz, el1) - un2[nz] * V2(C_.UV, 2, nz, el2);
        }
    } else {
      if (nod[0] <= C_.m.myDim_nod2D)
        for (int nz = ul1; nz <= nl1; nz++) {
          V2(Ur, 1, nz, nod[0]) = V2(Ur, 1, nz, nod[0]) + un1[nz] * V2(C_.UV, 1, nz, el1);
          V2(Ur, 2, nz, nod[0]) = V2(Ur, 2, nz, nod[0]) + un1[nz] * V2(C_.UV, 2, nz, el1);
        }
      if (nod[1] <= C_.m.myDim_nod2D)
        for (int nz = ul1; nz <= nl1; nz++) {
          V2(Ur, 1, nz, nod[1]) = V2(Ur, 1, nz, nod[1]) - un1[nz] * V2(C_.UV, 1, nz, el1);
          V2(Ur, 2, nz, nod[1]) = V2(Ur, 2, nz, nod[1]) - un1[nz] * V2(C_.UV, 2, nz, el1);
        }
    }
  }
  for (int n = 1; n <= C_.m.myDim_nod2D; n++)
    for (int nz = ULEVN(n); nz <= NLEVN(n) - 1; nz++) {
      V2(Ur, 1, nz, n) = V2(Ur, 1, nz, n) * AREASVOL_INV(nz, n);
      V2(Ur, 2, nz, n) = V2(Ur, 2, nz, n) * AREASVOL_INV(nz, n);
    }
  for (int el = 1; el <= C_.m.myDim_elem2D; el++) {
    double ar = C_.m.elem_area[el - 1];
    int n1 = EN(1, el), n2 = EN(2, el), n3 = EN(3, el);
    for (int nz = ULEV(el); nz <= NLEV(el) - 1; nz++)
      for (int c = 1; c <= 2; c++)
        V2(C_.UV_rhsAB, c, nz, el) = V2(C_.UV_rhsAB, c, nz, el) + ar * (V2(Ur, c, nz, n1) + V2(Ur, c, nz, n2) + V2(Ur, c, nz, n3)) / 3.0;
  }
  free(wu);
}

/* compute_vel_rhs: src/oce_ale_vel_rhs.F90:13-148 (no ice loading, no air pressure, no tides) */
static void relative_vorticity(void);
/* compute_vel_rhs_vinv (mom_adv = 3, vector-invariant form): src/oce_vel_rhs_vinv.F90:104-322.  The reference's vertical term is multiplied by
 * w = 0 (:122, the lines that would set w are commented out), i.e. uvert = 0 and UV_rhsAB - uvert * area = UV_rhsAB: left out.  hpressure exists
 * only with the linear free surface (pressure_bv, oce_ale_pressure_bv.F90:262): the scheme is meaningful for which_ALE = 'linfs' only. */
static void compute_vel_rhs_vinv(void) {
  const double eps = C_.p.epsilon, d0inv = 1. / DENSITY_0;
  double *KE = C_.KE_node;
  memset(KE, 0, sizeof(double) * (size_t)NLM1 * C_.N);
  for (int e = 1; e <= C_.m.myDim_elem2D; e++)
    for (int j = 1; j <= 3; j++)
      for (int nz = ULEV(e); nz <= NLEV(e) - 1; nz++)
        A2(KE, nz, EN(j, e)) = A2(KE, nz, EN(j, e)) + (V2(C_.UV, 1, nz, e) * V2(C_.UV, 1, nz, e) + V2(C_.UV, 2, nz, e) * V2(C_.UV, 2, nz, e)) * C_.m.elem_area[e - 1];
  for (int n = 1; n <= C_.m.myDim_nod2D; n++)
    for (int nz = ULEVN(n); nz <= NLEVN(n) - 1; nz++) A2(KE, nz, n) = A2(KE, nz, n) / (6. * AREASVOL(nz, n));
  for (int ed = 1; ed <= C_.m.myDim_edge2D; ed++)            /* zero at lateral walls */
    if (C_.m.myList_edge2D[ed - 1] > C_.m.edge2D_in)
      for (int k = 1; k <= 2; k++) for (int nz = 1; nz <= NLM1; nz++) A2(KE, nz, EDG(k, ed)) = 0.0;
  /* (exchange_nod(KE_node): single partition) */
  for (int e = 1; e <= C_.m.myDim_elem2D; e++)
    for (int nz = ULEV(e); nz <= NLEV(e) - 1; nz++) {
      V2(C_.UV_rhs, 1, nz, e) = -(0.5 + eps) * V2(C_.UV_rhsAB, 1, nz, e);
      V2(C_.UV_rhs, 2, nz, e) = -(0.5 + eps) * V2(C_.UV_rhsAB, 2, nz, e);
    }
  relative_vorticity();
  for (int e = 1; e <= C_.m.myDim_elem2D; e++) {
    const int en[3] = {EN(1, e), EN(2, e), EN(3, e)};
    double eta[3], ff[3];
    for (int k = 0; k < 3; k++) { eta[k] = G_ACC * C_.eta_n[en[k] - 1]; ff[k] = C_.m.coriolis_node[en[k] - 1]; }
    const double gg = C_.m.elem_area[e - 1];
    for (int nz = ULEV(e); nz <= NLEV(e) - 1; nz++) {
      double pre[3];
      for (int k = 0; k < 3; k++) pre[k] = -(eta[k] + A2L(C_.hpressure, nz, en[k]) * d0inv);
      double Fx = (GS(1, e) * pre[0] + GS(2, e) * pre[1]) + GS(3, e) * pre[2], Fy = (GS(4, e) * pre[0] + GS(5, e) * pre[1]) + GS(6, e) * pre[2];
      V2(C_.UV_rhs, 1, nz, e) = V2(C_.UV_rhs, 1, nz, e) + Fx * gg;
      V2(C_.UV_rhs, 2, nz, e) = V2(C_.UV_rhs, 2, nz, e) + Fy * gg;
      for (int k = 0; k < 3; k++) pre[k] = -A2(KE, nz, en[k]);
      Fx = (GS(1, e) * pre[0] + GS(2, e) * pre[1]) + GS(3, e) * pre[2]; Fy = (GS(4, e) * pre[0] + GS(5, e) * pre[1]) + GS(6, e) * pre[2];
      const double sfv = ((ff[0] + A2(C_.vorticity, nz, en[0])) + (ff[1] + A2(C_.vorticity, nz, en[1]))) + (ff[2] + A2(C_.vorticity, nz, en[2]));
      const double da = V2(C_.UV, 2, nz, e) * sfv / 3.0, db = -V2(C_.UV, 1, nz, e) * sfv / 3.0;
      V2(C_.UV_rhsAB, 1, nz, e) = (da + Fx) * gg;
      V2(C_.UV_rhsAB, 2, nz, e) = (db + Fy) * gg;
    }
  }
  double g2 = 1.5 + eps;
  if (!C_.first_step_done) { g2 = 1.0; C_.first_step_done = 1; }
  for (int e = 1; e <= C_.m.myDim_elem2D; e++) {
    const double ai = C_.p.dt / C_.m.elem_area[e - 1];
    for (int nz = ULEV(e); nz <= NLEV(e) - 1; nz++) {
      V2(C_.UV_rhs, 1, nz, e) = (V2(C_.UV_rhs, 1, nz, e) + V2(C_.UV_rhsAB, 1, nz, e) * g2) * ai;
      V2(C_.UV_rhs, 2, nz, e) = (V2(C_.UV_rhs, 2, nz, e) + V2(C_.UV_rhsAB, 2, nz, e) * g2) * ai;
    }
  }
}

void orc_compute_vel_rhs(void) {
  if (C_.p.mom_adv == 3) { compute_vel_rhs_vinv(); return; }
  double eps = C_.p.epsilon;
  for (int e = 1; e <= C_.m.myDim_elem2D; e++) {
    int nzmax = NLEV(e), nzmin = ULEV(e);
    for (int nz = nzmin; nz <= nzmax - 1; nz++) {
      V2(C_.UV_rhs, 1, nz, e) = -(0.5 + eps) * V2(C_.UV_rhsAB, 1, nz, e);
      V2(C_.UV_rhs, 2, nz, e) = -(0.5 + eps) * V2(C_.UV_rhsAB, 2, nz, e);
    }
    double pre[3];
    for (int k = 0; k < 3; k++) {                       /* surface potentials, src/oce_ale_vel_rhs.F90:52-76 */
      const int n = EN(k + 1, e) - 1;
      double p_eta = G_ACC * C_.eta_n[n], p_ice = 0.0, p_air = 0.0;
      if (C_.p.use_floatice) { p_ice = (C_.m_ice[n] * 910. + C_.m_snow[n] * 290.) * (1. / 1025.); p_ice = G_ACC * dmin(p_ice, C_.p.max_ice_loading); }
      if (C_.p.l_mslp) p_air = C_.press_air[n] / 1000;
      pre[k] = -(p_eta + p_ice + p_air);
      if (C_.p.use_global_tides) pre[k] = pre[k] - C_.ssh_gp[n];
    }
    double ff = C_.m.coriolis[e - 1] * C_.m.elem_area[e - 1];
    double Fx = GS(1, e) * pre[0] + GS(2, e) * pre[1] + GS(3, e) * pre[2];
    double Fy = GS(4, e) * pre[0] + GS(5, e) * pre[1] + GS(6, e) * pre[2];
    double ar = C_.m.elem_area[e - 1];
    for (int nz = nzmin; nz <= nzmax - 1; nz++) {
      V2(C_.UV_rhs, 1, nz, e) = V2(C_.UV_rhs, 1, nz, e) + (Fx - A2(C_.pgf_x, nz, e)) * ar;
      V2(C_.UV_rhs, 2, nz, e) = V2(C_.UV_rhs, 2, nz, e) + (Fy - A2(C_.pgf_y, nz, e)) * ar;
      V2(C_.UV_rhsAB, 1, nz, e) = V2(C_.UV, 2, nz, e) * ff;
      V2(C_.UV_rhsAB, 2, nz, e) = -V2(C_.UV, 1, nz, e) * ff;
    }
  }
  momentum_adv_scalar();
  double ff = 1.5 + eps;
  if (!C_.first_step_done) { ff = 1.0; C_.first_step_done = 1; }
  double dt = C_.p.dt;
  for (int e = 1; e <= C_.m.myDim_elem2D; e++) {
    double ar = C_.m.elem_area[e - 1];
    for (int nz = ULEV(e); nz <= NLEV(e) - 1; nz++) {
      V2(C_.UV_rhs, 1, nz, e) = dt * (V2(C_.UV_rhs, 1, nz, e) + V2(C_.UV_rhsAB, 1, nz, e) * ff) / ar;
      V2(C_.UV_rhs, 2, nz, e) = dt * (V2(C_.UV_rhs, 2, nz, e) + V2(C_.UV_rhsAB, 2, nz, e) * ff) / ar;
    }
  }
}

/* visc_filt_bcksct: src/oce_dyn.F90:563-649 */
void orc_visc_filt_bcksct(void) {
  double *Ub = C_.U_b, *Uc = C_.U_c;   /* (2,nl-1,E) and (2,nl-1,N): component 1 = U_*, 2 = V_* */
  memset(Ub, 0, sizeof(double) * 2 * NLM1 * C_.E);
  memset(Uc, 0, sizeof(double) * 2 * NLM1 * C_.N);
  double dt = C_.p.dt, g0 = C_.p.gamma0, g1 = C_.p.gamma1, g2 = C_.p.gamma2;
  for (int ed = 1; ed <= C_.D; ed++) {
    if (C_.m.myList_edge2D[ed - 1] > C_.m.edge2D_in) continue;
    int e1 = ETRI(1, ed), e2 = ETRI(2, ed);
    double a1 = C_.m.elem_area[e1 - 1], a2 = C_.m.elem_area[e2 - 1];
    double len = sqrt(a1 + a2);
    int nzmax = NLEV(e1) < NLEV(e2) ? NLEV(e1) : NLEV(e2);
    int nzmin = ULEV(e1) > ULEV(e2) ? ULEV(e1) : ULEV(e2);
    for (int nz = nzmin; nz <= nzmax - 1; nz++) {
      double u1 = V2(C_.UV, 1, nz, e1) - V2(C_.UV, 1, nz, e2);
      double v1 = V2(C_.UV, 2, nz, e1) - V2(C_.UV, 2, nz, e2);
      double vi = dt * dmax(g0, dmax(g1 * sqrt(u1 * u1 + v1 * v1), g2 * (u1 * u1 + v1 * v1))) * len;
      u1 = u1 * vi; v1 = v1 * vi;
      V2(Ub, 1, nz, e1) = V2(Ub, 1, nz, e1) - u1 / a1;
      V2(Ub, 1, nz, e2) = V2(Ub, 1, nz, e2) + u1 / a2;
      V2(Ub, 2, nz, e1) = V2(Ub, 2, nz, e1) - v1 / a1;
      V2(Ub, 2, nz, e2) = V2(Ub, 2, nz, e2) + v1 / a2;
    }
  }
  for (int n = 1; n <= C_.m.myDim_nod2D; n++)
    for (int nz = ULEVN(n); nz <= NLEVN(n) - 1; nz++) {
      double vi = 0.0, u1 = 0.0, v1 = 0.0;
      for (int k = 1; k <= C_.m.nod_in_elem2D_num[n - 1]; k++) {
        int e = NIE(k, n);
        double ar = C_.m.elem_area[e - 1];
        vi = vi + ar;
        u1 = u1 + V2(Ub, 1, nz, e) * ar;
        v1 = v1 + V2(Ub, 2, nz, e) * ar;
      }
      V2(Uc, 1, nz, n) = u1 / vi;
      V2(Uc, 2, nz, n) = v1 / vi;
    }
  double bs = C_.p.easy_bs_return;
  for (int e = 1; e <= C_.m.myDim_elem2D; e++) {
    int n1 = EN(1, e), n2 = EN(2, e), n3 = EN(3, e);
    for (int nz = ULEV(e); nz <= NLEV(e) - 1; nz++)
      for (int c = 1; c <= 2; c++)
        V2(C_.UV_rhs, c, nz, e) = V2(C_.UV_rhs, c, nz, e) + V2(Ub, c, nz, e) - bs * (V2(Uc, c, nz, n1) + V2(Uc, c, nz, n2) + V2(Uc, c, nz, n3)) / 3.0;
  }
}

/* visc_filt_bilapl (option 6, src/oce_dyn.F90:658-726) and visc_filt_bidiff (option 7, :734-801): both stages of the
 * biharmonic filter; U_b holds the reference's U_c / V_c */
static void visc_filt_biharmonic(int opt) {
  double *Uc = C_.U_b;
  memset(Uc, 0, sizeof(double) * 2 * NLM1 * C_.E);
  double dt = C_.p.dt, g0 = C_.p.gamma0, g1 = C_.p.gamma1, g2 = C_.p.gamma2;
  for (int ed = 1; ed <= C_.D; ed++) {
    if (C_.m.myList_edge2D[ed - 1] > C_.m.edge2D_in) continue;
    int e1 = ETRI(1, ed), e2 = ETRI(2, ed);
    double len = sqrt(C_.m.elem_area[e1 - 1] + C_.m.elem_area[e2 - 1]);
    int nzmax = NLEV(e1) < NLEV(e2) ? NLEV(e1) : NLEV(e2);
    int nzmin = ULEV(e1) > ULEV(e2) ? ULEV(e1) : ULEV(e2);
    for (int nz = nzmin; nz <= nzmax - 1; nz++) {
      double u1 = V2(C_.UV, 1, nz, e1) - V2(C_.UV, 1, nz, e2);
      double v1 = V2(C_.UV, 2, nz, e1) - V2(C_.UV, 2, nz, e2);
      if (opt == 7) {
        double vi = u1 * u1 + v1 * v1;
        vi = sqrt(dmax(g0, dmax(g1 * sqrt(vi), g2 * vi)) * len);
        u1 = u1 * vi; v1 = v1 * vi;
      }
      V2(Uc, 1, nz, e1) = V2(Uc, 1, nz, e1) - u1;
      V2(Uc, 1, nz, e2) = V2(Uc, 1, nz, e2) + u1;
      V2(Uc, 2, nz, e1) = V2(Uc, 2, nz, e1) - v1;
      V2(Uc, 2, nz, e2) = V2(Uc, 2, nz, e2) + v1;
    }
  }
  if (opt == 6)
    for (int e = 1; e <= C_.m.myDim_elem2D; e++) {
      double len = sqrt(C_.m.elem_area[e - 1]);
      for (int nz = ULEV(e); nz <= NLEV(e) - 1; nz++) {
        double u1 = V2(Uc, 1, nz, e) * V2(Uc, 1, nz, e) + V2(Uc, 2, nz, e) * V2(Uc, 2, nz, e);
        double vi = dmax(g0, dmax(g1 * sqrt(u1), g2 * u1)) * len * dt;
        V2(Uc, 1, nz, e) = -V2(Uc, 1, nz, e) * vi;
        V2(Uc, 2, nz, e) = -V2(Uc, 2, nz, e) * vi;
      }
    }
  if (opt == 4)                                  /* visc_filt_biharm(option = 1), src/oce_dyn.F90:314-331 */
    for (int e = 1; e <= C_.m.myDim_elem2D; e++) {
      double len = sqrt(C_.m.elem_area[e - 1]);
      for (int nz = ULEV(e); nz <= NLEV(e) - 1; nz++) {
        double vi = dmax(g0, g1 * sqrt(V2(C_.UV, 1, nz, e) * V2(C_.UV, 1, nz, e) + V2(C_.UV, 2, nz, e) * V2(C_.UV, 2, nz, e))) * len * dt;
        V2(Uc, 1, nz, e) = -V2(Uc, 1, nz, e) * vi;
        V2(Uc, 2, nz, e) = -V2(Uc, 2, nz, e) * vi;
      }
    }
  /* (exchange_elem(U_c), exchange_elem(V_c): single partition) */
  for (int ed = 1; ed <= C_.D; ed++) {
    if (C_.m.myList_edge2D[ed - 1] > C_.m.edge2D_in) continue;
    int e1 = ETRI(1, ed), e2 = ETRI(2, ed);
    double a1 = C_.m.elem_area[e1 - 1], a2 = C_.m.elem_area[e2 - 1];
    double len = sqrt(a1 + a2);
    int nzmax = NLEV(e1) < NLEV(e2) ? NLEV(e1) : NLEV(e2);
    int nzmin = ULEV(e1) > ULEV(e2) ? ULEV(e1) : ULEV(e2);
    for (int nz = nzmin; nz <= nzmax - 1; nz++) {
      double u1 = V2(Uc, 1, nz, e1) - V2(Uc, 1, nz, e2);
      double v1 = V2(Uc, 2, nz, e1) - V2(Uc, 2, nz, e2);
      if (opt == 7) {
        double du = V2(C_.UV, 1, nz, e1) - V2(C_.UV, 1, nz, e2);
        double dv = V2(C_.UV, 2, nz, e1) - V2(C_.UV, 2, nz, e2);
        double vi = du * du + dv * dv;
        vi = -dt * sqrt(dmax(g0, dmax(g1 * sqrt(vi), g2 * vi)) * len);
        u1 = vi * u1; v1 = vi * v1;
      }
      V2(C_.UV_rhs, 1, nz, e1) = V2(C_.UV_rhs, 1, nz, e1) - u1 / a1;
      V2(C_.UV_rhs, 1, nz, e2) = V2(C_.UV_rhs, 1, nz, e2) + u1 / a2;
      V2(C_.UV_rhs, 2, nz, e1) = V2(C_.UV_rhs, 2, nz, e1) - v1 / a1;
      V2(C_.UV_rhs, 2, nz, e2) = V2(C_.UV_rhs, 2, nz, e2) + v1 / a2;
    }
  }
}

/* Kv0_background_qiang(Kv0_b, geo_coord_nod2D(2,node)/rad, abs(zbar_3d_n(nz,node))): src/oce_ale_mixing_pp.F90:91-125, callers
 * :73-76 and oce_ale_mixing_kpp.F90:821-822 (Kv0_const=.false.) */
double orc_kv0_background_qiang(int n, int nz) {
  const double rad = 3.14159265358979 / 180.0;
  double lat = C_.m.geo_coord_nod2D[2 * (n - 1) + 1] / rad, dep = fabs(A2L(C_.zbar_3d_n, nz, n));
  double aux = (0.6 + 1.0598 / 3.1415926 * atan(4.5e-3 * (dep - 2500.0))) * 1.0e-5, ratio;
  if (fabs(lat) < 5.0) ratio = 1.0;
  else ratio = fmin(1.0 + 9.0 * (fabs(lat) - 5.0) / 10.0, 10.0);
  if (lat > 70.0) {
    if (dep <= 50.0) ratio = 4.0 + 6.0 * (50.0 - dep) / 50.0;
    else ratio = 4.0;
  }
  return aux * ratio;
}

/* relative_vorticity: src/oce_vel_rhs_vinv.F90:14-102 (circulation around the scalar control volumes / areasvol) */
static void relative_vorticity(void) {
  double *vo = C_.vorticity;
  for (int n = 1; n <= C_.m.myDim_nod2D; n++) for (int nz = 1; nz <= NLM1; nz++) A2(vo, nz, n) = 0.0;
  for (int ed = 1; ed <= C_.m.myDim_edge2D; ed++) {
    int n1 = EDG(1, ed), n2 = EDG(2, ed), e1 = ETRI(1, ed), e2 = ETRI(2, ed);
    int nl1 = NLEV(e1) - 1, ul1 = ULEV(e1), nl2 = 0, ul2 = 0;
    double dX1 = ECD(1, ed), dY1 = ECD(2, ed), dX2 = 0, dY2 = 0;
    if (e2 > 0) { dX2 = ECD(3, ed); dY2 = ECD(4, ed); nl2 = NLEV(e2) - 1; ul2 = ULEV(e2); }
    int nl12 = nl1 < nl2 ? nl1 : nl2, ul12 = ul1 > ul2 ? ul1 : ul2;
#define VADD(c1) { A2(vo, nz, n1) = A2(vo, nz, n1) + (c1); A2(vo, nz, n2) = A2(vo, nz, n2) - (c1); }
    for (int nz = ul1; nz <= ul12 - 1; nz++) VADD(dX1 * V2(C_.UV, 1, nz, e1) + dY1 * V2(C_.UV, 2, nz, e1));
    if (ul2 > 0) for (int nz = ul2; nz <= ul12 - 1; nz++) VADD(-dX2 * V2(C_.UV, 1, nz, e2) - dY2 * V2(C_.UV, 2, nz, e2));
    for (int nz = ul12; nz <= nl12; nz++) VADD(dX1 * V2(C_.UV, 1, nz, e1) + dY1 * V2(C_.UV, 2, nz, e1) - dX2 * V2(C_.UV, 1, nz, e2) - dY2 * V2(C_.UV, 2, nz, e2));
    for (int nz = nl12 + 1; nz <= nl1; nz++) VADD(dX1 * V2(C_.UV, 1, nz, e1) + dY1 * V2(C_.UV, 2, nz, e1));
    for (int nz = nl12 + 1; nz <= nl2; nz++) VADD(-dX2 * V2(C_.UV, 1, nz, e2) - dY2 * V2(C_.UV, 2, nz, e2));
#undef VADD
  }
  for (int n = 1; n <= C_.m.myDim_nod2D; n++)
    for (int nz = ULEVN(n); nz <= NLEVN(n) - 1; nz++) A2(vo, nz, n) = A2(vo, nz, n) / AREASVOL(nz, n);
  /* (exchange_nod(vorticity): single partition) */
}

/* h_viscosity_leith: src/oce_dyn.F90:461-561 (Leith + modified Leith coefficient, two rounds of node / element averaging) */
static void h_viscosity_leith(void) {
  const int nl = NL;
  double *Visc = C_.Visc, *aux = C_.leith_aux, *zbar_n = calloc((size_t)nl + 2, sizeof(double));
  const double dt = C_.p.dt, g1 = C_.p.gamma1, Div_c = C_.p.Div_c, Leith_c = C_.p.Leith_c;
  relative_vorticity();                                   /* mom_adv < 4 */
  memset(Visc, 0, sizeof(double) * (size_t)NLM1 * C_.E);
  for (int e = 1; e <= C_.m.myDim_elem2D; e++) {
    int nl1 = NLEV(e) - 1, ul1 = ULEV(e), en[3] = {EN(1, e), EN(2, e), EN(3, e)};
    for (int k = 0; k <= nl + 1; k++) zbar_n[k] = 0.0;
    zbar_n[nl1 + 1] = C_.m.zbar_e_bot[e - 1];
    for (int nz = nl1; nz >= ul1 + 1; nz--) zbar_n[nz] = zbar_n[nz + 1] + A2(C_.helem, nz, e);
    zbar_n[ul1] = zbar_n[ul1 + 1] + A2(C_.helem, ul1, e);
    const double ar = C_.m.elem_area[e - 1];
    for (int nz = ul1; nz <= nl1; nz++) {
      double dz = zbar_n[nz] - zbar_n[nz + 1], d3[3], v3[3];
      for (int j = 0; j < 3; j++) { d3[j] = (A2L(C_.Wvel, nz, en[j]) - A2L(C_.Wvel, nz + 1, en[j])) / dz; v3[j] = A2(C_.vorticity, nz, en[j]); }
      double xe = (GS(1, e) * d3[0] + GS(2, e) * d3[1]) + GS(3, e) * d3[2], ye = (GS(4, e) * d3[0] + GS(5, e) * d3[1]) + GS(6, e) * d3[2];
      double lx = (GS(1, e) * v3[0] + GS(2, e) * v3[1]) + GS(3, e) * v3[2], ly = (GS(4, e) * v3[0] + GS(5, e) * v3[1]) + GS(6, e) * v3[2];
      A2(Visc, nz, e) = dmin(g1 * ar * sqrt((Div_c * (xe * xe + ye * ye) + Leith_c * (lx * lx + ly * ly)) * ar), ar / dt);
    }
  }
  memset(aux, 0, sizeof(double) * (size_t)NLM1 * C_.N);
  for (int nt = 1; nt <= 2; nt++) {
    for (int n = 1; n <= C_.m.myDim_nod2D; n++)
      for (int nz = ULEVN(n); nz <= NLEVN(n) - 1; nz++) {
        double dz = 0.0, vi = 0.0;
        for (int k = 1; k <= C_.m.nod_in_elem2D_num[n - 1]; k++) {
          int el = NIE(k, n);
          dz = dz + C_.m.elem_area[el - 1];
          vi = vi + A2(Visc, nz, el) * C_.m.elem_area[el - 1];
        }
        A2(aux, nz, n) = vi / dz;
      }
    /* (exchange_nod(aux): single partition) */
    for (int e = 1; e <= C_.m.myDim_elem2D; e++) {
      int nl1 = NLEV(e) - 1, ul1 = ULEV(e), n1 = EN(1, e), n2 = EN(2, e), n3 = EN(3, e);
      for (int nz = 1; nz <= NLM1; nz++) A2(Visc, nz, e) = (nz >= ul1 && nz <= nl1) ? ((A2(aux, nz, n1) + A2(aux, nz, n2)) + A2(aux, nz, n3)) / 3.0 : 0.0;
    }
  }
  /* (exchange_elem(Visc): single partition) */
  free(zbar_n);
}

/* visc_filt_harmon (option 1, :236-273) and the harmonic Leith part of visc_filt_hbhmix (option 2, :376-458), which also leaves the
 * first stage of its biharmonic background in U_c; visc_filt_biharm(2) (option 3, :275-372) is the two-stage filter of option 4 with
 * the Leith coefficient */
static void visc_filt_leith(int opt) {
  double *Uc = C_.U_b, *Visc = C_.Visc;
  const double dt = C_.p.dt, g0 = C_.p.gamma0;
  memset(Uc, 0, sizeof(double) * 2 * NLM1 * C_.E);
  if (opt == 1 || opt == 2)
    for (int ed = 1; ed <= C_.D; ed++) {
      if (C_.m.myList_edge2D[ed - 1] > C_.m.edge2D_in) continue;
      int e1 = ETRI(1, ed), e2 = ETRI(2, ed);
      double a1 = C_.m.elem_area[e1 - 1], a2 = C_.m.elem_area[e2 - 1], len = sqrt(a1 + a2);
      int nzmax = NLEV(e1) < NLEV(e2) ? NLEV(e1) : NLEV(e2), nzmin = ULEV(e1) > ULEV(e2) ? ULEV(e1) : ULEV(e2);
      for (int nz = nzmin; nz <= nzmax - 1; nz++) {
        double u1 = V2(C_.UV, 1, nz, e1) - V2(C_.UV, 1, nz, e2), v1 = V2(C_.UV, 2, nz, e1) - V2(C_.UV, 2, nz, e2), vi;
        if (opt == 1) { vi = 0.5 * (A2(Visc, nz, e1) + A2(Visc, nz, e2)); vi = dmax(vi, g0 * len) * dt; }
        else {
          vi = dt * 0.5 * (A2(Visc, nz, e1) + A2(Visc, nz, e2));
          V2(Uc, 1, nz, e1) = V2(Uc, 1, nz, e1) - u1; V2(Uc, 1, nz, e2) = V2(Uc, 1, nz, e2) + u1;
          V2(Uc, 2, nz, e1) = V2(Uc, 2, nz, e1) - v1; V2(Uc, 2, nz, e2) = V2(Uc, 2, nz, e2) + v1;
        }
        u1 = u1 * vi; v1 = v1 * vi;
        V2(C_.UV_rhs, 1, nz, e1) = V2(C_.UV_rhs, 1, nz, e1) - u1 / a1; V2(C_.UV_rhs, 1, nz, e2) = V2(C_.UV_rhs, 1, nz, e2) + u1 / a2;
        V2(C_.UV_rhs, 2, nz, e1) = V2(C_.UV_rhs, 2, nz, e1) - v1 / a1; V2(C_.UV_rhs, 2, nz, e2) = V2(C_.UV_rhs, 2, nz, e2) + v1 / a2;
      }
    }
  if (opt == 1) return;
  if (opt == 3)                                           /* first stage of visc_filt_biharm */
    for (int ed = 1; ed <= C_.D; ed++) {
      if (C_.m.myList_edge2D[ed - 1] > C_.m.edge2D_in) continue;
      int e1 = ETRI(1, ed), e2 = ETRI(2, ed);
      int nzmax = NLEV(e1) < NLEV(e2) ? NLEV(e1) : NLEV(e2), nzmin = ULEV(e1) > ULEV(e2) ? ULEV(e1) : ULEV(e2);
      for (int nz = nzmin; nz <= nzmax - 1; nz++) {
        double u1 = V2(C_.UV, 1, nz, e1) - V2(C_.UV, 1, nz, e2), v1 = V2(C_.UV, 2, nz, e1) - V2(C_.UV, 2, nz, e2);
        V2(Uc, 1, nz, e1) = V2(Uc, 1, nz, e1) - u1; V2(Uc, 1, nz, e2) = V2(Uc, 1, nz, e2) + u1;
        V2(Uc, 2, nz, e1) = V2(Uc, 2, nz, e1) - v1; V2(Uc, 2, nz, e2) = V2(Uc, 2, nz, e2) + v1;
      }
    }
  for (int e = 1; e <= C_.m.myDim_elem2D; e++) {
    double len = sqrt(C_.m.elem_area[e - 1]);
    for (int nz = ULEV(e); nz <= NLEV(e) - 1; nz++) {
      double vi = (opt == 2) ? dt * g0 * len : dmax(A2(Visc, nz, e), g0 * len) * dt;
      V2(Uc, 1, nz, e) = -V2(Uc, 1, nz, e) * vi;
      V2(Uc, 2, nz, e) = -V2(Uc, 2, nz, e) * vi;
    }
  }
  /* (exchange_elem(U_c), exchange_elem(V_c): single partition) */
  for (int ed = 1; ed <= C_.D; ed++) {
    if (C_.m.myList_edge2D[ed - 1] > C_.m.edge2D_in) continue;
    int e1 = ETRI(1, ed), e2 = ETRI(2, ed);
    double a1 = C_.m.elem_area[e1 - 1], a2 = C_.m.elem_area[e2 - 1];
    int nzmax = NLEV(e1) < NLEV(e2) ? NLEV(e1) : NLEV(e2), nzmin = ULEV(e1) > ULEV(e2) ? ULEV(e1) : ULEV(e2);
    for (int nz = nzmin; nz <= nzmax - 1; nz++) {
      double u1 = V2(Uc, 1, nz, e1) - V2(Uc, 1, nz, e2), v1 = V2(Uc, 2, nz, e1) - V2(Uc, 2, nz, e2);
      V2(C_.UV_rhs, 1, nz, e1) = V2(C_.UV_rhs, 1, nz, e1) - u1 / a1; V2(C_.UV_rhs, 1, nz, e2) = V2(C_.UV_rhs, 1, nz, e2) + u1 / a2;
      V2(C_.UV_rhs, 2, nz, e1) = V2(C_.UV_rhs, 2, nz, e1) - v1 / a1; V2(C_.UV_rhs, 2, nz, e2) = V2(C_.UV_rhs, 2, nz, e2) + v1 / a2;
    }
  }
}

/* ---- visc_option = 8: backscatter_coef (src/oce_dyn.F90:967-996), visc_filt_dbcksc (:806-964), uke_update (:999-1152).  Single partition: the
 * exchange_* calls of the reference are no-ops.  which_toy = 'soufflet' as in the shipped namelist.config: the branch without the hard-coded
 * regional mask (:1122-1125). */
/* smooth_elem2D (src/gen_support.F90:183-212) applied to one level of an element field with `nc` interleaved components: N rounds of
 * element -> node (area-weighted mean over the whole cluster, dry cells included) -> element (mean of the three nodes) */
static void smooth_elem_level(double *arr, int nc, int comp, int nz, int nrounds, double *work) {
  for (int q = 0; q < nrounds; q++) {
    for (int n = 1; n <= C_.m.myDim_nod2D; n++) {
      double vol = 0., w = 0.;
      for (int j = 1; j <= C_.m.nod_in_elem2D_num[n - 1]; j++) {
        int e = NIE(j, n);
        w = w + arr[((size_t)(e - 1) * NLM1 + (nz - 1)) * nc + comp] * C_.m.elem_area[e - 1];
        vol = vol + C_.m.elem_area[e - 1];
      }
      work[n - 1] = w / vol;
    }
    for (int e = 1; e <= C_.m.myDim_elem2D; e++)
      arr[((size_t)(e - 1) * NLM1 + (nz - 1)) * nc + comp] = ((work[EN(1, e) - 1] + work[EN(2, e) - 1]) + work[EN(3, e) - 1]) / 3.0;
  }
}
static void backscatter_coef(void) {
  double dt = C_.p.dt;
  memset(C_.v_back, 0, sizeof(double) * NLM1 * C_.E);
  for (int e = 1; e <= C_.m.myDim_elem2D; e++) {
    double ar = C_.m.elem_area[e - 1];
    for (int nz = 1; nz <= NLEV(e) - 1; nz++)
      A2(C_.v_back, nz, e) = dmin(-C_.p.c_back * sqrt(ar) * sqrt(dmax(2.0 * A2(C_.uke, nz, e), 0.0)), 0.2 * ar / dt);
  }
}
static void uke_update(double *work) {
  const double c_min = 0.5, f_min = 1.e-6, r_max = 200000., pi = 3.14159265358979;
  int E = C_.m.myDim_elem2D, N = C_.m.myDim_nod2D;
  memset(C_.uke_back, 0, sizeof(double) * NLM1 * C_.E); memset(C_.uke_dis, 0, sizeof(double) * NLM1 * C_.E);
  for (int e = 1; e <= E; e++)
    for (int nz = 1; nz <= NLEV(e) - 1; nz++) {
      A2(C_.uke_dis, nz, e) = (V2(C_.UV, 1, nz, e) * V2(C_.UV_dis_tend, 1, nz, e) + V2(C_.UV, 2, nz, e) * V2(C_.UV_dis_tend, 2, nz, e));
      A2(C_.uke_back, nz, e) = (V2(C_.UV, 1, nz, e) * V2(C_.UV_back_tend, 1, nz, e) + V2(C_.UV, 2, nz, e) * V2(C_.UV_back_tend, 2, nz, e));
    }
  for (int nz = 1; nz <= NLM1; nz++) smooth_elem_level(C_.uke_back, 1, 0, nz, C_.p.smooth_back, work);
  /* U_work, V_work: area-weighted node means of UV over the whole cluster; V_work = U_work / vol (the reference's line :1066, kept) */
  double *Uw = calloc((size_t)2 * NLM1 * C_.N, sizeof(double)), *Vw = Uw + (size_t)NLM1 * C_.N;
  double *rosb = calloc((size_t)NLM1 * C_.E, sizeof(double));
  for (int nz = 1; nz <= NLM1; nz++)
    for (int n = 1; n <= N; n++) {
      double vol = 0., u = 0., v = 0.;
      for (int j = 1; j <= C_.m.nod_in_elem2D_num[n - 1]; j++) {
        int e = NIE(j, n);
        u = u + V2(C_.UV, 1, nz, e) * C_.m.elem_area[e - 1];
        v = v + V2(C_.UV, 2, nz, e) * C_.m.elem_area[e - 1];
        vol = vol + C_.m.elem_area[e - 1];
      }
      (void)v;
      u = u / vol;
      A2(Uw, nz, n) = u; A2(Vw, nz, n) = u / vol;
    }
  for (int e = 1; e <= E; e++) {
    int en[3] = {EN(1, e), EN(2, e), EN(3, e)};
    for (int nz = 1; nz <= NLEV(e) - 1; nz++) {
      double gu = (GS(1, e) * A2(Uw, nz, en[0]) + GS(2, e) * A2(Uw, nz, en[1])) + GS(3, e) * A2(Uw, nz, en[2]);
      double hv = (GS(4, e) * A2(Vw, nz, en[0]) + GS(5, e) * A2(Vw, nz, en[1])) + GS(6, e) * A2(Vw, nz, en[2]);
      double hu = (GS(4, e) * A2(Uw, nz, en[0]) + GS(5, e) * A2(Uw, nz, en[1])) + GS(6, e) * A2(Uw, nz, en[2]);
      double gv = (GS(1, e) * A2(Vw, nz, en[0]) + GS(2, e) * A2(Vw, nz, en[1])) + GS(3, e) * A2(Vw, nz, en[2]);
      A2(rosb, nz, e) = sqrt((gu - hv) * (gu - hv) + (hu + gv) * (hu + gv));
    }
  }
  for (int e = 1; e <= E; e++) {
    double scaling = 1.0;
    int en[3] = {EN(1, e), EN(2, e), EN(3, e)};
    if (C_.p.uke_scaling) {
      double reso = sqrt(C_.m.elem_area[e - 1] * 4.0 / sqrt(3.0)), rb = 0.0;
      for (int kk = 0; kk < 3; kk++) {
        int n = en[kk];
        double c1 = 0.0;
        int nzmax = C_.m.nlevels_nod2D_min[n - 1];            /* minval(nlevels(nod_in_elem2D(1:num, n))) */
        for (int nz = 1; nz <= nzmax - 1; nz++)
          c1 = c1 + A2(C_.hnode_new, nz, n) * (sqrt(dmax(A2L(C_.bvfreq, nz, n), 0.0)) + sqrt(dmax(A2L(C_.bvfreq, nz + 1, n), 0.0))) / 2.;
        c1 = dmax(c_min, c1 / pi);
        rb = rb + dmin(c1 / dmax(fabs(C_.m.coriolis_node[n - 1]), f_min), r_max);
      }
      rb = rb / 3.0;
      scaling = 1.0 / (1.0 + (C_.p.uke_scaling_factor * reso / rb));
    }
    for (int nz = 1; nz <= NLEV(e) - 1; nz++) {
      double fsum = (C_.m.coriolis_node[en[0] - 1] + C_.m.coriolis_node[en[1] - 1]) + C_.m.coriolis_node[en[2] - 1];
      A2(rosb, nz, e) = A2(rosb, nz, e) / dmax(fabs(fsum), f_min);
      A2(C_.uke_dis, nz, e) = scaling * 1.0 / (1.0 + A2(rosb, nz, e) / C_.p.rosb_dis) * A2(C_.uke_dis, nz, e);
    }
  }
  free(Uw); free(rosb);
  for (int nz = 1; nz <= NLM1; nz++) smooth_elem_level(C_.uke_dis, 1, 0, nz, C_.p.smooth_dis, work);
  for (int e = 1; e <= E; e++)
    for (int nz = 1; nz <= NLEV(e) - 1; nz++) {
      A2(C_.uke_rhs_old, nz, e) = A2(C_.uke_rhs, nz, e);
      A2(C_.uke_rhs, nz, e) = -A2(C_.uke_dis, nz, e) - A2(C_.uke_back, nz, e) + A2(C_.uke_dif, nz, e);
      A2(C_.uke, nz, e) = A2(C_.uke, nz, e) + 1.5 * A2(C_.uke_rhs, nz, e) - 0.5 * A2(C_.uke_rhs_old, nz, e);
    }
}
static void visc_filt_dbcksc(void) {
  double dt = C_.p.dt;
  size_t n2 = (size_t)2 * NLM1 * C_.E;
  double *Uc = C_.U_b, *back = calloc(n2, sizeof(double)), *dis = calloc(n2, sizeof(double)), *uked = C_.uke_dif;
  double *work = calloc(C_.N, sizeof(double));
  memset(Uc, 0, sizeof(double) * n2); memset(uked, 0, sizeof(double) * NLM1 * C_.E);
  for (int ed = 1; ed <= C_.D; ed++) {
    if (C_.m.myList_edge2D[ed - 1] > C_.m.edge2D_in) continue;
    int e1 = ETRI(1, ed), e2 = ETRI(2, ed);
    int nzmax = NLEV(e1) < NLEV(e2) ? NLEV(e1) : NLEV(e2);
    for (int nz = 1; nz <= nzmax - 1; nz++) {
      double u1 = V2(C_.UV, 1, nz, e1) - V2(C_.UV, 1, nz, e2), v1 = V2(C_.UV, 2, nz, e1) - V2(C_.UV, 2, nz, e2);
      V2(Uc, 1, nz, e1) = V2(Uc, 1, nz, e1) - u1; V2(Uc, 1, nz, e2) = V2(Uc, 1, nz, e2) + u1;
      V2(Uc, 2, nz, e1) = V2(Uc, 2, nz, e1) - v1; V2(Uc, 2, nz, e2) = V2(Uc, 2, nz, e2) + v1;
    }
  }
  for (int e = 1; e <= C_.m.myDim_elem2D; e++) {
    double len = sqrt(C_.m.elem_area[e - 1]);
    len = dt * len / 30.0;
    for (int nz = 1; nz <= NLEV(e) - 1; nz++) {
      double vi = dmax(0.2, sqrt(V2(C_.UV, 1, nz, e) * V2(C_.UV, 1, nz, e) + V2(C_.UV, 2, nz, e) * V2(C_.UV, 2, nz, e))) * len;
      V2(Uc, 1, nz, e) = -V2(Uc, 1, nz, e) * vi;
      V2(Uc, 2, nz, e) = -V2(Uc, 2, nz, e) * vi;
    }
  }
  for (int ed = 1; ed <= C_.D; ed++) {
    if (C_.m.myList_edge2D[ed - 1] > C_.m.edge2D_in) continue;
    int e1 = ETRI(1, ed), e2 = ETRI(2, ed);
    double a1 = C_.m.elem_area[e1 - 1], a2 = C_.m.elem_area[e2 - 1];
    double le1 = EDXY(1, ed) * (C_.m.elem_cos[e1 - 1] + C_.m.elem_cos[e2 - 1]) * 0.25, le2 = EDXY(2, ed);
    double len = sqrt(le1 * le1 + le2 * le2) * R_EARTH;
    le1 = ECD(1, ed) - ECD(3, ed); le2 = ECD(2, ed) - ECD(4, ed);
    double crosslen = sqrt(le1 * le1 + le2 * le2);
    int nzmax = NLEV(e1) < NLEV(e2) ? NLEV(e1) : NLEV(e2);
    for (int nz = 1; nz <= nzmax - 1; nz++) {
      double vi = dt * len * (A2(C_.v_back, nz, e1) + A2(C_.v_back, nz, e2)) / crosslen;
      double u1 = (V2(C_.UV, 1, nz, e1) - V2(C_.UV, 1, nz, e2)) * vi, v1 = (V2(C_.UV, 2, nz, e1) - V2(C_.UV, 2, nz, e2)) * vi;
      vi = dt * len * (C_.p.K_back * sqrt(a1 / C_.p.scale_area) + C_.p.K_back * sqrt(a2 / C_.p.scale_area)) / crosslen;
      double uke1 = (A2(C_.uke, nz, e1) - A2(C_.uke, nz, e2)) * vi;
      V2(back, 1, nz, e1) = V2(back, 1, nz, e1) - u1 / a1; V2(back, 1, nz, e2) = V2(back, 1, nz, e2) + u1 / a2;
      V2(back, 2, nz, e1) = V2(back, 2, nz, e1) - v1 / a1; V2(back, 2, nz, e2) = V2(back, 2, nz, e2) + v1 / a2;
      A2(uked, nz, e1) = A2(uked, nz, e1) - uke1 / a1; A2(uked, nz, e2) = A2(uked, nz, e2) + uke1 / a2;
      u1 = V2(Uc, 1, nz, e1) - V2(Uc, 1, nz, e2); v1 = V2(Uc, 2, nz, e1) - V2(Uc, 2, nz, e2);
      V2(dis, 1, nz, e1) = V2(dis, 1, nz, e1) - u1 / a1; V2(dis, 1, nz, e2) = V2(dis, 1, nz, e2) + u1 / a2;
      V2(dis, 2, nz, e1) = V2(dis, 2, nz, e1) - v1 / a1; V2(dis, 2, nz, e2) = V2(dis, 2, nz, e2) + v1 / a2;
    }
  }
  for (int nz = 1; nz <= NLM1; nz++) { smooth_elem_level(back, 2, 0, nz, C_.p.smooth_back_tend, work); smooth_elem_level(back, 2, 1, nz, C_.p.smooth_back_tend, work); }
  for (int e = 1; e <= C_.m.myDim_elem2D; e++)
    for (int nz = 1; nz <= NLEV(e) - 1; nz++) {
      V2(C_.UV_rhs, 1, nz, e) = V2(C_.UV_rhs, 1, nz, e) + V2(dis, 1, nz, e) + V2(back, 1, nz, e);
      V2(C_.UV_rhs, 2, nz, e) = V2(C_.UV_rhs, 2, nz, e) + V2(dis, 2, nz, e) + V2(back, 2, nz, e);
    }
  memcpy(C_.UV_dis_tend, dis, sizeof(double) * n2); memcpy(C_.UV_back_tend, back, sizeof(double) * n2);
  uke_update(work);
  free(back); free(dis); free(work);
}

/* viscosity_filter(visc_option): src/oce_dyn.F90:196-228 (options 1-8) */
void orc_viscosity_filter(void) {
  if (C_.p.visc_option == 8) { backscatter_coef(); visc_filt_dbcksc(); return; }
  if (C_.p.visc_option == 5) orc_visc_filt_bcksct();
  else if (C_.p.visc_option <= 3) { h_viscosity_leith(); visc_filt_leith(C_.p.visc_option); }
  else visc_filt_biharmonic(C_.p.visc_option);
}

/* impl_vert_visc_ale: src/oce_ale.F90:2348-2517 */
void orc_impl_vert_visc_ale(void) {
  int nl = NL;
  double *buf = calloc((size_t)10 * (nl + 2), sizeof(double));
  double *a = buf, *b = a + nl + 2, *c = b + nl + 2, *ur = c + nl + 2, *vr = ur + nl + 2, *cp = vr + nl + 2, *up = cp + nl + 2,
         *vp = up + nl + 2, *zbar_n = vp + nl + 2, *Z_n = zbar_n + nl + 2;
  double dt = C_.p.dt;
  for (int e = 1; e <= C_.m.myDim_elem2D; e++) {
    int en[3] = {EN(1, e), EN(2, e), EN(3, e)};
    int nzmin = ULEV(e), nzmax = NLEV(e);
#define WI3(nz) ((A2L(C_.Wvel_i, nz, en[0]) + A2L(C_.Wvel_i, nz, en[1]) + A2L(C_.Wvel_i, nz, en[2])) / 3.)
    zbar_n[nzmax] = C_.m.zbar_e_bot[e - 1];
    Z_n[nzmax - 1] = zbar_n[nzmax] + A2(C_.helem, nzmax - 1, e) / 2.0;
    for (int nz = nzmax - 1; nz >= nzmin + 1; nz--) {
      zbar_n[nz] = zbar_n[nz + 1] + A2(C_.helem, nz, e);
      Z_n[nz - 1] = zbar_n[nz] + A2(C_.helem, nz - 1, e) / 2.0;
    }
    zbar_n[nzmin] = zbar_n[nzmin + 1] + A2(C_.helem, nzmin, e);
    double zinv, wu, wd;
    for (int nz = nzmin + 1; nz <= nzmax - 2; nz++) {
      zinv = 1.0 * dt / (zbar_n[nz] - zbar_n[nz + 1]);
      a[nz] = -A2L(C_.Av, nz, e) / (Z_n[nz - 1] - Z_n[nz]) * zinv;
      c[nz] = -A2L(C_.Av, nz + 1, e) / (Z_n[nz] - Z_n[nz + 1]) * zinv;
      b[nz] = -a[nz] - c[nz] + 1.0;
      wu = WI3(nz); wd = WI3(nz + 1);
      a[nz] = a[nz] + dmin(0., wu) * zinv;
      b[nz] = b[nz] + dmax(0., wu) * zinv;
      b[nz] = b[nz] - dmin(0., wd) * zinv;
      c[nz] = c[nz] - dmax(0., wd) * zinv;
    }
    zinv = 1.0 * dt / (zbar_n[nzmax - 1] - zbar_n[nzmax]);
    a[nzmax - 1] = -A2L(C_.Av, nzmax - 1, e) / (Z_n[nzmax - 2] - Z_n[nzmax - 1]) * zinv;
    b[nzmax - 1] = -a[nzmax - 1] + 1.0;
    c[nzmax - 1] = 0.0;
    wu = WI3(nzmax - 1);
    a[nzmax - 1] = a[nzmax - 1] + dmin(0., wu) * zinv;
    b[nzmax - 1] = b[nzmax - 1] + dmax(0., wu) * zinv;
    zinv = 1.0 * dt / (zbar_n[nzmin] - zbar_n[nzmin + 1]);
    c[nzmin] = -A2L(C_.Av, nzmin + 1, e) / (Z_n[nzmin] - Z_n[nzmin + 1]) * zinv;
    a[nzmin] = 0.0;
    b[nzmin] = -c[nzmin] + 1.0;
    wu = WI3(nzmin); wd = WI3(nzmin + 1);
    b[nzmin] = b[nzmin] + wu * zinv;
    b[nzmin] = b[nzmin] - dmin(0., wd) * zinv;
    c[nzmin] = c[nzmin] - dmax(0., wd) * zinv;
    for (int nz = nzmin; nz <= nzmax - 1; nz++) { ur[nz] = V2(C_.UV_rhs, 1, nz, e); vr[nz] = V2(C_.UV_rhs, 2, nz, e); }
    ur[nzmin] = ur[nzmin] + zinv * C_.stress_surf[2 * (e - 1)] / DENSITY_0;
    vr[nzmin] = vr[nzmin] + zinv * C_.stress_surf[2 * (e - 1) + 1] / DENSITY_0;
    zinv = 1.0 * dt / (zbar_n[nzmax - 1] - zbar_n[nzmax]);
    double ub = V2(C_.UV, 1, nzmax - 1, e), vb = V2(C_.UV, 2, nzmax - 1, e);
    double friction = -C_.p.C_d * sqrt(ub * ub + vb * vb);
    ur[nzmax - 1] = ur[nzmax - 1] + zinv * friction * ub;
    vr[nzmax - 1] = vr[nzmax - 1] + zinv * friction * vb;
    for (int nz = nzmin + 1; nz <= nzmax - 2; nz++) {
      ur[nz] = ur[nz] - a[nz] * V2(C_.UV, 1, nz - 1, e) - (b[nz] - 1.0) * V2(C_.UV, 1, nz, e) - c[nz] * V2(C_.UV, 1, nz + 1, e);
      vr[nz] = vr[nz] - a[nz] * V2(C_.UV, 2, nz - 1, e) - (b[nz] - 1.0) * V2(C_.UV, 2, nz, e) - c[nz] * V2(C_.UV, 2, nz + 1, e);
    }
    ur[nzmin] = ur[nzmin] - (b[nzmin] - 1.0) * V2(C_.UV, 1, nzmin, e) - c[nzmin] * V2(C_.UV, 1, nzmin + 1, e);
    vr[nzmin] = vr[nzmin] - (b[nzmin] - 1.0) * V2(C_.UV, 2, nzmin, e) - c[nzmin] * V2(C_.UV, 2, nzmin + 1, e);
    ur[nzmax - 1] = ur[nzmax - 1] - a[nzmax - 1] * V2(C_.UV, 1, nzmax - 2, e) - (b[nzmax - 1] - 1.0) * V2(C_.UV, 1, nzmax - 1, e);
    vr[nzmax - 1] = vr[nzmax - 1] - a[nzmax - 1] * V2(C_.UV, 2, nzmax - 2, e) - (b[nzmax - 1] - 1.0) * V2(C_.UV, 2, nzmax - 1, e);
    cp[nzmin] = c[nzmin] / b[nzmin];
    up[nzmin] = ur[nzmin] / b[nzmin];
    vp[nzmin] = vr[nzmin] / b[nzmin];
    for (int nz = nzmin + 1; nz <= nzmax - 1; nz++) {
      double m = b[nz] - cp[nz - 1] * a[nz];
      cp[nz] = c[nz] / m;
      up[nz] = (ur[nz] - up[nz - 1] * a[nz]) / m;
      vp[nz] = (vr[nz] - vp[nz - 1] * a[nz]) / m;
    }
    ur[nzmax - 1] = up[nzmax - 1];
    vr[nzmax - 1] = vp[nzmax - 1];
    for (int nz = nzmax - 2; nz >= nzmin; nz--) {
      ur[nz] = up[nz] - cp[nz] * ur[nz + 1];
      vr[nz] = vp[nz] - cp[nz] * vr[nz + 1];
    }
    for (int nz = nzmin; nz <= nzmax - 1; nz++) { V2(C_.UV_rhs, 1, nz, e) = ur[nz]; V2(C_.UV_rhs, 2, nz, e) = vr[nz]; }
  }
  free(buf);
}

/* update_stiff_mat_ale: src/oce_ale.F90:1371-1470 */
void orc_update_stiff_mat_ale(void) {
  const int *rowptr = C_.m.ssh_rowptr, *col = C_.m.ssh_colind_loc;
  int *n_num = calloc(C_.N, sizeof(int));
  double factor = G_ACC * C_.p.dt * C_.p.alpha * C_.p.theta;
  for (int ed = 1; ed <= C_.m.myDim_edge2D; ed++) {
    for (int j = 1; j <= 2; j++) {
      int row = EDG(j, ed);
      if (row > C_.m.myDim_nod2D) continue;
      int offset = rowptr[row - 1] - rowptr[0];
      for (int n = 1; n <= rowptr[row] - rowptr[row - 1]; n++) n_num[col[offset + n - 1] - 1] = offset + n;
      for (int i = 1; i <= 2; i++) {
        int elem = ETRI(i, ed);
        if (elem < 1) continue;
        double fy[3];
        for (int k = 1; k <= 3; k++) fy[k - 1] = -C_.dhe[elem - 1] * (GS(k, elem) * ECD(2 * i, ed) - GS(3 + k, elem) * ECD(2 * i - 1, ed));
        if (i == 2) for (int k = 0; k < 3; k++) fy[k] = -fy[k];
        if (j == 2) for (int k = 0; k < 3; k++) fy[k] = -fy[k];
        for (int k = 1; k <= 3; k++) {
          int p = n_num[EN(k, elem) - 1];
          C_.ssh_values[p - 1] = C_.ssh_values[p - 1] + fy[k - 1] * factor;
        }
      }
    }
  }
  free(n_num);
}

/* compute_ssh_rhs_ale: src/oce_ale.F90:1478-1572 */
void orc_compute_ssh_rhs_ale(void) {
  double alpha = C_.p.alpha;
  for (int n = 0; n < C_.N; n++) C_.ssh_rhs[n] = 0.0;
  for (int ed = 1; ed <= C_.m.myDim_edge2D; ed++) {
    int e1 = ETRI(1, ed), e2 = ETRI(2, ed);
    double c1 = 0.0, c2 = 0.0;
    double dX1 = ECD(1, ed), dY1 = ECD(2, ed);
    for (int nz = ULEV(e1); nz <= NLEV(e1) - 1; nz++)
      c1 = c1 + alpha * ((V2(C_.UV, 2, nz, e1) + V2(C_.UV_rhs, 2, nz, e1)) * dX1 - (V2(C_.UV, 1, nz, e1) + V2(C_.UV_rhs, 1, nz, e1)) * dY1) *
                    A2(C_.helem, nz, e1);
    if (e2 > 0) {
      double dX2 = ECD(3, ed), dY2 = ECD(4, ed);
      for (int nz = ULEV(e2); nz <= NLEV(e2) - 1; nz++)
        c2 = c2 - alpha * ((V2(C_.UV, 2, nz, e2) + V2(C_.UV_rhs, 2, nz, e2)) * dX2 - (V2(C_.UV, 1, nz, e2) + V2(C_.UV_rhs, 1, nz, e2)) * dY2) *
                      A2(C_.helem, nz, e2);
    }
    C_.ssh_rhs[EDG(1, ed) - 1] = C_.ssh_rhs[EDG(1, ed) - 1] + (c1 + c2);
    C_.ssh_rhs[EDG(2, ed) - 1] = C_.ssh_rhs[EDG(2, ed) - 1] - (c1 + c2);
  }
  if (C_.p.which_ale != 0) {
    for (int n = 1; n <= C_.m.myDim_nod2D; n++)
      C_.ssh_rhs[n - 1] = C_.ssh_rhs[n - 1] - alpha * C_.water_flux[n - 1] * AREASVOL(ULEVN(n), n) + (1.0 - alpha) * C_.ssh_rhs_old[n - 1];
  } else
    for (int n = 1; n <= C_.m.myDim_nod2D; n++) C_.ssh_rhs[n - 1] = C_.ssh_rhs[n - 1] + (1.0 - alpha) * C_.ssh_rhs_old[n - 1];
}

/* update_vel: src/oce_dyn.F90:101-131 */
void orc_update_vel(void) {
  double fac = -G_ACC * C_.p.theta * C_.p.dt;
  for (int e = 1; e <= C_.m.myDim_elem2D; e++) {
    double eta[3];
    for (int k = 0; k < 3; k++) eta[k] = fac * C_.d_eta[EN(k + 1, e) - 1];
    double Fx = GS(1, e) * eta[0] + GS(2, e) * eta[1] + GS(3, e) * eta[2];
    double Fy = GS(4, e) * eta[0] + GS(5, e) * eta[1] + GS(6, e) * eta[2];
    for (int nz = ULEV(e); nz <= NLEV(e) - 1; nz++) {
      V2(C_.UV, 1, nz, e) = V2(C_.UV, 1, nz, e) + V2(C_.UV_rhs, 1, nz, e) + Fx;
      V2(C_.UV, 2, nz, e) = V2(C_.UV, 2, nz, e) + V2(C_.UV_rhs, 2, nz, e) + Fy;
    }
  }
  for (int n = 0; n < C_.N; n++) C_.eta_n[n] = C_.eta_n[n] + C_.d_eta[n];
}

/* compute_hbar_ale: src/oce_ale.F90:1585-1676 */
void orc_compute_hbar_ale(void) {
  for (int n = 0; n < C_.N; n++) C_.ssh_rhs_old[n] = 0.0;
  for (int ed = 1; ed <= C_.m.myDim_edge2D; ed++) {
    int e1 = ETRI(1, ed), e2 = ETRI(2, ed);
    double c1 = 0.0, c2 = 0.0, dX1 = ECD(1, ed), dY1 = ECD(2, ed);
    for (int nz = ULEV(e1); nz <= NLEV(e1) - 1; nz++)
      c1 = c1 + (V2(C_.UV, 2, nz, e1) * dX1 - V2(C_.UV, 1, nz, e1) * dY1) * A2(C_.helem, nz, e1);
    if (e2 > 0) {
      double dX2 = ECD(3, ed), dY2 = ECD(4, ed);
      for (int nz = ULEV(e2); nz <= NLEV(e2) - 1; nz++)
        c2 = c2 - (V2(C_.UV, 2, nz, e2) * dX2 - V2(C_.UV, 1, nz, e2) * dY2) * A2(C_.helem, nz, e2);
    }
    C_.ssh_rhs_old[EDG(1, ed) - 1] = C_.ssh_rhs_old[EDG(1, ed) - 1] + (c1 + c2);
    C_.ssh_rhs_old[EDG(2, ed) - 1] = C_.ssh_rhs_old[EDG(2, ed) - 1] - (c1 + c2);
  }
  if (C_.p.which_ale != 0)
    for (int n = 1; n <= C_.m.myDim_nod2D; n++)
      C_.ssh_rhs_old[n - 1] = C_.ssh_rhs_old[n - 1] - C_.water_flux[n - 1] * AREASVOL(ULEVN(n), n);
  for (int n = 0; n < C_.N; n++) C_.hbar_old[n] = C_.hbar[n];
  for (int n = 1; n <= C_.m.myDim_nod2D; n++)
    C_.hbar[n - 1] = C_.hbar_old[n - 1] + C_.ssh_rhs_old[n - 1] * C_.p.dt / AREASVOL(ULEVN(n), n);
  for (int e = 1; e <= C_.m.myDim_elem2D; e++) {
    int n1 = EN(1, e) - 1, n2 = EN(2, e) - 1, n3 = EN(3, e) - 1;
    if (ULEV(e) > 1) C_.dhe[e - 1] = 0.0;
    else C_.dhe[e - 1] = ((C_.hbar[n1] - C_.hbar_old[n1]) + (C_.hbar[n2] - C_.hbar_old[n2]) + (C_.hbar[n3] - C_.hbar_old[n3])) / 3.0;
  }
}

/* oce_ale.F90:2722 */
void orc_eta_update(void) {
  for (int n = 1; n <= C_.N; n++)
    if (ULEVN(n) == 1) C_.eta_n[n - 1] = C_.p.alpha * C_.hbar[n - 1] + (1.0 - C_.p.alpha) * C_.hbar_old[n - 1];
}

/* vert_vel_ale: src/oce_ale.F90:1692-2204 (linfs and zstar branches, no GM) */
void orc_vert_vel_ale(void) {
  double dt = C_.p.dt;
  memset(C_.Wvel, 0, sizeof(double) * NL * C_.N);
  for (int ed = 1; ed <= C_.m.myDim_edge2D; ed++) {
    int n1 = EDG(1, ed), n2 = EDG(2, ed), e1 = ETRI(1, ed), e2 = ETRI(2, ed);
    double dX1 = ECD(1, ed), dY1 = ECD(2, ed);
    for (int nz = NLEV(e1) - 1; nz >= ULEV(e1); nz--) {
      double c1 = (V2(C_.UV, 2, nz, e1) * dX1 - V2(C_.UV, 1, nz, e1) * dY1) * A2(C_.helem, nz, e1);
      A2L(C_.Wvel, nz, n1) = A2L(C_.Wvel, nz, n1) + c1;
      A2L(C_.Wvel, nz, n2) = A2L(C_.Wvel, nz, n2) - c1;
    }
    if (e2 > 0) {
      double dX2 = ECD(3, ed), dY2 = ECD(4, ed);
      for (int nz = NLEV(e2) - 1; nz >= ULEV(e2); nz--) {
        double c2 = -(V2(C_.UV, 2, nz, e2) * dX2 - V2(C_.UV, 1, nz, e2) * dY2) * A2(C_.helem, nz, e2);
        A2L(C_.Wvel, nz, n1) = A2L(C_.Wvel, nz, n1) + c2;
        A2L(C_.Wvel, nz, n2) = A2L(C_.Wvel, nz, n2) - c2;
      }
    }
  }
  for (int n = 1; n <= C_.m.myDim_nod2D; n++)
    for (int nz = NLEVN(n) - 1; nz >= ULEVN(n); nz--) A2L(C_.Wvel, nz, n) = A2L(C_.Wvel, nz, n) + A2L(C_.Wvel, nz + 1, n);
  for (int n = 1; n <= C_.m.myDim_nod2D; n++)
    for (int nz = ULEVN(n); nz <= NLEVN(n) - 1; nz++) A2L(C_.Wvel, nz, n) = A2L(C_.Wvel, nz, n) / AREA(nz, n);
  if (C_.p.which_ale == 2) {
    for (int n = 1; n <= C_.m.myDim_nod2D; n++) {
      int nzmin = ULEVN(n), nzmax = C_.m.nlevels_nod2D_min[n - 1] - 1;
      if (nzmin == 1) {
        double dd1 = A2L(C_.zbar_3d_n, nzmax, n);
        double dd = A2L(C_.zbar_3d_n, nzmin, n) - dd1;
        dd = (C_.hbar[n - 1] - C_.hbar_old[n - 1]) / dd;
        double dddt = dd / dt;
        for (int nz = nzmin; nz <= nzmax - 1; nz++) {
          A2L(C_.Wvel, nz, n) = A2L(C_.Wvel, nz, n) - (A2L(C_.zbar_3d_n, nz, n) - dd1) * dddt;
          A2(C_.hnode_new, nz, n) = A2(C_.hnode, nz, n) + (A2L(C_.zbar_3d_n, nz, n) - A2L(C_.zbar_3d_n, nz + 1, n)) * dd;
        }
      }
      A2L(C_.Wvel, nzmin, n) = A2L(C_.Wvel, nzmin, n) - C_.water_flux[n - 1];
    }
  }
  if (C_.p.which_ale == 1) {            /* zlevel (oce_ale.F90:1830-2023) */
    const int lz = C_.p.lzstar_lev;
    const double *zbar = C_.m.zbar;     /* zbar(k) = zbar[k-1] */
    for (int n = 1; n <= C_.m.myDim_nod2D; n++) {
      int nzmin = ULEVN(n), nzmax = C_.m.nlevels_nod2D_min[n - 1] - 1;
      if (nzmin == 1) {
        const double dhbar_total = C_.hbar[n - 1] - C_.hbar_old[n - 1];
        int changed = 0;                /* any(hnode(nzmin+1:nzmin+lz-1,n) /= zbar(nzmin+1:nzmin+lz-1)-zbar(nzmin+2:nzmin+lz)) */
        for (int k = 2; k <= lz; k++) if (A2(C_.hnode, nzmin + k - 1, n) != (zbar[nzmin + k - 2] - zbar[nzmin + k - 1])) changed = 1;
        if (dhbar_total < 0.0 && A2(C_.hnode, nzmin, n) + dhbar_total <= (zbar[nzmin - 1] - zbar[nzmin]) * C_.p.min_hnode) {
          /* the reference goes to its local-zstar fallback here (:1859-1942) and its update_thickness_ale then stops in a non-conformable PACK (:871-873, run-time error
           * under the compiler the reference is built with in this repository): not restated, reported */
          orc_ale_flag = 1;
          A2L(C_.Wvel, nzmin, n) = A2L(C_.Wvel, nzmin, n) - dhbar_total / dt;
          A2(C_.hnode_new, nzmin, n) = A2(C_.hnode, nzmin, n) + dhbar_total;
        } else if (dhbar_total > 0.0 && changed) {         /* return to zlevel (:1950-2003): refill the sub-surface layers first */
          double max_d[64];
          int nz = 0;
          for (int k = 1; k <= lz; k++) {
            max_d[k] = (zbar[nzmin + k - 2] - zbar[nzmin + k - 1]) - A2(C_.hnode, nzmin + k - 1, n);
            if (A2(C_.hnode, nzmin + k - 1, n) != (zbar[nzmin + k - 2] - zbar[nzmin + k - 1])) nz = k;
          }
          max_d[1] = 1000.0;
          nzmax = nz < nzmax - 1 ? nz : nzmax - 1;
          double rest = dhbar_total, integ = 0.0;
          for (nz = nzmax; nz >= 1; nz--) {
            const double d = dmin(rest, max_d[nz]);
            rest = rest - d;
            rest = dmax(0.0, rest);
            integ = integ + d;
            A2L(C_.Wvel, nzmin + nz - 1, n) = A2L(C_.Wvel, nzmin + nz - 1, n) - integ / dt;
            A2(C_.hnode_new, nzmin + nz - 1, n) = A2(C_.hnode, nzmin + nz - 1, n) + d;
          }
        } else {
          A2L(C_.Wvel, nzmin, n) = A2L(C_.Wvel, nzmin, n) - dhbar_total / dt;
          A2(C_.hnode_new, nzmin, n) = A2(C_.hnode, nzmin, n) + dhbar_total;
        }
      }
      A2L(C_.Wvel, nzmin, n) = A2L(C_.Wvel, nzmin, n) - C_.water_flux[n - 1];
    }
  }
  for (int n = 1; n <= C_.N; n++) A2L(C_.CFL_z, 1, n) = 0.0;
  for (int n = 1; n <= C_.N; n++)
    for (int nz = ULEVN(n); nz <= NLEVN(n) - 1; nz++) {
      double c1 = fabs(A2L(C_.Wvel, nz, n) * dt / A2(C_.hnode_new, nz, n));
      double c2 = fabs(A2L(C_.Wvel, nz + 1, n) * dt / A2(C_.hnode_new, nz, n));
      A2L(C_.CFL_z, nz, n) = A2L(C_.CFL_z, nz, n) + c1;
      A2L(C_.CFL_z, nz + 1, n) = c2;
    }
  for (int n = 1; n <= C_.N; n++)
    for (int nz = ULEVN(n); nz <= NLEVN(n); nz++) {
      double c1 = 1.0, c2 = 0.0;
      if (C_.p.w_split && (A2L(C_.CFL_z, nz, n) > C_.p.w_max_cfl)) {
        double dd = dmax((A2L(C_.CFL_z, nz, n) - C_.p.w_max_cfl), 0.0) / dmax(C_.p.w_max_cfl, 1.e-12);
        c1 = 1.0 / (1.0 + dd);
        c2 = dd / (1.0 + dd);
      }
      A2L(C_.Wvel_e, nz, n) = c1 * A2L(C_.Wvel, nz, n);
      A2L(C_.Wvel_i, nz, n) = c2 * A2L(C_.Wvel, nz, n);
    }
}

/* update_thickness_ale: src/oce_ale.F90:800-993 (zlevel and zstar branches; linfs: nothing) */
void orc_update_thickness_ale(void) {
  if (C_.p.which_ale == 1) {            /* zlevel (:817-943).  The element part's local-zstar case (:865-880) is the PACK the reference stops in: see vert_vel_ale */
    const int lz = C_.p.lzstar_lev;
    for (int e = 1; e <= C_.m.myDim_elem2D; e++) {
      int nzmin = ULEV(e);
      if (nzmin > 1) continue;
      int n1 = EN(1, e), n2 = EN(2, e), n3 = EN(3, e);
      A2(C_.helem, nzmin, e) = (A2(C_.hnode_new, nzmin, n1) + A2(C_.hnode_new, nzmin, n2) + A2(C_.hnode_new, nzmin, n3)) / 3.0;
    }
    for (int n = 1; n <= C_.N; n++) {
      int nzmin = C_.m.ulevels_nod2D_max[n - 1], nzmax = C_.m.nlevels_nod2D_min[n - 1] - 1;
      if (nzmin > 1) continue;
      int local = 0, top = nzmin;
      for (int k = 2; k <= lz; k++) if (A2(C_.hnode_new, nzmin + k - 1, n) - A2(C_.hnode, nzmin + k - 1, n) != 0.0) local = 1;
      if (local) {                      /* hnode of the layers the fallback / the return to zlevel changed */
        for (int k = 1; k <= lz; k++) if (A2(C_.hnode_new, nzmin + k - 1, n) - A2(C_.hnode, nzmin + k - 1, n) != 0.0) top = nzmin + k - 1;
        top = top < nzmax - 1 ? top : nzmax - 1;
      }
      for (int nz = top; nz >= nzmin; nz--) {
        A2(C_.hnode, nz, n) = A2(C_.hnode_new, nz, n);
        A2L(C_.zbar_3d_n, nz, n) = A2L(C_.zbar_3d_n, nz + 1, n) + A2(C_.hnode_new, nz, n);
        A2(C_.Z_3d_n, nz, n) = A2L(C_.zbar_3d_n, nz + 1, n) + A2(C_.hnode_new, nz, n) / 2.0;
      }
    }
    return;
  }
  if (C_.p.which_ale != 2) return;
  for (int n = 1; n <= C_.N; n++) {
    int nzmin = ULEVN(n), nzmax = C_.m.nlevels_nod2D_min[n - 1] - 2;
    if (nzmin > 1) continue;
    for (int nz = nzmax; nz >= nzmin; nz--) {
      A2(C_.hnode, nz, n) = A2(C_.hnode_new, nz, n);
      A2L(C_.zbar_3d_n, nz, n) = A2L(C_.zbar_3d_n, nz + 1, n) + A2(C_.hnode_new, nz, n);
      A2(C_.Z_3d_n, nz, n) = A2L(C_.zbar_3d_n, nz + 1, n) + A2(C_.hnode_new, nz, n) / 2.0;
    }
  }
  for (int e = 1; e <= C_.m.myDim_elem2D; e++) {
    int nzmin = ULEV(e), nzmax = NLEV(e) - 1;
    if (nzmin > 1) continue;
    int n1 = EN(1, e), n2 = EN(2, e), n3 = EN(3, e);
    for (int nz = nzmin; nz <= nzmax - 1; nz++)
      A2(C_.helem, nz, e) = (A2(C_.hnode, nz, n1) + A2(C_.hnode, nz, n2) + A2(C_.hnode, nz, n3)) / 3.0;
  }
}
