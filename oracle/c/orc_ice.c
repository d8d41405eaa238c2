/* ORACLE (test infrastructure): CPU restatement of the sea-ice rheologies EVPdynamics (classic EVP) and EVPdynamics_a (adaptive EVP), both below, and mEVP, EVPdynamics_m (src/ice_maEVP.F90:273-602; whichEVP = 1,
 * no cavities, no icepack), loop for loop in the reference's order.  Pinned bitwise on a run of the reference's own routine
 * (oracle/_ref, driver mode 'ice', one MPI rank: tests/golden/ice_evp_reference.npz, tests/test_ice.py).  Stand-alone: it works
 * on a mesh descriptor and the ice arrays, not on the ocean context. */
#include "../../include/fesom_gpu.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define G_ACC 9.81
#define DENSITY_0 1030.0
#define RHOICE 910.0
#define RHOSNO 290.0
#define INV_RHOWAT (1. / 1025.)            /* i_therm_param (src/ice_modules.F90) */

int orc_ice_evp(const fesom_mesh_desc *m, const fesom_ice_params *p, fesom_ice_state *s) {
  const int myN = m->myDim_nod2D, N = myN + m->eDim_nod2D, myE = m->myDim_elem2D, nl = m->nl;
  const double val3 = 1.0 / 3.0, vale = 1.0 / (p->ellipse * p->ellipse);
  const double det2 = 1.0 / (1.0 + p->alpha_evp), det1 = p->alpha_evp * det2, rdt = p->ice_dt;
  double *ua = malloc(sizeof(double) * N), *va = malloc(sizeof(double) * N), *rhs_a = calloc(N, sizeof(double)), *rhs_m = calloc(N, sizeof(double));
  double *urhs = calloc(N, sizeof(double)), *vrhs = calloc(N, sizeof(double)), *invt = calloc(myN, sizeof(double)), *mass = calloc(myN, sizeof(double));
  double *pfac = calloc(myE, sizeof(double));
  char *ice_nod = calloc(myN, 1), *ice_el = calloc(myE, 1), *bnd = calloc(N, 1);
  memcpy(ua, s->u_ice, sizeof(double) * N); memcpy(va, s->v_ice, sizeof(double) * N);
  /* coastal nodes: both ends of the owned boundary edges (:568-573; the same nodes carry bc_index_nod2D = 0, oce_mesh.F90:2404-2413) */
  for (int ed = 0; ed < m->myDim_edge2D; ed++)
    if (m->myList_edge2D[ed] > m->edge2D_in) { bnd[m->edges[2 * ed] - 1] = 1; bnd[m->edges[2 * ed + 1] - 1] = 1; }
  /* ssh2rhs inlined (:340-392) */
  for (int el = 0; el < myE; el++) {
    if (m->ulevels[el] > 1) continue;
    const int *en = m->elem2D_nodes + 3 * el;
    const double *gs = m->gradient_sca + 6 * (size_t)el;
    const double vol = m->elem_area[el];
    double e3[3];
    for (int k = 0; k < 3; k++) {
      e3[k] = s->elevation[en[k] - 1];
      if (p->use_floatice) {
        double pi = (RHOICE * s->m_ice[en[k] - 1] + RHOSNO * s->m_snow[en[k] - 1]) * INV_RHOWAT;
        pi = pi < p->max_ice_loading ? pi : p->max_ice_loading;
        e3[k] = e3[k] + pi;
      }
    }
    double bb = G_ACC * val3 * vol;
    const double aa = bb * ((gs[0] * e3[0] + gs[1] * e3[1]) + gs[2] * e3[2]);
    bb = bb * ((gs[3] * e3[0] + gs[4] * e3[1]) + gs[5] * e3[2]);
    for (int k = 0; k < 3; k++) { rhs_a[en[k] - 1] = rhs_a[en[k] - 1] - aa; rhs_m[en[k] - 1] = rhs_m[en[k] - 1] - bb; }
  }
  /* thickness, mass (:394-419) */
  for (int i = 0; i < myN; i++) {
    if (m->ulevels_nod2D[i] > 1) continue;
    if (s->a_ice[i] >= 0.01) {
      double it = (RHOICE * s->m_ice[i] + RHOSNO * s->m_snow[i]) / s->a_ice[i];
      invt[i] = 1.0 / (it > 9.0 ? it : 9.0);
      double ms = (s->m_ice[i] * RHOICE + s->m_snow[i] * RHOSNO);
      const double ar = m->area[(size_t)i * nl];
      mass[i] = ms / ((1.0 + ms * ms) * ar);
      rhs_a[i] = rhs_a[i] / ar; rhs_m[i] = rhs_m[i] / ar;
      ice_nod[i] = 1;
    }
  }
  /* pressure factor (:421-438) */
  for (int el = 0; el < myE; el++) {
    if (m->ulevels[el] > 1) continue;
    const int *en = m->elem2D_nodes + 3 * el;
    const double msum = ((s->m_ice[en[0] - 1] + s->m_ice[en[1] - 1]) + s->m_ice[en[2] - 1]) * val3;
    if (msum > 0.01) {
      ice_el[el] = 1;
      const double asum = ((s->a_ice[en[0] - 1] + s->a_ice[en[1] - 1]) + s->a_ice[en[2] - 1]) * val3;
      pfac[el] = det2 * p->Pstar * msum * exp(-p->c_pressure * (1.0 - asum));
    }
  }
  for (int sub = 0; sub < p->evp_rheol_steps; sub++) {
    for (int el = 0; el < myE; el++) {
      if (m->ulevels[el] > 1 || !ice_el[el]) continue;
      const int *en = m->elem2D_nodes + 3 * el;
      const double *dx = m->gradient_sca + 6 * (size_t)el, *dy = dx + 3;
      const double meancos = val3 * m->metric_factor[el];
      const double u1 = ua[en[0] - 1], u2 = ua[en[1] - 1], u3 = ua[en[2] - 1], v1 = va[en[0] - 1], v2 = va[en[1] - 1], v3 = va[en[2] - 1];
      const double eps11 = ((dx[0] * u1 + dx[1] * u2) + dx[2] * u3) - ((v1 + v2) + v3) * meancos;
      const double eps22 = (dy[0] * v1 + dy[1] * v2) + dy[2] * v3;
      const double eps12 = 0.5 * ((((dy[0] * u1 + dx[0] * v1) + (dy[1] * u2 + dx[1] * v2)) + (dy[2] * u3 + dx[2] * v3)) + ((u1 + u2) + u3) * meancos);
      const double eps1 = eps11 + eps22, eps2 = eps11 - eps22;
      const double delta = sqrt(eps1 * eps1 + vale * (eps2 * eps2 + 4.0 * (eps12 * eps12)));
      const double pressure = pfac[el] / (delta + p->delta_min);
      s->sigma12[el] = det1 * s->sigma12[el] + pressure * eps12 * vale;
      s->sigma11[el] = det1 * s->sigma11[el] + 0.5 * pressure * (eps1 - delta + eps2 * vale);
      s->sigma22[el] = det1 * s->sigma22[el] + 0.5 * pressure * (eps1 - delta - eps2 * vale);
      const double ar = m->elem_area[el], s11 = s->sigma11[el], s12 = s->sigma12[el], s22 = s->sigma22[el];
      for (int k = 0; k < 3; k++)
        if (en[k] <= myN) {
          urhs[en[k] - 1] = urhs[en[k] - 1] - ar * (s11 * dx[k] + s12 * (dy[k] + meancos));
          vrhs[en[k] - 1] = vrhs[en[k] - 1] - ar * (s12 * dx[k] + s22 * dy[k] - s11 * meancos);
        }
    }
    for (int i = 0; i < myN; i++) {
      if (m->ulevels_nod2D[i] > 1 || !ice_nod[i]) continue;
      urhs[i] = urhs[i] * mass[i] + rhs_a[i];
      vrhs[i] = vrhs[i] * mass[i] + rhs_m[i];
      const double du = ua[i] - s->u_w[i], dv = va[i] - s->v_w[i];
      const double umod = sqrt(du * du + dv * dv);
      const double drag = rdt * p->cd_oce_ice * umod * DENSITY_0 * invt[i];
      const double rhsu = s->u_ice[i] + drag * s->u_w[i] + rdt * (invt[i] * s->stress_atmice_x[i] + urhs[i]) + p->beta_evp * ua[i];
      const double rhsv = s->v_ice[i] + drag * s->v_w[i] + rdt * (invt[i] * s->stress_atmice_y[i] + vrhs[i]) + p->beta_evp * va[i];
      const double bd = 1.0 + p->beta_evp + drag, rc = rdt * m->coriolis_node[i];
      const double det = (bnd[i] ? 0.0 : 1.0) / (bd * bd + rc * rc);
      ua[i] = det * (bd * rhsu + rc * rhsv);
      va[i] = det * (bd * rhsv - rc * rhsu);
    }
    for (int i = 0; i < N; i++) if (bnd[i]) { ua[i] = 0.0; va[i] = 0.0; }
    /* (exchange_nod of u_ice_aux, v_ice_aux: single partition) */
    for (int i = 0; i < myN; i++) { urhs[i] = 0.0; vrhs[i] = 0.0; }
  }
  memcpy(s->u_ice, ua, sizeof(double) * N); memcpy(s->v_ice, va, sizeof(double) * N);
  free(ua); free(va); free(rhs_a); free(rhs_m); free(urhs); free(vrhs); free(invt); free(mass); free(pfac); free(ice_nod); free(ice_el); free(bnd);
  return 0;
}

/* Adaptive EVP, EVPdynamics_a (src/ice_maEVP.F90:785-888; whichEVP = 2): ssh2rhs (:130-202), per subcycle stress_tensor_a (:686-784) with the element's
 * own alpha, stress2rhs_m (:206-272), the node update with the node's beta (:831-856), coastal nodes (:860-877); after the subcycles find_alpha_field_a
 * (:611-683) and find_beta_field_a (:892-922).  Not the fused order of EVPdynamics_m above: every sum and product as these routines write them.
 * One partition.  Pinned bitwise on a run of the reference's own routine (tests/golden/ice_aevp_reference.npz, tests/test_ice.py). */
int orc_ice_evp_a(const fesom_mesh_desc *m, const fesom_ice_params *p, fesom_ice_state *s) {
  const int myN = m->myDim_nod2D, N = myN + m->eDim_nod2D, myE = m->myDim_elem2D, nl = m->nl;
  const double val3 = 1.0 / 3.0, vale = 1.0 / (p->ellipse * p->ellipse), rdt = p->ice_dt;
  double *ua = malloc(sizeof(double) * N), *va = malloc(sizeof(double) * N), *rhs_a = calloc(N, sizeof(double)), *rhs_m = calloc(N, sizeof(double));
  double *urhs = calloc(N, sizeof(double)), *vrhs = calloc(N, sizeof(double));
  char *bnd = calloc(N, 1);
  memcpy(ua, s->u_ice, sizeof(double) * N); memcpy(va, s->v_ice, sizeof(double) * N);
  for (int ed = 0; ed < m->myDim_edge2D; ed++)
    if (m->myList_edge2D[ed] > m->edge2D_in) { bnd[m->edges[2 * ed] - 1] = 1; bnd[m->edges[2 * ed + 1] - 1] = 1; }
  /* ssh2rhs */
  for (int el = 0; el < myE; el++) {
    if (m->ulevels[el] > 1) continue;
    const int *en = m->elem2D_nodes + 3 * el;
    const double *gs = m->gradient_sca + 6 * (size_t)el;
    double e3[3];
    for (int k = 0; k < 3; k++) {
      e3[k] = s->elevation[en[k] - 1];
      if (p->use_floatice) {
        double pi = (RHOICE * s->m_ice[en[k] - 1] + RHOSNO * s->m_snow[en[k] - 1]) * INV_RHOWAT;
        pi = pi < p->max_ice_loading ? pi : p->max_ice_loading;
        e3[k] = e3[k] + pi;
      }
    }
    double bb = G_ACC * val3 * m->elem_area[el];
    const double aa = bb * ((gs[0] * e3[0] + gs[1] * e3[1]) + gs[2] * e3[2]);
    bb = bb * ((gs[3] * e3[0] + gs[4] * e3[1]) + gs[5] * e3[2]);
    for (int k = 0; k < 3; k++) { rhs_a[en[k] - 1] = rhs_a[en[k] - 1] - aa; rhs_m[en[k] - 1] = rhs_m[en[k] - 1] - bb; }
  }
  for (int sub = 0; sub <= p->evp_rheol_steps; sub++) {      /* the last round only evaluates find_alpha_field_a on the final velocities */
    const int last = sub == p->evp_rheol_steps;
    for (int el = 0; el < myE; el++) {
      if (m->ulevels[el] > 1) continue;
      const double alpha = s->alpha_evp_array[el];
      const double det2 = 1.0 / (1.0 + alpha), det1 = alpha * det2;
      const int *en = m->elem2D_nodes + 3 * el;
      const double msum = ((s->m_ice[en[0] - 1] + s->m_ice[en[1] - 1]) + s->m_ice[en[2] - 1]) * val3;
      if (msum <= 0.01) continue;
      const double asum = ((s->a_ice[en[0] - 1] + s->a_ice[en[1] - 1]) + s->a_ice[en[2] - 1]) * val3;
      const double *dx = m->gradient_sca + 6 * (size_t)el, *dy = dx + 3;
      const double u1 = ua[en[0] - 1], u2 = ua[en[1] - 1], u3 = ua[en[2] - 1], v1 = va[en[0] - 1], v2 = va[en[1] - 1], v3 = va[en[2] - 1];
      const double vsum = (v1 + v2) + v3, usum = (u1 + u2) + u3, meancos = m->metric_factor[el];
      double eps11 = (dx[0] * u1 + dx[1] * u2) + dx[2] * u3;
      eps11 = eps11 - val3 * vsum * meancos;
      const double eps22 = (dy[0] * v1 + dy[1] * v2) + dy[2] * v3;
      double eps12 = 0.5 * (((dy[0] * u1 + dx[0] * v1) + (dy[1] * u2 + dx[1] * v2)) + (dy[2] * u3 + dx[2] * v3));
      eps12 = eps12 + 0.5 * val3 * usum * meancos;
      const double eps1 = eps11 + eps22, eps2 = eps11 - eps22;
      double delta = eps1 * eps1 + vale * (eps2 * eps2 + 4.0 * (eps12 * eps12));
      delta = sqrt(delta);
      if (last) {                                           /* find_alpha_field_a */
        const double pressure = p->Pstar * exp(-p->c_pressure * (1.0 - asum)) / (delta + p->delta_min);
        const double al = sqrt(p->ice_dt * p->c_aevp * pressure / RHOICE / m->elem_area[el]);
        s->alpha_evp_array[el] = al > 50.0 ? al : 50.0;
        continue;
      }
      const double pressure = p->Pstar * msum * exp(-p->c_pressure * (1.0 - asum)) / (delta + p->delta_min);
      const double r1 = pressure * (eps1 - delta), r2 = pressure * eps2 * vale, r3 = pressure * eps12 * vale;
      double si1 = s->sigma11[el] + s->sigma22[el], si2 = s->sigma11[el] - s->sigma22[el];
      si1 = det1 * si1 + det2 * r1;
      si2 = det1 * si2 + det2 * r2;
      s->sigma12[el] = det1 * s->sigma12[el] + det2 * r3;
      s->sigma11[el] = 0.5 * (si1 + si2);
      s->sigma22[el] = 0.5 * (si1 - si2);
    }
    if (last) break;
    /* stress2rhs_m */
    for (int i = 0; i < myN; i++) { urhs[i] = 0.0; vrhs[i] = 0.0; }
    for (int el = 0; el < myE; el++) {
      if (m->ulevels[el] > 1) continue;
      const int *en = m->elem2D_nodes + 3 * el;
      if ((s->a_ice[en[0] - 1] + s->a_ice[en[1] - 1]) + s->a_ice[en[2] - 1] < 0.01) continue;
      const double vol = m->elem_area[el], mf = m->metric_factor[el];
      const double *dx = m->gradient_sca + 6 * (size_t)el, *dy = dx + 3;
      const double s11 = s->sigma11[el], s12 = s->sigma12[el], s22 = s->sigma22[el];
      for (int k = 0; k < 3; k++) {
        const int row = en[k] - 1;
        urhs[row] = urhs[row] - vol * (s11 * dx[k] + s12 * dy[k]) - vol * s12 * val3 * mf;
        vrhs[row] = vrhs[row] - vol * (s12 * dx[k] + s22 * dy[k]) + vol * s11 * val3 * mf;
      }
    }
    for (int i = 0; i < myN; i++) {
      if (m->ulevels_nod2D[i] > 1) continue;
      double mass = (s->m_ice[i] * RHOICE + s->m_snow[i] * RHOSNO);
      mass = mass / (1.0 + mass * mass);
      const double ar = m->area[(size_t)i * nl];
      urhs[i] = (urhs[i] * mass + rhs_a[i]) / ar;
      vrhs[i] = (vrhs[i] * mass + rhs_m[i]) / ar;
    }
    /* node update (:831-856) */
    for (int i = 0; i < myN; i++) {
      if (m->ulevels_nod2D[i] > 1) continue;
      double thickness = (RHOICE * s->m_ice[i] + RHOSNO * s->m_snow[i]) / (s->a_ice[i] > 0.01 ? s->a_ice[i] : 0.01);
      thickness = thickness > 9.0 ? thickness : 9.0;
      const double inv_thickness = 1.0 / thickness;
      const double du = ua[i] - s->u_w[i], dv = va[i] - s->v_w[i];
      const double umod = sqrt(du * du + dv * dv);
      const double drag = rdt * p->cd_oce_ice * umod * DENSITY_0 * inv_thickness;
      double rhsu = s->u_ice[i] + drag * s->u_w[i] + rdt * (inv_thickness * s->stress_atmice_x[i] + urhs[i]);
      double rhsv = s->v_ice[i] + drag * s->v_w[i] + rdt * (inv_thickness * s->stress_atmice_y[i] + vrhs[i]);
      const double beta = s->beta_evp_array[i];
      rhsu = beta * ua[i] + rhsu;
      rhsv = beta * va[i] + rhsv;
      const double fc = rdt * m->coriolis_node[i];
      double det = (1.0 + beta + drag) * (1.0 + beta + drag) + fc * fc;
      det = (bnd[i] ? 0.0 : 1.0) / det;
      ua[i] = det * ((1.0 + beta + drag) * rhsu + fc * rhsv);
      va[i] = det * ((1.0 + beta + drag) * rhsv - fc * rhsu);
    }
    for (int i = 0; i < N; i++) if (bnd[i]) { ua[i] = 0.0; va[i] = 0.0; }
  }
  memcpy(s->u_ice, ua, sizeof(double) * N); memcpy(s->v_ice, va, sizeof(double) * N);
  /* find_beta_field_a */
  for (int i = 0; i < myN; i++) {
    if (m->ulevels_nod2D[i] > 1) continue;
    const int num = m->nod_in_elem2D_num[i];
    double b = s->alpha_evp_array[m->nod_in_elem2D[(size_t)i * m->max_nod_in_elem] - 1];
    for (int k = 1; k < num; k++) { const double a = s->alpha_evp_array[m->nod_in_elem2D[(size_t)i * m->max_nod_in_elem + k] - 1]; b = a > b ? a : b; }
    s->beta_evp_array[i] = b;
  }
  free(ua); free(va); free(rhs_a); free(rhs_m); free(urhs); free(vrhs); free(bnd);
  return 0;
}

/* Classic EVP, EVPdynamics (src/ice_EVP.F90:397-667; whichEVP = 0, the default of namelist.ice): inverse masses, ice strength and the sea-surface-slope term
 * (:448-541), per subcycle stress_tensor (:23-134), stress2rhs (:323-396) and the node update (:556-585; the velocities are updated in place), coastal nodes.
 * One partition.  Pinned bitwise on a run of the reference's own routine (tests/golden/ice_evp0_reference.npz, tests/test_ice.py). */
int orc_ice_evp0(const fesom_mesh_desc *m, const fesom_ice_params *p, fesom_ice_state *s) {
  const int myN = m->myDim_nod2D, N = myN + m->eDim_nod2D, myE = m->myDim_elem2D, nl = m->nl;
  const double rdt = p->ice_dt / (1.0 * p->evp_rheol_steps), ax = cos(p->theta_io), ay = sin(p->theta_io);
  const double vale = 1.0 / (p->ellipse * p->ellipse), dte = p->ice_dt / (1.0 * p->evp_rheol_steps);
  const double det1 = 1.0 / (1.0 + 0.5 * p->Tevp_inv * dte), det2 = 1.0 / (1.0 + 0.5 * p->Tevp_inv * dte), val3 = 1 / 3.0;
  double *inv_areamass = calloc(myN, sizeof(double)), *inv_mass = calloc(myN, sizeof(double)), *rhs_a = calloc(N, sizeof(double)), *rhs_m = calloc(N, sizeof(double));
  double *urhs = calloc(N, sizeof(double)), *vrhs = calloc(N, sizeof(double)), *strength = calloc(myE, sizeof(double));
  char *bnd = calloc(N, 1);
  for (int ed = 0; ed < m->myDim_edge2D; ed++)
    if (m->myList_edge2D[ed] > m->edge2D_in) { bnd[m->edges[2 * ed] - 1] = 1; bnd[m->edges[2 * ed + 1] - 1] = 1; }
  for (int n = 0; n < myN; n++) {
    if (m->ulevels_nod2D[n] > 1) continue;
    const double ms = RHOICE * s->m_ice[n] + RHOSNO * s->m_snow[n];
    inv_areamass[n] = ms > 1.e-3 ? 1. / (m->area[(size_t)n * nl] * ms) : 0.;
    if (s->a_ice[n] < 0.01) inv_mass[n] = 0.;
    else { inv_mass[n] = ms / s->a_ice[n]; inv_mass[n] = 1.0 / (inv_mass[n] > 9.0 ? inv_mass[n] : 9.0); }
  }
  const double use_pice = p->use_floatice ? 1.0 : 0.0;      /* (use_floatice of this struct = use_floatice .and. which_ALE /= 'linfs') */
  for (int el = 0; el < myE; el++) {
    if (m->ulevels[el] > 1) continue;
    const int *en = m->elem2D_nodes + 3 * el;
    const double m1 = s->m_ice[en[0] - 1], m2 = s->m_ice[en[1] - 1], m3 = s->m_ice[en[2] - 1], a1 = s->a_ice[en[0] - 1], a2 = s->a_ice[en[1] - 1], a3 = s->a_ice[en[2] - 1];
    if (m1 <= 0. || m2 <= 0. || m3 <= 0. || a1 <= 0. || a2 <= 0. || a3 <= 0.) continue;
    const double msum = ((m1 + m2) + m3) / 3.0, asum = ((a1 + a2) + a3) / 3.0;
    strength[el] = p->Pstar * msum * exp(-p->c_pressure * (1.0 - asum));
    strength[el] = 0.5 * strength[el];
    const double aa = 9.81 * m->elem_area[el] / 3.0;
    const double *gs = m->gradient_sca + 6 * (size_t)el;
    double e3[3];
    for (int k = 0; k < 3; k++) {
      e3[k] = s->elevation[en[k] - 1];
      {                                    /* (the ice load enters through p_ice * use_pice; with which_ALE = 'linfs' use_floatice of this struct is 0) */
        double pi = (RHOICE * s->m_ice[en[k] - 1] + RHOSNO * s->m_snow[en[k] - 1]) * INV_RHOWAT;
        pi = pi < p->max_ice_loading ? pi : p->max_ice_loading;
        e3[k] = e3[k] + pi * use_pice;
      }
    }
    const double ex = (gs[0] * e3[0] + gs[1] * e3[1]) + gs[2] * e3[2], ey = (gs[3] * e3[0] + gs[4] * e3[1]) + gs[5] * e3[2];
    for (int k = 0; k < 3; k++) { rhs_a[en[k] - 1] = rhs_a[en[k] - 1] - aa * ex; rhs_m[en[k] - 1] = rhs_m[en[k] - 1] - aa * ey; }
  }
  for (int n = 0; n < myN; n++) {
    if (m->ulevels_nod2D[n] > 1) continue;
    rhs_a[n] = rhs_a[n] / m->area[(size_t)n * nl]; rhs_m[n] = rhs_m[n] / m->area[(size_t)n * nl];
  }
  double *U = s->u_ice, *V = s->v_ice;
  for (int sub = 0; sub < p->evp_rheol_steps; sub++) {
    for (int el = 0; el < myE; el++) {                        /* stress_tensor */
      if (m->ulevels[el] > 1 || !(strength[el] > 0.)) continue;
      const int *en = m->elem2D_nodes + 3 * el;
      const double *dx = m->gradient_sca + 6 * (size_t)el, *dy = dx + 3, mf = m->metric_factor[el];
      const double u1 = U[en[0] - 1], u2 = U[en[1] - 1], u3 = U[en[2] - 1], v1 = V[en[0] - 1], v2 = V[en[1] - 1], v3 = V[en[2] - 1];
      const double e11 = ((dx[0] * u1 + dx[1] * u2) + dx[2] * u3) - mf * ((v1 + v2) + v3) / 3.0;
      const double e22 = (dy[0] * v1 + dy[1] * v2) + dy[2] * v3;
      const double e12 = 0.5 * ((((dy[0] * u1 + dy[1] * u2) + dy[2] * u3) + ((dx[0] * v1 + dx[1] * v2) + dx[2] * v3)) + mf * ((u1 + u2) + u3) / 3.0);
      const double delta = sqrt((e11 * e11 + e22 * e22) * (1.0 + vale) + 4.0 * vale * e12 * e12 + 2.0 * e11 * e22 * (1.0 - vale));
      const double delta_inv = 1.0 / (delta > p->delta_min ? delta : p->delta_min);
      double zeta = strength[el] * delta_inv;
      zeta = zeta * p->Tevp_inv;
      const double r1 = zeta * (e11 + e22) - strength[el] * p->Tevp_inv, r2 = zeta * (e11 - e22) * vale, r3 = zeta * e12 * vale;
      const double si1 = det1 * (s->sigma11[el] + s->sigma22[el] + dte * r1), si2 = det2 * (s->sigma11[el] - s->sigma22[el] + dte * r2);
      s->sigma12[el] = det2 * (s->sigma12[el] + dte * r3);
      s->sigma11[el] = 0.5 * (si1 + si2);
      s->sigma22[el] = 0.5 * (si1 - si2);
    }
    for (int n = 0; n < myN; n++) { urhs[n] = 0.0; vrhs[n] = 0.0; }      /* stress2rhs */
    for (int el = 0; el < myE; el++) {
      if (m->ulevels[el] > 1 || !(strength[el] > 0.)) continue;
      const int *en = m->elem2D_nodes + 3 * el;
      const double *gs = m->gradient_sca + 6 * (size_t)el, ar = m->elem_area[el], mf = m->metric_factor[el];
      const double s11 = s->sigma11[el], s12 = s->sigma12[el], s22 = s->sigma22[el];
      for (int k = 0; k < 3; k++) {
        urhs[en[k] - 1] = urhs[en[k] - 1] - ar * (s11 * gs[k] + s12 * gs[k + 3] + s12 * val3 * mf);
        vrhs[en[k] - 1] = vrhs[en[k] - 1] - ar * (s12 * gs[k] + s22 * gs[k + 3] - s11 * val3 * mf);
      }
    }
    for (int n = 0; n < myN; n++) {
      if (m->ulevels_nod2D[n] > 1) continue;
      if (inv_areamass[n] > 0.) { urhs[n] = urhs[n] * inv_areamass[n] + rhs_a[n]; vrhs[n] = vrhs[n] * inv_areamass[n] + rhs_m[n]; }
      else { urhs[n] = 0.; vrhs[n] = 0.; }
    }
    for (int n = 0; n < myN; n++) {                           /* node update (:556-585) */
      if (m->ulevels_nod2D[n] > 1) continue;
      if (s->a_ice[n] >= 0.01) {
        const double du = U[n] - s->u_w[n], dv = V[n] - s->v_w[n];
        const double umod = sqrt(du * du + dv * dv);
        const double drag = p->cd_oce_ice * umod * DENSITY_0 * inv_mass[n];
        const double rhsu = U[n] + rdt * (drag * (ax * s->u_w[n] - ay * s->v_w[n]) + inv_mass[n] * s->stress_atmice_x[n] + urhs[n]);
        const double rhsv = V[n] + rdt * (drag * (ax * s->v_w[n] + ay * s->u_w[n]) + inv_mass[n] * s->stress_atmice_y[n] + vrhs[n]);
        const double r_a = 1. + ax * drag * rdt, r_b = rdt * (m->coriolis_node[n] + ay * drag);
        const double det = 1.0 / (r_a * r_a + r_b * r_b);
        U[n] = det * (r_a * rhsu + r_b * rhsv);
        V[n] = det * (r_a * rhsv - r_b * rhsu);
      } else { U[n] = 0.0; V[n] = 0.0; }
    }
    for (int i = 0; i < N; i++) if (bnd[i]) { U[i] = 0.0; V[i] = 0.0; }
  }
  free(inv_areamass); free(inv_mass); free(rhs_a); free(rhs_m); free(urhs); free(vrhs); free(strength); free(bnd);
  return 0;
}

/* ---- FCT advection of m_ice, a_ice, m_snow: ice_TG_rhs_div, ice_fct_solve (ice_solve_high_order, ice_solve_low_order, ice_fem_fct x 3),
 * ice_update_for_div (src/ice_fct.F90) and cut_off (src/ice_thermo_oce.F90:2-63) = the "Advection part" of ice_timestep
 * (src/ice_setup_step.F90:213-232; no __oifs, no cavities).  One partition.  Pinned bitwise on a run of the reference's own routines
 * (tests/golden/ice_adv_reference.npz).  dbg: NULL or 20 arrays of N doubles that receive, in this order, u_ice, v_ice, rhs_m, rhs_a,
 * rhs_ms, rhs_mdiv, rhs_adiv, rhs_msdiv, m_icel, a_icel, m_snowl, dm_ice, da_ice, dm_snow (after ice_fct_solve), m_ice, a_ice, m_snow
 * (after ice_fct_solve) and (after ice_update_for_div). */
static double *mass_matrix_fill(const fesom_mesh_desc *m) {           /* ice_mass_matrix_fill (:634-709) */
  const int myN = m->myDim_nod2D, N = myN + m->eDim_nod2D;
  const int r0 = m->ssh_rowptr[0];
  double *mm = calloc((size_t)(m->ssh_rowptr[myN] - r0), sizeof(double));
  int *col_pos = calloc(N, sizeof(int));
  for (int el = 0; el < m->myDim_elem2D; el++) {
    const int *en = m->elem2D_nodes + 3 * el;
    for (int n = 0; n < 3; n++) {
      const int row = en[n];
      if (row > myN) continue;
      const int off = m->ssh_rowptr[row - 1] - r0, cn = m->ssh_rowptr[row] - m->ssh_rowptr[row - 1];
      for (int q = 0; q < cn; q++) col_pos[m->ssh_colind_loc[off + q] - 1] = q;
      for (int q = 0; q < 3; q++) {
        if (m->ulevels[el] > 1) continue;
        const int ipos = off + col_pos[en[q] - 1];
        mm[ipos] = mm[ipos] + m->elem_area[el] / 12.0;
        if (q == n) mm[ipos] = mm[ipos] + m->elem_area[el] / 12.0;
      }
    }
  }
  free(col_pos);
  return mm;
}
static double mm_row(const fesom_mesh_desc *m, const double *mm, const double *x, int row) {     /* sum(mass_matrix(clo:clo2) * x(nn_pos(1:cn, row))) */
  const int r0 = m->ssh_rowptr[0], a = m->ssh_rowptr[row] - r0, b = m->ssh_rowptr[row + 1] - r0;
  double s = 0.0;
  for (int q = a; q < b; q++) s = s + mm[q] * x[m->ssh_colind_loc[q] - 1];
  return s;
}
/* the three-sweep mass-matrix solve shared by ice_solve_high_order (:239-317) and ice_update_for_div (:804-892) */
static void mm_solve3(const fesom_mesh_desc *m, const double *mm, const double *rhs[3], double *d[3], double *l[3]) {
  const int myN = m->myDim_nod2D, nl = m->nl;
  for (int row = 0; row < myN; row++) {
    if (m->ulevels_nod2D[row] > 1) continue;
    for (int t = 0; t < 3; t++) d[t][row] = rhs[t][row] / m->area[(size_t)row * nl];
  }
  /* (exchange_nod: single partition) */
  for (int n = 1; n <= 2; n++) {
    for (int row = 0; row < myN; row++) {
      if (m->ulevels_nod2D[row] > 1) continue;
      for (int t = 0; t < 3; t++) {
        double rhs_new = rhs[t][row] - mm_row(m, mm, d[t], row);
        l[t][row] = d[t][row] + rhs_new / m->area[(size_t)row * nl];
      }
    }
    for (int row = 0; row < myN; row++) {
      if (m->ulevels_nod2D[row] > 1) continue;
      for (int t = 0; t < 3; t++) d[t][row] = l[t][row];
    }
  }
}
int orc_ice_adv(const fesom_mesh_desc *m, double ice_dt, double gamma, fesom_ice_state *s, double **dbg) {
  const int myN = m->myDim_nod2D, N = myN + m->eDim_nod2D, myE = m->myDim_elem2D, nl = m->nl;
  double *mm = mass_matrix_fill(m);
  double *buf = calloc((size_t)17 * N + 3 * (size_t)myE, sizeof(double));
  double *rhs[3] = {buf, buf + N, buf + 2 * N}, *rdiv[3] = {buf + 3 * N, buf + 4 * N, buf + 5 * N};
  double *lo[3] = {buf + 6 * N, buf + 7 * N, buf + 8 * N}, *d[3] = {buf + 9 * N, buf + 10 * N, buf + 11 * N};
  double *tmax = buf + 12 * N, *tmin = buf + 13 * N, *pplus = buf + 14 * N, *pminus = buf + 15 * N, *flx = buf + 17 * N;
  double *tr[3] = {s->m_ice, s->a_ice, s->m_snow};
  const double *u = s->u_ice, *v = s->v_ice;
#define DBG(i, a) if (dbg && dbg[i]) memcpy(dbg[i], a, sizeof(double) * N)
  DBG(0, u); DBG(1, v);
  /* ice_TG_rhs_div (:713-800) */
  for (int el = 0; el < myE; el++) {
    const int *en = m->elem2D_nodes + 3 * el;
    if (m->ulevels[el] > 1) continue;
    const double *dx = m->gradient_sca + 6 * (size_t)el, *dy = dx + 3, vol = m->elem_area[el];
    const double u3[3] = {u[en[0] - 1], u[en[1] - 1], u[en[2] - 1]}, v3[3] = {v[en[0] - 1], v[en[1] - 1], v[en[2] - 1]};
    const double um = (u3[0] + u3[1]) + u3[2], vm = (v3[0] + v3[1]) + v3[2];
    const double c1 = (um * um + ((u3[0] * u3[0] + u3[1] * u3[1]) + u3[2] * u3[2])) / 12.0;
    const double c2 = (vm * vm + ((v3[0] * v3[0] + v3[1] * v3[1]) + v3[2] * v3[2])) / 12.0;
    const double c3 = (um * vm + ((v3[0] * u3[0] + v3[1] * u3[1]) + v3[2] * u3[2])) / 12.0;
    const double c4 = ((dx[0] * u3[0] + dy[0] * v3[0]) + (dx[1] * u3[1] + dy[1] * v3[1])) + (dx[2] * u3[2] + dy[2] * v3[2]);
    for (int n = 0; n < 3; n++) {
      const int row = en[n] - 1;
      double ent[3], ent2[3];
      for (int q = 0; q < 3; q++) {
        ent[q] = vol * ice_dt * ((1.0 - 0.5 * ice_dt * c4) * (dx[n] * (um + u3[q]) + dy[n] * (vm + v3[q])) / 12.0 -
                                 0.5 * ice_dt * (c1 * dx[n] * dx[q] + c2 * dy[n] * dy[q] + c3 * (dx[n] * dy[q] + dx[q] * dy[n])));
        ent2[q] = 0.5 * ice_dt * (dx[n] * (um + u3[q]) + dy[n] * (vm + v3[q]) - dx[q] * (um + u3[n]) - dy[q] * (vm + v3[n]));
      }
      for (int t = 0; t < 3; t++) {
        const double a3[3] = {tr[t][en[0] - 1], tr[t][en[1] - 1], tr[t][en[2] - 1]};
        const double cx = vol * ice_dt * c4 * ((((a3[0] + a3[1]) + a3[2]) + a3[n]) + ((ent2[0] * a3[0] + ent2[1] * a3[1]) + ent2[2] * a3[2])) / 12.0;
        if (row < myN) {      /* (halo rows of the reference's arrays receive partial sums that nothing reads) */
          rhs[t][row] = (rhs[t][row] + ((ent[0] * a3[0] + ent[1] * a3[1]) + ent[2] * a3[2])) + cx;
          rdiv[t][row] = rdiv[t][row] - cx;
        }
      }
    }
  }
  DBG(2, rhs[0]); DBG(3, rhs[1]); DBG(4, rhs[2]); DBG(5, rdiv[0]); DBG(6, rdiv[1]); DBG(7, rdiv[2]);
  /* ice_fct_solve (:151-169): high order, low order, FCT per tracer */
  { const double *r3[3] = {rhs[0], rhs[1], rhs[2]}; mm_solve3(m, mm, r3, d, lo); }
  for (int row = 0; row < myN; row++) {                                   /* ice_solve_low_order (:173-235) */
    if (m->ulevels_nod2D[row] > 1) continue;
    for (int t = 0; t < 3; t++)
      lo[t][row] = (rhs[t][row] + gamma * mm_row(m, mm, tr[t], row)) / m->area[(size_t)row * nl] + (1.0 - gamma) * tr[t][row];
  }
  for (int t = 0; t < 3; t++) {                                           /* ice_fem_fct(t) (:321-630) */
    for (int el = 0; el < myE; el++) {
      const int *en = m->elem2D_nodes + 3 * el;
      if (m->ulevels[el] > 1) continue;
      const double vol = m->elem_area[el];
      double w[3];
      for (int k = 0; k < 3; k++) w[k] = gamma * tr[t][en[k] - 1] + d[t][en[k] - 1];
      for (int q = 0; q < 3; q++) {
        double sm = 0.0;                                                  /* sum(icoef(:,q) * w): icoef = 1, -2 on the diagonal */
        for (int k = 0; k < 3; k++) sm = sm + (k == q ? -2.0 : 1.0) * w[k];
        flx[3 * (size_t)el + q] = -sm * (vol / m->area[(size_t)(en[q] - 1) * nl]) / 12.0;
      }
    }
    for (int row = 0; row < myN; row++) {
      tmax[row] = 0.0; tmin[row] = 0.0;
      if (m->ulevels_nod2D[row] > 1) continue;
      const int r0 = m->ssh_rowptr[0], a = m->ssh_rowptr[row] - r0, b = m->ssh_rowptr[row + 1] - r0;
      double mx = lo[t][m->ssh_colind_loc[a] - 1], mn = mx;
      for (int q = a + 1; q < b; q++) { double x = lo[t][m->ssh_colind_loc[q] - 1]; if (x > mx) mx = x; if (x < mn) mn = x; }
      tmax[row] = mx - lo[t][row]; tmin[row] = mn - lo[t][row];
    }
    for (int i = 0; i < N; i++) { pplus[i] = 0.0; pminus[i] = 0.0; }
    for (int el = 0; el < myE; el++) {
      const int *en = m->elem2D_nodes + 3 * el;
      if (m->ulevels[el] > 1) continue;
      for (int q = 0; q < 3; q++) {
        const double f = flx[3 * (size_t)el + q];
        if (f > 0) pplus[en[q] - 1] = pplus[en[q] - 1] + f; else pminus[en[q] - 1] = pminus[en[q] - 1] + f;
      }
    }
    for (int n = 0; n < myN; n++) {
      if (m->ulevels_nod2D[n] > 1) continue;
      double f = pplus[n];
      pplus[n] = fabs(f) > 0 ? fmin(1.0, tmax[n] / f) : 0.0;
      f = pminus[n];
      pminus[n] = fabs(f) > 0 ? fmin(1.0, tmin[n] / f) : 0.0;
    }
    /* (exchange_nod(icepminus, icepplus): single partition) */
    for (int el = 0; el < myE; el++) {
      const int *en = m->elem2D_nodes + 3 * el;
      if (m->ulevels[el] > 1) continue;
      double ae = 1.0;
      for (int q = 0; q < 3; q++) {
        const double f = flx[3 * (size_t)el + q];
        if (f >= 0.) ae = fmin(ae, pplus[en[q] - 1]);
        if (f < 0.) ae = fmin(ae, pminus[en[q] - 1]);
      }
      for (int q = 0; q < 3; q++) flx[3 * (size_t)el + q] = ae * flx[3 * (size_t)el + q];
    }
    for (int n = 0; n < myN; n++) { if (m->ulevels_nod2D[n] > 1) continue; tr[t][n] = lo[t][n]; }
    for (int el = 0; el < myE; el++) {
      const int *en = m->elem2D_nodes + 3 * el;
      if (m->ulevels[el] > 1) continue;
      for (int q = 0; q < 3; q++) if (en[q] <= myN) tr[t][en[q] - 1] = tr[t][en[q] - 1] + flx[3 * (size_t)el + q];
    }
    /* (exchange_nod(m_ice, a_ice, m_snow): single partition) */
  }
  DBG(8, lo[0]); DBG(9, lo[1]); DBG(10, lo[2]); DBG(11, d[0]); DBG(12, d[1]); DBG(13, d[2]); DBG(14, tr[0]); DBG(15, tr[1]); DBG(16, tr[2]);
  /* ice_update_for_div (:804-892) */
  { const double *r3[3] = {rdiv[0], rdiv[1], rdiv[2]}; mm_solve3(m, mm, r3, d, lo); }
  for (int t = 0; t < 3; t++) for (int i = 0; i < N; i++) tr[t][i] = tr[t][i] + d[t][i];
  DBG(17, tr[0]); DBG(18, tr[1]); DBG(19, tr[2]);
  /* cut_off (src/ice_thermo_oce.F90:2-63) */
  for (int i = 0; i < N; i++) {
    if (s->a_ice[i] > 1.0) s->a_ice[i] = 1.0;
    if (s->a_ice[i] < 0.1e-8) s->a_ice[i] = 0.0;
    if (s->m_ice[i] < 0.1e-8) s->m_ice[i] = 0.0;
  }
#undef DBG
  free(mm); free(buf);
  return 0;
}
