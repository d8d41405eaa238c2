/* ORACLE (test infrastructure): CPU restatement of the sea-ice mEVP rheology, EVPdynamics_m (src/ice_maEVP.F90:273-602; whichEVP = 1,
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
