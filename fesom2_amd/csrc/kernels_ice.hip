// Sea-ice mEVP rheology on gfx950: EVPdynamics_m of the reference (src/ice_maEVP.F90:273-602; whichEVP = 1, Bouillon et al. 2013 /
// Kimmritz et al. 2015), the subcycled momentum solve of the sea-ice model: evp_rheol_steps (120) subcycles per ice step, each one
//   element loop : strain rates from the nodal velocities, viscous-plastic stresses (implicit relaxation with alpha_evp),
//                  stress divergence scattered to the three nodes
//   node loop    : + sea-surface slope, ocean drag and Coriolis implicitly, relaxation with beta_evp -> new velocities
//   boundary     : zero velocity on the coast; halo exchange.
// MI355X shape: ONE launch per subcycle, thread per node.  The element loop's scatter-add becomes a gather over the node's elements in
// increasing element index (= the order in which the reference's element loop adds to the node), and every node thread evaluates the
// stress update of its elements itself -- each element is updated by up to three threads with identical arithmetic (old stresses and
// old velocities are read from ping-pong buffers, so nobody sees a half-updated state), one designated thread stores the new stress.
// No atomics, deterministic, bit-identical to the CPU checker of the tests and to the reference's own routine
// (tests/test_ice.py).  The 120 launches of a call are captured once into a hipGraph and replayed (2-D problem: launch-bound).
// exp() of the pressure factor is evaluated on the HOST when the ice state is uploaded (glibc, as in the reference's build), the
// only libm call of the routine whose device version could differ in the last bit.
#include "dev.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define ICE_G 9.81
#define ICE_DENSITY_0 1030.0
#define ICE_RHOICE 910.0
#define ICE_RHOSNO 290.0
#define ICE_INV_RHOWAT (1. / 1025.)        // i_therm_param (src/ice_modules.F90)

namespace {
struct IceDM {
  int N, myN, myE, maxk;
  const int *en;             // (3, myE) 0-based
  const int *nie, *nie_num;  // (maxk, N) 0-based elements of a node in increasing index, -1 padded
  const double *gsca, *elem_area, *metric, *area1, *cori_n;
  const unsigned char *bnd;  // coastal nodes
  double *u_ice, *v_ice, *a_ice, *m_ice, *m_snow, *elev, *u_w, *v_w, *tax, *tay;
  double *sig[2];            // [parity] -> 3 * myE (sigma11 | sigma12 | sigma22)
  double *ua[2], *va[2];
  double *rhs_a, *rhs_m, *invt, *mass, *pfac, *efac;
  double *alpha, *beta;      // aEVP (whichEVP = 2): alpha_evp_array (myE), beta_evp_array (N)
  unsigned char *ice_nod, *ice_el;
  // FCT advection (src/ice_fct.F90): CSR pattern of the owned rows (= nn_pos / ssh_stiff, 0-based), consistent mass matrix, work arrays of 3 tracers
  const int *rp, *ci;
  const double *mm;
  double *tr3;               // m_ice | a_ice | m_snow (3 N), the state arrays above are views into it
  double *rhs, *rdiv, *lo, *dA, *dB, *pp;      // 3 N each; pp: icepplus (3 N) | icepminus (3 N)
  double *flx;               // (3 tracers, myE, 3)
  fesom_ice_params p;
};
struct IceCtx {
  bool ready = false;
  IceDM m;
  std::vector<void *> allocs;
  hipStream_t stream = nullptr;
  hipGraphExec_t graph = nullptr;
  int cur = 0;               // parity of the buffers that hold the current stresses
  std::vector<int> h_en; std::vector<double> h_efac;
  std::string err;
  // partition (npes > 1): com_nod2D lists of the rank, packed-message buffers
  int npes = 1;
  std::vector<int> sPE, sptr, rPE, rptr;
  const int *slist = nullptr, *rlist = nullptr;     // device, 0-based
  const int *sptr_d = nullptr, *rptr_d = nullptr;   // device copies of the 1-based block pointers
  int nsend = 0, nrecv = 0;
  double *hsend = nullptr, *hrecv = nullptr;
  std::vector<double> h_aice;                       // staging of a_ice for the host-side exp of the pressure factor
} I;

#define ICECHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { I.err = std::string(#x) + ": " + hipGetErrorString(e_); fprintf(stderr, "fesom_gpu_ice: %s\n", I.err.c_str()); return 1; } } while (0)

template <class T> T *ialloc(size_t n) {
  void *p = nullptr;
  if (hipMalloc(&p, (n ? n : 1) * sizeof(T)) != hipSuccess) return nullptr;
  hipMemset(p, 0, (n ? n : 1) * sizeof(T));
  I.allocs.push_back(p);
  return (T *)p;
}
template <class T> const T *iupload(const std::vector<T> &h) {
  T *p = ialloc<T>(h.size());
  if (p && !h.empty()) hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
  return p;
}

// ssh2rhs inlined (:340-392), thickness / mass (:394-419), start of the solver variables (:323-324)
__global__ void k_ice_prep_node(IceDM m) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m.N) return;
  m.ua[0][i] = m.u_ice[i]; m.va[0][i] = m.v_ice[i];
  if (i >= m.myN) return;
  const double val3 = 1.0 / 3.0;
  double ra = 0.0, rm = 0.0;
  for (int k = 0; k < m.nie_num[i]; k++) {
    const int el = m.nie[(size_t)m.maxk * i + k];
    const int *en = m.en + 3 * el;
    const double *gs = m.gsca + 6 * (size_t)el;
    double e3[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
      e3[q] = m.elev[en[q]];
      if (m.p.use_floatice) {
        double pi = (ICE_RHOICE * m.m_ice[en[q]] + ICE_RHOSNO * m.m_snow[en[q]]) * ICE_INV_RHOWAT;
        pi = pi < m.p.max_ice_loading ? pi : m.p.max_ice_loading;
        e3[q] = e3[q] + pi;
      }
    }
    double bb = ICE_G * val3 * m.elem_area[el];
    const double aa = bb * ((gs[0] * e3[0] + gs[1] * e3[1]) + gs[2] * e3[2]);
    bb = bb * ((gs[3] * e3[0] + gs[4] * e3[1]) + gs[5] * e3[2]);
    ra = ra - aa; rm = rm - bb;
  }
  double invt = 0.0, mass = 0.0;
  unsigned char on = 0;
  if (m.a_ice[i] >= 0.01) {
    double it = (ICE_RHOICE * m.m_ice[i] + ICE_RHOSNO * m.m_snow[i]) / m.a_ice[i];
    invt = 1.0 / (it > 9.0 ? it : 9.0);
    const double ms = (m.m_ice[i] * ICE_RHOICE + m.m_snow[i] * ICE_RHOSNO);
    mass = ms / ((1.0 + ms * ms) * m.area1[i]);
    ra = ra / m.area1[i]; rm = rm / m.area1[i];
    on = 1;
  }
  m.rhs_a[i] = ra; m.rhs_m[i] = rm; m.invt[i] = invt; m.mass[i] = mass; m.ice_nod[i] = on;
}
// pressure factor (:421-438); exp(-c_pressure (1 - asum)) comes from the host (efac)
__global__ void k_ice_prep_elem(IceDM m) {
  const int el = blockIdx.x * blockDim.x + threadIdx.x;
  if (el >= m.myE) return;
  const int *en = m.en + 3 * el;
  const double val3 = 1.0 / 3.0, det2 = 1.0 / (1.0 + m.p.alpha_evp);
  const double msum = ((m.m_ice[en[0]] + m.m_ice[en[1]]) + m.m_ice[en[2]]) * val3;
  double pf = 0.0;
  unsigned char on = 0;
  if (msum > 0.01) { on = 1; pf = det2 * m.p.Pstar * msum * m.efac[el]; }
  m.pfac[el] = pf; m.ice_el[el] = on;
}
// one subcycle (:452-590): stresses of the node's elements from the buffers of parity `par`, new stresses / velocities to parity 1 - par
__global__ void __launch_bounds__(128) k_ice_sub(IceDM m, int par) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m.N) return;
  const double *uo = m.ua[par], *vo = m.va[par];
  double *un = m.ua[1 - par], *vn = m.va[1 - par];
  if (i >= m.myN) { un[i] = uo[i]; vn[i] = vo[i]; return; }       // (halo values arrive by exchange in a partitioned run)
  const double *so = m.sig[par];
  double *sn = m.sig[1 - par];
  const size_t E = (size_t)m.myE;
  const double val3 = 1.0 / 3.0, vale = 1.0 / (m.p.ellipse * m.p.ellipse);
  const double det2 = 1.0 / (1.0 + m.p.alpha_evp), det1 = m.p.alpha_evp * det2, rdt = m.p.ice_dt;
  double urhs = 0.0, vrhs = 0.0;
  for (int k = 0; k < m.nie_num[i]; k++) {
    const int el = m.nie[(size_t)m.maxk * i + k];
    const int n1 = m.en[3 * el], n2 = m.en[3 * el + 1], n3 = m.en[3 * el + 2];
    const bool writer = (n1 < m.myN) ? (n1 == i) : ((n2 < m.myN) ? (n2 == i) : (n3 == i));   // the element's first owned node stores
    double s11 = so[el], s12 = so[E + el], s22 = so[2 * E + el];
    if (m.ice_el[el]) {
      const double *dx = m.gsca + 6 * (size_t)el, *dy = dx + 3;
      const double meancos = val3 * m.metric[el];
      const double u1 = uo[n1], u2 = uo[n2], u3 = uo[n3], v1 = vo[n1], v2 = vo[n2], v3 = vo[n3];
      const double eps11 = ((dx[0] * u1 + dx[1] * u2) + dx[2] * u3) - ((v1 + v2) + v3) * meancos;
      const double eps22 = (dy[0] * v1 + dy[1] * v2) + dy[2] * v3;
      const double eps12 = 0.5 * ((((dy[0] * u1 + dx[0] * v1) + (dy[1] * u2 + dx[1] * v2)) + (dy[2] * u3 + dx[2] * v3)) + ((u1 + u2) + u3) * meancos);
      const double eps1 = eps11 + eps22, eps2 = eps11 - eps22;
      const double delta = sqrt(eps1 * eps1 + vale * (eps2 * eps2 + 4.0 * (eps12 * eps12)));
      const double pressure = m.pfac[el] / (delta + m.p.delta_min);
      s12 = det1 * s12 + pressure * eps12 * vale;
      s11 = det1 * s11 + 0.5 * pressure * (eps1 - delta + eps2 * vale);
      s22 = det1 * s22 + 0.5 * pressure * (eps1 - delta - eps2 * vale);
      const int pos = (n1 == i) ? 0 : ((n2 == i) ? 1 : 2);
      const double ar = m.elem_area[el];
      urhs = urhs - ar * (s11 * dx[pos] + s12 * (dy[pos] + meancos));
      vrhs = vrhs - ar * (s12 * dx[pos] + s22 * dy[pos] - s11 * meancos);
    }
    if (writer) { sn[el] = s11; sn[E + el] = s12; sn[2 * E + el] = s22; }
  }
  double ua = uo[i], va = vo[i];
  if (m.ice_nod[i]) {
    urhs = urhs * m.mass[i] + m.rhs_a[i];
    vrhs = vrhs * m.mass[i] + m.rhs_m[i];
    const double uw = m.u_w[i], vw = m.v_w[i], invt = m.invt[i];
    const double du = ua - uw, dv = va - vw;
    const double umod = sqrt(du * du + dv * dv);
    const double drag = rdt * m.p.cd_oce_ice * umod * ICE_DENSITY_0 * invt;
    const double rhsu = m.u_ice[i] + drag * uw + rdt * (invt * m.tax[i] + urhs) + m.p.beta_evp * ua;
    const double rhsv = m.v_ice[i] + drag * vw + rdt * (invt * m.tay[i] + vrhs) + m.p.beta_evp * va;
    const double bd = 1.0 + m.p.beta_evp + drag, rc = rdt * m.cori_n[i];
    const double det = (m.bnd[i] ? 0.0 : 1.0) / (bd * bd + rc * rc);
    ua = det * (bd * rhsu + rc * rhsv);
    va = det * (bd * rhsv - rc * rhsu);
  }
  if (m.bnd[i]) { ua = 0.0; va = 0.0; }
  un[i] = ua; vn[i] = va;
}
// The same subcycle with EIGHT lanes per node: lane (node, k) evaluates element k of the node, so the dependent loads of a node's six
// elements (element id -> nodes -> velocities / stresses) are issued side by side instead of one element after the other, and the
// stress-divergence terms are then subtracted in element order by the node's first lane (shuffles inside the 8-lane group): the same
// operations in the same order as k_ice_sub, hence the same bits; 8.9 -> ~3 us per subcycle on pi.  Nodes with more than 8 elements
// (none on pi, rare elsewhere) take further rounds.
__global__ void __launch_bounds__(256) k_ice_sub8(IceDM m, int par) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = t >> 3, k0 = t & 7;
  const bool live = i < m.N;
  const double *uo = m.ua[par], *vo = m.va[par];
  double *un = m.ua[1 - par], *vn = m.va[1 - par];
  const bool owned = live && i < m.myN;
  const double *so = m.sig[par];
  double *sn = m.sig[1 - par];
  const size_t E = (size_t)m.myE;
  const double val3 = 1.0 / 3.0, vale = 1.0 / (m.p.ellipse * m.p.ellipse);
  const double det2 = 1.0 / (1.0 + m.p.alpha_evp), det1 = m.p.alpha_evp * det2, rdt = m.p.ice_dt;
  const int num = owned ? m.nie_num[i] : 0;
  int nmax = num;                                          // rounds of the group = of the wave (shuffles need every lane)
  for (int sft = 32; sft >= 1; sft >>= 1) nmax = max(nmax, __shfl_xor(nmax, sft, 64));
  double urhs = 0.0, vrhs = 0.0;
  for (int base = 0; base < nmax; base += 8) {
    const int k = base + k0;
    double tu = 0.0, tv = 0.0;
    if (k < num) {
      const int el = m.nie[(size_t)m.maxk * i + k];
      const int n1 = m.en[3 * el], n2 = m.en[3 * el + 1], n3 = m.en[3 * el + 2];
      const bool writer = (n1 < m.myN) ? (n1 == i) : ((n2 < m.myN) ? (n2 == i) : (n3 == i));
      double s11 = so[el], s12 = so[E + el], s22 = so[2 * E + el];
      if (m.ice_el[el]) {
        const double *dx = m.gsca + 6 * (size_t)el, *dy = dx + 3;
        const double meancos = val3 * m.metric[el];
        const double u1 = uo[n1], u2 = uo[n2], u3 = uo[n3], v1 = vo[n1], v2 = vo[n2], v3 = vo[n3];
        const double eps11 = ((dx[0] * u1 + dx[1] * u2) + dx[2] * u3) - ((v1 + v2) + v3) * meancos;
        const double eps22 = (dy[0] * v1 + dy[1] * v2) + dy[2] * v3;
        const double eps12 = 0.5 * ((((dy[0] * u1 + dx[0] * v1) + (dy[1] * u2 + dx[1] * v2)) + (dy[2] * u3 + dx[2] * v3)) + ((u1 + u2) + u3) * meancos);
        const double eps1 = eps11 + eps22, eps2 = eps11 - eps22;
        const double delta = sqrt(eps1 * eps1 + vale * (eps2 * eps2 + 4.0 * (eps12 * eps12)));
        const double pressure = m.pfac[el] / (delta + m.p.delta_min);
        s12 = det1 * s12 + pressure * eps12 * vale;
        s11 = det1 * s11 + 0.5 * pressure * (eps1 - delta + eps2 * vale);
        s22 = det1 * s22 + 0.5 * pressure * (eps1 - delta - eps2 * vale);
        const int pos = (n1 == i) ? 0 : ((n2 == i) ? 1 : 2);
        const double ar = m.elem_area[el];
        tu = ar * (s11 * dx[pos] + s12 * (dy[pos] + meancos));
        tv = ar * (s12 * dx[pos] + s22 * dy[pos] - s11 * meancos);
      }
      if (writer) { sn[el] = s11; sn[E + el] = s12; sn[2 * E + el] = s22; }
    }
    // ordered subtraction by the group's first lane: urhs = (((urhs - t_0) - t_1) - ...); a slot without an (ice) element holds +0.0
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const double qu = __shfl(tu, q, 8), qv = __shfl(tv, q, 8);
      urhs = urhs - qu; vrhs = vrhs - qv;
    }
  }
  if (!live || k0 != 0) return;
  if (!owned) { un[i] = uo[i]; vn[i] = vo[i]; return; }
  double ua = uo[i], va = vo[i];
  if (m.ice_nod[i]) {
    urhs = urhs * m.mass[i] + m.rhs_a[i];
    vrhs = vrhs * m.mass[i] + m.rhs_m[i];
    const double uw = m.u_w[i], vw = m.v_w[i], invt = m.invt[i];
    const double du = ua - uw, dv = va - vw;
    const double umod = sqrt(du * du + dv * dv);
    const double drag = rdt * m.p.cd_oce_ice * umod * ICE_DENSITY_0 * invt;
    const double rhsu = m.u_ice[i] + drag * uw + rdt * (invt * m.tax[i] + urhs) + m.p.beta_evp * ua;
    const double rhsv = m.v_ice[i] + drag * vw + rdt * (invt * m.tay[i] + vrhs) + m.p.beta_evp * va;
    const double bd = 1.0 + m.p.beta_evp + drag, rc = rdt * m.cori_n[i];
    const double det = (m.bnd[i] ? 0.0 : 1.0) / (bd * bd + rc * rc);
    ua = det * (bd * rhsu + rc * rhsv);
    va = det * (bd * rhsv - rc * rhsu);
  }
  if (m.bnd[i]) { ua = 0.0; va = 0.0; }
  un[i] = ua; vn[i] = va;
}
__global__ void k_ice_finish(IceDM m, int par) {             // u_ice = u_ice_aux (:599-600)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m.N) return;
  m.u_ice[i] = m.ua[par][i]; m.v_ice[i] = m.va[par][i];
}

// halo of (u_ice_aux, v_ice_aux) after a subcycle (exchange_nod_begin / _end, ice_maEVP.F90:588-596): per neighbour the u items then the v items
__global__ void k_ice_pack(const double *__restrict__ u, const double *__restrict__ v, const int *__restrict__ list, const int *__restrict__ ptr, int npe, int nitems,
                           double *__restrict__ buf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nitems) return;
  int p = 0;
  while (p + 1 < npe && i >= ptr[p + 1] - 1) p++;
  const int first = ptr[p] - 1, cnt = ptr[p + 1] - ptr[p];
  buf[(size_t)first * 2 + (i - first)] = u[list[i]];
  buf[(size_t)first * 2 + cnt + (i - first)] = v[list[i]];
}
__global__ void k_ice_unpack(double *__restrict__ u, double *__restrict__ v, const int *__restrict__ list, const int *__restrict__ ptr, int npe, int nitems,
                             const double *__restrict__ buf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nitems) return;
  int p = 0;
  while (p + 1 < npe && i >= ptr[p + 1] - 1) p++;
  const int first = ptr[p] - 1, cnt = ptr[p + 1] - ptr[p];
  u[list[i]] = buf[(size_t)first * 2 + (i - first)];
  v[list[i]] = buf[(size_t)first * 2 + cnt + (i - first)];
}


// ====================================================================================================================
// FCT advection of m_ice, a_ice, m_snow: ice_TG_rhs_div, ice_fct_solve (ice_solve_high_order, ice_solve_low_order, ice_fem_fct x 3),
// ice_update_for_div (src/ice_fct.F90) and cut_off (src/ice_thermo_oce.F90:2-63) = the "Advection part" of ice_timestep
// (src/ice_setup_step.F90:213-232).  Thread per node / per element, the three tracers side by side (they do not depend on each other);
// the element loops' scatter-adds are gathers over the node's elements in increasing element index, each node thread forms the
// element terms it needs itself (same arithmetic in up to three threads); the mass-matrix sweeps ping-pong between two buffers.
// 8 launches per step on one partition.
// ====================================================================================================================
__device__ __forceinline__ double ice_mm_row(const IceDM &m, const double *x, int row) {       // sum(mass_matrix(clo:clo2) * x(nn_pos(1:cn, row)))
  double s = 0.0;
  for (int q = m.rp[row]; q < m.rp[row + 1]; q++) s = s + m.mm[q] * x[m.ci[q]];
  return s;
}
// ice_TG_rhs_div (:713-800) + the starts of ice_solve_high_order (:255-264) and ice_solve_low_order (:189-212)
__global__ void k_ice_adv_tg(IceDM m, double gamma) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m.myN) return;
  const double dt = m.p.ice_dt;
  const size_t N = m.N;
  double rhs[3] = {0.0, 0.0, 0.0}, rdv[3] = {0.0, 0.0, 0.0};
  for (int k = 0; k < m.nie_num[i]; k++) {
    const int el = m.nie[(size_t)m.maxk * i + k];
    const int *en = m.en + 3 * el;
    const double *dx = m.gsca + 6 * (size_t)el, *dy = dx + 3, vol = m.elem_area[el];
    const int n = (en[0] == i) ? 0 : ((en[1] == i) ? 1 : 2);
    const double u3[3] = {m.u_ice[en[0]], m.u_ice[en[1]], m.u_ice[en[2]]}, v3[3] = {m.v_ice[en[0]], m.v_ice[en[1]], m.v_ice[en[2]]};
    const double um = (u3[0] + u3[1]) + u3[2], vm = (v3[0] + v3[1]) + v3[2];
    const double c1 = (um * um + ((u3[0] * u3[0] + u3[1] * u3[1]) + u3[2] * u3[2])) / 12.0;
    const double c2 = (vm * vm + ((v3[0] * v3[0] + v3[1] * v3[1]) + v3[2] * v3[2])) / 12.0;
    const double c3 = (um * vm + ((v3[0] * u3[0] + v3[1] * u3[1]) + v3[2] * u3[2])) / 12.0;
    const double c4 = ((dx[0] * u3[0] + dy[0] * v3[0]) + (dx[1] * u3[1] + dy[1] * v3[1])) + (dx[2] * u3[2] + dy[2] * v3[2]);
    double ent[3], ent2[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
      ent[q] = vol * dt * ((1.0 - 0.5 * dt * c4) * (dx[n] * (um + u3[q]) + dy[n] * (vm + v3[q])) / 12.0 -
                           0.5 * dt * (c1 * dx[n] * dx[q] + c2 * dy[n] * dy[q] + c3 * (dx[n] * dy[q] + dx[q] * dy[n])));
      ent2[q] = 0.5 * dt * (dx[n] * (um + u3[q]) + dy[n] * (vm + v3[q]) - dx[q] * (um + u3[n]) - dy[q] * (vm + v3[n]));
    }
#pragma unroll
    for (int t = 0; t < 3; t++) {
      const double *tr = m.tr3 + t * N;
      const double a3[3] = {tr[en[0]], tr[en[1]], tr[en[2]]};
      const double cx = vol * dt * c4 * ((((a3[0] + a3[1]) + a3[2]) + a3[n]) + ((ent2[0] * a3[0] + ent2[1] * a3[1]) + ent2[2] * a3[2])) / 12.0;
      rhs[t] = (rhs[t] + ((ent[0] * a3[0] + ent[1] * a3[1]) + ent[2] * a3[2])) + cx;
      rdv[t] = rdv[t] - cx;
    }
  }
  const double ar = m.area1[i];
#pragma unroll
  for (int t = 0; t < 3; t++) {
    const double *tr = m.tr3 + t * N;
    m.rhs[t * N + i] = rhs[t]; m.rdiv[t * N + i] = rdv[t];
    m.dA[t * N + i] = rhs[t] / ar;
    m.lo[t * N + i] = (rhs[t] + gamma * ice_mm_row(m, tr, i)) / ar + (1.0 - gamma) * tr[i];
  }
}
// one sweep of the mass-matrix iteration (:273-308, :840-878): dst = src + (R - M src) / area.  FIN: the last sweep of ice_update_for_div on one
// partition, followed by m_ice = m_ice + dm_ice (:880-882) and cut_off
template <bool FIN>
__global__ void k_ice_adv_sweep(IceDM m, const double *__restrict__ R, const double *__restrict__ src, double *__restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m.myN) return;
  const size_t N = m.N;
  double v[3];
#pragma unroll
  for (int t = 0; t < 3; t++) {
    const double rn = R[t * N + i] - ice_mm_row(m, src + t * N, i);
    v[t] = src[t * N + i] + rn / m.area1[i];
    dst[t * N + i] = v[t];
  }
  if (FIN) {
    double mi = m.tr3[i] + v[0], ai = m.tr3[N + i] + v[1];
    m.tr3[2 * N + i] = m.tr3[2 * N + i] + v[2];
    if (ai > 1.0) ai = 1.0;
    if (ai < 0.1e-8) ai = 0.0;
    if (mi < 0.1e-8) mi = 0.0;
    m.tr3[i] = mi; m.tr3[N + i] = ai;
  }
}
// partitions: m_ice = m_ice + dm_ice on owned AND halo nodes (the increments come in by exchange), then cut_off
__global__ void k_ice_adv_add_cut(IceDM m, const double *__restrict__ d) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m.N) return;
  const size_t N = m.N;
  double mi = m.tr3[i] + d[i], ai = m.tr3[N + i] + d[N + i];
  m.tr3[2 * N + i] = m.tr3[2 * N + i] + d[2 * N + i];
  if (ai > 1.0) ai = 1.0;
  if (ai < 0.1e-8) ai = 0.0;
  if (mi < 0.1e-8) mi = 0.0;
  m.tr3[i] = mi; m.tr3[N + i] = ai;
}
// antidiffusive element fluxes of ice_fem_fct (:345-380)
__global__ void k_ice_adv_flux(IceDM m, double gamma, const double *__restrict__ d) {
  const int el = blockIdx.x * blockDim.x + threadIdx.x;
  if (el >= m.myE) return;
  const int *en = m.en + 3 * el;
  const size_t N = m.N;
  const double vol = m.elem_area[el];
#pragma unroll
  for (int t = 0; t < 3; t++) {
    const double *tr = m.tr3 + t * N;
    double w[3];
#pragma unroll
    for (int k = 0; k < 3; k++) w[k] = gamma * tr[en[k]] + d[t * N + en[k]];
#pragma unroll
    for (int q = 0; q < 3; q++) {
      double sm = 0.0;
#pragma unroll
      for (int k = 0; k < 3; k++) sm = sm + (k == q ? -2.0 : 1.0) * w[k];
      m.flx[((size_t)t * m.myE + el) * 3 + q] = -sm * (vol / m.area1[en[q]]) / 12.0;
    }
  }
}
// admissible increments and the sums of the positive / negative fluxes into a node -> limiting factors (:381-467)
__global__ void k_ice_adv_lim(IceDM m) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m.myN) return;
  const size_t N = m.N;
#pragma unroll
  for (int t = 0; t < 3; t++) {
    const double *lo = m.lo + t * N;
    double mx = lo[m.ci[m.rp[i]]], mn = mx;
    for (int q = m.rp[i] + 1; q < m.rp[i + 1]; q++) { const double x = lo[m.ci[q]]; if (x > mx) mx = x; if (x < mn) mn = x; }
    const double tmax = mx - lo[i], tmin = mn - lo[i];
    double pp = 0.0, pm = 0.0;
    for (int k = 0; k < m.nie_num[i]; k++) {
      const int el = m.nie[(size_t)m.maxk * i + k];
      const int *en = m.en + 3 * el;
      const int n = (en[0] == i) ? 0 : ((en[1] == i) ? 1 : 2);
      const double f = m.flx[((size_t)t * m.myE + el) * 3 + n];
      if (f > 0) pp = pp + f; else pm = pm + f;
    }
    m.pp[t * N + i] = fabs(pp) > 0 ? fmin(1.0, tmax / pp) : 0.0;
    m.pp[(3 + t) * N + i] = fabs(pm) > 0 ? fmin(1.0, tmin / pm) : 0.0;
  }
}
// limited fluxes added to the low-order solution (:468-600) + the start of ice_update_for_div (:820-828)
__global__ void k_ice_adv_upd(IceDM m, double *__restrict__ d0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m.myN) return;
  const size_t N = m.N;
#pragma unroll
  for (int t = 0; t < 3; t++) {
    double v = m.lo[t * N + i];
    for (int k = 0; k < m.nie_num[i]; k++) {
      const int el = m.nie[(size_t)m.maxk * i + k];
      const int *en = m.en + 3 * el;
      const double *f3 = m.flx + ((size_t)t * m.myE + el) * 3;
      double ae = 1.0;
#pragma unroll
      for (int q = 0; q < 3; q++) {
        if (f3[q] >= 0.) ae = fmin(ae, m.pp[t * N + en[q]]);
        if (f3[q] < 0.) ae = fmin(ae, m.pp[(3 + t) * N + en[q]]);
      }
      const int n = (en[0] == i) ? 0 : ((en[1] == i) ? 1 : 2);
      v = v + ae * f3[n];
    }
    m.tr3[t * N + i] = v;
    d0[t * N + i] = m.rdiv[t * N + i] / m.area1[i];
  }
}
// halo messages of W node fields that lie N apart (base + f N): per neighbour the items of field 0, then of field 1, ...
__global__ void k_ice_packw(const double *__restrict__ base, size_t N, int W, const int *__restrict__ list, const int *__restrict__ ptr, int npe, int nitems,
                            double *__restrict__ buf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nitems) return;
  int p = 0;
  while (p + 1 < npe && i >= ptr[p + 1] - 1) p++;
  const int first = ptr[p] - 1, cnt = ptr[p + 1] - ptr[p];
  for (int f = 0; f < W; f++) buf[(size_t)first * W + (size_t)f * cnt + (i - first)] = base[f * N + list[i]];
}
__global__ void k_ice_unpackw(double *__restrict__ base, size_t N, int W, const int *__restrict__ list, const int *__restrict__ ptr, int npe, int nitems,
                              const double *__restrict__ buf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nitems) return;
  int p = 0;
  while (p + 1 < npe && i >= ptr[p + 1] - 1) p++;
  const int first = ptr[p] - 1, cnt = ptr[p + 1] - ptr[p];
  for (int f = 0; f < W; f++) base[f * N + list[i]] = buf[(size_t)first * W + (size_t)f * cnt + (i - first)];
}

// ---- adaptive EVP, EVPdynamics_a (src/ice_maEVP.F90:785-888; whichEVP = 2, one partition).  Not the fused order of EVPdynamics_m: every sum and product as
// ssh2rhs / stress_tensor_a / stress2rhs_m / the node loop / find_alpha_field_a / find_beta_field_a write them.  Two launches per subcycle: the element stresses
// (in place, each element is independent), then the node gather over the node's elements in increasing index (= the order of the reference's scatter) + update.
__global__ void k_ice_a_prep(IceDM m) {                      // u_ice_aux = u_ice; ssh2rhs (:130-202) as a node gather
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m.N) return;
  m.ua[0][i] = m.u_ice[i]; m.va[0][i] = m.v_ice[i];
  if (i >= m.myN) return;
  const double val3 = 1.0 / 3.0;
  double ra = 0.0, rm = 0.0;
  for (int k = 0; k < m.nie_num[i]; k++) {
    const int el = m.nie[(size_t)m.maxk * i + k];
    const int *en = m.en + 3 * el;
    const double *gs = m.gsca + 6 * (size_t)el;
    double e3[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
      e3[q] = m.elev[en[q]];
      if (m.p.use_floatice) {
        double pi = (ICE_RHOICE * m.m_ice[en[q]] + ICE_RHOSNO * m.m_snow[en[q]]) * ICE_INV_RHOWAT;
        pi = pi < m.p.max_ice_loading ? pi : m.p.max_ice_loading;
        e3[q] = e3[q] + pi;
      }
    }
    double bb = ICE_G * val3 * m.elem_area[el];
    const double aa = bb * ((gs[0] * e3[0] + gs[1] * e3[1]) + gs[2] * e3[2]);
    bb = bb * ((gs[3] * e3[0] + gs[4] * e3[1]) + gs[5] * e3[2]);
    ra = ra - aa; rm = rm - bb;
  }
  m.rhs_a[i] = ra; m.rhs_m[i] = rm;
}
// strain rates of stress_tensor_a / find_alpha_field_a (:719-741, :644-662); returns false where the element carries no ice (msum <= 0.01)
__device__ __forceinline__ bool ice_a_strain(const IceDM &m, int el, const double *uo, const double *vo, double &eps1, double &eps2, double &eps12, double &delta, double &msum) {
  const int *en = m.en + 3 * el;
  const double val3 = 1.0 / 3.0, vale = 1.0 / (m.p.ellipse * m.p.ellipse);
  msum = ((m.m_ice[en[0]] + m.m_ice[en[1]]) + m.m_ice[en[2]]) * val3;
  if (msum <= 0.01) return false;
  const double *dx = m.gsca + 6 * (size_t)el, *dy = dx + 3;
  const double u1 = uo[en[0]], u2 = uo[en[1]], u3 = uo[en[2]], v1 = vo[en[0]], v2 = vo[en[1]], v3 = vo[en[2]];
  const double vsum = (v1 + v2) + v3, usum = (u1 + u2) + u3, meancos = m.metric[el];
  double eps11 = (dx[0] * u1 + dx[1] * u2) + dx[2] * u3;
  eps11 = eps11 - val3 * vsum * meancos;
  const double eps22 = (dy[0] * v1 + dy[1] * v2) + dy[2] * v3;
  eps12 = 0.5 * (((dy[0] * u1 + dx[0] * v1) + (dy[1] * u2 + dx[1] * v2)) + (dy[2] * u3 + dx[2] * v3));
  eps12 = eps12 + 0.5 * val3 * usum * meancos;
  eps1 = eps11 + eps22; eps2 = eps11 - eps22;
  delta = eps1 * eps1 + vale * (eps2 * eps2 + 4.0 * (eps12 * eps12));
  delta = sqrt(delta);
  return true;
}
__global__ void k_ice_a_stress(IceDM m, int par, int sp) {    // stress_tensor_a; sp = parity of the stress buffers (in place)
  const int el = blockIdx.x * blockDim.x + threadIdx.x;
  if (el >= m.myE) return;
  double eps1, eps2, eps12, delta, msum;
  if (!ice_a_strain(m, el, m.ua[par], m.va[par], eps1, eps2, eps12, delta, msum)) return;
  const size_t E = (size_t)m.myE;
  double *sg = m.sig[sp];
  const double vale = 1.0 / (m.p.ellipse * m.p.ellipse), alpha = m.alpha[el];
  const double det2 = 1.0 / (1.0 + alpha), det1 = alpha * det2;
  const double pressure = m.p.Pstar * msum * m.efac[el] / (delta + m.p.delta_min);      // (exp(-c_pressure (1 - asum)) from the host: efac)
  const double r1 = pressure * (eps1 - delta), r2 = pressure * eps2 * vale, r3 = pressure * eps12 * vale;
  double si1 = sg[el] + sg[2 * E + el], si2 = sg[el] - sg[2 * E + el];
  si1 = det1 * si1 + det2 * r1;
  si2 = det1 * si2 + det2 * r2;
  sg[E + el] = det1 * sg[E + el] + det2 * r3;
  sg[el] = 0.5 * (si1 + si2);
  sg[2 * E + el] = 0.5 * (si1 - si2);
}
__global__ void k_ice_a_node(IceDM m, int par, int sp) {      // stress2rhs_m (:206-272) + the node update (:831-856) + coastal nodes (:860-867)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m.N) return;
  const double *uo = m.ua[par], *vo = m.va[par];
  double *un = m.ua[1 - par], *vn = m.va[1 - par];
  if (i >= m.myN) { un[i] = uo[i]; vn[i] = vo[i]; return; }
  const size_t E = (size_t)m.myE;
  const double *sg = m.sig[sp];
  const double val3 = 1.0 / 3.0, rdt = m.p.ice_dt;
  double urhs = 0.0, vrhs = 0.0;
  for (int k = 0; k < m.nie_num[i]; k++) {
    const int el = m.nie[(size_t)m.maxk * i + k];
    const int n1 = m.en[3 * el], n2 = m.en[3 * el + 1], n3 = m.en[3 * el + 2];
    if ((m.a_ice[n1] + m.a_ice[n2]) + m.a_ice[n3] < 0.01) continue;
    const double vol = m.elem_area[el], mf = m.metric[el];
    const double *dx = m.gsca + 6 * (size_t)el, *dy = dx + 3;
    const double s11 = sg[el], s12 = sg[E + el], s22 = sg[2 * E + el];
    const int pos = (n1 == i) ? 0 : ((n2 == i) ? 1 : 2);
    urhs = urhs - vol * (s11 * dx[pos] + s12 * dy[pos]) - vol * s12 * val3 * mf;
    vrhs = vrhs - vol * (s12 * dx[pos] + s22 * dy[pos]) + vol * s11 * val3 * mf;
  }
  double mass = (m.m_ice[i] * ICE_RHOICE + m.m_snow[i] * ICE_RHOSNO);
  mass = mass / (1.0 + mass * mass);
  urhs = (urhs * mass + m.rhs_a[i]) / m.area1[i];
  vrhs = (vrhs * mass + m.rhs_m[i]) / m.area1[i];
  const double ai = m.a_ice[i];
  double thickness = (ICE_RHOICE * m.m_ice[i] + ICE_RHOSNO * m.m_snow[i]) / (ai > 0.01 ? ai : 0.01);
  thickness = thickness > 9.0 ? thickness : 9.0;
  const double inv_thickness = 1.0 / thickness;
  double ua = uo[i], va = vo[i];
  const double uw = m.u_w[i], vw = m.v_w[i];
  const double du = ua - uw, dv = va - vw;
  const double umod = sqrt(du * du + dv * dv);
  const double drag = rdt * m.p.cd_oce_ice * umod * ICE_DENSITY_0 * inv_thickness;
  double rhsu = m.u_ice[i] + drag * uw + rdt * (inv_thickness * m.tax[i] + urhs);
  double rhsv = m.v_ice[i] + drag * vw + rdt * (inv_thickness * m.tay[i] + vrhs);
  const double beta = m.beta[i];
  rhsu = beta * ua + rhsu;
  rhsv = beta * va + rhsv;
  const double fc = rdt * m.cori_n[i];
  double det = (1.0 + beta + drag) * (1.0 + beta + drag) + fc * fc;
  det = (m.bnd[i] ? 0.0 : 1.0) / det;
  ua = det * ((1.0 + beta + drag) * rhsu + fc * rhsv);
  va = det * ((1.0 + beta + drag) * rhsv - fc * rhsu);
  if (m.bnd[i]) { ua = 0.0; va = 0.0; }
  un[i] = ua; vn[i] = va;
}
__global__ void k_ice_a_alpha(IceDM m, int par) {             // find_alpha_field_a (:611-683) on the final velocities
  const int el = blockIdx.x * blockDim.x + threadIdx.x;
  if (el >= m.myE) return;
  double eps1, eps2, eps12, delta, msum;
  if (!ice_a_strain(m, el, m.ua[par], m.va[par], eps1, eps2, eps12, delta, msum)) return;
  const double pressure = m.p.Pstar * m.efac[el] / (delta + m.p.delta_min);
  const double al = sqrt(m.p.ice_dt * m.p.c_aevp * pressure / ICE_RHOICE / m.elem_area[el]);
  m.alpha[el] = al > 50.0 ? al : 50.0;
}
__global__ void k_ice_a_beta(IceDM m) {                       // find_beta_field_a (:892-922): the largest alpha of the node's elements
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m.myN) return;
  double b = m.alpha[m.nie[(size_t)m.maxk * i]];
  for (int k = 1; k < m.nie_num[i]; k++) { const double a = m.alpha[m.nie[(size_t)m.maxk * i + k]]; b = a > b ? a : b; }
  m.beta[i] = b;
}
// ---- classic EVP, EVPdynamics (src/ice_EVP.F90:397-667; whichEVP = 0, the default of namelist.ice).  The velocities are updated in place (a node reads only its
// own old value; the stresses of a subcycle are formed in a launch of their own before).
__device__ __forceinline__ bool ice_c_has_ice(const IceDM &m, const int *en) {      // (:471-474)
  return !(m.m_ice[en[0]] <= 0. || m.m_ice[en[1]] <= 0. || m.m_ice[en[2]] <= 0. || m.a_ice[en[0]] <= 0. || m.a_ice[en[1]] <= 0. || m.a_ice[en[2]] <= 0.);
}
__global__ void k_ice_c_prep_node(IceDM m) {                 // inverse masses (:448-465) and the sea-surface-slope term (:467-541) as a node gather
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m.myN) return;
  const double ms = ICE_RHOICE * m.m_ice[i] + ICE_RHOSNO * m.m_snow[i];
  m.mass[i] = ms > 1.e-3 ? 1. / (m.area1[i] * ms) : 0.;      // inv_areamass
  double im = 0.;
  if (!(m.a_ice[i] < 0.01)) { im = ms / m.a_ice[i]; im = 1.0 / (im > 9.0 ? im : 9.0); }
  m.invt[i] = im;                                            // inv_mass
  const double use_pice = m.p.use_floatice ? 1.0 : 0.0;
  double ra = 0.0, rm = 0.0;
  for (int k = 0; k < m.nie_num[i]; k++) {
    const int el = m.nie[(size_t)m.maxk * i + k];
    const int *en = m.en + 3 * el;
    if (!ice_c_has_ice(m, en)) continue;
    const double *gs = m.gsca + 6 * (size_t)el;
    const double aa = 9.81 * m.elem_area[el] / 3.0;
    double e3[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
      double pi = (ICE_RHOICE * m.m_ice[en[q]] + ICE_RHOSNO * m.m_snow[en[q]]) * ICE_INV_RHOWAT;
      pi = pi < m.p.max_ice_loading ? pi : m.p.max_ice_loading;
      e3[q] = m.elev[en[q]] + pi * use_pice;
    }
    const double ex = (gs[0] * e3[0] + gs[1] * e3[1]) + gs[2] * e3[2], ey = (gs[3] * e3[0] + gs[4] * e3[1]) + gs[5] * e3[2];
    ra = ra - aa * ex; rm = rm - aa * ey;
  }
  m.rhs_a[i] = ra / m.area1[i]; m.rhs_m[i] = rm / m.area1[i];
}
__global__ void k_ice_c_prep_elem(IceDM m) {                 // ice strength (:475-483); exp(-c_pressure (1 - asum)) from the host (efac)
  const int el = blockIdx.x * blockDim.x + threadIdx.x;
  if (el >= m.myE) return;
  const int *en = m.en + 3 * el;
  double st = 0.0;
  if (ice_c_has_ice(m, en)) {
    const double msum = ((m.m_ice[en[0]] + m.m_ice[en[1]]) + m.m_ice[en[2]]) / 3.0;
    st = m.p.Pstar * msum * m.efac[el];
    st = 0.5 * st;
  }
  m.pfac[el] = st;
}
__global__ void k_ice_c_stress(IceDM m, int sp) {             // stress_tensor (:23-134), in place
  const int el = blockIdx.x * blockDim.x + threadIdx.x;
  if (el >= m.myE) return;
  const double strength = m.pfac[el];
  if (!(strength > 0.)) return;
  const size_t E = (size_t)m.myE;
  double *sg = m.sig[sp];
  const int *en = m.en + 3 * el;
  const double *dx = m.gsca + 6 * (size_t)el, *dy = dx + 3, mf = m.metric[el];
  const double vale = 1.0 / (m.p.ellipse * m.p.ellipse), dte = m.p.ice_dt / (1.0 * m.p.evp_rheol_steps);
  const double det1 = 1.0 / (1.0 + 0.5 * m.p.Tevp_inv * dte), det2 = 1.0 / (1.0 + 0.5 * m.p.Tevp_inv * dte);
  const double u1 = m.u_ice[en[0]], u2 = m.u_ice[en[1]], u3 = m.u_ice[en[2]], v1 = m.v_ice[en[0]], v2 = m.v_ice[en[1]], v3 = m.v_ice[en[2]];
  const double e11 = ((dx[0] * u1 + dx[1] * u2) + dx[2] * u3) - mf * ((v1 + v2) + v3) / 3.0;
  const double e22 = (dy[0] * v1 + dy[1] * v2) + dy[2] * v3;
  const double e12 = 0.5 * ((((dy[0] * u1 + dy[1] * u2) + dy[2] * u3) + ((dx[0] * v1 + dx[1] * v2) + dx[2] * v3)) + mf * ((u1 + u2) + u3) / 3.0);
  const double delta = sqrt((e11 * e11 + e22 * e22) * (1.0 + vale) + 4.0 * vale * e12 * e12 + 2.0 * e11 * e22 * (1.0 - vale));
  const double delta_inv = 1.0 / (delta > m.p.delta_min ? delta : m.p.delta_min);
  double zeta = strength * delta_inv;
  zeta = zeta * m.p.Tevp_inv;
  const double r1 = zeta * (e11 + e22) - strength * m.p.Tevp_inv, r2 = zeta * (e11 - e22) * vale, r3 = zeta * e12 * vale;
  const double si1 = det1 * (sg[el] + sg[2 * E + el] + dte * r1), si2 = det2 * (sg[el] - sg[2 * E + el] + dte * r2);
  sg[E + el] = det2 * (sg[E + el] + dte * r3);
  sg[el] = 0.5 * (si1 + si2);
  sg[2 * E + el] = 0.5 * (si1 - si2);
}
__global__ void k_ice_c_node(IceDM m, int sp, double ax, double ay) {      // stress2rhs (:323-396) as a node gather + the node update (:556-585) + coastal nodes
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m.N) return;
  if (i >= m.myN) { if (m.bnd[i]) { m.u_ice[i] = 0.0; m.v_ice[i] = 0.0; } return; }
  const size_t E = (size_t)m.myE;
  const double *sg = m.sig[sp];
  const double val3 = 1 / 3.0, rdt = m.p.ice_dt / (1.0 * m.p.evp_rheol_steps);
  double urhs = 0.0, vrhs = 0.0;
  for (int k = 0; k < m.nie_num[i]; k++) {
    const int el = m.nie[(size_t)m.maxk * i + k];
    if (!(m.pfac[el] > 0.)) continue;
    const int n1 = m.en[3 * el], n2 = m.en[3 * el + 1];
    const int pos = (n1 == i) ? 0 : ((n2 == i) ? 1 : 2);
    const double *gs = m.gsca + 6 * (size_t)el;
    const double ar = m.elem_area[el], mf = m.metric[el], s11 = sg[el], s12 = sg[E + el], s22 = sg[2 * E + el];
    urhs = urhs - ar * (s11 * gs[pos] + s12 * gs[pos + 3] + s12 * val3 * mf);
    vrhs = vrhs - ar * (s12 * gs[pos] + s22 * gs[pos + 3] - s11 * val3 * mf);
  }
  const double iam = m.mass[i];
  if (iam > 0.) { urhs = urhs * iam + m.rhs_a[i]; vrhs = vrhs * iam + m.rhs_m[i]; } else { urhs = 0.; vrhs = 0.; }
  double U = m.u_ice[i], V = m.v_ice[i];
  if (m.a_ice[i] >= 0.01) {
    const double uw = m.u_w[i], vw = m.v_w[i], im = m.invt[i];
    const double du = U - uw, dv = V - vw;
    const double umod = sqrt(du * du + dv * dv);
    const double drag = m.p.cd_oce_ice * umod * ICE_DENSITY_0 * im;
    const double rhsu = U + rdt * (drag * (ax * uw - ay * vw) + im * m.tax[i] + urhs);
    const double rhsv = V + rdt * (drag * (ax * vw + ay * uw) + im * m.tay[i] + vrhs);
    const double r_a = 1. + ax * drag * rdt, r_b = rdt * (m.cori_n[i] + ay * drag);
    const double det = 1.0 / (r_a * r_a + r_b * r_b);
    U = det * (r_a * rhsu + r_b * rhsv);
    V = det * (r_a * rhsv - r_b * rhsu);
  } else { U = 0.0; V = 0.0; }
  if (m.bnd[i]) { U = 0.0; V = 0.0; }
  m.u_ice[i] = U; m.v_ice[i] = V;
}
// The classic subcycle FUSED as k_ice_sub8 is (eight lanes per node, lane (node, k) evaluates element k of the node from the stresses and velocities of parity `par`,
// the element's first owned node stores its new stresses to parity 1 - par, the node's first lane subtracts the stress-divergence terms in element order and updates the
// node): one launch per subcycle instead of two, the dependent loads of a node's elements side by side.  Same operations in the same order as k_ice_c_stress + k_ice_c_node.
__global__ void __launch_bounds__(256) k_ice_c_sub8(IceDM m, int par, double ax, double ay) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = t >> 3, k0 = t & 7;
  const bool live = i < m.N;
  const double *uo = m.ua[par], *vo = m.va[par];
  double *un = m.ua[1 - par], *vn = m.va[1 - par];
  const bool owned = live && i < m.myN;
  const double *so = m.sig[par];
  double *sn = m.sig[1 - par];
  const size_t E = (size_t)m.myE;
  const double val3 = 1 / 3.0, vale = 1.0 / (m.p.ellipse * m.p.ellipse), dte = m.p.ice_dt / (1.0 * m.p.evp_rheol_steps), rdt = dte;
  const double det1 = 1.0 / (1.0 + 0.5 * m.p.Tevp_inv * dte), det2 = 1.0 / (1.0 + 0.5 * m.p.Tevp_inv * dte);
  const int num = owned ? m.nie_num[i] : 0;
  int nmax = num;
  for (int sft = 32; sft >= 1; sft >>= 1) nmax = max(nmax, __shfl_xor(nmax, sft, 64));
  double urhs = 0.0, vrhs = 0.0;
  for (int base = 0; base < nmax; base += 8) {
    const int k = base + k0;
    double tu = 0.0, tv = 0.0;
    if (k < num) {
      const int el = m.nie[(size_t)m.maxk * i + k];
      const int n1 = m.en[3 * el], n2 = m.en[3 * el + 1], n3 = m.en[3 * el + 2];
      const bool writer = (n1 < m.myN) ? (n1 == i) : ((n2 < m.myN) ? (n2 == i) : (n3 == i));
      double s11 = so[el], s12 = so[E + el], s22 = so[2 * E + el];
      const double strength = m.pfac[el];
      if (strength > 0.) {
        const double *dx = m.gsca + 6 * (size_t)el, *dy = dx + 3, mf = m.metric[el];
        const double u1 = uo[n1], u2 = uo[n2], u3 = uo[n3], v1 = vo[n1], v2 = vo[n2], v3 = vo[n3];
        const double e11 = ((dx[0] * u1 + dx[1] * u2) + dx[2] * u3) - mf * ((v1 + v2) + v3) / 3.0;
        const double e22 = (dy[0] * v1 + dy[1] * v2) + dy[2] * v3;
        const double e12 = 0.5 * ((((dy[0] * u1 + dy[1] * u2) + dy[2] * u3) + ((dx[0] * v1 + dx[1] * v2) + dx[2] * v3)) + mf * ((u1 + u2) + u3) / 3.0);
        const double delta = sqrt((e11 * e11 + e22 * e22) * (1.0 + vale) + 4.0 * vale * e12 * e12 + 2.0 * e11 * e22 * (1.0 - vale));
        const double delta_inv = 1.0 / (delta > m.p.delta_min ? delta : m.p.delta_min);
        double zeta = strength * delta_inv;
        zeta = zeta * m.p.Tevp_inv;
        const double r1 = zeta * (e11 + e22) - strength * m.p.Tevp_inv, r2 = zeta * (e11 - e22) * vale, r3 = zeta * e12 * vale;
        const double si1 = det1 * (s11 + s22 + dte * r1), si2 = det2 * (s11 - s22 + dte * r2);
        s12 = det2 * (s12 + dte * r3);
        s11 = 0.5 * (si1 + si2);
        s22 = 0.5 * (si1 - si2);
        const int pos = (n1 == i) ? 0 : ((n2 == i) ? 1 : 2);
        const double *gs = m.gsca + 6 * (size_t)el;
        const double ar = m.elem_area[el];
        tu = ar * (s11 * gs[pos] + s12 * gs[pos + 3] + s12 * val3 * mf);
        tv = ar * (s12 * gs[pos] + s22 * gs[pos + 3] - s11 * val3 * mf);
      }
      if (writer) { sn[el] = s11; sn[E + el] = s12; sn[2 * E + el] = s22; }
    }
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const double qu = __shfl(tu, q, 8), qv = __shfl(tv, q, 8);
      urhs = urhs - qu; vrhs = vrhs - qv;
    }
  }
  if (!live || k0 != 0) return;
  if (!owned) { double U = uo[i], V = vo[i]; if (m.bnd[i]) { U = 0.0; V = 0.0; } un[i] = U; vn[i] = V; return; }
  const double iam = m.mass[i];
  if (iam > 0.) { urhs = urhs * iam + m.rhs_a[i]; vrhs = vrhs * iam + m.rhs_m[i]; } else { urhs = 0.; vrhs = 0.; }
  double U = uo[i], V = vo[i];
  if (m.a_ice[i] >= 0.01) {
    const double uw = m.u_w[i], vw = m.v_w[i], im = m.invt[i];
    const double du = U - uw, dv = V - vw;
    const double umod = sqrt(du * du + dv * dv);
    const double drag = m.p.cd_oce_ice * umod * ICE_DENSITY_0 * im;
    const double rhsu = U + rdt * (drag * (ax * uw - ay * vw) + im * m.tax[i] + urhs);
    const double rhsv = V + rdt * (drag * (ax * vw + ay * uw) + im * m.tay[i] + vrhs);
    const double r_a = 1. + ax * drag * rdt, r_b = rdt * (m.cori_n[i] + ay * drag);
    const double det = 1.0 / (r_a * r_a + r_b * r_b);
    U = det * (r_a * rhsu + r_b * rhsv);
    V = det * (r_a * rhsv - r_b * rhsu);
  } else { U = 0.0; V = 0.0; }
  if (m.bnd[i]) { U = 0.0; V = 0.0; }
  un[i] = U; vn[i] = V;
}
__global__ void k_ice_c_start(IceDM m, int par) {            // the solver copies of the velocities (the fused subcycles alternate between two buffers)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m.N) return;
  m.ua[par][i] = m.u_ice[i]; m.va[par][i] = m.v_ice[i];
}
// one EVPdynamics call, fused subcycles; the stresses start in parity `par`; returns the parity they end in
int enqueue_call_c(hipStream_t s, int par) {
  const IceDM &m = I.m;
  const double ax = cos(m.p.theta_io), ay = sin(m.p.theta_io);      // (host libm, as the reference's)
  hipLaunchKernelGGL(k_ice_c_prep_node, dim3((m.myN + 255) / 256), dim3(256), 0, s, m);
  hipLaunchKernelGGL(k_ice_c_prep_elem, dim3((m.myE + 255) / 256), dim3(256), 0, s, m);
  hipLaunchKernelGGL(k_ice_c_start, dim3((m.N + 255) / 256), dim3(256), 0, s, m, par);
  for (int k = 0; k < m.p.evp_rheol_steps; k++) {
    hipLaunchKernelGGL(k_ice_c_sub8, dim3((unsigned)(((size_t)m.N * 8 + 255) / 256)), dim3(256), 0, s, m, par, ax, ay);
    par = 1 - par;
  }
  hipLaunchKernelGGL(k_ice_finish, dim3((m.N + 255) / 256), dim3(256), 0, s, m, par);
  return par;
}
// The adaptive subcycle fused the same way (k_ice_a_stress + k_ice_a_node in one launch, eight lanes per node).  stress2rhs_m subtracts TWO terms per element
// (:246-253): both travel to the node's first lane, ((urhs - A) - B) and ((vrhs - C) + D) in element order.
__global__ void __launch_bounds__(256) k_ice_a_sub8(IceDM m, int par) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = t >> 3, k0 = t & 7;
  const bool live = i < m.N;
  const double *uo = m.ua[par], *vo = m.va[par];
  double *un = m.ua[1 - par], *vn = m.va[1 - par];
  const bool owned = live && i < m.myN;
  const double *so = m.sig[par];
  double *sn = m.sig[1 - par];
  const size_t E = (size_t)m.myE;
  const double val3 = 1.0 / 3.0, vale = 1.0 / (m.p.ellipse * m.p.ellipse), rdt = m.p.ice_dt;
  const int num = owned ? m.nie_num[i] : 0;
  int nmax = num;
  for (int sft = 32; sft >= 1; sft >>= 1) nmax = max(nmax, __shfl_xor(nmax, sft, 64));
  double urhs = 0.0, vrhs = 0.0;
  for (int base = 0; base < nmax; base += 8) {
    const int k = base + k0;
    double tA = 0.0, tB = 0.0, tC = 0.0, tD = 0.0;
    if (k < num) {
      const int el = m.nie[(size_t)m.maxk * i + k];
      const int n1 = m.en[3 * el], n2 = m.en[3 * el + 1], n3 = m.en[3 * el + 2];
      const bool writer = (n1 < m.myN) ? (n1 == i) : ((n2 < m.myN) ? (n2 == i) : (n3 == i));
      double s11 = so[el], s12 = so[E + el], s22 = so[2 * E + el];
      double eps1, eps2, eps12, delta, msum;
      if (ice_a_strain(m, el, uo, vo, eps1, eps2, eps12, delta, msum)) {
        const double alpha = m.alpha[el];
        const double det2 = 1.0 / (1.0 + alpha), det1 = alpha * det2;
        const double pressure = m.p.Pstar * msum * m.efac[el] / (delta + m.p.delta_min);
        const double r1 = pressure * (eps1 - delta), r2 = pressure * eps2 * vale, r3 = pressure * eps12 * vale;
        double si1 = s11 + s22, si2 = s11 - s22;
        si1 = det1 * si1 + det2 * r1;
        si2 = det1 * si2 + det2 * r2;
        s12 = det1 * s12 + det2 * r3;
        s11 = 0.5 * (si1 + si2);
        s22 = 0.5 * (si1 - si2);
      }
      if (writer) { sn[el] = s11; sn[E + el] = s12; sn[2 * E + el] = s22; }
      if (!((m.a_ice[n1] + m.a_ice[n2]) + m.a_ice[n3] < 0.01)) {
        const double vol = m.elem_area[el], mf = m.metric[el];
        const double *dx = m.gsca + 6 * (size_t)el, *dy = dx + 3;
        const int pos = (n1 == i) ? 0 : ((n2 == i) ? 1 : 2);
        tA = vol * (s11 * dx[pos] + s12 * dy[pos]); tB = vol * s12 * val3 * mf;
        tC = vol * (s12 * dx[pos] + s22 * dy[pos]); tD = vol * s11 * val3 * mf;
      }
    }
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const double qa = __shfl(tA, q, 8), qb = __shfl(tB, q, 8), qc = __shfl(tC, q, 8), qd = __shfl(tD, q, 8);
      urhs = urhs - qa - qb; vrhs = vrhs - qc + qd;
    }
  }
  if (!live || k0 != 0) return;
  if (!owned) { un[i] = uo[i]; vn[i] = vo[i]; return; }
  double mass = (m.m_ice[i] * ICE_RHOICE + m.m_snow[i] * ICE_RHOSNO);
  mass = mass / (1.0 + mass * mass);
  urhs = (urhs * mass + m.rhs_a[i]) / m.area1[i];
  vrhs = (vrhs * mass + m.rhs_m[i]) / m.area1[i];
  const double ai = m.a_ice[i];
  double thickness = (ICE_RHOICE * m.m_ice[i] + ICE_RHOSNO * m.m_snow[i]) / (ai > 0.01 ? ai : 0.01);
  thickness = thickness > 9.0 ? thickness : 9.0;
  const double inv_thickness = 1.0 / thickness;
  double ua = uo[i], va = vo[i];
  const double uw = m.u_w[i], vw = m.v_w[i];
  const double du = ua - uw, dv = va - vw;
  const double umod = sqrt(du * du + dv * dv);
  const double drag = rdt * m.p.cd_oce_ice * umod * ICE_DENSITY_0 * inv_thickness;
  double rhsu = m.u_ice[i] + drag * uw + rdt * (inv_thickness * m.tax[i] + urhs);
  double rhsv = m.v_ice[i] + drag * vw + rdt * (inv_thickness * m.tay[i] + vrhs);
  const double beta = m.beta[i];
  rhsu = beta * ua + rhsu;
  rhsv = beta * va + rhsv;
  const double fc = rdt * m.cori_n[i];
  double det = (1.0 + beta + drag) * (1.0 + beta + drag) + fc * fc;
  det = (m.bnd[i] ? 0.0 : 1.0) / det;
  ua = det * ((1.0 + beta + drag) * rhsu + fc * rhsv);
  va = det * ((1.0 + beta + drag) * rhsv - fc * rhsu);
  if (m.bnd[i]) { ua = 0.0; va = 0.0; }
  un[i] = ua; vn[i] = va;
}
// one EVPdynamics_a call, fused subcycles; the stresses start in parity `sp` and (even number of subcycles) return to it; returns the parity they end in
int enqueue_call_a8(hipStream_t s, int sp) {
  const IceDM &m = I.m;
  hipLaunchKernelGGL(k_ice_a_prep, dim3((m.N + 255) / 256), dim3(256), 0, s, m);      // (u_ice_aux in ua[0])
  int par = sp;
  if (par == 1) {
    hipMemcpyAsync(m.ua[1], m.ua[0], sizeof(double) * m.N, hipMemcpyDeviceToDevice, s);
    hipMemcpyAsync(m.va[1], m.va[0], sizeof(double) * m.N, hipMemcpyDeviceToDevice, s);
  }
  for (int k = 0; k < m.p.evp_rheol_steps; k++) {
    hipLaunchKernelGGL(k_ice_a_sub8, dim3((unsigned)(((size_t)m.N * 8 + 255) / 256)), dim3(256), 0, s, m, par);
    par = 1 - par;
  }
  hipLaunchKernelGGL(k_ice_finish, dim3((m.N + 255) / 256), dim3(256), 0, s, m, par);
  hipLaunchKernelGGL(k_ice_a_alpha, dim3((m.myE + 255) / 256), dim3(256), 0, s, m, par);
  hipLaunchKernelGGL(k_ice_a_beta, dim3((m.myN + 255) / 256), dim3(256), 0, s, m);
  return par;
}
// one EVPdynamics_a call: the stresses stay in sig[sp] (in place); the velocities alternate and end in u_ice / v_ice
void enqueue_call_a(hipStream_t s, int sp) {
  const IceDM &m = I.m;
  hipLaunchKernelGGL(k_ice_a_prep, dim3((m.N + 255) / 256), dim3(256), 0, s, m);
  int par = 0;
  for (int k = 0; k < m.p.evp_rheol_steps; k++) {
    hipLaunchKernelGGL(k_ice_a_stress, dim3((m.myE + 255) / 256), dim3(256), 0, s, m, par, sp);
    hipLaunchKernelGGL(k_ice_a_node, dim3((m.N + 127) / 128), dim3(128), 0, s, m, par, sp);
    par = 1 - par;
  }
  hipLaunchKernelGGL(k_ice_finish, dim3((m.N + 255) / 256), dim3(256), 0, s, m, par);
  hipLaunchKernelGGL(k_ice_a_alpha, dim3((m.myE + 255) / 256), dim3(256), 0, s, m, par);
  hipLaunchKernelGGL(k_ice_a_beta, dim3((m.myN + 255) / 256), dim3(256), 0, s, m);
}

// one EVPdynamics_m call; the stresses start in parity I.cur; returns the parity they end in
int enqueue_call(hipStream_t s, int par) {
  const IceDM &m = I.m;
  hipLaunchKernelGGL(k_ice_prep_node, dim3((m.N + 255) / 256), dim3(256), 0, s, m);
  hipLaunchKernelGGL(k_ice_prep_elem, dim3((m.myE + 255) / 256), dim3(256), 0, s, m);
  // the solver variables start in ua[0]; the stresses in sig[par]: make both parities agree by starting the velocities there as well
  if (par == 1) {
    hipMemcpyAsync(m.ua[1], m.ua[0], sizeof(double) * m.N, hipMemcpyDeviceToDevice, s);
    hipMemcpyAsync(m.va[1], m.va[0], sizeof(double) * m.N, hipMemcpyDeviceToDevice, s);
  }
  static const bool one_lane = getenv("FESOM_GPU_ICE_ONE_LANE") != nullptr;      // (the thread-per-node form, kept for comparison)
  for (int k = 0; k < m.p.evp_rheol_steps; k++) {
    if (one_lane) hipLaunchKernelGGL(k_ice_sub, dim3((m.N + 127) / 128), dim3(128), 0, s, m, par);
    else hipLaunchKernelGGL(k_ice_sub8, dim3((unsigned)(((size_t)m.N * 8 + 255) / 256)), dim3(256), 0, s, m, par);
    par = 1 - par;
  }
  hipLaunchKernelGGL(k_ice_finish, dim3((m.N + 255) / 256), dim3(256), 0, s, m, par);
  return par;
}
}  // namespace

extern "C" {
int fesom_gpu_ice_finalize(void) {
  if (I.stream) hipStreamSynchronize(I.stream);
  if (I.graph) { hipGraphExecDestroy(I.graph); I.graph = nullptr; }
  for (void *p : I.allocs) hipFree(p);
  I.allocs.clear();
  if (I.stream) { hipStreamDestroy(I.stream); I.stream = nullptr; }
  I.ready = false;
  return 0;
}
const char *fesom_gpu_ice_last_error(void) { return I.err.c_str(); }

int fesom_gpu_ice_init(const fesom_mesh_desc *d, const fesom_part_desc *part, const fesom_ice_params *par) {
  if (I.ready) fesom_gpu_ice_finalize();
  I.err.clear();
  if (fesom_internal_select_device(I.err)) { fprintf(stderr, "fesom_gpu_ice: %s\n", I.err.c_str()); return 2; }       // the device of this rank, as the ocean core (csrc/api.hip)
  I.npes = part ? part->npes : 1;
  if (par->evp_rheol_steps < 1) { I.err = "fesom_gpu_ice_init: evp_rheol_steps < 1"; return 3; }
  for (int e = 0; e < d->myDim_elem2D; e++) if (d->ulevels[e] != 1) { I.err = "fesom_gpu_ice_init: cavities (ulevels > 1) are not supported"; return 3; }
  ICECHK(hipStreamCreate(&I.stream));
  IceDM &m = I.m;
  memset(&m, 0, sizeof(m));
  m.p = *par;
  m.myN = d->myDim_nod2D; m.N = d->myDim_nod2D + d->eDim_nod2D; m.myE = d->myDim_elem2D; m.maxk = d->max_nod_in_elem;
  const size_t N = m.N, E = m.myE;
  I.h_en.resize(3 * E);
  for (size_t q = 0; q < 3 * E; q++) I.h_en[q] = d->elem2D_nodes[q] - 1;
  m.en = iupload(I.h_en);
  {   // elements of a node, restricted to the owned elements (0-based, increasing)
    std::vector<int> nie((size_t)m.maxk * N, -1), num(N, 0);
    for (size_t n = 0; n < N; n++)
      for (int k = 0; k < d->nod_in_elem2D_num[n]; k++) {
        const int el = d->nod_in_elem2D[(size_t)m.maxk * n + k] - 1;
        if (el >= 0 && el < m.myE) nie[(size_t)m.maxk * n + num[n]++] = el;
      }
    m.nie = iupload(nie); m.nie_num = iupload(num);
  }
  m.gsca = iupload(std::vector<double>(d->gradient_sca, d->gradient_sca + 6 * E));
  m.elem_area = iupload(std::vector<double>(d->elem_area, d->elem_area + E));
  m.metric = iupload(std::vector<double>(d->metric_factor, d->metric_factor + E));
  m.cori_n = iupload(std::vector<double>(d->coriolis_node, d->coriolis_node + N));
  {
    std::vector<double> a1(N);
    for (size_t n = 0; n < N; n++) a1[n] = d->area[n * (size_t)d->nl];
    m.area1 = iupload(a1);
    std::vector<unsigned char> bnd(N, 0);
    for (int ed = 0; ed < d->myDim_edge2D; ed++)
      if (d->myList_edge2D[ed] > d->edge2D_in) { bnd[d->edges[2 * ed] - 1] = 1; bnd[d->edges[2 * ed + 1] - 1] = 1; }
    m.bnd = iupload(bnd);
  }
  double **nf[] = {&m.u_ice, &m.v_ice, &m.elev, &m.u_w, &m.v_w, &m.tax, &m.tay, &m.ua[0], &m.ua[1], &m.va[0], &m.va[1],
                   &m.rhs_a, &m.rhs_m, &m.invt, &m.mass};
  for (auto f : nf) *f = ialloc<double>(N);
  m.tr3 = ialloc<double>(3 * N);                      // the advected fields side by side
  if (m.tr3) { m.m_ice = m.tr3; m.a_ice = m.tr3 + N; m.m_snow = m.tr3 + 2 * N; }
  {   // FCT advection: pattern of the owned rows (nn_pos = ssh_stiff%colind_loc: the node itself first, then its neighbours in edge order) and the
      // consistent mass matrix assembled as ice_mass_matrix_fill does (src/ice_fct.F90:634-709: element loop, area/12 per pair, twice on the diagonal)
    const int r0 = d->ssh_rowptr[0], nnz = d->ssh_rowptr[m.myN] - r0;
    std::vector<int> rp(m.myN + 1), ci(nnz), col_pos(N, 0);
    for (int i = 0; i <= m.myN; i++) rp[i] = d->ssh_rowptr[i] - r0;
    for (int q = 0; q < nnz; q++) ci[q] = d->ssh_colind_loc[q] - 1;
    std::vector<double> mm(nnz, 0.0);
    for (size_t el = 0; el < E; el++) {
      const int *en = &I.h_en[3 * el];
      for (int n = 0; n < 3; n++) {
        const int row = en[n];
        if (row >= m.myN) continue;
        for (int q = rp[row]; q < rp[row + 1]; q++) col_pos[ci[q]] = q;
        for (int q = 0; q < 3; q++) {
          const int ipos = col_pos[en[q]];
          mm[ipos] = mm[ipos] + d->elem_area[el] / 12.0;
          if (q == n) mm[ipos] = mm[ipos] + d->elem_area[el] / 12.0;
        }
      }
    }
    m.rp = iupload(rp); m.ci = iupload(ci); m.mm = iupload(mm);
    double **wf[] = {&m.rhs, &m.rdiv, &m.lo, &m.dA, &m.dB};
    for (auto f : wf) *f = ialloc<double>(3 * N);
    m.pp = ialloc<double>(6 * N); m.flx = ialloc<double>(9 * E);
  }
  m.sig[0] = ialloc<double>(3 * E); m.sig[1] = ialloc<double>(3 * E); m.pfac = ialloc<double>(E); m.efac = ialloc<double>(E);
  m.alpha = m.beta = nullptr;
  if (par->whichEVP == 2) {      // alpha_evp_array = beta_evp_array = alpha_evp at the start (src/ice_setup_step.F90:85-89)
    m.alpha = ialloc<double>(E); m.beta = ialloc<double>(N);
    std::vector<double> ha(E, par->alpha_evp), hb(N, par->alpha_evp);
    if (m.alpha) hipMemcpy(m.alpha, ha.data(), sizeof(double) * E, hipMemcpyHostToDevice);
    if (m.beta) hipMemcpy(m.beta, hb.data(), sizeof(double) * N, hipMemcpyHostToDevice);
  }
  m.ice_nod = ialloc<unsigned char>(N); m.ice_el = ialloc<unsigned char>(E);
  for (void *p : I.allocs) if (!p) { I.err = "fesom_gpu_ice_init: device allocation failed"; return 1; }
  I.h_efac.assign(E, 0.0);
  I.cur = 0;
  if (I.npes > 1) {     // com_nod2D of this rank (gen_modules_partitioning.F90:17-29)
    const fesom_com_desc &c = part->com_nod2D;
    I.rPE.assign(c.rPE, c.rPE + c.rPEnum); I.rptr.assign(c.rptr, c.rptr + c.rPEnum + 1);
    I.sPE.assign(c.sPE, c.sPE + c.sPEnum); I.sptr.assign(c.sptr, c.sptr + c.sPEnum + 1);
    I.nrecv = I.rptr.back() - 1; I.nsend = I.sptr.back() - 1;
    std::vector<int> rl(I.nrecv), sl(I.nsend);
    for (int q = 0; q < I.nrecv; q++) rl[q] = c.rlist[q] - 1;
    for (int q = 0; q < I.nsend; q++) sl[q] = c.slist[q] - 1;
    I.rlist = iupload(rl); I.slist = iupload(sl);
    I.sptr_d = iupload(I.sptr); I.rptr_d = iupload(I.rptr);
    I.hsend = ialloc<double>(6 * (size_t)I.nsend); I.hrecv = ialloc<double>(6 * (size_t)I.nrecv);     // up to 6 fields per exchange (the FCT limiting factors)
  }
  ICECHK(hipDeviceSynchronize());
  I.ready = true;
  return 0;
}

#define ICE_READY() if (!I.ready) { I.err = "fesom_gpu_ice: not initialised"; return 1; }
int fesom_gpu_ice_upload(const fesom_ice_state *st) {
  ICE_READY();
  const IceDM &m = I.m;
  ICECHK(hipStreamSynchronize(I.stream));
  struct { const double *h; double *d; } nodef[] = {{st->u_ice, m.u_ice}, {st->v_ice, m.v_ice}, {st->a_ice, m.a_ice}, {st->m_ice, m.m_ice}, {st->m_snow, m.m_snow},
      {st->elevation, m.elev}, {st->u_w, m.u_w}, {st->v_w, m.v_w}, {st->stress_atmice_x, m.tax}, {st->stress_atmice_y, m.tay}};
  for (auto &f : nodef) if (f.h) ICECHK(hipMemcpy(f.d, f.h, sizeof(double) * m.N, hipMemcpyHostToDevice));
  const size_t E = m.myE;
  double *sg = m.sig[I.cur];
  if (st->sigma11) ICECHK(hipMemcpy(sg, st->sigma11, sizeof(double) * E, hipMemcpyHostToDevice));
  if (st->sigma12) ICECHK(hipMemcpy(sg + E, st->sigma12, sizeof(double) * E, hipMemcpyHostToDevice));
  if (st->sigma22) ICECHK(hipMemcpy(sg + 2 * E, st->sigma22, sizeof(double) * E, hipMemcpyHostToDevice));
  if (m.alpha && st->alpha_evp_array) ICECHK(hipMemcpy(m.alpha, st->alpha_evp_array, sizeof(double) * E, hipMemcpyHostToDevice));
  if (m.beta && st->beta_evp_array) ICECHK(hipMemcpy(m.beta, st->beta_evp_array, sizeof(double) * m.N, hipMemcpyHostToDevice));
  if (st->a_ice) {      // exp of the pressure factor on the host (glibc exp, the reference's)
    const double val3 = 1.0 / 3.0;
    for (size_t el = 0; el < E; el++) {
      const int *en = &I.h_en[3 * el];
      double asum = ((st->a_ice[en[0]] + st->a_ice[en[1]]) + st->a_ice[en[2]]) * val3;
      if (m.p.whichEVP == 0) asum = ((st->a_ice[en[0]] + st->a_ice[en[1]]) + st->a_ice[en[2]]) / 3.0;      // (the classic EVP divides, ice_EVP.F90:478)
      I.h_efac[el] = exp(-m.p.c_pressure * (1.0 - asum));
    }
    ICECHK(hipMemcpy(m.efac, I.h_efac.data(), sizeof(double) * E, hipMemcpyHostToDevice));
  }
  return 0;
}
int fesom_gpu_ice_evp(int ncalls) {
  ICE_READY();
  if (I.m.p.whichEVP == 2 || I.m.p.whichEVP == 0) {          // adaptive EVP: EVPdynamics_a; classic EVP: EVPdynamics
    if (I.npes > 1) { I.err = "fesom_gpu_ice_evp: partitioned context, call fesom_gpu_ice_evp_partitioned"; return 1; }
    static const bool unfused = getenv("FESOM_GPU_ICE_UNFUSED") != nullptr;      // (the two-launch form of the classic subcycle, kept for comparison)
    if (!unfused) {
      const bool adaptive = I.m.p.whichEVP == 2;
      for (int c = 0; c < ncalls; c++) {
        if (I.m.p.evp_rheol_steps % 2 == 0 && I.cur == 0) {      // an even number of subcycles returns to parity 0 -> one fixed graph
          if (!I.graph) {
            hipGraph_t g;
            ICECHK(hipStreamBeginCapture(I.stream, hipStreamCaptureModeGlobal));
            if (adaptive) enqueue_call_a8(I.stream, 0); else enqueue_call_c(I.stream, 0);
            ICECHK(hipStreamEndCapture(I.stream, &g));
            ICECHK(hipGraphInstantiate(&I.graph, g, nullptr, nullptr, 0));
            hipGraphDestroy(g);
          }
          ICECHK(hipGraphLaunch(I.graph, I.stream));
        } else I.cur = adaptive ? enqueue_call_a8(I.stream, I.cur) : enqueue_call_c(I.stream, I.cur);
      }
      ICECHK(hipGetLastError());
      return 0;
    }
    auto enqueue = [&]() {
      if (I.m.p.whichEVP == 2) enqueue_call_a(I.stream, I.cur);
      else {
        const IceDM &m = I.m;
        const double ax = cos(m.p.theta_io), ay = sin(m.p.theta_io);      // (host libm, as the reference's)
        hipLaunchKernelGGL(k_ice_c_prep_node, dim3((m.myN + 255) / 256), dim3(256), 0, I.stream, m);
        hipLaunchKernelGGL(k_ice_c_prep_elem, dim3((m.myE + 255) / 256), dim3(256), 0, I.stream, m);
        for (int k = 0; k < m.p.evp_rheol_steps; k++) {
          hipLaunchKernelGGL(k_ice_c_stress, dim3((m.myE + 255) / 256), dim3(256), 0, I.stream, m, I.cur);
          hipLaunchKernelGGL(k_ice_c_node, dim3((m.N + 127) / 128), dim3(128), 0, I.stream, m, I.cur, ax, ay);
        }
      }
    };
    // the stresses stay in sig[I.cur] (in place): the launches of a call never change -> one fixed graph (2 launches per subcycle, launch-bound on pi)
    static const bool no_graph = getenv("FESOM_GPU_ICE_NO_GRAPH") != nullptr;
    for (int c = 0; c < ncalls; c++) {
      if (no_graph) { enqueue(); continue; }
      if (!I.graph) {
        hipGraph_t g;
        ICECHK(hipStreamBeginCapture(I.stream, hipStreamCaptureModeGlobal));
        enqueue();
        ICECHK(hipStreamEndCapture(I.stream, &g));
        ICECHK(hipGraphInstantiate(&I.graph, g, nullptr, nullptr, 0));
        hipGraphDestroy(g);
      }
      ICECHK(hipGraphLaunch(I.graph, I.stream));
    }
    ICECHK(hipGetLastError());
    return 0;
  }
  for (int c = 0; c < ncalls; c++) {
    if (I.m.p.evp_rheol_steps % 2 == 0 && I.cur == 0) {      // the usual case: an even number of subcycles returns to parity 0 -> one fixed graph
      if (!I.graph) {
        hipGraph_t g;
        ICECHK(hipStreamBeginCapture(I.stream, hipStreamCaptureModeGlobal));
        enqueue_call(I.stream, 0);
        ICECHK(hipStreamEndCapture(I.stream, &g));
        ICECHK(hipGraphInstantiate(&I.graph, g, nullptr, nullptr, 0));
        hipGraphDestroy(g);
      }
      ICECHK(hipGraphLaunch(I.graph, I.stream));
    } else I.cur = enqueue_call(I.stream, I.cur);
  }
  ICECHK(hipGetLastError());
  return 0;
}
// Partitioned run (one rank per GPU): the same call with the halo of (u_ice_aux, v_ice_aux) exchanged after every subcycle, as the
// reference does (ice_maEVP.F90:588-596).  t == NULL: the library's built-in RCCL transport (fesom_gpu_comm_init); else the host's
// callbacks (exchange is called with kind 0 = com_nod2D and two values per item).
int fesom_gpu_ice_evp_partitioned(int ncalls, const fesom_transport *t) {
  ICE_READY();
  if (I.npes < 2) return fesom_gpu_ice_evp(ncalls);
  if (t && !t->exchange) { I.err = "ice_evp_partitioned: transport callback missing"; return 1; }
  const IceDM &m = I.m;
  const int *sptr_d = I.sptr_d, *rptr_d = I.rptr_d;
  hipStream_t s = I.stream;
  // halo of (u_ice_aux, v_ice_aux) in the buffers of parity `par` (exchange_nod after every subcycle)
  auto halo_uv = [&](double *hu, double *hv) -> int {
    if (I.nsend > 0) hipLaunchKernelGGL(k_ice_pack, dim3((I.nsend + 255) / 256), dim3(256), 0, s, (const double *)hu, (const double *)hv, I.slist, sptr_d, (int)I.sPE.size(), I.nsend, I.hsend);
    if (t) {
      if (hipStreamSynchronize(s) != hipSuccess) { I.err = "ice_evp_partitioned: stream"; return 1; }                      // (a host transport reads the packed buffer)
      if (t->exchange(t->ctx, 0, I.hsend, I.hrecv, 2)) { I.err = "ice_evp_partitioned: transport exchange failed"; return 1; }
    } else if (fesom_internal_rccl_exchange((int)I.sPE.size(), I.sPE.data(), I.sptr.data(), (int)I.rPE.size(), I.rPE.data(), I.rptr.data(), I.hsend, I.hrecv, 2, s)) {
      I.err = "ice_evp_partitioned: built-in transport failed (fesom_gpu_comm_init?)"; return 1;
    }
    if (I.nrecv > 0) hipLaunchKernelGGL(k_ice_unpack, dim3((I.nrecv + 255) / 256), dim3(256), 0, s, hu, hv, I.rlist, rptr_d, (int)I.rPE.size(), I.nrecv, I.hrecv);
    return 0;
  };
  auto halo = [&](int par) -> int { return halo_uv(m.ua[par], m.va[par]); };
  if (m.p.whichEVP == 0) {       // classic EVP (EVPdynamics): the velocities themselves are exchanged after every subcycle (ice_EVP.F90:600)
    const double ax = cos(m.p.theta_io), ay = sin(m.p.theta_io);
    for (int c = 0; c < ncalls; c++) {
      hipLaunchKernelGGL(k_ice_c_prep_node, dim3((m.myN + 255) / 256), dim3(256), 0, s, m);
      hipLaunchKernelGGL(k_ice_c_prep_elem, dim3((m.myE + 255) / 256), dim3(256), 0, s, m);
      for (int k = 0; k < m.p.evp_rheol_steps; k++) {
        hipLaunchKernelGGL(k_ice_c_stress, dim3((m.myE + 255) / 256), dim3(256), 0, s, m, I.cur);
        hipLaunchKernelGGL(k_ice_c_node, dim3((m.N + 127) / 128), dim3(128), 0, s, m, I.cur, ax, ay);
        if (halo_uv(m.u_ice, m.v_ice)) return 1;
      }
    }
    ICECHK(hipGetLastError());
    return 0;
  }
  if (m.p.whichEVP == 2) {       // adaptive EVP (EVPdynamics_a): every rank updates the stresses of all its elements, the node update of its own nodes, then the halo (:879)
    for (int c = 0; c < ncalls; c++) {
      hipLaunchKernelGGL(k_ice_a_prep, dim3((m.N + 255) / 256), dim3(256), 0, s, m);
      int par = 0;
      for (int k = 0; k < m.p.evp_rheol_steps; k++) {
        hipLaunchKernelGGL(k_ice_a_stress, dim3((m.myE + 255) / 256), dim3(256), 0, s, m, par, I.cur);
        hipLaunchKernelGGL(k_ice_a_node, dim3((m.N + 127) / 128), dim3(128), 0, s, m, par, I.cur);
        par = 1 - par;
        if (halo(par)) return 1;
      }
      hipLaunchKernelGGL(k_ice_finish, dim3((m.N + 255) / 256), dim3(256), 0, s, m, par);
      hipLaunchKernelGGL(k_ice_a_alpha, dim3((m.myE + 255) / 256), dim3(256), 0, s, m, par);
      hipLaunchKernelGGL(k_ice_a_beta, dim3((m.myN + 255) / 256), dim3(256), 0, s, m);
    }
    ICECHK(hipGetLastError());
    return 0;
  }
  for (int c = 0; c < ncalls; c++) {
    int par = I.cur;
    hipLaunchKernelGGL(k_ice_prep_node, dim3((m.N + 255) / 256), dim3(256), 0, s, m);
    hipLaunchKernelGGL(k_ice_prep_elem, dim3((m.myE + 255) / 256), dim3(256), 0, s, m);
    if (par == 1) {
      ICECHK(hipMemcpyAsync(m.ua[1], m.ua[0], sizeof(double) * m.N, hipMemcpyDeviceToDevice, s));
      ICECHK(hipMemcpyAsync(m.va[1], m.va[0], sizeof(double) * m.N, hipMemcpyDeviceToDevice, s));
    }
    for (int k = 0; k < m.p.evp_rheol_steps; k++) {
      hipLaunchKernelGGL(k_ice_sub8, dim3((unsigned)(((size_t)m.N * 8 + 255) / 256)), dim3(256), 0, s, m, par);
      par = 1 - par;
      if (I.nsend > 0) hipLaunchKernelGGL(k_ice_pack, dim3((I.nsend + 255) / 256), dim3(256), 0, s, m.ua[par], m.va[par], I.slist, sptr_d, (int)I.sPE.size(), I.nsend, I.hsend);
      if (t) {
        ICECHK(hipStreamSynchronize(s));                      // (a host transport reads the packed buffer)
        if (t->exchange(t->ctx, 0, I.hsend, I.hrecv, 2)) { I.err = "ice_evp_partitioned: transport exchange failed"; return 1; }
      } else if (fesom_internal_rccl_exchange((int)I.sPE.size(), I.sPE.data(), I.sptr.data(), (int)I.rPE.size(), I.rPE.data(), I.rptr.data(), I.hsend, I.hrecv, 2, s)) {
        I.err = "ice_evp_partitioned: built-in transport failed (fesom_gpu_comm_init?)"; return 1;
      }
      if (I.nrecv > 0) hipLaunchKernelGGL(k_ice_unpack, dim3((I.nrecv + 255) / 256), dim3(256), 0, s, m.ua[par], m.va[par], I.rlist, rptr_d, (int)I.rPE.size(), I.nrecv, I.hrecv);
    }
    hipLaunchKernelGGL(k_ice_finish, dim3((m.N + 255) / 256), dim3(256), 0, s, m, par);
    I.cur = par;
  }
  ICECHK(hipGetLastError());
  return 0;
}

// exp of the pressure factor for the CURRENT a_ice (it changes with every advection step): evaluated on the host with glibc's exp like at upload,
// the one libm call of the ice dynamics whose device version could differ in the last bit -> one small device-host-device round trip per ice step
static int ice_refresh_efac() {
  const IceDM &m = I.m;
  ICECHK(hipStreamSynchronize(I.stream));
  I.h_aice.resize(m.N);
  ICECHK(hipMemcpy(I.h_aice.data(), m.a_ice, sizeof(double) * m.N, hipMemcpyDeviceToHost));
  const double val3 = 1.0 / 3.0;
  for (size_t el = 0; el < (size_t)m.myE; el++) {
    const int *en = &I.h_en[3 * el];
    double asum = ((I.h_aice[en[0]] + I.h_aice[en[1]]) + I.h_aice[en[2]]) * val3;
    if (m.p.whichEVP == 0) asum = ((I.h_aice[en[0]] + I.h_aice[en[1]]) + I.h_aice[en[2]]) / 3.0;
    I.h_efac[el] = exp(-m.p.c_pressure * (1.0 - asum));
  }
  ICECHK(hipMemcpy(m.efac, I.h_efac.data(), sizeof(double) * m.myE, hipMemcpyHostToDevice));
  return 0;
}
static int ice_exchange(double *base, int W, const fesom_transport *t) {          // halo of W node fields N apart
  const IceDM &m = I.m;
  hipStream_t s = I.stream;
  if (I.nsend > 0) hipLaunchKernelGGL(k_ice_packw, dim3((I.nsend + 255) / 256), dim3(256), 0, s, base, (size_t)m.N, W, I.slist, I.sptr_d, (int)I.sPE.size(), I.nsend, I.hsend);
  if (t) {
    ICECHK(hipStreamSynchronize(s));
    if (t->exchange(t->ctx, 0, I.hsend, I.hrecv, W)) { I.err = "ice_advect_partitioned: transport exchange failed"; return 1; }
  } else if (fesom_internal_rccl_exchange((int)I.sPE.size(), I.sPE.data(), I.sptr.data(), (int)I.rPE.size(), I.rPE.data(), I.rptr.data(), I.hsend, I.hrecv, W, s)) {
    I.err = "ice_advect_partitioned: built-in transport failed (fesom_gpu_comm_init?)"; return 1;
  }
  if (I.nrecv > 0) hipLaunchKernelGGL(k_ice_unpackw, dim3((I.nrecv + 255) / 256), dim3(256), 0, s, base, (size_t)m.N, W, I.rlist, I.rptr_d, (int)I.rPE.size(), I.nrecv, I.hrecv);
  return 0;
}
// ncalls x the advection part of ice_timestep (ice_TG_rhs_div, ice_fct_solve, ice_update_for_div, cut_off) on the device-resident state with the
// current ice velocities; the pressure factor of the next EVP call follows the new concentration
int fesom_gpu_ice_advect(int ncalls) {
  ICE_READY();
  if (I.npes > 1) { I.err = "fesom_gpu_ice_advect: partitioned context, call fesom_gpu_ice_advect_partitioned"; return 1; }
  const IceDM &m = I.m;
  hipStream_t s = I.stream;
  const dim3 gn((m.myN + 127) / 128), ge((m.myE + 127) / 128), b(128);
  const double gamma = m.p.ice_gamma_fct;
  for (int c = 0; c < ncalls; c++) {
    hipLaunchKernelGGL(k_ice_adv_tg, gn, b, 0, s, m, gamma);
    hipLaunchKernelGGL(k_ice_adv_sweep<false>, gn, b, 0, s, m, (const double *)m.rhs, (const double *)m.dA, m.dB);
    hipLaunchKernelGGL(k_ice_adv_sweep<false>, gn, b, 0, s, m, (const double *)m.rhs, (const double *)m.dB, m.dA);
    hipLaunchKernelGGL(k_ice_adv_flux, ge, b, 0, s, m, gamma, (const double *)m.dA);
    hipLaunchKernelGGL(k_ice_adv_lim, gn, b, 0, s, m);
    hipLaunchKernelGGL(k_ice_adv_upd, gn, b, 0, s, m, m.dB);
    hipLaunchKernelGGL(k_ice_adv_sweep<false>, gn, b, 0, s, m, (const double *)m.rdiv, (const double *)m.dB, m.dA);
    hipLaunchKernelGGL(k_ice_adv_sweep<true>, gn, b, 0, s, m, (const double *)m.rdiv, (const double *)m.dA, m.dB);
    if (ice_refresh_efac()) return 1;
  }
  ICECHK(hipGetLastError());
  return 0;
}
// partitions: the exchanges of the reference (exchange_nod of the high-order increments at the start and after each sweep, of the low-order
// solution, of the limiting factors, of the advected fields, of the divergence increments), three tracers per message
int fesom_gpu_ice_advect_partitioned(int ncalls, const fesom_transport *t) {
  ICE_READY();
  if (I.npes < 2) return fesom_gpu_ice_advect(ncalls);
  if (t && !t->exchange) { I.err = "ice_advect_partitioned: transport callback missing"; return 1; }
  const IceDM &m = I.m;
  hipStream_t s = I.stream;
  const dim3 gn((m.myN + 127) / 128), ga((m.N + 127) / 128), ge((m.myE + 127) / 128), b(128);
  const double gamma = m.p.ice_gamma_fct;
  for (int c = 0; c < ncalls; c++) {
    hipLaunchKernelGGL(k_ice_adv_tg, gn, b, 0, s, m, gamma);
    if (ice_exchange(m.dA, 3, t) || ice_exchange(m.lo, 3, t)) return 1;
    hipLaunchKernelGGL(k_ice_adv_sweep<false>, gn, b, 0, s, m, (const double *)m.rhs, (const double *)m.dA, m.dB);
    if (ice_exchange(m.dB, 3, t)) return 1;
    hipLaunchKernelGGL(k_ice_adv_sweep<false>, gn, b, 0, s, m, (const double *)m.rhs, (const double *)m.dB, m.dA);
    if (ice_exchange(m.dA, 3, t)) return 1;
    hipLaunchKernelGGL(k_ice_adv_flux, ge, b, 0, s, m, gamma, (const double *)m.dA);
    hipLaunchKernelGGL(k_ice_adv_lim, gn, b, 0, s, m);
    if (ice_exchange(m.pp, 6, t)) return 1;
    hipLaunchKernelGGL(k_ice_adv_upd, gn, b, 0, s, m, m.dB);
    if (ice_exchange(m.tr3, 3, t) || ice_exchange(m.dB, 3, t)) return 1;
    hipLaunchKernelGGL(k_ice_adv_sweep<false>, gn, b, 0, s, m, (const double *)m.rdiv, (const double *)m.dB, m.dA);
    if (ice_exchange(m.dA, 3, t)) return 1;
    hipLaunchKernelGGL(k_ice_adv_sweep<false>, gn, b, 0, s, m, (const double *)m.rdiv, (const double *)m.dA, m.dB);
    if (ice_exchange(m.dB, 3, t)) return 1;
    hipLaunchKernelGGL(k_ice_adv_add_cut, ga, b, 0, s, m, (const double *)m.dB);
    if (ice_refresh_efac()) return 1;
  }
  ICECHK(hipGetLastError());
  return 0;
}
int fesom_gpu_ice_download(const fesom_ice_state *st) {
  ICE_READY();
  const IceDM &m = I.m;
  ICECHK(hipStreamSynchronize(I.stream));
  if (st->u_ice) ICECHK(hipMemcpy(st->u_ice, m.u_ice, sizeof(double) * m.N, hipMemcpyDeviceToHost));
  if (st->v_ice) ICECHK(hipMemcpy(st->v_ice, m.v_ice, sizeof(double) * m.N, hipMemcpyDeviceToHost));
  if (st->a_ice) ICECHK(hipMemcpy(st->a_ice, m.a_ice, sizeof(double) * m.N, hipMemcpyDeviceToHost));      // (advected: fesom_gpu_ice_advect)
  if (st->m_ice) ICECHK(hipMemcpy(st->m_ice, m.m_ice, sizeof(double) * m.N, hipMemcpyDeviceToHost));
  if (st->m_snow) ICECHK(hipMemcpy(st->m_snow, m.m_snow, sizeof(double) * m.N, hipMemcpyDeviceToHost));
  const size_t E = m.myE;
  const double *sg = m.sig[I.cur];
  if (st->sigma11) ICECHK(hipMemcpy(st->sigma11, sg, sizeof(double) * E, hipMemcpyDeviceToHost));
  if (st->sigma12) ICECHK(hipMemcpy(st->sigma12, sg + E, sizeof(double) * E, hipMemcpyDeviceToHost));
  if (st->sigma22) ICECHK(hipMemcpy(st->sigma22, sg + 2 * E, sizeof(double) * E, hipMemcpyDeviceToHost));
  if (m.alpha && st->alpha_evp_array) ICECHK(hipMemcpy(st->alpha_evp_array, m.alpha, sizeof(double) * E, hipMemcpyDeviceToHost));
  if (m.beta && st->beta_evp_array) ICECHK(hipMemcpy(st->beta_evp_array, m.beta, sizeof(double) * m.N, hipMemcpyDeviceToHost));
  return 0;
}
int fesom_gpu_ice_time_ms(int ncalls, double *ms_per_call) {
  ICE_READY();
  hipEvent_t e0, e1;
  ICECHK(hipEventCreate(&e0)); ICECHK(hipEventCreate(&e1));
  if (fesom_gpu_ice_evp(1)) return 1;                          // warm-up (builds the graph)
  ICECHK(hipStreamSynchronize(I.stream));
  ICECHK(hipEventRecord(e0, I.stream));
  if (fesom_gpu_ice_evp(ncalls)) return 1;
  ICECHK(hipEventRecord(e1, I.stream));
  ICECHK(hipEventSynchronize(e1));
  float ms = 0;
  ICECHK(hipEventElapsedTime(&ms, e0, e1));
  *ms_per_call = ms / (ncalls > 0 ? ncalls : 1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return 0;
}
}
