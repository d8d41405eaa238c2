// K-profile parameterisation (mix_scheme='KPP', the reference's default) for gfx950: src/oce_ale_mixing_kpp.F90
//   oce_mixing_KPP :240-432, bldepth :446-650, wscale :660-727, ri_iwmix :732-850, blmix_kpp :958-1145, enhance :1152-1191,
//   smooth_nod3D src/gen_support.F90:78-178, followed by Kv = Kv_double(:,:,1) and mo_convect (src/oce_ale.F90:2609-2611,
//   src/oce_mo_conv.F90, use_momix=.false.).
// One wavefront per node column, lane = level.  Everything the reference does with sequential searches over a column
// (first level whose bulk Richardson number exceeds Ricr, first interface below hbl) is a ballot + find-first over the lanes;
// per-column scalars (ustar, Bo, hbl, kbl, caseA, the matching coefficients at hbl) are computed redundantly by all lanes.
// Supported switches: use_sw_pene (sw_3d from the forcing), double_diffusion=.false., use_kpp_nonlclflx=.false.; Kv0_const either way;
// module switches as in the source (smooth_blmc=.true., the others .false.).
//   k_kpp_col     dVsq, ustar, Bo, ri_iwmix, bldepth, blmix_kpp, enhance          (owned nodes)
//   k_kpp_smooth  one sweep of smooth_nod3D for the three blmc fields               (owned nodes; halo by exchange)
//   k_kpp_final   max(interior, blmc) inside the boundary layer, ghats, Kv + mo_convect node part
//   k_kpp_elem    node -> element average of the viscosity (+ minmix) + mo_convect element part
#include "dev.h"
#include <string.h>

#define KPP_NNI 890
#define KPP_NNJ 480
#define KTBL(t, i, j) (t)[(size_t)(j) * (KPP_NNI + 2) + (i)]
#define K_EPSLN 1.0e-40
#define K_EPS 0.1
#define K_VONK 0.4
#define K_CONC1 5.0
#define K_ZMIN (-4.e-7)
#define K_ZMAX 0.0

__device__ __forceinline__ void kpp_wscale(const DM &m, double zehat, double us, double &wm, double &ws) {
  if (zehat <= K_ZMAX) {
    const double deltaz = m.kpp_deltaz, deltau = m.kpp_deltau;
    double zdiff = zehat - K_ZMIN;
    int iz = (int)(zdiff / deltaz);
    iz = iz < KPP_NNI ? iz : KPP_NNI;
    iz = iz > 0 ? iz : 0;
    int izp1 = iz + 1;
    double udiff = us - 0.0;
    int ju = (int)dmin_(udiff / deltau, (double)KPP_NNJ);
    ju = ju > 0 ? ju : 0;
    int jup1 = ju + 1;
    double zfrac = zdiff / deltaz - (double)iz;
    double ufrac = udiff / deltau - (double)ju;
    double fzfrac = 1. - zfrac;
    double wam = fzfrac * KTBL(m.kpp_wmt, iz, jup1) + zfrac * KTBL(m.kpp_wmt, izp1, jup1);
    double wbm = fzfrac * KTBL(m.kpp_wmt, iz, ju) + zfrac * KTBL(m.kpp_wmt, izp1, ju);
    wm = (1. - ufrac) * wbm + ufrac * wam;
    double was = fzfrac * KTBL(m.kpp_wst, iz, jup1) + zfrac * KTBL(m.kpp_wst, izp1, jup1);
    double wbs = fzfrac * KTBL(m.kpp_wst, iz, ju) + zfrac * KTBL(m.kpp_wst, izp1, ju);
    ws = (1. - ufrac) * wbs + ufrac * was;
  } else {
    double u3 = us * us * us;
    wm = K_VONK * us * u3 / (u3 + K_CONC1 * zehat + K_EPSLN);
    ws = wm;
  }
}

__global__ void __launch_bounds__(BLOCK) k_kpp_col(DM m) {
  const int n = col_id(m), l = lane_id(), nz = l + 1;
  if (n >= m.myN) return;
  const int nzmin = m.ulev_n[n], nzmax = m.nlev_n[n];
  const bool lay = (nz >= nzmin && nz <= nzmax - 1);                 // layers
  const bool inner = (nz >= nzmin + 1 && nz <= nzmax - 1);           // interior interfaces
  const bool ifc = (nz >= nzmin && nz <= nzmax);                     // all interfaces of the column
  const size_t nlN = (size_t)m.nl * m.N;
  double u = 0.0, v = 0.0, z = 0.0, hn = 0.0;
  if (lay) { u = DV2(m.Unode, 1, nz, n); v = DV2(m.Unode, 2, nz, n); z = DA2(m.Z_3d_n, nz, n); hn = DA2(m.hnode, nz, n); }
  const double u_up = shup(u), v_up = shup(v), z_up = shup(z), hn_up = shup(hn);
  const double bv = ifc ? DA2L(m.bvfreq, nz, n) : 0.0;
  const double zbs = ifc ? DA2L(m.zbar_3d_n, nz, n) : 0.0;           // signed interface depth
  const double zk = fabs(zbs);
  // ---- dVsq (:263-312)
  const double usurf = bcast(u, nzmin - 1), vsurf = bcast(v, nzmin - 1);
  double dVsq = 0.0;
  if (inner) {
    double u_loc = 0.5 * (u_up + u), v_loc = 0.5 * (v_up + v);
    double du = usurf - u_loc, dv = vsurf - v_loc;
    dVsq = du * du + dv * dv;
  }
  { double last = bcast(dVsq, nzmax - 2); if (nz == nzmax) dVsq = last; }
  // ---- friction velocity and surface buoyancy forcing (:339-345)
  const double sx = m.stress_atmoce_x[n], sy = m.stress_atmoce_y[n];
  const double ustar = sqrt(sqrt(sx * sx + sy * sy) * (1.0 / D_RHO0));
  const double Bo = -D_G * (DA2(m.sw_alpha, nzmin, n) * m.heat_flux[n] / D_VCPW + DA2(m.sw_beta, nzmin, n) * m.water_flux[n] * DTR(m.tr_arr, nzmin, n, 1));
  // ---- ri_iwmix (:732-850): interior values on the interfaces nzmin..nzmax
  double visc = 0.0, kv1 = 0.0;
  if (inner) {
    double dz_inv = 1.0 / (z_up - z);
    double du = u_up - u, dv = v_up - v;
    double shear = du * du + dv * dv;
    shear = shear * dz_inv * dz_inv;
    double ri = dmax_(bv, 0.0) / (shear + K_EPSLN);
    double Rigg = dmax_(ri, 0.0);
    double ratio = dmin_(Rigg / 0.8, 1.0);
    double frit = 1.0 - ratio * ratio;
    frit = frit * frit * frit;
    visc = m.p.visc_sh_limit * frit + m.p.A_ver;
    kv1 = m.p.diff_sh_limit * frit + (m.p.Kv0_const ? m.p.K_ver : kv0_background_qiang(m.lat_deg[n], fabs(DA2L(m.zbar_3d_n, nz, n))));
  }
  {
    double vf = bcast(visc, nzmin), vl = bcast(visc, nzmax - 2), kf = bcast(kv1, nzmin), kl = bcast(kv1, nzmax - 2);
    if (nz == nzmin) { visc = vf; kv1 = kf; }
    if (nz == nzmax) { visc = vl; kv1 = kl; }
  }
  double kv2 = kv1;
  if (m.p.double_diffusion) {        // ddmix (:857-934): salt fingering / diffusive convection on the interior interfaces; exp is the device's
    double a1 = kv1, a2 = kv2;
    if (inner) {
      const double alphaDT = DA2(m.sw_alpha, nz - 1, n) * DTR(m.tr_arr, nz - 1, n, 0), betaDS = DA2(m.sw_beta, nz - 1, n) * DTR(m.tr_arr, nz - 1, n, 1);
      if (alphaDT > betaDS && betaDS > 0.0) {
        const double Rrho = dmin_(alphaDT / betaDS, 1.9);
        double diffdd = 1.0 - ((Rrho - 1.0) / (1.9 - 1.0));
        diffdd = 1.e-4 * diffdd * diffdd * diffdd;
        a1 = a1 + 0.7 * diffdd; a2 = a2 + diffdd;
      } else if (alphaDT < 0.0 && alphaDT > betaDS) {
        const double Rrho = alphaDT / betaDS;
        const double diffdd = 1.5e-6 * 0.909 * exp(4.6 * exp(-0.54 * (1.0 / Rrho - 1.0)));
        double prandtl = 0.15 * Rrho;
        if (Rrho > 0.5) prandtl = (1.85 - 0.85 / Rrho) * Rrho;
        a1 = a1 + diffdd; a2 = a2 + prandtl * diffdd;
      }
    }
    kv1 = a1; kv2 = a2;
    const double f1 = bcast(kv1, nzmin), l1 = bcast(kv1, nzmax - 2), f2 = bcast(kv2, nzmin), l2 = bcast(kv2, nzmax - 2);
    if (nz == nzmin) { kv1 = f1; kv2 = f2; }
    if (nz == nzmax) { kv1 = l1; kv2 = l2; }
  }
  // ---- bldepth (:446-650).  Without short-wave penetration bfsfc = Bo throughout; with it (use_sw_pene) the buoyancy forcing
  // at level nz includes the short-wave flux absorbed above it, so bfsfc / stable depend on the lane inside the search.
  const bool sw = m.p.use_sw_pene != 0;
  const double sw_l = (sw && ifc) ? DA2L(m.sw_3d, nz, n) : 0.0;
  const double sw_min = bcast(sw_l, nzmin - 1);
  const double coeff_sw = sw ? D_G * DA2(m.sw_alpha, nzmin, n) : 0.0;
  double bfsfc = Bo, stable;
  double hbl, caseA;
  int kbl;
  {
    const bool rng = (nz >= nzmin + 1 && nz <= nzmax);
    const double bf_l = sw ? Bo + coeff_sw * (sw_min - sw_l) : Bo;                // bfsfc at the top of iteration nz (:525)
    const double st_l = 0.5 + copysign(0.5, bf_l);
    double sigma = st_l + (1.0 - st_l) * K_EPS;
    double zehat = K_VONK * sigma * zk * bf_l, wm, ws;
    kpp_wscale(m, zehat, ustar, wm, ws);
    double Vtsq = zk * ws * sqrt(fabs(bv)) * m.kpp_Vtc;
    double Ritop = zk * (rng ? DA2L(m.dbsfc, nz, n) : 0.0);
    double Rib_k = Ritop / (dVsq + Vtsq + K_EPSLN);
    const double zkm1 = shup(zk), sw_up = shup(sw_l);
    unsigned long long hit = __ballot(rng && Rib_k > m.p.Ricr);
    if (hit) {
      const int f = __ffsll((long long)hit) - 1;
      const double Rib_km1 = (f == nzmin) ? 0.0 : bcast(Rib_k, f - 1);
      const double Rf = bcast(Rib_k, f), zf = bcast(zk, f), zfm1 = bcast(zkm1, f);
      const double dzup = zf - zfm1;
      hbl = zfm1 + dzup * (m.p.Ricr - Rib_km1) / (Rf - Rib_km1 + K_EPSLN);
      bfsfc = bcast(bf_l, f); stable = bcast(st_l, f);
    } else {
      hbl = bcast(zk, nzmax - 1);
      bfsfc = Bo; stable = 0.5 + copysign(0.5, bfsfc);
      if (nzmax >= nzmin + 1) {
        if (sw) {                     // the loop ran to its end: bfsfc of the last iteration's interpolation to hbl (:573-584)
          const double zk_l = bcast(zk, nzmax - 1), zkm1_l = bcast(zkm1, nzmax - 1), dzup = zk_l - zkm1_l;
          const double s_k = bcast(sw_l, nzmax - 1), s_km1 = bcast(sw_up, nzmax - 1);
          bfsfc = Bo + coeff_sw * (sw_min - (s_km1 + (s_k - s_km1) * (hbl - zkm1_l) / dzup));
          stable = 0.5 + copysign(0.5, bfsfc);
          bfsfc = bfsfc + stable * K_EPSLN;
        } else stable = bcast(st_l, nzmax - 1);
      }
    }
    if (bfsfc > 0.0 && nzmin == 1) {
      double hekman = 0.7 * ustar / dmax_(fabs(m.coriolis_node[n]), K_EPSLN);
      double hmonob = 1.0 * ustar * ustar * ustar / K_VONK / (bfsfc + K_EPSLN);
      double hlimit = stable * dmin_(hekman, hmonob);
      hbl = dmin_(hbl, hlimit);
      hbl = dmax_(hbl, bcast(zk, 1));
    }
    unsigned long long below = __ballot(rng && zk > hbl);
    kbl = below ? __ffsll((long long)below) : nzmax;                 // lane index + 1 = level
    const double zb_k = bcast(zbs, kbl - 1), zb_km1 = bcast(zbs, kbl - 2);
    if (sw) {     // :627-640.  Reference quirk: coeff_sw here is still that of the LAST node of the first loop = the last owned node
      const int nlast = m.myN - 1;
      const double coeff_last = D_G * DA2(m.sw_alpha, m.ulev_n[nlast], nlast);
      const double s_k = bcast(sw_l, kbl - 1), s_km1 = bcast(sw_l, kbl - 2);
      bfsfc = Bo + coeff_last * (sw_min - (s_km1 + (s_k - s_km1) * (hbl + zb_km1) / (zb_km1 - zb_k)));
      stable = 0.5 + copysign(0.5, bfsfc);
      bfsfc = bfsfc + stable * K_EPSLN;
    }
    const double dzup = zb_km1 - zb_k;
    caseA = 0.5 + copysign(0.5, fabs(zb_k) - 0.5 * dzup - hbl);
  }
  // ---- blmix_kpp (:958-1145)
  const int nl1 = nzmax, nu1 = nzmin;
  double bl[3] = {0.0, 0.0, 0.0};
  double gh = (nz <= m.nlm1) ? DA2(m.kpp_ghats, nz, n) : 0.0;          // ghats keeps what is not rewritten
  double dk[3] = {m.kpp_dkm1[3 * (size_t)n], m.kpp_dkm1[3 * (size_t)n + 1], m.kpp_dkm1[3 * (size_t)n + 2]};
  if (!(nl1 < 3 || nl1 - nu1 < 2)) {
    double dth = 0.5 * (hn_up + hn);
    if (nz == nu1) dth = hn * 0.5;
    if (nz == nl1) dth = hn_up * 0.5;
    double sigma = stable * 1.0 + (1.0 - stable) * K_EPS;
    double zehat = K_VONK * sigma * hbl * bfsfc, wm, ws;
    kpp_wscale(m, zehat, ustar, wm, ws);
    const int ica = (int)(caseA + K_EPSLN);
    int kn = ica * (kbl - 1) + (1 - ica) * kbl;
    kn = kn < nl1 - 1 ? kn : nl1 - 1;
    const int knm1 = kn - 1 > nu1 ? kn - 1 : nu1, knp1 = kn + 1 < nl1 ? kn + 1 : nl1;
    const double delhat = fabs(bcast(z, kn - 1)) - hbl;
    const double dth_kn = bcast(dth, kn - 1), dth_knp1 = bcast(dth, knp1 - 1);
    const double R = 1.0 - delhat / dth_kn;
    const double dcv[3] = {visc, kv1, kv2};                            // diff_col(:,1:3) on the interfaces nu1..nl1
    double p[3], h[3];
    for (int j = 0; j < 3; j++) {
      const double c_m1 = bcast(dcv[j], knm1 - 1), c_0 = bcast(dcv[j], kn - 1), c_p1 = bcast(dcv[j], knp1 - 1);
      double dvdzup = (c_m1 - c_0) / dth_kn;
      double dvdzdn = (c_0 - c_p1) / dth_knp1;
      p[j] = 0.5 * ((1.0 - R) * (dvdzup + fabs(dvdzup)) + R * (dvdzdn + fabs(dvdzdn)));
      h[j] = c_0 + p[j] * delhat;
    }
    const double us2 = ustar * ustar;
    const double f1 = stable * K_CONC1 * bfsfc / (us2 * us2 + K_EPSLN);
    double gat[3], dat[3];                                              // 0 momentum (wm), 1 temperature, 2 salinity (ws)
    for (int j = 0; j < 3; j++) {
      const double w = (j == 0) ? wm : ws;
      gat[j] = h[j] / (hbl + K_EPSLN) / (w + K_EPSLN);
      dat[j] = -p[j] / (w + K_EPSLN) + f1 * h[j];
      dat[j] = dmin_(dat[j], 0.0);
    }
    {
      const bool in_bl = (nz >= nu1 + 1 && nz <= nl1 - 1 && nz < kbl);
      double sig = fabs(z) / (hbl + K_EPSLN);
      double sg = stable * sig + (1.0 - stable) * dmin_(sig, K_EPS);
      double ze = K_VONK * sg * hbl * bfsfc, wml, wsl;
      kpp_wscale(m, ze, ustar, wml, wsl);
      const double a1 = sig - 2.0, a2 = 3.0 - 2.0 * sig, a3 = sig - 1.0;
      if (in_bl) {
        for (int j = 0; j < 3; j++) {
          const double Gj = a1 + a2 * gat[j] + a3 * dat[j];
          bl[j] = hbl * ((j == 0) ? wml : wsl) * sig * (1.0 + sig * Gj);
        }
        gh = (1.0 - stable) * m.kpp_cg / (wsl * hbl + K_EPSLN);
      }
    }
    {
      double sig = bcast(zk, kbl - 2) / (hbl + K_EPSLN);
      double sg = stable * sig + (1.0 - stable) * dmin_(sig, K_EPS);
      double ze = K_VONK * sg * hbl * bfsfc, wmk, wsk;
      kpp_wscale(m, ze, ustar, wmk, wsk);
      const double a1 = sig - 2.0, a2 = 3.0 - 2.0 * sig, a3 = sig - 1.0;
      for (int j = 0; j < 3; j++) {
        const double Gj = a1 + a2 * gat[j] + a3 * dat[j];
        dk[j] = hbl * ((j == 0) ? wmk : wsk) * sig * (1.0 + sig * Gj);
      }
    }
  }
  // ---- enhance (:1152-1191) at the interface k = kbl-1
  {
    const int k = kbl - 1;
    const double zb_k = bcast(zbs, k - 1), zb_k1 = bcast(zbs, k);
    const double delta = (hbl + zb_k) / (zb_k - zb_k1), omd = 1.0 - delta;
    const double dcv[3] = {visc, kv1, kv2};
    for (int j = 0; j < 3; j++) {
      const double vk = bcast(dcv[j], k - 1), bk = bcast(bl[j], k - 1);
      const double dkmp5 = caseA * vk + (1.0 - caseA) * bk;
      const double dstar = omd * omd * dk[j] + delta * delta * dkmp5;
      if (nz == k) bl[j] = omd * vk + delta * dstar;
    }
    if (nz == k) gh = (1.0 - caseA) * gh;
  }
  // ---- results
  if (nz <= m.nl) { m.kpp_blmc[(size_t)n * m.nl + l] = bl[0]; m.kpp_blmc[nlN + (size_t)n * m.nl + l] = bl[1]; m.kpp_blmc[2 * nlN + (size_t)n * m.nl + l] = bl[2]; }
  if (ifc) { DA2L(m.kpp_viscA, nz, n) = visc; DA2L(m.kpp_Kv1, nz, n) = kv1; DA2L(m.kpp_Kv2, nz, n) = kv2; }
  if (nz <= m.nlm1) DA2(m.kpp_ghats, nz, n) = gh;
  if (l == 0) {
    m.kpp_hbl[n] = hbl; m.kpp_kbl[n] = kbl; m.kpp_caseA[n] = caseA;
    m.kpp_dkm1[3 * (size_t)n] = dk[0]; m.kpp_dkm1[3 * (size_t)n + 1] = dk[1]; m.kpp_dkm1[3 * (size_t)n + 2] = dk[2];
  }
}

// one sweep of smooth_nod3D (gen_support.F90:95-140 / :145-173) for blmc(:,:,1..3), the three fields of a node in one wave.  src and dst are
// different buffers (the reference gathers into work_array before it overwrites arr); levels outside uln..nln keep the
// value 0 they have in blmc.  The patch areas are summed again in every sweep, in the same order (same value as `vol`).
__global__ void __launch_bounds__(BLOCK) k_kpp_smooth(DM m, const double *src, double *dst) {
  const int n = col_id(m), l = lane_id(), nz = l + 1;
  if (n >= m.myN) return;
  const size_t nlN = (size_t)m.nl * m.N;
  const int uln = m.ulev_n[n], nln = m.nlev_n[n] < m.nl ? m.nlev_n[n] : m.nl;
  const int num = m.nie_num[n];
  // element cluster of the node, lane-parallel (lane k = k-th element): ids, level range, area, the 3 nodes -- read ONCE for the three fields
  // (a wave per node and field repeated this chain three times); then batches of loads for KB elements x 3 fields and the sums in the
  // reference's element order, field by field (no dependent load chains)
  int el_l = 0, n1_l = 0, n2_l = 0, n3_l = 0, lo_l = 1, hi_l = 0;
  double ar_l = 0.0;
  if (l < num) {
    el_l = m.nie[(size_t)m.maxk * n + l];
    n1_l = m.elem_nodes[3 * el_l]; n2_l = m.elem_nodes[3 * el_l + 1]; n3_l = m.elem_nodes[3 * el_l + 2];
    lo_l = uln > m.ulev[el_l] ? uln : m.ulev[el_l];
    int nle = m.nlev[el_l] < m.nl ? m.nlev[el_l] : m.nl;
    hi_l = nln < nle ? nln : nle;
    ar_l = m.elem_area[el_l];
  }
  const int nzc = nz <= m.nl ? nz : m.nl;
  constexpr int KB = 4;                                   // elements per batch
  double work[3] = {0.0, 0.0, 0.0}, vol = 0.0;
  for (int k0 = 0; k0 < num; k0 += KB) {
    double v1[3][KB], v2[3][KB], v3[3][KB];
#pragma unroll
    for (int k = 0; k < KB; k++) {
      const int kk = (k0 + k < num) ? k0 + k : 0;
      const int a1 = rdlane(n1_l, kk), a2 = rdlane(n2_l, kk), a3 = rdlane(n3_l, kk);
#pragma unroll
      for (int f = 0; f < 3; f++) {
        const double *a = src + (size_t)f * nlN;
        v1[f][k] = DA2L(a, nzc, a1); v2[f][k] = DA2L(a, nzc, a2); v3[f][k] = DA2L(a, nzc, a3);
      }
    }
#pragma unroll
    for (int k = 0; k < KB; k++) {
      const int kk = k0 + k;
      if (kk < num) {
        const double ar = bcast(ar_l, kk);
        const bool on = nz >= rdlane(lo_l, kk) && nz <= rdlane(hi_l, kk);
        const double nv = vol + ar;
        vol = on ? nv : vol;
#pragma unroll
        for (int f = 0; f < 3; f++) {
          const double nw = work[f] + ar * (v1[f][k] + v2[f][k] + v3[f][k]);
          work[f] = on ? nw : work[f];
        }
      }
    }
  }
  if (nz > m.nl) return;
  const bool in = nz >= uln && nz <= nln;
  if (in) vol = 1. / (3. * vol);
#pragma unroll
  for (int f = 0; f < 3; f++) dst[(size_t)f * nlN + (size_t)n * m.nl + l] = in ? work[f] * vol : 0.0;
}

// The same sweep with the distinct nodes of the cluster staged ONCE per wave (DM::cl_nb: 7 columns x 3 fields on a regular mesh instead of the 18 x 3
// gathers of the element loop): all loads issued together, then field by field through a wave-private LDS image from which every element takes
// its 3 nodes (DM::cl_pos).  Same sums in the same order.  MAXU = upper bound of the distinct nodes (DM::cl_maxu).
template <int MAXU>
__global__ void __launch_bounds__(BLOCK) k_kpp_smooth_u(DM m, const double *src, double *dst) {
  __shared__ double img[COLS_PER_BLOCK][MAXU][WAVE];
  const int n = col_id(m), l = lane_id(), nz = l + 1, w = threadIdx.x >> 6;
  if (n >= m.myN) return;
  const size_t nlN = (size_t)m.nl * m.N;
  const int uln = m.ulev_n[n], nln = m.nlev_n[n] < m.nl ? m.nlev_n[n] : m.nl;
  const int num = m.nie_num[n], nu = m.cl_nbn[n];
  int nb_l = 0, pos_l = 0, lo_l = 1, hi_l = 0;
  double ar_l = 0.0;
  if (l < nu) nb_l = m.cl_nb[(size_t)m.cl_maxu * n + l];
  if (l < num) {
    const int el = m.nie[(size_t)m.maxk * n + l];
    pos_l = m.cl_pos[(size_t)m.maxk * n + l];
    lo_l = uln > m.ulev[el] ? uln : m.ulev[el];
    const int nle = m.nlev[el] < m.nl ? m.nlev[el] : m.nl;
    hi_l = nln < nle ? nln : nle;
    ar_l = m.elem_area[el];
  }
  const int nzc = nz <= m.nl ? nz : m.nl;
  double v[3][MAXU];
#pragma unroll
  for (int q = 0; q < MAXU; q++) {
    const int a = rdlane(nb_l, q < nu ? q : 0);
#pragma unroll
    for (int f = 0; f < 3; f++) v[f][q] = DA2L(src + (size_t)f * nlN, nzc, a);
  }
  double work[3] = {0.0, 0.0, 0.0}, vol = 0.0;
#pragma unroll
  for (int f = 0; f < 3; f++) {
#pragma unroll
    for (int q = 0; q < MAXU; q++) img[w][q][l] = v[f][q];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double wk = 0.0, vl = 0.0;
    for (int k = 0; k < num; k++) {
      const int pk = rdlane(pos_l, k);
      const double ar = bcast(ar_l, k);
      const bool on = nz >= rdlane(lo_l, k) && nz <= rdlane(hi_l, k);
      const double nw = wk + ar * (img[w][pk & 0xff][l] + img[w][(pk >> 8) & 0xff][l] + img[w][(pk >> 16) & 0xff][l]);
      const double nv = vl + ar;
      wk = on ? nw : wk; vl = on ? nv : vl;
    }
    work[f] = wk; vol = vl;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (nz > m.nl) return;
  const bool in = nz >= uln && nz <= nln;
  if (in) vol = 1. / (3. * vol);
#pragma unroll
  for (int f = 0; f < 3; f++) dst[(size_t)f * nlN + (size_t)n * m.nl + l] = in ? work[f] * vol : 0.0;
}

// :377-392 + Kv = Kv_double(:,:,1) + mo_convect node part (oce_mo_conv.F90:47-57)
__device__ __forceinline__ void kpp_final_body(const DM &m, int n) {
  const int l = lane_id(), nz = l + 1;
  if (n >= m.myN) return;
  const int nzmin = m.ulev_n[n], nzmax = m.nlev_n[n];
  if (nz < nzmin || nz > nzmax) return;
  const size_t nlN = (size_t)m.nl * m.N;
  double visc = DA2L(m.kpp_viscA, nz, n), kv1 = DA2L(m.kpp_Kv1, nz, n), kv2 = DA2L(m.kpp_Kv2, nz, n);
  if (nz >= nzmin + 1 && nz <= nzmax - 1) {
    if (nz < m.kpp_kbl[n]) {
      visc = dmax_(visc, m.kpp_blmc[(size_t)n * m.nl + l]);
      kv1 = dmax_(kv1, m.kpp_blmc[nlN + (size_t)n * m.nl + l]);
      kv2 = dmax_(kv2, m.kpp_blmc[2 * nlN + (size_t)n * m.nl + l]);
      DA2L(m.kpp_viscA, nz, n) = visc; DA2L(m.kpp_Kv1, nz, n) = kv1; DA2L(m.kpp_Kv2, nz, n) = kv2;
    } else DA2(m.kpp_ghats, nz, n) = 0.0;
  }
  double kv = kv1;
  if (nz >= nzmin + 1 && nz <= nzmax - 1) {
    if (m.p.use_momix) kv = kv + momix_mo(m, nz, n);
    if (m.p.use_instabmix && DA2L(m.bvfreq, nz, n) < 0.) kv = dmax_(kv, m.p.instabmix_kv);
    if (nzmin <= 1 && m.p.use_windmix && nz <= m.p.windmix_nl + 1) kv = dmax_(kv, m.p.windmix_kv);
  }
  DA2L(m.Kv, nz, n) = kv;
}

// :400-416 + mo_convect element part (oce_mo_conv.F90:62-77)
// viscosity of node n at interface k as k_kpp_final leaves it.  FUSED: the node part of the same launch may or may not have stored it yet; the
// maximum with blmc applied to either value gives the same number (max(max(a, b), b) = max(a, b)), 8-byte loads are not torn.
template <bool FUSED>
__device__ __forceinline__ double kpp_visc_final(const DM &m, int k, int n) {
  double v = DA2L(m.kpp_viscA, k, n);
  if (FUSED && k >= m.ulev_n[n] + 1 && k <= m.nlev_n[n] - 1 && k < m.kpp_kbl[n]) v = dmax_(v, m.kpp_blmc[(size_t)n * m.nl + k - 1]);
  return v;
}
template <bool FUSED>
__device__ __forceinline__ void kpp_elem_body(const DM &m, int e) {
  const int nz = lane_id() + 1;
  if (e >= m.myE) return;
  const int nzmin = m.ulev[e], nzmax = m.nlev[e];
  if (nz < nzmin || nz > nzmax) return;
  const int n1 = m.elem_nodes[3 * e], n2 = m.elem_nodes[3 * e + 1], n3 = m.elem_nodes[3 * e + 2];
  const int k = nz < nzmax ? nz : nzmax - 1;                               // viscAE(nlevels) = viscAE(nlevels-1)
  double av;
  bool unstable;
  if (m.exp_batch & 1) {
    // every gather of the column in one batch of independent loads (the short-circuit forms below issue them one behind the other: on CORE2-class
    // meshes the kernel waits for memory 84 % of its time), the conditions applied as selects afterwards: same values, same order of the sum
    double v1 = DA2L(m.kpp_viscA, k, n1), v2 = DA2L(m.kpp_viscA, k, n2), v3 = DA2L(m.kpp_viscA, k, n3);
    double b1 = 0.0, b2 = 0.0, b3 = 0.0, f1 = 0.0, f2 = 0.0, f3 = 0.0;
    if (FUSED) { b1 = m.kpp_blmc[(size_t)n1 * m.nl + k - 1]; b2 = m.kpp_blmc[(size_t)n2 * m.nl + k - 1]; b3 = m.kpp_blmc[(size_t)n3 * m.nl + k - 1]; }
    if (m.p.use_instabmix) { f1 = DA2L(m.bvfreq, nz, n1); f2 = DA2L(m.bvfreq, nz, n2); f3 = DA2L(m.bvfreq, nz, n3); }
    if (FUSED) {
      const bool c1 = k >= m.ulev_n[n1] + 1 && k <= m.nlev_n[n1] - 1 && k < m.kpp_kbl[n1], c2 = k >= m.ulev_n[n2] + 1 && k <= m.nlev_n[n2] - 1 && k < m.kpp_kbl[n2],
                 c3 = k >= m.ulev_n[n3] + 1 && k <= m.nlev_n[n3] - 1 && k < m.kpp_kbl[n3];
      v1 = c1 ? dmax_(v1, b1) : v1; v2 = c2 ? dmax_(v2, b2) : v2; v3 = c3 ? dmax_(v3, b3) : v3;
    }
    av = (v1 + v2 + v3) / 3.0;
    unstable = f1 < 0. || f2 < 0. || f3 < 0.;
  } else {
    av = (kpp_visc_final<FUSED>(m, k, n1) + kpp_visc_final<FUSED>(m, k, n2) + kpp_visc_final<FUSED>(m, k, n3)) / 3.0;
    unstable = m.p.use_instabmix && nz >= nzmin + 1 && nz <= nzmax - 1 && (DA2L(m.bvfreq, nz, n1) < 0. || DA2L(m.bvfreq, nz, n2) < 0. || DA2L(m.bvfreq, nz, n3) < 0.);
  }
  if (nz == nzmin && av < 3.0e-3) av = 3.0e-3;                            // minmix on the first interface only
  if (nz >= nzmin + 1 && nz <= nzmax - 1) {
    if (m.p.use_instabmix && unstable)
      av = dmax_(av, m.p.instabmix_kv);
    if (m.p.use_momix && m.momix_elem[e]) av = av + ((momix_mo(m, nz, n1) + momix_mo(m, nz, n2)) + momix_mo(m, nz, n3)) / 3.0;
    if (nzmin <= 1 && m.p.use_windmix && nz <= m.p.windmix_nl + 1) av = dmax_(av, m.p.windmix_kv);
  }
  DA2L(m.Av, nz, e) = av;
}
__global__ void __launch_bounds__(BLOCK) k_kpp_final(DM m) { kpp_final_body(m, col_id(m)); }
__global__ void __launch_bounds__(BLOCK) k_kpp_elem(DM m) { kpp_elem_body<false>(m, col_id(m)); }
// single partition: both in ONE launch (first ncolE column slots are element columns), one dependent launch less on the critical chain
__global__ void __launch_bounds__(BLOCK) k_kpp_final_elem(DM m, int ncolE) {
  const int c = col_id(m);
  if (c < ncolE) kpp_elem_body<true>(m, c); else kpp_final_body(m, c - ncolE);
}

#define LAUNCH_COL(k, ncol, ...) hipLaunchKernelGGL(k, dim3(nblocks(ncol)), dim3(BLOCK), 0, s, __VA_ARGS__)
static void smooth(const DM &m, hipStream_t s, const double *src, double *dst) {
  static const int env = getenv("FESOM_GPU_EXP_KPPU") ? atoi(getenv("FESOM_GPU_EXP_KPPU")) : -1;
  const bool staged = (env >= 0 ? env != 0 : true) && m.cl_maxu <= 12;      // (430 -> 228 us per sweep on the basin, 10.7 -> 10.0 us on pi)
  if (staged && m.cl_maxu <= 8) hipLaunchKernelGGL((k_kpp_smooth_u<8>), dim3(nblocks(m.myN)), dim3(BLOCK), 0, s, m, src, dst);
  else if (staged) hipLaunchKernelGGL((k_kpp_smooth_u<12>), dim3(nblocks(m.myN)), dim3(BLOCK), 0, s, m, src, dst);
  else hipLaunchKernelGGL(k_kpp_smooth, dim3(nblocks(m.myN)), dim3(BLOCK), 0, s, m, src, dst);      // the three blmc fields in one wave per node
}
int launch_named_kpp(const DM &m, hipStream_t s, const char *name) {
  if (m.p.mix_scheme != 1) return -1;
  if (!strcmp(name, "k_kpp_col")) { LAUNCH_COL(k_kpp_col, m.myN, m); return 0; }
  if (!strcmp(name, "k_kpp_smooth1")) { smooth(m, s, m.kpp_blmc, m.kpp_sA); return 0; }      // blmc -> sA -> sB -> blmc
  if (!strcmp(name, "k_kpp_smooth2")) { smooth(m, s, m.kpp_sA, m.kpp_sB); return 0; }
  if (!strcmp(name, "k_kpp_smooth3")) { smooth(m, s, m.kpp_sB, m.kpp_blmc); return 0; }
  if (!strcmp(name, "k_kpp_final")) { LAUNCH_COL(k_kpp_final, m.myN, m); return 0; }
  if (!strcmp(name, "k_kpp_elem")) { LAUNCH_COL(k_kpp_elem, m.myE, m); return 0; }
  if (!strcmp(name, "k_kpp_final_elem")) { hipLaunchKernelGGL(k_kpp_final_elem, dim3(nblocks(m.myE) + nblocks(m.myN)), dim3(BLOCK), 0, s, m, nblocks(m.myE) * COLS_PER_BLOCK); return 0; }
  if (!strcmp(name, "mixing_kpp")) {                        // oce_mixing_KPP + Kv = Kv_double(:,:,1) + mo_convect
    launch_momix(m, s);
    LAUNCH_COL(k_kpp_col, m.myN, m);
    smooth(m, s, m.kpp_blmc, m.kpp_sA); smooth(m, s, m.kpp_sA, m.kpp_sB); smooth(m, s, m.kpp_sB, m.kpp_blmc);
    if (m.N == m.myN) hipLaunchKernelGGL(k_kpp_final_elem, dim3(nblocks(m.myE) + nblocks(m.myN)), dim3(BLOCK), 0, s, m, nblocks(m.myE) * COLS_PER_BLOCK);
    else { LAUNCH_COL(k_kpp_final, m.myN, m); LAUNCH_COL(k_kpp_elem, m.myE, m); }
    return 0;
  }
  if (!strcmp(name, "mo_convect")) return 0;                // fused into k_kpp_final / k_kpp_elem
  return -1;
}
