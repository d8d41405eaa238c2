/* ORACLE -- TEST INFRASTRUCTURE ONLY (see orc.h).
 *
 * K-profile parameterisation of the reference (src/oce_ale_mixing_kpp.F90, module o_mixing_KPP_mod):
 *   oce_mixing_kpp_init :97-201    constants Vtc, cg and the wm/ws look-up tables
 *   oce_mixing_KPP      :240-432   driver: dVsq, ustar, Bo, the calls below, smoothing of blmc, max with the interior values,
 *                                  node -> element average of the viscosity
 *   bldepth             :446-650   boundary layer depth from the bulk Richardson number
 *   wscale              :660-727   turbulent velocity scales from the tables
 *   ri_iwmix            :732-850   interior (shear instability + background) mixing
 *   blmix_kpp           :958-1145  boundary layer profiles, nonlocal coefficient ghats
 *   enhance             :1152-1191 enhanced diffusivity at the kbl-1 interface
 *   smooth_nod3D        src/gen_support.F90:78-178
 * Supported options: use_sw_pene (sw_3d from the forcing), double_diffusion=.false., use_kpp_nonlclflx=.false.; Kv0_const either way
 * (the reference's defaults);
 * module switches smooth_blmc=.true., smooth_hbl/smooth_Ri_hor/smooth_Ri_ver/limit_hbl_ekmmob=.false. as in the source.
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NNI 890
#define NNJ 480
#define TBL(t, i, j) (t)[(size_t)(j) * (NNI + 2) + (i)]          /* wmt(0:nni+1, 0:nnj+1), first index fastest */
static const double epsln = 1.0e-40, epsilon_kpp = 0.1, vonk = 0.4, conc1 = 5.0;
static const double zmin = -4.e-7, zmax = 0.0, umin = 0.0, umax = 0.04;
#define VCPW 4.2e6

/* Table construction.  Which library call the reference's build (amdflang -O2) makes for each power was pinned on the
 * reference run (tests/golden/pi_default_reference.npz): x**(1/2) -> sqrt, x**(1/4) and x**(1/3) -> pow; the constant
 * expression of cg is folded at compile time to the value cbrt gives. */
void orc_kpp_tables(double *wmt, double *wst, double *deltaz, double *deltau) {
  const double conam = 1.257, concm = 8.380, conc2 = 16.0, zetam = -0.2, conas = -28.86, concs = 98.96, conc3 = 16.0, zetas = -1.0;
  *deltaz = (zmax - zmin) / (double)(NNI + 1);
  *deltau = (umax - umin) / (double)(NNJ + 1);
  for (int i = 0; i <= NNI + 1; i++) {
    double zehat = *deltaz * (double)i + zmin;
    for (int j = 0; j <= NNJ + 1; j++) {
      double usta = *deltau * (double)j + umin;
      double u3 = usta * usta * usta;
      double zeta = zehat / (u3 + epsln);
      if (zehat >= 0.) {
        TBL(wmt, i, j) = vonk * usta / (1. + conc1 * zeta);
        TBL(wst, i, j) = TBL(wmt, i, j);
      } else {
        if (zeta > zetam) TBL(wmt, i, j) = vonk * usta * pow(1. - conc2 * zeta, 1. / 4.);
        else TBL(wmt, i, j) = vonk * pow(conam * u3 - concm * zehat, 1. / 3.);
        if (zeta > zetas) TBL(wst, i, j) = vonk * usta * sqrt(1. - conc3 * zeta);
        else TBL(wst, i, j) = vonk * pow(conas * u3 - concs * zehat, 1. / 3.);
      }
    }
  }
}

void orc_kpp_init(void) {
  const double cstar = 10.0, concs = 98.96;
  size_t N = C_.N, nl = NL;
  C_.kpp_wmt = malloc(sizeof(double) * (NNI + 2) * (NNJ + 2));
  C_.kpp_wst = malloc(sizeof(double) * (NNI + 2) * (NNJ + 2));
  orc_kpp_tables(C_.kpp_wmt, C_.kpp_wst, &C_.kpp_deltaz, &C_.kpp_deltau);
  C_.kpp_Vtc = C_.p.concv * sqrt(0.2 / concs / epsilon_kpp) / (vonk * vonk) / C_.p.Ricr;
  C_.kpp_cg = cstar * vonk * cbrt(concs * vonk * epsilon_kpp);
  C_.kpp_kbl = calloc(N ? N : 1, sizeof(int));
  C_.kpp_work = calloc(nl * N, sizeof(double));
  C_.kpp_vol = calloc(nl * N, sizeof(double));
}

static void wscale(double zehat, double us, double *wm, double *ws) {
  const double *wmt = C_.kpp_wmt, *wst = C_.kpp_wst;
  const double deltaz = C_.kpp_deltaz, deltau = C_.kpp_deltau;
  if (zehat <= zmax) {
    double zdiff = zehat - zmin;
    int iz = (int)(zdiff / deltaz);
    iz = iz < NNI ? iz : NNI;
    iz = iz > 0 ? iz : 0;
    int izp1 = iz + 1;
    double udiff = us - umin;
    int ju = (int)dmin(udiff / deltau, (double)NNJ);
    ju = ju > 0 ? ju : 0;
    int jup1 = ju + 1;
    double zfrac = zdiff / deltaz - (double)iz;
    double ufrac = udiff / deltau - (double)ju;
    double fzfrac = 1. - zfrac;
    double wam = fzfrac * TBL(wmt, iz, jup1) + zfrac * TBL(wmt, izp1, jup1);
    double wbm = fzfrac * TBL(wmt, iz, ju) + zfrac * TBL(wmt, izp1, ju);
    *wm = (1. - ufrac) * wbm + ufrac * wam;
    double was = fzfrac * TBL(wst, iz, jup1) + zfrac * TBL(wst, izp1, jup1);
    double wbs = fzfrac * TBL(wst, iz, ju) + zfrac * TBL(wst, izp1, ju);
    *ws = (1. - ufrac) * wbs + ufrac * was;
  } else {
    double u3 = us * us * us;
    *wm = vonk * us * u3 / (u3 + conc1 * zehat + epsln);
    *ws = *wm;
  }
}

/* ri_iwmix :732-850 */
static void ri_iwmix(void) {
  const double Riinfty = 0.8;
  for (int n = 1; n <= C_.m.myDim_nod2D; n++) {
    int nzmin = ULEVN(n), nzmax = NLEVN(n);
    for (int nz = nzmin + 1; nz <= nzmax - 1; nz++) {
      double dz_inv = 1.0 / (A2(C_.Z_3d_n, nz - 1, n) - A2(C_.Z_3d_n, nz, n));
      double du = V2(C_.Unode, 1, nz - 1, n) - V2(C_.Unode, 1, nz, n), dv = V2(C_.Unode, 2, nz - 1, n) - V2(C_.Unode, 2, nz, n);
      double shear = du * du + dv * dv;
      shear = shear * dz_inv * dz_inv;
      A2L(C_.kpp_Kv1, nz, n) = dmax(A2L(C_.bvfreq, nz, n), 0.0) / (shear + epsln);
    }
    A2L(C_.kpp_Kv1, nzmin, n) = A2L(C_.kpp_Kv1, nzmin + 1, n);
    A2L(C_.kpp_Kv1, nzmax, n) = A2L(C_.kpp_Kv1, nzmax - 1, n);
  }
  for (int n = 1; n <= C_.m.myDim_nod2D; n++) {
    int nzmin = ULEVN(n), nzmax = NLEVN(n);
    for (int nz = nzmin + 1; nz <= nzmax - 1; nz++) {
      double Rigg = dmax(A2L(C_.kpp_Kv1, nz, n), 0.0);
      double ratio = dmin(Rigg / Riinfty, 1.0);
      double frit = 1.0 - ratio * ratio;
      frit = frit * frit * frit;
      A2L(C_.kpp_viscA, nz, n) = C_.p.visc_sh_limit * frit + C_.p.A_ver;
      A2L(C_.kpp_Kv1, nz, n) = C_.p.diff_sh_limit * frit + (C_.p.Kv0_const ? C_.p.K_ver : orc_kv0_background_qiang(n, nz));
      A2L(C_.kpp_Kv2, nz, n) = A2L(C_.kpp_Kv1, nz, n);
    }
    A2L(C_.kpp_viscA, nzmin, n) = A2L(C_.kpp_viscA, nzmin + 1, n);
    A2L(C_.kpp_Kv1, nzmin, n) = A2L(C_.kpp_Kv1, nzmin + 1, n);
    A2L(C_.kpp_Kv2, nzmin, n) = A2L(C_.kpp_Kv2, nzmin + 1, n);
    A2L(C_.kpp_viscA, nzmax, n) = A2L(C_.kpp_viscA, nzmax - 1, n);
    A2L(C_.kpp_Kv1, nzmax, n) = A2L(C_.kpp_Kv1, nzmax - 1, n);
    A2L(C_.kpp_Kv2, nzmax, n) = A2L(C_.kpp_Kv2, nzmax - 1, n);
  }
}

/* ddmix :857-934 (double_diffusion): salt fingering / diffusive convection added to the interior diffusivities of heat (Kv1) and salt (Kv2) */
static void ddmix(void) {
  const double Rrho0 = 1.9, dsfmax = 1.e-4, viscosity_molecular = 1.5e-6;
  for (int n = 1; n <= C_.m.myDim_nod2D; n++) {
    int nzmin = ULEVN(n), nzmax = NLEVN(n);
    for (int nz = nzmin + 1; nz <= nzmax - 1; nz++) {
      double alphaDT = A2(C_.sw_alpha, nz - 1, n) * TR(nz - 1, n, 1), betaDS = A2(C_.sw_beta, nz - 1, n) * TR(nz - 1, n, 2);
      if (alphaDT > betaDS && betaDS > 0.0) {
        double Rrho = dmin(alphaDT / betaDS, Rrho0);
        double diffdd = 1.0 - ((Rrho - 1.0) / (Rrho0 - 1.0));
        diffdd = dsfmax * diffdd * diffdd * diffdd;
        A2L(C_.kpp_Kv1, nz, n) = A2L(C_.kpp_Kv1, nz, n) + 0.7 * diffdd;
        A2L(C_.kpp_Kv2, nz, n) = A2L(C_.kpp_Kv2, nz, n) + diffdd;
      } else if (alphaDT < 0.0 && alphaDT > betaDS) {
        double Rrho = alphaDT / betaDS;
        double diffdd = viscosity_molecular * 0.909 * exp(4.6 * exp(-0.54 * (1.0 / Rrho - 1.0)));
        double prandtl = 0.15 * Rrho;
        if (Rrho > 0.5) prandtl = (1.85 - 0.85 / Rrho) * Rrho;
        A2L(C_.kpp_Kv1, nz, n) = A2L(C_.kpp_Kv1, nz, n) + diffdd;
        A2L(C_.kpp_Kv2, nz, n) = A2L(C_.kpp_Kv2, nz, n) + prandtl * diffdd;
      }
    }
    A2L(C_.kpp_Kv1, nzmin, n) = A2L(C_.kpp_Kv1, nzmin + 1, n); A2L(C_.kpp_Kv2, nzmin, n) = A2L(C_.kpp_Kv2, nzmin + 1, n);
    A2L(C_.kpp_Kv1, nzmax, n) = A2L(C_.kpp_Kv1, nzmax - 1, n); A2L(C_.kpp_Kv2, nzmax, n) = A2L(C_.kpp_Kv2, nzmax - 1, n);
  }
}

/* bldepth :446-650 (use_sw_pene=.false.) */
static void bldepth(void) {
  const double cekman = 0.7, cmonob = 1.0, Ricr = C_.p.Ricr;
  for (int n = 1; n <= C_.m.myDim_nod2D; n++) {
    C_.kpp_kbl[n - 1] = NLEVN(n);
    C_.kpp_hbl[n - 1] = fabs(A2L(C_.zbar_3d_n, NLEVN(n), n));
  }
  for (int n = 1; n <= C_.m.myDim_nod2D; n++) {
    int nzmin = ULEVN(n), nzmax = NLEVN(n);
    double Rib_km1 = 0.0;
    const int sw = C_.p.use_sw_pene;
    const double coeff_sw = sw ? G_ACC * A2(C_.sw_alpha, nzmin, n) : 0.0;
    C_.kpp_bfsfc[n - 1] = C_.kpp_Bo[n - 1];
    for (int nz = nzmin + 1; nz <= nzmax; nz++) {
      double zk = fabs(A2L(C_.zbar_3d_n, nz, n)), zkm1 = fabs(A2L(C_.zbar_3d_n, nz - 1, n));
      if (sw) C_.kpp_bfsfc[n - 1] = C_.kpp_Bo[n - 1] + coeff_sw * (A2L(C_.sw_3d, nzmin, n) - A2L(C_.sw_3d, nz, n));
      C_.kpp_stable[n - 1] = 0.5 + copysign(0.5, C_.kpp_bfsfc[n - 1]);
      double sigma = C_.kpp_stable[n - 1] + (1.0 - C_.kpp_stable[n - 1]) * epsilon_kpp;
      double zehat = vonk * sigma * zk * C_.kpp_bfsfc[n - 1], wm, ws;
      wscale(zehat, C_.kpp_ustar[n - 1], &wm, &ws);
      double bvsq = A2L(C_.bvfreq, nz, n);
      double Vtsq = zk * ws * sqrt(fabs(bvsq)) * C_.kpp_Vtc;
      double Ritop = zk * A2L(C_.dbsfc, nz, n);
      double Rib_k = Ritop / (A2L(C_.kpp_dVsq, nz, n) + Vtsq + epsln);
      double dzup = zk - zkm1;
      if (Rib_k > Ricr) {
        C_.kpp_hbl[n - 1] = zkm1 + dzup * (Ricr - Rib_km1) / (Rib_k - Rib_km1 + epsln);
        C_.kpp_kbl[n - 1] = nz;
        break;
      } else Rib_km1 = Rib_k;
      if (sw) {                                          /* :573-584 */
        C_.kpp_bfsfc[n - 1] = C_.kpp_Bo[n - 1] + coeff_sw * (A2L(C_.sw_3d, nzmin, n) -
                              (A2L(C_.sw_3d, nz - 1, n) + (A2L(C_.sw_3d, nz, n) - A2L(C_.sw_3d, nz - 1, n)) * (C_.kpp_hbl[n - 1] - zkm1) / dzup));
        C_.kpp_stable[n - 1] = 0.5 + copysign(0.5, C_.kpp_bfsfc[n - 1]);
        C_.kpp_bfsfc[n - 1] = C_.kpp_bfsfc[n - 1] + C_.kpp_stable[n - 1] * epsln;
      }
    }
    if (C_.kpp_bfsfc[n - 1] > 0.0 && nzmin == 1) {
      double us = C_.kpp_ustar[n - 1];
      double hekman = cekman * us / dmax(fabs(C_.m.coriolis_node[n - 1]), epsln);
      double hmonob = cmonob * us * us * us / vonk / (C_.kpp_bfsfc[n - 1] + epsln);
      double hlimit = C_.kpp_stable[n - 1] * dmin(hekman, hmonob);
      C_.kpp_hbl[n - 1] = dmin(C_.kpp_hbl[n - 1], hlimit);
      C_.kpp_hbl[n - 1] = dmax(C_.kpp_hbl[n - 1], fabs(A2L(C_.zbar_3d_n, 2, n)));
    }
  }
  for (int n = 1; n <= C_.m.myDim_nod2D; n++) {
    int nzmin = ULEVN(n), nzmax = NLEVN(n);
    int kbl = nzmax;
    for (int nz = nzmin + 1; nz <= nzmax; nz++)
      if (fabs(A2L(C_.zbar_3d_n, nz, n)) > C_.kpp_hbl[n - 1]) { kbl = nz; break; }
    C_.kpp_kbl[n - 1] = kbl;
    if (C_.p.use_sw_pene) {                              /* :627-640 */
      /* Reference quirk: this second node loop does not set coeff_sw again, it still holds the value of the LAST node of the
       * first loop (the rank's last owned node).  kpp_sw_node(n) names that node: the last node on one partition; the tests
       * that compare with a 2-rank reference run set it to the last owned node of the rank that owns n. */
      const int nl_ = (int)C_.kpp_sw_node[n - 1];
      const double coeff_sw = G_ACC * A2(C_.sw_alpha, ULEVN(nl_), nl_);
      C_.kpp_bfsfc[n - 1] = C_.kpp_Bo[n - 1] + coeff_sw * (A2L(C_.sw_3d, nzmin, n) -
                            (A2L(C_.sw_3d, kbl - 1, n) + (A2L(C_.sw_3d, kbl, n) - A2L(C_.sw_3d, kbl - 1, n)) * (C_.kpp_hbl[n - 1] + A2L(C_.zbar_3d_n, kbl - 1, n)) /
                                                              (A2L(C_.zbar_3d_n, kbl - 1, n) - A2L(C_.zbar_3d_n, kbl, n))));
      C_.kpp_stable[n - 1] = 0.5 + copysign(0.5, C_.kpp_bfsfc[n - 1]);
      C_.kpp_bfsfc[n - 1] = C_.kpp_bfsfc[n - 1] + C_.kpp_stable[n - 1] * epsln;
    }
    double dzup = A2L(C_.zbar_3d_n, kbl - 1, n) - A2L(C_.zbar_3d_n, kbl, n);
    C_.kpp_caseA[n - 1] = 0.5 + copysign(0.5, fabs(A2L(C_.zbar_3d_n, kbl, n)) - 0.5 * dzup - C_.kpp_hbl[n - 1]);
  }
}

#define BLMC(j, nz, n) A2L(C_.kpp_blmc[(j) - 1], nz, n)

/* blmix_kpp :958-1145 */
static void blmix_kpp(void) {
  double dthick[128], dc[3][128];
  for (int j = 0; j < 3; j++) memset(C_.kpp_blmc[j], 0, sizeof(double) * (size_t)NL * C_.N);
  for (int n = 1; n <= C_.m.myDim_nod2D; n++) {
    int nl1 = NLEVN(n), nu1 = ULEVN(n);
    if (nl1 < 3) continue;
    if (nl1 - nu1 < 2) continue;
    for (int nz = nu1 + 1; nz <= nl1 - 1; nz++) dthick[nz] = 0.5 * (A2(C_.hnode, nz - 1, n) + A2(C_.hnode, nz, n));
    dthick[nu1] = A2(C_.hnode, nu1, n) * 0.5;
    dthick[nl1] = A2(C_.hnode, nl1 - 1, n) * 0.5;
    for (int nz = nu1; nz <= nl1 - 1; nz++) { dc[0][nz] = A2L(C_.kpp_viscA, nz, n); dc[1][nz] = A2L(C_.kpp_Kv1, nz, n); dc[2][nz] = A2L(C_.kpp_Kv2, nz, n); }
    for (int j = 0; j < 3; j++) dc[j][nl1] = dc[j][nl1 - 1];
    const double stable = C_.kpp_stable[n - 1], hbl = C_.kpp_hbl[n - 1], bfsfc = C_.kpp_bfsfc[n - 1], us = C_.kpp_ustar[n - 1];
    const int kbl = C_.kpp_kbl[n - 1];
    double sigma = stable * 1.0 + (1.0 - stable) * epsilon_kpp;
    double zehat = vonk * sigma * hbl * bfsfc, wm, ws;
    wscale(zehat, us, &wm, &ws);
    int ica = (int)(C_.kpp_caseA[n - 1] + epsln);
    int kn = ica * (kbl - 1) + (1 - ica) * kbl;
    kn = kn < nl1 - 1 ? kn : nl1 - 1;
    int knm1 = kn - 1 > nu1 ? kn - 1 : nu1;
    int knp1 = kn + 1 < nl1 ? kn + 1 : nl1;
    double delhat = fabs(A2(C_.Z_3d_n, kn, n)) - hbl;
    double R = 1.0 - delhat / dthick[kn];
    double p[3], h[3];                                  /* 0: momentum, 1: T, 2: S (columns of diff_col) */
    for (int j = 0; j < 3; j++) {
      double dvdzup = (dc[j][knm1] - dc[j][kn]) / dthick[kn];
      double dvdzdn = (dc[j][kn] - dc[j][knp1]) / dthick[knp1];
      p[j] = 0.5 * ((1.0 - R) * (dvdzup + fabs(dvdzup)) + R * (dvdzdn + fabs(dvdzdn)));
      h[j] = dc[j][kn] + p[j] * delhat;
    }
    const double viscp = p[0], diftp = p[1], difsp = p[2], visch = h[0], difth = h[1], difsh = h[2];
    double us2 = us * us;
    double f1 = stable * conc1 * bfsfc / (us2 * us2 + epsln);
    double gat1m = visch / (hbl + epsln) / (wm + epsln);
    double dat1m = -viscp / (wm + epsln) + f1 * visch;
    dat1m = dmin(dat1m, 0.0);
    double gat1s = difsh / (hbl + epsln) / (ws + epsln);
    double dat1s = -difsp / (ws + epsln) + f1 * difsh;
    dat1s = dmin(dat1s, 0.0);
    double gat1t = difth / (hbl + epsln) / (ws + epsln);
    double dat1t = -diftp / (ws + epsln) + f1 * difth;
    dat1t = dmin(dat1t, 0.0);
    for (int nz = nu1 + 1; nz <= nl1 - 1; nz++) {
      if (nz >= kbl) break;
      double sig = fabs(A2(C_.Z_3d_n, nz, n)) / (hbl + epsln);
      sigma = stable * sig + (1.0 - stable) * dmin(sig, epsilon_kpp);
      zehat = vonk * sigma * hbl * bfsfc;
      wscale(zehat, us, &wm, &ws);
      double a1 = sig - 2.0, a2 = 3.0 - 2.0 * sig, a3 = sig - 1.0;
      double Gm = a1 + a2 * gat1m + a3 * dat1m, Gs = a1 + a2 * gat1s + a3 * dat1s, Gt = a1 + a2 * gat1t + a3 * dat1t;
      BLMC(1, nz, n) = hbl * wm * sig * (1.0 + sig * Gm);
      BLMC(2, nz, n) = hbl * ws * sig * (1.0 + sig * Gt);
      BLMC(3, nz, n) = hbl * ws * sig * (1.0 + sig * Gs);
      A2(C_.kpp_ghats, nz, n) = (1.0 - stable) * C_.kpp_cg / (ws * hbl + epsln);
    }
    double sig = fabs(A2L(C_.zbar_3d_n, kbl - 1, n)) / (hbl + epsln);
    sigma = stable * sig + (1.0 - stable) * dmin(sig, epsilon_kpp);
    zehat = vonk * sigma * hbl * bfsfc;
    wscale(zehat, us, &wm, &ws);
    double a1 = sig - 2.0, a2 = 3.0 - 2.0 * sig, a3 = sig - 1.0;
    double Gm = a1 + a2 * gat1m + a3 * dat1m, Gs = a1 + a2 * gat1s + a3 * dat1s, Gt = a1 + a2 * gat1t + a3 * dat1t;
    C_.kpp_dkm1[3 * (n - 1) + 0] = hbl * wm * sig * (1.0 + sig * Gm);
    C_.kpp_dkm1[3 * (n - 1) + 1] = hbl * ws * sig * (1.0 + sig * Gt);
    C_.kpp_dkm1[3 * (n - 1) + 2] = hbl * ws * sig * (1.0 + sig * Gs);
  }
}

/* enhance :1152-1191 */
static void enhance(void) {
  for (int n = 1; n <= C_.m.myDim_nod2D; n++) {
    int k = C_.kpp_kbl[n - 1] - 1;
    const double caseA = C_.kpp_caseA[n - 1], hbl = C_.kpp_hbl[n - 1];
    double delta = (hbl + A2L(C_.zbar_3d_n, k, n)) / (A2L(C_.zbar_3d_n, k, n) - A2L(C_.zbar_3d_n, k + 1, n));
    const double omd = 1.0 - delta;
    double *intr[3] = {C_.kpp_viscA, C_.kpp_Kv1, C_.kpp_Kv2};
    for (int j = 0; j < 3; j++) {
      double v = A2L(intr[j], k, n);
      double dkmp5 = caseA * v + (1.0 - caseA) * BLMC(j + 1, k, n);
      double dstar = omd * omd * C_.kpp_dkm1[3 * (n - 1) + j] + delta * delta * dkmp5;
      BLMC(j + 1, k, n) = omd * v + delta * dstar;
    }
    A2(C_.kpp_ghats, k, n) = (1.0 - caseA) * A2(C_.kpp_ghats, k, n);
  }
}

/* smooth_nod3D, src/gen_support.F90:78-178 (nlev = nl; single partition: the exchanges are no-ops) */
static void smooth_nod3D(double *arr, int nsmooth) {
  double *work = C_.kpp_work, *vol = C_.kpp_vol;
  const int nlev = NL;
  for (int q = 0; q < nsmooth; q++) {
    for (int n = 1; n <= C_.m.myDim_nod2D; n++) {
      int uln = ULEVN(n), nln = nlev < NLEVN(n) ? nlev : NLEVN(n);
      for (int nz = 1; nz <= nln; nz++) { A2L(work, nz, n) = 0.; if (q == 0) A2L(vol, nz, n) = 0.; }
      for (int j = 1; j <= C_.m.nod_in_elem2D_num[n - 1]; j++) {
        int el = NIE(j, n);
        int ule = uln > ULEV(el) ? uln : ULEV(el);
        int nle = NLEV(el) < nlev ? NLEV(el) : nlev;
        nle = nln < nle ? nln : nle;
        double ar = C_.m.elem_area[el - 1];
        for (int nz = ule; nz <= nle; nz++) {
          if (q == 0) A2L(vol, nz, n) = A2L(vol, nz, n) + ar;
          A2L(work, nz, n) = A2L(work, nz, n) + ar * (A2L(arr, nz, EN(1, el)) + A2L(arr, nz, EN(2, el)) + A2L(arr, nz, EN(3, el)));
        }
      }
      if (q == 0) for (int nz = uln; nz <= nln; nz++) A2L(vol, nz, n) = 1. / (3. * A2L(vol, nz, n));
    }
    for (int n = 1; n <= C_.m.myDim_nod2D; n++) {
      int uln = ULEVN(n), nln = nlev < NLEVN(n) ? nlev : NLEVN(n);
      for (int nz = uln; nz <= nln; nz++) A2L(arr, nz, n) = A2L(work, nz, n) * A2L(vol, nz, n);
    }
  }
}

/* oce_mixing_KPP :240-432 followed by `Kv = Kv_double(:,:,1)` (src/oce_ale.F90:2609-2610) */
void orc_mixing_kpp(void) {
  if (!C_.kpp_wmt) orc_kpp_init();
  memset(C_.kpp_viscA, 0, sizeof(double) * (size_t)NL * C_.N);
  for (int n = 1; n <= C_.m.myDim_nod2D; n++) {
    int nzmin = ULEVN(n), nzmax = NLEVN(n);
    A2L(C_.kpp_dVsq, nzmin, n) = 0.0;
    double usurf = V2(C_.Unode, 1, nzmin, n), vsurf = V2(C_.Unode, 2, nzmin, n);
    for (int nz = nzmin + 1; nz <= nzmax - 1; nz++) {
      double u_loc = 0.5 * (V2(C_.Unode, 1, nz - 1, n) + V2(C_.Unode, 1, nz, n));
      double v_loc = 0.5 * (V2(C_.Unode, 2, nz - 1, n) + V2(C_.Unode, 2, nz, n));
      double du = usurf - u_loc, dv = vsurf - v_loc;
      A2L(C_.kpp_dVsq, nz, n) = du * du + dv * dv;
    }
    A2L(C_.kpp_dVsq, nzmax, n) = A2L(C_.kpp_dVsq, nzmax - 1, n);
  }
  for (int n = 1; n <= C_.m.myDim_nod2D; n++) {
    int nzmin = ULEVN(n);
    double sx = C_.stress_atmoce_x[n - 1], sy = C_.stress_atmoce_y[n - 1];
    C_.kpp_ustar[n - 1] = sqrt(sqrt(sx * sx + sy * sy) * (1.0 / DENSITY_0));
    C_.kpp_Bo[n - 1] = -G_ACC * (A2(C_.sw_alpha, nzmin, n) * C_.heat_flux[n - 1] / VCPW +
                                A2(C_.sw_beta, nzmin, n) * C_.water_flux[n - 1] * TR(nzmin, n, 2));
  }
  ri_iwmix();
  if (C_.p.double_diffusion) ddmix();
  bldepth();
  blmix_kpp();
  enhance();
  for (int j = 0; j < 3; j++) smooth_nod3D(C_.kpp_blmc[j], 3);
  for (int n = 1; n <= C_.m.myDim_nod2D; n++) {
    int nzmin = ULEVN(n), nzmax = NLEVN(n);
    for (int nz = nzmin + 1; nz <= nzmax - 1; nz++) {
      if (nz < C_.kpp_kbl[n - 1]) {
        A2L(C_.kpp_viscA, nz, n) = dmax(A2L(C_.kpp_viscA, nz, n), BLMC(1, nz, n));
        A2L(C_.kpp_Kv1, nz, n) = dmax(A2L(C_.kpp_Kv1, nz, n), BLMC(2, nz, n));
        A2L(C_.kpp_Kv2, nz, n) = dmax(A2L(C_.kpp_Kv2, nz, n), BLMC(3, nz, n));
      } else A2(C_.kpp_ghats, nz, n) = 0.0;
    }
  }
  const double minmix = 3.0e-3;
  for (int e = 1; e <= C_.m.myDim_elem2D; e++) {
    int n1 = EN(1, e), n2 = EN(2, e), n3 = EN(3, e), nzmin = ULEV(e), nzmax = NLEV(e);
    for (int nz = nzmin; nz <= nzmax - 1; nz++)
      A2L(C_.Av, nz, e) = (A2L(C_.kpp_viscA, nz, n1) + A2L(C_.kpp_viscA, nz, n2) + A2L(C_.kpp_viscA, nz, n3)) / 3.0;
    A2L(C_.Av, nzmax, e) = A2L(C_.Av, nzmax - 1, e);
    if (A2L(C_.Av, nzmin, e) < minmix) A2L(C_.Av, nzmin, e) = minmix;
  }
  memcpy(C_.Kv, C_.kpp_Kv1, sizeof(double) * (size_t)NL * C_.N);
}
