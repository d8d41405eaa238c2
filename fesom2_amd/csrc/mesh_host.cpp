// Host mesh layer (setup, untimed).  Builds every array of the reference's t_mesh and
// the static/initial ALE arrays from the ASCII mesh files, in the reference's operation
// order so that the results are bit-identical to mesh_setup/ocean_setup:
//   read_mesh              src/oce_mesh.F90:147-696      test_tri        :1353-1395
//   load_edges             :1419-1642                    find_neighbors  :1650-1800
//   find_levels(_min_e2n)  :699-815,1322-1350            mesh_areas      :1840-2090
//   mesh_auxiliary_arrays  :2097-2300                    r2g/g2r         src/gen_modules_rotate_grid.F90
//   init_ale / bottom thickness / init_thickness_ale     src/oce_ale.F90:82-795
//   init_stiff_mat_ale     src/oce_ale.F90:1088-1354     find_up_downwind_triangles src/oce_muscl_adv.F90:124-281
// Everything is computed on the GLOBAL mesh in global order; a rank's local arrays are an
// extraction through dist_<npes>/my_list (owned lists are in increasing global order, so
// owned values do not depend on the partition).
#include "../../include/fesom_gpu.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include <fstream>
#include <sstream>

namespace {
const double PI = 3.14159265358979;          // o_PARAM pi (src/oce_modules.F90:11), deliberately truncated
const double RAD = PI / 180.0;
const double R_EARTH = 6367500.0;
const double G_ACC = 9.81;
const double OMEGA = 2 * PI / (3600.0 * 24.0);

typedef std::vector<double> dvec;
typedef std::vector<int> ivec;

struct Mesh {
  fesom_mesh_opts o;
  int N2 = 0, E2 = 0, D2 = 0, D2in = 0, nl = 0, maxk = 0, nza = 0;
  double cyc = 0;
  double r2g_m[3][3];
  // global arrays (1-based content, 0-based storage)
  dvec coord, geo, depth, zbar, Z;
  ivec elem_nodes, edges, edge_tri, elem_edges, elem_nb, nie, nie_num;
  ivec nlev, ulev, nlev_n, ulev_n, nlev_n_min, ulev_n_max;
  dvec elem_area, area, area_inv, areasvol, areasvol_inv, resol;
  dvec grad_sca, grad_vec, edge_dxdy, edge_cross, elem_cos, metric, cori, cori_n, cen_y;
  ivec rowptr, colind, colind_loc; dvec values;
  ivec updn;
  dvec zbar_n_bot, zbar_n_srf, bot_n_th, zbar_e_bot, zbar_e_srf, bot_e_th, cavity_depth;
  ivec list_n, list_e, list_d;
  // initial state
  dvec hnode, hnode_new, helem, zbar3, Z3, eta, d_eta, ssh_rhs, ssh_rhs_old, hbar, hbar_old, dhe;
  dvec tr, tr_old, UV, UVab, W, We, Wi, values_state;
  fesom_mesh_desc desc;
  fesom_part_desc part;
  fesom_state_desc st;
  ivec com_dummy;
  // ---- partition (npes > 1): node owner, local lists, communication lists, local copies of every array
  struct Com { ivec rPE, rptr, rlist, sPE, sptr, slist; };
  ivec owner;                                   // rank of every global node (0-based ranks)
  int myN = 0, eN = 0, myE = 0, eE = 0, eX = 0, myD = 0, eD = 0;
  Com cn, ce, cf;                               // com_nod2D, com_elem2D, com_elem2D_full (local 1-based lists after localisation)
  std::vector<dvec> ld;                         // local double arrays (kept alive for desc / state pointers)
  std::vector<ivec> li;
  bool local = false;
};

bool read_all_tokens(const std::string &fn, std::vector<std::string> &tok) {
  std::ifstream f(fn);
  if (!f) return false;
  std::stringstream ss; ss << f.rdbuf();
  std::string t;
  while (ss >> t) tok.push_back(t);
  return true;
}

inline void trim_cyclic(double &b, double cyc) {
  if (b > cyc / 2.0) b = b - cyc;
  if (b < -cyc / 2.0) b = b + cyc;
}

void set_rot(Mesh &m) {
  double al = m.o.alphaEuler_deg * RAD, be = m.o.betaEuler_deg * RAD, ga = m.o.gammaEuler_deg * RAD;
  m.r2g_m[0][0] = cos(ga) * cos(al) - sin(ga) * cos(be) * sin(al);
  m.r2g_m[0][1] = cos(ga) * sin(al) + sin(ga) * cos(be) * cos(al);
  m.r2g_m[0][2] = sin(ga) * sin(be);
  m.r2g_m[1][0] = -sin(ga) * cos(al) - cos(ga) * cos(be) * sin(al);
  m.r2g_m[1][1] = -sin(ga) * sin(al) + cos(ga) * cos(be) * cos(al);
  m.r2g_m[1][2] = cos(ga) * sin(be);
  m.r2g_m[2][0] = sin(be) * sin(al);
  m.r2g_m[2][1] = -sin(be) * cos(al);
  m.r2g_m[2][2] = cos(be);
}
void r2g(const Mesh &m, double &glon, double &glat, double rlon, double rlat) {
  double xr = cos(rlat) * cos(rlon), yr = cos(rlat) * sin(rlon), zr = sin(rlat);
  double xg = m.r2g_m[0][0] * xr + m.r2g_m[1][0] * yr + m.r2g_m[2][0] * zr;
  double yg = m.r2g_m[0][1] * xr + m.r2g_m[1][1] * yr + m.r2g_m[2][1] * zr;
  double zg = m.r2g_m[0][2] * xr + m.r2g_m[1][2] * yr + m.r2g_m[2][2] * zr;
  glat = asin(zg);
  if (yg == 0. && xg == 0.) glon = 0.0; else glon = atan2(yg, xg);
}
void g2r(const Mesh &m, double glon, double glat, double &rlon, double &rlat) {
  double xg = cos(glat) * cos(glon), yg = cos(glat) * sin(glon), zg = sin(glat);
  double xr = m.r2g_m[0][0] * xg + m.r2g_m[0][1] * yg + m.r2g_m[0][2] * zg;
  double yr = m.r2g_m[1][0] * xg + m.r2g_m[1][1] * yg + m.r2g_m[1][2] * zg;
  double zr = m.r2g_m[2][0] * xg + m.r2g_m[2][1] * yg + m.r2g_m[2][2] * zg;
  rlat = asin(zr);
  if (yr == 0. && xr == 0.) rlon = 0.0; else rlon = atan2(yr, xr);
}

#define CO(k, n) m.coord[2 * ((n) - 1) + (k) - 1]
#define EN(j, e) m.elem_nodes[3 * ((e) - 1) + (j) - 1]
#define ED(j, d) m.edges[2 * ((d) - 1) + (j) - 1]
#define ET(j, d) m.edge_tri[2 * ((d) - 1) + (j) - 1]
#define EE(j, e) m.elem_edges[3 * ((e) - 1) + (j) - 1]
#define ENB(j, e) m.elem_nb[3 * ((e) - 1) + (j) - 1]
#define NIE(j, n) m.nie[m.maxk * ((n) - 1) + (j) - 1]

void edge_center(const Mesh &m, int n1, int n2, double &x, double &y) {
  double a1 = CO(1, n1), a2 = CO(2, n1), b1 = CO(1, n2), b2 = CO(2, n2);
  if (a1 - b1 > m.cyc / 2.0) a1 = a1 - m.cyc;
  if (a1 - b1 < -m.cyc / 2.0) b1 = b1 - m.cyc;
  x = 0.5 * (a1 + b1);
  y = 0.5 * (a2 + b2);
}
void elem_center(const Mesh &m, int e, double &x, double &y) {
  double ax[3];
  for (int k = 0; k < 3; k++) ax[k] = CO(1, EN(k + 1, e));
  double amin = std::min(ax[0], std::min(ax[1], ax[2]));
  for (int k = 0; k < 3; k++) {
    if (ax[k] - amin >= m.cyc / 2.0) ax[k] = ax[k] - m.cyc;
    if (ax[k] - amin < -m.cyc / 2.0) ax[k] = ax[k] + m.cyc;
  }
  x = (ax[0] + ax[1] + ax[2]) / 3.0;
  y = (CO(2, EN(1, e)) + CO(2, EN(2, e)) + CO(2, EN(3, e))) / 3.0;
}


// Edge list of a mesh that comes without edges.out / edge_tri.out / edgenum.out (generated meshes): restatement of the
// reference partitioner's find_edges_ini (src/fvom_init.F90:315-650).  Internal edges first, then boundary edges; within
// each group in the order (lower node ascending, neighbours in order of discovery over the node's elements); edge_tri(1)
// is the triangle to the left of the edge.  For the reference's own meshes it reproduces their edge files entry for entry
// (tests/test_abi_and_host.py::test_generated_edges_equal_mesh_files).
void generate_edges(Mesh &m) {
  const int N = m.N2, E = m.E2;
  std::vector<std::vector<int>> ne(N + 1), nn(N + 1);
  for (int e = 1; e <= E; e++) for (int q = 1; q <= 3; q++) ne[EN(q, e)].push_back(e);
  std::vector<char> seen(N + 1, 0);
  for (int n = 1; n <= N; n++) {
    for (int e : ne[n]) for (int q = 1; q <= 3; q++) { int k = EN(q, e); if (k != n && !seen[k]) { seen[k] = 1; nn[n].push_back(k); } }
    for (int k : nn[n]) seen[k] = 0;
  }
  m.edges.clear(); m.edge_tri.clear();
  for (int pass = 0; pass < 2; pass++) {               // 0: internal (two triangles), 1: boundary (one)
    for (int n = 1; n <= N; n++)
      for (int node : nn[n]) {
        if (node < n) continue;
        int flag = 0, el[2] = {0, 0};
        for (int e : ne[n]) for (int q = 1; q <= 3; q++) if (EN(q, e) == node) { if (flag < 2) el[flag] = e; flag++; break; }
        if ((pass == 0 && flag == 2) || (pass == 1 && flag == 1)) {
          m.edges.push_back(n); m.edges.push_back(node);
          m.edge_tri.push_back(el[0]); m.edge_tri.push_back(pass == 0 ? el[1] : -999);
        }
      }
    if (pass == 0) m.D2in = (int)m.edges.size() / 2;
  }
  m.D2 = (int)m.edges.size() / 2;
  for (int d = 1; d <= m.D2; d++) {                     // orientation: first triangle on the left of (edges(1) -> edges(2))
    if (ET(1, d) <= 0) std::swap(ET(1, d), ET(2, d));
    double xc[2], xe[2];
    elem_center(m, ET(1, d), xc[0], xc[1]);
    xc[0] -= CO(1, ED(1, d)); xc[1] -= CO(2, ED(1, d));
    xe[0] = CO(1, ED(2, d)) - CO(1, ED(1, d)); xe[1] = CO(2, ED(2, d)) - CO(2, ED(1, d));
    trim_cyclic(xe[0], m.cyc); trim_cyclic(xc[0], m.cyc);
    if (xc[0] * xe[1] - xc[1] * xe[0] > 0.0) {
      if (ET(2, d) > 0) std::swap(ET(1, d), ET(2, d));
      else std::swap(ED(1, d), ED(2, d));
    }
  }
  for (auto &v : m.edge_tri) if (v < 0) v = 0;
}

bool load_files(Mesh &m, const std::string &dir) {
  std::vector<std::string> t;
  if (!read_all_tokens(dir + "/nod2d.out", t)) return false;
  m.N2 = atoi(t[0].c_str());
  m.coord.resize(2 * m.N2);
  for (int n = 0; n < m.N2; n++) {
    double lon = atof(t[1 + 4 * n + 1].c_str()), lat = atof(t[1 + 4 * n + 2].c_str());
    double x = lon * RAD, y = lat * RAD;
    if (m.o.force_rotation) { double rx = x, ry = y; g2r(m, rx, ry, x, y); }
    m.coord[2 * n] = x; m.coord[2 * n + 1] = y;
  }
  t.clear();
  if (!read_all_tokens(dir + "/elem2d.out", t)) return false;
  m.E2 = atoi(t[0].c_str());
  m.elem_nodes.resize(3 * m.E2);
  for (int i = 0; i < 3 * m.E2; i++) m.elem_nodes[i] = atoi(t[1 + i].c_str());
  t.clear();
  if (!read_all_tokens(dir + "/aux3d.out", t)) return false;
  m.nl = atoi(t[0].c_str());
  m.zbar.resize(m.nl);
  for (int i = 0; i < m.nl; i++) m.zbar[i] = atof(t[1 + i].c_str());
  if (m.zbar[1] > 0) for (auto &z : m.zbar) z = -z;
  m.Z.resize(m.nl - 1);
  for (int i = 0; i < m.nl - 1; i++) { m.Z[i] = m.zbar[i] + m.zbar[i + 1]; m.Z[i] = 0.5 * m.Z[i]; }
  m.depth.resize(m.N2);
  for (int n = 0; n < m.N2; n++) {
    double x = atof(t[1 + m.nl + n].c_str());
    if (x > 0) x = -x;
    if (x > m.zbar[4]) x = m.zbar[4];
    m.depth[n] = x;
  }
  t.clear();
  const bool gen_edges = getenv("FESOM_MESH_GENERATE_EDGES") != nullptr;      // (tests: ignore the edge files)
  if (gen_edges || !read_all_tokens(dir + "/edgenum.out", t)) {
    t.clear();
    generate_edges(m);
  } else {
    m.D2 = atoi(t[0].c_str()); m.D2in = atoi(t[1].c_str());
    t.clear();
    if (!read_all_tokens(dir + "/edges.out", t)) return false;
    m.edges.resize(2 * m.D2);
    for (int i = 0; i < 2 * m.D2; i++) m.edges[i] = atoi(t[i].c_str());
    t.clear();
    if (!read_all_tokens(dir + "/edge_tri.out", t)) return false;
    m.edge_tri.resize(2 * m.D2);
    for (int i = 0; i < 2 * m.D2; i++) { int v = atoi(t[i].c_str()); m.edge_tri[i] = v < 0 ? 0 : v; }
    t.clear();
  }
  if (!read_all_tokens(dir + "/elvls.out", t)) return false;
  m.nlev.resize(m.E2);
  for (int i = 0; i < m.E2; i++) m.nlev[i] = atoi(t[i].c_str());
  t.clear();
  if (!read_all_tokens(dir + "/nlvls.out", t)) return false;
  m.nlev_n.resize(m.N2);
  for (int i = 0; i < m.N2; i++) m.nlev_n[i] = atoi(t[i].c_str());
  m.ulev.assign(m.E2, 1);
  m.ulev_n.assign(m.N2, 1);
  m.cavity_depth.assign(m.N2, 0.0);
  if (m.o.use_cavity) {        // find_levels_cavity (src/oce_mesh.F90:897-1280): upper levels of elements and nodes, draft of the ice shelf at the nodes
    t.clear();
    if (!read_all_tokens(dir + "/cavity_elvls.out", t) || (int)t.size() < m.E2) return false;
    for (int i = 0; i < m.E2; i++) m.ulev[i] = atoi(t[i].c_str());
    t.clear();
    if (!read_all_tokens(dir + "/cavity_nlvls.out", t) || (int)t.size() < m.N2) return false;
    for (int i = 0; i < m.N2; i++) m.ulev_n[i] = atoi(t[i].c_str());
    t.clear();
    if (!read_all_tokens(dir + "/cavity_depth.out", t) || (int)t.size() < m.N2) return false;
    for (int i = 0; i < m.N2; i++) m.cavity_depth[i] = (double)atoi(t[i].c_str());       // (the reference reads it into an integer buffer, :1253)
    t.clear();
  }
  return true;
}

void test_tri(Mesh &m) {
  for (int e = 1; e <= m.E2; e++) {
    int n1 = EN(1, e), n2 = EN(2, e), n3 = EN(3, e);
    double a1 = CO(1, n1), a2 = CO(2, n1);
    double b1 = CO(1, n2) - a1, b2 = CO(2, n2) - a2, c1 = CO(1, n3) - a1, c2 = CO(2, n3) - a2;
    trim_cyclic(b1, m.cyc); trim_cyclic(c1, m.cyc);
    double r = b1 * c2 - b2 * c1;
    if (r > 0.0) { EN(2, e) = n3; EN(3, e) = n2; }
  }
}

void topology(Mesh &m) {
  // elem_edges (load_edges): edges appended in increasing edge index, then re-ordered so that edge q is opposite node q
  m.elem_edges.assign(3 * m.E2, 0);
  ivec aux(m.E2, 0);
  for (int d = 1; d <= m.D2; d++)
    for (int k = 1; k <= 2; k++) {
      int q = ET(k, d);
      if (q > 0) { aux[q - 1]++; EE(aux[q - 1], q) = d; }
    }
  for (int e = 1; e <= m.E2; e++) {
    int el[3] = {EE(1, e), EE(2, e), EE(3, e)};
    for (int q = 1; q <= 3; q++)
      for (int k = 0; k < 3; k++)
        if (ED(1, el[k]) != EN(q, e) && ED(2, el[k]) != EN(q, e)) { EE(q, e) = el[k]; break; }
  }
  m.elem_nb.assign(3 * m.E2, 0);
  for (int e = 1; e <= m.E2; e++)
    for (int j = 1; j <= 3; j++) {
      int e1 = ET(1, EE(j, e));
      if (e1 == e) e1 = ET(2, EE(j, e));
      ENB(j, e) = e1;
    }
  m.nie_num.assign(m.N2, 0);
  for (int e = 1; e <= m.E2; e++) for (int j = 1; j <= 3; j++) m.nie_num[EN(j, e) - 1]++;
  m.maxk = *std::max_element(m.nie_num.begin(), m.nie_num.end());
  m.nie.assign((size_t)m.maxk * m.N2, 0);
  std::fill(m.nie_num.begin(), m.nie_num.end(), 0);
  for (int e = 1; e <= m.E2; e++)
    for (int j = 1; j <= 3; j++) {
      int n = EN(j, e);
      m.nie_num[n - 1]++;
      NIE(m.nie_num[n - 1], n) = e;
    }
  m.nlev_n_min.resize(m.N2); m.ulev_n_max.resize(m.N2);
  for (int n = 1; n <= m.N2; n++) {
    int mn = 1 << 30, mx = 0;
    for (int j = 1; j <= m.nie_num[n - 1]; j++) {
      mn = std::min(mn, m.nlev[NIE(j, n) - 1]);
      mx = std::max(mx, m.ulev[NIE(j, n) - 1]);
    }
    m.nlev_n_min[n - 1] = mn; m.ulev_n_max[n - 1] = mx;
  }
}

void areas(Mesh &m) {
  int nl = m.nl;
  m.elem_area.resize(m.E2);
  for (int e = 1; e <= m.E2; e++) {
    int n1 = EN(1, e), n2 = EN(2, e), n3 = EN(3, e);
    double ay = (CO(2, n1) + CO(2, n2) + CO(2, n3)) / 3.0;
    ay = cos(ay);
    double a1 = CO(1, n2) - CO(1, n1), a2 = CO(2, n2) - CO(2, n1);
    double b1 = CO(1, n3) - CO(1, n1), b2 = CO(2, n3) - CO(2, n1);
    trim_cyclic(a1, m.cyc); trim_cyclic(b1, m.cyc);
    a1 = a1 * ay; b1 = b1 * ay;
    m.elem_area[e - 1] = 0.5 * fabs(a1 * b2 - b1 * a2);
  }
  m.area.assign((size_t)nl * m.N2, 0.0);
  for (int n = 1; n <= m.N2; n++)
    for (int j = 1; j <= m.nie_num[n - 1]; j++) {
      int e = NIE(j, n);
      for (int nz = m.ulev[e - 1]; nz <= m.nlev[e - 1] - 1; nz++)
        m.area[(size_t)nl * (n - 1) + nz - 1] += m.elem_area[e - 1] / 3.0;
    }
  m.areasvol.assign((size_t)nl * m.N2, 0.0);
  if (m.o.use_cavity) {      // :1927-1977: directly under the ice the scalar cell takes the area of its LOWER face where a cavity triangle sits on the upper one
    std::vector<int> contrib((size_t)nl * m.N2, 0);
    for (int n = 1; n <= m.N2; n++)
      for (int j = 1; j <= m.nie_num[n - 1]; j++) {
        int e = NIE(j, n);
        for (int nz = 1; nz <= m.ulev[e - 1] - 1; nz++) contrib[(size_t)nl * (n - 1) + nz - 1]++;
      }
    for (int n = 1; n <= m.N2; n++) {
      const int nzmin = m.ulev_n[n - 1], nzmax = m.nlev_n[n - 1] - 1;
      for (int nz = nzmin; nz <= nzmax; nz++)
        m.areasvol[(size_t)nl * (n - 1) + nz - 1] = contrib[(size_t)nl * (n - 1) + nz - 1] > 0 ? m.area[(size_t)nl * (n - 1) + std::min(nz + 1, nzmax) - 1]
                                                                                              : m.area[(size_t)nl * (n - 1) + nz - 1];
    }
  } else
  for (int n = 1; n <= m.N2; n++)
    for (int nz = m.ulev_n[n - 1]; nz <= m.nlev_n[n - 1] - 1; nz++)
      m.areasvol[(size_t)nl * (n - 1) + nz - 1] = m.area[(size_t)nl * (n - 1) + nz - 1];
  for (auto &a : m.elem_area) a = a * R_EARTH * R_EARTH;
  for (auto &a : m.area) a = a * R_EARTH * R_EARTH;
  for (auto &a : m.areasvol) a = a * R_EARTH * R_EARTH;
  m.area_inv.assign((size_t)nl * m.N2, 0.0);
  for (int n = 1; n <= m.N2; n++)
    for (int nz = m.ulev_n[n - 1]; nz <= m.nlev_n[n - 1]; nz++) {
      double a = m.area[(size_t)nl * (n - 1) + nz - 1];
      m.area_inv[(size_t)nl * (n - 1) + nz - 1] = a > 0.0 ? 1.0 / a : 0.0;
    }
  m.areasvol_inv = m.area_inv;
  if (m.o.use_cavity) {      // :2015-2028
    m.areasvol_inv.assign((size_t)nl * m.N2, 0.0);
    for (int n = 1; n <= m.N2; n++)
      for (int nz = m.ulev_n[n - 1]; nz <= m.nlev_n[n - 1] - 1; nz++) {
        double a = m.areasvol[(size_t)nl * (n - 1) + nz - 1];
        m.areasvol_inv[(size_t)nl * (n - 1) + nz - 1] = a > 0.0 ? 1.0 / a : 0.0;
      }
  }
  m.resol.resize(m.N2);
  for (int n = 1; n <= m.N2; n++)
    m.resol[n - 1] = sqrt(m.areasvol[(size_t)nl * (n - 1) + m.ulev_n[n - 1] - 1] / PI) * 2.0;
  dvec work(m.N2);
  for (int q = 0; q < 3; q++) {
    for (int n = 1; n <= m.N2; n++) {
      double vol = 0.0; work[n - 1] = 0.0;
      for (int j = 1; j <= m.nie_num[n - 1]; j++) {
        int e = NIE(j, n);
        double s = (m.resol[EN(1, e) - 1] + m.resol[EN(2, e) - 1] + m.resol[EN(3, e) - 1]);
        work[n - 1] = work[n - 1] + s / 3.0 * m.elem_area[e - 1];
        vol = vol + m.elem_area[e - 1];
      }
      work[n - 1] = work[n - 1] / vol;
    }
    m.resol = work;
  }
}

void auxiliary(Mesh &m) {
  m.cori_n.resize(m.N2); m.geo.resize(2 * m.N2);
  for (int n = 1; n <= m.N2; n++) {
    double lon, lat;
    r2g(m, lon, lat, CO(1, n), CO(2, n));
    m.cori_n[n - 1] = 2 * OMEGA * sin(lat);
    if (lon > 2.0 * PI) lon = lon - 2.0 * PI;
    if (lon < -2.0 * PI) lon = lon + 2.0 * PI;
    m.geo[2 * (n - 1)] = lon; m.geo[2 * (n - 1) + 1] = lat;
  }
  m.cori.resize(m.E2); m.elem_cos.resize(m.E2); m.metric.resize(m.E2);
  dvec cx(m.E2), cy(m.E2);
  for (int e = 1; e <= m.E2; e++) {
    double ax, ay, lon, lat;
    elem_center(m, e, ax, ay);
    r2g(m, lon, lat, ax, ay);
    m.cori[e - 1] = 2 * OMEGA * sin(lat);
    cx[e - 1] = ax; cy[e - 1] = ay;
    m.elem_cos[e - 1] = cos(ay);
  }
  // reference quirk (oce_mesh.F90:2183): the whole metric_factor array is assigned in every iteration,
  // so every entry ends up tan(center_y(last element))/r_earth
  { double v = tan(cy[m.E2 - 1]) / R_EARTH; std::fill(m.metric.begin(), m.metric.end(), v); }
  m.cen_y = cy;
  m.edge_dxdy.resize(2 * m.D2); m.edge_cross.resize(4 * m.D2);
  for (int d = 1; d <= m.D2; d++) {
    int n1 = ED(1, d), n2 = ED(2, d);
    double a1 = CO(1, n2) - CO(1, n1), a2 = CO(2, n2) - CO(2, n1);
    trim_cyclic(a1, m.cyc);
    m.edge_dxdy[2 * (d - 1)] = a1; m.edge_dxdy[2 * (d - 1) + 1] = a2;
    double ex, ey; edge_center(m, n1, n2, ex, ey);
    for (int i = 0; i < 2; i++) {
      int el = ET(i + 1, d);
      if (el > 0) {
        double b1 = cx[el - 1] - ex, b2 = cy[el - 1] - ey;
        trim_cyclic(b1, m.cyc);
        b1 = b1 * m.elem_cos[el - 1];
        b1 = b1 * R_EARTH; b2 = b2 * R_EARTH;
        m.edge_cross[4 * (d - 1) + 2 * i] = b1; m.edge_cross[4 * (d - 1) + 2 * i + 1] = b2;
      } else {
        m.edge_cross[4 * (d - 1) + 2 * i] = 0.0; m.edge_cross[4 * (d - 1) + 2 * i + 1] = 0.0;
      }
    }
  }
  m.grad_sca.resize(6 * m.E2); m.grad_vec.resize(6 * m.E2);
  for (int e = 1; e <= m.E2; e++) {
    int n1 = EN(1, e), n2 = EN(2, e), n3 = EN(3, e);
    double dX31 = CO(1, n3) - CO(1, n1); trim_cyclic(dX31, m.cyc); dX31 = m.elem_cos[e - 1] * dX31;
    double dX21 = CO(1, n2) - CO(1, n1); trim_cyclic(dX21, m.cyc); dX21 = m.elem_cos[e - 1] * dX21;
    double dY31 = CO(2, n3) - CO(2, n1), dY21 = CO(2, n2) - CO(2, n1);
    double df = -0.5 * R_EARTH / m.elem_area[e - 1];
    double *gs = &m.grad_sca[6 * (e - 1)];
    gs[0] = (-dY31 + dY21) * df; gs[1] = dY31 * df; gs[2] = -dY21 * df;
    gs[3] = (dX31 - dX21) * df;  gs[4] = -dX31 * df; gs[5] = dX21 * df;
  }
  for (int e = 1; e <= m.E2; e++) {
    double a1 = cx[e - 1], a2 = cy[e - 1], x[3], y[3];
    for (int j = 1; j <= 3; j++) {
      int el = ENB(j, e);
      if (el > 0) {
        x[j - 1] = cx[el - 1] - a1; trim_cyclic(x[j - 1], m.cyc);
        y[j - 1] = cy[el - 1] - a2;
      } else {
        int d = EE(j, e); double b1, b2;
        edge_center(m, ED(1, d), ED(2, d), b1, b2);
        x[j - 1] = (b1 - a1); trim_cyclic(x[j - 1], m.cyc);
        x[j - 1] = 2 * x[j - 1];
        y[j - 1] = 2 * (b2 - a2);
      }
    }
    for (int j = 0; j < 3; j++) { x[j] = x[j] * m.elem_cos[e - 1] * R_EARTH; y[j] = y[j] * R_EARTH; }
    double cxx = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
    double cxy = x[0] * y[0] + x[1] * y[1] + x[2] * y[2];
    double cyy = y[0] * y[0] + y[1] * y[1] + y[2] * y[2];
    double dd = cxy * cxy - cxx * cyy;
    for (int j = 0; j < 3; j++) {
      m.grad_vec[6 * (e - 1) + j] = (cxy * y[j] - cyy * x[j]) / dd;
      m.grad_vec[6 * (e - 1) + 3 + j] = (cxy * x[j] - cxx * y[j]) / dd;
    }
  }
}

void ale_init(Mesh &m) {
  int nl = m.nl;
  m.zbar_e_bot.assign(m.E2, 0.0); m.bot_e_th.assign(m.E2, 0.0);
  for (int e = 1; e <= m.E2; e++) {
    int nle = m.nlev[e - 1];
    if (m.o.use_partial_cell) {
      double dd = (m.depth[EN(1, e) - 1] + m.depth[EN(2, e) - 1] + m.depth[EN(3, e) - 1]) / 3.0;
      if (m.zbar[nle - 2] - m.zbar[nle - 1] <= 0.0 /*partial_cell_thresh*/) {
        m.zbar_e_bot[e - 1] = m.zbar[nle - 1];
        m.bot_e_th[e - 1] = m.zbar[nle - 2] - m.zbar_e_bot[e - 1];
        continue;
      }
      if (dd < m.zbar[nle - 1]) {
        if (nle == nl) m.zbar_e_bot[e - 1] = std::max(dd, m.zbar[nle - 1] + (m.zbar[nle - 1] - m.Z[nle - 2]));
        else m.zbar_e_bot[e - 1] = std::max(m.Z[nle - 1], dd);
      } else {
        m.zbar_e_bot[e - 1] = std::min(m.Z[nle - 2], dd);
      }
      m.bot_e_th[e - 1] = m.zbar[nle - 2] - m.zbar_e_bot[e - 1];
    } else {
      m.bot_e_th[e - 1] = m.zbar[nle - 2] - m.zbar[nle - 1];
      m.zbar_e_bot[e - 1] = m.zbar[nle - 1];
    }
  }
  m.zbar_n_bot.assign(m.N2, 0.0); m.bot_n_th.assign(m.N2, 0.0);
  for (int n = 1; n <= m.N2; n++) {
    int nln = m.nlev_n[n - 1];
    if (m.o.use_partial_cell) {
      double mn = m.zbar_e_bot[NIE(1, n) - 1];
      for (int j = 2; j <= m.nie_num[n - 1]; j++) mn = std::min(mn, m.zbar_e_bot[NIE(j, n) - 1]);
      m.zbar_n_bot[n - 1] = mn;
    } else m.zbar_n_bot[n - 1] = m.zbar[nln - 1];
    m.bot_n_th[n - 1] = m.zbar[nln - 2] - m.zbar_n_bot[n - 1];
  }
  m.zbar_n_srf.assign(m.N2, m.zbar[0]); m.zbar_e_srf.assign(m.E2, m.zbar[0]);
  if (m.o.use_cavity) {      // init_surface_elem_depth / init_surface_node_depth (src/oce_ale.F90:422-545)
    for (int e = 1; e <= m.E2; e++) {
      const int ule = m.ulev[e - 1];
      if (ule == 1) continue;
      double zs = m.zbar[ule - 1];
      if (m.o.use_cavity_partial_cell && !(m.zbar[ule - 1] - m.zbar[ule] <= m.o.cavity_partial_cell_thresh)) {
        // partial surface cell: the mean draft of the element, at most half a layer away from the full-cell interface (:455-486)
        const double dd = (m.cavity_depth[EN(1, e) - 1] + m.cavity_depth[EN(2, e) - 1] + m.cavity_depth[EN(3, e) - 1]) / 3.0;
        zs = (dd < m.zbar[ule - 1]) ? std::max(m.Z[ule - 1], dd) : std::min(m.Z[ule - 2], dd);
      }
      m.zbar_e_srf[e - 1] = zs;
    }
    for (int n = 1; n <= m.N2; n++) {
      const int uln = m.ulev_n[n - 1];
      if (uln == 1) continue;
      if (m.o.use_cavity_partial_cell) {                    // the highest surface of the elements around the node (:524-538)
        double mx = m.zbar_e_srf[NIE(1, n) - 1];
        for (int j = 2; j <= m.nie_num[n - 1]; j++) mx = std::max(mx, m.zbar_e_srf[NIE(j, n) - 1]);
        m.zbar_n_srf[n - 1] = mx;
      } else m.zbar_n_srf[n - 1] = m.zbar[uln - 1];
    }
  }
  m.zbar3.assign((size_t)nl * m.N2, 0.0); m.Z3.assign((size_t)(nl - 1) * m.N2, 0.0);
  for (int n = 1; n <= m.N2; n++) {
    double *zb = &m.zbar3[(size_t)nl * (n - 1)] - 1;   // 1-based
    double *zz = &m.Z3[(size_t)(nl - 1) * (n - 1)] - 1;
    int nzmin = m.ulev_n[n - 1], nzmax = m.nlev_n[n - 1];
    for (int k = 1; k <= nzmin - 1; k++) zb[k] = m.zbar[k - 1];
    zb[nzmin] = m.zbar_n_srf[n - 1];
    for (int k = nzmin + 1; k <= nzmax - 1; k++) zb[k] = m.zbar[k - 1];
    zb[nzmax] = m.zbar_n_bot[n - 1];
    for (int k = 1; k <= nzmin - 1; k++) zz[k] = m.Z[k - 1];
    zz[nzmin] = zb[nzmin] + (zb[nzmin + 1] - m.zbar_n_srf[n - 1]) / 2;
    for (int k = nzmin + 1; k <= nzmax - 2; k++) zz[k] = m.Z[k - 1];
    zz[nzmax - 1] = zb[nzmax - 1] + (m.zbar_n_bot[n - 1] - zb[nzmax - 1]) / 2;
  }
  // init_thickness_ale (hbar = hbar_old = 0 at start)
  m.hbar.assign(m.N2, 0.0); m.hbar_old.assign(m.N2, 0.0); m.dhe.assign(m.E2, 0.0);
  m.ssh_rhs_old.assign(m.N2, 0.0); m.eta.assign(m.N2, 0.0); m.d_eta.assign(m.N2, 0.0); m.ssh_rhs.assign(m.N2, 0.0);
  m.hnode.assign((size_t)(nl - 1) * m.N2, 0.0); m.helem.assign((size_t)(nl - 1) * m.E2, 0.0);
  for (int n = 1; n <= m.N2; n++)
    m.ssh_rhs_old[n - 1] = (m.hbar[n - 1] - m.hbar_old[n - 1]) * m.areasvol[(size_t)nl * (n - 1) + m.ulev_n[n - 1] - 1] / m.o.dt;
  for (int n = 1; n <= m.N2; n++) m.eta[n - 1] = m.o.alpha * m.hbar_old[n - 1] + (1.0 - m.o.alpha) * m.hbar[n - 1];
  for (int n = 1; n <= m.N2; n++) {
    double *h = &m.hnode[(size_t)(nl - 1) * (n - 1)] - 1;
    double *zb = &m.zbar3[(size_t)nl * (n - 1)] - 1;
    int nzmin = m.ulev_n[n - 1], nzmax = m.nlev_n[n - 1] - 1;
    if (m.o.which_ale == 0) {
      for (int nz = nzmin; nz <= nzmax - 1; nz++) h[nz] = (zb[nz] - zb[nz + 1]);
    } else if (m.o.which_ale == 1) {   // zlevel (oce_ale.F90:630-660): the whole ssh variation in the top layer
      h[nzmin] = (nzmin == 1 ? m.hbar[n - 1] : 0.0) + (zb[nzmin] - zb[nzmin + 1]);
      for (int nz = nzmin + 1; nz <= nzmax - 1; nz++) h[nz] = (zb[nz] - zb[nz + 1]);
    } else {   // zstar
      if (nzmin == 1) {
        int nmin = m.nlev_n_min[n - 1];
        double dd = m.zbar[nzmin - 1] - m.zbar[nmin - 2];
        for (int nz = nzmin; nz <= nmin - 2; nz++) h[nz] = (m.zbar[nz - 1] - m.zbar[nz]) * (1.0 + m.hbar[n - 1] / dd);
        for (int nz = nmin - 1; nz <= nzmax - 1; nz++) h[nz] = (m.zbar[nz - 1] - m.zbar[nz]);
      } else
        for (int nz = nzmin; nz <= nzmax - 1; nz++) h[nz] = (zb[nz] - zb[nz + 1]);
    }
    h[nzmax] = m.bot_n_th[n - 1];
  }
  for (int e = 1; e <= m.E2; e++) {
    double *he = &m.helem[(size_t)(nl - 1) * (e - 1)] - 1;
    int nzmin = m.ulev[e - 1], nzmax = m.nlev[e - 1] - 1;
    int n1 = EN(1, e), n2 = EN(2, e), n3 = EN(3, e);
    if (m.o.which_ale == 0) {
      he[nzmin] = (m.zbar_e_srf[e - 1] - m.zbar[nzmin]);
      for (int nz = nzmin + 1; nz <= nzmax - 1; nz++) he[nz] = (m.zbar[nz - 1] - m.zbar[nz]);
    } else if (m.o.which_ale == 1) {   // zlevel (oce_ale.F90:663-690)
      m.dhe[e - 1] = (m.hbar[n1 - 1] + m.hbar[n2 - 1] + m.hbar[n3 - 1]) / 3.0;
      he[nzmin] = (nzmin == 1 ? m.dhe[e - 1] : 0.0) + (m.zbar_e_srf[e - 1] - m.zbar[nzmin]);
      for (int nz = nzmin + 1; nz <= nzmax - 1; nz++) he[nz] = (m.zbar[nz - 1] - m.zbar[nz]);
    } else {
      m.dhe[e - 1] = (m.hbar[n1 - 1] + m.hbar[n2 - 1] + m.hbar[n3 - 1]) / 3.0;
      for (int nz = nzmin; nz <= nzmax - 1; nz++)
        he[nz] = (m.hnode[(size_t)(nl - 1) * (n1 - 1) + nz - 1] + m.hnode[(size_t)(nl - 1) * (n2 - 1) + nz - 1] +
                  m.hnode[(size_t)(nl - 1) * (n3 - 1) + nz - 1]) / 3.0;
      if (nzmin > 1) { m.dhe[e - 1] = 0.0; he[nzmin] = m.zbar_e_srf[e - 1] - m.zbar[nzmin]; }      // under the shelf the surface is fixed (oce_ale.F90:757-763)
    }
    he[nzmax] = m.bot_e_th[e - 1];
  }
  m.hnode_new = m.hnode;
}

void stiff_mat(Mesh &m) {
  int N = m.N2;
  ivec n_num(N, 1);
  std::vector<ivec> n_pos(N);
  for (int n = 1; n <= N; n++) n_pos[n - 1].push_back(n);
  for (int d = 1; d <= m.D2; d++) {
    int n1 = ED(1, d), n2 = ED(2, d);
    n_pos[n1 - 1].push_back(n2); n_pos[n2 - 1].push_back(n1);
  }
  m.rowptr.resize(N + 1); m.rowptr[0] = 1;
  for (int n = 1; n <= N; n++) m.rowptr[n] = m.rowptr[n - 1] + (int)n_pos[n - 1].size();
  m.nza = m.rowptr[N] - 1;
  m.colind.resize(m.nza); m.values.assign(m.nza, 0.0);
  for (int n = 1; n <= N; n++)
    for (size_t k = 0; k < n_pos[n - 1].size(); k++) m.colind[m.rowptr[n - 1] - 1 + k] = n_pos[n - 1][k];
  m.colind_loc = m.colind;
  ivec pos(N, 0);
  double factor = G_ACC * m.o.dt * m.o.alpha * m.o.theta;
  for (int d = 1; d <= m.D2; d++) {
    for (int i = 1; i <= 2; i++) {
      int el = ET(i, d);
      if (el < 1) continue;
      double fy[3];
      const double *gs = &m.grad_sca[6 * (el - 1)];
      double c2i = m.edge_cross[4 * (d - 1) + 2 * i - 1], c2im1 = m.edge_cross[4 * (d - 1) + 2 * i - 2];
      for (int k = 0; k < 3; k++) fy[k] = (m.zbar_e_bot[el - 1] - m.zbar_e_srf[el - 1]) * (gs[k] * c2i - gs[3 + k] * c2im1);
      if (i == 2) for (int k = 0; k < 3; k++) fy[k] = -fy[k];
      for (int j = 1; j <= 2; j++) {
        int row = ED(j, d);
        for (int q = m.rowptr[row - 1]; q <= m.rowptr[row] - 1; q++) pos[m.colind[q - 1] - 1] = q;
        for (int k = 0; k < 3; k++) {
          int p = pos[EN(k + 1, el) - 1];
          if (j == 1) m.values[p - 1] = m.values[p - 1] + fy[k] * factor;
          else m.values[p - 1] = m.values[p - 1] - fy[k] * factor;
        }
      }
    }
  }
  for (int row = 1; row <= N; row++) {
    if (m.ulev_n[row - 1] > 1) continue;
    int off = m.rowptr[row - 1];
    m.values[off - 1] = m.values[off - 1] + m.areasvol[(size_t)m.nl * (row - 1) + m.ulev_n[row - 1] - 1] / m.o.dt;
  }
}

void updn_tri(Mesh &m) {
  m.updn.assign(2 * m.D2, 0);
  for (int d = 1; d <= m.D2; d++) {
    int en[2] = {ED(1, d), ED(2, d)};
    double x1 = CO(1, en[1]) - CO(1, en[0]), x2 = CO(2, en[1]) - CO(2, en[0]);
    if (x1 > m.cyc / 2.0) x1 = x1 - m.cyc;
    if (x1 < -m.cyc / 2.0) x1 = x1 + m.cyc;
    for (int side = 0; side < 2; side++) {
      x1 = -x1; x2 = -x2;            // first node: x=-x ; second node: x=-x again
      int nd = en[side];
      for (int k = 1; k <= m.nie_num[nd - 1]; k++) {
        int e = NIE(k, nd);
        int p[3] = {EN(1, e), EN(2, e), EN(3, e)};
        int i0, ib, ic;
        if (p[0] == nd) { i0 = 0; ib = 1; ic = 2; }
        else if (p[1] == nd) { i0 = 1; ib = 0; ic = 2; }
        else { i0 = 2; ib = 0; ic = 1; }
        double b1 = CO(1, p[ib]) - CO(1, p[i0]), b2 = CO(2, p[ib]) - CO(2, p[i0]);
        double c1 = CO(1, p[ic]) - CO(1, p[i0]), c2 = CO(2, p[ic]) - CO(2, p[i0]);
        if (b1 > m.cyc / 2.0) b1 = b1 - m.cyc;
        if (b1 < -m.cyc / 2.0) b1 = b1 + m.cyc;
        if (c1 > m.cyc / 2.0) c1 = c1 - m.cyc;
        if (c1 < -m.cyc / 2.0) c1 = c1 + m.cyc;
        double cr = c1 * c1 + c2 * c2;
        double bx = (b1 * c1 + b2 * c2) / cr;
        double by = (-b1 * c2 + b2 * c1) / cr;
        double xx = (x1 * c1 + x2 * c2) / cr;
        double xy = (-x1 * c2 + x2 * c1) / cr;
        double ab = atan2(by, bx), ax = atan2(xy, xx);
        bool hit = ((ab > 0.0) && (ax > 0.0) && (ax < ab)) || ((ab < 0.0) && (ax < 0.0) && (ax > ab)) ||
                   ((ab == ax) || (ax == 0.0));
        if (hit) m.updn[2 * (d - 1) + side] = e;   // no exit in the reference: the last matching element wins
      }
    }
  }
}


// =====================================================================================================================
// Partition layer (npes > 1).  The node -> rank map comes from the reference's own partition files
// (dist_<npes>/my_list*.out: owned nodes), from a coarser merge of a finer one (ranks r/k of dist_<k*npes>), or from a
// recursive coordinate bisection when no file fits.  Everything else follows the reference's rules:
//   communication_nodn  src/gen_comm.F90:12-215     halo nodes = foreign nodes of the elements around owned nodes
//   communication_elemn :218-515                    my elements = any node owned; small halo = edge neighbours without
//                                                   owned node; full halo (+eXDim) = node-patch neighbours
//   mymesh              :517-641                    my edges = any node owned; external edges of my elements
//   com_global2local    src/oce_local.F90:11-117    lists -> local indices
// Local arrays are extractions of the global ones (owned lists are in increasing global order and halo values are the
// owners' values, so owned results do not depend on the partition); connectivity is translated to local indices, 0 = not
// local (load_edges src/oce_mesh.F90:1419-1642, find_neighbors :1650-1800).
// =====================================================================================================================
bool read_ints(const std::string &fn, ivec &v) {
  std::vector<std::string> tok;
  if (!read_all_tokens(fn, tok)) return false;
  v.resize(tok.size());
  for (size_t i = 0; i < tok.size(); i++) v[i] = atoi(tok[i].c_str());
  return true;
}

bool owner_from_files(Mesh &m, const std::string &dir, int np, ivec &own) {
  own.assign(m.N2, -1);
  for (int r = 0; r < np; r++) {
    char fn[64]; snprintf(fn, sizeof(fn), "/dist_%d/my_list%05d.out", np, r);
    ivec v;
    if (!read_ints(dir + fn, v) || v.size() < 3) return false;
    int myN = v[1];
    if ((int)v.size() < 3 + myN) return false;
    for (int k = 0; k < myN; k++) { int g = v[3 + k]; if (g < 1 || g > m.N2) return false; own[g - 1] = r; }
  }
  for (int n = 0; n < m.N2; n++) if (own[n] < 0) return false;
  return true;
}

void rcb(const Mesh &m, ivec &idx, int lo, int hi, int r0, int nr, ivec &own) {        // recursive coordinate bisection
  if (nr == 1) { for (int i = lo; i < hi; i++) own[idx[i]] = r0; return; }
  double mn[2] = {1e300, 1e300}, mx[2] = {-1e300, -1e300};
  for (int i = lo; i < hi; i++) for (int k = 0; k < 2; k++) { double c = m.coord[2 * idx[i] + k]; mn[k] = std::min(mn[k], c); mx[k] = std::max(mx[k], c); }
  int ax = (mx[0] - mn[0]) * cos(0.5 * (mn[1] + mx[1])) > (mx[1] - mn[1]) ? 0 : 1;
  int nl = nr / 2, cut = lo + (int)((long long)(hi - lo) * nl / nr);
  std::nth_element(idx.begin() + lo, idx.begin() + cut, idx.begin() + hi, [&](int a, int b) {
    double ca = m.coord[2 * a + ax], cb = m.coord[2 * b + ax];
    return ca < cb || (ca == cb && a < b);
  });
  rcb(m, idx, lo, cut, r0, nl, own);
  rcb(m, idx, cut, hi, r0 + nl, nr - nl, own);
}

bool make_owner(Mesh &m, const std::string &dir, int np) {
  if (owner_from_files(m, dir, np, m.owner)) return true;
  for (int k = 2; k <= 64; k++) {                              // merge a finer reference partition: rank = r / k
    ivec fine;
    if (owner_from_files(m, dir, np * k, fine)) { m.owner.resize(m.N2); for (int n = 0; n < m.N2; n++) m.owner[n] = fine[n] / k; return true; }
  }
  ivec idx(m.N2);
  for (int n = 0; n < m.N2; n++) idx[n] = n;
  m.owner.assign(m.N2, 0);
  rcb(m, idx, 0, m.N2, 0, np, m.owner);
  return true;
}

void build_com(int np, const ivec &recv_from, int nglob, const std::vector<std::vector<int>> &send_to, const ivec &mylist, Mesh::Com &c) {
  c = Mesh::Com();
  c.rptr.push_back(1); c.sptr.push_back(1);
  for (int r = 0; r < np; r++) {
    int nr = 0, ns = 0;
    for (int g = 0; g < nglob; g++) if (recv_from[g] == r) nr++;
    for (size_t l = 0; l < mylist.size(); l++) if (std::find(send_to[l].begin(), send_to[l].end(), r) != send_to[l].end()) ns++;
    if (nr) { c.rPE.push_back(r); c.rptr.push_back(c.rptr.back() + nr); for (int g = 0; g < nglob; g++) if (recv_from[g] == r) c.rlist.push_back(g + 1); }
    if (ns) {
      c.sPE.push_back(r); c.sptr.push_back(c.sptr.back() + ns);
      for (size_t l = 0; l < mylist.size(); l++) if (std::find(send_to[l].begin(), send_to[l].end(), r) != send_to[l].end()) c.slist.push_back(mylist[l]);
    }
  }
}

void partition(Mesh &m, int np, int me) {
  const ivec &own = m.owner;
  auto add = [](std::vector<int> &v, int r) { if (std::find(v.begin(), v.end(), r) == v.end()) v.push_back(r); };
  // ---- nodes (communication_nodn)
  m.list_n.clear();
  for (int n = 1; n <= m.N2; n++) if (own[n - 1] == me) m.list_n.push_back(n);
  m.myN = (int)m.list_n.size();
  {
    ivec recv(m.N2, -1);
    std::vector<std::vector<int>> send(m.myN);
    for (int l = 0; l < m.myN; l++) {
      int n = m.list_n[l];
      for (int i = 1; i <= m.nie_num[n - 1]; i++) {
        int el = NIE(i, n);
        for (int j = 1; j <= 3; j++) {
          int nod = EN(j, el);
          if (own[nod - 1] != me) { recv[nod - 1] = own[nod - 1]; add(send[l], own[nod - 1]); }
        }
      }
    }
    build_com(np, recv, m.N2, send, m.list_n, m.cn);
    m.eN = (int)m.cn.rlist.size();
  }
  // ---- elements (communication_elemn)
  ivec mye;
  for (int el = 1; el <= m.E2; el++) if (own[EN(1, el) - 1] == me || own[EN(2, el) - 1] == me || own[EN(3, el) - 1] == me) mye.push_back(el);
  m.myE = (int)mye.size();
  {
    ivec recv(m.E2, -1);
    std::vector<std::vector<int>> send(m.myE);
    auto none_mine = [&](int e) { return own[EN(1, e) - 1] != me && own[EN(2, e) - 1] != me && own[EN(3, e) - 1] != me; };
    auto any_foreign = [&](int e) { return own[EN(1, e) - 1] != me || own[EN(2, e) - 1] != me || own[EN(3, e) - 1] != me; };
    auto visit = [&](int l, int el, int elem) {
      if (none_mine(elem) && recv[elem - 1] == -1) recv[elem - 1] = own[EN(1, elem) - 1];     // first node's rank is the "main" owner
      if (own[EN(1, el) - 1] == me && any_foreign(elem))
        for (int i = 1; i <= 3; i++) {
          int ep = own[EN(i, elem) - 1];
          if (own[EN(1, el) - 1] == ep || own[EN(2, el) - 1] == ep || own[EN(3, el) - 1] == ep) continue;
          add(send[l], ep);
        }
    };
    for (int l = 0; l < m.myE; l++)
      for (int n = 1; n <= 3; n++) { int elem = ENB(n, mye[l]); if (elem >= 1) visit(l, mye[l], elem); }
    build_com(np, recv, m.E2, send, mye, m.ce);
    m.eE = (int)m.ce.rlist.size();
    for (int l = 0; l < m.myE; l++)
      for (int n = 1; n <= 3; n++) {
        int nod = EN(n, mye[l]);
        for (int j = 1; j <= m.nie_num[nod - 1]; j++) visit(l, mye[l], NIE(j, nod));
      }
    build_com(np, recv, m.E2, send, mye, m.cf);
  }
  m.list_e = mye;
  for (int g : m.ce.rlist) m.list_e.push_back(g);
  m.eX = 0;
  for (int g : m.cf.rlist) if (std::find(m.ce.rlist.begin(), m.ce.rlist.end(), g) == m.ce.rlist.end()) { m.list_e.push_back(g); m.eX++; }
  // ---- edges (mymesh)
  m.list_d.clear();
  for (int d = 1; d <= m.D2; d++) if (own[ED(1, d) - 1] == me || own[ED(2, d) - 1] == me) m.list_d.push_back(d);
  m.myD = (int)m.list_d.size();
  {
    std::vector<char> have(m.D2, 0);
    for (int l = 0; l < m.myE; l++)
      for (int q = 1; q <= 3; q++) {
        int e = EE(q, mye[l]);
        if (own[ED(1, e) - 1] != me && own[ED(2, e) - 1] != me && !have[e - 1]) { have[e - 1] = 1; m.list_d.push_back(e); }
      }
  }
  m.eD = (int)m.list_d.size() - m.myD;
  for (int g : m.cn.rlist) m.list_n.push_back(g);
  // ---- com_global2local
  ivec gn(m.N2 + 1, 0), ge(m.E2 + 1, 0);
  for (int l = 0; l < m.myN; l++) gn[m.list_n[l]] = l + 1;
  for (int k = 0; k < m.eN; k++) m.cn.rlist[k] = m.myN + k + 1;
  for (auto &v : m.cn.slist) v = gn[v];
  for (int l = 0; l < m.myE; l++) ge[mye[l]] = l + 1;
  for (int k = 0; k < m.eE; k++) { ge[m.ce.rlist[k]] = m.myE + k + 1; m.ce.rlist[k] = m.myE + k + 1; }
  for (auto &v : m.ce.slist) v = ge[v];
  { int x = 0; for (auto &v : m.cf.rlist) { if (ge[v] > 0) v = ge[v]; else { x++; v = m.myE + m.eE + x; } } }
  for (auto &v : m.cf.slist) v = ge[v];
}

template <class T> std::vector<T> gather(const std::vector<T> &src, int width, const ivec &list, int count) {
  std::vector<T> out((size_t)width * count);
  for (int l = 0; l < count; l++) memcpy(&out[(size_t)width * l], &src[(size_t)width * (list[l] - 1)], sizeof(T) * width);
  return out;
}

void make_local(Mesh &m, int np, int me) {
  const int Nl = m.myN + m.eN, El = m.myE + m.eE, EX = El + m.eX, Dl = m.myD + m.eD, nl = m.nl, n1 = nl - 1;
  ivec gn(m.N2 + 1, 0), ge(m.E2 + 1, 0), gd(m.D2 + 1, 0);
  for (int l = 0; l < Nl; l++) gn[m.list_n[l]] = l + 1;
  for (int l = 0; l < EX; l++) ge[m.list_e[l]] = l + 1;
  for (int l = 0; l < Dl; l++) gd[m.list_d[l]] = l + 1;
  m.li.reserve(64); m.ld.reserve(96);
  auto KI = [&](ivec v) -> const int * { m.li.push_back(std::move(v)); return m.li.back().data(); };
  auto KD = [&](dvec v) -> const double * { m.ld.push_back(std::move(v)); return m.ld.back().data(); };
  auto tr = [&](ivec v, const ivec &map) { for (auto &x : v) x = (x > 0) ? map[x] : 0; return v; };
  fesom_mesh_desc &d = m.desc;
  d.myDim_nod2D = m.myN; d.eDim_nod2D = m.eN; d.myDim_elem2D = m.myE; d.eDim_elem2D = m.eE; d.eXDim_elem2D = m.eX;
  d.myDim_edge2D = m.myD; d.eDim_edge2D = m.eD;
  d.myList_nod2D = m.list_n.data(); d.myList_elem2D = m.list_e.data(); d.myList_edge2D = m.list_d.data();
  d.coord_nod2D = KD(gather(m.coord, 2, m.list_n, Nl)); d.geo_coord_nod2D = KD(gather(m.geo, 2, m.list_n, Nl));
  d.elem2D_nodes = KI(tr(gather(m.elem_nodes, 3, m.list_e, EX), gn));
  d.edges = KI(tr(gather(m.edges, 2, m.list_d, Dl), gn));
  d.edge_tri = KI(tr(gather(m.edge_tri, 2, m.list_d, Dl), ge));
  d.elem_edges = KI(tr(gather(m.elem_edges, 3, m.list_e, m.myE), gd));
  d.elem_neighbors = KI(tr(gather(m.elem_nb, 3, m.list_e, m.myE), ge));
  d.nod_in_elem2D = KI(tr(gather(m.nie, m.maxk, m.list_n, Nl), ge));
  d.nod_in_elem2D_num = KI(gather(m.nie_num, 1, m.list_n, Nl));
  d.nlevels = KI(gather(m.nlev, 1, m.list_e, EX)); d.ulevels = KI(gather(m.ulev, 1, m.list_e, EX));
  d.nlevels_nod2D = KI(gather(m.nlev_n, 1, m.list_n, Nl)); d.ulevels_nod2D = KI(gather(m.ulev_n, 1, m.list_n, Nl));
  d.nlevels_nod2D_min = KI(gather(m.nlev_n_min, 1, m.list_n, Nl)); d.ulevels_nod2D_max = KI(gather(m.ulev_n_max, 1, m.list_n, Nl));
  d.depth = KD(gather(m.depth, 1, m.list_n, Nl));
  d.elem_area = KD(gather(m.elem_area, 1, m.list_e, EX));
  d.area = KD(gather(m.area, nl, m.list_n, Nl)); d.area_inv = KD(gather(m.area_inv, nl, m.list_n, Nl));
  d.areasvol = KD(gather(m.areasvol, nl, m.list_n, Nl)); d.areasvol_inv = KD(gather(m.areasvol_inv, nl, m.list_n, Nl));
  d.mesh_resolution = KD(gather(m.resol, 1, m.list_n, Nl));
  d.gradient_sca = KD(gather(m.grad_sca, 6, m.list_e, m.myE)); d.gradient_vec = KD(gather(m.grad_vec, 6, m.list_e, m.myE));
  d.edge_dxdy = KD(gather(m.edge_dxdy, 2, m.list_d, Dl)); d.edge_cross_dxdy = KD(gather(m.edge_cross, 4, m.list_d, Dl));
  d.elem_cos = KD(gather(m.elem_cos, 1, m.list_e, EX));
  {   // the quirk of oce_mesh.F90:2183 on a partition: every rank ends up with tan(latitude of ITS last owned element)/r_earth in all its
      // owned entries (halo entries take their owner's value in the reference; nothing on the hot path reads them: the rank's own value here)
    dvec mf(EX, tan(m.cen_y[m.list_e[m.myE - 1] - 1]) / R_EARTH);
    d.metric_factor = KD(mf);
  }
  d.coriolis = KD(gather(m.cori, 1, m.list_e, m.myE)); d.coriolis_node = KD(gather(m.cori_n, 1, m.list_n, Nl));
  d.edge_up_dn_tri = KI(tr(gather(m.updn, 2, m.list_d, m.myD), ge));
  d.zbar_n_bot = KD(gather(m.zbar_n_bot, 1, m.list_n, Nl)); d.zbar_n_srf = KD(gather(m.zbar_n_srf, 1, m.list_n, Nl));
  d.bottom_node_thickness = KD(gather(m.bot_n_th, 1, m.list_n, Nl));
  d.zbar_e_bot = KD(gather(m.zbar_e_bot, 1, m.list_e, El)); d.zbar_e_srf = KD(gather(m.zbar_e_srf, 1, m.list_e, El));
  d.bottom_elem_thickness = KD(gather(m.bot_e_th, 1, m.list_e, m.myE));
  // SSH operator rows of the owned nodes: columns in local numbering and in the PE-contiguous global numbering of the
  // reference (oce_ale.F90:1297-1344: rank r owns positions part(r) .. part(r+1)-1 in owned order)
  {
    ivec cnt(np + 1, 0), pos(m.N2 + 1, 0), nnz_before(np + 1, 0);
    for (int n = 1; n <= m.N2; n++) { int r = m.owner[n - 1]; pos[n] = ++cnt[r + 1]; nnz_before[r + 1] += m.rowptr[n] - m.rowptr[n - 1]; }
    for (int r = 1; r <= np; r++) { cnt[r] += cnt[r - 1]; nnz_before[r] += nnz_before[r - 1]; }
    for (int n = 1; n <= m.N2; n++) pos[n] += cnt[m.owner[n - 1]];
    ivec rp(m.myN + 1), cg, cl; dvec va;
    rp[0] = nnz_before[me] + 1;
    for (int l = 0; l < m.myN; l++) {
      int g = m.list_n[l];
      for (int q = m.rowptr[g - 1]; q < m.rowptr[g]; q++) { int c = m.colind[q - 1]; cg.push_back(pos[c]); cl.push_back(gn[c]); va.push_back(m.values[q - 1]); }
      rp[l + 1] = rp[l] + (m.rowptr[g] - m.rowptr[g - 1]);
    }
    d.ssh_nza = (int)va.size();
    d.ssh_rowptr = KI(rp); d.ssh_colind = KI(cg); d.ssh_colind_loc = KI(cl); d.ssh_values = KD(va);
    m.values_state = m.ld.back();
  }
  // initial ALE state (sizes as fesom_state_desc)
  m.hnode = gather(m.hnode, n1, m.list_n, Nl); m.hnode_new = gather(m.hnode_new, n1, m.list_n, Nl);
  m.helem = gather(m.helem, n1, m.list_e, m.myE);
  m.zbar3 = gather(m.zbar3, nl, m.list_n, Nl); m.Z3 = gather(m.Z3, n1, m.list_n, Nl);
  m.eta = gather(m.eta, 1, m.list_n, Nl); m.d_eta = gather(m.d_eta, 1, m.list_n, Nl); m.ssh_rhs = gather(m.ssh_rhs, 1, m.list_n, Nl);
  m.ssh_rhs_old = gather(m.ssh_rhs_old, 1, m.list_n, Nl); m.hbar = gather(m.hbar, 1, m.list_n, Nl); m.hbar_old = gather(m.hbar_old, 1, m.list_n, Nl);
  m.dhe = gather(m.dhe, 1, m.list_e, m.myE);
  // communication descriptors
  m.part.npes = np; m.part.mype = me;
  auto setc = [&](fesom_com_desc &c, const Mesh::Com &s) {
    c.rPEnum = (int)s.rPE.size(); c.sPEnum = (int)s.sPE.size();
    c.rPE = s.rPE.data(); c.rptr = s.rptr.data(); c.rlist = s.rlist.data(); c.sPE = s.sPE.data(); c.sptr = s.sptr.data(); c.slist = s.slist.data();
  };
  setc(m.part.com_nod2D, m.cn); setc(m.part.com_elem2D, m.ce); setc(m.part.com_elem2D_full, m.cf);
  m.local = true;
}

void fill_desc(Mesh &m) {
  fesom_mesh_desc &d = m.desc;
  memset(&d, 0, sizeof(d));
  d.nod2D = m.N2; d.elem2D = m.E2; d.edge2D = m.D2; d.edge2D_in = m.D2in; d.nl = m.nl;
  d.myDim_nod2D = m.N2; d.eDim_nod2D = 0; d.myDim_elem2D = m.E2; d.eDim_elem2D = 0; d.eXDim_elem2D = 0;
  d.myDim_edge2D = m.D2; d.eDim_edge2D = 0; d.max_nod_in_elem = m.maxk; d.ssh_nza = m.nza;
  m.list_n.resize(m.N2); m.list_e.resize(m.E2); m.list_d.resize(m.D2);
  for (int i = 0; i < m.N2; i++) m.list_n[i] = i + 1;
  for (int i = 0; i < m.E2; i++) m.list_e[i] = i + 1;
  for (int i = 0; i < m.D2; i++) m.list_d[i] = i + 1;
  d.myList_nod2D = m.list_n.data(); d.myList_elem2D = m.list_e.data(); d.myList_edge2D = m.list_d.data();
  d.coord_nod2D = m.coord.data(); d.geo_coord_nod2D = m.geo.data();
  d.elem2D_nodes = m.elem_nodes.data(); d.edges = m.edges.data(); d.edge_tri = m.edge_tri.data();
  d.elem_edges = m.elem_edges.data(); d.elem_neighbors = m.elem_nb.data();
  d.nod_in_elem2D = m.nie.data(); d.nod_in_elem2D_num = m.nie_num.data();
  d.nlevels = m.nlev.data(); d.ulevels = m.ulev.data(); d.nlevels_nod2D = m.nlev_n.data(); d.ulevels_nod2D = m.ulev_n.data();
  d.nlevels_nod2D_min = m.nlev_n_min.data(); d.ulevels_nod2D_max = m.ulev_n_max.data();
  d.zbar = m.zbar.data(); d.Z = m.Z.data(); d.depth = m.depth.data();
  d.elem_area = m.elem_area.data(); d.area = m.area.data(); d.area_inv = m.area_inv.data();
  d.areasvol = m.areasvol.data(); d.areasvol_inv = m.areasvol_inv.data(); d.mesh_resolution = m.resol.data();
  d.gradient_sca = m.grad_sca.data(); d.gradient_vec = m.grad_vec.data(); d.edge_dxdy = m.edge_dxdy.data();
  d.edge_cross_dxdy = m.edge_cross.data(); d.elem_cos = m.elem_cos.data(); d.metric_factor = m.metric.data();
  d.coriolis = m.cori.data(); d.coriolis_node = m.cori_n.data();
  d.ssh_rowptr = m.rowptr.data(); d.ssh_colind = m.colind.data(); d.ssh_colind_loc = m.colind_loc.data();
  d.ssh_values = m.values.data(); d.edge_up_dn_tri = m.updn.data();
  d.zbar_n_bot = m.zbar_n_bot.data(); d.zbar_n_srf = m.zbar_n_srf.data(); d.bottom_node_thickness = m.bot_n_th.data();
  d.zbar_e_bot = m.zbar_e_bot.data(); d.zbar_e_srf = m.zbar_e_srf.data(); d.bottom_elem_thickness = m.bot_e_th.data();
  memset(&m.part, 0, sizeof(m.part));
  m.part.npes = 1; m.part.mype = 0;
  m.com_dummy.assign(2, 1);
  fesom_com_desc *cs[3] = {&m.part.com_nod2D, &m.part.com_elem2D, &m.part.com_elem2D_full};
  for (auto c : cs) { c->rptr = m.com_dummy.data(); c->sptr = m.com_dummy.data(); }
}
}  // namespace

extern "C" {

void *fesom_mesh_load(const char *meshdir, const fesom_mesh_opts *opts) {
  if (opts->npes < 1 || opts->mype < 0 || opts->mype >= opts->npes) { fprintf(stderr, "fesom_mesh_load: bad npes/mype\n"); return nullptr; }
  if (opts->which_ale < 0 || opts->which_ale > 2) { fprintf(stderr, "fesom_mesh_load: which_ale must be 0 (linfs), 1 (zlevel) or 2 (zstar)\n"); return nullptr; }
  Mesh *mp = new Mesh();
  Mesh &m = *mp;
  m.o = *opts;
  m.cyc = opts->cyclic_length_deg * RAD;
  set_rot(m);
  if (!load_files(m, meshdir)) { fprintf(stderr, "fesom_mesh_load: cannot read mesh files in %s\n", meshdir); delete mp; return nullptr; }
  test_tri(m);
  topology(m);
  areas(m);
  auxiliary(m);
  ale_init(m);
  stiff_mat(m);
  updn_tri(m);
  fill_desc(m);
  if (opts->npes > 1) {
    make_owner(m, meshdir, opts->npes);
    partition(m, opts->npes, opts->mype);
    make_local(m, opts->npes, opts->mype);
  }
  return mp;
}

const fesom_mesh_desc *fesom_mesh_get_desc(void *h) { return &((Mesh *)h)->desc; }
const fesom_part_desc *fesom_mesh_get_part(void *h) { return &((Mesh *)h)->part; }

const fesom_state_desc *fesom_mesh_get_initial_state(void *h, int ntr) {
  Mesh &m = *(Mesh *)h;
  size_t nl = m.nl, N = m.local ? m.myN + m.eN : m.N2, E = m.local ? m.myE + m.eE : m.E2;
  m.tr.assign((nl - 1) * N * ntr, 0.0); m.tr_old.assign((nl - 1) * N * ntr, 0.0);
  m.UV.assign(2 * (nl - 1) * E, 0.0); m.UVab.assign(2 * (nl - 1) * E, 0.0);
  m.W.assign(nl * N, 0.0); m.We.assign(nl * N, 0.0); m.Wi.assign(nl * N, 0.0);
  if (!m.local) m.values_state = m.values;
  fesom_state_desc &s = m.st;
  s.tr_arr = m.tr.data(); s.tr_arr_old = m.tr_old.data(); s.UV = m.UV.data(); s.UV_rhsAB = m.UVab.data();
  s.eta_n = m.eta.data(); s.d_eta = m.d_eta.data(); s.ssh_rhs = m.ssh_rhs.data(); s.ssh_rhs_old = m.ssh_rhs_old.data();
  s.hbar = m.hbar.data(); s.hbar_old = m.hbar_old.data(); s.dhe = m.dhe.data();
  s.hnode = m.hnode.data(); s.hnode_new = m.hnode_new.data(); s.helem = m.helem.data();
  s.zbar_3d_n = m.zbar3.data(); s.Z_3d_n = m.Z3.data();
  s.Wvel = m.W.data(); s.Wvel_e = m.We.data(); s.Wvel_i = m.Wi.data(); s.ssh_values = m.values_state.data();
  return &s;
}

void fesom_mesh_free(void *h) { delete (Mesh *)h; }
}
