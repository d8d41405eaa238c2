// C-ABI of libfesom_gpu.so (include/fesom_gpu.h): device state manager, step orchestration, hipGraph replay.
// Host side mirrors the call surface of oce_timestep_ale (src/oce_ale.F90:2521-2799): the order of kernel
// groups in enqueue_step() is the order of subroutine calls there.
#include "dev.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <chrono>
#include <map>
#include <algorithm>
#include <dlfcn.h>
#include <rccl/rccl.h>

int launch_named_dyn(const DM &m, hipStream_t s, const char *name, int arg, int first_step);
int launch_named_tra(const DM &m, hipStream_t s, const char *name, int arg);
void solver_prepare();
extern "C" int fesom_xinv_build(int n, const int *rp, const int *ci, const double *vals, const double *scale, int ld, float *out, int *bandwidth);
extern "C" void fesom_xinv_sparsify(int n, int ld, const float *M, double tau, int *rowptr, unsigned short *cols, float *vals);
#define XINV_DROP 1.0e-4      /* entries of the inverse below this fraction of their row's largest are dropped */
#include "ras_host.h"
int launch_named_ras(const DM &m, hipStream_t s, const char *name);
void tile_prepare_tra();
void tile_prepare_dyn();
void launch_init_density_ref(const DM &m, hipStream_t s);

namespace {
struct Field { void *p; size_t count; int slabs; };   // slabs>1: one slab of `count` values per tracer
struct Ctx {
  bool ready = false;
  DM m;
  hipStream_t stream = nullptr;
  bool ext_stream = false;                              // stream handed in by the host (fesom_gpu_set_stream): not ours to destroy
  hipStream_t side[3] = {nullptr, nullptr, nullptr};   // forked branches of the step DAG
  int cur_tr = 0;
  int part_iters = -1;
  double *frc_dev = nullptr, *frc_pin[2] = {nullptr, nullptr};     // surface forcing: device block + pinned staging
  hipEvent_t frc_ev[2] = {nullptr, nullptr};
  size_t frc_count = 0;
  int frc_cur = 0;
  double *mon_col = nullptr, *mon_out = nullptr;       // step monitor scratch (fesom_gpu_step_info)
  bool serial = false;
  std::map<std::string, Field> fields;
  std::vector<void *> allocs;
  int first_step = 1;
  hipGraphExec_t graph[2] = {nullptr, nullptr};
  bool use_graph = true;
  std::string err;
  // partition + halo exchange (npes > 1)
  int npes = 1, mype = 0;
  struct Halo { std::vector<int> rPE, rptr, sPE, sptr, slist_h; const int *rlist = nullptr, *slist = nullptr, *slist_q = nullptr; const int *rptr_d = nullptr, *sptr_d = nullptr; int nrecv = 0, nsend = 0; } halo[3];   // slist_q: node send list as positions of the solver's patch order
  double *hsend = nullptr, *hrecv = nullptr; size_t hcap = 0;        // channel 0: exchanges on the step's stream
  double *hsend1 = nullptr, *hrecv1 = nullptr; size_t hcap1 = 0;     // channel 1: the exchange in flight on the communication stream
  hipStream_t cstream = nullptr; hipEvent_t ev_prod = nullptr, ev_done = nullptr;
  long long n_async = 0;
  bool x_pending = false;                             // an exchange is in flight on the communication stream
  bool toy_znum_global = false;                       // partitioned Soufflet channel: the per-bin element counts have been summed over the ranks
  // communication statistics of the partitioned step (fesom_gpu_comm_stats)
  long long n_exch = 0, n_allred = 0, n_parts = 0;
  // interior / boundary split of the node-column kernels behind an exchange (partitioned runs): owned nodes whose edge neighbours are all owned,
  // owned nodes with a halo neighbour, and the latter plus the halo nodes themselves
  struct ColList { const int *d = nullptr; int n = 0; } sub_int, sub_cb, sub_cbh;
  bool dref_done = false;                             // the reference density profile has been formed (first state upload)
  bool solver_only = false;                           // the context holds the distributed SSH solver alone (fesom_gpu_psolver_init_dist): no ocean step
  bool precond_agreed = false;                        // partitioned runs: all ranks have settled on one SSH preconditioner
  int generation = 0;                                 // counts fesom_gpu_init calls (cached plans of the partitioned step belong to one)
  bool comm_timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> comm_ev;
} G;

// ---- built-in transport: RCCL point-to-point over xGMI, issued by the library on its own stream --------------------------
// The reference posts MPI_Isend/Irecv per neighbour and waits (src/gen_halo_exchange.F90:129-164, 317-363); here every exchange
// is ONE RCCL group of ncclSend/ncclRecv (one pair per neighbour of the com list) on the stream that runs the pack and unpack
// kernels, so the step needs no host callback and no host synchronisation per exchange.  librccl is loaded on first use
// (dlopen by soname: inside a PyTorch process that is the librccl torch has already loaded; FESOM_GPU_RCCL_LIB overrides, which
// the tests use to put a shared-memory stand-in between two ranks that share one GPU).
struct RcclApi {
  void *h = nullptr;
  ncclComm_t comm = nullptr;
  int nranks = 0, rank = -1;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
} R;
int rccl_load() {
  if (R.h) return 0;
  const char *ov = getenv("FESOM_GPU_RCCL_LIB");
  const char *cand[] = {ov, "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
  for (const char *c : cand) {
    if (!c || !*c) continue;
    R.h = dlopen(c, RTLD_NOW | RTLD_LOCAL);
    if (R.h) break;
    if (c == ov) { G.err = std::string("comm: cannot load FESOM_GPU_RCCL_LIB=") + ov + ": " + dlerror(); return 1; }
  }
  if (!R.h) { G.err = std::string("comm: cannot load librccl: ") + dlerror(); return 1; }
  bool ok = true;
#define SYM(f, n) do { *(void **)(&R.f) = dlsym(R.h, n); ok = ok && R.f; } while (0)
  SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
  SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv");
  SYM(AllReduce, "ncclAllReduce"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  if (!ok) { G.err = "comm: librccl lacks a required symbol"; dlclose(R.h); R.h = nullptr; return 1; }
  return 0;
}
#define NCCLCHK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { G.err = std::string(#x) + ": " + R.GetErrorString(r_); fprintf(stderr, "fesom_gpu: %s\n", G.err.c_str()); return 1; } } while (0)


#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { G.err = std::string(#x) + ": " + hipGetErrorString(e_); fprintf(stderr, "fesom_gpu: %s\n", G.err.c_str()); return 1; } } while (0)

bool g_prog_ready = false;                         // the program of the partitioned step (build_program) belongs to this context
bool g_alloc_failed = false;                       // any device allocation / upload of fesom_gpu_init that failed (checked at its end)
template <class T> T *dev_alloc(size_t n) {
  void *p = nullptr;
  if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) { g_alloc_failed = true; return nullptr; }
  hipMemset(p, 0, std::max<size_t>(n, 1) * sizeof(T));
  G.allocs.push_back(p);
  return (T *)p;
}
template <class T> const T *dev_upload(const std::vector<T> &h) {
  T *p = dev_alloc<T>(h.size());
  if (p && !h.empty() && hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) g_alloc_failed = true;
  return p;
}
const double *dev_upload_d(const double *h, size_t n) {
  double *p = dev_alloc<double>(n);
  if (p && n && hipMemcpy(p, h, n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) g_alloc_failed = true;
  return p;
}
// Explicit-inverse preconditioner (csrc/precond_host.cpp): built on the host from the operator the run starts with, uploaded once.
// The last matrix is kept per process and reused when the same operator comes again (tests and benches re-initialise often).
struct XinvCache { std::vector<int> rp, ci; std::vector<double> vals; std::vector<int> mp; std::vector<unsigned short> mc; std::vector<float> mv; } XC;
static double xinv_drop() { const char *e = getenv("FESOM_GPU_XINV_DROP"); const double v = e ? atof(e) : 0.0; return v > 0.0 ? v : XINV_DROP; }     // (experiments: tools/xinv_sweep.py)
int xinv_device(int n, const int *rp, const int *ci, const double *vals, const double *scale, DM &m, std::vector<void *> &owner) {
  const int nza = rp[n];
  const bool hit = !scale && (int)XC.rp.size() == n + 1 && (int)XC.vals.size() == nza && !XC.mp.empty() && !memcmp(XC.rp.data(), rp, sizeof(int) * (n + 1)) &&
                   !memcmp(XC.ci.data(), ci, sizeof(int) * nza) && !memcmp(XC.vals.data(), vals, sizeof(double) * nza);
  if (!hit) {
    const int ld = (n + 255) / 256 * 256;
    std::vector<float> M((size_t)n * ld, 0.0f);
    XC.mp.clear();
    if (fesom_xinv_build(n, rp, ci, vals, scale, ld, M.data(), nullptr)) return 1;
    XC.mp.assign(n + 1, 0);
    fesom_xinv_sparsify(n, ld, M.data(), xinv_drop(), XC.mp.data(), nullptr, nullptr);
    XC.mc.assign(XC.mp[n], 0); XC.mv.assign(XC.mp[n], 0.0f);
    fesom_xinv_sparsify(n, ld, M.data(), xinv_drop(), XC.mp.data(), XC.mc.data(), XC.mv.data());
    XC.rp.assign(rp, rp + n + 1); XC.ci.assign(ci, ci + nza); XC.vals.assign(vals, vals + nza);
    if (scale) XC.rp.clear();                              // (a partition's block: not cached)
  }
  void *dp = nullptr, *dc = nullptr, *dv = nullptr;
  if (hipMalloc(&dp, XC.mp.size() * sizeof(int)) != hipSuccess) return 1;
  owner.push_back(dp);
  if (hipMalloc(&dc, XC.mc.size() * sizeof(unsigned short) + 8) != hipSuccess) return 1;
  owner.push_back(dc);
  if (hipMalloc(&dv, XC.mv.size() * sizeof(float) + 8) != hipSuccess) return 1;
  owner.push_back(dv);
  if (hipMemcpy(dp, XC.mp.data(), XC.mp.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(dc, XC.mc.data(), XC.mc.size() * sizeof(unsigned short), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(dv, XC.mv.data(), XC.mv.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return 1;
  m.sv_mp = (const int *)dp; m.sv_mc = (const unsigned short *)dc; m.sv_minv = (const float *)dv;
  return 0;
}
// RAS-Chebyshev preconditioner (csrc/ras_host.h): plan built on the host from the operator the run starts with (frozen, like the
// reference's ILU factors), uploaded once.  `ld` = length of the solver vectors' owned part rounded up to 64 (ELL leading dimension).
static int env_int(const char *n, int dflt) { const char *e = getenv(n); return e && *e ? atoi(e) : dflt; }
int ras_device(int n, int ncols, const int *rp, const int *ci, const double *vals, int maxnnz, DM &m, std::vector<void *> &owner, std::vector<int> *inv_out = nullptr) {
  RasPlan pl;
  const char *ek = getenv("FESOM_GPU_RAS_KAPPA");
  if (fesom_ras_build(n, rp, ci, vals, nullptr, env_int("FESOM_GPU_RAS_PATCH", RAS_PATCH_MAX), env_int("FESOM_GPU_RAS_OVL", RAS_OVERLAP), env_int("FESOM_GPU_RAS_DEG", RAS_DEG),
                      ek && atof(ek) > 1.0 ? atof(ek) : RAS_KAPPA, pl)) return 1;
  auto up = [&](const void *h, size_t bytes) -> void * {
    void *d = nullptr;
    if (hipMalloc(&d, bytes ? bytes : 8) != hipSuccess) return nullptr;
    owner.push_back(d);
    if (bytes && hipMemcpy(d, h, bytes, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
  };
  std::vector<double> cheb(128, 0.0);
  cheb[0] = pl.inv_theta;
  for (int k = 1; k < pl.deg; k++) { cheb[1 + k] = pl.c1[k]; cheb[64 + k] = pl.c2[k]; }
  const int W = maxnnz <= 8 ? 8 : maxnnz <= 10 ? 10 : 16, NP = (n + 63) / 64 * 64;
  std::vector<int> colsq((size_t)W * NP, 0);                  // ELL pattern of the products with A_s, as positions (halo columns keep their index)
  for (int q = 0; q < NP; q++)
    for (int k = 0; k < W; k++) {
      int c = q < n ? q : 0;
      if (q < n) { const int i = pl.perm[q]; if (rp[i] + k < rp[i + 1]) { const int cc = ci[rp[i] + k]; c = cc < n ? pl.inv[cc] : cc; } }
      colsq[(size_t)k * NP + q] = c;
    }
  (void)ncols;
  m.rs_pinfo = (const int *)up(pl.pinfo.data(), pl.pinfo.size() * sizeof(int));
  m.rs_extq = (const int *)up(pl.extq.data(), pl.extq.size() * sizeof(int));
  m.rs_perm = (const int *)up(pl.perm.data(), pl.perm.size() * sizeof(int));
  m.rs_inv = (const int *)up(pl.inv.data(), pl.inv.size() * sizeof(int));
  m.rs_colsq = (const int *)up(colsq.data(), colsq.size() * sizeof(int));
  m.rs_lv = (const float *)up(pl.lv.data(), pl.lv.size() * sizeof(float));
  m.rs_lc = (const unsigned short *)up(pl.lc.data(), pl.lc.size() * sizeof(unsigned short));
  m.rs_dsc = (const double *)up(pl.dsc.data(), pl.dsc.size() * sizeof(double));
  m.rs_cheb = (const double *)up(cheb.data(), cheb.size() * sizeof(double));
  if (!m.rs_pinfo || !m.rs_extq || !m.rs_perm || !m.rs_inv || !m.rs_colsq || !m.rs_lv || !m.rs_lc || !m.rs_dsc || !m.rs_cheb) { m.rs_pinfo = nullptr; return 1; }
  m.rs_P = pl.P; m.rs_NS = pl.NS; m.rs_rpt = pl.rpt; m.rs_woff = pl.woff; m.rs_deg = pl.deg;
  if (inv_out) *inv_out = pl.inv;
  return 0;
}
double *field(const char *name, size_t n, int slabs = 1) {
  double *p = dev_alloc<double>(n * slabs);
  G.fields[name] = Field{p, n, slabs};
  return p;
}
// static ELL column pattern [k][NP] (uint16), padding entries point to the row itself (their B entry is 0)
// Row order of the one-workgroup solve: stable sort by the number of entries, descending (rows of equal width keep their order);
// `sorted` = false gives the identity.  perm[position] = row, inv[row] = position, wid[position/64] = widest row of the group.
void solver_row_order(const int *rp, int n, int maxnnz, bool sorted, std::vector<int> &perm, std::vector<int> &inv, std::vector<int> &wid) {
  const int NP = (n + 63) / 64 * 64, W = maxnnz <= 10 ? 10 : 16;
  perm.assign(NP, 0); inv.assign(NP, 0); wid.assign(NP / 64, W);
  int q = 0;
  if (sorted) for (int w = maxnnz; w >= 0; w--) for (int i = 0; i < n; i++) if (rp[i + 1] - rp[i] == w) perm[q++] = i;
  for (; q < n; q++) perm[q] = q;                      // (sorted == false)
  if (!sorted) for (int i = 0; i < n; i++) perm[i] = i;
  for (int i = n; i < NP; i++) perm[i] = i;
  for (int i = 0; i < NP; i++) inv[perm[i]] = i;
  if (sorted)
    for (int g = 0; g < NP / 64; g++) {
      int w = 0;
      for (int i = 64 * g; i < 64 * g + 64 && i < n; i++) w = std::max(w, rp[perm[i] + 1] - rp[perm[i]]);
      wid[g] = w;
    }
}
// static ELL column pattern in the solver's row order: entry k of the row at position q, column ids as positions
std::vector<unsigned short> ell_cols(const int *rp, const int *ci, int n, int maxnnz, const std::vector<int> &perm, const std::vector<int> &inv) {
  int W = maxnnz <= 10 ? 10 : 16, NP = (n + 63) / 64 * 64;
  std::vector<unsigned short> c((size_t)W * NP, 0);
  for (int q = 0; q < NP; q++)
    for (int k = 0; k < W; k++) {
      const int i = perm[q];
      unsigned short v = (unsigned short)(q < n ? q : 0);
      if (q < n && rp[i] + k < rp[i + 1]) v = (unsigned short)inv[ci[rp[i] + k]];
      c[(size_t)k * NP + q] = v;
    }
  return c;
}
std::vector<int> minus1(const int *a, size_t n) {
  std::vector<int> v(n);
  for (size_t i = 0; i < n; i++) v[i] = a[i] > 0 ? a[i] - 1 : -1;
  return v;
}

// `n` = step number (mstep of the reference); only the Soufflet toy hooks look at it (zonal means every 10th step)
void enqueue_step(hipStream_t s, int first_step, int n) {
  const DM &m = G.m;
  const bool toy = m.p.toy_soufflet != 0;
  if (toy && n % 10 == 0) launch_named_toy(m, s, "compute_zonal_mean");    // before_oce_step (oce_setup_step.F90:625-630)
  launch_dynamics_pre(m, s, first_step);     // compute_vel_nodes .. impl_vert_visc_ale
  launch_ssh_rhs(m, s);                      // update_stiff_mat_ale, compute_ssh_rhs_ale
  launch_solver(m, s);                       // solve_ssh_ale
  if (toy) launch_named_toy(m, s, "relax_zonal_vel");                      // oce_ale.F90:2696
  const bool gm = m.p.Fer_GM != 0;
  if (gm || m.p.Redi) launch_named_gm(m, s, "init_Redi_GM");               // before vert_vel_ale touches hnode_new (oce_ale.F90:2729-2739)
  if (gm) { launch_named_gm(m, s, "fer_solve_Gamma"); launch_named_gm(m, s, "fer_gamma2vel"); launch_named_gm(m, s, "fer_wvel"); }
  launch_dynamics_post(m, s);                // update_vel, compute_hbar_ale, eta_n, vert_vel_ale
  if (m.p.SPP) launch_named_tra(m, s, "k_spp", 0);                          // solve_tracers_ale :120-121
  if (gm) launch_named_gm(m, s, "bolus_add");                               // solve_tracers_ale :127-131
  launch_tracer(m, s, -1);                   // solve_tracers_ale, all tracers per launch
  if (gm) launch_named_gm(m, s, "bolus_remove");                            // :165-169
  if (toy) for (int tr = 0; tr < m.ntr; tr++) launch_named_toy(m, s, "relax_zonal_temp");   // once per tracer, oce_ale_tracer.F90:150
  else if (m.p.clim_relax > 1.0e-8) launch_named_tra(m, s, "relax_to_clim", 0);
  launch_thickness(m, s);                    // update_thickness_ale
}

// One step as a DAG over 4 streams (captured into one hipGraph).  pi is latency-bound (42+ dependent launches, each a
// few thousand wavefronts), so independent branches run concurrently: the four pre-solver branches of the momentum
// equation, the tracer preparation of T and S (hidden under the SSH solve) and the T and S advection chains.
struct Dag {
  std::vector<hipEvent_t> evs;
  size_t next = 0;
  void reset() { next = 0; }                         // eager replay: the same events are re-recorded every step
  hipEvent_t ev() {
    if (next == evs.size()) { hipEvent_t e; hipEventCreateWithFlags(&e, hipEventDisableTiming); evs.push_back(e); }
    return evs[next++];
  }
  void dep(hipStream_t to, hipStream_t from) { if (to == from) return; hipEvent_t e = ev(); hipEventRecord(e, from); hipStreamWaitEvent(to, e, 0); }     // (same stream: already ordered)
  ~Dag() { for (auto e : evs) hipEventDestroy(e); }
};
static std::string g_bad_launch;
static int g_exp_skip_side = -1, g_exp_step = 0;     // timing experiment (FESOM_GPU_EXP_SKIP_SIDE=<step>): from that step on only the critical chain is launched
int K(hipStream_t s, const char *k, int arg = 0, int fs = 0) {
  if (g_exp_skip_side >= 0 && g_exp_step >= g_exp_skip_side && s != G.stream) return 0;
  int rc = launch_named_dyn(G.m, s, k, arg, fs);
  if (rc < 0) rc = launch_named_tra(G.m, s, k, arg);
  if (rc < 0) rc = launch_named_kpp(G.m, s, k);
  if (rc != 0 && g_bad_launch.empty()) g_bad_launch = k;          // a phase of the step that no launcher knows: the step must not pass for done
  return rc;
}
void enqueue_step_dag(hipStream_t s0, int first_step, Dag &d, int n) {
  // The critical chain stays on s0 (momentum -> SSH solve -> W -> tracer advection/diffusion -> thickness); side streams
  // carry what only has to be READY by then.  All tracers go through the same launches (grid.y = tracer).
  const DM &m = G.m;
  static const int nside = getenv("FESOM_GPU_SIDE_STREAMS") ? atoi(getenv("FESOM_GPU_SIDE_STREAMS")) : 3;     // timing experiment: 1 or 2 = fewer side streams
  hipStream_t s1 = G.side[0], s2 = nside >= 2 ? G.side[1] : G.side[0], s3 = nside >= 3 ? G.side[2] : G.side[0];
  { static bool once = false; if (!once) { once = true; const char *e = getenv("FESOM_GPU_EXP_SKIP_SIDE"); if (e) g_exp_skip_side = atoi(e); } g_exp_step++; }
  const bool toy = m.p.toy_soufflet != 0;
  if (toy && n % 10 == 0) launch_named_toy(m, s0, "compute_zonal_mean");   // before_oce_step (oce_setup_step.F90:625-630)
  d.dep(s1, s0); d.dep(s2, s0); d.dep(s3, s0);
  // s1: pressure -> PGF -> velocity rhs ; SSH operator update + row scales ; later dhe
  K(s1, "k_pressure_bv");
  hipEvent_t ev_pb = d.ev(); hipEventRecord(ev_pb, s1);
  K(s1, "k_pgf");
  if (m.p.mom_adv == 3) K(s1, "k_vel_rhs_step", 0, first_step);      // k_vinv_ke, k_leith_vort, k_vinv_elem
  else {
    K(s2, "k_momadv_node");
    d.dep(s1, s2);
    K(s1, "k_vel_rhs", 0, first_step);
  }
  hipEvent_t ev_rhs = d.ev(); hipEventRecord(ev_rhs, s1);
  if (m.p.which_ale != 0) K(s1, "k_stiff_update");
  launch_row_scale(m, s1);
  hipEvent_t ev_op = d.ev(); hipEventRecord(ev_op, s1);
  // s3: viscosity stencil, then everything nobody waits for soon (sigma/slope, tracer preparation)
  if (m.p.visc_option <= 3) K(s3, "h_viscosity_leith");      // UV, Wvel, helem of the incoming state only
  if (m.p.visc_option != 1 && m.p.visc_option != 8) K(s3, "k_visc_elem");
  if (m.p.visc_option == 5) K(s3, "k_visc_node");
  hipEvent_t ev_visc = d.ev(); hipEventRecord(ev_visc, s3);
  if (s3 != s1) hipStreamWaitEvent(s3, ev_pb, 0);
  K(s3, "k_sigma_slope");
  const bool gm = m.p.Fer_GM != 0;
  hipEvent_t ev_gm = nullptr;
  const bool redi = m.p.Redi != 0;
  if (gm || redi) {    // bolus velocities / Ki: need bvfreq, sigma_xy, helem and the OLD hnode_new -> before vert_vel_ale on s0
    launch_named_gm(m, s3, "init_Redi_GM");
    if (gm) { launch_named_gm(m, s3, "fer_solve_Gamma"); launch_named_gm(m, s3, "fer_gamma2vel"); launch_named_gm(m, s3, "fer_wvel"); }
    ev_gm = d.ev(); hipEventRecord(ev_gm, s3);
  }
  const bool fuse_updn = m.use_tile != 0;             // CORE2-class meshes: k_flux_hor evaluates fill_up_dn_grad on the fly (kernels_tra.hip)
  K(s3, "k_tr_ab", 0); K(s3, "k_tr_grad_elem", 0); if (!fuse_updn) K(s3, "k_updn_grad", 0);
  if (m.p.with_diffusion && !redi) K(s3, "k_diff_flux", 0);        // with Redi it needs tr_z of the new thicknesses: see s1 below
  hipEvent_t ev_prep = d.ev(); hipEventRecord(ev_prep, s3);
  // s0: critical chain
  K(s0, "k_vel_nodes");
  hipStreamWaitEvent(s0, ev_pb, 0);
  if (m.p.use_momix && m.p.mix_scheme == 2) K(s0, "k_momix");    // mixing length of mo_convect (forcing + ice state only); "mixing_kpp" launches it itself
  if (m.p.mix_scheme == 2) K(s0, "k_pp");          // element (Av) and node (Kv) part in one launch
  if (m.p.mix_scheme == 1) K(s0, "mixing_kpp");       // k_kpp_col, 3 smoothing sweeps, k_kpp_final, k_kpp_elem
  if (s3 != s1) hipStreamWaitEvent(s0, ev_rhs, 0);      // (one side stream: ev_visc is recorded behind ev_rhs and ev_op on it)
  hipStreamWaitEvent(s0, ev_visc, 0);
  if (m.p.visc_option == 8) K(s0, "viscosity_filter");   // backscatter_coef + visc_filt_dbcksc + uke_update: needs the complete UV_rhs, bvfreq
  else if (m.p.visc_option != 5) K(s0, "k_visc_apply");  // second stage of the biharmonic filters (visc_option 6, 7): in place on UV_rhs
  K(s0, "k_impl_visc");                            // incl. the Thomas sweep
  K(s0, "k_edge_transport");
  if (s3 != s1) hipStreamWaitEvent(s0, ev_op, 0);
  launch_solver(m, s0, 1, 1);                      // set-up gathers ssh_rhs (k_ssh_rhs_node fused); row scales from s1
  if (toy) launch_named_toy(m, s0, "relax_zonal_vel");                     // oce_ale.F90:2696
  K(s0, "k_update_vel"); K(s0, "k_edge_transport1");
  if (ev_gm) hipStreamWaitEvent(s0, ev_gm, 0);
  K(s0, "k_vert_vel_hbar");                        // k_hbar_node fused
  if (gm) launch_named_gm(m, s0, "bolus_add");     // solve_tracers_ale :127-131 (k_vert_vel reads UV itself: the addition cannot ride in its launch)
  hipEvent_t ev_w = d.ev(); hipEventRecord(ev_w, s0);
  hipStreamWaitEvent(s1, ev_w, 0);
  K(s1, "k_dhe");
  if (redi) K(s1, "k_tr_z", 0);                    // vertical tracer gradient: only the Redi terms read it (the named routine init_tracers_AB always forms it)
  hipStreamWaitEvent(s0, ev_prep, 0);
  hipEvent_t ev_df = nullptr;
  if (redi && m.p.with_diffusion) { hipStreamWaitEvent(s1, ev_prep, 0); K(s1, "k_diff_flux", 0); ev_df = d.ev(); hipEventRecord(ev_df, s1); }
  K(s0, fuse_updn ? "k_flux_hor_fused" : "k_flux_hor", 0); K(s0, "k_fct_lo_node", 0); K(s0, "k_fct_node", 0);
  // (k_fct_edge_limit only materialises the limited flux field adv_flux_hor, which no kernel of the step reads -- k_tr_update limits on
  // the fly: it is part of the named routine adv_tracers_ale, not of the running step)
  if (ev_df) hipStreamWaitEvent(s0, ev_df, 0);
  K(s0, "k_tr_update", 0);                         // incl. the Thomas sweep
  if (m.p.smooth_bh_tra) { K(s0, "k_bh1", 0); K(s0, "k_bh2", 0); }      // diff_part_bh at the end of diff_tracers_ale
  if (toy) for (int tr = 0; tr < m.ntr; tr++) launch_named_toy(m, s0, "relax_zonal_temp");   // once per tracer
  else if (m.p.clim_relax > 1.0e-8) launch_named_tra(m, s0, "relax_to_clim", 0);
  // bolus_remove (:165-169) rides in the thickness launch, which reads neither UV nor the vertical velocities: one launch less on the chain (pi: 0.410 -> 0.407 ms
  // per step; FESOM_GPU_NO_BOLUS_FOLD=1 keeps k_bolus)
  static const bool no_fold = getenv("FESOM_GPU_NO_BOLUS_FOLD") && atoi(getenv("FESOM_GPU_NO_BOLUS_FOLD")) != 0;
  const bool fold_remove = gm && !no_fold && m.p.which_ale != 0;
  if (gm && !fold_remove) launch_named_gm(m, s0, "bolus_remove");          // :165-169
  d.dep(s0, s1); d.dep(s0, s2); d.dep(s0, s3);
  launch_thickness(m, s0, fold_remove);
}

int build_graph(int which) {
  hipGraph_t g;
  Dag d;                                            // events must outlive the capture
  HIPCHK(hipStreamBeginCapture(G.stream, hipStreamCaptureModeGlobal));
  if (G.serial || G.m.p.SPP) enqueue_step(G.stream, which, 1); else enqueue_step_dag(G.stream, which, d, 1);     // (SPP: see fesom_gpu_run_steps)
  HIPCHK(hipStreamEndCapture(G.stream, &g));
  HIPCHK(hipGraphInstantiate(&G.graph[which], g, nullptr, nullptr, 0));
  hipGraphDestroy(g);
  return 0;
}
}  // namespace

// Device of this process, shared by every context of the library (ocean core, sea ice, communicator): FESOM_GPU_DEVICE, else LOCAL_RANK
// (one rank per GPU), else 0.  A mis-set multi-rank launch must not pile every rank on device 0 and still report numbers; a context
// must not end up on another device than the one an earlier context of the process is bound to.
int fesom_internal_select_device(std::string &err) {
  static int bound = -1;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { err = "no HIP device: the MI355X path has no CPU fallback"; return 1; }
  const char *dv = getenv("FESOM_GPU_DEVICE");
  if (!dv) dv = getenv("LOCAL_RANK");
  const int dev = dv ? atoi(dv) : 0;
  if (dev < 0 || dev >= ndev) {
    err = std::string("device index ") + std::to_string(dev) + " (FESOM_GPU_DEVICE / LOCAL_RANK) but only " + std::to_string(ndev) + " HIP device(s) are visible";
    return 1;
  }
  if (bound >= 0 && bound != dev && (G.ready || R.comm)) {
    err = std::string("device index ") + std::to_string(dev) + " requested, but a context of this process is already bound to device " + std::to_string(bound);
    return 1;
  }
  if (hipSetDevice(dev) != hipSuccess) { err = "hipSetDevice failed"; return 1; }
  bound = dev;
  return 0;
}

int fesom_internal_rccl_exchange(int ns, const int *sPE, const int *sptr, int nr, const int *rPE, const int *rptr, double *sd, double *rd, int W, hipStream_t s) {
  if (!R.comm) { G.err = "built-in transport not initialised (fesom_gpu_comm_init)"; return 1; }
  NCCLCHK(R.GroupStart());
  for (int p = 0; p < ns; p++) {
    const size_t first = (size_t)(sptr[p] - 1), cnt = (size_t)(sptr[p + 1] - sptr[p]);
    if (cnt) NCCLCHK(R.Send(sd + first * W, cnt * W, ncclDouble, sPE[p], R.comm, s));
  }
  for (int p = 0; p < nr; p++) {
    const size_t first = (size_t)(rptr[p] - 1), cnt = (size_t)(rptr[p + 1] - rptr[p]);
    if (cnt) NCCLCHK(R.Recv(rd + first * W, cnt * W, ncclDouble, rPE[p], R.comm, s));
  }
  NCCLCHK(R.GroupEnd());
  return 0;
}

extern "C" {

const char *fesom_gpu_last_error(void) { return G.err.c_str(); }

int fesom_gpu_finalize(void) {
  if (G.stream) hipStreamSynchronize(G.stream);
  for (int i = 0; i < 2; i++) if (G.graph[i]) { hipGraphExecDestroy(G.graph[i]); G.graph[i] = nullptr; }
  for (void *p : G.allocs) hipFree(p);
  G.allocs.clear(); G.fields.clear(); G.mon_col = G.mon_out = nullptr;
  for (int b = 0; b < 2; b++) { if (G.frc_pin[b]) { hipHostFree(G.frc_pin[b]); G.frc_pin[b] = nullptr; hipEventDestroy(G.frc_ev[b]); G.frc_ev[b] = nullptr; } }
  G.frc_dev = nullptr;
  if (G.stream && !G.ext_stream) hipStreamDestroy(G.stream);
  G.stream = nullptr; G.ext_stream = false;
  for (int i = 0; i < 3; i++) if (G.side[i]) { hipStreamDestroy(G.side[i]); G.side[i] = nullptr; }
  G.ready = false; G.solver_only = false;
  return 0;
}

int fesom_gpu_init(const fesom_mesh_desc *d, const fesom_part_desc *part, const fesom_params *par) {
  if (G.ready) fesom_gpu_finalize();
  G.err.clear();
  g_alloc_failed = false;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { G.err = "no HIP device: the MI355X path has no CPU fallback"; fprintf(stderr, "fesom_gpu: %s\n", G.err.c_str()); return 2; }
  if (d->nl > 64) { G.err = "fesom_gpu_init: nl > 64 levels not supported by the one-wave-per-column kernels"; return 3; }
  if (par->which_ale < 0 || par->which_ale > 2) { G.err = "fesom_gpu_init: which_ale must be linfs (0), zlevel (1) or zstar (2)"; return 3; }
  if (par->which_ale == 1 && (par->lzstar_lev < 1 || par->lzstar_lev > 32 || par->lzstar_lev + 1 > d->nl - 1)) { G.err = "fesom_gpu_init: which_ALE='zlevel' needs 1 <= lzstar_lev <= 32 and lzstar_lev + 1 layers"; return 3; }
  if ((par->mom_adv != 2 && par->mom_adv != 3) || par->visc_option < 1 || par->visc_option > 8) { G.err = "fesom_gpu_init: only mom_adv=2 or 3, visc_option=1..8 are implemented"; return 3; }
  if (par->visc_option == 8 && part && part->npes > 1) { G.err = "fesom_gpu_init: visc_option=8 (backscatter with the uke budget) is built for one partition"; return 3; }
  // (mom_adv = 3 reads hpressure, which the reference forms with which_ALE='linfs' only, oce_ale_pressure_bv.F90:262; with zstar / zlevel the array keeps
  //  the zeros of array_setup, oce_setup_step.F90:384, and compute_vel_rhs_vinv runs without a baroclinic pressure term -- kept as it is: run pi_pp_vinv)
  const bool pgf_cav = par->which_ale == 0 && par->use_cavity && par->use_cavity_partial_cell;       // linfs with partial cells at the shelf base: 'sergey', 'shchepetkin', 'easypgf' (src/oce_ale_pressure_bv.F90:388-403)
  if (par->which_pgf != 0 && par->which_pgf != 1 && !(par->which_pgf == 2 && par->which_ale == 0) && !(par->which_pgf == 3) && !(par->which_pgf == 4 && pgf_cav) && !(par->which_ale == 0 && !par->use_partial_cell && !pgf_cav)) {
    G.err = "fesom_gpu_init: which_pgf must be 'shchepetkin' (0), 'cubicspline' (1), with linfs 'nemo' (2), 'easypgf' (3), or with linfs and use_cavity_partial_cell 'sergey' (4)"; return 3;
  }
  if (par->tra_adv_ver < 0 || par->tra_adv_ver > 3 || par->tra_adv_hor < 0 || par->tra_adv_hor > 2) {
    G.err = "fesom_gpu_init: tra_adv_ver must be QR4C (0), CDIFF (1), UPW1 (2) or PPM (3), tra_adv_hor MFCT (0), MUSCL (1) or UPW1 (2), tra_adv_lim='FCT'"; return 3;
  }
  if (par->tra_adv_lim < 0 || par->tra_adv_lim > 1) { G.err = "fesom_gpu_init: tra_adv_lim must be 'FCT' (0) or 'NON' (1)"; return 3; }
  if (par->mix_scheme < 0 || par->mix_scheme > 2) { G.err = "fesom_gpu_init: mix_scheme must be 0 (constant), 1 (KPP) or 2 (PP); the cvmix schemes are not implemented"; return 3; }
  if (!par->use_cavity)
    for (int e = 0; e < d->myDim_elem2D + d->eDim_elem2D; e++)
      if (d->ulevels[e] != 1) { G.err = "fesom_gpu_init: the mesh has ice-shelf cavities (ulevels > 1) but use_cavity is off"; return 3; }
  if (par->use_cavity && par->which_pgf != 0 && par->which_pgf != 3 && par->which_pgf != 1 && !(par->which_pgf == 2 && par->which_ale == 0 && !pgf_cav) && !(par->which_pgf == 4 && pgf_cav) && !(par->which_ale == 0 && !par->use_partial_cell && !pgf_cav)) {
    G.err = "fesom_gpu_init: with cavities which_pgf must be 'shchepetkin', 'easypgf' or (linfs with use_cavity_partial_cell) 'sergey'; 'cubicspline' and 'nemo' only with their own conditions"; return 3; }
  if (fesom_internal_select_device(G.err)) { fprintf(stderr, "fesom_gpu: %s\n", G.err.c_str()); return 2; }
  HIPCHK(hipStreamCreate(&G.stream));
  for (int i = 0; i < 3; i++) HIPCHK(hipStreamCreateWithFlags(&G.side[i], hipStreamNonBlocking));   // (stream priorities: no effect, measured)
  G.serial = getenv("FESOM_GPU_SERIAL") != nullptr;
  // Measured on MI355X/ROCm 7.2 (pi): eager 4-stream DAG 0.80 ms/step, hipGraph of the same DAG 0.86, serial chain 0.90
  // (graph replay serialises most branches; kernels are >= 5 us so the host launch rate is not the limit).
  // Default: the eager 4-stream DAG (pooled events).  FESOM_GPU_GRAPH=1 replays the same DAG as ONE hipGraph; it is no faster on pi
  // and the HIP runtime bundled with PyTorch 2.10 (ROCm 7.0) recurses without bound in hipStreamEndCapture for this multi-stream
  // capture (the ROCm 7.2 runtime of /opt/rocm is fine), so it stays opt-in for hosts that link the system runtime.
  G.use_graph = getenv("FESOM_GPU_GRAPH") != nullptr;
  if (G.use_graph) {
    // the HIP runtime bundled with PyTorch wheels (ROCm 7.0) recurses without bound in hipStreamEndCapture for this multi-stream
    // capture: refuse graph mode there instead of crashing; the system runtime of /opt/rocm (7.2) is fine
    Dl_info di;
    if (dladdr((void *)hipStreamEndCapture, &di) && di.dli_fname && strstr(di.dli_fname, "/torch/")) {
      fprintf(stderr, "fesom_gpu: FESOM_GPU_GRAPH ignored: libamdhip64 comes from a PyTorch wheel (%s), whose stream capture of a multi-stream step is broken; eager stream DAG used\n", di.dli_fname);
      G.use_graph = false;
    }
  }
  DM &m = G.m;
  memset(&m, 0, sizeof(m));
  m.p = *par;
  m.N = d->myDim_nod2D + d->eDim_nod2D; m.E = d->myDim_elem2D + d->eDim_elem2D; m.D = d->myDim_edge2D + d->eDim_edge2D;
  m.myN = d->myDim_nod2D; m.myE = d->myDim_elem2D; m.myD = d->myDim_edge2D;
  {   // shape of the kernels that end in a Thomas sweep: tiles on CORE2-class meshes (dev.h:TL_MIN_COLUMNS), FESOM_GPU_TILE=0/1 overrides
    const char *tl = getenv("FESOM_GPU_TILE");
    m.use_tile = tl ? atoi(tl) : (m.myN >= TL_MIN_COLUMNS ? 1 : 0);
    if (m.use_tile < 0 || m.use_tile > 4) m.use_tile = 1;
    { const char *e2 = getenv("FESOM_GPU_TRU_NT2"); m.tru_nt2 = e2 ? atoi(e2) : 1; }     // both tracers of a column in one wave (pi: 55 against 59 us; 0 = one tracer per wave)
  }
  m.nl = d->nl; m.nlm1 = d->nl - 1; m.ntr = par->num_tracers; m.maxk = d->max_nod_in_elem; m.nza = d->ssh_nza; m.edge2D_in = d->edge2D_in;
  const size_t N = m.N, E = m.E, D = m.D, nl = m.nl, n1 = m.nlm1;
  const int EX = m.E + d->eXDim_elem2D;
  m.EX = EX;
  // ---- connectivity (0-based)
  std::vector<int> en = minus1(d->elem2D_nodes, 3 * (size_t)EX), ed = minus1(d->edges, 2 * D), et = minus1(d->edge_tri, 2 * D);
  m.elem_nodes = dev_upload(en); m.edges = dev_upload(ed); m.edge_tri = dev_upload(et);
  m.nie = dev_upload(minus1(d->nod_in_elem2D, (size_t)m.maxk * N));
  m.nie_num = dev_upload(std::vector<int>(d->nod_in_elem2D_num, d->nod_in_elem2D_num + N));
  m.nlev = dev_upload(std::vector<int>(d->nlevels, d->nlevels + EX)); m.ulev = dev_upload(std::vector<int>(d->ulevels, d->ulevels + EX));   // incl. the extended element halo
  m.nlev_n = dev_upload(std::vector<int>(d->nlevels_nod2D, d->nlevels_nod2D + N));
  m.ulev_n = dev_upload(std::vector<int>(d->ulevels_nod2D, d->ulevels_nod2D + N));
  m.nlev_n_min = dev_upload(std::vector<int>(d->nlevels_nod2D_min, d->nlevels_nod2D_min + N));
  m.ulev_n_max = dev_upload(std::vector<int>(d->ulevels_nod2D_max, d->ulevels_nod2D_max + N));
  m.edge_glob = dev_upload(std::vector<int>(d->myList_edge2D, d->myList_edge2D + D));
  m.updn = dev_upload(minus1(d->edge_up_dn_tri, 2 * (size_t)m.myD));
  // node -> incident owned edges in increasing edge order (= order of the reference's scatter-adds)
  {
    std::vector<int> ptr(N + 1, 0), idx, sgn;
    for (int e = 0; e < m.myD; e++) { ptr[ed[2 * e] + 1]++; ptr[ed[2 * e + 1] + 1]++; }
    for (size_t n = 0; n < N; n++) ptr[n + 1] += ptr[n];
    idx.resize(ptr[N]); sgn.resize(ptr[N]);
    std::vector<int> fill(ptr.begin(), ptr.end() - 1);
    for (int e = 0; e < m.myD; e++) {
      int a = ed[2 * e], b = ed[2 * e + 1];
      idx[fill[a]] = e; sgn[fill[a]++] = 1;
      idx[fill[b]] = e; sgn[fill[b]++] = -1;
    }
    m.ne_ptr = dev_upload(ptr); m.ne_idx = dev_upload(idx); m.ne_sgn = dev_upload(sgn);
    std::vector<unsigned> rng(idx.size());
    for (size_t q = 0; q < idx.size(); q++) {
      int e = idx[q], e1 = et[2 * e], e2 = et[2 * e + 1];
      int lo = d->ulevels[e1], hi = d->nlevels[e1] - 1;
      if (e2 >= 0) { lo = std::min(lo, d->ulevels[e2]); hi = std::max(hi, d->nlevels[e2] - 1); }
      rng[q] = (unsigned)lo | ((unsigned)hi << 8);
    }
    m.ne_rng = dev_upload(rng);
  }
  // element -> its internal edges (increasing) with the edge_tri slot it occupies
  {
    std::vector<int> idx(3 * E, 0), side(3 * E, 0), cnt(E, 0);
    for (size_t e = 0; e < D; e++) {
      if (d->myList_edge2D[e] > d->edge2D_in) continue;
      for (int k = 0; k < 2; k++) {
        int el = et[2 * e + k];
        if (el < 0 || el >= (int)E) continue;
        if (cnt[el] < 3) { idx[3 * el + cnt[el]] = (int)e; side[3 * el + cnt[el]] = k + 1; cnt[el]++; }
      }
    }
    m.ee_idx = dev_upload(idx); m.ee_side = dev_upload(side);
  }
  // SSH operator (local 0-based CSR) + stiffness update lists in the reference's scatter order (oce_ale.F90:1399-1450)
  {
    std::vector<int> rp(m.myN + 1), ci(m.nza);
    for (int i = 0; i <= m.myN; i++) rp[i] = d->ssh_rowptr[i] - d->ssh_rowptr[0];
    for (int j = 0; j < m.nza; j++) ci[j] = d->ssh_colind_loc[j] - 1;
    m.rowptr = dev_upload(rp); m.colind = dev_upload(ci);
    m.ssh_maxnnz = 0;
    for (int i = 0; i < m.myN; i++) m.ssh_maxnnz = std::max(m.ssh_maxnnz, rp[i + 1] - rp[i]);
    {   // natural order wherever other phases read the ELL operator: partitioned runs, multi-workgroup solve
      const bool one_wg = (!part || part->npes <= 1) && m.myN <= 4096 && m.ssh_maxnnz <= 10;
      std::vector<int> perm, inv, wid;
      solver_row_order(rp.data(), m.myN, m.ssh_maxnnz, one_wg, perm, inv, wid);
      m.sv_cols = (unsigned short *)dev_upload(ell_cols(rp.data(), ci.data(), m.myN, m.ssh_maxnnz, perm, inv));
      m.sv_perm = dev_upload(perm); m.sv_inv = dev_upload(inv); m.sv_wid = dev_upload(wid);
    }
    {
      const int W = m.ssh_maxnnz <= 10 ? 10 : 16, NP = (m.myN + 63) / 64 * 64;
      std::vector<int> c32((size_t)W * NP, 0);
      for (int i = 0; i < NP; i++)
        for (int k = 0; k < W; k++) c32[(size_t)k * NP + i] = (i < m.myN && rp[i] + k < rp[i + 1]) ? ci[rp[i] + k] : (i < m.myN ? i : 0);
      m.sv_colsi = dev_upload(c32);
    }
    std::vector<std::vector<std::pair<int, double>>> lists(m.nza);
    std::vector<int> pos(N, -1);
    for (int e = 0; e < m.myD; e++)
      for (int j = 0; j < 2; j++) {
        int row = ed[2 * e + j];
        if (row >= m.myN) continue;
        for (int q = rp[row]; q < rp[row + 1]; q++) pos[ci[q]] = q;
        for (int i = 0; i < 2; i++) {
          int el = et[2 * e + i];
          if (el < 0) continue;
          const double *gs = d->gradient_sca + 6 * (size_t)el;
          const double *ec = d->edge_cross_dxdy + 4 * (size_t)e;
          for (int k = 0; k < 3; k++) {
            double coef = gs[k] * ec[2 * i + 1] - gs[3 + k] * ec[2 * i];
            int sg = ((i == 1) ? -1 : 1) * ((j == 1) ? -1 : 1);
            lists[pos[en[3 * el + k]]].push_back({sg * (el + 1), coef});
          }
        }
      }
    std::vector<int> sp(m.nza + 1, 0), se; std::vector<double> sc;
    for (int p = 0; p < m.nza; p++) {
      sp[p + 1] = sp[p] + (int)lists[p].size();
      for (auto &c : lists[p]) { se.push_back(c.first); sc.push_back(c.second); }
    }
    m.su_ptr = dev_upload(sp); m.su_elem = dev_upload(se); m.su_coef = dev_upload(sc);
  }
  // ---- geometry
  m.elem_area = dev_upload_d(d->elem_area, EX); m.area = dev_upload_d(d->area, nl * N); m.areasvol = dev_upload_d(d->areasvol, nl * N);
  m.areasvol_inv = dev_upload_d(d->areasvol_inv, nl * N); m.gsca = dev_upload_d(d->gradient_sca, 6 * (size_t)m.myE);
  m.ecd = dev_upload_d(d->edge_cross_dxdy, 4 * D); m.edxy = dev_upload_d(d->edge_dxdy, 2 * D); m.elem_cos = dev_upload_d(d->elem_cos, EX);
  m.coriolis = dev_upload_d(d->coriolis, m.myE); m.zbar_e_bot = dev_upload_d(d->zbar_e_bot, E); m.zbar_e_srf = dev_upload_d(d->zbar_e_srf, E); m.zbar_n_bot = dev_upload_d(d->zbar_n_bot, N);
  m.zbar = dev_upload_d(d->zbar, nl); m.Z = dev_upload_d(d->Z, n1);
  // ---- fields
#define F(f, c) m.f = field(#f, c)
#define FT(f, c) m.f = field(#f, c, m.ntr)
  F(tr_arr, n1 * N * m.ntr); F(tr_arr_old, n1 * N * m.ntr);
  F(density_m_rho0, n1 * N); F(hnode, n1 * N); F(hnode_new, n1 * N); F(Z_3d_n, n1 * N); F(sw_alpha, n1 * N); F(sw_beta, n1 * N);
  FT(del_ttf, n1 * N); FT(fct_LO, n1 * N); FT(fct_ttf_max, n1 * N); FT(fct_ttf_min, n1 * N); FT(fct_plus, n1 * N); FT(fct_minus, n1 * N); F(Ki, n1 * N);
  F(bvfreq, nl * N); F(hpressure, nl * N); F(zbar_3d_n, nl * N); F(Wvel, nl * N); F(Wvel_e, nl * N); F(Wvel_i, nl * N); F(CFL_z, nl * N);
  F(Kv, nl * N); FT(tr_z, nl * N); FT(adv_flux_ver, nl * N);
  F(Unode, 2 * n1 * N); F(Unode_rhs, 2 * n1 * N); F(sigma_xy, 2 * n1 * N); F(neutral_slope, 3 * n1 * N); F(slope_tapered, 3 * n1 * N); F(U_c, 2 * n1 * N);
  F(eta_n, N); F(d_eta, N); F(ssh_rhs, N); F(ssh_rhs_old, N); F(hbar, N); F(hbar_old, N); F(MLD1, N); F(MLD2, N);
  F(UV, 2 * n1 * E); F(UV_rhs, 2 * n1 * E); F(UV_rhsAB, 2 * n1 * E); FT(tr_xy, 2 * n1 * EX); FT(tr_xy_ab, 2 * n1 * EX); F(U_b, 2 * n1 * E);
  F(pgf_x, n1 * E); F(pgf_y, n1 * E); F(helem, n1 * E); F(Av, nl * E); F(dhe, E);
  FT(adv_flux_hor, n1 * D); FT(adv_flux_raw, n1 * D); FT(flux_lo_hor, n1 * D); FT(diff_flux, n1 * D); FT(edge_up_dn_grad, 4 * n1 * D); F(edge_c12, D);
  m.cl_grad = nullptr; m.cl_need = nullptr;
  m.density_ref = nullptr;
  if (par->use_density_ref || par->use_cavity) F(density_ref, n1 * N);       // filled by k_init_density_ref at the first state upload (ocean_setup: init_ref_density)
  G.dref_done = false;
  if (m.use_tile && !getenv("FESOM_GPU_NO_CLUSTER_GRAD")) {
    // k_flux_hor<FUSED> forms fill_up_dn_grad on the fly; where the two upwind triangles of an edge do not cover a node's column (ragged bottom, boundary
    // edges) it needs the cluster mean of the gradient at that node -- per NODE in k_cluster_grad instead of once per incident edge
    std::vector<unsigned char> need(N, 0);
    const int *ud = d->edge_up_dn_tri;
    for (int e = 0; e < m.myD; e++) {
      const int a = ed[2 * e], b = ed[2 * e + 1];
      if (ud[2 * e] > 0 && ud[2 * e + 1] > 0) {
        const int nzmin = std::max(d->ulevels_nod2D_max[a], d->ulevels_nod2D_max[b]), nzmax = std::min(d->nlevels_nod2D_min[a], d->nlevels_nod2D_min[b]);
        if (nzmin > d->ulevels_nod2D[a] || nzmax - 1 < d->nlevels_nod2D[a] - 1) need[a] = 1;
        if (nzmin > d->ulevels_nod2D[b] || nzmax - 1 < d->nlevels_nod2D[b] - 1) need[b] = 1;
      } else { need[a] = 1; need[b] = 1; }
    }
    m.cl_need = dev_upload(need);
    FT(cl_grad, 2 * n1 * N);
  }
  F(ssh_values, m.nza);
  if ((par->which_pgf == 0 && !(par->which_ale == 0 && !par->use_partial_cell && !pgf_cav)) || par->which_pgf == 4) { F(pgf_A, n1 * N); F(pgf_B, n1 * N); }      // shchepetkin variants, 'sergey'
  if (par->visc_option <= 3) { F(Visc, n1 * E); F(leith_aux, n1 * N); }
  if (par->visc_option == 8) {
    F(uke, n1 * E); F(v_back, n1 * E); F(uke_rhs, n1 * E); F(uke_rhs_old, n1 * E); F(uke_dif, n1 * E); F(uke_dis, n1 * E); F(uke_back, n1 * E);
    F(UV_dis_tend, 2 * n1 * E); F(UV_back_tend, 2 * n1 * E); F(v8_work, 2 * n1 * N); F(v8_rb, N);
    if (!m.coriolis_node) m.coriolis_node = dev_upload_d(d->coriolis_node, N);
  }
  if (par->smooth_bh_tra) FT(bh_tmp, n1 * N);
  m.ale_flag = dev_alloc<int>(1); HIPCHK(hipMemset(m.ale_flag, 0, sizeof(int)));
  if (par->SPP) { std::vector<double> gl(N); for (size_t n = 0; n < N; n++) gl[n] = d->geo_coord_nod2D[2 * n + 1]; m.geo_lat = dev_upload(gl); }
  if (par->visc_option <= 3 || par->mom_adv == 3) F(vorticity, n1 * N);
  if (par->mom_adv == 3) {
    F(KE_node, n1 * N);
    std::vector<unsigned char> wall(N, 0);
    for (int e = 0; e < m.myD; e++) if (d->myList_edge2D[e] > d->edge2D_in) { wall[ed[2 * e]] = 1; wall[ed[2 * e + 1]] = 1; }
    m.wall_node = dev_upload(wall);
    if (!m.coriolis_node) m.coriolis_node = dev_upload_d(d->coriolis_node, N);
  }
  if (par->use_momix) {       // where mo_convect applies the Monin-Obukhov mixing (oce_mo_conv.F90:28-31, :95); rad = pi/180 with the reference's pi (oce_modules.F90:11-12)
    F(mixlength, N);
    const double rad = 3.14159265358979 / 180.0, lim = par->momix_lat * rad;
    std::vector<int> fn(N), fe(E, 0);
    for (size_t n = 0; n < N; n++) fn[n] = !(d->geo_coord_nod2D[2 * n + 1] > lim) && d->ulevels_nod2D[n] <= 1;
    for (int e = 0; e < m.myE; e++) {
      const int n1 = d->elem2D_nodes[3 * e] - 1, n2 = d->elem2D_nodes[3 * e + 1] - 1, n3 = d->elem2D_nodes[3 * e + 2] - 1;
      fe[e] = ((d->geo_coord_nod2D[2 * n1 + 1] + d->geo_coord_nod2D[2 * n2 + 1]) + d->geo_coord_nod2D[2 * n3 + 1]) / 3.0 <= lim && d->ulevels[e] <= 1;
    }
    m.momix_node = dev_upload(fn); m.momix_elem = dev_upload(fe);
  }
  if (par->Fer_GM) { F(fer_K, nl * N); F(fer_gamma, 2 * nl * N); F(fer_Wvel, nl * N); F(fer_c, N); F(fer_UV, 2 * n1 * E); }
  {   // surface forcing: ONE device block (one host->device copy per fesom_gpu_set_forcing), fields are views into it
    G.frc_count = 2 * E + 7 * N + (par->use_sw_pene ? nl * N : 0) + (par->use_momix ? 3 * N : 0) + (par->use_floatice ? 2 * N : 0) + (par->l_mslp ? N : 0) + (par->use_global_tides ? N : 0) + (par->SPP ? 2 * N : 0);
    G.frc_dev = dev_alloc<double>(G.frc_count);
    double *q = G.frc_dev;
    auto view = [&](const char *name, size_t cnt) { double *r = q; G.fields[name] = Field{r, cnt, 1}; q += cnt; return r; };
    m.stress_surf = view("stress_surf", 2 * E); m.heat_flux = view("heat_flux", N); m.water_flux = view("water_flux", N);
    m.virtual_salt = view("virtual_salt", N); m.relax_salt = view("relax_salt", N); m.real_salt_flux = view("real_salt_flux", N);
    m.stress_atmoce_x = view("stress_atmoce_x", N); m.stress_atmoce_y = view("stress_atmoce_y", N);
    m.sw_3d = par->use_sw_pene ? view("sw_3d", nl * N) : nullptr;
    if (par->use_momix) { m.u_ice = view("u_ice", N); m.v_ice = view("v_ice", N); m.a_ice = view("a_ice", N); }
    if (par->use_floatice) { m.m_ice = view("m_ice", N); m.m_snow = view("m_snow", N); }
    if (par->l_mslp) m.press_air = view("press_air", N);
    if (par->use_global_tides) m.ssh_gp = view("ssh_gp", N);
    if (par->SPP) { m.thdgr = view("thdgr", N); m.S_oc = view("S_oc_array", N); }
  }
  {   // kernels that gather the 3 nodes of every element of a node's cluster (smooth_nod3D of KPP, compute_sigma_xy): the DISTINCT nodes of the cluster (7 on a regular mesh against 18 gathers)
      // are staged once per wave (k_kpp_smooth_u); list of distinct nodes in order of first appearance + where each element's nodes sit in it
    int maxu = 1;
    std::vector<std::vector<int>> U(m.myN);
    std::vector<int> pos((size_t)m.maxk * m.myN, 0), cnt(m.myN, 0);
    for (int n = 0; n < m.myN; n++) {
      std::vector<int> &u = U[n];
      for (int k = 0; k < d->nod_in_elem2D_num[n]; k++) {
        const int el = d->nod_in_elem2D[(size_t)m.maxk * n + k] - 1;
        int pk = 0;
        for (int j = 0; j < 3; j++) {
          const int nd = en[3 * (size_t)el + j];
          size_t q = std::find(u.begin(), u.end(), nd) - u.begin();
          if (q == u.size()) u.push_back(nd);
          pk |= (int)q << (8 * j);
        }
        pos[(size_t)m.maxk * n + k] = pk;
      }
      cnt[n] = (int)u.size();
      maxu = std::max(maxu, cnt[n]);
    }
    std::vector<int> nb((size_t)maxu * m.myN, 0);
    for (int n = 0; n < m.myN; n++) for (size_t q = 0; q < U[n].size(); q++) nb[(size_t)maxu * n + q] = U[n][q];
    m.cl_nb = dev_upload(nb); m.cl_nbn = dev_upload(cnt); m.cl_pos = dev_upload(pos); m.cl_maxu = maxu;
  }
  if (par->mix_scheme == 1) {
    F(dbsfc, nl * N);
    F(kpp_viscA, nl * N); F(kpp_Kv1, nl * N); F(kpp_Kv2, nl * N); F(kpp_ghats, n1 * N); F(kpp_hbl, N); F(kpp_caseA, N); F(kpp_dkm1, 3 * N);
    m.kpp_blmc = field("kpp_blmc", nl * N, 3); m.kpp_sA = field("kpp_sA", nl * N, 3); m.kpp_sB = field("kpp_sB", nl * N, 3);
    for (int j = 0; j < 3; j++) G.fields[std::string("kpp_blmc") + char('1' + j)] = Field{m.kpp_blmc + (size_t)j * nl * N, nl * N, 1};
    m.kpp_kbl = dev_alloc<int>(N);
    m.coriolis_node = dev_upload_d(d->coriolis_node, N);
    // constants and look-up tables of oce_mixing_kpp_init (oce_ale_mixing_kpp.F90:97-201).  Which library call the reference's
    // build makes for each power is pinned on the reference run of tests/golden (sqrt for **(1/2), pow otherwise; cg as cbrt)
    const double epsln = 1.0e-40, eps_kpp = 0.1, vonk = 0.4, conc1 = 5.0, zmin = -4.e-7, zmax = 0.0, umin = 0.0, umax = 0.04;
    const double conam = 1.257, concm = 8.380, conc2 = 16.0, zetam = -0.2, conas = -28.86, concs = 98.96, conc3 = 16.0, zetas = -1.0;
    const int nni = 890, nnj = 480;
    m.kpp_deltaz = (zmax - zmin) / (double)(nni + 1); m.kpp_deltau = (umax - umin) / (double)(nnj + 1);
    m.kpp_Vtc = par->concv * sqrt(0.2 / concs / eps_kpp) / (vonk * vonk) / par->Ricr;
    m.kpp_cg = 10.0 * vonk * cbrt(concs * vonk * eps_kpp);
    std::vector<double> wmt((size_t)(nni + 2) * (nnj + 2)), wst(wmt.size());
    for (int i = 0; i <= nni + 1; i++) {
      const double zehat = m.kpp_deltaz * (double)i + zmin;
      for (int j = 0; j <= nnj + 1; j++) {
        const double usta = m.kpp_deltau * (double)j + umin, u3 = usta * usta * usta, zeta = zehat / (u3 + epsln);
        double wm, ws;
        if (zehat >= 0.) { wm = vonk * usta / (1. + conc1 * zeta); ws = wm; }
        else {
          wm = zeta > zetam ? vonk * usta * pow(1. - conc2 * zeta, 1. / 4.) : vonk * pow(conam * u3 - concm * zehat, 1. / 3.);
          ws = zeta > zetas ? vonk * usta * sqrt(1. - conc3 * zeta) : vonk * pow(conas * u3 - concs * zehat, 1. / 3.);
        }
        wmt[(size_t)j * (nni + 2) + i] = wm; wst[(size_t)j * (nni + 2) + i] = ws;
      }
    }
    m.kpp_wmt = dev_upload(wmt); m.kpp_wst = dev_upload(wst);
  }
  if (par->toy_soufflet) { F(Tclim, n1 * N); F(Uclim, n1 * E); F(toy_zvel, n1 * 100); F(toy_ztem, n1 * 100); }
  else if (par->clim_relax > 1.0e-8) { F(Tclim, n1 * N); F(Sclim, n1 * N); F(relax2clim, N); }
  F(sv_vals, 16 * (N + 64)); F(sv_dinv, N + 64); F(sv_b, N + 64); F(sv_r, N + 64); F(sv_r0, N + 64); F(sv_p, N + 64); F(sv_v, N + 64); F(sv_s, N + 64); F(sv_t, N + 64);
  F(sv_ph, N + 64); F(sv_x0, 16 * (N + 64)); F(sv_snap, N); F(sv_ph2, N + 64); F(sv_v2, N + 64);
  if (!part || part->npes <= 1) F(sv_rdinv, N + 64);                 // 1 / D of every row, formed with the row scales off the critical chain (one partition: no halo of it is needed)
  F(sv_part, 8 * ((N + 255) / 256 + 1)); F(sv_red, 8); F(sv_kry, 48);
  F(sv_resid, 1); F(sv_h1, N); F(sv_h2, N); F(sv_h3, N); F(sv_scale, N + 64);
  F(sv_bn, N + 64); F(sv_x, N + 64); F(sv_pd, N + 64); F(sv_sn, N + 64); F(sv_sh, N + 64);
  m.sv_extrap = 1;
#undef F
#undef FT
  m.sv_info = dev_alloc<int>(4);
  m.lat_deg = nullptr;
  if (!par->Kv0_const) {                // geo_coord_nod2D(2,node)/rad with the reference's rad = pi/180, pi = 3.14159265358979 (oce_modules.F90:11-12)
    const double rad = 3.14159265358979 / 180.0;
    std::vector<double> lat(N);
    for (size_t n = 0; n < N; n++) lat[n] = d->geo_coord_nod2D[2 * n + 1] / rad;
    m.lat_deg = dev_upload(lat);
  }
  m.nb_lay = nullptr;
  if (par->tra_adv_hor == 1) {          // nboundary_lay: the layer below which a node touches the boundary (oce_muscl_adv.F90:74-104, owned edges)
    std::vector<int> nb(N, (int)nl - 1);
    for (size_t e = 0; e < (size_t)d->myDim_edge2D; e++) {
      const int n1 = d->edges[2 * e] - 1, n2 = d->edges[2 * e + 1] - 1, t1 = d->edge_tri[2 * e], t2 = d->edge_tri[2 * e + 1];
      if (t1 <= 0 || t2 <= 0) { nb[n1] = 0; nb[n2] = 0; }
      else {
        const int lv = std::min(d->nlevels[t1 - 1], d->nlevels[t2 - 1]) - 1;
        nb[n1] = std::min(nb[n1], lv); nb[n2] = std::min(nb[n2], lv);
      }
    }
    m.nb_lay = dev_upload(nb);
  }
  if (par->Fer_GM || par->Redi) {
    // mesh-only part of the horizontal GM scaling (init_Redi_GM, src/oce_fer_gm.F90:204-232; scaling_Rossby is rejected above)
    std::vector<double> sc(N, 1.0);
    for (size_t n = 0; n < N; n++) {
      double reso = d->mesh_resolution[n], scaling = 1.;
      if (par->scaling_resolution) scaling = scaling * pow(reso / 100000., par->K_GM_resscalorder);
      if (reso / 1000.0 < par->K_GM_rampmax) scaling = scaling * std::max((reso / 1000.0 - par->K_GM_rampmin) / (par->K_GM_rampmax - par->K_GM_rampmin), 0.);
      sc[n] = scaling;
    }
    m.gm_scal_static = dev_upload(sc);
    m.gm_scal_A = m.gm_scal_B = m.mesh_resolution = nullptr;
    if (par->scaling_Rossby) {      // the Rossby factor comes first in the reference's product (:196-225): the two mesh-only factors separately (1 where a factor is off)
      std::vector<double> fa(N, 1.0), fb(N, 1.0);
      for (size_t n = 0; n < N; n++) {
        const double reso = d->mesh_resolution[n];
        if (par->scaling_resolution) fa[n] = pow(reso / 100000., par->K_GM_resscalorder);
        if (reso / 1000.0 < par->K_GM_rampmax) fb[n] = std::max((reso / 1000.0 - par->K_GM_rampmin) / (par->K_GM_rampmax - par->K_GM_rampmin), 0.);
      }
      m.gm_scal_A = dev_upload(fa); m.gm_scal_B = dev_upload(fb); m.mesh_resolution = dev_upload_d(d->mesh_resolution, N);
      if (!m.coriolis_node) m.coriolis_node = dev_upload_d(d->coriolis_node, N);
    }
    for (size_t n = 0; n < N; n++) { double q = d->mesh_resolution[n] / 100000.0; sc[n] = par->K_hor * (q * q); }
    m.redi_k0 = dev_upload(sc);
    m.exp_batch = getenv("FESOM_GPU_EXP_BATCH") ? atoi(getenv("FESOM_GPU_EXP_BATCH")) : 127;
    m.gm_nzl = d->myDim_nod2D > 0 ? d->ulevels_nod2D_max[d->myDim_nod2D - 1] : 1;
    m.MLD1_ind = dev_alloc<int>(N);
  }
  G.npes = part ? part->npes : 1; G.mype = part ? part->mype : 0;
  G.precond_agreed = false; G.x_pending = false; G.generation++; g_prog_ready = false;
  G.hsend = G.hrecv = nullptr; G.hcap = 0; G.hsend1 = G.hrecv1 = nullptr; G.hcap1 = 0; G.toy_znum_global = false;
  if (part && part->npes > 1) {
    const fesom_com_desc *cs[3] = {&part->com_nod2D, &part->com_elem2D, &part->com_elem2D_full};
    for (int k = 0; k < 3; k++) {
      Ctx::Halo &h = G.halo[k];
      const fesom_com_desc &c = *cs[k];
      h.rPE.assign(c.rPE, c.rPE + c.rPEnum); h.rptr.assign(c.rptr, c.rptr + c.rPEnum + 1);
      h.sPE.assign(c.sPE, c.sPE + c.sPEnum); h.sptr.assign(c.sptr, c.sptr + c.sPEnum + 1);
      h.nrecv = h.rptr.back() - 1; h.nsend = h.sptr.back() - 1;
      h.slist_h = minus1(c.slist, h.nsend);
      h.rlist = dev_upload(minus1(c.rlist, h.nrecv)); h.slist = dev_upload(h.slist_h); h.slist_q = nullptr;
      h.rptr_d = dev_upload(h.rptr); h.sptr_d = dev_upload(h.sptr);
    }
  }
  G.sub_int = G.sub_cb = G.sub_cbh = Ctx::ColList();
  if (part && part->npes > 1) {
    std::vector<char> bnd(m.myN, 0);
    for (int e = 0; e < m.myD; e++) {
      const int a = ed[2 * e], b = ed[2 * e + 1];
      if (a < m.myN && b >= m.myN) bnd[a] = 1;
      if (b < m.myN && a >= m.myN) bnd[b] = 1;
    }
    std::vector<int> li, lb;
    for (int n = 0; n < m.myN; n++) (bnd[n] ? lb : li).push_back(n);
    std::vector<int> lbh(lb);
    for (int n = m.myN; n < m.N; n++) lbh.push_back(n);
    G.sub_int.d = dev_upload(li); G.sub_int.n = (int)li.size();
    G.sub_cb.d = dev_upload(lb); G.sub_cb.n = (int)lb.size();
    G.sub_cbh.d = dev_upload(lbh); G.sub_cbh.n = (int)lbh.size();
  }
  if (par->toy_soufflet) {
    // static tables of the Soufflet hooks: compute_zonal_mean_ini (toy_channel_soufflet.F90:104-155) and the interpolation
    // headers of relax_zonal_vel / relax_zonal_temp (:57-70, :89-100); module constants :19-23
    const double lat0 = 0.0, ysize = 2000000.0;
    const double Ly = ysize / D_REARTH, dy = Ly / 100.0;
    auto interp = [&](double yy, int &nn, int &nn1, double &a) {
      a = 0;
      if (yy < dy / 2) { nn = 1; nn1 = 1; }
      else { nn = (int)floor(yy / dy - 0.5) + 1; nn1 = nn + 1; if (nn1 > 100) nn1 = nn; a = yy / dy + 0.5 - (double)nn; }
    };
    std::vector<int> bpos(m.myE), enn(2 * (size_t)m.myE), nnn(2 * N), bptr(101, 0), bidx;
    std::vector<double> ea(m.myE), na(N), znum(100, 0.0);
    const double *cn = d->coord_nod2D;
    for (int e = 0; e < m.myE; e++) {
      double ymean = ((cn[2 * en[3 * e] + 1] + cn[2 * en[3 * e + 1] + 1]) + cn[2 * en[3 * e + 2] + 1]) / 3.0;
      bpos[e] = (int)floor((ymean - lat0) / dy) + 1;
      if (bpos[e] < 1 || bpos[e] > 100) { G.err = "toy_soufflet: element outside the 100 latitude bins of the channel"; return 1; }
      interp(ymean - lat0, enn[2 * e], enn[2 * e + 1], ea[e]);
      if (en[3 * e] < m.myN) { bptr[bpos[e]]++; znum[bpos[e] - 1] += 1.0; }        // each element once: first node owned
    }
    for (int b = 0; b < 100; b++) bptr[b + 1] += bptr[b];
    bidx.resize(bptr[100]);
    { std::vector<int> fill(bptr.begin(), bptr.end() - 1);
      for (int e = 0; e < m.myE; e++) if (en[3 * e] < m.myN) bidx[fill[bpos[e] - 1]++] = e; }
    for (size_t n = 0; n < N; n++) interp(cn[2 * n + 1] - lat0, nnn[2 * n], nnn[2 * n + 1], na[n]);
    m.toy_bptr = dev_upload(bptr); m.toy_bidx = dev_upload(bidx); m.toy_e_nn = dev_upload(enn); m.toy_n_nn = dev_upload(nnn);
    m.toy_e_a = dev_upload(ea); m.toy_n_a = dev_upload(na); m.toy_znum = dev_upload(znum);
  }
  for (auto &kv : G.fields) if (!kv.second.p) { G.err = "device allocation failed"; return 1; }
  if (g_alloc_failed) { G.err = "fesom_gpu_init: a device allocation or upload of the mesh / halo lists failed (out of device memory?)"; fprintf(stderr, "fesom_gpu: %s\n", G.err.c_str()); return 1; }
  HIPCHK(hipMemcpy(m.ssh_values, d->ssh_values, sizeof(double) * m.nza, hipMemcpyHostToDevice));
  {   // Ki = K_hor*(mesh_resolution/100000)**2 (oce_setup_step.F90:328-331); Av/Kv constant when no mixing scheme is selected
    std::vector<double> ki(n1 * N), av(nl * E, par->A_ver), kv(nl * N, par->K_ver);
    for (size_t n = 0; n < N; n++) { double r = d->mesh_resolution[n] / 100000.0; for (size_t k = 0; k < n1; k++) ki[n * n1 + k] = par->K_hor * (r * r); }
    HIPCHK(hipMemcpy(m.Ki, ki.data(), sizeof(double) * ki.size(), hipMemcpyHostToDevice));
    if (par->Fer_GM) { std::vector<double> fk(nl * N, 500.0); HIPCHK(hipMemcpy(m.fer_K, fk.data(), sizeof(double) * fk.size(), hipMemcpyHostToDevice)); }      // fer_K=500 (oce_setup_step.F90:359; read back only under an ice shelf, kernels_gm.hip)
    if (par->mix_scheme == 0) {
      HIPCHK(hipMemcpy(m.Av, av.data(), sizeof(double) * av.size(), hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(m.Kv, kv.data(), sizeof(double) * kv.size(), hipMemcpyHostToDevice));
    }
  }
  m.sv_minv = nullptr; m.sv_mp = nullptr; m.sv_mc = nullptr; m.sv_xi_its = par->solver_xinv_its; m.sv_solves = 0;
  if (G.use_graph && m.sv_xi_its == 0) m.sv_xi_its = 1;          // (a captured step replays a fixed launch sequence)
  if (par->solver_precond == 1 && G.npes <= 1 && m.myN <= 4096 && m.ssh_maxnnz <= 10 && m.myN >= 64) {
    // explicit inverse of the row-scaled operator this run starts with (frozen, like the reference's ILU factors)
    std::vector<int> rp(m.myN + 1), ci(m.nza);
    for (int i = 0; i <= m.myN; i++) rp[i] = d->ssh_rowptr[i] - d->ssh_rowptr[0];
    for (int j = 0; j < m.nza; j++) ci[j] = d->ssh_colind_loc[j] - 1;
    if (xinv_device(m.myN, rp.data(), ci.data(), d->ssh_values, nullptr, m, G.allocs)) { m.sv_minv = nullptr; G.err = "fesom_gpu_init: the explicit-inverse SSH preconditioner could not be built (singular operator or out of memory)"; return 1; }
  }
  if (par->solver_precond == 1 && !m.sv_minv && m.ssh_maxnnz <= 16) {
    // operators beyond the explicit inverse and every partitioned run: RAS-Chebyshev (solver_ras.hip), frozen at the operator this run
    // starts with.  On a partition the patches cover the rank's owned rows (halo columns are outside every patch), so applying the
    // preconditioner needs no communication -- the role of the per-rank ILU factors in the reference's RAS (bicgstab_ras.c:49-259).
    // The choice does not depend on the rank's size, so all ranks take the same solver path (checked once more at the first step).
    std::vector<int> rp(m.myN + 1), ci(m.nza), inv;
    for (int i = 0; i <= m.myN; i++) rp[i] = d->ssh_rowptr[i] - d->ssh_rowptr[0];
    for (int j = 0; j < m.nza; j++) ci[j] = d->ssh_colind_loc[j] - 1;
    const char *off = getenv("FESOM_GPU_RAS_OFF_ON_RANK");          // (tests: a rank whose block does not qualify -- all ranks must then fall back together)
    if ((off && G.npes > 1 && atoi(off) == G.mype) || ras_device(m.myN, m.N, rp.data(), ci.data(), d->ssh_values, m.ssh_maxnnz, m, G.allocs, &inv)) m.rs_pinfo = nullptr;      // (does not qualify: Jacobi)
    if (m.rs_pinfo && G.npes > 1) {
      Ctx::Halo &h = G.halo[0];
      std::vector<int> sq(h.slist_h.size());
      for (size_t k = 0; k < sq.size(); k++) sq[k] = (h.slist_h[k] >= 0 && h.slist_h[k] < m.myN) ? inv[h.slist_h[k]] : h.slist_h[k];
      h.slist_q = dev_upload(sq);
      if (g_alloc_failed) { G.err = "fesom_gpu_init: device allocation failed (halo send list of the solver)"; return 1; }
    }
  }
  solver_prepare();
  tile_prepare_tra(); tile_prepare_dyn();
  G.first_step = 1;
  G.ready = true;
  HIPCHK(hipDeviceSynchronize());
  return 0;
}

#define NEED_CTX() if (!G.ready) { G.err = "fesom_gpu: not initialised"; return 1; }
// (a context that holds the distributed SSH solver alone, fesom_gpu_psolver_init_dist, has no ocean state: only the halo / copy / communicator calls work on it)
#define NEED_READY() do { NEED_CTX(); if (G.solver_only) { G.err = "fesom_gpu: the context holds the distributed SSH solver only (fesom_gpu_psolver_init_dist); call fesom_gpu_init for the ocean step"; return 1; } } while (0)

// zlevel: a column whose surface layer would fall below min_hnode needs the reference's local-zstar fallback (oce_ale.F90:1859-1942), which is not built (the
// reference itself stops in update_thickness_ale's non-conformable PACK there under flang): reported at the next synchronising call
static int check_ale_flag() {
  if (G.m.p.which_ale != 1) return 0;
  int f = 0;
  HIPCHK(hipMemcpy(&f, G.m.ale_flag, sizeof(int), hipMemcpyDeviceToHost));
  if (f) { G.err = "which_ALE='zlevel': the surface layer of a column fell below min_hnode of its resting thickness; the local-zstar fallback of vert_vel_ale is not implemented"; return 1; }
  return 0;
}
static int copy_state(const fesom_state_desc *st, bool up) {
  NEED_READY();
  struct { const char *n; double *h; } tab[] = {
      {"tr_arr", st->tr_arr}, {"tr_arr_old", st->tr_arr_old}, {"UV", st->UV}, {"UV_rhsAB", st->UV_rhsAB}, {"eta_n", st->eta_n},
      {"d_eta", st->d_eta}, {"ssh_rhs", st->ssh_rhs}, {"ssh_rhs_old", st->ssh_rhs_old}, {"hbar", st->hbar}, {"hbar_old", st->hbar_old},
      {"dhe", st->dhe}, {"hnode", st->hnode}, {"hnode_new", st->hnode_new}, {"helem", st->helem}, {"zbar_3d_n", st->zbar_3d_n},
      {"Z_3d_n", st->Z_3d_n}, {"Wvel", st->Wvel}, {"Wvel_e", st->Wvel_e}, {"Wvel_i", st->Wvel_i}, {"ssh_values", st->ssh_values}};
  HIPCHK(hipStreamSynchronize(G.stream));
  if (!up && check_ale_flag()) return 1;
  for (auto &t : tab) {
    if (!t.h) continue;
    Field &f = G.fields[t.n];
    size_t cnt = f.count;                 // the reference allocates these two for owned elements only (oce_ale.F90:108,112)
    if (!strcmp(t.n, "helem")) cnt = (size_t)G.m.nlm1 * G.m.myE;
    if (!strcmp(t.n, "dhe")) cnt = (size_t)G.m.myE;
    if (up) HIPCHK(hipMemcpy(f.p, t.h, cnt * sizeof(double), hipMemcpyHostToDevice));
    else HIPCHK(hipMemcpy(t.h, f.p, cnt * sizeof(double), hipMemcpyDeviceToHost));
  }
  if (up && G.m.density_ref && !G.dref_done && st->Z_3d_n) {      // init_ref_density of ocean_setup: from the initial layer depths, once
    launch_init_density_ref(G.m, G.stream);
    HIPCHK(hipStreamSynchronize(G.stream));
    G.dref_done = true;
  }
  return 0;
}
int fesom_gpu_upload_state(const fesom_state_desc *st) { return copy_state(st, true); }
int fesom_gpu_download_state(const fesom_state_desc *st) { return copy_state(st, false); }

int fesom_gpu_set_forcing(const fesom_forcing_desc *f) {
  NEED_READY();
  // The host arrays are pageable: they are gathered into one of two pinned staging buffers and go to the device block in ONE
  // asynchronous copy on the library's stream, so the caller is not synchronised with the previous step (the event of a
  // staging buffer, recorded two calls ago, bounds how far the host may run ahead).
  if (!G.frc_pin[0]) {
    for (int b = 0; b < 2; b++) {
      HIPCHK(hipHostMalloc((void **)&G.frc_pin[b], G.frc_count * sizeof(double), hipHostMallocDefault));
      HIPCHK(hipEventCreateWithFlags(&G.frc_ev[b], hipEventDisableTiming));
    }
  }
  const int b = (G.frc_cur ^= 1);
  HIPCHK(hipEventSynchronize(G.frc_ev[b]));
  const size_t N = G.m.N, E2 = (size_t)2 * G.m.E, myE2 = (size_t)2 * G.m.myE;     // stress_surf is (2,myDim_elem2D) on the host (oce_setup_step.F90:249)
  double *q = G.frc_pin[b];
  auto put = [&](const double *h, size_t host_cnt, size_t dev_cnt) {
    if (h) memcpy(q, h, host_cnt * sizeof(double)); else memset(q, 0, host_cnt * sizeof(double));
    if (dev_cnt > host_cnt) memset(q + host_cnt, 0, (dev_cnt - host_cnt) * sizeof(double));
    q += dev_cnt;
  };
  put(f->stress_surf, myE2, E2); put(f->heat_flux, N, N); put(f->water_flux, N, N); put(f->virtual_salt, N, N);
  put(f->relax_salt, N, N); put(f->real_salt_flux, N, N); put(f->stress_atmoce_x, N, N); put(f->stress_atmoce_y, N, N);
  if (G.m.sw_3d) put(f->sw_3d, (size_t)G.m.nl * N, (size_t)G.m.nl * N);
  if (G.m.p.use_momix) { put(f->u_ice, N, N); put(f->v_ice, N, N); put(f->a_ice, N, N); }
  if (G.m.p.use_floatice) { put(f->m_ice, N, N); put(f->m_snow, N, N); }
  if (G.m.p.l_mslp) put(f->press_air, N, N);
  if (G.m.p.use_global_tides) put(f->ssh_gp, N, N);
  if (G.m.p.SPP) { put(f->thdgr, N, N); put(f->S_oc_array, N, N); }
  HIPCHK(hipMemcpyAsync(G.frc_dev, G.frc_pin[b], G.frc_count * sizeof(double), hipMemcpyHostToDevice, G.stream));
  HIPCHK(hipEventRecord(G.frc_ev[b], G.stream));
  return 0;
}

static int call_named(const char *name, int arg);
// ---- partitioned step driven by the library.  The same phases, exchange points and solver loop as fesom2_amd/parallel.py:run_step,
// which the tests probe phase by phase; both issue the same kernels on the same data, so their results are bit-identical for the same
// transport.  Built for few, large messages and no host synchronisation outside the one convergence read-back of the SSH solve:
//   * the step is a PROGRAM (vector of ops) built once per context: kernel ops hold the launcher of their family, exchange ops their
//     resolved field lists, message layout and buffer offsets -- nothing is looked up by name while a step is enqueued;
//   * an exchange point may carry node AND element fields (one RCCL group; one callback per kind with the host transport), and the
//     kernel order inside a step is chosen so that fields that are ready at the same time travel together: 13 exchange points per
//     step with the reference's default physics (KPP + GM + Redi) outside the SSH solve, 8 with PP (the reference: ~50 calls);
//   * exchanges whose consumer is far away run on the communication stream (pack, RCCL group, unpack behind an event) while the
//     step's stream carries on -- the overlap the reference has at one place (src/oce_tracer_mod.F90:68-81).
namespace {
struct Sub { double *p; int W; bool q; };          // q: a solver vector in the patch order of the RAS preconditioner (send list as positions)
struct XSpec { int kind; std::vector<const char *> names; };
struct XPlan {
  // two layouts of the packed buffers: kind-major (a contiguous region per part, neighbour blocks inside: what the host transport's
  // exchange(kind, ...) callback expects) and neighbour-major (ONE message per neighbour, the parts' blocks one after the other inside
  // it: what the built-in transport sends -- one ncclSend / ncclRecv pair per neighbour and exchange point)
  struct Part { int kind, Wtot; std::vector<Sub> subs; size_t soff, roff; const long long *sbase_d = nullptr, *rbase_d = nullptr; };
  struct Msg { int peer; size_t soff, scnt, roff, rcnt; };
  std::vector<Part> parts;
  std::vector<Msg> msgs;
  size_t stot = 0, rtot = 0;
};
int halo_subfields(int kind, int nf, const char *const *names, std::vector<Sub> &subs, int &Wtot);
int halo_reserve(size_t doubles, int ch);
int xplan_build(const std::vector<XSpec> &spec, XPlan &x) {
  x.parts.clear(); x.msgs.clear(); x.stot = x.rtot = 0;
  for (const XSpec &sp : spec) {
    if (sp.kind < 0 || sp.kind > 2) { G.err = "halo: bad kind"; return 1; }
    XPlan::Part pt;
    pt.kind = sp.kind;
    if (halo_subfields(sp.kind, (int)sp.names.size(), sp.names.data(), pt.subs, pt.Wtot)) return 1;
    pt.soff = x.stot; pt.roff = x.rtot;
    x.stot += (size_t)G.halo[sp.kind].nsend * pt.Wtot; x.rtot += (size_t)G.halo[sp.kind].nrecv * pt.Wtot;
    x.parts.push_back(pt);
  }
  // neighbour-major layout
  std::vector<int> peers;
  for (const XPlan::Part &pt : x.parts) { const Ctx::Halo &h = G.halo[pt.kind]; peers.insert(peers.end(), h.sPE.begin(), h.sPE.end()); peers.insert(peers.end(), h.rPE.begin(), h.rPE.end()); }
  std::sort(peers.begin(), peers.end()); peers.erase(std::unique(peers.begin(), peers.end()), peers.end());
  std::vector<std::vector<long long>> sb(x.parts.size()), rb(x.parts.size());
  for (size_t k = 0; k < x.parts.size(); k++) { sb[k].assign(G.halo[x.parts[k].kind].sPE.size(), 0); rb[k].assign(G.halo[x.parts[k].kind].rPE.size(), 0); }
  size_t so = 0, ro = 0;
  for (int pe : peers) {
    XPlan::Msg mg{pe, so, 0, ro, 0};
    for (size_t k = 0; k < x.parts.size(); k++) {
      const Ctx::Halo &h = G.halo[x.parts[k].kind];
      for (size_t p = 0; p < h.sPE.size(); p++) if (h.sPE[p] == pe) { sb[k][p] = (long long)so; so += (size_t)(h.sptr[p + 1] - h.sptr[p]) * x.parts[k].Wtot; }
      for (size_t p = 0; p < h.rPE.size(); p++) if (h.rPE[p] == pe) { rb[k][p] = (long long)ro; ro += (size_t)(h.rptr[p + 1] - h.rptr[p]) * x.parts[k].Wtot; }
    }
    mg.scnt = so - mg.soff; mg.rcnt = ro - mg.roff;
    x.msgs.push_back(mg);
  }
  if (x.parts.size() > 1)
    for (size_t k = 0; k < x.parts.size(); k++) {
      x.parts[k].sbase_d = dev_upload(sb[k]); x.parts[k].rbase_d = dev_upload(rb[k]);
      if (g_alloc_failed) { G.err = "halo: device allocation failed (message layout)"; return 1; }
    }
  return 0;
}
__global__ void k_halo_pack(const double *__restrict__ f, int W, const int *__restrict__ list, const int *__restrict__ ptr, int npe, int nitems,
                            int Wtot, int Woff, double *__restrict__ buf, const long long *__restrict__ pbase);
__global__ void k_halo_unpack(double *__restrict__ f, int W, const int *__restrict__ list, const int *__restrict__ ptr, int npe, int nitems,
                              int Wtot, int Woff, const double *__restrict__ buf, const long long *__restrict__ pbase);
// merged: the neighbour-major layout (built-in transport)
int xplan_pack(const XPlan &x, int ch, bool merged) {
  if (halo_reserve(std::max(x.stot, x.rtot), ch)) return 1;
  hipStream_t st = ch ? G.cstream : G.stream;
  double *sbuf = ch ? G.hsend1 : G.hsend;
  for (const XPlan::Part &pt : x.parts) {
    const Ctx::Halo &h = G.halo[pt.kind];
    const long long *pb = merged ? pt.sbase_d : nullptr;
    int off = 0;
    for (const Sub &sb : pt.subs) {
      const long long tot = (long long)h.nsend * sb.W;
      if (tot > 0) hipLaunchKernelGGL(k_halo_pack, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, sb.p, sb.W, sb.q ? h.slist_q : h.slist, h.sptr_d, (int)h.sPE.size(), h.nsend, pt.Wtot, off,
                                      pb ? sbuf : sbuf + pt.soff, pb);
      off += sb.W;
    }
  }
  return 0;
}
int xplan_unpack(const XPlan &x, int ch, bool merged) {
  hipStream_t st = ch ? G.cstream : G.stream;
  const double *rbuf = ch ? G.hrecv1 : G.hrecv;
  for (const XPlan::Part &pt : x.parts) {
    const Ctx::Halo &h = G.halo[pt.kind];
    const long long *pb = merged ? pt.rbase_d : nullptr;
    int off = 0;
    for (const Sub &sb : pt.subs) {
      const long long tot = (long long)h.nrecv * sb.W;
      if (tot > 0) hipLaunchKernelGGL(k_halo_unpack, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, sb.p, sb.W, h.rlist, h.rptr_d, (int)h.rPE.size(), h.nrecv, pt.Wtot, off,
                                      pb ? rbuf : rbuf + pt.roff, pb);
      off += sb.W;
    }
  }
  return 0;
}
// the bytes of one exchange point: ONE group of ncclSend / ncclRecv through the built-in transport, one pair per neighbour (reference:
// one MPI_Isend / MPI_Irecv per neighbour and FIELD, src/gen_halo_exchange.F90:129-164), or one callback per part through the host's transport
int xplan_move(const XPlan &x, const fesom_transport *t, int ch, hipStream_t stream) {
  double *sbuf = ch ? G.hsend1 : G.hsend, *rbuf = ch ? G.hrecv1 : G.hrecv;
  if (t) {
    for (const XPlan::Part &pt : x.parts)
      if (t->exchange(t->ctx, pt.kind, sbuf + pt.soff, rbuf + pt.roff, pt.Wtot)) { G.err = "step_partitioned: transport exchange failed"; return 1; }
    return 0;
  }
  NCCLCHK(R.GroupStart());
  for (const XPlan::Msg &mg : x.msgs) if (mg.scnt) NCCLCHK(R.Send(sbuf + mg.soff, mg.scnt, ncclDouble, mg.peer, R.comm, stream));
  for (const XPlan::Msg &mg : x.msgs) if (mg.rcnt) NCCLCHK(R.Recv(rbuf + mg.roff, mg.rcnt, ncclDouble, mg.peer, R.comm, stream));
  NCCLCHK(R.GroupEnd());
  return 0;
}

typedef int (*FamFn)(const DM &, hipStream_t, const char *, int, int);
int fam_dyn(const DM &m, hipStream_t s, const char *n, int a, int fs) { return launch_named_dyn(m, s, n, a, fs); }
int fam_tra(const DM &m, hipStream_t s, const char *n, int a, int) { return launch_named_tra(m, s, n, a); }
int fam_kpp(const DM &m, hipStream_t s, const char *n, int, int) { return launch_named_kpp(m, s, n); }
int fam_gm(const DM &m, hipStream_t s, const char *n, int, int) { return launch_named_gm(m, s, n); }
int fam_toy(const DM &m, hipStream_t s, const char *n, int, int) { return launch_named_toy(m, s, n); }
int fam_sol(const DM &m, hipStream_t s, const char *n, int, int) { return launch_named_dsolve(m, s, n); }
int fam_ras(const DM &m, hipStream_t s, const char *n, int, int) { return launch_named_ras(m, s, n); }

struct PStep {
  const fesom_transport *t;                       // nullptr: the built-in RCCL transport (fesom_gpu_comm_init)
  int rc = 0;
  void k(FamFn f, const char *name, int arg = 0, int sub = 0) {
    if (rc) return;
    if (sub) {                                     // the same kernel over a list of columns (DM::sub_list)
      const Ctx::ColList &cl = sub == 1 ? G.sub_int : sub == 2 ? G.sub_cb : G.sub_cbh;
      if (cl.n == 0) return;
      DM ms = G.m;
      ms.sub_list = cl.d; ms.sub_n = cl.n;
      if (f(ms, G.stream, name, arg, G.first_step) != 0) { rc = 1; if (G.err.empty()) G.err = std::string("step_partitioned: unknown phase ") + name; }
      return;
    }
    if (f(G.m, G.stream, name, arg, G.first_step) != 0) { rc = 1; if (G.err.empty()) G.err = std::string("step_partitioned: unknown phase ") + name; }
  }
  // One exchange in flight on the communication stream (built-in transport only); any other communication waits for it first, so the
  // operations of the communicator stay in program order on every rank.
  bool &pending = G.x_pending;
  void Wt() {
    if (!pending) return;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (G.comm_timing) { hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, G.stream); }
    hipStreamWaitEvent(G.stream, G.ev_done, 0);
    if (e0) { hipEventRecord(e1, G.stream); G.comm_ev.push_back({e0, e1}); }       // what the step's stream still had to wait for
    pending = false;
  }
  void X(const XPlan &x, bool async) {
    if (rc) return;
    Wt();
    const bool as = async && !t && G.cstream;
    const int ch = as ? 1 : 0;
    hipStream_t st = as ? G.cstream : G.stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (as) { hipEventRecord(G.ev_prod, G.stream); hipStreamWaitEvent(G.cstream, G.ev_prod, 0); }
    else if (G.comm_timing) { hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, G.stream); }
    if (xplan_pack(x, ch, !t) || xplan_move(x, t, ch, st) || xplan_unpack(x, ch, !t)) { rc = 1; return; }
    if (as) { hipEventRecord(G.ev_done, G.cstream); pending = true; G.n_async++; }
    if (e0) { hipEventRecord(e1, G.stream); G.comm_ev.push_back({e0, e1}); }
    G.n_exch++; G.n_parts += (long long)x.parts.size();
  }
  void ARp(double *buf, int n) {                  // global sum over the ranks of n doubles at a device address, in place
    if (rc) return;
    Wt();
    if (t) { if (t->allreduce_sum(t->ctx, buf, n)) { rc = 1; G.err = "step_partitioned: transport allreduce failed"; } }
    else if (R.AllReduce(buf, buf, (size_t)n, ncclDouble, ncclSum, R.comm, G.stream) != ncclSuccess) { rc = 1; G.err = "step_partitioned: ncclAllReduce failed"; }
    G.n_allred++;
  }
  void AR(int n) { ARp(G.m.sv_red, n); }
  // compute_zonal_mean of the Soufflet channel on a partition (src/toy_channel_soufflet.F90:157-217): local sums, the two global sums
  // of the reference (plus, once, the one of compute_zonal_mean_ini :141-150 for the element counts), division
  void zonal() {
    k(fam_toy, "toy_zonal_sum");
    if (!G.toy_znum_global) { ARp((double *)G.m.toy_znum, 100); G.toy_znum_global = !rc; }
    ARp(G.m.toy_zvel, 100 * G.m.nlm1); ARp(G.m.toy_ztem, 100 * G.m.nlm1);
    k(fam_toy, "toy_zonal_div");
  }
};

// the program of one step
enum OpType { O_KERNEL, O_XCHG, O_XCHG_ASYNC, O_WAIT, O_SOLVE, O_ZONAL10 };
struct Op { OpType type; FamFn fam; const char *name; int arg; XPlan x; int sub = 0; };      // sub: 0 all columns, 1 interior nodes, 2 nodes with a halo neighbour, 3 those + the halo nodes
std::vector<Op> g_prog;
int g_last_part_its = 8;

int build_program() {
  const fesom_params &p = G.m.p;
  std::vector<Op> &P = g_prog;
  P.clear();
  int bad = 0;
  auto K = [&](FamFn f, const char *name, int arg = 0) { P.push_back(Op{O_KERNEL, f, name, arg, XPlan()}); };
  auto X = [&](std::vector<XSpec> spec, bool async = false) {
    Op o{async ? O_XCHG_ASYNC : O_XCHG, nullptr, nullptr, 0, XPlan()};
    bad |= xplan_build(spec, o.x);
    P.push_back(o);
  };
  auto W = [&]() { P.push_back(Op{O_WAIT, nullptr, nullptr, 0, XPlan()}); };
  const bool toy = p.toy_soufflet != 0, kpp = p.mix_scheme == 1, v5 = p.visc_option == 5;
  if (toy) P.push_back(Op{O_ZONAL10, nullptr, nullptr, 0, XPlan()});       // before_oce_step (oce_setup_step.F90:625-630)
  // head: everything that only needs the incoming state, so that Unode and (visc_option 5..7) U_b leave in ONE message, in flight under
  // pressure / PGF / slopes / the KPP column part
  K(fam_dyn, "k_vel_nodes");
  const bool ub_early = p.visc_option >= 4;                                // (options 1-3 read the Leith coefficient, formed further down)
  if (ub_early) { K(fam_dyn, "k_visc_elem"); X({{0, {"Unode"}}, {1, {"U_b"}}}, true); }
  else X({{0, {"Unode"}}}, true);
  K(fam_dyn, "k_pressure_bv"); K(fam_dyn, "k_pgf"); K(fam_dyn, "k_sigma_slope");
  if (p.use_momix) K(fam_dyn, "k_momix");
  if (kpp) K(fam_kpp, "k_kpp_col");
  W();
  std::vector<const char *> n2;                                              // second node message: whatever is ready by now
  if (p.mom_adv == 3) { K(fam_dyn, "k_vinv_ke"); n2.push_back("KE_node"); }
  else { K(fam_dyn, "k_momadv_node"); n2.push_back("Unode_rhs"); }
  if (v5 && ub_early) { K(fam_dyn, "k_visc_node"); n2.push_back("U_c"); }
  if (p.Redi) n2.push_back("slope_tapered");
  if (kpp) n2.push_back("kpp_blmc");
  X({{0, n2}});
  if (p.mix_scheme == 2) K(fam_dyn, "k_pp");          // (the shear of oce_mixing_PP is read at the three nodes of every owned element)
  if (kpp) {
    K(fam_kpp, "k_kpp_smooth1"); X({{0, {"kpp_sA"}}});
    K(fam_kpp, "k_kpp_smooth2"); X({{0, {"kpp_sB"}}});
    K(fam_kpp, "k_kpp_smooth3");
    K(fam_kpp, "k_kpp_final"); X({{0, {"kpp_viscA", "Kv"}}});
    K(fam_kpp, "k_kpp_elem");
  }
  if (p.mom_adv == 3) { K(fam_dyn, "k_leith_vort"); X({{0, {"vorticity"}}}); K(fam_dyn, "k_vinv_elem"); }
  else K(fam_dyn, "k_vel_rhs");
  if (p.visc_option <= 3) {    // h_viscosity_leith with its exchange_nod(vorticity), 2 x exchange_nod(aux), exchange_elem(Visc)
    K(fam_dyn, "k_leith_vort"); X({{0, {"vorticity"}}}); K(fam_dyn, "k_leith_elem");
    for (int nt = 0; nt < 2; nt++) { K(fam_dyn, "k_leith_node"); X({{0, {"leith_aux"}}}); K(fam_dyn, "k_leith_avg"); }
    X({{1, {"Visc"}}});
    if (p.visc_option != 1) { K(fam_dyn, "k_visc_elem"); X({{1, {"U_b"}}}); }
  }
  if (!v5) K(fam_dyn, "k_visc_apply");
  K(fam_dyn, "k_impl_visc");
  if (p.which_ale != 0) K(fam_dyn, "k_stiff_update");
  K(fam_dyn, "k_edge_transport"); K(fam_dyn, "k_ssh_rhs_node");
  P.push_back(Op{O_SOLVE, nullptr, nullptr, 0, XPlan()});
  X({{0, {"d_eta"}}});
  if (toy) K(fam_toy, "relax_zonal_vel");                    // oce_ale.F90:2696
  if (p.Redi && !p.Fer_GM) { K(fam_gm, "init_Redi_GM"); X({{0, {"Ki"}}}); }
  if (p.Fer_GM) {
    K(fam_gm, "init_Redi_GM");
    if (p.Redi) X({{0, {"fer_c", "fer_K", "Ki"}}}); else X({{0, {"fer_c", "fer_K"}}});
    K(fam_gm, "fer_solve_Gamma"); X({{0, {"fer_gamma"}}});
    K(fam_gm, "fer_gamma2vel"); K(fam_gm, "fer_wvel");     // (owned edges only touch elements this rank computes itself: the halos can wait)
  }
  K(fam_dyn, "k_update_vel"); K(fam_dyn, "k_edge_transport1"); K(fam_dyn, "k_vert_vel_hbar");
  {   // ONE message for everything the dynamics leave behind: element velocities and the node fields of vert_vel_ale / compute_hbar_ale
    std::vector<const char *> nn = {"Wvel", "Wvel_e", "Wvel_i", "hnode_new", "hbar", "hbar_old", "eta_n", "ssh_rhs_old"}, ee = {"UV"};
    if (p.Fer_GM) { nn.push_back("fer_Wvel"); ee.push_back("fer_UV"); }
    X({{0, nn}, {1, ee}});
  }
  K(fam_dyn, "k_dhe");
  if (p.Fer_GM) K(fam_gm, "bolus_add");
  if (p.SPP) K(fam_tra, "k_spp", 0);                         // solve_tracers_ale :120-121 (owned and halo nodes, as the reference)
  K(fam_tra, "k_tr_ab", 0); K(fam_tra, "k_tr_grad_elem", 0); X({{2, {"tr_xy_ab"}}}, true);       // the reference's own overlap (oce_tracer_mod.F90:68-81)
  K(fam_tra, "k_tr_z", 0);
  W();
  K(fam_tra, "k_updn_grad", 0);
  K(fam_tra, "k_flux_hor", 0); K(fam_tra, "k_fct_lo_node", 0);
  auto KS = [&](FamFn f, const char *name, int sub) { Op o{O_KERNEL, f, name, 0, XPlan()}; o.sub = sub; P.push_back(o); };
  if (!p.tra_adv_lim) {                                        // (no low-order solution, no limiter with tra_adv_lim='NON')
    // INTERIOR / BOUNDARY SPLIT (the reference's mechanism: src/gen_halo_exchange.F90:129-164 posts, :317-363 waits; its one use: src/oce_tracer_mod.F90:68-81):
    // the low-order solution travels on the communication stream while the limiter works on the nodes whose neighbours are all owned; the
    // nodes next to the halo follow once it has arrived.  The same for the limiting factors and the tracer update.
    if (p.with_diffusion && !p.Redi) K(fam_tra, "k_diff_flux", 0);
    if (p.Redi) X({{0, {"fct_LO", "tr_z"}}}, true); else X({{0, {"fct_LO"}}}, true);
    KS(fam_tra, "k_fct_node", 1);
    W();
    if (p.with_diffusion && p.Redi) K(fam_tra, "k_diff_flux", 0);     // (with Redi it reads the halo of tr_z)
    KS(fam_tra, "k_fct_node", 2);
    X({{0, {"fct_plus", "fct_minus"}}}, true);
    KS(fam_tra, "k_tr_update", 1);
    W();
    K(fam_tra, "k_fct_edge_limit", 0);                          // (reads the factors at both nodes of every owned edge, halo nodes included)
    KS(fam_tra, "k_tr_update", 3);
  } else {
    if (p.Redi) X({{0, {"tr_z"}}});
    if (p.with_diffusion) K(fam_tra, "k_diff_flux", 0);
    K(fam_tra, "k_fct_edge_limit", 0); K(fam_tra, "k_tr_update", 0);
  }
  if (p.smooth_bh_tra) { K(fam_tra, "k_bh1", 0); X({{0, {"bh_tmp"}}}); K(fam_tra, "k_bh2", 0); }     // (the tracer halo still holds the values of the previous exchange, as in the reference)
  if (toy) for (int tr = 0; tr < G.m.ntr; tr++) K(fam_toy, "relax_zonal_temp");     // once per tracer of the loop, always on tracer 1 (oce_ale_tracer.F90:150)
  else if (p.clim_relax > 1.0e-8) K(fam_tra, "relax_to_clim", 0);
  X({{0, {"tr_arr"}}}, true);                                 // in flight under the thickness update and the head of the next step
  if (p.Fer_GM) K(fam_gm, "bolus_remove");
  K(fam_dyn, "k_thick_node"); K(fam_dyn, "k_thick_elem");
  if (bad) return 1;
  g_prog_ready = true;
  return 0;
}

// Every rank must take the same solver path (their exchanges and all-reduces have to match): a rank whose block does not qualify for the
// RAS preconditioner takes it from all of them.  One global sum + one read-back, at the first partitioned step.
int agree_on_preconditioner(PStep &S) {
  if (G.precond_agreed) return 0;
  const double mine = G.m.rs_pinfo ? 1.0 : 0.0;
  HIPCHK(hipMemcpyAsync(G.m.sv_red, &mine, sizeof(double), hipMemcpyHostToDevice, G.stream));
  S.AR(1);
  if (S.rc) return 1;
  double all = 0.0;
  HIPCHK(hipMemcpyAsync(&all, G.m.sv_red, sizeof(double), hipMemcpyDeviceToHost, G.stream));
  HIPCHK(hipStreamSynchronize(G.stream));
  if (all != (double)G.npes) G.m.rs_pinfo = nullptr;
  G.precond_agreed = true;
  return 0;
}

// Partitioned SSH solve: BiCGstab over the owned rows, halo of the gathered vector before each product with A_s, global sum of the
// partial dot products after it (pARMS does the same with MPI, lib/parms/src/bicgstab_ras.c:49-259, parms_comm.c:205-356).  Krylov scalars
// and the convergence flag live on the device; iterations are enqueued in chunks and the flag is read back once per solve as a rule (one
// iteration more than the last solve needed first; phases behind the convergence are no-ops, so the result does not depend on the chunks).
int part_solve(PStep &S) {
  const DM &m = G.m;
  static XPlan x_x, x_ph, x_sh, x_dinv, x_s;
  static int key = -1;                                   // the context (fesom_gpu_init call) the cached message plans belong to
  if (key != G.generation) {
    if (xplan_build({{0, {"sv_x"}}}, x_x) || xplan_build({{0, {"sv_ph"}}}, x_ph) || xplan_build({{0, {"sv_sh"}}}, x_sh) || xplan_build({{0, {"sv_dinv"}}}, x_dinv) ||
        xplan_build({{0, {"sv_s"}}}, x_s)) return 1;
    key = G.generation;
  }
  const int maxits = m.sv_maxits > 0 ? m.sv_maxits : 2000;
  static double *hk = nullptr;
  if (!hk && hipHostMalloc((void **)&hk, 16 * sizeof(double)) != hipSuccess) { G.err = "step_partitioned: pinned allocation failed"; return 1; }
  const bool ras = m.rs_pinfo != nullptr;
  S.k(fam_sol, "ds_scale");
  if (ras) {
    S.k(fam_ras, "dsr_setup"); S.X(x_x, false);
    S.k(fam_ras, "dsr_init"); S.AR(1); S.k(fam_ras, "dsr_scal_init");
  } else {
    S.X(x_dinv, false);
    S.k(fam_sol, "ds_setup"); S.X(x_s, false);
    S.k(fam_sol, "ds_init"); S.AR(1); S.k(fam_sol, "ds_scal_init"); S.k(fam_sol, "ds_p");
  }
  int total = 0, chunk = std::max(1, g_last_part_its + 1);
  const int flag = ras ? 8 : 7;
  for (;;) {
    for (int i = 0; i < chunk && !S.rc; i++) {
      if (ras) {
        S.k(fam_ras, "dsr_prec0"); S.X(x_ph, false); S.k(fam_ras, "dsr_spmv1"); S.AR(1); S.k(fam_ras, "dsr_scal_alpha");
        S.k(fam_ras, "dsr_prec1"); S.X(x_sh, false); S.k(fam_ras, "dsr_spmv2"); S.AR(4); S.k(fam_ras, "dsr_scal_omega"); S.k(fam_ras, "dsr_update");
      } else {
        S.X(x_ph, false); S.k(fam_sol, "ds_spmv1"); S.AR(1); S.k(fam_sol, "ds_scal_alpha"); S.k(fam_sol, "ds_s");
        S.X(x_s, false); S.k(fam_sol, "ds_spmv2"); S.AR(4); S.k(fam_sol, "ds_scal_omega"); S.k(fam_sol, "ds_update"); S.k(fam_sol, "ds_p");
      }
    }
    if (S.rc) return 1;
    total += chunk;
    HIPCHK(hipMemcpyAsync(hk, m.sv_kry, 16 * sizeof(double), hipMemcpyDeviceToHost, G.stream));
    HIPCHK(hipStreamSynchronize(G.stream));
    if (hk[flag] != 0.0 || hk[6] >= (double)maxits) break;
    chunk = 2;
  }
  (void)total;
  G.part_iters = (int)hk[6]; g_last_part_its = G.part_iters;
  S.k(ras ? fam_ras : fam_sol, ras ? "dsr_finish" : "ds_finish");
  const double tol = m.sv_tol > 0.0 ? m.sv_tol : 1e-10;
  if (!(hk[5] < tol * tol)) {          // the reference's BiCGstab reports this too (bicgstab_ras.c:237); a step must not go on with an unconverged SSH
    char b[200];
    snprintf(b, sizeof b, "step_partitioned: the SSH solve did not converge: %d iterations, ||scaled residual|| = %.3e (tolerance %.1e)", G.part_iters, sqrt(hk[5] > 0.0 ? hk[5] : 0.0), tol);
    G.err = b; fprintf(stderr, "fesom_gpu: %s\n", b);
    return 1;
  }
  return 0;
}
}  // namespace

int fesom_gpu_step_partitioned(int n, const fesom_transport *t) {
  NEED_READY();
  if (G.npes < 2) return fesom_gpu_step(n);
  if (t && (!t->exchange || !t->allreduce_sum)) { G.err = "step_partitioned: transport callbacks missing"; return 1; }
  if (!t && (!R.comm || R.nranks != G.npes || R.rank != G.mype)) {
    G.err = "step_partitioned: no transport given and the built-in RCCL transport is not initialised for this partition (fesom_gpu_comm_init)"; return 1;
  }
  PStep S{t};
  if (agree_on_preconditioner(S)) return 1;
  if (!g_prog_ready && build_program()) return 1;
  for (const Op &o : g_prog) {
    if (S.rc) break;
    switch (o.type) {
      case O_KERNEL: S.k(o.fam, o.name, o.arg, o.sub); break;
      case O_XCHG: S.X(o.x, false); break;
      case O_XCHG_ASYNC: S.X(o.x, true); break;
      case O_WAIT: S.Wt(); break;
      case O_SOLVE: if (part_solve(S)) S.rc = 1; break;
      case O_ZONAL10: if (n % 10 == 0) S.zonal(); break;
    }
  }
  S.Wt();
  G.first_step = 0;
  HIPCHK(hipGetLastError());
  return S.rc;
}

// zonal means of a partitioned Soufflet channel outside a step (the set-up calls compute_zonal_mean once before the first step,
// toy_channel_soufflet.F90:343); t == NULL: the built-in transport
int fesom_gpu_toy_zonal_mean(const fesom_transport *t) {
  NEED_READY();
  if (!G.m.p.toy_soufflet) { G.err = "toy_zonal_mean: not a Soufflet channel run"; return 1; }
  if (G.npes < 2) return call_named("compute_zonal_mean", 0);
  if (t && !t->allreduce_sum) { G.err = "toy_zonal_mean: transport callbacks missing"; return 1; }
  if (!t && (!R.comm || R.nranks != G.npes)) { G.err = "toy_zonal_mean: the built-in transport is not initialised"; return 1; }
  PStep S{t};
  S.zonal();
  HIPCHK(hipGetLastError());
  return S.rc;
}

int fesom_gpu_profile_step(int n, double ms[7]) {
  NEED_READY();
  if (G.npes > 1 || G.m.p.toy_soufflet) { G.err = "profile_step: single partition without the toy hooks only"; return 1; }
  (void)n;
  const fesom_params &p = G.m.p;
  hipEvent_t t[11];
  for (auto &e : t) HIPCHK(hipEventCreate(&e));
  int bad = 0;
  auto c = [&](const char *name, int arg = 0) { bad |= call_named(name, arg); };
  auto T = [&](int i) { hipEventRecord(t[i], G.stream); };
  c("compute_vel_nodes");                                                   // fvom_main.F90:216, outside the reference's t0..t10
  T(0);
  c("pressure_bv"); c("pressure_force"); c("sw_alpha_beta"); c("compute_sigma_xy"); c("compute_neutral_slope");
  if (p.mix_scheme == 2) { c("mixing_pp"); c("mo_convect"); }
  if (p.mix_scheme == 1) { c("mixing_kpp"); c("mo_convect"); }
  T(1);
  c("compute_vel_rhs"); c("viscosity_filter"); if (p.i_vert_visc) c("impl_vert_visc_ale");
  T(2);
  if (p.which_ale != 0) c("update_stiff_mat_ale");
  c("compute_ssh_rhs_ale");
  T(10);                                                                    // t30
  c("solve_ssh");
  T(3);
  c("update_vel");
  T(4);
  c("compute_hbar_ale"); c("eta_update");
  T(5);
  if (p.Fer_GM || p.Redi) c("init_Redi_GM");
  if (p.Fer_GM) { c("fer_solve_Gamma"); c("fer_gamma2vel"); }
  T(6);
  c("vert_vel_ale"); if (p.Fer_GM) c("fer_wvel");
  T(7);
  if (p.Fer_GM) c("bolus_add");
  for (int tr = 1; tr <= G.m.ntr; tr++) { c("init_tracers_AB", tr); c("adv_tracers_ale", tr); c("diff_tracers_ale", tr); }
  if (p.Fer_GM) c("bolus_remove");
  c("salinity_clamp");
  T(8);
  c("update_thickness_ale");
  T(9);
  HIPCHK(hipEventSynchronize(t[9]));
  auto dtm = [&](int a, int b) { float x = 0; hipEventElapsedTime(&x, t[a], t[b]); return (double)x; };
  ms[0] = dtm(0, 1);                                  // rtime_oce_mixpres
  ms[1] = dtm(1, 2) + dtm(6, 7) + dtm(3, 4);          // rtime_oce_dyn
  ms[2] = dtm(2, 3) + dtm(4, 5);                      // rtime_oce_dynssh (includes the solve, as in the reference)
  ms[3] = dtm(10, 3);                                 // rtime_oce_solvessh
  ms[4] = dtm(5, 6);                                  // rtime_oce_GMRedi
  ms[5] = dtm(7, 8);                                  // rtime_oce_solvetra
  ms[6] = dtm(0, 9);                                  // rtime_oce
  for (auto &e : t) hipEventDestroy(e);
  if (bad) { if (G.err.empty()) G.err = "profile_step: a routine of the chain is unknown"; return 1; }
  return 0;
}

int fesom_gpu_step_info(fesom_step_info *out) {
  NEED_READY();
  static_assert(sizeof(fesom_step_info) == 42 * sizeof(double), "fesom_step_info = 42 doubles");
  if (!G.mon_col) { G.mon_col = dev_alloc<double>((size_t)10 * G.m.N); G.mon_out = dev_alloc<double>(64); }
  launch_step_info(G.m, G.stream, G.mon_col, G.mon_out);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, G.mon_out, sizeof(fesom_step_info), hipMemcpyDeviceToHost, G.stream));
  HIPCHK(hipStreamSynchronize(G.stream));
  return check_ale_flag();
}

int fesom_gpu_get_field(const char *name, double *out, long long count) {
  NEED_CTX();
  auto it = G.fields.find(name);
  if (it == G.fields.end() || (size_t)count != it->second.count) { G.err = std::string("get_field: bad name/count ") + name; return 1; }
  HIPCHK(hipDeviceSynchronize());
  const double *src = (const double *)it->second.p + (it->second.slabs > 1 ? (size_t)G.cur_tr * it->second.count : 0);   // slab of the tracer last worked on
  HIPCHK(hipMemcpy(out, src, sizeof(double) * count, hipMemcpyDeviceToHost));
  return 0;
}
int fesom_gpu_set_field(const char *name, const double *in, long long count) {
  NEED_CTX();
  auto it = G.fields.find(name);
  if (it == G.fields.end() || (size_t)count != it->second.count) { G.err = std::string("set_field: bad name/count ") + name; return 1; }
  HIPCHK(hipDeviceSynchronize());
  double *dst = (double *)it->second.p + (it->second.slabs > 1 ? (size_t)G.cur_tr * it->second.count : 0);
  HIPCHK(hipMemcpy(dst, in, sizeof(double) * count, hipMemcpyHostToDevice));
  return 0;
}

static int call_named(const char *name, int arg) {
  const DM &m = G.m;
  if (!strcmp(name, "first_step")) { G.first_step = arg; return 0; }
  if (!strcmp(name, "solve_ssh") || !strcmp(name, "k_solver")) return launch_solver(m, G.stream);
  if (!strcmp(name, "step")) { enqueue_step(G.stream, G.first_step, arg); G.first_step = 0; return 0; }
  if (!strcmp(name, "solver_snapshot")) {       // keep the pre-solve iterate so that the solve can be replayed for timing
    return hipMemcpyAsync(m.sv_snap, m.d_eta, sizeof(double) * m.N, hipMemcpyDeviceToDevice, G.stream) != hipSuccess;
  }
  if (!strcmp(name, "k_solver_replay")) {
    if (hipMemcpyAsync(m.d_eta, m.sv_snap, sizeof(double) * m.N, hipMemcpyDeviceToDevice, G.stream) != hipSuccess) return 1;
    const int solves = m.sv_solves;                // a replay for timing is not a solve of the run (the default iteration schedule counts them)
    const int rc = launch_solver(m, G.stream);
    G.m.sv_solves = solves;
    return rc;
  }
  int fs = G.first_step;
  if (!strcmp(name, "init_tracers_AB") || !strcmp(name, "adv_tracers_ale") || !strcmp(name, "diff_tracers_ale") || !strncmp(name, "k_t", 3) ||
      !strncmp(name, "k_f", 3) || !strcmp(name, "k_updn_grad") || !strcmp(name, "k_diff_flux"))
    G.cur_tr = (arg >= 1 && arg <= m.ntr) ? arg - 1 : 0;
  int rc = launch_named_dyn(m, G.stream, name, arg, fs);
  if (rc == 0) { if (!strcmp(name, "compute_vel_rhs")) G.first_step = 0; return 0; }
  rc = launch_named_tra(m, G.stream, name, arg);
  if (rc == 0) return 0;
  rc = launch_named_toy(m, G.stream, name);
  if (rc == 0) return 0;
  rc = launch_named_kpp(m, G.stream, name);
  if (rc == 0) return 0;
  rc = launch_named_gm(m, G.stream, name);
  if (rc == 0) return 0;
  rc = launch_named_dsolve(m, G.stream, name);
  if (rc == 0) return 0;
  rc = launch_named_ras(m, G.stream, name);
  if (rc == 0) return 0;
  G.err = std::string("fesom_gpu_call: unknown routine ") + name;
  return 1;
}
int fesom_gpu_call(const char *routine, int arg) {
  NEED_READY();
  int rc = call_named(routine, arg);
  if (rc) return rc;
  HIPCHK(hipGetLastError());
  return 0;
}

int fesom_gpu_run_steps(int n_first, int nsteps) {
  NEED_READY();
  static const bool timing = getenv("FESOM_GPU_TIMING") != nullptr;
  auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < nsteps; k++) {
    int which = G.first_step ? 1 : 0;
    const int n = n_first + k;
    // graph replay only where the step is a fixed launch sequence: the toy hooks depend on the step number, the multi-workgroup SSH
    // solve reads its convergence flag back (a host synchronisation is illegal inside a capture)
    const bool solver_syncs = !G.m.sv_minv && (G.m.myN > 4096 || G.m.ssh_maxnnz > 10);
    if (G.use_graph && !G.m.p.toy_soufflet && !solver_syncs) {
      if (!G.graph[which] && build_graph(which)) return 1;
      HIPCHK(hipGraphLaunch(G.graph[which], G.stream));
    } else if (!G.serial && !G.m.p.SPP) {               // (SPP changes the salinity at the head of solve_tracers_ale: the tracer preparation cannot be hoisted
      static Dag dag;                                   //  to the start of the step as the DAG does -> the serial order)  pooled events
      dag.reset();
      enqueue_step_dag(G.stream, which, dag, n);
    } else enqueue_step(G.stream, which, n);
    G.first_step = 0;
  }
  if (!g_bad_launch.empty()) { G.err = "fesom_gpu_run_steps: the step names a kernel no launcher knows: " + g_bad_launch; fprintf(stderr, "fesom_gpu: %s\n", G.err.c_str()); return 1; }
  if (timing) {
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    fprintf(stderr, "[fesom_gpu] host enqueue: %d steps in %.1f us (%.1f us/step)\n", nsteps, us, us / (nsteps ? nsteps : 1));
  }
  HIPCHK(hipGetLastError());
  return 0;
}
int fesom_gpu_step(int n) {          // asynchronous: the step is enqueued on the library's stream (graph replay); download_state,
  return fesom_gpu_run_steps(n, 1);  // get_field, step_info and fesom_gpu_sync synchronise
}

int fesom_gpu_last_solver_iterations(void) {
  if (!G.ready) return -1;
  if (G.npes > 1 && G.part_iters >= 0) return G.part_iters;      // partitioned solve (fesom_gpu_step_partitioned)
  int it = -1;
  hipStreamSynchronize(G.stream);
  hipMemcpy(&it, G.m.sv_info, sizeof(int), hipMemcpyDeviceToHost);
  return it;
}
int fesom_gpu_tile_shape(void) { return G.ready ? G.m.use_tile : -1; }
int fesom_gpu_solver_kind(void) { return !G.ready ? -1 : (G.m.sv_minv ? 1 : G.m.rs_pinfo ? 2 : 0); }
// solves of this run that the explicit-inverse iterations did not finish (the Jacobi safety net took over); synchronises
int fesom_gpu_solver_safety_net_count(void) {
  if (!G.ready) return -1;
  int v[4] = {0, 0, 0, 0};
  hipStreamSynchronize(G.stream);
  hipMemcpy(v, G.m.sv_info, sizeof(v), hipMemcpyDeviceToHost);
  return v[2];
}
double fesom_gpu_last_solver_residual(void) {
  if (!G.ready) return -1.0;
  double r = -1.0;
  hipStreamSynchronize(G.stream);
  hipMemcpy(&r, G.m.sv_resid, sizeof(double), hipMemcpyDeviceToHost);
  return r;
}

// Average device time of one launch of a routine / kernel (or "step"): nrep launches are captured into one
// hipGraph (so the host launch rate does not bound short kernels) and timed with HIP events on the library's
// own stream.  The figure includes the ~1.5 us dependent-launch boundary of back-to-back kernels.
int fesom_gpu_kernel_time_ms(const char *group_in, int nrep, double *ms_per_launch) {
  NEED_READY();
  // "<name>:all" = the per-tracer kernels as the step launches them (all tracers in one launch, grid.y)
  std::string gname(group_in);
  int targ = 1;
  if (gname.size() > 4 && gname.compare(gname.size() - 4, 4, ":all") == 0) { gname.resize(gname.size() - 4); targ = 0; }
  const char *group = gname.c_str();
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  int fs = G.first_step;
  G.first_step = 0;
  hipGraph_t g; hipGraphExec_t ge;
  HIPCHK(hipStreamSynchronize(G.stream));
  if ((G.m.myN > 4096 || G.m.ssh_maxnnz > 10) && (!strcmp(group, "k_solver_replay") || !strcmp(group, "solve_ssh") || !strcmp(group, "k_solver") || !strcmp(group, "step"))) {
    // the multi-workgroup SSH solve reads its convergence flag back between chunks: no stream capture, plain events
    HIPCHK(hipEventRecord(e0, G.stream));
    for (int i = 0; i < nrep; i++) if (call_named(group, 1)) return 1;
    HIPCHK(hipEventRecord(e1, G.stream));
    HIPCHK(hipEventSynchronize(e1));
    float ms0 = 0;
    HIPCHK(hipEventElapsedTime(&ms0, e0, e1));
    *ms_per_launch = ms0 / nrep;
    hipEventDestroy(e0); hipEventDestroy(e1);
    G.first_step = fs;
    return 0;
  }
  HIPCHK(hipStreamBeginCapture(G.stream, hipStreamCaptureModeGlobal));
  int bad = 0;
  for (int i = 0; i < nrep; i++) bad |= call_named(group, targ);
  HIPCHK(hipStreamEndCapture(G.stream, &g));
  if (bad) { hipGraphDestroy(g); return 1; }
  HIPCHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  HIPCHK(hipGraphLaunch(ge, G.stream));        // warm-up
  HIPCHK(hipStreamSynchronize(G.stream));
  HIPCHK(hipEventRecord(e0, G.stream));
  HIPCHK(hipGraphLaunch(ge, G.stream));
  HIPCHK(hipEventRecord(e1, G.stream));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  *ms_per_launch = ms / nrep;
  hipEventDestroy(e0); hipEventDestroy(e1); hipGraphExecDestroy(ge); hipGraphDestroy(g);
  G.first_step = fs;
  return 0;
}

// ---- psolver_init / psolve / psolver_final with the reference's signatures (src/psolve.c:16,117,152) -----
// The reference's functions are void and report through stderr + exit (psolve.c:203 "ERROR: matrix data is static"; pARMS aborts
// via MPI_Abort): a caller linked against this library cannot see a status either, so every violated precondition or failed HIP
// call prints one line and terminates the process with a non-zero status instead of returning an unsolved `sol`.
static struct { bool ok = false; int n = 0, nza = 0; DM m; std::vector<void *> al; } PS;
[[noreturn]] static void ps_die(const std::string &msg) {
  fprintf(stderr, "fesom_gpu psolver: %s\n", msg.c_str());
  fflush(stderr);
  exit(3);
}
#define PSCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) ps_die(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)
// (bodies as internal functions: the exported names below may be interposed by a host adapter that defines psolver_init / psolve itself and forwards)
static void ps_final_impl(void) { for (void *p : PS.al) hipFree(p); PS.al.clear(); PS.ok = false; }
static void ps_init_impl(int *id, int *stype, int *pctype, int *pcilutype, int *ilulevel, int *fillin, double *droptol, int *maxits,
                         int *restart, double *soltol, int *part, int *rptr, int *cols, double *vals, int *reuse, int *fcomm) {
  // solver / preconditioner selectors of pARMS (SOLBICGS_RAS, PCILUK, fill level ...) have one answer here: BiCGstab with this
  // library's frozen preconditioner; `reuse` (keep the factors of the first matrix) is how the preconditioner is always used
  (void)id; (void)stype; (void)pctype; (void)pcilutype; (void)ilulevel; (void)fillin; (void)droptol; (void)restart; (void)reuse; (void)fcomm;
  ps_final_impl();
  // one GPU = one partition.  The row partition is part[0..npes]; the rank count is not an argument (psolve.c:31-33 asks MPI), so
  // what can be checked is that this rank's block starts at row 0 and that every column lies inside it -- a block of a
  // multi-rank partition has off-block columns or a non-zero offset and is refused (use fesom_gpu_step_partitioned there).
  if (part[0] != 0) ps_die("psolver_init: part[0] != 0 -- the rows of this rank are a block of a multi-rank partition; this entry point solves single-partition systems only");
  const int n = part[1] - part[0];
  if (n < 1) ps_die("psolver_init: no rows");
  if (rptr[0] != 0) ps_die("psolver_init: rptr[0] must be 0 (0-based CSR, src/oce_ale.F90:2302-2304)");
  const int nza = rptr[n];
  int maxnnz = 0;
  for (int i = 0; i < n; i++) {
    if (rptr[i + 1] < rptr[i]) ps_die("psolver_init: rptr is not non-decreasing");
    maxnnz = std::max(maxnnz, rptr[i + 1] - rptr[i]);
  }
  if (maxnnz > 16) ps_die("psolver_init: more than 16 entries in a row (the SSH operator of a triangular mesh has <= ~10)");
  for (int j = 0; j < nza; j++)
    if (cols[j] < 0 || cols[j] >= n) ps_die("psolver_init: column index outside [0, n) -- rows of a multi-rank partition (global columns) are not supported here");
  for (int i = 0; i < n; i++)
    if (rptr[i + 1] == rptr[i] || cols[rptr[i]] != i) ps_die("psolver_init: the first entry of every row must be the diagonal (src/oce_ale.F90:1128-1151)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) ps_die("no HIP device: the MI355X path has no CPU fallback");
  memset(&PS.m, 0, sizeof(PS.m));
  auto A = [&](size_t bytes) { void *p = nullptr; PSCHK(hipMalloc(&p, bytes ? bytes : 8)); PSCHK(hipMemset(p, 0, bytes ? bytes : 8)); PS.al.push_back(p); return p; };
  DM &m = PS.m;
  m.myN = m.N = n; m.nza = nza;
  m.sv_tol = (soltol && *soltol > 0.0) ? *soltol : 0.0;            // absolute tolerance on the row-scaled residual (psolve.c:97, bicgstab_ras.c:78)
  m.sv_maxits = (maxits && *maxits > 0) ? *maxits : 0;
  int *rp = (int *)A(sizeof(int) * (n + 1)), *ci = (int *)A(sizeof(int) * nza);
  PSCHK(hipMemcpy(rp, rptr, sizeof(int) * (n + 1), hipMemcpyHostToDevice));
  PSCHK(hipMemcpy(ci, cols, sizeof(int) * nza, hipMemcpyHostToDevice));
  m.rowptr = rp; m.colind = ci;
  m.ssh_maxnnz = maxnnz;
  m.ssh_values = (double *)A(sizeof(double) * nza);
  PSCHK(hipMemcpy(m.ssh_values, vals, sizeof(double) * nza, hipMemcpyHostToDevice));
  m.sv_vals = (double *)A(sizeof(double) * 16 * (n + 64));
  double **vecs[] = {&m.sv_scale, &m.sv_dinv, &m.sv_b, &m.sv_r, &m.sv_r0, &m.sv_p, &m.sv_v, &m.sv_s, &m.sv_t, &m.sv_ph, &m.d_eta, &m.ssh_rhs,
                     &m.sv_bn, &m.sv_x, &m.sv_pd, &m.sv_sn, &m.sv_sh};
  for (auto v : vecs) *v = (double *)A(sizeof(double) * (n + 64));
  m.sv_x0 = (double *)A(sizeof(double) * 16 * (n + 64));
  m.sv_info = (int *)A(16); m.sv_resid = (double *)A(8);
  {
    std::vector<int> perm, inv, wid;
    solver_row_order(rptr, n, m.ssh_maxnnz, n <= 4096 && m.ssh_maxnnz <= 10, perm, inv, wid);
    for (auto pv : {std::make_pair(&perm, &m.sv_perm), std::make_pair(&inv, &m.sv_inv), std::make_pair(&wid, &m.sv_wid)}) {
      int *dp = (int *)A(pv.first->size() * sizeof(int));
      PSCHK(hipMemcpy(dp, pv.first->data(), pv.first->size() * sizeof(int), hipMemcpyHostToDevice));
      *pv.second = dp;
    }
    std::vector<unsigned short> ec = ell_cols(rptr, cols, n, m.ssh_maxnnz, perm, inv);
    m.sv_cols = (unsigned short *)A(ec.size() * sizeof(unsigned short));
    PSCHK(hipMemcpy(m.sv_cols, ec.data(), ec.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
    const int W = m.ssh_maxnnz <= 10 ? 10 : 16, NP = (n + 63) / 64 * 64;      // 32-bit pattern + work space of the multi-workgroup phases
    std::vector<int> c32((size_t)W * NP, 0);
    for (int i = 0; i < NP; i++)
      for (int k = 0; k < W; k++) c32[(size_t)k * NP + i] = (i < n && rptr[i] + k < rptr[i + 1]) ? cols[rptr[i] + k] : (i < n ? i : 0);
    int *c32d = (int *)A(c32.size() * sizeof(int));
    PSCHK(hipMemcpy(c32d, c32.data(), c32.size() * sizeof(int), hipMemcpyHostToDevice));
    m.sv_colsi = c32d;
    m.sv_part = (double *)A(sizeof(double) * 8 * ((n + 255) / 256 + 1)); m.sv_red = (double *)A(64); m.sv_kry = (double *)A(48 * sizeof(double));
  }
  // frozen preconditioner from the matrix of this call (the reference computes its ILU factors from the first matrix, psolve.c:117-150):
  // the explicit inverse where it fits, FESOM_GPU_PRECOND=jacobi opts out
  const char *pc = getenv("FESOM_GPU_PRECOND");
  if (!(pc && !strcmp(pc, "jacobi")) && n <= 4096 && n >= 64 && maxnnz <= 10) {
    if (xinv_device(n, rptr, cols, vals, nullptr, m, PS.al)) ps_die("psolver_init: the explicit-inverse preconditioner could not be built (singular matrix or out of device memory)");
  } else if (!(pc && !strcmp(pc, "jacobi")) && n > 4096) {
    m.rs_inv = nullptr;
    if (ras_device(n, n, rptr, cols, vals, maxnnz, m, PS.al)) m.rs_pinfo = nullptr;               // (does not qualify: Jacobi)
  }
  solver_prepare();
  PS.n = n; PS.nza = nza; PS.ok = true;
}
// ---- distributed variant: the rows of a multi-rank partition (what psolver_init receives from solve_ssh_ale with npes > 1: `part` = prefix of
// the owned-row counts, `cols` = global contiguous numbering, src/oce_ale.F90:1298-1344, src/psolve.c:16-115).  The library holds no MPI: the
// host adapter (fesom2_amd/fortran/fesom_gpu_psolve_mpi.c, compiled by the integrator with the application's mpi.h) works out the halo of the
// row block with MPI, hands it over here together with a transport (MPI callbacks, or NULL = the built-in RCCL transport after
// fesom_gpu_comm_init) and forwards psolve.  The solve is the partitioned BiCGstab of fesom_gpu_step_partitioned (part_solve: RAS-Chebyshev
// preconditioner on the owned block, halo of the gathered vector before each product, global sums of the partial dot products), in a context
// that holds the solver alone.
//   rglob[0 .. nrecv)  global rows of the halo in receive order (grouped by rPE, counts rcnt), sloc[0 .. nsend) owned rows (0-based, local)
//   to send, grouped by sPE with counts scnt.
static struct { bool ok = false; int n = 0, nza = 0; const fesom_transport *t = nullptr; fesom_transport tcopy; } PSD;
int fesom_gpu_psolver_init_dist(int npes, int mype, const int *part, const int *rptr, const int *cols, const double *vals, int maxits, double soltol,
                                int nr, const int *rPE, const int *rcnt, const int *rglob, int ns, const int *sPE, const int *scnt, const int *sloc,
                                const fesom_transport *t) {
  if (G.ready) fesom_gpu_finalize();
  ps_final_impl();
  PSD.ok = false;
  G.err.clear(); g_alloc_failed = false;
  if (npes < 2 || mype < 0 || mype >= npes) { G.err = "psolver_init_dist: needs npes >= 2 and 0 <= mype < npes (one partition: psolver_init)"; return 1; }
  if (t && (!t->exchange || !t->allreduce_sum)) { G.err = "psolver_init_dist: transport callbacks missing"; return 1; }
  const int n = part[mype + 1] - part[mype], g0 = part[mype];
  if (n < 1 || rptr[0] != 0) { G.err = "psolver_init_dist: no rows, or rptr[0] != 0"; return 1; }
  const int nza = rptr[n];
  int nrecv = 0, nsend = 0;
  for (int p = 0; p < nr; p++) nrecv += rcnt[p];
  for (int p = 0; p < ns; p++) nsend += scnt[p];
  std::map<int, int> hpos;                           // global row of a halo entry -> local column
  for (int k = 0; k < nrecv; k++) hpos[rglob[k]] = n + k;
  std::vector<int> rp(rptr, rptr + n + 1), ci(nza);
  int maxnnz = 0;
  for (int i = 0; i < n; i++) {
    if (rp[i + 1] <= rp[i]) { G.err = "psolver_init_dist: empty row or rptr not increasing"; return 1; }
    maxnnz = std::max(maxnnz, rp[i + 1] - rp[i]);
    for (int j = rp[i]; j < rp[i + 1]; j++) {
      const int g = cols[j];
      if (g >= g0 && g < g0 + n) ci[j] = g - g0;
      else {
        auto it = hpos.find(g);
        if (it == hpos.end()) { G.err = "psolver_init_dist: a column of the row block is neither owned nor in the halo list"; return 1; }
        ci[j] = it->second;
      }
    }
    if (ci[rp[i]] != i) { G.err = "psolver_init_dist: the first entry of every row must be the diagonal (src/oce_ale.F90:1128-1151)"; return 1; }
  }
  if (maxnnz > 16) { G.err = "psolver_init_dist: more than 16 entries in a row"; return 1; }
  for (int k = 0; k < nsend; k++) if (sloc[k] < 0 || sloc[k] >= n) { G.err = "psolver_init_dist: send list entry outside the owned rows"; return 1; }
  if (fesom_internal_select_device(G.err)) { fprintf(stderr, "fesom_gpu: %s\n", G.err.c_str()); return 2; }
  HIPCHK(hipStreamCreate(&G.stream));
  G.serial = true; G.use_graph = false;
  DM &m = G.m;
  memset(&m, 0, sizeof(m));
  m.myN = n; m.N = n + nrecv; m.nza = nza; m.ssh_maxnnz = maxnnz; m.nl = 2; m.nlm1 = 1; m.ntr = 1;
  m.p.solver_precond = 1;
  const size_t N = m.N;
  m.sv_tol = soltol > 0.0 ? soltol : 0.0; m.sv_maxits = maxits > 0 ? maxits : 0;
  m.rowptr = dev_upload(rp); m.colind = dev_upload(ci);
  {
    std::vector<int> perm, inv, wid;
    solver_row_order(rp.data(), n, maxnnz, false, perm, inv, wid);
    m.sv_cols = (unsigned short *)dev_upload(ell_cols(rp.data(), ci.data(), n, maxnnz, perm, inv));
    m.sv_perm = dev_upload(perm); m.sv_inv = dev_upload(inv); m.sv_wid = dev_upload(wid);
    const int W = maxnnz <= 10 ? 10 : 16, NP = (n + 63) / 64 * 64;
    std::vector<int> c32((size_t)W * NP, 0);
    for (int i = 0; i < NP; i++)
      for (int k = 0; k < W; k++) c32[(size_t)k * NP + i] = (i < n && rp[i] + k < rp[i + 1]) ? ci[rp[i] + k] : (i < n ? i : 0);
    m.sv_colsi = dev_upload(c32);
  }
#define F(f, c) m.f = field(#f, c)
  F(ssh_values, nza); F(d_eta, N); F(ssh_rhs, N);
  F(sv_vals, 16 * (N + 64)); F(sv_dinv, N + 64); F(sv_b, N + 64); F(sv_r, N + 64); F(sv_r0, N + 64); F(sv_p, N + 64); F(sv_v, N + 64); F(sv_s, N + 64); F(sv_t, N + 64);
  F(sv_ph, N + 64); F(sv_x0, 16 * (N + 64)); F(sv_snap, N); F(sv_ph2, N + 64); F(sv_v2, N + 64);
  F(sv_part, 8 * ((N + 255) / 256 + 1)); F(sv_red, 8); F(sv_kry, 48);
  F(sv_resid, 1); F(sv_h1, N); F(sv_h2, N); F(sv_h3, N); F(sv_scale, N + 64);
  F(sv_bn, N + 64); F(sv_x, N + 64); F(sv_pd, N + 64); F(sv_sn, N + 64); F(sv_sh, N + 64);
#undef F
  m.sv_extrap = 0;                                   // warm start from the caller's `sol`, as the reference (psolve.c:155-221)
  m.sv_info = dev_alloc<int>(4);
  HIPCHK(hipMemcpy(m.ssh_values, vals, sizeof(double) * nza, hipMemcpyHostToDevice));
  G.npes = npes; G.mype = mype;
  G.precond_agreed = false; G.x_pending = false; G.generation++; g_prog_ready = false;
  G.hsend = G.hrecv = nullptr; G.hcap = 0; G.hsend1 = G.hrecv1 = nullptr; G.hcap1 = 0;
  for (int k = 0; k < 3; k++) G.halo[k] = Ctx::Halo();
  {
    Ctx::Halo &h = G.halo[0];
    h.rPE.assign(rPE, rPE + nr); h.sPE.assign(sPE, sPE + ns);
    h.rptr.assign(nr + 1, 1); h.sptr.assign(ns + 1, 1);                       // 1-based prefixes, as the reference's com_struct
    for (int p = 0; p < nr; p++) h.rptr[p + 1] = h.rptr[p] + rcnt[p];
    for (int p = 0; p < ns; p++) h.sptr[p + 1] = h.sptr[p] + scnt[p];
    h.nrecv = nrecv; h.nsend = nsend;
    h.slist_h.assign(sloc, sloc + nsend);
    std::vector<int> rl(nrecv);
    for (int k = 0; k < nrecv; k++) rl[k] = n + k;
    h.rlist = dev_upload(rl); h.slist = dev_upload(h.slist_h); h.slist_q = nullptr;
    h.rptr_d = dev_upload(h.rptr); h.sptr_d = dev_upload(h.sptr);
  }
  G.sub_int = G.sub_cb = G.sub_cbh = Ctx::ColList();
  {
    std::vector<int> inv;
    const char *pc = getenv("FESOM_GPU_PRECOND");
    if ((pc && !strcmp(pc, "jacobi")) || ras_device(n, m.N, rp.data(), ci.data(), vals, maxnnz, m, G.allocs, &inv)) m.rs_pinfo = nullptr;      // (does not qualify: Jacobi; the ranks agree at the first solve)
    if (m.rs_pinfo) {
      Ctx::Halo &h = G.halo[0];
      std::vector<int> sq(h.slist_h.size());
      for (size_t k = 0; k < sq.size(); k++) sq[k] = inv[h.slist_h[k]];
      h.slist_q = dev_upload(sq);
    }
  }
  for (auto &kv : G.fields) if (!kv.second.p) { G.err = "psolver_init_dist: device allocation failed"; return 1; }
  if (g_alloc_failed) { G.err = "psolver_init_dist: a device allocation or upload failed"; return 1; }
  solver_prepare();
  if (t) { PSD.tcopy = *t; PSD.t = &PSD.tcopy; } else PSD.t = nullptr;
  PSD.n = n; PSD.nza = nza; PSD.ok = true;
  G.first_step = 0; G.solver_only = true; G.ready = true;
  HIPCHK(hipDeviceSynchronize());
  return 0;
}
int fesom_gpu_psolve_dist(const double *rhs, const double *vals, double *sol, int newvals) {
  if (!PSD.ok || !G.ready || !G.solver_only) { G.err = "psolve_dist: fesom_gpu_psolver_init_dist has not been called"; return 1; }
  DM &m = G.m;
  if (!PSD.t && (!R.comm || R.nranks != G.npes || R.rank != G.mype)) { G.err = "psolve_dist: no transport given and the built-in RCCL transport is not initialised for this partition (fesom_gpu_comm_init)"; return 1; }
  if (newvals) HIPCHK(hipMemcpyAsync(m.ssh_values, vals, sizeof(double) * PSD.nza, hipMemcpyHostToDevice, G.stream));
  HIPCHK(hipMemcpyAsync(m.ssh_rhs, rhs, sizeof(double) * PSD.n, hipMemcpyHostToDevice, G.stream));
  HIPCHK(hipMemcpyAsync(m.d_eta, sol, sizeof(double) * PSD.n, hipMemcpyHostToDevice, G.stream));
  PStep S{PSD.t};
  if (agree_on_preconditioner(S)) return 1;
  if (part_solve(S) || S.rc) return 1;
  S.Wt();
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(G.stream));
  HIPCHK(hipMemcpy(sol, m.d_eta, sizeof(double) * PSD.n, hipMemcpyDeviceToHost));
  return 0;
}
int fesom_gpu_psolver_iterations(void) { return (PSD.ok && G.ready && G.solver_only) ? G.part_iters : -1; }
static void ps_solve_impl(int *id, double *rhs, double *vals, double *sol, int *newvals) {
  (void)id;
  if (!PS.ok) ps_die("psolve: psolver_init has not been called");
  DM &m = PS.m;
  if (*newvals) PSCHK(hipMemcpy(m.ssh_values, vals, sizeof(double) * PS.nza, hipMemcpyHostToDevice));
  PSCHK(hipMemcpy(m.ssh_rhs, rhs, sizeof(double) * PS.n, hipMemcpyHostToDevice));
  PSCHK(hipMemcpy(m.d_eta, sol, sizeof(double) * PS.n, hipMemcpyHostToDevice));
  if (launch_solver(m, 0)) ps_die("psolve: no solver kernel for this operator");
  PSCHK(hipGetLastError());
  PSCHK(hipDeviceSynchronize());
  PSCHK(hipMemcpy(sol, m.d_eta, sizeof(double) * PS.n, hipMemcpyDeviceToHost));
}
void psolver_init(int *id, int *stype, int *pctype, int *pcilutype, int *ilulevel, int *fillin, double *droptol, int *maxits,
                  int *restart, double *soltol, int *part, int *rptr, int *cols, double *vals, int *reuse, int *fcomm) {
  ps_init_impl(id, stype, pctype, pcilutype, ilulevel, fillin, droptol, maxits, restart, soltol, part, rptr, cols, vals, reuse, fcomm);
}
void psolve(int *id, double *rhs, double *vals, double *sol, int *newvals) { ps_solve_impl(id, rhs, vals, sol, newvals); }
void psolver_final(void) { ps_final_impl(); }
// the same three under library-prefixed names: what a host adapter that defines psolver_init / psolve / psolver_final itself forwards to on one rank
void fesom_gpu_psolver_init(int *id, int *stype, int *pctype, int *pcilutype, int *ilulevel, int *fillin, double *droptol, int *maxits,
                            int *restart, double *soltol, int *part, int *rptr, int *cols, double *vals, int *reuse, int *fcomm) {
  ps_init_impl(id, stype, pctype, pcilutype, ilulevel, fillin, droptol, maxits, restart, soltol, part, rptr, cols, vals, reuse, fcomm);
}
void fesom_gpu_psolve(int *id, double *rhs, double *vals, double *sol, int *newvals) { ps_solve_impl(id, rhs, vals, sol, newvals); }
void fesom_gpu_psolver_final(void) { ps_final_impl(); if (PSD.ok) { PSD.ok = false; if (G.ready && G.solver_only) fesom_gpu_finalize(); } }
}

// =====================================================================================================================
// Halo exchange (row e).  The library packs / unpacks, the HOST moves the bytes: an MPI host calls MPI_Isend/Irecv on
// the device buffers (GPU-aware MPI), this repository's Python host uses torch.distributed (RCCL over xGMI on a multi-GPU
// node, gloo with host staging in the 2-rank tests).  Mirrors exchange_nod / exchange_elem of the reference
// (src/gen_halo_exchange.F90:58-1035): `kind` 0 = com_nod2D, 1 = com_elem2D, 2 = com_elem2D_full; several fields that
// are exchanged at the same point of the step travel in ONE message per neighbour.
// Message layout: for every neighbour p (order of sPE / rPE) a contiguous block; inside it field after field, each
// [items of p][values per item] (an item is a node / element column, contiguous in memory: vertical index fastest).
// =====================================================================================================================
namespace {
__global__ void k_halo_pack(const double *__restrict__ f, int W, const int *__restrict__ list, const int *__restrict__ ptr, int npe, int nitems,
                            int Wtot, int Woff, double *__restrict__ buf, const long long *__restrict__ pbase) {
  long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (long long)nitems * W) return;
  int i = (int)(g / W), w = (int)(g % W);
  int p = 0;
  while (p + 1 < npe && i >= ptr[p + 1] - 1) p++;
  int first = ptr[p] - 1, cnt = ptr[p + 1] - ptr[p];
  const size_t base = pbase ? (size_t)pbase[p] : (size_t)first * Wtot;       // start of neighbour p's block of this part
  buf[base + (size_t)cnt * Woff + (size_t)(i - first) * W + w] = f[(size_t)list[i] * W + w];
}
__global__ void k_halo_unpack(double *__restrict__ f, int W, const int *__restrict__ list, const int *__restrict__ ptr, int npe, int nitems,
                              int Wtot, int Woff, const double *__restrict__ buf, const long long *__restrict__ pbase) {
  long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (long long)nitems * W) return;
  int i = (int)(g / W), w = (int)(g % W);
  int p = 0;
  while (p + 1 < npe && i >= ptr[p + 1] - 1) p++;
  int first = ptr[p] - 1, cnt = ptr[p + 1] - ptr[p];
  const size_t base = pbase ? (size_t)pbase[p] : (size_t)first * Wtot;
  f[(size_t)list[i] * W + w] = buf[base + (size_t)cnt * Woff + (size_t)(i - first) * W + w];
}
int halo_subfields(int kind, int nf, const char *const *names, std::vector<Sub> &subs, int &Wtot) {
  const DM &m = G.m;
  const size_t items = kind == 0 ? m.N : kind == 1 ? m.E : m.EX;
  subs.clear(); Wtot = 0;
  for (int k = 0; k < nf; k++) {
    auto it = G.fields.find(names[k]);
    if (it == G.fields.end()) { G.err = std::string("halo: unknown field ") + names[k]; return 1; }
    size_t cnt = it->second.count; int slabs = it->second.slabs;
    if (!strcmp(names[k], "tr_arr") || !strcmp(names[k], "tr_arr_old")) { slabs = m.ntr; cnt /= m.ntr; }   // (nz, node, tracer)
    bool q = false;
    if (!strncmp(names[k], "sv_", 3)) {                                                                   // solver vectors are padded
      cnt = items;
      // with the RAS preconditioner the owned part of the Krylov vectors is in the patch order (solver_ras.hip): the send list as positions
      q = G.m.rs_pinfo && kind == 0 && (!strcmp(names[k], "sv_x") || !strcmp(names[k], "sv_ph") || !strcmp(names[k], "sv_sh"));
      if (q && !G.halo[0].slist_q) { G.err = "halo: the send list in the solver's patch order is missing"; return 1; }
    }
    if (cnt % items) { G.err = std::string("halo: field size does not match the exchange kind: ") + names[k]; return 1; }
    for (int sl = 0; sl < slabs; sl++) { subs.push_back(Sub{(double *)it->second.p + (size_t)sl * cnt, (int)(cnt / items), q}); Wtot += (int)(cnt / items); }
  }
  return 0;
}
int halo_reserve(size_t doubles, int ch) {
  size_t &capr = ch ? G.hcap1 : G.hcap;
  if (doubles <= capr) return 0;
  size_t cap = doubles * 2;
  double *a = dev_alloc<double>(cap), *b2 = dev_alloc<double>(cap);      // (old buffers are released at finalize)
  if (!a || !b2) { G.err = "halo: buffer allocation failed"; return 1; }
  // dev_alloc clears the new buffers on the NULL stream, which the non-blocking communication stream does not wait for: make sure the
  // clearing is over before a pack kernel writes into them (buffers grow a few times in the first step only)
  if (hipDeviceSynchronize() != hipSuccess) { G.err = "halo: device synchronisation failed"; return 1; }
  if (ch) { G.hsend1 = a; G.hrecv1 = b2; } else { G.hsend = a; G.hrecv = b2; }
  capr = cap;
  return 0;
}
}  // namespace

// single exchanges for hosts that drive the step phase by phase (fesom2_amd/parallel.py): pack / unpack on the step's stream
static XPlan g_hx;                                      // plan of the exchange between fesom_gpu_halo_pack and fesom_gpu_halo_unpack
static int halo_pack_on(int kind, int nfields, const char *const *names, void **send_dev, void **recv_dev, int *values_per_item) {
  NEED_CTX();
  if (G.npes < 2) { G.err = "halo: single partition"; return 1; }
  std::vector<XSpec> spec(1);
  spec[0].kind = kind; spec[0].names.assign(names, names + nfields);
  if (xplan_build(spec, g_hx) || xplan_pack(g_hx, 0, false)) return 1;
  // stream-ordered: fesom_gpu_copy / fesom_gpu_sync wait for the pack kernels; a transport on the same stream
  // (fesom_gpu_set_stream) needs no host wait at all
  HIPCHK(hipGetLastError());
  *send_dev = G.hsend; *recv_dev = G.hrecv; *values_per_item = g_hx.parts[0].Wtot;
  return 0;
}
static int halo_unpack_on(int kind, int nfields, const char *const *names) {
  NEED_CTX();
  std::vector<XSpec> spec(1);
  spec[0].kind = kind; spec[0].names.assign(names, names + nfields);
  XPlan x;
  if (xplan_build(spec, x) || xplan_unpack(x, 0, false)) return 1;
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" {
int fesom_gpu_halo_info(int kind, int *npes, int *mype, int *nr, int *rPE, int *rcnt, int *ns, int *sPE, int *scnt) {
  NEED_CTX();
  if (kind < 0 || kind > 2) { G.err = "halo: bad kind"; return 1; }
  const Ctx::Halo &h = G.halo[kind];
  *npes = G.npes; *mype = G.mype; *nr = (int)h.rPE.size(); *ns = (int)h.sPE.size();
  for (size_t p = 0; p < h.rPE.size(); p++) { rPE[p] = h.rPE[p]; rcnt[p] = h.rptr[p + 1] - h.rptr[p]; }
  for (size_t p = 0; p < h.sPE.size(); p++) { sPE[p] = h.sPE[p]; scnt[p] = h.sptr[p + 1] - h.sptr[p]; }
  return 0;
}
// packs the send halo of `names` into the device send buffer; returns both device buffers and the number of values per
// item (a neighbour's block holds count(p) * values_per_item doubles, blocks are consecutive in sPE / rPE order)
int fesom_gpu_halo_pack(int kind, int nfields, const char *const *names, void **send_dev, void **recv_dev, int *values_per_item) {
  return halo_pack_on(kind, nfields, names, send_dev, recv_dev, values_per_item);
}
int fesom_gpu_halo_unpack(int kind, int nfields, const char *const *names) { return halo_unpack_on(kind, nfields, names); }
// plain copies for hosts that stage through host memory (dir 0: device -> host, 1: host -> device); synchronous
int fesom_gpu_copy(void *dst, const void *src, long long bytes, int dir) {
  NEED_CTX();
  HIPCHK(hipStreamSynchronize(G.stream));
  HIPCHK(hipMemcpy(dst, src, (size_t)bytes, dir == 0 ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice));
  return 0;
}
// device address of a named field (for a host transport that works on device memory, e.g. an all-reduce of sv_red)
int fesom_gpu_field_ptr(const char *name, void **dev, long long *count) {
  NEED_CTX();
  auto it = G.fields.find(name);
  if (it == G.fields.end()) { G.err = std::string("field_ptr: unknown field ") + name; return 1; }
  *dev = it->second.p; *count = (long long)it->second.count;
  return 0;
}
int fesom_gpu_sync(void) { NEED_CTX(); HIPCHK(hipStreamSynchronize(G.stream)); return check_ale_flag(); }
// run every kernel of the library on the host's stream (e.g. torch.cuda.current_stream().cuda_stream), so that the host's
// stream-ordered transport (RCCL) and the library's pack / unpack / compute kernels need no host synchronisation
int fesom_gpu_set_stream(void *hip_stream) {
  NEED_CTX();
  HIPCHK(hipStreamSynchronize(G.stream));
  if (!G.ext_stream) hipStreamDestroy(G.stream);
  G.stream = (hipStream_t)hip_stream; G.ext_stream = true;
  return 0;
}

// ---- built-in RCCL transport (see RcclApi above) -------------------------------------------------------------------------
// Rank 0 obtains the 128-byte RCCL unique id and hands it to the other ranks by whatever the host has (MPI_Bcast in a Fortran
// host, torch.distributed / a file in Python); then every rank calls fesom_gpu_comm_init.  Independent of fesom_gpu_init.
int fesom_gpu_comm_unique_id(void *id128) {
  if (rccl_load()) return 1;
  ncclUniqueId id;
  NCCLCHK(R.GetUniqueId(&id));
  memcpy(id128, &id, sizeof(id));
  return 0;
}
int fesom_gpu_comm_init(const void *id128, int nranks, int rank) {
  if (fesom_internal_select_device(G.err)) return 1;            // (independent of fesom_gpu_init: bind the communicator to this rank's device)
  if (rccl_load()) return 1;
  if (R.comm) { R.CommDestroy(R.comm); R.comm = nullptr; }
  if (nranks < 1 || rank < 0 || rank >= nranks) { G.err = "comm_init: bad rank / nranks"; return 1; }
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  NCCLCHK(R.CommInitRank(&R.comm, nranks, id, rank));
  R.nranks = nranks; R.rank = rank;
  if (!G.cstream && !getenv("FESOM_GPU_NO_OVERLAP")) {     // communication stream of the asynchronous exchanges
    if (hipStreamCreateWithFlags(&G.cstream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&G.ev_prod, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&G.ev_done, hipEventDisableTiming) != hipSuccess) { G.cstream = nullptr; G.err = "comm_init: cannot create the communication stream"; return 1; }
  }
  return 0;
}
int fesom_gpu_comm_finalize(void) {
  if (R.comm) { if (G.ready && G.stream) hipStreamSynchronize(G.stream); if (G.cstream) hipStreamSynchronize(G.cstream); R.CommDestroy(R.comm); R.comm = nullptr; }
  if (G.cstream) { hipStreamDestroy(G.cstream); hipEventDestroy(G.ev_prod); hipEventDestroy(G.ev_done); G.cstream = nullptr; }
  R.nranks = 0; R.rank = -1;
  return 0;
}
// Self test of the transport on the library's stream: a ring shift (every rank sends `n` doubles to rank+1 and receives from
// rank-1 in one group) and an in-place sum; returns 0 if both arrive as expected.  Works with one rank too (send to self).
int fesom_gpu_comm_selftest(int n) {
  NEED_CTX();
  if (!R.comm) { G.err = "comm_selftest: fesom_gpu_comm_init has not been called"; return 1; }
  if (n < 1) n = 1;
  double *buf = nullptr;
  HIPCHK(hipMalloc((void **)&buf, sizeof(double) * (2 * (size_t)n + 2)));
  std::vector<double> h(2 * (size_t)n + 2, 0.0);
  for (int i = 0; i < n; i++) h[i] = 1000.0 * R.rank + i;
  h[2 * (size_t)n] = R.rank + 1.0; h[2 * (size_t)n + 1] = 0.5;
  HIPCHK(hipMemcpyAsync(buf, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, G.stream));
  const int to = (R.rank + 1) % R.nranks, from = (R.rank + R.nranks - 1) % R.nranks;
  NCCLCHK(R.GroupStart());
  NCCLCHK(R.Send(buf, (size_t)n, ncclDouble, to, R.comm, G.stream));
  NCCLCHK(R.Recv(buf + n, (size_t)n, ncclDouble, from, R.comm, G.stream));
  NCCLCHK(R.GroupEnd());
  NCCLCHK(R.AllReduce(buf + 2 * (size_t)n, buf + 2 * (size_t)n, 2, ncclDouble, ncclSum, R.comm, G.stream));
  HIPCHK(hipMemcpyAsync(h.data(), buf, sizeof(double) * h.size(), hipMemcpyDeviceToHost, G.stream));
  HIPCHK(hipStreamSynchronize(G.stream));
  hipFree(buf);
  for (int i = 0; i < n; i++) if (h[(size_t)n + i] != 1000.0 * from + i) { G.err = "comm_selftest: ring shift delivered wrong data"; return 2; }
  if (h[2 * (size_t)n] != 0.5 * R.nranks * (R.nranks + 1) || h[2 * (size_t)n + 1] != 0.5 * R.nranks) { G.err = "comm_selftest: all-reduce delivered a wrong sum"; return 3; }
  return 0;
}
// exchanges / all-reduces issued by fesom_gpu_step_partitioned since the last call, and -- if timing was switched on with
// fesom_gpu_comm_timing(1) -- the device time (ms, HIP events on the library's stream) from the first pack kernel to the last
// unpack kernel of those exchanges, summed.  Resets the counters.
int fesom_gpu_comm_timing(int on) { G.comm_timing = on != 0; return 0; }
// exchange points, message parts (an exchange point that carries node and element fields has two), all-reduces and exchanges that ran on
// the communication stream since the last fesom_gpu_comm_stats call (not reset here)
int fesom_gpu_comm_counts(long long out[4]) { NEED_CTX(); out[0] = G.n_exch; out[1] = G.n_parts; out[2] = G.n_allred; out[3] = G.n_async; return 0; }
int fesom_gpu_comm_stats(long long *exchanges, long long *allreduces, double *exchange_ms) {
  NEED_CTX();
  double ms = 0.0;
  if (!G.comm_ev.empty()) {
    HIPCHK(hipStreamSynchronize(G.stream));
    for (auto &pr : G.comm_ev) { float t = 0; if (hipEventElapsedTime(&t, pr.first, pr.second) == hipSuccess) ms += t; hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    G.comm_ev.clear();
  }
  if (exchanges) *exchanges = G.n_exch;
  if (allreduces) *allreduces = G.n_allred;
  if (exchange_ms) *exchange_ms = ms;
  G.n_exch = G.n_allred = G.n_parts = G.n_async = 0;
  return 0;
}
}
