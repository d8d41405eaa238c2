// TEST INFRASTRUCTURE -- a shared-memory stand-in for the nine librccl entry points that libfesom_gpu.so's built-in transport
// uses (fesom2_amd/csrc/api.hip: RcclApi), so that the transport's peers / offsets / counts / group structure can be exercised
// between two or four REAL processes that share the one GPU of a test box (RCCL itself refuses two ranks on one device).
// Selected with FESOM_GPU_RCCL_LIB=<this .so>; never part of the product.  Semantics kept: ncclSend/ncclRecv are stream-ordered
// point-to-point messages matched per (source, destination) in posting order and completed at ncclGroupEnd; a receive whose byte
// count differs from the matching send fails (ncclInvalidArgument) instead of silently truncating; ncclAllReduce sums in rank order.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <vector>

namespace {
constexpr int MAXR = 8;
constexpr size_t CAP = 6u << 20;
struct Mailbox { std::atomic<uint64_t> written, consumed; uint64_t bytes; char data[CAP]; };
struct Shm {
  std::atomic<int> count; std::atomic<int> sense;
  double red[MAXR][8192];
  Mailbox mb[MAXR][MAXR];
};
struct Comm { Shm *s; int n, rank; int local_sense; char name[64]; };
struct Op { bool send; void *buf; size_t bytes; int peer; Comm *c; hipStream_t st; };
thread_local int depth = 0;
thread_local std::vector<Op> ops;

template <class F> bool spin(F f) {
  auto t0 = std::chrono::steady_clock::now();
  while (!f()) {
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return false;
    usleep(20);
  }
  return true;
}
bool barrier(Comm *c) {
  c->local_sense = 1 - c->local_sense;
  if (c->s->count.fetch_add(1) + 1 == c->n) { c->s->count.store(0); c->s->sense.store(c->local_sense); return true; }
  return spin([&] { return c->s->sense.load() == c->local_sense; });
}
ncclResult_t flush() {
  for (auto &o : ops)
    if (o.send) {
      Mailbox &m = o.c->s->mb[o.c->rank][o.peer];
      if (o.bytes > CAP) return ncclInvalidArgument;
      if (!spin([&] { return m.written.load() == m.consumed.load(); })) return ncclSystemError;
      if (hipStreamSynchronize(o.st) != hipSuccess) return ncclUnhandledCudaError;
      if (hipMemcpy(m.data, o.buf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
      m.bytes = o.bytes;
      m.written.fetch_add(1);
    }
  for (auto &o : ops)
    if (!o.send) {
      Mailbox &m = o.c->s->mb[o.peer][o.c->rank];
      if (!spin([&] { return m.written.load() > m.consumed.load(); })) return ncclSystemError;
      if (m.bytes != o.bytes) { fprintf(stderr, "fake_rccl: rank %d expects %zu bytes from %d, which sent %llu\n", o.c->rank, o.bytes, o.peer, (unsigned long long)m.bytes); return ncclInvalidArgument; }
      if (hipStreamSynchronize(o.st) != hipSuccess) return ncclUnhandledCudaError;
      if (hipMemcpy(o.buf, m.data, o.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
      m.consumed.fetch_add(1);
    }
  ops.clear();
  return ncclSuccess;
}
size_t tsize(ncclDataType_t t) { return t == ncclDouble || t == ncclInt64 || t == ncclUint64 ? 8 : t == ncclFloat || t == ncclInt32 || t == ncclUint32 ? 4 : 1; }
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "/fesom_fake_rccl_%d_%lld", (int)getpid(), (long long)std::chrono::steady_clock::now().time_since_epoch().count());
  return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
  if (nranks > MAXR) return ncclInvalidArgument;
  int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
  if (fd < 0) return ncclSystemError;
  if (ftruncate(fd, sizeof(Shm)) != 0) return ncclSystemError;
  void *p = mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return ncclSystemError;
  Comm *c = new Comm{(Shm *)p, nranks, rank, 0, {0}};
  strncpy(c->name, id.internal, sizeof(c->name) - 1);
  *comm = (ncclComm_t)c;
  return barrier(c) ? ncclSuccess : ncclSystemError;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  Comm *c = (Comm *)comm;
  if (!c) return ncclSuccess;
  barrier(c);
  munmap(c->s, sizeof(Shm));
  if (c->rank == 0) shm_unlink(c->name);
  delete c;
  return ncclSuccess;
}
ncclResult_t ncclGroupStart() { depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd() { if (--depth == 0) return flush(); return ncclSuccess; }
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t st) {
  Comm *c = (Comm *)comm;
  if (peer < 0 || peer >= c->n) return ncclInvalidArgument;
  for (auto &o : ops) if (o.send && o.peer == peer) return ncclInvalidUsage;     // one message per peer and group (what the library issues)
  ops.push_back(Op{true, (void *)buf, count * tsize(t), peer, c, st});
  return depth ? ncclSuccess : flush();
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t st) {
  Comm *c = (Comm *)comm;
  if (peer < 0 || peer >= c->n) return ncclInvalidArgument;
  ops.push_back(Op{false, buf, count * tsize(t), peer, c, st});
  return depth ? ncclSuccess : flush();
}
ncclResult_t ncclAllReduce(const void *sb, void *rb, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t comm, hipStream_t st) {
  Comm *c = (Comm *)comm;
  if (t != ncclDouble || op != ncclSum || count > 8192) return ncclInvalidArgument;
  if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
  if (hipMemcpy(c->s->red[c->rank], sb, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  if (!barrier(c)) return ncclSystemError;
  static thread_local double out[8192];
  for (size_t i = 0; i < count; i++) { double a = 0.0; for (int r = 0; r < c->n; r++) a += c->s->red[r][i]; out[i] = a; }
  if (!barrier(c)) return ncclSystemError;
  if (hipMemcpy(rb, out, count * 8, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
}
const char *ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "success";
    case ncclInvalidArgument: return "fake_rccl: invalid argument (peer out of range, message too large or byte counts of a send/recv pair differ)";
    case ncclInvalidUsage: return "fake_rccl: two sends to one peer in a group";
    case ncclSystemError: return "fake_rccl: shared memory or timeout";
    default: return "fake_rccl: HIP error";
  }
}
}
