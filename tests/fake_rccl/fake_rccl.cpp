// TEST-ONLY stand-in for librccl.so (never loaded by the product unless MPCQ_RCCL_LIB names it): the four entry points libmpcq.so
// resolves -- ncclGetUniqueId, ncclCommInitRank, ncclAllReduce, ncclCommDestroy -- with the reduction carried out over a POSIX
// shared-memory segment named after the unique id.  It lets the multi-rank control flow of bench.py (unique id broadcast, one
// communicator per rank, mpcq_comm_share for the second engine, SUM + MAX reductions of the tracking statistic) run through the
// RCCL branch of libmpcq under torch.distributed.run on a machine without GPUs, on top of the lane emulator (whose "device
// pointers" are host pointers).  Blocking, float64 only, ops SUM (0) and MAX (2) -- what mpcq_allreduce_tracking_stats issues.
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>

namespace {
constexpr int MAX_RANKS = 64, MAX_COUNT = 16;
struct Seg {
  std::atomic<int> joined;          // ranks that have mapped the segment
  std::atomic<long> arrived;        // contributions written, over all reductions so far
  double slot[2][MAX_RANKS][MAX_COUNT];   // [parity of the reduction's sequence number][rank][element]
};
struct Comm { Seg* seg; int rank, nranks; long seq; char name[80]; };
struct Id128 { char b[128]; };
bool wait_until(const std::atomic<long>& v, long target) {
  const time_t t0 = time(nullptr);
  while (v.load(std::memory_order_acquire) < target) {
    sched_yield();
    if (time(nullptr) - t0 > 120) return false;   // a rank died: fail the call instead of hanging the test
  }
  return true;
}
}  // namespace

extern "C" {
int ncclGetUniqueId(void* id) {
  std::memset(id, 0, 128);
  std::snprintf((char*)id, 128, "/fake_rccl_%d_%ld_%d", (int)getpid(), (long)time(nullptr), rand());
  return 0;
}
int ncclCommInitRank(void** comm, int nranks, Id128 id, int rank) {
  if (nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return 4;   // ncclInvalidArgument
  Comm* c = new Comm();
  std::snprintf(c->name, sizeof(c->name), "%s", id.b);
  const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);   // a fresh segment is zero-filled: counters start at 0
  if (fd < 0 || ftruncate(fd, sizeof(Seg)) != 0) { delete c; return 2; }
  c->seg = (Seg*)mmap(nullptr, sizeof(Seg), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (c->seg == MAP_FAILED) { delete c; return 2; }
  c->rank = rank; c->nranks = nranks; c->seq = 0;
  c->seg->joined.fetch_add(1, std::memory_order_acq_rel);
  const time_t t0 = time(nullptr);
  while (c->seg->joined.load(std::memory_order_acquire) < nranks) {   // like the real call: returns when every rank has joined
    sched_yield();
    if (time(nullptr) - t0 > 120) return 6;
  }
  *comm = c;
  return 0;
}
int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, void* /*stream*/) {
  Comm* c = (Comm*)comm;
  if (!c || dtype != 8 || count > (size_t)MAX_COUNT || (op != 0 && op != 2)) return 4;
  const int par = (int)(c->seq & 1);
  std::memcpy(c->seg->slot[par][c->rank], send, count * sizeof(double));
  c->seg->arrived.fetch_add(1, std::memory_order_acq_rel);
  if (!wait_until(c->seg->arrived, (c->seq + 1) * c->nranks)) return 6;
  double out[MAX_COUNT];
  for (size_t k = 0; k < count; ++k) {
    double v = c->seg->slot[par][0][k];
    for (int r = 1; r < c->nranks; ++r) { const double w = c->seg->slot[par][r][k]; v = op == 0 ? v + w : (w > v ? w : v); }   // rank order: every rank gets the same bits
    out[k] = v;
  }
  std::memcpy(recv, out, count * sizeof(double));
  c->seq += 1;   // (slot[par] is rewritten by reduction seq + 2, which no rank enters before all have contributed to seq + 1, i.e. read seq)
  return 0;
}
int ncclCommDestroy(void* comm) {
  Comm* c = (Comm*)comm;
  if (!c) return 0;
  shm_unlink(c->name);
  munmap(c->seg, sizeof(Seg));
  delete c;
  return 0;
}
const char* ncclGetErrorString(int r) { return r == 0 ? "success" : (r == 6 ? "fake rccl: timed out waiting for the other ranks" : "fake rccl: error"); }
}
