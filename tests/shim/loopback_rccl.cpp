// TEST INFRASTRUCTURE -- not part of the product.  A loop-back stand-in for the five RCCL entry points
// edgegan_amd/csrc/comm.cpp binds by dlopen (ncclGetUniqueId, ncclCommInitRank, ncclAllReduce,
// ncclCommDestroy, ncclGetErrorString, ncclGetVersion), so that the engine's multi-rank path (apply_group: event-ordered
// all-reduce + RMSProp on the side stream, comm_wait_groups, comm_reduce_scalars, bench.py's barriers) can be
// run with SEVERAL RANKS ON ONE GPU: the real RCCL refuses two ranks on one device (ncclInvalidUsage), and
// the boxes the gpu tests run on have one.  Selected with EG_RCCL_LIB=<this .so> (comm.cpp: load_lib).
//
// An all-reduce here is: drain the stream, copy the rank's buffer into its slot of a file-backed shared
// mapping, meet the other ranks, reduce the slots in rank order on the host (every rank computes the same
// bits), meet again, copy the result back.  Every call also publishes (count, op) and fails with
// ncclInvalidUsage when the ranks disagree -- a collective that one rank enqueued and another did not is
// reported instead of hanging; a rank that never arrives is reported after EG_LOOPBACK_TIMEOUT_S (120 s).
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>

namespace {
constexpr int kMaxRanks = 8;
constexpr int kMagic = 0x4c425243;   // "LBRC"
struct Header {
  std::atomic<int> magic;
  std::atomic<int> arrived;
  std::atomic<int> generation;
  std::atomic<long long> call[kMaxRanks];   // count * 16 + op of the call each rank is in
  std::atomic<int> failed;
};
struct Comm {
  Header* h = nullptr;
  float* slots = nullptr;
  size_t cap = 0;          // floats per slot
  size_t bytes = 0;
  int rank = 0, nranks = 1;
  char path[160] = {0};
  std::vector<float> tmp;
};
struct UniqueId {
  char internal[128];
};
double timeout_s() {
  const char* e = getenv("EG_LOOPBACK_TIMEOUT_S");
  return e ? atof(e) : 120.0;
}
// sense-reversing barrier over the mapping; false = a rank did not arrive in time (or another rank failed)
bool meet(Comm* c) {
  Header* h = c->h;
  const int gen = h->generation.load();
  if (h->arrived.fetch_add(1) + 1 == c->nranks) {
    h->arrived.store(0);
    h->generation.fetch_add(1);
    return h->failed.load() == 0;
  }
  const auto t0 = std::chrono::steady_clock::now();
  while (h->generation.load() == gen) {
    sched_yield();
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) {
      h->failed.store(1);
      return false;
    }
  }
  return h->failed.load() == 0;
}
}  // namespace

extern "C" {

int ncclGetUniqueId(UniqueId* id) {
  memset(id->internal, 0, sizeof id->internal);
  unsigned long long r = 0;
  FILE* f = fopen("/dev/urandom", "rb");
  if (!f || fread(&r, sizeof r, 1, f) != 1) r = (unsigned long long)getpid() * 2654435761u + (unsigned long long)time(nullptr);
  if (f) fclose(f);
  const char* dir = getenv("EG_RDZV_DIR");
  snprintf(id->internal, sizeof id->internal, "%s/eg_loopback_rccl_%016llx", dir ? dir : "/tmp", r);
  return 0;
}

int ncclCommInitRank(void** out, int nranks, UniqueId id, int rank) {
  if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return 5;
  Comm* c = new Comm;
  c->rank = rank;
  c->nranks = nranks;
  const char* cap_mb = getenv("EG_LOOPBACK_CAP_MB");
  c->cap = (size_t)(cap_mb ? atoi(cap_mb) : 96) * (1u << 20) / sizeof(float);
  c->bytes = 4096 + (size_t)nranks * c->cap * sizeof(float);
  strncpy(c->path, id.internal, sizeof c->path - 1);
  int fd = -1;
  if (rank == 0) {
    fd = open(c->path, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) return 2;
  } else {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      struct stat st;
      fd = open(c->path, O_RDWR);
      if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size == c->bytes) break;
      if (fd >= 0) close(fd);
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) return 2;
      usleep(2000);
    }
  }
  void* m = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) return 2;
  c->h = reinterpret_cast<Header*>(m);
  c->slots = reinterpret_cast<float*>(static_cast<char*>(m) + 4096);
  if (rank == 0) c->h->magic.store(kMagic);   // the file starts zeroed: counters are 0
  const auto t0 = std::chrono::steady_clock::now();
  while (c->h->magic.load() != kMagic) {
    usleep(1000);
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) return 2;
  }
  if (!meet(c)) return 2;
  *out = c;
  return 0;
}

// dtype 7 = ncclFloat32 only; op 0 sum, 2 max, 3 min (what comm.cpp uses)
int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t stream) {
  Comm* c = static_cast<Comm*>(comm);
  if (!c || dtype != 7 || !(op == 0 || op == 2 || op == 3) || count > c->cap) return 5;
  if (hipStreamSynchronize(stream) != hipSuccess) return 1;
  float* mine = c->slots + (size_t)c->rank * c->cap;
  if (hipMemcpy(mine, send, count * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  c->h->call[c->rank].store((long long)count * 16 + op);
  if (!meet(c)) return 2;
  for (int r = 0; r < c->nranks; ++r)
    if (c->h->call[r].load() != (long long)count * 16 + op) {
      fprintf(stderr, "loopback_rccl: rank %d is in all-reduce(count %zu, op %d), rank %d in (count %lld, op %lld)\n",
              c->rank, count, op, r, c->h->call[r].load() / 16, c->h->call[r].load() % 16);
      c->h->failed.store(1);
    }
  c->tmp.resize(count);
  float* t = c->tmp.data();
  memcpy(t, c->slots, count * sizeof(float));
  for (int r = 1; r < c->nranks; ++r) {
    const float* s = c->slots + (size_t)r * c->cap;
    if (op == 0) for (size_t i = 0; i < count; ++i) t[i] += s[i];
    else if (op == 2) for (size_t i = 0; i < count; ++i) t[i] = t[i] > s[i] ? t[i] : s[i];
    else for (size_t i = 0; i < count; ++i) t[i] = t[i] < s[i] ? t[i] : s[i];
  }
  if (!meet(c)) return c->h->failed.load() ? 5 : 2;   // nobody overwrites a slot before everyone has read it
  if (hipMemcpy(recv, t, count * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return 1;
  return 0;
}

int ncclCommDestroy(void* comm) {
  Comm* c = static_cast<Comm*>(comm);
  if (!c) return 0;
  if (c->rank == 0) unlink(c->path);   // the mapping lives until the last rank unmaps it
  munmap(c->h, c->bytes);
  delete c;
  return 0;
}

int ncclGetVersion(int* v) {   // (comm_info: reported in bench.py's line; 0.0.1 marks the stand-in)
  *v = 1;
  return 0;
}

const char* ncclGetErrorString(int r) {
  switch (r) {
    case 0: return "success";
    case 1: return "loopback: HIP call failed";
    case 2: return "loopback: a rank did not arrive (system error)";
    case 5: return "loopback: invalid usage (ranks disagree on the collective, or bad arguments)";
    default: return "loopback: error";
  }
}
}
