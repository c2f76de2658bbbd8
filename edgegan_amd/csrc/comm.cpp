// Data-parallel gradient exchange: one process per GPU, ONE RCCL all-reduce (sum, fp32) per
// optimised network group over xGMI.  RCCL is bound lazily with dlopen so that
// single-GPU use has no dependency on it, and so that a host process that already carries a
// librccl (e.g. through torch) shares that copy instead of loading a second one.
#include <dlfcn.h>

#include "engine.h"

namespace eg {

namespace {
struct NcclUniqueId {
  char internal[128];
};
typedef int (*fn_get_id)(NcclUniqueId*);
typedef int (*fn_init_rank)(void**, int, NcclUniqueId, int);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_destroy)(void*);
typedef const char* (*fn_errstr)(int);

void* load_lib(Ctx& c) {
  if (c.nccl_lib) return c.nccl_lib;
  // EG_RCCL_LIB: bind this library instead (a site's own RCCL build; the gpu tests' loop-back stand-in that lets
  // several ranks share ONE device, tests/shim/loopback_rccl.cpp) -- no fallback when it is set and does not load
  if (const char* forced = getenv("EG_RCCL_LIB")) {
    c.nccl_lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
    if (c.nccl_lib) return c.nccl_lib;
    fail(3, "cannot load EG_RCCL_LIB=%s: %s", forced, dlerror());
  }
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    c.nccl_lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (c.nccl_lib) return c.nccl_lib;
  }
  fail(3, "cannot load librccl: %s", dlerror());
}
template <class F>
F sym(Ctx& c, const char* name) {
  void* p = dlsym(load_lib(c), name);
  if (!p) fail(3, "librccl lacks %s", name);
  return reinterpret_cast<F>(p);
}
void check(Ctx& c, int r, const char* what) {
  if (r != 0) {
    const char* msg = "?";
    if (void* p = dlsym(c.nccl_lib, "ncclGetErrorString")) msg = reinterpret_cast<fn_errstr>(p)(r);
    fail(3, "RCCL %s failed: %d (%s)", what, r, msg);
  }
}
}  // namespace

void comm_unique_id(Ctx& c, char* out128) {
  NcclUniqueId id;
  check(c, sym<fn_get_id>(c, "ncclGetUniqueId")(&id), "ncclGetUniqueId");
  memcpy(out128, id.internal, 128);
}

// The exchange's footprint on the compute units.  RCCL's all-reduce kernels run one workgroup per channel, concurrently with
// the engine's matrix kernels, which take a whole CU each (150 KB of LDS, __launch_bounds__(512, 1)): every CU a channel
// lands on is a CU whose resident compute block is slowed, and RCCL picks its channel count for bandwidth, not for that.
//   EG_COMM_CHANNELS=n  caps the channels (exported as NCCL_MAX_NCHANNELS before the communicator is created, unless the
//                       caller already set that variable; NCCL_MIN_NCHANNELS and RCCL's other variables pass through untouched:
//                       RCCL reads the process environment)
//   EG_COMM_CUS=n       creates the communication stream with a CU mask of n compute units (a multiple of 8: n / 8 per XCD, the
//                       highest-numbered ones; mask bit i = XCD i % 8, CU i / 8 -- the interleaved layout of multi-XCD parts), so the
//                       exchange cannot spread beyond them.  (The complementary mask on the COMPUTE lanes was measured and removed:
//                       a CU-masked stream costs the matrix kernels 29 % whatever its size, profiles/r06j_cu_mask_cost.txt.)
// Both are measured knobs, not defaults: the reference is single-GPU (edgegan/train.py:77) and nothing beyond one device could
// be measured by the builder.
void comm_cu_mask(int ncu, uint32_t out[8]) {
  for (int w = 0; w < 8; ++w) out[w] = 0;
  for (int i = 256 - ncu; i < 256; ++i) out[i >> 5] |= 1u << (i & 31);
}
int comm_reserved_cus() {
  const char* e = getenv("EG_COMM_CUS");
  if (!e) return 0;
  const int n = atoi(e);
  EG_CHECK(n >= 8 && n <= 128 && n % 8 == 0, "EG_COMM_CUS: a multiple of 8 in 8..128, got %s", e);
  return n;
}

void comm_init(Ctx& c, const char* id128, int rank, int nranks) {
  EG_CHECK(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank/nranks");
  NcclUniqueId id;
  memcpy(id.internal, id128, 128);
  EG_HIP(hipSetDevice(c.cfg.device));
  if (const char* ch = getenv("EG_COMM_CHANNELS")) {
    EG_CHECK(atoi(ch) >= 1 && atoi(ch) <= 128, "EG_COMM_CHANNELS: 1..128, got %s", ch);
    setenv("NCCL_MAX_NCHANNELS", ch, 0);
  }
  check(c, sym<fn_init_rank>(c, "ncclCommInitRank")(&c.nccl_comm, nranks, id, rank), "ncclCommInitRank");
  c.rank = rank;
  c.nranks = nranks;
  if (!c.comm_stream) {
    if (const int ncu = comm_reserved_cus()) {
      uint32_t mask[8];
      comm_cu_mask(ncu, mask);
      EG_HIP(hipExtStreamCreateWithCUMask(&c.comm_stream, 8, mask));
      c.comm_cus = ncu;
    } else {
      EG_HIP(hipStreamCreateWithFlags(&c.comm_stream, hipStreamNonBlocking));
    }
    for (int g = 0; g < G_COUNT; ++g) {
      EG_HIP(hipEventCreateWithFlags(&c.ev_ready[g], hipEventDisableTiming));
      EG_HIP(hipEventCreateWithFlags(&c.ev_applied[g], hipEventDisableTiming));
    }
  }
  // Every rank must issue the same collectives in the same order, and the order of an iteration's seven all-reduces is a
  // function of the lane schedule (EG_LANES) and of the graph (classes, switched-off discriminators): ranks that disagree
  // would pair different groups' buffers -- or hang.  Checked here, once, instead of discovered at step one.
  if (nranks > 1) {
    const float key[6] = {(float)c.lanes_mask, (float)c.cfg.num_classes, (float)c.cfg.B, (float)(c.cfg.H * 4096 + c.cfg.W),
                          (float)((c.cfg.use_image_dis ? 1 : 0) | (c.cfg.use_edge_dis ? 2 : 0)), (float)((c.cfg.bf16 ? 1 : 0) | (c.cfg.x3 ? 2 : 0))};
    float lo[6], hi[6];
    for (int i = 0; i < 6; ++i) lo[i] = hi[i] = key[i];
    comm_reduce_scalars(c, lo, 6, 3);
    comm_reduce_scalars(c, hi, 6, 2);
    for (int i = 0; i < 6; ++i)
      if (lo[i] != hi[i])
        fail(3, "data-parallel ranks disagree on %s (this rank %g, over the ranks %g .. %g): every rank must run the same lane schedule "
                "and graph, or their collectives do not pair", i == 0 ? "EG_LANES" : i == 1 ? "num_classes" : i == 2 ? "the per-GPU batch size"
                : i == 3 ? "the image size" : i == 4 ? "use_image_discriminator / use_edge_discriminator" : "compute_dtype", key[i], lo[i], hi[i]);
  }
}

void comm_allreduce(Ctx& c, float* buf, size_t n, hipStream_t stream, int group) {
  EG_CHECK(c.nccl_comm, "comm_allreduce without comm_init");
  ++c.comm_calls;
  c.comm_bytes += (long long)n * 4;
  Ctx::CommRec* rec = nullptr;
  if (c.comm_timing && c.comm_recs.size() < 4096) {   // event pair around the collective, on the stream it runs on (eg_comm_timing)
    c.comm_recs.emplace_back();
    rec = &c.comm_recs.back();
    rec->group = group;
    rec->bytes = (long long)n * 4;
    EG_HIP(hipEventCreate(&rec->a));
    EG_HIP(hipEventCreate(&rec->b));
    EG_HIP(hipEventRecord(rec->a, stream));
  }
  // ncclFloat32 = 7, ncclSum = 0
  check(c, sym<fn_allreduce>(c, "ncclAllReduce")(buf, buf, n, 7, 0, c.nccl_comm, stream), "ncclAllReduce");
  if (rec) EG_HIP(hipEventRecord(rec->b, stream));
}

// The collectives recorded since comm_timing was switched on, in issue order: group id (engine.h: GroupId; G_G1 stands for the
// generators' joint exchange), payload bytes, milliseconds between the two events (start = the stream reaches the collective,
// i.e. this rank's gradients are final; end = the reduced buffer is complete: the wait for the slowest rank is inside).
int comm_timing_read(Ctx& c, float* ms, int* group, long long* bytes, int cap) {
  if (c.comm_stream) EG_HIP(hipStreamSynchronize(c.comm_stream));
  int n = 0;
  for (auto& r : c.comm_recs) {
    if (n < cap) {
      float t = 0.f;
      EG_HIP(hipEventElapsedTime(&t, r.a, r.b));
      ms[n] = t;
      group[n] = r.group;
      bytes[n] = r.bytes;
      ++n;
    }
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  c.comm_recs.clear();
  return n;
}

// Every lane's stream waits (not only the active one): pending[g] is cleared for the whole context, so a lane that
// did not fork from the active lane after this point must not be able to read the group's weights early.
void comm_wait_groups(Ctx& c, unsigned mask) {
  for (int g = 0; g < G_COUNT; ++g)
    if ((mask >> g & 1u) && c.pending[g]) {
      for (int i = 0; i < Ctx::kLanes; ++i)
        if (c.lanes[i].stream) EG_HIP(hipStreamWaitEvent(c.lanes[i].stream, c.ev_applied[g], 0));
      c.pending[g] = false;
    }
}

// Host scalars reduced over the ranks on the communication stream (also the host barrier of the data-parallel
// loop: returns when every rank has arrived).  op: 0 sum, 2 max, 3 min (ncclRedOp_t)
void comm_reduce_scalars(Ctx& c, float* inout, int n, int op) {
  EG_CHECK(c.nccl_comm, "comm_reduce_scalars without comm_init");
  EG_CHECK(n >= 1 && n <= 64 && (op == 0 || op == 2 || op == 3), "comm_reduce_scalars: n in 1..64, op in {0,2,3}");
  if (!c.comm_scal) EG_HIP(hipMalloc(&c.comm_scal, 64 * sizeof(float)));
  EG_HIP(hipMemcpyAsync(c.comm_scal, inout, n * sizeof(float), hipMemcpyHostToDevice, c.comm_stream));
  check(c, sym<fn_allreduce>(c, "ncclAllReduce")(c.comm_scal, c.comm_scal, (size_t)n, 7, op, c.nccl_comm, c.comm_stream),
        "ncclAllReduce(scalars)");
  EG_HIP(hipMemcpyAsync(inout, c.comm_scal, n * sizeof(float), hipMemcpyDeviceToHost, c.comm_stream));
  EG_HIP(hipStreamSynchronize(c.comm_stream));
}

void comm_wait_all(Ctx& c) { comm_wait_groups(c, ~0u); }

// What the exchange of this context looks like from outside (bench.py prints it so that a scaling line is
// self-describing): RCCL version code (ncclGetVersion), the file the bound library was mapped from, and the gradient
// all-reduces / bytes enqueued so far.
void comm_info(Ctx& c, int* version, char* libpath, int cap, long long* calls, long long* bytes) {
  *version = 0;
  if (libpath && cap > 0) libpath[0] = 0;
  if (c.nccl_lib) {
    typedef int (*fn_ver)(int*);
    if (void* p = dlsym(c.nccl_lib, "ncclGetVersion")) (void)reinterpret_cast<fn_ver>(p)(version);
    Dl_info di;
    void* any = dlsym(c.nccl_lib, "ncclAllReduce");
    if (any && dladdr(any, &di) && di.dli_fname && libpath && cap > 0) snprintf(libpath, cap, "%s", di.dli_fname);
  }
  *calls = c.comm_calls;
  *bytes = c.comm_bytes;
}

void comm_destroy(Ctx& c) {
  if (c.comm_stream) {
    (void)hipStreamSynchronize(c.comm_stream);
    for (auto& r : c.comm_recs) {
      (void)hipEventDestroy(r.a);
      (void)hipEventDestroy(r.b);
    }
    c.comm_recs.clear();
    for (int g = 0; g < G_COUNT; ++g) {
      (void)hipEventDestroy(c.ev_ready[g]);
      (void)hipEventDestroy(c.ev_applied[g]);
      c.pending[g] = false;
    }
    (void)hipStreamDestroy(c.comm_stream);
    c.comm_stream = nullptr;
    if (c.apply_stream) {
      (void)hipStreamSynchronize(c.apply_stream);
      (void)hipStreamDestroy(c.apply_stream);
    }
    c.apply_stream = nullptr;
    (void)hipFree(c.comm_scal);
    c.comm_scal = nullptr;
  }
  if (c.nccl_comm) {
    (void)sym<fn_destroy>(c, "ncclCommDestroy")(c.nccl_comm);
    c.nccl_comm = nullptr;
  }
  c.nranks = 1;
  c.rank = 0;
}

}  // namespace eg
