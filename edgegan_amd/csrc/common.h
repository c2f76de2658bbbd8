// EdgeGAN MI355X engine -- shared host-side types (context, arena, tensors, tape).
// gfx950 only; no portability layers.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace eg {

// ---------------------------------------------------------------------------------------------
// errors: C++ exceptions inside the library, translated to int codes at the C ABI (capi.cpp).
// ---------------------------------------------------------------------------------------------
struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

[[noreturn]] inline void fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  throw Error(code, buf);
}

#define EG_HIP(expr)                                                                      \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess)                                                                 \
      ::eg::fail(2, "HIP error %d (%s) at %s:%d: %s", (int)_e, hipGetErrorString(_e),     \
                 __FILE__, __LINE__, #expr);                                              \
  } while (0)

#define EG_CHECK(cond, ...)                          \
  do {                                               \
    if (!(cond)) ::eg::fail(1, __VA_ARGS__);         \
  } while (0)

// ---------------------------------------------------------------------------------------------
// Arena: chunked bump allocator over HBM.  Sized for 288 GB parts: nothing is freed inside a
// training iteration, the pointer is reset between iterations, chunks persist, so after the first
// iteration there is no hipMalloc on the path and every buffer address is replay-stable.
// ---------------------------------------------------------------------------------------------
class Arena {
 public:
  explicit Arena(size_t chunk_bytes = (size_t)512 << 20) : chunk_bytes_(chunk_bytes) {}
  ~Arena() { release(); }
  void* alloc(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes == 0) bytes = 256;
    while (true) {
      if (cur_ < chunks_.size()) {
        Chunk& c = chunks_[cur_];
        if (off_ + bytes <= c.size) {
          void* p = (char*)c.base + off_;
          off_ += bytes;
          used_ += bytes;
          if (used_ > peak_) peak_ = used_;
          return p;
        }
        ++cur_;
        off_ = 0;
        continue;
      }
      size_t sz = bytes > chunk_bytes_ ? bytes : chunk_bytes_;
      void* base = nullptr;
      EG_HIP(hipMalloc(&base, sz));
      chunks_.push_back({base, sz});
      total_ += sz;
    }
  }
  float* allocf(size_t n) { return (float*)alloc(n * sizeof(float)); }
  void reset() {
    cur_ = 0;
    off_ = 0;
    used_ = 0;
  }
  void release() {
    for (auto& c : chunks_) (void)hipFree(c.base);
    chunks_.clear();
    reset();
    total_ = 0;
  }
  size_t peak() const { return peak_; }
  size_t reserved() const { return total_; }
  void swap(Arena& o) {
    chunks_.swap(o.chunks_);
    std::swap(chunk_bytes_, o.chunk_bytes_);
    std::swap(cur_, o.cur_);
    std::swap(off_, o.off_);
    std::swap(used_, o.used_);
    std::swap(peak_, o.peak_);
    std::swap(total_, o.total_);
  }

 private:
  struct Chunk {
    void* base;
    size_t size;
  };
  std::vector<Chunk> chunks_;
  size_t chunk_bytes_, cur_ = 0, off_ = 0, used_ = 0, peak_ = 0, total_ = 0;
};

// ---------------------------------------------------------------------------------------------
// Tensor: NHWC fp32 activation (2-D tensors use n x c with h = w = 1).  `g` is the gradient
// buffer, allocated on first use; `g_set` says whether it already holds a value (first writer
// stores, later writers accumulate -- no memsets).
// ---------------------------------------------------------------------------------------------
struct Param;
struct Tensor {
  float* d = nullptr;
  float* g = nullptr;
  int n = 0, h = 1, w = 1, c = 0;
  bool needs_grad = false;
  bool g_set = false;
  bool leaf_grad = false;   // gradient chain of this tensor runs on the lane's leaf stream (classifier's image pyramid: ops.cpp)
  bool g_inplace = false;   // the producer's backward rewrites g in place (conv + fused activation): g must not be shared
  int fused_act = 0;        // ... that activation (Act enum); g_act_done: its derivative was already applied to g by the
  bool g_act_done = false;  //     input-gradient kernel that wrote it (ConvArgs::epi_act)
  // The gradient of a tensor whose only reader is a 2x2 average pool (pool2_add behind a convolution: g_up_ok) is 0.25 x the 2x
  // nearest-neighbour upsampling of the pooled tensor's gradient: g_up = that tensor, g itself is NOT written -- the convolution's
  // two gradient kernels read the small tensor (ConvArgs::x_up, WgradArgs::dy_up; ops.cpp: materialize_up where they cannot)
  // out = prelu(x) with a learned slope (prelu_tail): x and the slope -- the single input-gradient convolution that writes out's
  // gradient may apply the PReLU's backward in its epilogue (ConvArgs::epi_act = ACT_PRELU_X; then g_act_done, and x's gradient
  // IS this buffer)
  // bf16 mode: the gradient of a convolution's output (g_plane_ok, set by conv2d) whose writer is an element-wise backward kernel
  // with a plane store is kept as that plane alone (g_plane_only; kernels.h: GRAD_PLANE_ONLY) -- both gradient kernels of the
  // convolution read planes; a reader that wants fp32 widens its own copy (ops.cpp: grad_f32_view), a second writer widens in place
  bool g_plane_ok = false;
  bool g_plane_only = false;
  Tensor* prelu_pre = nullptr;
  Param* prelu_leak = nullptr;
  bool g_up_ok = false;
  Tensor* g_up = nullptr;
  size_t g_epi_wrap = 0;    //     ... with the activation output of element i - g_epi_wrap for elements i >= g_epi_wrap (ConvArgs::epi_wrap)
  // forward-only convolution outputs left as their K-split slabs for the InstanceNorm pass that is their only reader
  // (ops.cpp: conv2d / deconv2d with Ctx::feeds_in, in_act): d is then NOT written
  const float* slabs = nullptr;
  const float* slab_bias = nullptr;
  const float* slab_base = nullptr;   // an accumulating convolution's slabs: the tensor they are added to (d itself)
  int nslab = 0;
  // The same for the GRADIENT: an input-gradient convolution that is the first writer of g may leave its K-split slabs
  // un-summed (g_slabs / g_nslab, stride numel()) when this tensor's producer declared that its backward reads them
  // (g_slab_ok: prelu, in_act, gate_prelu, pool2 -- ops.cpp); a second writer sums them first (grad_acc)
  const float* g_slabs = nullptr;
  int g_nslab = 0;
  bool g_slab_ok = false;
  // column sums of g (a bias gradient) may be taken from a SMALLER tensor with the same column sums: pool2_add's backward
  // writes g = up-sample(out.g) / 4, whose column sums are those of out.g (a quarter of the rows); cleared by grad_acc
  const float* g_colsum_src = nullptr;
  size_t g_colsum_rows = 0;
  // ... or from per-block column partials [g_colpart_rows][c] that the FIRST writer of g stored next to it (kernels.h: colpart)
  // because this tensor's producer -- a convolution with a trainable bias -- asked for them (want_colpart); cleared by grad_acc
  bool want_colpart = false;
  const float* g_colpart = nullptr;
  int g_colpart_rows = 0;
  // bf16 planes ([pixel][c] bf16) of d / g for the plane-fed kernels of the bf16 mode (k_conv_bf16p.hip).  Written by the
  // tensor's own producer kernel as a second store (then *_bf_stream == nullptr: the plane is visible wherever d / g is),
  // or on demand by launch_cvt_bf16 on the stream recorded here (ops.cpp: tensor_plane decides who else may read it)
  unsigned short* d_bf = nullptr;
  unsigned short* g_bf = nullptr;
  // bf16 mode: d was never written -- the tensor exists as its plane only (its readers are plane-fed dense contractions, which
  // round their operands to bf16 anyway).  A reader that wants fp32 widens the plane first (ops.cpp: need_f32; exact)
  bool d_plane_only = false;
  hipStream_t d_bf_stream = nullptr, g_bf_stream = nullptr;
  size_t numel() const { return (size_t)n * h * w * c; }
  size_t pixels() const { return (size_t)n * h * w; }
};
using T = Tensor*;

// Derived bf16 copy of a filter that outlives a sub-step.  It is built lazily on whichever lane's stream first needs it;
// `ready` is recorded behind the build and every OTHER stream waits for it before its first read (lanes run
// concurrently: ops.cpp limb_planes).
struct LimbCache {
  unsigned short* buf = nullptr;
  bool valid = false;
  int planes = 0;
  hipEvent_t ready = nullptr;
  hipStream_t built_on = nullptr;
  unsigned waited = 0;   // bit per stream (ops.cpp: stream_bit) that has already ordered itself behind `ready`
  // what the planes are built from (set at the first build): the batched rebuild (ops.cpp: limb_planes) refreshes every
  // stale cache of the same domain (network group / the spectral-norm weights) in one launch
  const float* src = nullptr;
  int T = 0, R = 0, Cc = 0, o = 0, domain = -1;
  void release() {
    if (buf) (void)hipFree(buf);
    if (ready) (void)hipEventDestroy(ready);
    buf = nullptr;
    ready = nullptr;
  }
};

// A model variable (TF variable of SURVEY Appendix C).  w/g/ms/mom point into the owning
// network group's flat buffers (one all-reduce + one RMSProp launch per group).
struct Param {
  std::string name;
  std::vector<int> shape;
  bool trainable = true;
  bool dead = false;         // trainable in TF but never receives a gradient (quirk N8)
  int init_kind = 0;         // 0 const, 1 normal, 2 truncated normal, 3 xavier uniform
  float init_param = 0.f;
  size_t numel = 0;
  float* w = nullptr;
  float* g = nullptr;
  float* ms = nullptr;
  float* mom = nullptr;
  bool want_grad = false;    // set per sub-step for the networks being optimised
  bool g_set = false;
  // per-tap transposed copy of a conv filter, kept across sub-steps until the variable changes (the discriminators'
  // filters are transposed for the input gradient in sub-steps 1-3, 5 and 7 but only change in their own sub-step)
  float* wT = nullptr;
  bool wT_valid = false;
  bool cache_T = false;      // network variables only (per-op test scaffolding re-uses Param slots)
  // phase-merged 3x3 form of a stride-2 filter with 3 image channels (ops.cpp: merged_filter), same life cycle as wT
  float* wM = nullptr;
  bool wM_valid = false;
  // bf16 limb planes of a filter (k_gemm_x3.hip), K-step-major, [0] as stored / [1] per-tap transposed; same life cycle
  LimbCache wL[2];
  void invalidate_derived() { wT_valid = wM_valid = wL[0].valid = wL[1].valid = false; }
};

struct Ctx;

// Tape of backward closures; run in reverse.
struct Tape {
  std::vector<std::function<void()>> ops;
  std::function<void()> after;   // runs when a backward pass ends (the engine flushes its queued reductions here)
  void push(std::function<void()> f) { ops.push_back(std::move(f)); }
  void backward() {
    for (size_t i = ops.size(); i-- > 0;) ops[i]();
    if (after) after();
  }
  void backward_range(size_t lo, size_t hi) {
    for (size_t i = hi; i-- > lo;) ops[i]();
    if (after) after();
  }
  void clear() { ops.clear(); }
};

}  // namespace eg
