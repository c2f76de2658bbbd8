// Host-callable launchers for the gfx950 kernels (k_gemm.hip, k_norm.hip, k_elem.hip).
// Every launcher enqueues on the given stream and returns immediately.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>

namespace eg {

// A/B and timing switches (EG_NO_*, EG_X3_*, EG_TRACE_GEMM, ...) exist only in builds made with -DEG_EXPERIMENTS
// (make EXPERIMENTS=1 -> libedgegan_hip_exp.so, selected with EG_LIB_PATH by tools/gpu_ab*.py); the shipped library
// ignores them.  Runtime options of the product proper -- EG_LANES, EG_DEFER_REDUCE, EG_RCCL_LIB -- read getenv directly.
inline const char* dbg_env(const char* name) {
#ifdef EG_EXPERIMENTS
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}


// ---------------------------------------------------------------------------------------------
// Tap-gather implicit GEMM (conv forward, conv input-gradient, transposed conv), NHWC fp32:
//   Y[pix_out(m)][n] (+)= bias[n] + sum_{tap in phase} sum_c X[pix_in(m,tap)][c] * Wt[wt(tap)][c][n]
// One launch covers up to 4 output phases (sub-pixel classes of a stride-2 transposed conv).
// ---------------------------------------------------------------------------------------------
struct ConvPhase {
  int OHp, OWp;      // phase sub-grid extents (rows m enumerate (nb, oy, ox) over it)
  int ophy, ophx;    // output pixel = (oy*os + ophy, ox*os + ophx)
  int Ty, Tx;        // taps in this phase
  int kh0, kw0;      // filter tap = (kh0 + ty*kst, kw0 + tx*kst)
  int offy, offx;    // input pixel = (oy*S + ty*D + offy, ox*S + tx*D + offx)
};

struct ConvArgs {
  const float* x;    // [NB][IH][IW] pixels, x_ps floats per pixel, C channels used
  const unsigned short* x_bf;   // optional bf16 PLANE of x ([pixel][xbf_ps] bf16, written by x's producer or by launch_cvt_bf16):
  int xbf_ps;                   //   pixel pitch of the plane in elements, xbf_bytes its extent (buffer-load bound; 0 = >= 2 GB).
  unsigned xbf_bytes;           //   The plane-fed kernel (k_conv_bf16p.hip) stages it by LDS-DMA; x itself is then not read
  int x_is_activation;          // x is a forward activation (written once per sub-step): its bf16 copy may be kept and reused
  const float* wgt;  // element (tap wt, c, n) at wgt[wt*w_ts + c*ldw + n]
  const float* bias; // [N] or null
  const unsigned short* wgt_bf;  // optional bf16 filter, element (tap wt, n, c) at wgt_bf[wt*wbf_ts + n*ldwbf + c]:
  long long wbf_ts;              // when set and the shape qualifies the contraction runs on the bf16 MFMA
  int ldwbf;
  int limbs;                     // 3: wgt_bf holds the three bf16 limb planes of the fp32 filter (plane stride wbf_plane
  long long wbf_plane;           //    elements) and the contraction runs as exact-product limb MFMAs (k_gemm_x3.hip)
  int x3_tile;                   //    tile family chosen by conv_pick_ksplit for the limb kernels: 1 = 2x128x128 ping-pong, 2 = 64x64
  unsigned x_bytes, wbf_bytes;   //    extents of x and of the limb planes (buffer-load bounds; 0 = unknown / >= 2 GB)
  float* y;          // [NB][OHf][OWf] pixels, y_ps floats per pixel
  int NB, IH, IW, C, x_ps;
  int OHf, OWf, N, y_ps;
  int KW, kst, S, D, os;
  long long w_ts;
  int ldw;
  int reflect;       // 1: out-of-range input index is mirrored (tf.pad REFLECT), 0: contributes zero
  int accumulate;    // 1: y += result
  int act;           // fused activation of the output (Act enum; only the small-C kernel applies it: ops.cpp conv2d)
  int shuf;          // 1 (N == 12, tiles <*,16> only): channel n = (py*2+px)*3 + ch is stored to output pixel
                     //    (2*oy+py, 2*ox+px), channel ch of a [NB][OHf][OWf][y_ps] image; bias is indexed by ch
  int ksplit;        // > 1: K is split over ksplit blocks writing partial slabs to ws, reduced afterwards
  float* ws;         // ksplit * y_total floats
  size_t y_total;    // NB*OHf*OWf*y_ps
  int nphase;
  ConvPhase ph[4];
  int keep_slabs;   // K-split launches: leave the slabs in ws (no conv_split_reduce); the caller's consumer sums them
  // Activation backward in the epilogue (input-gradient launches): y holds dL/d(act output); with epi_act = ACT_RELU /
  // ACT_LRELU and epi_ref = that activation's OUTPUT (same layout as y) the kernel stores y * act'(out) instead -- the
  // in-place pass of the producing conv + activation is then skipped.  Honoured by the un-split halo kernel only; the
  // launcher reports it in epi_applied.
  const float* epi_ref;
  int epi_act;
  int epi_applied;
  int x_up;          // 1: the gathered operand is 0.25 x the 2x nearest-neighbour upsampling of a [NB][IH/2][IW/2][x_ps] tensor at x
                     //    (x_bf: its plane) -- the gradient of a 2x2 average pool read without materialising it (halo kernel and
                     //    plane-fed kernel, stride-1 launches; launch_conv_gemm checks).  x_bytes / xbf_bytes: the small tensor's.
  const unsigned short* epi_ref_bf;   // ReLU / LeakyReLU forms: the activation's output as its bf16 plane (a plane-only tensor); the sign is all they need
  const float* epi_leak;   // epi_act == ACT_PRELU_X: y holds dL/d(prelu output), epi_ref the PReLU's INPUT x, *epi_leak its slope lk; the
                           //   kernel stores (lk x >= x ? lk y : y) -- prelu_bwd's expression (k_elem.hip) -- and, with epi_bf, its bf16 plane
  unsigned short* epi_bf;
  int epi_skip_f32;        // with epi_bf, un-split launches of the kernels that honour it: the fp32 output is NOT stored (its readers take the plane)
  int epi_soft;            // run_conv: a K-split launch drops the epilogue request and keeps its slabs instead (keep_slabs_soft)
  int keep_slabs_soft;
  unsigned long long epi_wrap;   // > 0: output elements at or beyond this index take epi_ref[index - epi_wrap] (the stacked
                                 // discriminator sub-step: the tangent rows are masked by the primal rows' activation)
  int x3q;           // experiments build: run_conv chose the two-blocks-per-CU halo kernel (k_gemm_x3q.hip) and ITS K split for this launch
  int y_skip;             // with y_bf (conv_pk): the fp32 output is NOT stored -- the tensor lives as its plane alone (Tensor::d_plane_only)
  unsigned short* y_bf;   // bf16 mode, conv_pk only: also store the (activated) output's bf16 plane [pixel][y_ps] -- the next
                          // convolution reads it by LDS-DMA and no conversion pass runs (d_conv_0 + LeakyReLU -> d_conv_1)
  int pk;            // f32x3 / bf16 modes: thin forward layers (3 or 8 input channels) may run the packed-K limb kernel (k_thin_pk.hip)
  // Second output of the epilogue (conv.py:217-220, activation.py:23-27: the PReLU that follows Conv_1 of an MRU unit):
  // y2 = max(*y2_leak * y, y) of the value just stored to y, as fp32 (same layout as y; may be null) and / or as its bf16
  // plane.  Written by the un-split launches of the kernels that report CONV_CAP_Y2 in `caps`; everybody else leaves it to
  // the caller's prelu pass.
  const float* y2_leak;
  float* y2;
  unsigned short* y2_bf;
  // pool2_add in the epilogue (conv.py:239-243: ht_new = mean_pool(... + h_new); un-split stride-1 launches of the kernels that
  // report CONV_CAP_POOL): INSTEAD of y the kernel stores  pool_y = mean of each 2x2 block of (y incl. bias) + pool_b  at the
  // pooled resolution ([NB][OHf/2][OWf/2][y_ps]) and, with pool_leak, prelu(pool_y, *pool_leak) as pool_y2 / its bf16 plane.
  // The 2x2 partners of an accumulator sit in the SAME register of lanes l ^ 1 and l ^ TW of the 32-row MFMA tile.
  float* pool_y;
  const float* pool_b;
  const float* pool_leak;
  float* pool_y2;
  unsigned short* pool_y2_bf;
  int caps;          // OUT, set by launch_conv_gemm: what the kernel that ran honoured (CONV_CAP_*); ksplit is written back too
  int tuned;         // conv_pick_ksplit took (x3_tile, ksplit) from the measured per-shape table (x3_tuned.h): run_conv keeps them
};
// bit 1 of the `accumulate` argument of the gradient writers that can store a bf16 plane (in_bwd, in_tangent_bwd's tangent output,
// prelu_bwd, minmax_bwd -- their float4 forms): compute, store the plane, do NOT store the fp32 gradient (bf16 mode: both
// gradient kernels of the convolution behind it read the plane; ops.cpp: Tensor::g_plane_only)
enum { GRAD_ACCUMULATE = 1, GRAD_PLANE_ONLY = 2 };
enum ConvCaps { CONV_CAP_EPI_ACT = 1, CONV_CAP_Y2 = 2, CONV_CAP_POOL = 4,
                CONV_CAP_TANH = 8 /* ConvArgs::act == ACT_TANH applied to the stored value (the phase-merged scatter kernel) */ };

// A tensor that may still be the K-split slabs of the convolution that produced it (ConvArgs::keep_slabs): its value is
// bias[c] + slab 0 + slab 1 + ... (+ base), conv_split_reduce's order, formed by the CONSUMER's load instead of a
// reduction pass of its own (one launch, one write and one read of the tensor less).  n <= 1: an ordinary tensor at p.
struct SlabSrc {
  const float* p;              // the tensor, or slab 0
  int n;                       // slabs (0 / 1: plain tensor)
  int C;                       // channels (the bias period), % 4 == 0 when n > 1
  unsigned long long stride;   // floats between slabs
  const float* bias;           // [C] or null
  const float* base;           // added last (the tensor an accumulating convolution adds to), or null
};
inline SlabSrc plain_src(const float* p) { return SlabSrc{p, 1, 0, 0ull, nullptr, nullptr}; }
// y (+)= bias + slab 0 + ... : the stand-alone reduction (k_gemm.hip), for readers that do not take a SlabSrc
void launch_conv_split_reduce(const float* ws, float* y, const float* bias, size_t total, int y_ps, int N, int ksplit,
                              int accumulate, hipStream_t s);

// flops (2*M*N*K summed over phases) are returned for the profiler.
double launch_conv_gemm(ConvArgs& a, hipStream_t s, const char** kernel_name);
bool conv_up_kernel(const ConvArgs& a);     // would launch_conv_gemm run a kernel that honours ConvArgs::x_up for these arguments?
bool conv_x3_takes_halo(const ConvArgs& a, bool big_tile);
int conv_pick_ksplit(ConvArgs& a);   // also sets a.x3_tile for limb-mode launches
#ifdef EG_EXPERIMENTS
// k_gemm_x3q.hip (experiments build, EG_X3Q=1): the f32x3 halo convolution at two blocks per CU (128 x 128 tiles, 16-channel
// K chunks) -- built and parity-green in round 4, equal to the one-block kernel within +-5 % per layer: docs/experiments.md
bool conv_x3q_eligible(const ConvArgs& a);
int conv_x3q_pick_ksplit(const ConvArgs& a);
const char* launch_conv_x3q(const ConvArgs& a, hipStream_t s);
#endif

// ---------------------------------------------------------------------------------------------
// Weight gradient: dW[wt(tap)][c][n] = sum_m X[pix_in(m,tap)][c] * dY[m][n], m over NB*OH*OW
// "output" pixels, pix_in = (oy*S + kh - pad, ox*S + kw - pad) (mirrored when reflect).
// Split over m with a fixed-order second-stage reduction (deterministic, no float atomics).
// ---------------------------------------------------------------------------------------------
struct WgradArgs {
  const float* x;    // gathered operand [NB][IH][IW][x_ps], C channels
  const float* dy;   // dense operand    [NB][OH][OW][dy_ps], N channels
  float* dw;         // [KH*KW][C][N] with tap stride w_ts, row stride ldw
  float* ws;         // workspace for split partials: nsplit * (KH*KW*w_ts) floats
  int NB, IH, IW, C, x_ps;
  int OH, OW, N, dy_ps;
  int KH, KW, S, pad;
  long long w_ts;
  int ldw;
  int reflect;
  int accumulate;    // 1: dw += result
  int nsplit;
  int bf16;          // 1: round both operands to bf16 and use the bf16 MFMA (fp32 accumulate)
  int limbs;         // 3: fp32 result from three-limb bf16 operand splits on the bf16 MFMA (k_gemm_x3.hip)
  unsigned x_bytes, dy_bytes;   // extents of x / dy for the limb kernel's buffer loads (0 = unknown or >= 2 GB)
  int defer_reduce;  // 1: the caller runs the second-stage reduction of the split partials itself (launch_wgrad sets
                     //    reduce_deferred when it left the reduction out: the float4 form applies)
  int reduce_deferred;
  int dbg;           // experiments build only: limiter probes of the filter-gradient kernel (k_gemm_x3.hip)
  // bf16 mode: bf16 planes of both operands ([pixel][C] / [pixel][N] bf16, from the tensors' producers or launch_cvt_bf16);
  // when both are there and the shape fits, the plane-fed kernel (k_wgrad_bf16p.hip) stages them by LDS-DMA
  const unsigned short* x_bf;
  const unsigned short* dy_bf;
  unsigned xbf_bytes, dybf_bytes;
  int dy_up;         // 1: the dense operand is 0.25 x the 2x nearest-neighbour upsampling of a [NB][OH/2][OW/2][dy_ps] tensor at dy
                     //    (dy_bf: its plane; ConvArgs::x_up) -- row-halo limb kernel and plane-fed kernel only (wgrad_up_kernel)
};
#ifdef __HIPCC__
// XCD-aware block map (MI355X: 8 XCDs with private L2s; block b of a launch is observed on XCD b % 8): dispatch index ->
// virtual index such that every XCD owns one contiguous range of virtual indices (bijective for any block count)
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned total, int off = 0) {
  if (off) return b;
  const unsigned q = total >> 3, r = total & 7, x = b & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}
// ConvArgs::y2 / y2_bf: PReLU of the quad just stored at element offset `off` of y (lk = *a.y2_leak, read once per kernel);
// the plane is rounded to nearest even, the rounding of cvt_bf16 / st_bf4 (k_elem.hip)
__device__ __forceinline__ void conv_epi_y2(const ConvArgs& a, size_t off, const float4& v, float lk) {
  const float4 p = make_float4(fmaxf(lk * v.x, v.x), fmaxf(lk * v.y, v.y), fmaxf(lk * v.z, v.z), fmaxf(lk * v.w, v.w));
  if (a.y2) *reinterpret_cast<float4*>(a.y2 + off) = p;
  if (a.y2_bf) {
    typedef __bf16 bf16x2_e __attribute__((ext_vector_type(2)));
    typedef float f32x2_e __attribute__((ext_vector_type(2)));
    const f32x2_e lo = {p.x, p.y}, hi = {p.z, p.w};
    uint2 o;
    o.x = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2_e));
    o.y = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2_e));
    *reinterpret_cast<uint2*>(a.y2_bf + off) = o;
  }
}
// bf16 plane of a stored quad (round to nearest even: cvt_bf16's rounding)
__device__ __forceinline__ void conv_epi_plane(unsigned short* dst, const float4& v) {
  typedef __bf16 bf16x2_e __attribute__((ext_vector_type(2)));
  typedef float f32x2_e __attribute__((ext_vector_type(2)));
  const f32x2_e lo = {v.x, v.y}, hi = {v.z, v.w};
  uint2 o;
  o.x = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2_e));
  o.y = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2_e));
  *reinterpret_cast<uint2*>(dst) = o;
}
// ConvArgs::pool_y: one accumulator quad of the pooled epilogue.  v = the quad of output pixel (py, px) of a tile TW wide; on
// return the lanes with even (py, px) have stored  0.25 * (sum over the 2x2 block) + bias + pool_b  (and its PReLU) at element
// offset `poff` of the pooled tensors.  Every lane of the wave must call it (cross-lane reads).
__device__ __forceinline__ void conv_epi_pool(const ConvArgs& a, float4 v, int xr /* = TW */, bool owner, size_t poff, int n, float plk) {
  v.x += __shfl_xor(v.x, 1, 64); v.y += __shfl_xor(v.y, 1, 64); v.z += __shfl_xor(v.z, 1, 64); v.w += __shfl_xor(v.w, 1, 64);
  v.x += __shfl_xor(v.x, xr, 64); v.y += __shfl_xor(v.y, xr, 64); v.z += __shfl_xor(v.z, xr, 64); v.w += __shfl_xor(v.w, xr, 64);
  if (!owner) return;
  float4 r = make_float4(0.25f * v.x, 0.25f * v.y, 0.25f * v.z, 0.25f * v.w);
  if (a.bias) {
    const float4 b4 = *reinterpret_cast<const float4*>(a.bias + n);
    r.x += b4.x; r.y += b4.y; r.z += b4.z; r.w += b4.w;
  }
  if (a.pool_b) {
    const float4 q = *reinterpret_cast<const float4*>(a.pool_b + poff);
    r.x += q.x; r.y += q.y; r.z += q.z; r.w += q.w;
  }
  *reinterpret_cast<float4*>(a.pool_y + poff) = r;
  if (a.pool_leak) {
    const float4 p = make_float4(fmaxf(plk * r.x, r.x), fmaxf(plk * r.y, r.y), fmaxf(plk * r.z, r.z), fmaxf(plk * r.w, r.w));
    if (a.pool_y2) *reinterpret_cast<float4*>(a.pool_y2 + poff) = p;
    if (a.pool_y2_bf) {
      typedef __bf16 bf16x2_p __attribute__((ext_vector_type(2)));
      typedef float f32x2_p __attribute__((ext_vector_type(2)));
      const f32x2_p lo = {p.x, p.y}, hi = {p.z, p.w};
      uint2 o;
      o.x = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2_p));
      o.y = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2_p));
      *reinterpret_cast<uint2*>(a.pool_y2_bf + poff) = o;
    }
  }
}
#endif
// k_elem.hip: a memory-free v_mfma_f32_32x32x16_bf16 loop, blocks x 8 waves x iters x 16 MFMAs (scratch >= blocks * 512 floats)
void launch_mfma_probe(float* scratch, int blocks, int iters, int random_operands, hipStream_t s);
int wgrad_x3_rowhalo_units(const WgradArgs& a);   // k_gemm_x3.hip: units of the row-halo filter-gradient kernel (0: it does not apply)
// k_thin_wgrad.hip: filter gradients with 3 / 8 gathered channels, (tap, channel) packed into the MFMA's rows
bool wgrad_pk_eligible(const WgradArgs& a);
int wgrad_pk_splits(const WgradArgs& a);
const char* launch_wgrad_pk(const WgradArgs& a, float* out, int dense, hipStream_t s);
int wgrad_pick_splits(const WgradArgs& a);
size_t wgrad_workspace_floats(const WgradArgs& a);
double launch_wgrad(const WgradArgs& a, hipStream_t s, const char** kernel_name);
bool wgrad_up_kernel(const WgradArgs& a);   // would launch_wgrad run a kernel that honours WgradArgs::dy_up for these arguments?

// plane-fed bf16 filter gradient (k_wgrad_bf16p.hip): units > 0 when it takes the launch
int wgrad_bf16p_units(const WgradArgs& a);
const char* launch_wgrad_bf16p(const WgradArgs& a, float* out, int dense, hipStream_t s);

// thin-layer kernels (k_thin.hip): 3x3 stride-1 convolution with <= 12 output channels as one product over the input
// pixels + col2im in LDS
bool conv_scatter_eligible(const ConvArgs& a);
const char* launch_conv_scatter(const ConvArgs& a, hipStream_t s);
// thin forward convolutions (3 / 8 input channels; 3x3 stride 1, 4x4 stride 2) with (tap, channel) packed into K on limb MFMAs
bool conv_pk_eligible(const ConvArgs& a);
const char* launch_conv_pk(const ConvArgs& a, hipStream_t s);
// convolution with <= 8 input channels: input rows and the whole filter staged in LDS, MFMA operands built from LDS
bool conv_smallc_eligible(const ConvArgs& a);
const char* launch_conv_smallc(const ConvArgs& a, hipStream_t s);

// Second stages of the split reductions (filter gradients: sum of the pixel-split partial slabs; bias gradients: sum of
// the row-block partials) are tiny launches, ~150 per iteration; they are queued and run as ONE launch per batch
// (ops.cpp: flush_deferred).  Same arithmetic and summation order as wgrad_reduce_v_kernel / col_sum_stage2.
struct RedDesc {
  const float* src;
  float* dst;
  int kind;          // 0: filter-gradient slabs (float4 form), 1: bias-gradient row blocks
  int T, C, N;       // kind 0: taps, channels, outputs; kind 1: C = columns
  long long w_ts;
  int ldw;
  int n;             // kind 0: nsplit, kind 1: row blocks
  int accumulate;
};
constexpr int kRedBatch = 24;
struct RedBatch {
  int count;
  unsigned start[kRedBatch + 1];   // first block of each descriptor
  RedDesc d[kRedBatch];
};
unsigned red_desc_blocks(const RedDesc& d);
void launch_reduce_batch(const RedBatch& b, hipStream_t s);
size_t col_sum_partial_floats(size_t R, int C);
int launch_col_sum_stage1(const float* dy, size_t R, int C, float* part, hipStream_t s);   // returns the row blocks

// bf16-operand kernels (k_gemm_bf16.hip); the tile choice is made by launch_conv_gemm / launch_wgrad
bool conv_bf16_eligible(const ConvArgs& a);
const char* launch_conv_gemm_bf16(const ConvArgs& a, int maxM, bool big_tile, hipStream_t s);
const char* launch_wgrad_bf16(const WgradArgs& a, float* out, int dense, bool big_tile, hipStream_t s);
void launch_cvt_bf16(const float* in, unsigned short* out, size_t n, hipStream_t s);
void launch_cvt_from_bf16(const unsigned short* in, float* out, size_t n, hipStream_t s);
// dx = dout * act'(out) with the activation's output given as its bf16 plane (ReLU / LeakyReLU: the sign decides); dx and / or its
// plane dx_bf are stored (n % 4 == 0)
void launch_act_bwd_bf(const unsigned short* out_bf, const float* dout, float* dx, unsigned short* dx_bf, size_t n, int act, hipStream_t s);   // exact widening, n % 4 == 0
void launch_tap_transpose_bf16(const float* in, unsigned short* out, int T, int R, int Cc, hipStream_t s);

// plane-fed bf16 kernel for stride-1 layers (k_conv_bf16p.hip): the gathered operand arrives from a bf16 plane by LDS-DMA
bool conv_bf16p_eligible(const ConvArgs& a);
int conv_bf16p_pick_ksplit(const ConvArgs& a);
const char* launch_conv_bf16p(const ConvArgs& a, hipStream_t s);

// halo-staged limb / bf16 kernel for stride-1 layers (k_gemm_x3h.hip)
bool conv_x3h_eligible(const ConvArgs& a);
bool conv_x3h_pool_ok(const ConvArgs& a);      // this launch can take ConvArgs::pool_y (the pooled epilogue)
bool conv_bf16p_pool_ok(const ConvArgs& a);
const char* launch_conv_gemm_x3h(const ConvArgs& a, hipStream_t s);

// fp32-via-three-bf16-limbs kernels (k_gemm_x3.hip)
bool conv_x3_eligible(const ConvArgs& a);
const char* launch_conv_gemm_x3(const ConvArgs& a, int maxM, bool big_tile, hipStream_t s, int* caps = nullptr);   // caps |= CONV_CAP_* of the kernel chosen
const char* launch_wgrad_x3(const WgradArgs& a, float* out, int dense, bool big_tile, hipStream_t s);
// src [T][R][Cc] fp32 -> three bf16 limb planes, K-step-major [plane][T][K/32][Nrows][32]; (Nrows, K) = (R, Cc), or (Cc, R)
// when transposed; K % 32 == 0
void launch_split_bf16x3(const float* in, unsigned short* out, int T, int R, int Cc, int transposed, size_t plane,
                         int nplanes /* 3, or 1: the first limb only = the value rounded to bf16 */, hipStream_t s);
// the same for up to kSplitMax filters in one launch (all stale limb-plane caches of a network group)
constexpr int kSplitMax = 40;
struct SplitDesc {
  const float* in;
  unsigned short* out;
  unsigned long long plane;
  int R, Cc, T, transposed, nplanes, gx, gy, pad_;
};
struct SplitBatch {
  int count;
  unsigned start[kSplitMax + 1];
  SplitDesc d[kSplitMax];
};
void launch_split_bf16x3_batch(const SplitBatch& b, hipStream_t s);

// Phase-merged filter of a stride-2 layer with 3 image channels: the 4 sub-pixel phases of a transposed convolution
// (or of a stride-2 convolution's input gradient) become 12 output channels n = (py*2+px)*3 + ch of ONE 3x3 stride-1
// convolution over the coarse grid, wm[(dy+1)*3 + (dx+1)][c][n] = w[kh][kw] with kh = py + 1 - 2*dy (0 where kh, kw fall
// outside the k x k filter).  mode 0: w is a deconv filter [k*k][3][C]; mode 1: w is a conv filter HWIO [k*k][3][C]
// (both: image channel before the contracted channel).
void launch_merge_phase_filter(const float* w, float* wm, int k, int C, hipStream_t s);

// [T][R][Cc] -> [T][Cc][R] (per-tap transpose), optional scale by 1/(*inv_of) when inv_of != null.
void launch_tap_transpose(const float* in, float* out, int T, int R, int Cc, const float* inv_of,
                          hipStream_t s);

// ---------------------------------------------------------------------------------------------
// norm / activation kernels (k_norm.hip).  act: 0 none, 1 relu, 2 lrelu(max(x,0.2x)), 3 tanh
// ---------------------------------------------------------------------------------------------
enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_LRELU = 2, ACT_TANH = 3,
           ACT_PRELU_X = 4 };   // ConvArgs::epi_act only: PReLU backward from the PRE-activation (epi_ref) and the learned slope (epi_leak)

// InstanceNorm (x-mean)/(sqrt(var)+1e-5) + act, per (n,c) over H*W. stats[n][c] = {mean, sigma}.
// nslab > 1: x is the first of nslab K-split slabs (slab_stride floats apart) of a convolution output that was never
// summed; the load pass forms bias + slab 0 + slab 1 + ... in conv_split_reduce's order (C % 32 == 0, HW <= 1024 only)
// and, with xsum, also stores that sum (training: the pre-norm tensor the backward reads)
void launch_in_fwd(const float* x, float* out, float* stats, int NB, int HW, int C, int act, hipStream_t s,
                   int nslab = 0, size_t slab_stride = 0, const float* slab_bias = nullptr, float* xsum = nullptr,
                   unsigned short* out_bf = nullptr,   // *_bf: also store the bf16 plane of the output (bf16 mode, C % 32 == 0)
                   float* chunk_scratch = nullptr);
// chunk_scratch (launch_in_fwd / launch_in_bwd / launch_in_tangent_fwd): in_chunk_scratch_floats(NB, HW, C) floats of the caller's,
// private to the launch until it has run.  With it, maps of more than 1,024 pixels (in_chunked_shape) take the chunked two-stage
// form -- pixel chunks x channel groups x images instead of one block per (image, channel group).
bool in_chunked_shape(int HW, int C);
size_t in_chunk_scratch_floats(int NB, int HW, int C);
// dx (+)= J * (dout * act'(out))   (J = IN Jacobian, symmetric)
// dout_slabs (C % 32 == 0 and HW <= 1024 only): dout is that SlabSrc instead
void launch_in_bwd(const float* x, const float* out, const float* dout, const float* stats, float* dx,
                   int accumulate, int NB, int HW, int C, int act, hipStream_t s, unsigned short* dx_bf = nullptr,
                   const SlabSrc* dout_slabs = nullptr, float* colpart = nullptr, float* chunk_scratch = nullptr);
// colpart (here and in the other gradient writers below): per-block column sums of the gradient JUST WRITTEN -- the first
// stage of the bias gradient of the convolution whose output gradient this is (conv.py:31-34: db = sum over pixels of dY), so
// that no pass re-reads dY for it.  [rows][C] floats; the launcher returns `rows` through colpart_rows_* below (0: this form of
// the kernel does not produce them, the caller runs col_sum_stage1 as before).  Only meaningful for a first writer.
int in_bwd_colpart_rows(int NB, int HW, int C);
bool in_bwd_takes_slabs(int HW, int C);
// tangent forward: tout = act'(out_primal) * (J * tx)
void launch_in_tangent_fwd(const float* x, const float* out, const float* stats, const float* tx, float* tout,
                           int NB, int HW, int C, int act, hipStream_t s, unsigned short* tout_bf = nullptr,
                           float* chunk_scratch = nullptr);
// tangent backward: given dtout, produce dtx (+)= ..., dx_primal (+)= ...
void launch_in_tangent_bwd(const float* x, const float* out, const float* stats, const float* tx,
                           const float* dtout, float* dtx, int acc_dtx, float* dx, int acc_dx, int NB, int HW,
                           int C, int act, hipStream_t s, unsigned short* dtx_bf = nullptr, float* chunk_scratch = nullptr);
// plain activation (no norm): fwd, bwd (dx (+)= dout*act'(out)), tangent (tout = act'(out)*tx)
void launch_act_fwd(const float* x, float* out, size_t n, int act, hipStream_t s);
void launch_act_bwd(const float* out, const float* dout, float* dx, int accumulate, size_t n, int act,
                    hipStream_t s);

// BatchNorm (training statistics) + ReLU over [R][C] rows: y = relu(gamma*(x-mean)/sqrt(var+eps)+beta)
// stats = {mean[C], var[C]}
// out_bf: also the output's bf16 plane (bf16 mode: what g_dconv_1 reads); returns whether the kernel that ran stored it
bool launch_bn_relu_fwd(const float* x, const float* gamma, const float* beta, float* out, float* stats,
                        int R, int C, float eps, hipStream_t s, unsigned short* out_bf = nullptr);
void launch_bn_relu_bwd(const float* x, const float* out, const float* dout, const float* gamma,
                        const float* stats, float* dx, int acc_dx, float* dgamma, int acc_dg, float* dbeta,
                        int acc_db, int R, int C, float eps, hipStream_t s);
// moving statistics update: mm = d*mm + (1-d)*mean, mv = d*mv + (1-d)*var*R/(R-1), applied `times` times
// BatchNorm (batch statistics, learned scale / shift) + activation as the `norm` of a block (normalization.py:19-25 with
// E_norm / G_norm / D_norm = 'batch'): two-stage over row chunks; scratch = bn_scratch_floats(R, C) floats per call in flight.
// stats = [mean | biased var] (2 C floats).  The tangent pair serves the gradient penalty through a BatchNorm discriminator.
size_t bn_scratch_floats(int R, int C);
void launch_bn_act_fwd(const float* x, const float* gamma, const float* beta, float* out, float* stats, float* scratch, float* mm,
                       float* mv, int times, int R, int C, float eps, int act, hipStream_t s);
void launch_bn_act_bwd(const float* x, const float* out, const float* dout, const float* gamma, const float* stats, float* scratch,
                       float* dx, int acc_dx, float* dgamma, int acc_dg, float* dbeta, int acc_db, int R, int C, float eps, int act,
                       hipStream_t s);
void launch_bn_act_tangent_fwd(const float* x, const float* out, const float* gamma, const float* stats, const float* tx, float* tout,
                               float* scratch, int R, int C, float eps, int act, hipStream_t s);
void launch_bn_act_tangent_bwd(const float* x, const float* out, const float* gamma, const float* stats, const float* tx,
                               const float* dtout, float* scratch, float* dtx, int acc_dtx, float* dx, int acc_dx, float* dgamma,
                               int acc_dg, int R, int C, float eps, int act, hipStream_t s);
void launch_bn_moving_update(float* mm, float* mv, const float* stats, int C, int R, float decay, int times,
                             hipStream_t s);

// ---------------------------------------------------------------------------------------------
// elementwise / data movement / reductions / losses / optimizer (k_elem.hip)
// ---------------------------------------------------------------------------------------------
void launch_fill(float* p, float v, size_t n, hipStream_t s);
void launch_copy(const float* src, float* dst, size_t n, hipStream_t s);
// y (+)= a*x
void launch_axpy(const float* x, float* y, float a, int accumulate, size_t n, hipStream_t s);
// out = a + b ; out = relu(a+b)
void launch_add(const float* a, const float* b, float* out, int relu, size_t n, hipStream_t s);
// dst[n][h][wo + w][c] = src[n][h][w][c]  (copy a [NB][H][Ws][C] block into a wider [NB][H][Wd][C] at column wo)
// to_dst=1: wide[.., wo+w, ..] (+)= narrow ; to_dst=0: narrow (+)= wide[.., wo+w, ..]
void launch_copy_cols(const float* narrow, float* wide, int NB, int H, int Ws, int Wd, int wo, int C, int to_dst,
                      int accumulate, hipStream_t s);
// xhat = real + alpha[n]*(fake-real)
void launch_blend(const float* real, const float* fake, const float* alpha, float* out, int NB, size_t per,
                  hipStream_t s);
// gradient of tf.pad(REFLECT, p): fold [NB][H+2p][W+2p][C] back onto [NB][H][W][C]
void launch_reflect_fold(const float* dpad, float* dx, int accumulate, int NB, int H, int W, int C, int p,
                         hipStream_t s);
// bias gradient: db[c] (+)= sum_rows dy[r][c]
size_t col_sum_scratch_floats(size_t R, int C);
void launch_col_sum(const float* dy, float* db, int accumulate, size_t R, int C, float* scratch, hipStream_t s);
// 2x2 mean pool NHWC fwd / bwd
void launch_pool2_fwd(const float* x, float* y, int NB, int H, int W, int C, hipStream_t s);
// y = pool2(relu(a + b)) (+ y's bf16 plane) in one pass; false: the shape has no float4 form (the caller runs the separate ops)
// forward-only encoder head: relu + global mean, the two FC8 linears, z_r = mu + eps_r exp(log sigma) (z1 may be null)
void launch_enc_head(const float* x, const float* wmu, const float* bmu, const float* wls, const float* bls, const float* eps, float* mu,
                     float* ls, float* z0, float* z1, int NB, int HW, int C, int Z, int round_bf16, hipStream_t s);
bool launch_add_relu_pool2_fwd(const float* a, const float* b, float* y, unsigned short* y_bf, int NB, int H, int W, int C, hipStream_t s);
void launch_pool2_bwd(const float* dy, float* dx, int accumulate, int NB, int H, int W, int C, hipStream_t s,
                      unsigned short* dx_bf = nullptr,   // dx_bf: also store the bf16 plane of dx (k_elem.hip: st_bf4)
                      const SlabSrc* dy_slabs = nullptr);   // float4 form only (ops.cpp: flat_slab_ok)
// y = pool2(a + b), a/b [NB][H][W][C]; returns false (nothing launched) when the float4 form does not apply
bool launch_add_pool2_fwd(const float* a, const float* b, float* y, int NB, int H, int W, int C, hipStream_t s);
// y = pool2(a) + b with b [NB][H/2][W/2][C]; same return convention
// a_slabs: a is that SlabSrc; leak / y2 / y2_bf: also store prelu(y, *leak) (the next MRU unit's prelu(h), conv.py:185) and
// its bf16 plane
bool launch_pool2_add_fwd(const float* a, const float* b, float* y, int NB, int H, int W, int C, hipStream_t s,
                          const SlabSrc* a_slabs = nullptr, const float* leak = nullptr, float* y2 = nullptr,
                          unsigned short* y2_bf = nullptr);
// global spatial mean: y[n][c] = mean_hw act(x[n][hw][c]) ; pre_relu applies relu first
void launch_gmean_fwd(const float* x, float* y, int NB, int HW, int C, int pre_relu, hipStream_t s);
void launch_gmean_bwd(const float* x, const float* dy, float* dx, int accumulate, int NB, int HW, int C,
                      int pre_relu, hipStream_t s);
// separable resize driven by host-built tap tables (TF legacy bicubic and its transpose):
// y[n][oy][ox][c] (+)= sum_a sum_b ty_w[oy][a]*tx_w[ox][b]*x[n][ty_i[oy][a]][tx_i[ox][b]][c]
void launch_resize(const float* x, float* y, int accumulate, int NB, int IH, int IW, int OH, int OW, int C,
                   const int* ty_i, const float* ty_w, int tyn, const int* tx_i, const float* tx_w, int txn,
                   hipStream_t s);
// prelu / fixed leak: y = max(leak*x, x); leak read from device scalar (leak_p) or constant
// x_slabs (float4 form only): x is that SlabSrc; xsum != null also stores the summed x (what prelu_bwd reads);
// y == null (float4 form, y_bf given): only the bf16 plane of the output is stored
void launch_prelu_fwd(const float* x, const float* leak_p, float leak_c, float* y, size_t n, hipStream_t s,
                      unsigned short* y_bf = nullptr, const SlabSrc* x_slabs = nullptr, float* xsum = nullptr);
// dx (+)= ..., dleak (+)= sum(dy*x [x<0 or tie]) (when dleak != null)
size_t prelu_scratch_floats();
void launch_prelu_bwd(const float* x, const float* dy, const float* leak_p, float leak_c, float* dx, int acc_dx,
                      float* dleak, int acc_dleak, size_t n, float* scratch, hipStream_t s, unsigned short* dx_bf = nullptr,
                      const SlabSrc* dy_slabs = nullptr, float* colpart = nullptr, int C = 0);
int flat_colpart_rows(size_t n, int C);   // prelu_bwd / gate_prelu_bwd (their float4 forms): partial rows for [..][C] tensors of n elements, 0 = no
// per-(n,c) spatial min-max normalisation (conv.py:197-198) fwd/bwd; mm[n][c] = {min,max}
// pre_leak != 0: y = minmax(lrelu_{pre_leak}(x)) with the lrelu applied on load / its slope on the gradient store
// x_slabs (C % 8 == 0 only): x is that SlabSrc and the summed value is stored to xsum (what the backward reads)
void launch_minmax_fwd(const float* x, float* y, float* mm, int NB, int HW, int C, hipStream_t s, float pre_leak = 0.f,
                       const SlabSrc* x_slabs = nullptr, float* xsum = nullptr, float* chunk_scratch = nullptr);
void launch_minmax_bwd(const float* x, const float* mm, const float* dy, float* dx, int accumulate, int NB,
                       int HW, int C, hipStream_t s, float pre_leak = 0.f, unsigned short* dx_bf = nullptr, float* colpart = nullptr,
                       float* chunk_scratch = nullptr);
int minmax_bwd_colpart_rows(int NB, int HW, int C);
// chunk_scratch: minmax_chunk_scratch_floats(NB, HW, C) floats of the caller's; with it, maps of more than 4,096 pixels
// (minmax_chunked_shape) take the chunked two-stage form (no column partials there: minmax_bwd_colpart_rows returns 0)
bool minmax_chunked_shape(int NB, int HW, int C);
size_t minmax_chunk_scratch_floats(int NB, int HW, int C);
// gate merge: out = h + rg*img ; backward
void launch_gate_fwd(const float* h, const float* rg, const float* img, float* out, size_t n, hipStream_t s);
void launch_gate_bwd(const float* rg, const float* img, const float* dout, float* dh, int acc_h, float* drg,
                     float* dimg, size_t n, hipStream_t s);
// out = prelu(h + rg*img, *leak) ; backward with the pre-activation recomputed; dleak (+)= sum(dout * pre [leak*pre >= pre])
void launch_gate_prelu_fwd(const float* h, const float* rg, const float* img, const float* leak, float* out, size_t n,
                           hipStream_t s, unsigned short* out_bf = nullptr);
void launch_gate_prelu_bwd(const float* h, const float* rg, const float* img, const float* leak, const float* dout, float* dh,
                           int acc_h, float* drg, float* dimg, float* dleak, int acc_dleak, size_t n, float* scratch,
                           hipStream_t s, const SlabSrc* dout_slabs = nullptr, float* colpart_img = nullptr, int C = 0);
// add bias over last dim in place (y[r][c] += b[c]), optional prelu/lrelu fused variants are separate ops
void launch_bias_add(float* y, const float* b, size_t R, int C, hipStream_t s);

// N==1 linear head: y[r] = dot(x[r], w) + b ; backward pieces
void launch_gemv_fwd(const float* x, const float* w, const float* b, float* y, int R, int K, hipStream_t s);
void launch_gemv_bwd_x(const float* dy, const float* w, float* dx, int accumulate, int R, int K, hipStream_t s);
void launch_gemv_bwd_w(const float* x, const float* dy, float* dw, int acc_w, float* db, int acc_b, int R,
                       int K, hipStream_t s);

// ---- spectral norm (normalization.py:38-76), batched over a table of weights -----------------
constexpr int SN_SCAL_FLOATS = 8 + 256;
struct SNDesc {
  const float* W;   // [K][Co]
  const float* u;   // [Co]
  float* Wbar;      // [K][Co] = W/sigma
  float* t;         // [K]  scratch: W u^T  (then v)
  float* sv;        // [sn_sv_floats(Co)]: v W in the first Co floats, then the K-split partials
  float* scal;      // [SN_SCAL_FLOATS]: 0 nt, 1 ns, 2 sigma, 3 dot(dWbar,W), 4 P=dot(dv,v), 5 ds coefficient, 8.. the 256 partials of dot(dWbar,W)
  const float* dWbar;  // backward input
  float* dW;        // backward output
  float* dv;        // [K] scratch
  int K, Co;
  int acc_dW;
};
int sn_sv_floats(int Co);
void launch_sn_fwd(const SNDesc* d_table, int count, int maxK, int maxCo, hipStream_t s);
void launch_sn_bwd(const SNDesc* d_table, int count, int maxK, int maxCo, hipStream_t s);

// ---- losses -------------------------------------------------------------------------------------
// scal layout documented in engine.cpp
// WGAN: out[0] = mean(fake) - mean(real) over B
void launch_mean_diff(const float* fake, const float* real, float* out, int B, hipStream_t s);
void launch_mean_neg(const float* x, float* out, int B, hipStream_t s);
// GP: per-sample norm of g (per floats each); v = lambda*(2/B)*(|g|-1)*g/|g| ; gp[0] = lambda*mean((|g|-1)^2)
void launch_gp_v(const float* g, float* v, float* norms, float* gp, int B, size_t per, float lambda,
                 hipStream_t s);
// GP head (quirk N10): s = 1 + sig'(d), seeds
//   mode 0: out[b] = 1 + sig(d)(1-sig(d))                                  (first-order seed)
//   mode 1: dd[b] = s'(d)*td[b] ; dtd[b] = s(d)                            (reverse seeds)
void launch_gp_head(const float* d, const float* td, float* o1, float* o2, int B, int mode, hipStream_t s);
// focal AC loss (functional.py:5-16): logits [B][NC], labels from z last column (float ids)
// mode 0: d-loss on real logits -> loss[0], dlogits ; mode 1: g-loss 0.5*mean(CE) -> loss[0], dlogits
void launch_acgan_loss(const float* logits, const float* zlast, int z_stride, float* loss, float* dlogits, int B,
                       int NC, int mode, float* scratch /*[B]*/, hipStream_t s);
// L1: loss = w*mean|t - r| ; dr = -w/(n)*sign(t-r)
void launch_l1(const float* target, int t_stride, const float* recon, float* loss, float* drecon, int B, int Z,
               float weight, hipStream_t s);
// reparam: z = mu + eps*exp(ls); backward
void launch_reparam_fwd(const float* mu, const float* ls, const float* eps_p, float* z, size_t n, hipStream_t s);
void launch_reparam_bwd(const float* ls, const float* eps_p, const float* dz, float* dmu, float* dls, size_t n,
                        hipStream_t s);
// one-hot concat: out[b] = [z[b][:zd], onehot(int(z[b][zd]))]
void launch_z_onehot(const float* z, float* out, int B, int zd, int NC, hipStream_t s);
void launch_class_onehot(const float* zrec, const int* classes, float* out, int B, int zd, int NC, hipStream_t s);

// RMSProp over a flat range; gscale multiplies g first (DP averaging)
void launch_rmsprop(float* w, const float* g, float* ms, float* mom, size_t n, float lr, float gscale,
                    hipStream_t s);

}  // namespace eg
