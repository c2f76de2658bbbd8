// HBM-bound kernels of the EdgeGAN step: data movement, pooling, legacy-bicubic resize, PReLU,
// min-max gate, spectral norm, losses, RMSProp.  NHWC fp32.  All reductions are fixed-order
// (two-stage, no float atomics) so replicas stay bit-identical under data parallelism.
#include <algorithm>

#include "common.h"
#include "kernels.h"
#include "cq_reduce.h"

namespace eg {

static inline int ew_blocks(size_t n, int cap = 4096) {
  size_t b = (n + 255) / 256;
  return (int)(b > (size_t)cap ? cap : (b < 1 ? 1 : b));
}
#define GRID_STRIDE(i, n) \
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (n); i += (size_t)gridDim.x * blockDim.x)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
// block-wide sum for 256-thread blocks; result valid in thread 0 (and broadcast via LDS)
__device__ __forceinline__ float block_sum(float v, float* sm /*[4]*/) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  return sm[0] + sm[1] + sm[2] + sm[3];
}

// ---------------------------------------------------------------------------------------------
// bf16 plane of a tensor, written by its producer as a second store (bf16 mode: kernels.h, k_conv_bf16p.hip): element quad i4
// of the plane = the quad just stored to the fp32 tensor, rounded to nearest even -- the rounding the staging pass of the
// round-3 kernels applied on the operand's way into LDS (v_cvt_pk_bf16_f32), so the products are the same numbers
__device__ __forceinline__ void st_bf4(unsigned short* __restrict__ bf, size_t i4, const float4& v) {
  typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  const f32x2_ lo = {v.x, v.y}, hi = {v.z, v.w};
  uint2 o;
  o.x = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2_));
  o.y = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2_));
  reinterpret_cast<uint2*>(bf)[i4] = o;
}

__global__ void fill_kernel(float* p, float v, size_t n) { GRID_STRIDE(i, n) p[i] = v; }
__global__ void copy_kernel(const float* __restrict__ a, float* __restrict__ b, size_t n) { GRID_STRIDE(i, n) b[i] = a[i]; }
__global__ void axpy_kernel(const float* __restrict__ x, float* __restrict__ y, float a, int acc, size_t n) {
  GRID_STRIDE(i, n) y[i] = acc ? y[i] + a * x[i] : a * x[i];
}
__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, int relu,
                           size_t n) {
  GRID_STRIDE(i, n) {
    const float v = a[i] + b[i];
    o[i] = (relu && v < 0.f) ? 0.f : v;
  }
}
void launch_fill(float* p, float v, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(fill_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, p, v, n);
}
void launch_copy(const float* a, float* b, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(copy_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, a, b, n);
}
void launch_axpy(const float* x, float* y, float a, int acc, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(axpy_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, x, y, a, acc, n);
}
void launch_add(const float* a, const float* b, float* o, int relu, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(add_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, a, b, o, relu, n);
}

// narrow [NB][H][Ws][C] <-> column window wo of wide [NB][H][Wd][C]
__global__ void copy_cols_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int Ws, int Wd, int wo,
                                 int C, int to_dst, int acc, size_t n) {
  GRID_STRIDE(i, n) {
    const int c = (int)(i % C);
    size_t t = i / C;
    const int w = (int)(t % Ws);
    t /= Ws;  // t = nb*H + h
    const size_t j = (t * Wd + wo + w) * C + c;
    if (to_dst) {
      dst[j] = acc ? dst[j] + src[i] : src[i];
    } else {
      float* s2 = const_cast<float*>(src);
      s2[i] = acc ? s2[i] + dst[j] : dst[j];
    }
  }
}
void launch_copy_cols(const float* src, float* dst, int NB, int H, int Ws, int Wd, int wo, int C, int to_dst,
                      int accumulate, hipStream_t s) {
  const size_t n = (size_t)NB * H * Ws * C;
  hipLaunchKernelGGL(copy_cols_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, src, dst, H, Ws, Wd, wo, C, to_dst,
                     accumulate, n);
}

__global__ void blend_kernel(const float* __restrict__ real, const float* __restrict__ fake,
                             const float* __restrict__ alpha, float* __restrict__ out, size_t per, size_t n) {
  GRID_STRIDE(i, n) {
    const float a = alpha[i / per];
    out[i] = real[i] + a * (fake[i] - real[i]);
  }
}
void launch_blend(const float* real, const float* fake, const float* alpha, float* out, int NB, size_t per,
                  hipStream_t s) {
  const size_t n = (size_t)NB * per;
  hipLaunchKernelGGL(blend_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, real, fake, alpha, out, per, n);
}

// fold the gradient of a REFLECT-padded tensor back onto the unpadded one (tf.pad REFLECT gradient,
// conv.py:23-26): dx[i] = sum of dpad[P] over every padded index P that mirrors onto i.
__global__ void reflect_fold_kernel(const float* __restrict__ dpad, float* __restrict__ dx, int acc, int H, int W,
                                    int C, int p, size_t n) {
  const int Hp = H + 2 * p, Wp = W + 2 * p;
  GRID_STRIDE(i, n) {
    const int c = (int)(i % C);
    size_t t = i / C;
    const int x_ = (int)(t % W);
    t /= W;
    const int y_ = (int)(t % H);
    const size_t nb = t / H;
    int ys[3], xs[3], ny = 0, nx = 0;
    ys[ny++] = y_ + p;
    if (y_ >= 1 && y_ <= p) ys[ny++] = p - y_;
    if (y_ <= H - 2 && y_ >= H - 1 - p) ys[ny++] = 2 * (H - 1) - y_ + p;
    xs[nx++] = x_ + p;
    if (x_ >= 1 && x_ <= p) xs[nx++] = p - x_;
    if (x_ <= W - 2 && x_ >= W - 1 - p) xs[nx++] = 2 * (W - 1) - x_ + p;
    float s = 0.f;
    for (int a = 0; a < ny; ++a)
      for (int b = 0; b < nx; ++b) s += dpad[((nb * Hp + ys[a]) * Wp + xs[b]) * C + c];
    dx[i] = acc ? dx[i] + s : s;
  }
}
void launch_reflect_fold(const float* dpad, float* dx, int accumulate, int NB, int H, int W, int C, int p,
                         hipStream_t s) {
  const size_t n = (size_t)NB * H * W * C;
  hipLaunchKernelGGL(reflect_fold_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, dpad, dx, accumulate, H, W, C, p, n);
}

// ---- column sums (bias gradients): two-stage, fixed order -------------------------------------
#define CS_RB 256
__global__ __launch_bounds__(256) void col_sum_stage1(const float* __restrict__ dy, float* __restrict__ part, size_t R,
                                                      int C) {
  __shared__ float red[4 * 64];
  const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  const size_t rows_per = (R + gridDim.y - 1) / gridDim.y;
  const size_t r0 = blockIdx.y * rows_per, r1 = (r0 + rows_per < R) ? r0 + rows_per : R;
  float s = 0.f;
  if (c < C)
    for (size_t r = r0 + py; r < r1; r += 4) s += dy[r * C + c];
  red[py * 64 + cx] = s;
  __syncthreads();
  if (py == 0 && c < C) part[(size_t)blockIdx.y * C + c] = red[cx] + red[64 + cx] + red[128 + cx] + red[192 + cx];
}
__global__ __launch_bounds__(256) void col_sum_stage2(const float* __restrict__ part, float* __restrict__ db, int acc,
                                                      int C, int nrb) {
  __shared__ float red[4 * 64];
  const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  float s = 0.f;
  if (c < C)
    for (int b = py; b < nrb; b += 4) s += part[(size_t)b * C + c];
  red[py * 64 + cx] = s;
  __syncthreads();
  if (py == 0 && c < C) {
    const float t = (red[cx] + red[64 + cx]) + (red[128 + cx] + red[192 + cx]);
    db[c] = acc ? db[c] + t : t;
  }
}
// float4 variant: block = 8 channel quads x 32 row lanes, more row blocks
__global__ __launch_bounds__(256) void col_sum_stage1_v(const float* __restrict__ dy, float* __restrict__ part, size_t R,
                                                        int C) {
  __shared__ float4 red[4 * 8];
  const int cq = threadIdx.x % 8, pl = threadIdx.x / 8;
  const int c = (blockIdx.x * 8 + cq) * 4;
  const size_t rows_per = (R + gridDim.y - 1) / gridDim.y;
  const size_t r0 = blockIdx.y * rows_per, r1 = (r0 + rows_per < R) ? r0 + rows_per : R;
  float4 s[1] = {f4_zero()};
  for (size_t r = r0 + pl; r < r1; r += 32) {
    const float4 v = ld4(dy + r * C + c);
    s[0].x += v.x; s[0].y += v.y; s[0].z += v.z; s[0].w += v.w;
  }
  cq_reduce<1, 8>(s, red);
  if (pl == 0) st4(part + (size_t)blockIdx.y * C + c, s[0]);
}
// narrow form (C <= 128, not a multiple of 32: the 3 / 8 / 14 / 100-channel biases): the block walks its rows as one
// flat stream with a stride that is a multiple of C, so every thread stays on one channel and all lanes load
__global__ __launch_bounds__(256) void col_sum_stage1_narrow(const float* __restrict__ dy, float* __restrict__ part,
                                                             size_t R, int C) {
  __shared__ float red[256];
  const int S = (256 / C) * C;
  const size_t rows_per = (R + gridDim.x - 1) / gridDim.x;
  const size_t r0 = blockIdx.x * rows_per, r1 = (r0 + rows_per < R) ? r0 + rows_per : R;
  float s0 = 0.f, s1 = 0.f;
  if ((int)threadIdx.x < S && r0 < r1) {
    const size_t e1 = r1 * C;
    size_t e = r0 * C + threadIdx.x;
    for (; e + S < e1; e += 2 * (size_t)S) {
      s0 += dy[e];
      s1 += dy[e + S];
    }
    if (e < e1) s0 += dy[e];
  }
  red[threadIdx.x] = s0 + s1;
  __syncthreads();
  if ((int)threadIdx.x < C) {
    float t = 0.f;
    for (int i = threadIdx.x; i < S; i += C) t += red[i];
    part[(size_t)blockIdx.x * C + threadIdx.x] = t;
  }
}
static int col_sum_row_blocks(size_t R) {
  size_t nrb = (R + 255) / 256;
  return nrb > CS_RB ? CS_RB : (nrb < 1 ? 1 : (int)nrb);
}
size_t col_sum_scratch_floats(size_t R, int C) { return (size_t)col_sum_row_blocks(R) * C; }
size_t col_sum_partial_floats(size_t R, int C) { return col_sum_scratch_floats(R, C); }
int launch_col_sum_stage1(const float* dy, size_t R, int C, float* scratch, hipStream_t s) {
  const int nrb = col_sum_row_blocks(R);
  if (C % 32 == 0 && (((uintptr_t)dy & 15) == 0))
    hipLaunchKernelGGL(col_sum_stage1_v, dim3(C / 32, nrb), dim3(256), 0, s, dy, scratch, R, C);
  else if (C <= 128)
    hipLaunchKernelGGL(col_sum_stage1_narrow, dim3(nrb), dim3(256), 0, s, dy, scratch, R, C);
  else
    hipLaunchKernelGGL(col_sum_stage1, dim3((C + 63) / 64, nrb), dim3(256), 0, s, dy, scratch, R, C);
  return nrb;
}
void launch_col_sum(const float* dy, float* db, int accumulate, size_t R, int C, float* scratch, hipStream_t s) {
  const int nrb = launch_col_sum_stage1(dy, R, C, scratch, s);
  hipLaunchKernelGGL(col_sum_stage2, dim3((C + 63) / 64), dim3(256), 0, s, scratch, db, accumulate, C, nrb);
}

__global__ void bias_add_kernel(float* __restrict__ y, const float* __restrict__ b, int C, size_t n) {
  GRID_STRIDE(i, n) y[i] += b[i % C];
}
void launch_bias_add(float* y, const float* b, size_t R, int C, hipStream_t s) {
  const size_t n = R * C;
  hipLaunchKernelGGL(bias_add_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, y, b, C, n);
}

// ---- 2x2 mean pool (pooling.py:4-8; tf.nn.avg_pool 2x2 on even extents, encoder.py:68) ---------
__global__ void pool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C, size_t n) {
  const int Ho = H / 2, Wo = W / 2;
  GRID_STRIDE(i, n) {
    const int c = (int)(i % C);
    size_t t = i / C;
    const int ox = (int)(t % Wo);
    t /= Wo;
    const int oy = (int)(t % Ho);
    const size_t nb = t / Ho;
    const float* p = x + ((nb * H + 2 * oy) * W + 2 * ox) * C + c;
    // add_n order of pooling.py: [::2,::2] + [1::2,::2] + [::2,1::2] + [1::2,1::2]
    y[i] = (p[0] + p[(size_t)W * C] + p[C] + p[(size_t)W * C + C]) * 0.25f;
  }
}
__global__ void pool2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int acc, int H, int W, int C,
                                 size_t n) {
  const int Ho = H / 2, Wo = W / 2;
  GRID_STRIDE(i, n) {  // over input elements
    const int c = (int)(i % C);
    size_t t = i / C;
    const int x_ = (int)(t % W);
    t /= W;
    const int y_ = (int)(t % H);
    const size_t nb = t / H;
    const float g = 0.25f * dy[((nb * Ho + y_ / 2) * Wo + x_ / 2) * C + c];
    dx[i] = acc ? dx[i] + g : g;
  }
}
// float4 forms (C % 4 == 0, < 2^31 elements): one thread per OUTPUT quad, 32-bit index arithmetic
__global__ void pool2_fwd_v_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C4, unsigned n4) {
  const int Ho = H / 2, Wo = W / 2;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n4; i += gridDim.x * 256u) {
    const unsigned c = i % C4;
    unsigned t = i / C4;
    const unsigned ox = t % Wo;
    t /= Wo;
    const unsigned oy = t % Ho, nb = t / Ho;
    const float4* p = reinterpret_cast<const float4*>(x) + ((size_t)(nb * H + 2 * oy) * W + 2 * ox) * C4 + c;
    const float4 a = p[0], b = p[(size_t)W * C4], d = p[C4], e = p[(size_t)W * C4 + C4];
    reinterpret_cast<float4*>(y)[i] = make_float4((a.x + b.x + d.x + e.x) * 0.25f, (a.y + b.y + d.y + e.y) * 0.25f,
                                                  (a.z + b.z + d.z + e.z) * 0.25f, (a.w + b.w + d.w + e.w) * 0.25f);
  }
}
__global__ void pool2_bwd_v_kernel(const SlabSrc dy, float* __restrict__ dx, int acc, int H, int W, int C4,
                                   unsigned n4, unsigned short* __restrict__ dx_bf) {
  const int Ho = H / 2, Wo = W / 2;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n4; i += gridDim.x * 256u) {
    const unsigned c = i % C4;
    unsigned t = i / C4;
    const unsigned ox = t % Wo;
    t /= Wo;
    const unsigned oy = t % Ho, nb = t / Ho;
    const float4 g0 = slab_ld4(dy, (size_t)i * 4, (int)c * 4);
    const float4 g = make_float4(0.25f * g0.x, 0.25f * g0.y, 0.25f * g0.z, 0.25f * g0.w);
    float4* p = reinterpret_cast<float4*>(dx) + ((size_t)(nb * H + 2 * oy) * W + 2 * ox) * C4 + c;
    const size_t offs[4] = {0, (size_t)C4, (size_t)W * C4, (size_t)W * C4 + C4};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float4 v = g;
      if (acc) {
        const float4 o = p[offs[k]];
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
      }
      p[offs[k]] = v;
      if (dx_bf) st_bf4(dx_bf, (size_t)(p - reinterpret_cast<float4*>(dx)) + offs[k], v);
    }
  }
}
// y = pool2(relu(a + b)) (+ its bf16 plane): the tail of a residual block (conv.py:80-85 + encoder.py:63-66) of a forward-only
// graph in one pass -- add_kernel's sum and ReLU, pool2_fwd_v_kernel's summation order, cvt_bf16's rounding: the same bits as the
// three launches it replaces
__global__ void add_relu_pool2_fwd_v_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y,
                                            unsigned short* __restrict__ y_bf, int H, int W, int C4, unsigned n4) {
  const int Ho = H / 2, Wo = W / 2;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n4; i += gridDim.x * 256u) {
    const unsigned c = i % C4;
    unsigned t = i / C4;
    const unsigned ox = t % Wo;
    t /= Wo;
    const unsigned oy = t % Ho, nb = t / Ho;
    const size_t base = ((size_t)(nb * H + 2 * oy) * W + 2 * ox) * C4 + c;
    const float4* pa = reinterpret_cast<const float4*>(a) + base;
    const float4* pb = reinterpret_cast<const float4*>(b) + base;
    const size_t o1 = (size_t)W * C4, o2 = C4, o3 = (size_t)W * C4 + C4;
    const float4 a0 = pa[0], a1 = pa[o1], a2 = pa[o2], a3 = pa[o3];
    const float4 b0 = pb[0], b1 = pb[o1], b2 = pb[o2], b3 = pb[o3];
    auto rl = [](float u, float v) {
      const float w = u + v;
      return w < 0.f ? 0.f : w;
    };
    float4 r;
    r.x = (rl(a0.x, b0.x) + rl(a1.x, b1.x) + rl(a2.x, b2.x) + rl(a3.x, b3.x)) * 0.25f;
    r.y = (rl(a0.y, b0.y) + rl(a1.y, b1.y) + rl(a2.y, b2.y) + rl(a3.y, b3.y)) * 0.25f;
    r.z = (rl(a0.z, b0.z) + rl(a1.z, b1.z) + rl(a2.z, b2.z) + rl(a3.z, b3.z)) * 0.25f;
    r.w = (rl(a0.w, b0.w) + rl(a1.w, b1.w) + rl(a2.w, b2.w) + rl(a3.w, b3.w)) * 0.25f;
    reinterpret_cast<float4*>(y)[i] = r;
    if (y_bf) st_bf4(y_bf, i, r);
  }
}
bool launch_add_relu_pool2_fwd(const float* a, const float* b, float* y, unsigned short* y_bf, int NB, int H, int W, int C, hipStream_t s) {
  const size_t n = (size_t)NB * (H / 2) * (W / 2) * C;
  if (C % 4 != 0 || n >= (1ull << 31) || ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)y)) & 15) != 0) return false;
  hipLaunchKernelGGL(add_relu_pool2_fwd_v_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, s, a, b, y, y_bf, H, W, C / 4, (unsigned)(n / 4));
  return true;
}
// y = pool2(a + b): the element-wise sum first (as the reference does), then pooling.py's add_n order
__global__ void add_pool2_fwd_v_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y,
                                       int H, int W, int C4, unsigned n4) {
  const int Ho = H / 2, Wo = W / 2;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n4; i += gridDim.x * 256u) {
    const unsigned c = i % C4;
    unsigned t = i / C4;
    const unsigned ox = t % Wo;
    t /= Wo;
    const unsigned oy = t % Ho, nb = t / Ho;
    const size_t base = ((size_t)(nb * H + 2 * oy) * W + 2 * ox) * C4 + c;
    const float4* pa = reinterpret_cast<const float4*>(a) + base;
    const float4* pb = reinterpret_cast<const float4*>(b) + base;
    const size_t o1 = (size_t)W * C4, o2 = C4, o3 = (size_t)W * C4 + C4;
    const float4 a0 = pa[0], a1 = pa[o1], a2 = pa[o2], a3 = pa[o3];
    const float4 b0 = pb[0], b1 = pb[o1], b2 = pb[o2], b3 = pb[o3];
    float4 r;
    r.x = ((a0.x + b0.x) + (a1.x + b1.x) + (a2.x + b2.x) + (a3.x + b3.x)) * 0.25f;
    r.y = ((a0.y + b0.y) + (a1.y + b1.y) + (a2.y + b2.y) + (a3.y + b3.y)) * 0.25f;
    r.z = ((a0.z + b0.z) + (a1.z + b1.z) + (a2.z + b2.z) + (a3.z + b3.z)) * 0.25f;
    r.w = ((a0.w + b0.w) + (a1.w + b1.w) + (a2.w + b2.w) + (a3.w + b3.w)) * 0.25f;
    reinterpret_cast<float4*>(y)[i] = r;
  }
}
// y = pool2(a) + b, b already at the pooled resolution (the MRU unit's output with the 1x1 projection evaluated on the
// pooled hidden state: ops.cpp pool2_add)
// a may still be the K-split slabs of the convolution that produced it (SlabSrc); lkp != null: y2 = prelu(y, *lkp) and its
// bf16 plane are stored too (the next MRU unit's prelu(h): conv.py:185, activation.py:23-27)
__global__ void pool2_add_fwd_v_kernel(const SlabSrc a, const float* __restrict__ b, float* __restrict__ y,
                                       int H, int W, int C4, unsigned n4, const float* __restrict__ lkp, float* __restrict__ y2,
                                       unsigned short* __restrict__ y2_bf) {
  const int Ho = H / 2, Wo = W / 2;
  const float lk = lkp ? *lkp : 0.f;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n4; i += gridDim.x * 256u) {
    const unsigned c = i % C4;
    unsigned t = i / C4;
    const unsigned ox = t % Wo;
    t /= Wo;
    const unsigned oy = t % Ho, nb = t / Ho;
    const size_t p0 = (((size_t)(nb * H + 2 * oy) * W + 2 * ox) * C4 + c) * 4;
    const size_t o1 = (size_t)W * C4 * 4, o2 = (size_t)C4 * 4, o3 = o1 + o2;
    const float4 a0 = slab_ld4(a, p0, (int)c * 4), a1 = slab_ld4(a, p0 + o1, (int)c * 4), a2 = slab_ld4(a, p0 + o2, (int)c * 4),
                 a3 = slab_ld4(a, p0 + o3, (int)c * 4), q = reinterpret_cast<const float4*>(b)[i];
    const float4 r = make_float4((a0.x + a1.x + a2.x + a3.x) * 0.25f + q.x, (a0.y + a1.y + a2.y + a3.y) * 0.25f + q.y,
                                 (a0.z + a1.z + a2.z + a3.z) * 0.25f + q.z, (a0.w + a1.w + a2.w + a3.w) * 0.25f + q.w);
    reinterpret_cast<float4*>(y)[i] = r;
    if (lkp) {
      const float4 o = make_float4(fmaxf(lk * r.x, r.x), fmaxf(lk * r.y, r.y), fmaxf(lk * r.z, r.z), fmaxf(lk * r.w, r.w));
      if (y2) reinterpret_cast<float4*>(y2)[i] = o;
      if (y2_bf) st_bf4(y2_bf, i, o);
    }
  }
}
bool launch_pool2_add_fwd(const float* a, const float* b, float* y, int NB, int H, int W, int C, hipStream_t s,
                          const SlabSrc* a_slabs, const float* leak, float* y2, unsigned short* y2_bf) {
  const size_t n = (size_t)NB * (H / 2) * (W / 2) * C;
  if (!(C % 4 == 0 && H % 2 == 0 && W % 2 == 0 && (size_t)NB * H * W * C < (1ull << 31) &&
        ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)y) | ((uintptr_t)y2)) & 15) == 0)) {
    EG_CHECK(!a_slabs && !leak, "pool2_add: slab input / second output requested from the two-pass form");
    return false;
  }
  hipLaunchKernelGGL(pool2_add_fwd_v_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, s, a_slabs ? *a_slabs : plain_src(a), b, y,
                     H, W, C / 4, (unsigned)(n / 4), leak, y2, y2_bf);
  return true;
}
bool launch_add_pool2_fwd(const float* a, const float* b, float* y, int NB, int H, int W, int C, hipStream_t s) {
  const size_t n = (size_t)NB * (H / 2) * (W / 2) * C;
  if (!(C % 4 == 0 && H % 2 == 0 && W % 2 == 0 && (size_t)NB * H * W * C < (1ull << 31) &&
        ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)y)) & 15) == 0))
    return false;
  hipLaunchKernelGGL(add_pool2_fwd_v_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, s, a, b, y, H, W, C / 4,
                     (unsigned)(n / 4));
  return true;
}
void launch_pool2_fwd(const float* x, float* y, int NB, int H, int W, int C, hipStream_t s) {
  const size_t n = (size_t)NB * (H / 2) * (W / 2) * C;
  if (C % 4 == 0 && (size_t)NB * H * W * C < (1ull << 31) && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0) {
    hipLaunchKernelGGL(pool2_fwd_v_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, s, x, y, H, W, C / 4, (unsigned)(n / 4));
    return;
  }
  hipLaunchKernelGGL(pool2_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, x, y, H, W, C, n);
}
void launch_pool2_bwd(const float* dy, float* dx, int accumulate, int NB, int H, int W, int C, hipStream_t s,
                      unsigned short* dx_bf, const SlabSrc* dy_slabs) {
  const size_t n = (size_t)NB * H * W * C;
  if (C % 4 == 0 && n < (1ull << 31) && H % 2 == 0 && W % 2 == 0 && ((((uintptr_t)dy) | ((uintptr_t)dx)) & 15) == 0) {
    const size_t no4 = n / 16;   // output quads
    hipLaunchKernelGGL(pool2_bwd_v_kernel, dim3(ew_blocks(no4)), dim3(256), 0, s, dy_slabs ? *dy_slabs : plain_src(dy), dx,
                       accumulate, H, W, C / 4, (unsigned)no4, dx_bf);
    return;
  }
  EG_CHECK(!dx_bf && !dy_slabs, "pool2_bwd: a bf16 plane / a slab input was requested from the scalar form");
  hipLaunchKernelGGL(pool2_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, dy, dx, accumulate, H, W, C, n);
}

// ---- global spatial mean (encoder.py:69-70 relu + avg_pool 8x8 SAME on 2x2; classifier.py:111) --
__global__ void gmean_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int HW, int C, int pre_relu,
                                 size_t n) {
  GRID_STRIDE(i, n) {
    const int c = (int)(i % C);
    const size_t nb = i / C;
    float s = 0.f;
    for (int p = 0; p < HW; ++p) {
      float v = x[(nb * HW + p) * C + c];
      if (pre_relu && v < 0.f) v = 0.f;
      s += v;
    }
    y[i] = s / HW;
  }
}
__global__ void gmean_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                 int acc, int HW, int C, int pre_relu, size_t n) {
  GRID_STRIDE(i, n) {  // over x elements
    const int c = (int)(i % C);
    const size_t nb = i / ((size_t)HW * C);
    float g = dy[nb * C + c] / HW;
    if (pre_relu && !(x[i] > 0.f)) g = 0.f;
    dx[i] = acc ? dx[i] + g : g;
  }
}
void launch_gmean_fwd(const float* x, float* y, int NB, int HW, int C, int pre_relu, hipStream_t s) {
  const size_t n = (size_t)NB * C;
  hipLaunchKernelGGL(gmean_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, x, y, HW, C, pre_relu, n);
}
void launch_gmean_bwd(const float* x, const float* dy, float* dx, int accumulate, int NB, int HW, int C,
                      int pre_relu, hipStream_t s) {
  const size_t n = (size_t)NB * HW * C;
  hipLaunchKernelGGL(gmean_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, x, dy, dx, accumulate, HW, C, pre_relu, n);
}

// ---- separable resize by host-built tap tables (TF legacy bicubic, edgegan.py:211-213) ---------
// table row o: `taps` entries (index, weight); rows padded with weight 0.
__global__ void resize_kernel(const float* __restrict__ x, float* __restrict__ y, int acc, int IH, int IW, int OH,
                              int OW, int C, const int* __restrict__ ty_i, const float* __restrict__ ty_w, int tyn,
                              const int* __restrict__ tx_i, const float* __restrict__ tx_w, int txn, size_t n) {
  GRID_STRIDE(i, n) {
    const int c = (int)(i % C);
    size_t t = i / C;
    const int ox = (int)(t % OW);
    t /= OW;
    const int oy = (int)(t % OH);
    const size_t nb = t / OH;
    float s = 0.f;
    for (int a = 0; a < tyn; ++a) {
      const float wy = ty_w[oy * tyn + a];
      if (wy == 0.f) continue;
      const float* row = x + ((nb * IH + ty_i[oy * tyn + a]) * IW) * C + c;
      float rs = 0.f;
      for (int b = 0; b < txn; ++b) {
        const float wx = tx_w[ox * txn + b];
        if (wx != 0.f) rs += wx * row[(size_t)tx_i[ox * txn + b] * C];
      }
      s += wy * rs;
    }
    y[i] = acc ? y[i] + s : s;
  }
}
// 3-channel images (every call of the reference): one thread per output PIXEL, so the tap tables and the index
// arithmetic are read once for the three channels
__global__ void resize3_kernel(const float* __restrict__ x, float* __restrict__ y, int acc, int IH, int IW, int OH,
                               int OW, const int* __restrict__ ty_i, const float* __restrict__ ty_w, int tyn,
                               const int* __restrict__ tx_i, const float* __restrict__ tx_w, int txn, size_t npix) {
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < npix; i += gridDim.x * 256u) {   // 32-bit index arithmetic
    const int ox = (int)(i % (unsigned)OW);
    unsigned t = i / (unsigned)OW;
    const int oy = (int)(t % (unsigned)OH);
    const size_t nb = t / (unsigned)OH;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int a = 0; a < tyn; ++a) {
      const float wy = ty_w[oy * tyn + a];
      if (wy == 0.f) continue;
      const float* row = x + ((nb * IH + ty_i[oy * tyn + a]) * IW) * 3;
      float r0 = 0.f, r1 = 0.f, r2 = 0.f;
      for (int b = 0; b < txn; ++b) {
        const float wx = tx_w[ox * txn + b];
        if (wx != 0.f) {
          const float* px = row + (size_t)tx_i[ox * txn + b] * 3;
          r0 += wx * px[0];
          r1 += wx * px[1];
          r2 += wx * px[2];
        }
      }
      s0 += wy * r0;
      s1 += wy * r1;
      s2 += wy * r2;
    }
    float* o = y + (size_t)i * 3;
    if (acc) {
      s0 += o[0];
      s1 += o[1];
      s2 += o[2];
    }
    o[0] = s0;
    o[1] = s1;
    o[2] = s2;
  }
}
// the same sums with the column taps in registers and no branch in the tap loops (round 6): the table loads of a pixel are
// tyn + TN index / weight pairs instead of tyn (1 + txn), and the 3 TN pixel loads of a row are issued back to back (the looped
// form above branched on every zero weight: one dependent load chain per tap, 29 us for 1 M output pixels).  A tap of weight 0
// (table padding; its index is a valid pixel) contributes nothing through a select, in the order of the looped form: same bits.
template <int TN>
__global__ __launch_bounds__(256) void resize3_taps_kernel(const float* __restrict__ x, float* __restrict__ y, int acc, int IH, int IW,
                                                           int OH, int OW, const int* __restrict__ ty_i, const float* __restrict__ ty_w,
                                                           int tyn, const int* __restrict__ tx_i, const float* __restrict__ tx_w, int txn,
                                                           size_t npix) {
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < npix; i += gridDim.x * 256u) {
    const int ox = (int)(i % (unsigned)OW);
    unsigned t = i / (unsigned)OW;
    const int oy = (int)(t % (unsigned)OH);
    const size_t nb = t / (unsigned)OH;
    float wx[TN];
    int ix[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const bool vb = b < txn;
      wx[b] = vb ? tx_w[ox * txn + b] : 0.f;
      ix[b] = vb ? tx_i[ox * txn + b] * 3 : 0;
    }
    const float* img = x + nb * IH * IW * 3;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int a = 0; a < tyn; ++a) {
      const float wy = ty_w[oy * tyn + a];
      if (wy == 0.f) continue;   // (a wave's pixels share their row: uniform in practice)
      const float* row = img + (size_t)ty_i[oy * tyn + a] * IW * 3;
      float r0 = 0.f, r1 = 0.f, r2 = 0.f;
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const float* px = row + ix[b];
        const float p0 = px[0], p1 = px[1], p2 = px[2];
        const bool on = wx[b] != 0.f;
        r0 = on ? r0 + wx[b] * p0 : r0;
        r1 = on ? r1 + wx[b] * p1 : r1;
        r2 = on ? r2 + wx[b] * p2 : r2;
      }
      s0 += wy * r0;
      s1 += wy * r1;
      s2 += wy * r2;
    }
    float* o = y + (size_t)i * 3;
    if (acc) {
      s0 += o[0];
      s1 += o[1];
      s2 += o[2];
    }
    o[0] = s0;
    o[1] = s1;
    o[2] = s2;
  }
}
void launch_resize(const float* x, float* y, int accumulate, int NB, int IH, int IW, int OH, int OW, int C,
                   const int* ty_i, const float* ty_w, int tyn, const int* tx_i, const float* tx_w, int txn,
                   hipStream_t s) {
  if (C == 3 && (size_t)NB * OH * OW < (1ull << 31)) {
    const size_t npix = (size_t)NB * OH * OW;
#define EG_RESIZE3(TN)                                                                                                        \
  hipLaunchKernelGGL(resize3_taps_kernel<TN>, dim3(ew_blocks(npix)), dim3(256), 0, s, x, y, accumulate, IH, IW, OH, OW, ty_i, ty_w, \
                     tyn, tx_i, tx_w, txn, npix)
    if (txn <= 4) EG_RESIZE3(4);
    else if (txn <= 8) EG_RESIZE3(8);
    else if (txn <= 12) EG_RESIZE3(12);
    else
      hipLaunchKernelGGL(resize3_kernel, dim3(ew_blocks(npix)), dim3(256), 0, s, x, y, accumulate, IH, IW, OH, OW, ty_i,
                         ty_w, tyn, tx_i, tx_w, txn, npix);
#undef EG_RESIZE3
    return;
  }
  const size_t n = (size_t)NB * OH * OW * C;
  hipLaunchKernelGGL(resize_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, x, y, accumulate, IH, IW, OH, OW, C, ty_i,
                     ty_w, tyn, tx_i, tx_w, txn, n);
}

// ---- PReLU / fixed-leak lrelu (activation.py:23-32): y = max(leak*x, x) ------------------------
__global__ void prelu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ leak_p, float leak_c,
                                 float* __restrict__ y, size_t n) {
  const float lk = leak_p ? *leak_p : leak_c;
  GRID_STRIDE(i, n) {
    const float v = x[i];
    y[i] = fmaxf(lk * v, v);
  }
}
__global__ __launch_bounds__(256) void prelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                        const float* __restrict__ leak_p, float leak_c,
                                                        float* __restrict__ dx, int acc_dx, float* __restrict__ part,
                                                        size_t n) {
  __shared__ float sm[4];
  const float lk = leak_p ? *leak_p : leak_c;
  float s = 0.f;
  GRID_STRIDE(i, n) {
    const float v = x[i], g = dy[i];
    const bool first = (lk * v >= v);  // tf.maximum sends the gradient to the first argument on ties
    if (dx) {
      const float gx = first ? lk * g : g;
      dx[i] = acc_dx ? dx[i] + gx : gx;
    }
    if (first) s += g * v;
  }
  if (part) {
    const float tot = block_sum(s, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
  }
}
// Column partials of a flat float4 kernel over an [..][C] tensor (kernels.h: colpart): with C4 = C / 4 a divisor of 256 and a
// grid stride that is a multiple of 256 quads, thread t only ever touches channel quad t % C4; the block sums its threads'
// running sums per quad (fixed order) and stores row blockIdx.x of the [blocks][C] table
__device__ __forceinline__ void flat_colpart_store(const float4& cs, float4* redc /*[256]*/, float* __restrict__ colpart, int C4) {
  redc[threadIdx.x] = cs;
  __syncthreads();
  if ((int)threadIdx.x < C4) {
    float4 t = redc[threadIdx.x];
    for (int j = threadIdx.x + C4; j < 256; j += C4) {
      const float4 v = redc[j];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    reinterpret_cast<float4*>(colpart)[(size_t)blockIdx.x * C4 + threadIdx.x] = t;
  }
}
constexpr int kFlatColpartBlocks = 512;   // rows of the table = blocks of the launch (second stage: 128 adds per lane)
int flat_colpart_rows(size_t n, int C) {
  if (C % 4 != 0 || 256 % (C / 4) != 0 || (n & 3) != 0 || n / 4 < 256) return 0;
  const size_t b = (n / 4 + 255) / 256;
  return (int)(b > (size_t)kFlatColpartBlocks ? kFlatColpartBlocks : b);
}

// float4 form (n % 4 == 0, 16-byte aligned buffers)
__global__ __launch_bounds__(256) void prelu_bwd_v_kernel(const float* __restrict__ x, const SlabSrc dy,
                                                          const float* __restrict__ leak_p, float leak_c,
                                                          float* __restrict__ dx, int acc_dx, float* __restrict__ part,
                                                          size_t n4, unsigned short* __restrict__ dx_bf,
                                                          float* __restrict__ colpart, int C4) {
  __shared__ float sm[4];
  __shared__ float4 redc[256];
  const float lk = leak_p ? *leak_p : leak_c;
  float s = 0.f;
  float4 cs = f4_zero();
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i], g = slab_ld4_flat(dy, i);
    const bool f0 = lk * v.x >= v.x, f1 = lk * v.y >= v.y, f2 = lk * v.z >= v.z, f3 = lk * v.w >= v.w;
    if (dx) {
      float4 o = make_float4(f0 ? lk * g.x : g.x, f1 ? lk * g.y : g.y, f2 ? lk * g.z : g.z, f3 ? lk * g.w : g.w);
      if (acc_dx & 1) {
        const float4 d = reinterpret_cast<const float4*>(dx)[i];
        o.x += d.x; o.y += d.y; o.z += d.z; o.w += d.w;
      }
      if (!(acc_dx & 2)) reinterpret_cast<float4*>(dx)[i] = o;   // (GRAD_PLANE_ONLY)
      if (dx_bf) st_bf4(dx_bf, i, o);
      cs.x += o.x; cs.y += o.y; cs.z += o.z; cs.w += o.w;
    }
    s += (f0 ? g.x * v.x : 0.f) + (f1 ? g.y * v.y : 0.f) + (f2 ? g.z * v.z : 0.f) + (f3 ? g.w * v.w : 0.f);
  }
  if (part) {
    const float tot = block_sum(s, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
  }
  if (colpart) flat_colpart_store(cs, redc, colpart, C4);
}
__global__ void prelu_fwd_v_kernel(const SlabSrc x, const float* __restrict__ leak_p, float leak_c,
                                   float* __restrict__ y, size_t n4, unsigned short* __restrict__ y_bf, float* xsum) {   // (xsum may alias the slabs' base)
  const float lk = leak_p ? *leak_p : leak_c;
  GRID_STRIDE(i, n4) {
    const float4 v = slab_ld4_flat(x, i);
    if (xsum) reinterpret_cast<float4*>(xsum)[i] = v;
    const float4 o = make_float4(fmaxf(lk * v.x, v.x), fmaxf(lk * v.y, v.y), fmaxf(lk * v.z, v.z), fmaxf(lk * v.w, v.w));
    if (y) reinterpret_cast<float4*>(y)[i] = o;
    if (y_bf) st_bf4(y_bf, i, o);
  }
}
__global__ void sum_partials_kernel(const float* __restrict__ part, int np, float* __restrict__ out, int acc,
                                    float scale) {
  __shared__ float sm[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < np; i += 256) s += part[i];
  const float tot = block_sum(s, sm);
  if (threadIdx.x == 0) *out = acc ? *out + tot * scale : tot * scale;
}
static bool ew_vec_ok(size_t n, const void* a, const void* b, const void* c = nullptr) {
  return (n & 3) == 0 && ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c)) & 15) == 0;
}
void launch_prelu_fwd(const float* x, const float* leak_p, float leak_c, float* y, size_t n, hipStream_t s, unsigned short* y_bf,
                      const SlabSrc* x_slabs, float* xsum) {
  EG_CHECK(!(y_bf || x_slabs) || ew_vec_ok(n, x, y, xsum), "prelu_fwd: a bf16 plane / a slab input was requested from the scalar form");
  EG_CHECK(y || (y_bf && ew_vec_ok(n, x, y, xsum)), "prelu_fwd: no output");
  if (ew_vec_ok(n, x, y, xsum)) {
    // (experiments: EG_DUP_PRELU_FWD = N extra, idempotent launches -- what does an HBM-bound pass on a lane cost the iteration?)
    static const int dup = dbg_env("EG_DUP_PRELU_FWD") ? atoi(dbg_env("EG_DUP_PRELU_FWD")) : 0;
    for (int r = 0; r <= dup; ++r)
      hipLaunchKernelGGL(prelu_fwd_v_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, s, x_slabs ? *x_slabs : plain_src(x), leak_p,
                         leak_c, y, n / 4, y_bf, xsum);
    return;
  }
  hipLaunchKernelGGL(prelu_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, x, leak_p, leak_c, y, n);
}
size_t prelu_scratch_floats() { return 2048; }
void launch_prelu_bwd(const float* x, const float* dy, const float* leak_p, float leak_c, float* dx, int acc_dx,
                      float* dleak, int acc_dleak, size_t n, float* scratch, hipStream_t s, unsigned short* dx_bf,
                      const SlabSrc* dy_slabs, float* colpart, int C) {
  int nb;
  EG_CHECK(!(acc_dx & GRAD_PLANE_ONLY) || dx_bf, "prelu_bwd: GRAD_PLANE_ONLY without a plane");
  if (ew_vec_ok(n, x, dy, dx)) {
    nb = colpart ? flat_colpart_rows(n, C) : ew_blocks(n / 4, 2048);
    EG_CHECK(nb > 0 && (!colpart || dx), "prelu_bwd: column partials were requested from a form that has none");
    hipLaunchKernelGGL(prelu_bwd_v_kernel, dim3(nb), dim3(256), 0, s, x, dy_slabs ? *dy_slabs : plain_src(dy), leak_p, leak_c, dx,
                       acc_dx, dleak ? scratch : nullptr, n / 4, dx_bf, colpart, colpart ? C / 4 : 1);
  } else {
    EG_CHECK(!dx_bf && !dy_slabs && !colpart, "prelu_bwd: a bf16 plane / a slab input / column partials were requested from the scalar form");
    nb = ew_blocks(n, 1024);
    hipLaunchKernelGGL(prelu_bwd_kernel, dim3(nb), dim3(256), 0, s, x, dy, leak_p, leak_c, dx, acc_dx,
                       dleak ? scratch : nullptr, n);
  }
  if (dleak) hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, scratch, nb, dleak, acc_dleak, 1.0f);
}

// ---- per-(n,c) spatial min-max normalisation of the MRU update gate (conv.py:197-198) ----------
// lk != 0: the fixed-leak lrelu in front of it (conv.py:190-196, activation.py:30-32: maximum(lk x, x)) is applied while
// loading, u = lrelu(x), and its derivative (slope lk where lk x >= x, i.e. also at 0: Appendix B.8) while storing the
// gradient -- lrelu's output is never materialised, its backward is not a pass of its own
__device__ __forceinline__ float mm_pre(float v, float lk) { return lk != 0.f ? fmaxf(lk * v, v) : v; }
__device__ __forceinline__ float mm_slope(float v, float lk) { return (lk != 0.f && lk * v >= v) ? lk : 1.f; }
__device__ __forceinline__ float4 mm_pre4(float4 v, float lk) {
  return make_float4(mm_pre(v.x, lk), mm_pre(v.y, lk), mm_pre(v.z, lk), mm_pre(v.w, lk));
}
__global__ __launch_bounds__(256) void minmax_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         float* __restrict__ mm, int HW, int C, float lk) {
  __shared__ float red[2 * 4 * 64];
  const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
  const int n = blockIdx.x, c = blockIdx.y * 64 + cx;
  const bool cv = c < C;
  const size_t base = (size_t)n * HW * C + c;
  float mn = INFINITY, mx = -INFINITY;
  if (cv)
    for (int p = py; p < HW; p += 4) {
      const float v = mm_pre(x[base + (size_t)p * C], lk);
      mn = fminf(mn, v);
      mx = fmaxf(mx, v);
    }
  red[py * 64 + cx] = mn;
  red[(4 + py) * 64 + cx] = mx;
  __syncthreads();
  mn = fminf(fminf(red[cx], red[64 + cx]), fminf(red[128 + cx], red[192 + cx]));
  mx = fmaxf(fmaxf(red[256 + cx], red[320 + cx]), fmaxf(red[384 + cx], red[448 + cx]));
  if (cv) {
    const float inv = 1.f / (mx - mn);
    for (int p = py; p < HW; p += 4) y[base + (size_t)p * C] = (mm_pre(x[base + (size_t)p * C], lk) - mn) * inv;
    if (py == 0) {
      mm[((size_t)n * C + c) * 2 + 0] = mn;
      mm[((size_t)n * C + c) * 2 + 1] = mx;
    }
  }
}
__global__ __launch_bounds__(256) void minmax_bwd_kernel(const float* __restrict__ x, const float* __restrict__ mm,
                                                         const float* __restrict__ dy, float* __restrict__ dx, int acc,
                                                         int HW, int C, float lk) {
  __shared__ float red[4 * 4 * 64];
  const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
  const int n = blockIdx.x, c = blockIdx.y * 64 + cx;
  const bool cv = c < C;
  const size_t base = (size_t)n * HW * C + c;
  const float mn = cv ? mm[((size_t)n * C + c) * 2] : 0.f, mx = cv ? mm[((size_t)n * C + c) * 2 + 1] : 1.f;
  float s1 = 0.f, s2 = 0.f, cmn = 0.f, cmx = 0.f;
  if (cv)
    for (int p = py; p < HW; p += 4) {
      const float v = mm_pre(x[base + (size_t)p * C], lk), g = dy[base + (size_t)p * C];
      s1 += g;
      s2 += g * (v - mn);
      cmn += (v == mn) ? 1.f : 0.f;
      cmx += (v == mx) ? 1.f : 0.f;
    }
  red[(0 * 4 + py) * 64 + cx] = s1;
  red[(1 * 4 + py) * 64 + cx] = s2;
  red[(2 * 4 + py) * 64 + cx] = cmn;
  red[(3 * 4 + py) * 64 + cx] = cmx;
  __syncthreads();
  float v4[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    v4[k] = red[(k * 4 + 0) * 64 + cx] + red[(k * 4 + 1) * 64 + cx] + red[(k * 4 + 2) * 64 + cx] +
            red[(k * 4 + 3) * 64 + cx];
  if (cv) {
    const float d = mx - mn;
    const float dmn = (-v4[0] / d + v4[1] / (d * d)) / v4[2];   // split equally among ties (Appendix B.8)
    const float dmx = (-v4[1] / (d * d)) / v4[3];
    for (int p = py; p < HW; p += 4) {
      const size_t i = base + (size_t)p * C;
      const float xv = x[i], v = mm_pre(xv, lk);
      float g = dy[i] / d;
      if (v == mn) g += dmn;
      if (v == mx) g += dmx;
      g *= mm_slope(xv, lk);
      dx[i] = acc ? dx[i] + g : g;
    }
  }
}
// xs.n > 1: x is still the K-split slabs of the gate's hidden-state convolution (+ the image-row convolution's output as
// the base): the first pass sums them and stores the sum to xsum, which the second pass (the same thread re-reads its own
// stores) and the backward read
// NW waves per block: 4 where the channel groups alone fill the chip (CQ = 8), 16 for the 8-channel gate of the classifier's
// first unit (64 blocks of 4,096 pixels: at 256 threads each thread walked 32 dependent iterations per pass -- 23 / 45 us)
template <int CQ, int NW>
// (xsum carries no __restrict__: with the FEED_MINMAX hand-over it IS SlabSrc::base, the tensor the slabs are added to)
__global__ __launch_bounds__(64 * NW) void minmax_fwd_v(const SlabSrc xs, float* xsum, float* __restrict__ y,
                                                        float* __restrict__ mm, int HW, int C, float lk) {
  constexpr int PL = 64 * NW / CQ;
  __shared__ float4 red[2 * NW * CQ];
  const int cq = threadIdx.x % CQ, pl = threadIdx.x / CQ;
  const int n = blockIdx.x, c = (blockIdx.y * CQ + cq) * 4;
  const size_t base = (size_t)n * HW * C + c;
  const float* x = xs.n > 1 ? xsum : xs.p;
  float4 mn = make_float4(INFINITY, INFINITY, INFINITY, INFINITY), mx = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  // (round 6: the pixel loops take four pixels per step with their loads issued together -- one memory round trip per FOUR
  //  iterations instead of one per iteration; min / max and the stores are order-independent)
  auto upd = [&](const float4& raw) __attribute__((always_inline)) {
    const float4 v = mm_pre4(raw, lk);
    mn.x = fminf(mn.x, v.x); mn.y = fminf(mn.y, v.y); mn.z = fminf(mn.z, v.z); mn.w = fminf(mn.w, v.w);
    mx.x = fmaxf(mx.x, v.x); mx.y = fmaxf(mx.y, v.y); mx.z = fmaxf(mx.z, v.z); mx.w = fmaxf(mx.w, v.w);
  };
  if (xs.n > 1) {
    for (int p = pl; p < HW; p += PL) {
      const float4 raw = slab_ld4(xs, base + (size_t)p * C, c);
      st4(xsum + base + (size_t)p * C, raw);
      upd(raw);
    }
  } else {
    int p = pl;
    for (; p + 3 * PL < HW; p += 4 * PL) {
      float4 r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) r[u] = ld4(x + base + (size_t)(p + u * PL) * C);
#pragma unroll
      for (int u = 0; u < 4; ++u) upd(r[u]);
    }
    for (; p < HW; p += PL) upd(ld4(x + base + (size_t)p * C));
  }
  cq_reduce_minmax_w<CQ, NW>(mn, mx, red);
  const float4 inv = make_float4(1.f / (mx.x - mn.x), 1.f / (mx.y - mn.y), 1.f / (mx.z - mn.z), 1.f / (mx.w - mn.w));
  auto emit = [&](size_t i, const float4& raw) __attribute__((always_inline)) {
    const float4 v = mm_pre4(raw, lk);
    st4(y + i, make_float4((v.x - mn.x) * inv.x, (v.y - mn.y) * inv.y, (v.z - mn.z) * inv.z, (v.w - mn.w) * inv.w));
  };
  {
    int p = pl;
    for (; p + 3 * PL < HW; p += 4 * PL) {
      float4 r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) r[u] = ld4(x + base + (size_t)(p + u * PL) * C);
#pragma unroll
      for (int u = 0; u < 4; ++u) emit(base + (size_t)(p + u * PL) * C, r[u]);
    }
    for (; p < HW; p += PL) emit(base + (size_t)p * C, ld4(x + base + (size_t)p * C));
  }
  if (pl == 0) {
    float* mp = mm + ((size_t)n * C + c) * 2;
    mp[0] = mn.x; mp[1] = mx.x; mp[2] = mn.y; mp[3] = mx.y; mp[4] = mn.z; mp[5] = mx.z; mp[6] = mn.w; mp[7] = mx.w;
  }
}

template <int CQ, int NW>
__global__ __launch_bounds__(64 * NW) void minmax_bwd_v(const float* __restrict__ x, const float* __restrict__ mm,
                                                        const float* __restrict__ dy, float* __restrict__ dx, int acc,
                                                        int HW, int C, float lk, unsigned short* __restrict__ dx_bf,
                                                        float* __restrict__ colpart) {
  constexpr int PL = 64 * NW / CQ;
  __shared__ float4 red[4 * NW * CQ];
  const int cq = threadIdx.x % CQ, pl = threadIdx.x / CQ;
  const int n = blockIdx.x, c = (blockIdx.y * CQ + cq) * 4;
  const size_t base = (size_t)n * HW * C + c;
  const float4 a = ld4(mm + ((size_t)n * C + c) * 2), b = ld4(mm + ((size_t)n * C + c) * 2 + 4);
  const float mn[4] = {a.x, a.z, b.x, b.z}, mx[4] = {a.y, a.w, b.y, b.w};
  float4 s[4] = {f4_zero(), f4_zero(), f4_zero(), f4_zero()};  // sum g, sum g*(x-mn), count==mn, count==mx
  auto acc1 = [&](const float4& xr, const float4& g) __attribute__((always_inline)) {
    const float4 v = mm_pre4(xr, lk);
    s[0].x += g.x; s[0].y += g.y; s[0].z += g.z; s[0].w += g.w;
    s[1].x += g.x * (v.x - mn[0]); s[1].y += g.y * (v.y - mn[1]); s[1].z += g.z * (v.z - mn[2]); s[1].w += g.w * (v.w - mn[3]);
    s[2].x += v.x == mn[0] ? 1.f : 0.f; s[2].y += v.y == mn[1] ? 1.f : 0.f; s[2].z += v.z == mn[2] ? 1.f : 0.f; s[2].w += v.w == mn[3] ? 1.f : 0.f;
    s[3].x += v.x == mx[0] ? 1.f : 0.f; s[3].y += v.y == mx[1] ? 1.f : 0.f; s[3].z += v.z == mx[2] ? 1.f : 0.f; s[3].w += v.w == mx[3] ? 1.f : 0.f;
  };
  {   // (four pixels per step, their eight loads issued together; the sums keep the one-pixel loop's order)
    int p = pl;
    for (; p + 3 * PL < HW; p += 4 * PL) {
      float4 xr[4], g[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const size_t i = base + (size_t)(p + u * PL) * C;
        xr[u] = ld4(x + i);
        g[u] = ld4(dy + i);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc1(xr[u], g[u]);
    }
    for (; p < HW; p += PL) {
      const size_t i = base + (size_t)p * C;
      acc1(ld4(x + i), ld4(dy + i));
    }
  }
  cq_reduce_w<4, CQ, NW>(s, red);
  const float S1[4] = {s[0].x, s[0].y, s[0].z, s[0].w}, S2[4] = {s[1].x, s[1].y, s[1].z, s[1].w},
              Cn[4] = {s[2].x, s[2].y, s[2].z, s[2].w}, Cx[4] = {s[3].x, s[3].y, s[3].z, s[3].w};
  float d[4], dmn[4], dmx[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    d[j] = mx[j] - mn[j];
    dmn[j] = (-S1[j] / d[j] + S2[j] / (d[j] * d[j])) / Cn[j];
    dmx[j] = (-S2[j] / (d[j] * d[j])) / Cx[j];
  }
  float4 cs[1] = {f4_zero()};
  auto apply2 = [&](size_t i, const float4& xv, const float4& g, const float4& t) __attribute__((always_inline)) {
    const float4 v = mm_pre4(xv, lk);
    const float xx[4] = {xv.x, xv.y, xv.z, xv.w}, vv[4] = {v.x, v.y, v.z, v.w}, gg[4] = {g.x, g.y, g.z, g.w};
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      r[j] = gg[j] / d[j];
      if (vv[j] == mn[j]) r[j] += dmn[j];
      if (vv[j] == mx[j]) r[j] += dmx[j];
      r[j] *= mm_slope(xx[j], lk);
    }
    float4 o = make_float4(r[0], r[1], r[2], r[3]);
    if (acc & 1) { o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w; }
    if (!(acc & 2)) st4(dx + i, o);      // (bit 1 of the accumulate argument: the gradient is kept as its bf16 plane only, kernels.h: GRAD_PLANE_ONLY)
    if (dx_bf) st_bf4(dx_bf, i >> 2, o);
    cs[0].x += o.x; cs[0].y += o.y; cs[0].z += o.z; cs[0].w += o.w;
  };
  {
    int p = pl;
    for (; p + 3 * PL < HW; p += 4 * PL) {
      float4 xr[4], g[4], t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const size_t i = base + (size_t)(p + u * PL) * C;
        xr[u] = ld4(x + i);
        g[u] = ld4(dy + i);
        t[u] = (acc & 1) ? ld4(dx + i) : f4_zero();
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) apply2(base + (size_t)(p + u * PL) * C, xr[u], g[u], t[u]);
    }
    for (; p < HW; p += PL) {
      const size_t i = base + (size_t)p * C;
      apply2(i, ld4(x + i), ld4(dy + i), (acc & 1) ? ld4(dx + i) : f4_zero());
    }
  }
  if (colpart) {   // row n of the gate bias gradient's first stage (kernels.h: colpart)
    cq_reduce_w<1, CQ, NW>(cs, red);
    if (pl == 0) st4(colpart + (size_t)n * C + c, cs[0]);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The min-max gate on LARGE maps (H W > 4,096: configs[4]'s 256 x 512 geometry -- up to 65,536 pixels per map).  The per-(n, c)
// kernels above give a whole map to one block: 16 blocks at batch 16 for the 8-channel gate, 180 us per launch.  Here a map is
// cut into chunks of MMC_ROWS rows; minima / maxima and the backward's four sums come from chunk partials combined per (n, c) in
// chunk order (min / max are order-free, the sums have a fixed order), the apply passes are element-wise over quads.  Same
// formulas and tie split as minmax_fwd_v / minmax_bwd_v.  Block = CG channels x 256 / CG row lanes (CG = 8 or 64).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int MMC_ROWS = 512;
template <int CG>
__global__ __launch_bounds__(256) void mmc_minmax_kernel(const float* __restrict__ x, float* __restrict__ part, int HW, int C, float lk) {
  constexpr int L = 256 / CG;
  __shared__ float red[2 * L * CG];
  const int cx = threadIdx.x % CG, py = threadIdx.x / CG;
  const int c = blockIdx.y * CG + cx, n = blockIdx.z, nch = gridDim.x;
  const bool cv = c < C;
  const int r0 = blockIdx.x * MMC_ROWS, r1 = min(HW, r0 + MMC_ROWS);
  const float* xp = x + (size_t)n * HW * C + c;
  float mn = INFINITY, mx = -INFINITY;
  if (cv)
    for (int p0 = r0 + py; p0 < r1; p0 += 8 * L) {   // eight rows per step, their loads issued together
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int p = p0 + u * L;
        v[u] = xp[(size_t)(p < r1 ? p : r0) * C];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (p0 + u * L < r1) {
          const float t = mm_pre(v[u], lk);
          mn = fminf(mn, t);
          mx = fmaxf(mx, t);
        }
    }
  red[py * CG + cx] = mn;
  red[(L + py) * CG + cx] = mx;
  __syncthreads();
  if (py == 0 && cv) {
    for (int q = 1; q < L; ++q) {
      mn = fminf(mn, red[q * CG + cx]);
      mx = fmaxf(mx, red[(L + q) * CG + cx]);
    }
    float* o = part + (((size_t)n * nch + blockIdx.x) * C + c) * 2;
    o[0] = mn;
    o[1] = mx;
  }
}
__global__ void mmc_minmax_finalize_kernel(const float* __restrict__ part, int nch, int C, float* __restrict__ mm) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
  if (c >= C) return;
  float mn = INFINITY, mx = -INFINITY;
  for (int k0 = 0; k0 < nch; k0 += 8) {   // (eight chunks' partials in flight)
    float2 pv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      pv[u] = *reinterpret_cast<const float2*>(part + (((size_t)n * nch + min(k0 + u, nch - 1)) * C + c) * 2);
#pragma unroll
    for (int u = 0; u < 8; ++u) {   // (a repeated last chunk changes no minimum / maximum)
      mn = fminf(mn, pv[u].x);
      mx = fmaxf(mx, pv[u].y);
    }
  }
  mm[((size_t)n * C + c) * 2] = mn;
  mm[((size_t)n * C + c) * 2 + 1] = mx;
}
__global__ __launch_bounds__(256) void mmc_fwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ mm,
                                                            float* __restrict__ y, int HW, int C, float lk) {
  const int n = blockIdx.y, c4n = C >> 2;
  const size_t per = (size_t)HW * c4n, ib = (size_t)n * HW * C;
  for (size_t q = blockIdx.x * (size_t)256 + threadIdx.x; q < per; q += (size_t)gridDim.x * 256) {
    const int c = (int)(q % c4n) * 4;
    const size_t i = ib + q * 4;
    const float4 a = ld4(mm + ((size_t)n * C + c) * 2), b = ld4(mm + ((size_t)n * C + c) * 2 + 4);
    const float4 v = mm_pre4(ld4(x + i), lk);
    st4(y + i, make_float4((v.x - a.x) * (1.f / (a.y - a.x)), (v.y - a.z) * (1.f / (a.w - a.z)), (v.z - b.x) * (1.f / (b.y - b.x)),
                           (v.w - b.z) * (1.f / (b.w - b.z))));
  }
}
// chunk partials of S1 = sum g, S2 = sum g (v - mn), the tie counts of mn and mx
template <int CG>
__global__ __launch_bounds__(256) void mmc_bwd_sums_kernel(const float* __restrict__ x, const float* __restrict__ mm,
                                                           const float* __restrict__ dy, float* __restrict__ part, int HW, int C,
                                                           float lk) {
  constexpr int L = 256 / CG;
  __shared__ float red[4 * L * CG];
  const int cx = threadIdx.x % CG, py = threadIdx.x / CG;
  const int c = blockIdx.y * CG + cx, n = blockIdx.z, nch = gridDim.x;
  const bool cv = c < C;
  const int r0 = blockIdx.x * MMC_ROWS, r1 = min(HW, r0 + MMC_ROWS);
  const size_t base = (size_t)n * HW * C + c;
  const float mn = cv ? mm[((size_t)n * C + c) * 2] : 0.f, mx = cv ? mm[((size_t)n * C + c) * 2 + 1] : 1.f;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (cv)
    for (int p0 = r0 + py; p0 < r1; p0 += 4 * L) {   // four rows per step, their eight loads issued together
      float xr[4], gr[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int p = p0 + u * L;
        const size_t i = base + (size_t)(p < r1 ? p : r0) * C;
        xr[u] = x[i];
        gr[u] = dy[i];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (p0 + u * L < r1) {
          const float v = mm_pre(xr[u], lk), g = gr[u];
          s[0] += g;
          s[1] += g * (v - mn);
          s[2] += v == mn ? 1.f : 0.f;
          s[3] += v == mx ? 1.f : 0.f;
        }
    }
#pragma unroll
  for (int k = 0; k < 4; ++k) red[(k * L + py) * CG + cx] = s[k];
  __syncthreads();
  if (py == 0 && cv) {
    float* o = part + (((size_t)n * nch + blockIdx.x) * C + c) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float t = s[k];
      for (int q = 1; q < L; ++q) t += red[(k * L + q) * CG + cx];
      o[k] = t;
    }
  }
}
// coef[(n C + c) 3 ..] = d = mx - mn, dmn, dmx (the tie split of minmax_bwd_v)
__global__ void mmc_bwd_finalize_kernel(const float* __restrict__ part, const float* __restrict__ mm, int nch, int C,
                                        float* __restrict__ coef) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
  if (c >= C) return;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < nch; k0 += 4) {   // (four chunks' partials in flight, summed in chunk order)
    float4 pv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) pv[u] = ld4(part + (((size_t)n * nch + min(k0 + u, nch - 1)) * C + c) * 4);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (k0 + u < nch) {
        s[0] += pv[u].x; s[1] += pv[u].y; s[2] += pv[u].z; s[3] += pv[u].w;
      }
  }
  const float d = mm[((size_t)n * C + c) * 2 + 1] - mm[((size_t)n * C + c) * 2];
  float* o = coef + ((size_t)n * C + c) * 3;
  o[0] = d;
  o[1] = (-s[0] / d + s[1] / (d * d)) / s[2];
  o[2] = (-s[1] / (d * d)) / s[3];
}
__global__ __launch_bounds__(256) void mmc_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ mm,
                                                            const float* __restrict__ dy, const float* __restrict__ coef,
                                                            float* __restrict__ dx, int acc, unsigned short* __restrict__ dx_bf, int HW,
                                                            int C, float lk) {
  const int n = blockIdx.y, c4n = C >> 2;
  const size_t per = (size_t)HW * c4n, ib = (size_t)n * HW * C;
  for (size_t q = blockIdx.x * (size_t)256 + threadIdx.x; q < per; q += (size_t)gridDim.x * 256) {
    const int c = (int)(q % c4n) * 4;
    const size_t i = ib + q * 4;
    const float* mp = mm + ((size_t)n * C + c) * 2;
    const float* cp = coef + ((size_t)n * C + c) * 3;
    const float4 xv = ld4(x + i), g = ld4(dy + i), v = mm_pre4(xv, lk);
    const float xx[4] = {xv.x, xv.y, xv.z, xv.w}, vv[4] = {v.x, v.y, v.z, v.w}, gg[4] = {g.x, g.y, g.z, g.w};
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      r[j] = gg[j] / cp[3 * j];
      if (vv[j] == mp[2 * j]) r[j] += cp[3 * j + 1];
      if (vv[j] == mp[2 * j + 1]) r[j] += cp[3 * j + 2];
      r[j] *= mm_slope(xx[j], lk);
    }
    float4 o = make_float4(r[0], r[1], r[2], r[3]);
    if (acc & 1) { const float4 t = ld4(dx + i); o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w; }
    if (!(acc & 2)) st4(dx + i, o);
    if (dx_bf) st_bf4(dx_bf, i >> 2, o);
  }
}
// (more than 4,096 pixels per map.  Also taking the 64 x 64 maps when the per-map blocks do not fill the chip -- batch 16 -- measured
//  +-0 at 256 x 512: profiles/r06ad)
bool minmax_chunked_shape(int NB, int HW, int C) { return C % 8 == 0 && HW > 4096; }
size_t minmax_chunk_scratch_floats(int NB, int HW, int C) {
  const size_t nch = (size_t)(HW + MMC_ROWS - 1) / MMC_ROWS;
  return (size_t)NB * nch * C * 4 + (size_t)NB * C * 3;
}
static inline int mmc_ew_blocks(int HW, int C) { return (int)std::min<size_t>(2048, ((size_t)HW * (C >> 2) + 255) / 256); }

void launch_minmax_fwd(const float* x, float* y, float* mm, int NB, int HW, int C, hipStream_t s, float pre_leak,
                       const SlabSrc* x_slabs, float* xsum, float* chunk_scratch) {
  const SlabSrc xs = x_slabs ? *x_slabs : plain_src(x);
  EG_CHECK(xs.n <= 1 || (C % 8 == 0 && xsum), "minmax_fwd: a slab input needs C %% 8 == 0 and a destination for the sum");
  if (chunk_scratch && xs.n <= 1 && minmax_chunked_shape(NB, HW, C)) {   // large maps: chunked two-stage form
    const int nch = (HW + MMC_ROWS - 1) / MMC_ROWS;
    if (C % 64 == 0) hipLaunchKernelGGL(mmc_minmax_kernel<64>, dim3(nch, C / 64, NB), dim3(256), 0, s, x, chunk_scratch, HW, C, pre_leak);
    else hipLaunchKernelGGL(mmc_minmax_kernel<8>, dim3(nch, C / 8, NB), dim3(256), 0, s, x, chunk_scratch, HW, C, pre_leak);
    hipLaunchKernelGGL(mmc_minmax_finalize_kernel, dim3((C + 63) / 64, NB), dim3(64), 0, s, chunk_scratch, nch, C, mm);
    hipLaunchKernelGGL(mmc_fwd_apply_kernel, dim3(mmc_ew_blocks(HW, C), NB), dim3(256), 0, s, x, mm, y, HW, C, pre_leak);
    return;
  }
  if (C % 32 == 0) {
    hipLaunchKernelGGL((minmax_fwd_v<8, 4>), dim3(NB, C / 32), dim3(256), 0, s, xs, xsum, y, mm, HW, C, pre_leak);
    return;
  }
  if (C % 8 == 0) {
    hipLaunchKernelGGL((minmax_fwd_v<2, 16>), dim3(NB, C / 8), dim3(1024), 0, s, xs, xsum, y, mm, HW, C, pre_leak);
    return;
  }
  hipLaunchKernelGGL(minmax_fwd_kernel, dim3(NB, (C + 63) / 64), dim3(256), 0, s, x, y, mm, HW, C, pre_leak);
}
int minmax_bwd_colpart_rows(int NB, int HW, int C) { return (C % 8 == 0 && !minmax_chunked_shape(NB, HW, C)) ? NB : 0; }
void launch_minmax_bwd(const float* x, const float* mm, const float* dy, float* dx, int accumulate, int NB,
                       int HW, int C, hipStream_t s, float pre_leak, unsigned short* dx_bf, float* colpart, float* chunk_scratch) {
  if (chunk_scratch && !colpart && minmax_chunked_shape(NB, HW, C)) {   // large maps: chunked two-stage form
    EG_CHECK(!(accumulate & GRAD_PLANE_ONLY) || dx_bf, "minmax_bwd: GRAD_PLANE_ONLY without a plane");
    EG_CHECK(!dx_bf || C % 32 == 0, "minmax_bwd: a bf16 plane was requested with C %% 32 != 0");
    const int nch = (HW + MMC_ROWS - 1) / MMC_ROWS;
    float* coef = chunk_scratch + (size_t)NB * nch * C * 4;
    if (C % 64 == 0) hipLaunchKernelGGL(mmc_bwd_sums_kernel<64>, dim3(nch, C / 64, NB), dim3(256), 0, s, x, mm, dy, chunk_scratch, HW, C, pre_leak);
    else hipLaunchKernelGGL(mmc_bwd_sums_kernel<8>, dim3(nch, C / 8, NB), dim3(256), 0, s, x, mm, dy, chunk_scratch, HW, C, pre_leak);
    hipLaunchKernelGGL(mmc_bwd_finalize_kernel, dim3((C + 63) / 64, NB), dim3(64), 0, s, chunk_scratch, mm, nch, C, coef);
    hipLaunchKernelGGL(mmc_bwd_apply_kernel, dim3(mmc_ew_blocks(HW, C), NB), dim3(256), 0, s, x, mm, dy, coef, dx, accumulate, dx_bf, HW, C,
                       pre_leak);
    return;
  }
  EG_CHECK(!colpart || C % 8 == 0, "minmax_bwd: column partials need C %% 8 == 0");
  EG_CHECK(!(accumulate & GRAD_PLANE_ONLY) || dx_bf, "minmax_bwd: GRAD_PLANE_ONLY without a plane");
  if (C % 32 == 0) {
    hipLaunchKernelGGL((minmax_bwd_v<8, 4>), dim3(NB, C / 32), dim3(256), 0, s, x, mm, dy, dx, accumulate, HW, C, pre_leak, dx_bf, colpart);
    return;
  }
  EG_CHECK(!dx_bf, "minmax_bwd: a bf16 plane was requested with C %% 32 != 0");
  if (C % 8 == 0) {
    hipLaunchKernelGGL((minmax_bwd_v<2, 16>), dim3(NB, C / 8), dim3(1024), 0, s, x, mm, dy, dx, accumulate, HW, C, pre_leak, nullptr, colpart);
    return;
  }
  hipLaunchKernelGGL(minmax_bwd_kernel, dim3(NB, (C + 63) / 64), dim3(256), 0, s, x, mm, dy, dx, accumulate, HW, C, pre_leak);
}

__global__ void gate_fwd_kernel(const float* __restrict__ h, const float* __restrict__ rg,
                                const float* __restrict__ img, float* __restrict__ o, size_t n) {
  GRID_STRIDE(i, n) o[i] = h[i] + rg[i] * img[i];
}
__global__ void gate_bwd_kernel(const float* __restrict__ rg, const float* __restrict__ img,
                                const float* __restrict__ d, float* __restrict__ dh, int acc_h,
                                float* __restrict__ drg, float* __restrict__ dimg, size_t n) {
  GRID_STRIDE(i, n) {
    const float g = d[i];
    if (dh) dh[i] = acc_h ? dh[i] + g : g;
    drg[i] = g * img[i];
    dimg[i] = g * rg[i];
  }
}
void launch_gate_fwd(const float* h, const float* rg, const float* img, float* out, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(gate_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, h, rg, img, out, n);
}
void launch_gate_bwd(const float* rg, const float* img, const float* dout, float* dh, int acc_h, float* drg,
                     float* dimg, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(gate_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, rg, img, dout, dh, acc_h, drg, dimg, n);
}

// gate merge + PReLU in one pass (ops.cpp gate_prelu): same expressions, in the same order, as gate_* followed by prelu_*
__global__ void gate_prelu_fwd_kernel(const float* __restrict__ h, const float* __restrict__ rg, const float* __restrict__ img,
                                      const float* __restrict__ leak_p, float* __restrict__ o, size_t n) {
  const float lk = *leak_p;
  GRID_STRIDE(i, n) {
    const float v = h[i] + rg[i] * img[i];
    o[i] = fmaxf(lk * v, v);
  }
}
__global__ __launch_bounds__(256) void gate_prelu_bwd_kernel(const float* __restrict__ h, const float* __restrict__ rg,
                                                             const float* __restrict__ img, const float* __restrict__ leak_p,
                                                             const float* __restrict__ d, float* __restrict__ dh, int acc_h,
                                                             float* __restrict__ drg, float* __restrict__ dimg,
                                                             float* __restrict__ part, size_t n) {
  __shared__ float sm[4];
  const float lk = *leak_p;
  float s = 0.f;
  GRID_STRIDE(i, n) {
    const float r = rg[i], m = img[i];
    const float v = h[i] + r * m, gy = d[i];
    const bool first = (lk * v >= v);   // tf.maximum sends the gradient to the first argument on ties
    const float g = first ? lk * gy : gy;
    if (first) s += gy * v;
    if (dh) dh[i] = acc_h ? dh[i] + g : g;
    drg[i] = g * m;
    dimg[i] = g * r;
  }
  if (part) {
    const float tot = block_sum(s, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
  }
}
// float4 forms (n % 4 == 0, 16-byte aligned buffers)
__global__ void gate_prelu_fwd_v_kernel(const float* __restrict__ h, const float* __restrict__ rg, const float* __restrict__ img,
                                        const float* __restrict__ leak_p, float* __restrict__ o, size_t n4,
                                        unsigned short* __restrict__ o_bf) {
  const float lk = *leak_p;
  GRID_STRIDE(i, n4) {
    const float4 a = reinterpret_cast<const float4*>(h)[i], r = reinterpret_cast<const float4*>(rg)[i],
                 m = reinterpret_cast<const float4*>(img)[i];
    const float v0 = a.x + r.x * m.x, v1 = a.y + r.y * m.y, v2 = a.z + r.z * m.z, v3 = a.w + r.w * m.w;
    const float4 y = make_float4(fmaxf(lk * v0, v0), fmaxf(lk * v1, v1), fmaxf(lk * v2, v2), fmaxf(lk * v3, v3));
    if (o) reinterpret_cast<float4*>(o)[i] = y;       // (null: the output lives as its bf16 plane only, Tensor::d_plane_only)
    if (o_bf) st_bf4(o_bf, i, y);
  }
}
__global__ __launch_bounds__(256) void gate_prelu_bwd_v_kernel(const float* __restrict__ h, const float* __restrict__ rg,
                                                               const float* __restrict__ img, const float* __restrict__ leak_p,
                                                               const SlabSrc d, float* __restrict__ dh, int acc_h,
                                                               float* __restrict__ drg, float* __restrict__ dimg,
                                                               float* __restrict__ part, size_t n4, float* __restrict__ colpart,
                                                               int C4) {
  __shared__ float sm[4];
  __shared__ float4 redc[256];
  const float lk = *leak_p;
  float s = 0.f;
  float4 cs = f4_zero();   // column sums of dimg: the image convolution's bias gradient (kernels.h: colpart)
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 a = reinterpret_cast<const float4*>(h)[i], r = reinterpret_cast<const float4*>(rg)[i],
                 m = reinterpret_cast<const float4*>(img)[i], gy = slab_ld4_flat(d, i);
    const float v[4] = {a.x + r.x * m.x, a.y + r.y * m.y, a.z + r.z * m.z, a.w + r.w * m.w};
    const float gg[4] = {gy.x, gy.y, gy.z, gy.w};
    float g[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool first = (lk * v[j] >= v[j]);
      g[j] = first ? lk * gg[j] : gg[j];
      s += first ? gg[j] * v[j] : 0.f;
    }
    if (dh) {
      float4 o = make_float4(g[0], g[1], g[2], g[3]);
      if (acc_h) {
        const float4 q = reinterpret_cast<const float4*>(dh)[i];
        o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w;
      }
      reinterpret_cast<float4*>(dh)[i] = o;
    }
    reinterpret_cast<float4*>(drg)[i] = make_float4(g[0] * m.x, g[1] * m.y, g[2] * m.z, g[3] * m.w);
    const float4 di = make_float4(g[0] * r.x, g[1] * r.y, g[2] * r.z, g[3] * r.w);
    reinterpret_cast<float4*>(dimg)[i] = di;
    cs.x += di.x; cs.y += di.y; cs.z += di.z; cs.w += di.w;
  }
  if (part) {
    const float tot = block_sum(s, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
  }
  if (colpart) flat_colpart_store(cs, redc, colpart, C4);
}
static bool gp_vec_ok(size_t n, const void* a, const void* b, const void* c, const void* d, const void* e = nullptr,
                      const void* f = nullptr, const void* g = nullptr) {
  return (n & 3) == 0 && ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)d) | ((uintptr_t)e) | ((uintptr_t)f) |
                           ((uintptr_t)g)) & 15) == 0;
}
void launch_gate_prelu_fwd(const float* h, const float* rg, const float* img, const float* leak, float* out, size_t n,
                           hipStream_t s, unsigned short* out_bf) {
  if (gp_vec_ok(n, h, rg, img, out)) {
    hipLaunchKernelGGL(gate_prelu_fwd_v_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, s, h, rg, img, leak, out, n / 4, out_bf);
    return;
  }
  EG_CHECK(!out_bf && out, "gate_prelu_fwd: a bf16 plane / a plane-only output was requested from the scalar form");
  hipLaunchKernelGGL(gate_prelu_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, h, rg, img, leak, out, n);
}
void launch_gate_prelu_bwd(const float* h, const float* rg, const float* img, const float* leak, const float* dout, float* dh,
                           int acc_h, float* drg, float* dimg, float* dleak, int acc_dleak, size_t n, float* scratch,
                           hipStream_t s, const SlabSrc* dout_slabs, float* colpart_img, int C) {
  int nb;
  if (gp_vec_ok(n, h, rg, img, dout, dh, drg, dimg)) {
    nb = colpart_img ? flat_colpart_rows(n, C) : ew_blocks(n / 4, 2048);
    EG_CHECK(nb > 0, "gate_prelu_bwd: column partials were requested from a form that has none");
    hipLaunchKernelGGL(gate_prelu_bwd_v_kernel, dim3(nb), dim3(256), 0, s, h, rg, img, leak, dout_slabs ? *dout_slabs : plain_src(dout),
                       dh, acc_h, drg, dimg, dleak ? scratch : nullptr, n / 4, colpart_img, colpart_img ? C / 4 : 1);
  } else {
    EG_CHECK(!dout_slabs && !colpart_img, "gate_prelu_bwd: a slab input / column partials were requested from the scalar form");
    nb = ew_blocks(n, 2048);
    hipLaunchKernelGGL(gate_prelu_bwd_kernel, dim3(nb), dim3(256), 0, s, h, rg, img, leak, dout, dh, acc_h, drg, dimg,
                       dleak ? scratch : nullptr, n);
  }
  if (dleak) hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, scratch, nb, dleak, acc_dleak, 1.0f);
}

// ---- N == 1 linear head (d_linear_5, discriminator.py:74) --------------------------------------
// one 1024-thread block per row: with only B (64..128) rows there are fewer blocks than CUs, so each block must
// keep many loads in flight itself (8 independent 16-byte loads per thread at K = 32768)
__global__ __launch_bounds__(1024) void gemv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, float* __restrict__ y, int K) {
  __shared__ float sm[16];
  const float* xr = x + (size_t)blockIdx.x * K;
  float s = 0.f;
  if ((K & 3) == 0 && (((uintptr_t)xr | (uintptr_t)w) & 15) == 0) {
    const float4* x4 = reinterpret_cast<const float4*>(xr);
    const float4* w4 = reinterpret_cast<const float4*>(w);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int k = threadIdx.x; k < K / 4; k += 1024) {
      const float4 a = x4[k], c = w4[k];
      s0 += a.x * c.x; s1 += a.y * c.y; s2 += a.z * c.z; s3 += a.w * c.w;
    }
    s = (s0 + s1) + (s2 + s3);
  } else {
    for (int k = threadIdx.x; k < K; k += 1024) s += xr[k] * w[k];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) tot += sm[i];
    y[blockIdx.x] = tot + (b ? b[0] : 0.f);
  }
}
__global__ void gemv_bwd_x_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                  int acc, int K, size_t n) {
  GRID_STRIDE(i, n) {
    const float g = dy[i / K] * w[i % K];
    dx[i] = acc ? dx[i] + g : g;
  }
}
// dw[k] (+)= sum_r x[r][k] dy[r]: block = 16 column quads x 16 row lanes (64 columns), lane l sums rows l, l+16, ...;
// the 16 lane totals are added in lane order (deterministic).  db (+)= sum_r dy[r] by block 0.
__global__ __launch_bounds__(256) void gemv_bwd_w_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         float* __restrict__ dw, int acc_w, float* __restrict__ db,
                                                         int acc_b, int R, int K) {
  __shared__ float4 red[16][17];
  const int q = threadIdx.x & 15, l = threadIdx.x >> 4;
  const int k = (blockIdx.x * 16 + q) * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (k < K)
    for (int r = l; r < R; r += 16) {
      const float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * K + k);
      const float g = dy[r];
      s.x += v.x * g; s.y += v.y * g; s.z += v.z * g; s.w += v.w * g;
    }
  red[l][q] = s;
  __syncthreads();
  if (l == 0 && k < K) {
    float4 t = red[0][q];
#pragma unroll
    for (int i = 1; i < 16; ++i) {
      const float4 v = red[i][q];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    float4* o = reinterpret_cast<float4*>(dw + k);
    if (acc_w) {
      const float4 v = *o;
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    *o = t;
  }
  if (db && blockIdx.x == 0 && threadIdx.x == 0) {
    float sb = 0.f;
    for (int r = 0; r < R; ++r) sb += dy[r];
    db[0] = acc_b ? db[0] + sb : sb;
  }
}
// scalar form for K % 4 != 0 or unaligned buffers
__global__ void gemv_bwd_w_scalar_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw,
                                         int acc_w, float* __restrict__ db, int acc_b, int R, int K) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < K) {
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += x[(size_t)r * K + k] * dy[r];
    dw[k] = acc_w ? dw[k] + s : s;
  }
  if (db && k == 0) {
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += dy[r];
    db[0] = acc_b ? db[0] + s : s;
  }
}
void launch_gemv_fwd(const float* x, const float* w, const float* b, float* y, int R, int K, hipStream_t s) {
  hipLaunchKernelGGL(gemv_fwd_kernel, dim3(R), dim3(1024), 0, s, x, w, b, y, K);
}
void launch_gemv_bwd_x(const float* dy, const float* w, float* dx, int accumulate, int R, int K, hipStream_t s) {
  const size_t n = (size_t)R * K;
  hipLaunchKernelGGL(gemv_bwd_x_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, dy, w, dx, accumulate, K, n);
}
void launch_gemv_bwd_w(const float* x, const float* dy, float* dw, int acc_w, float* db, int acc_b, int R, int K,
                       hipStream_t s) {
  if ((K & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)dw)) & 15) == 0)
    hipLaunchKernelGGL(gemv_bwd_w_kernel, dim3((K / 4 + 15) / 16), dim3(256), 0, s, x, dy, dw, acc_w, db, acc_b, R, K);
  else
    hipLaunchKernelGGL(gemv_bwd_w_scalar_kernel, dim3((K + 255) / 256), dim3(256), 0, s, x, dy, dw, acc_w, db, acc_b, R, K);
}

// ---- spectral norm, batched over a device table of weights (normalization.py:38-76) -----------
#define SN_EPS 1e-12f
// t[k] = sum_c W[k][c]*vec[c]*coef ; one wave per row
__global__ __launch_bounds__(256) void sn_rowdot_kernel(const SNDesc* __restrict__ tab, int bwd) {
  const SNDesc d = tab[blockIdx.y];
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (k >= d.K) return;
  const float* vec = bwd ? d.sv : d.u;
  const float coef = bwd ? d.scal[5] : 1.f;
  const float* wr = d.W + (size_t)k * d.Co;
  float s = 0.f;
  for (int c = lane; c < d.Co; c += 64) s += wr[c] * vec[c];
  s = wave_sum(s);
  if (lane == 0) (bwd ? d.dv : d.t)[k] = s * coef;
}
// nt = |t| ; t <- t/(nt+eps)   (single block per weight)
__global__ __launch_bounds__(256) void sn_normalize_t_kernel(const SNDesc* __restrict__ tab) {
  __shared__ float sm[4];
  const SNDesc d = tab[blockIdx.x];
  float s = 0.f;
  for (int k = threadIdx.x; k < d.K; k += 256) s += d.t[k] * d.t[k];
  const float nt = sqrtf(block_sum(s, sm));
  const float inv = 1.f / (nt + SN_EPS);
  for (int k = threadIdx.x; k < d.K; k += 256) d.t[k] *= inv;
  if (threadIdx.x == 0) d.scal[0] = nt;
}
// sv[c] = sum_k v[k]*W[k][c]: the K rows are split over SN_KS blocks (the 23 weights x 12 column blocks alone
// leave the chip idle on a 6912-row serial loop); partials go to sv[(1+z)*Co + c], summed in fixed order by sn_sigma
#define SN_KS 16
__global__ __launch_bounds__(256) void sn_coldot_kernel(const SNDesc* __restrict__ tab) {
  __shared__ float red[4 * 64];
  const SNDesc d = tab[blockIdx.y];
  const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  if (blockIdx.x * 64 >= d.Co) return;
  const int per = (d.K + SN_KS - 1) / SN_KS;
  const int k0 = blockIdx.z * per, k1 = min(d.K, k0 + per);
  float s = 0.f;
  if (c < d.Co)
    for (int k = k0 + py; k < k1; k += 4) s += d.t[k] * d.W[(size_t)k * d.Co + c];
  red[py * 64 + cx] = s;
  __syncthreads();
  if (py == 0 && c < d.Co) d.sv[(size_t)(1 + blockIdx.z) * d.Co + c] = red[cx] + red[64 + cx] + red[128 + cx] + red[192 + cx];
}
// sv = sum of the K-split partials; ns = |sv| ; sigma = ns^2/(ns+eps)  (= v W u'^T with u' = sv/(ns+eps))
__global__ __launch_bounds__(256) void sn_sigma_kernel(const SNDesc* __restrict__ tab) {
  __shared__ float sm[4];
  const SNDesc d = tab[blockIdx.x];
  float s = 0.f;
  for (int c = threadIdx.x; c < d.Co; c += 256) {
    float v = 0.f;
#pragma unroll
    for (int z = 0; z < SN_KS; ++z) v += d.sv[(size_t)(1 + z) * d.Co + c];
    d.sv[c] = v;
    s += v * v;
  }
  const float ns = sqrtf(block_sum(s, sm));
  if (threadIdx.x == 0) {
    d.scal[1] = ns;
    d.scal[2] = ns * ns / (ns + SN_EPS);
  }
}
__global__ void sn_scale_kernel(const SNDesc* __restrict__ tab) {
  const SNDesc d = tab[blockIdx.y];
  const float inv = 1.f / d.scal[2];
  const size_t n = (size_t)d.K * d.Co;
  GRID_STRIDE(i, n) d.Wbar[i] = d.W[i] * inv;
}
int sn_sv_floats(int Co) { return (1 + SN_KS) * Co; }
void launch_sn_fwd(const SNDesc* tab, int count, int maxK, int maxCo, hipStream_t s) {
  hipLaunchKernelGGL(sn_rowdot_kernel, dim3((maxK + 3) / 4, count), dim3(256), 0, s, tab, 0);
  hipLaunchKernelGGL(sn_normalize_t_kernel, dim3(count), dim3(256), 0, s, tab);
  hipLaunchKernelGGL(sn_coldot_kernel, dim3((maxCo + 63) / 64, count, SN_KS), dim3(256), 0, s, tab);
  hipLaunchKernelGGL(sn_sigma_kernel, dim3(count), dim3(256), 0, s, tab);
  hipLaunchKernelGGL(sn_scale_kernel, dim3(256, count), dim3(256), 0, s, tab);
}
// backward ------------------------------------------------------------------------------------
// stage 1: partial dots of dWbar.W into scal[8 .. 8 + SN_DOT_PARTS) -- one partial per block, summed in block order by stage 2.
// Round 6: 16-byte loads, four quads of each operand in flight per thread, 256 parts per weight (the 64-part scalar form
// walked the 5.3 M elements of the classifier's last unit one dependent load at a time: 145 us for 137 MB)
#define SN_DOT_PARTS (SN_SCAL_FLOATS - 8)
__global__ __launch_bounds__(256) void sn_bwd_dot_kernel(const SNDesc* __restrict__ tab) {
  __shared__ float sm[4];
  const SNDesc d = tab[blockIdx.y];
  const size_t n = (size_t)d.K * d.Co;
  float s = 0.f;
  if ((n & 3) == 0 && ((((uintptr_t)d.dWbar) | ((uintptr_t)d.W)) & 15) == 0) {
    const float4* __restrict__ a4 = reinterpret_cast<const float4*>(d.dWbar);
    const float4* __restrict__ w4 = reinterpret_cast<const float4*>(d.W);
    const size_t n4 = n >> 2, stride = (size_t)gridDim.x * 256;
    size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (; i + 3 * stride < n4; i += 4 * stride) {
      const float4 a0 = a4[i], a1 = a4[i + stride], a2 = a4[i + 2 * stride], a3 = a4[i + 3 * stride];
      const float4 b0 = w4[i], b1 = w4[i + stride], b2 = w4[i + 2 * stride], b3 = w4[i + 3 * stride];
      s0 += a0.x * b0.x + a0.y * b0.y + a0.z * b0.z + a0.w * b0.w;
      s1 += a1.x * b1.x + a1.y * b1.y + a1.z * b1.z + a1.w * b1.w;
      s2 += a2.x * b2.x + a2.y * b2.y + a2.z * b2.z + a2.w * b2.w;
      s3 += a3.x * b3.x + a3.y * b3.y + a3.z * b3.z + a3.w * b3.w;
    }
    for (; i < n4; i += stride) {
      const float4 a0 = a4[i], b0 = w4[i];
      s0 += a0.x * b0.x + a0.y * b0.y + a0.z * b0.z + a0.w * b0.w;
    }
    s = (s0 + s1) + (s2 + s3);
  } else {
    GRID_STRIDE(i, n) s += d.dWbar[i] * d.W[i];
  }
  const float tot = block_sum(s, sm);
  if (threadIdx.x == 0) d.scal[8 + blockIdx.x] = tot;
}
// stage 2: dot, dsigma, coefficient for ds = coef * sv
__global__ void sn_bwd_coef_kernel(const SNDesc* __restrict__ tab, int nparts) {
  const SNDesc d = tab[blockIdx.x];
  float dot = 0.f;   // one wave: lane l sums parts l, l + 64, ...; then the fixed-order wave sum
  for (int i = threadIdx.x; i < nparts; i += 64) dot += d.scal[8 + i];
  dot = wave_sum(dot);
  if (threadIdx.x != 0) return;
  const float sigma = d.scal[2], ns = d.scal[1];
  const float dsigma = -dot / (sigma * sigma);
  const float q = ns + SN_EPS;
  const float dns = dsigma * (ns * ns + 2.f * ns * SN_EPS) / (q * q);
  d.scal[3] = dot;
  d.scal[5] = dns / ns;  // ds[c] = scal[5] * sv[c]
}
// P = sum_k dv[k]*v[k]
__global__ __launch_bounds__(256) void sn_bwd_p_kernel(const SNDesc* __restrict__ tab) {
  __shared__ float sm[4];
  const SNDesc d = tab[blockIdx.x];
  float s = 0.f;
  for (int k = threadIdx.x; k < d.K; k += 256) s += d.dv[k] * d.t[k];
  const float P = block_sum(s, sm);
  if (threadIdx.x == 0) d.scal[4] = P;
}
// dW = dWbar/sigma + v (x) ds + dt (x) u ,  dt[k] = dv[k]/(nt+eps) - P v[k]/nt
__global__ void sn_bwd_final_kernel(const SNDesc* __restrict__ tab) {
  const SNDesc d = tab[blockIdx.y];
  const float inv_sigma = 1.f / d.scal[2], nt = d.scal[0], P = d.scal[4], coef = d.scal[5];
  const float invq = 1.f / (nt + SN_EPS);
  const size_t n = (size_t)d.K * d.Co;
  if ((d.Co & 3) == 0 && ((((uintptr_t)d.dWbar) | ((uintptr_t)d.dW) | ((uintptr_t)d.sv) | ((uintptr_t)d.u)) & 15) == 0) {
    // a quad = four consecutive columns of one row k (the same expressions per element as the scalar form below)
    const unsigned co4 = (unsigned)d.Co >> 2;
    const size_t n4 = n >> 2;
    GRID_STRIDE(q, n4) {
      const unsigned k = (unsigned)(q / co4), c4 = (unsigned)(q - (size_t)k * co4);
      const float v = d.t[k];
      const float dt = d.dv[k] * invq - P * v / nt;
      const float4 a = reinterpret_cast<const float4*>(d.dWbar)[q];
      const float4 sv = reinterpret_cast<const float4*>(d.sv)[c4], u = reinterpret_cast<const float4*>(d.u)[c4];
      float4 g;
      g.x = a.x * inv_sigma + v * coef * sv.x + dt * u.x;
      g.y = a.y * inv_sigma + v * coef * sv.y + dt * u.y;
      g.z = a.z * inv_sigma + v * coef * sv.z + dt * u.z;
      g.w = a.w * inv_sigma + v * coef * sv.w + dt * u.w;
      float4* o = reinterpret_cast<float4*>(d.dW) + q;
      if (d.acc_dW) {
        const float4 p = *o;
        g.x += p.x; g.y += p.y; g.z += p.z; g.w += p.w;
      }
      *o = g;
    }
    return;
  }
  GRID_STRIDE(i, n) {
    const int k = (int)(i / d.Co), c = (int)(i % d.Co);
    const float v = d.t[k];
    const float dt = d.dv[k] * invq - P * v / nt;
    const float g = d.dWbar[i] * inv_sigma + v * coef * d.sv[c] + dt * d.u[c];
    d.dW[i] = d.acc_dW ? d.dW[i] + g : g;
  }
}
void launch_sn_bwd(const SNDesc* tab, int count, int maxK, int maxCo, hipStream_t s) {
  const int nparts = SN_DOT_PARTS;
  hipLaunchKernelGGL(sn_bwd_dot_kernel, dim3(nparts, count), dim3(256), 0, s, tab);
  hipLaunchKernelGGL(sn_bwd_coef_kernel, dim3(count), dim3(64), 0, s, tab, nparts);
  hipLaunchKernelGGL(sn_rowdot_kernel, dim3((maxK + 3) / 4, count), dim3(256), 0, s, tab, 1);
  hipLaunchKernelGGL(sn_bwd_p_kernel, dim3(count), dim3(256), 0, s, tab);
  hipLaunchKernelGGL(sn_bwd_final_kernel, dim3(256, count), dim3(256), 0, s, tab);
}

// ---- losses (edgegan/nn/functional.py) ----------------------------------------------------------
__global__ void mean_diff_kernel(const float* fake, const float* real, float* out, int B) {
  __shared__ float sm[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) s += fake[i] - real[i];
  const float t = block_sum(s, sm);
  if (threadIdx.x == 0) out[0] = t / B;
}
__global__ void mean_neg_kernel(const float* x, float* out, int B) {
  __shared__ float sm[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) s -= x[i];
  const float t = block_sum(s, sm);
  if (threadIdx.x == 0) out[0] = t / B;
}
void launch_mean_diff(const float* fake, const float* real, float* out, int B, hipStream_t s) {
  hipLaunchKernelGGL(mean_diff_kernel, dim3(1), dim3(256), 0, s, fake, real, out, B);
}
void launch_mean_neg(const float* x, float* out, int B, hipStream_t s) {
  hipLaunchKernelGGL(mean_neg_kernel, dim3(1), dim3(256), 0, s, x, out, B);
}

// gradient penalty (functional.py:26-29, lambda from train.py:52)
// one 1024-thread block per sample: 16-byte loads, three quads in flight per thread, fixed-order sum (round 6; the
// 256-thread scalar form took 39 us for the 12.6 MB of a batch of 64)
__global__ __launch_bounds__(1024) void gp_norm_kernel(const float* __restrict__ g, float* __restrict__ norms,
                                                       size_t per) {
  __shared__ float sm[16];
  const float* p = g + (size_t)blockIdx.x * per;
  float s = 0.f;
  if ((per & 3) == 0 && (((uintptr_t)g) & 15) == 0) {
    const float4* __restrict__ p4 = reinterpret_cast<const float4*>(p);
    const size_t n4 = per >> 2;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    size_t i = threadIdx.x;
    for (; i + 2048 < n4; i += 3072) {
      const float4 a = p4[i], b = p4[i + 1024], c = p4[i + 2048];
      s0 += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
      s1 += b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
      s2 += c.x * c.x + c.y * c.y + c.z * c.z + c.w * c.w;
    }
    for (; i < n4; i += 1024) {
      const float4 a = p4[i];
      s0 += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
    }
    s = s0 + s1 + s2;
  } else {
    for (size_t i = threadIdx.x; i < per; i += 1024) s += p[i] * p[i];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += sm[w];
    norms[blockIdx.x] = sqrtf(t);
  }
}
__global__ void gp_v_kernel(const float* __restrict__ g, const float* __restrict__ norms, float* __restrict__ v,
                            size_t per, int B, float lambda, size_t n) {
  GRID_STRIDE(i, n) {
    const float nr = norms[i / per];
    v[i] = lambda * (2.f / B) * (nr - 1.f) / nr * g[i];
  }
}
__global__ void gp_loss_kernel(const float* norms, float* gp, int B, float lambda) {
  __shared__ float sm[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) s += (norms[i] - 1.f) * (norms[i] - 1.f);
  const float t = block_sum(s, sm);
  if (threadIdx.x == 0) gp[0] = lambda * t / B;
}
void launch_gp_v(const float* g, float* v, float* norms, float* gp, int B, size_t per, float lambda,
                 hipStream_t s) {
  hipLaunchKernelGGL(gp_norm_kernel, dim3(B), dim3(1024), 0, s, g, norms, per);
  const size_t n = (size_t)B * per;
  if (v) hipLaunchKernelGGL(gp_v_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, g, norms, v, per, B, lambda, n);
  hipLaunchKernelGGL(gp_loss_kernel, dim3(1), dim3(256), 0, s, norms, gp, B, lambda);
}
// quirk N10: the penalised function is sum_b sigmoid(d_b) + d_b
__global__ void gp_head_kernel(const float* d, const float* td, float* o1, float* o2, int B, int mode) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float sg = 1.f / (1.f + expf(-d[b]));
  const float sp = sg * (1.f - sg);
  if (mode == 0) {
    o1[b] = 1.f + sp;
  } else {
    o1[b] = sp * (1.f - 2.f * sg) * td[b];
    o2[b] = 1.f + sp;
  }
}
void launch_gp_head(const float* d, const float* td, float* o1, float* o2, int B, int mode, hipStream_t s) {
  hipLaunchKernelGGL(gp_head_kernel, dim3((B + 255) / 256), dim3(256), 0, s, d, td, o1, o2, B, mode);
}

// focal AC-GAN loss (functional.py:5-16); one thread per sample, NC <= 64
__global__ void acgan_kernel(const float* __restrict__ logits, const float* __restrict__ zlast, int z_stride,
                             float* __restrict__ per_loss, float* __restrict__ dlogits, int B, int NC, int mode) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* l = logits + (size_t)b * NC;
  const int y = (int)zlast[(size_t)b * z_stride];
  float mx = l[0];
  for (int j = 1; j < NC; ++j) mx = fmaxf(mx, l[j]);
  float se = 0.f;
  for (int j = 0; j < NC; ++j) se += expf(l[j] - mx);
  const float lse = mx + logf(se);
  const float ce = lse - l[y];
  const float py = expf(l[y] - lse);
  float coef_pj, coef_y;  // dl_j = coef_pj * p_j + (j==y) * coef_y
  if (mode == 0) {
    per_loss[b] = (1.f - py) * (1.f - py) * ce;
    // f = (1-py)^2 * (-log py) ; df/dpy = 2(1-py) log py - (1-py)^2/py ; dpy/dl_j = py (delta - p_j).
    // k = df/dpy * py / B with py cancelled algebraically: py = exp(l_y - lse) underflows to 0 for a logit gap > ~88
    // and (1-py)^2/py * py would be inf * 0 = NaN
    const float k = (2.f * (1.f - py) * (-ce) * py - (1.f - py) * (1.f - py)) / B;
    coef_pj = -k;
    coef_y = k;
  } else {
    per_loss[b] = 0.5f * ce;
    coef_pj = 0.5f / B;
    coef_y = -0.5f / B;
  }
  if (dlogits)
    for (int j = 0; j < NC; ++j)
      dlogits[(size_t)b * NC + j] = coef_pj * expf(l[j] - lse) + (j == y ? coef_y : 0.f);
}
void launch_acgan_loss(const float* logits, const float* zlast, int z_stride, float* loss, float* dlogits, int B,
                       int NC, int mode, float* scratch, hipStream_t s) {
  hipLaunchKernelGGL(acgan_kernel, dim3((B + 255) / 256), dim3(256), 0, s, logits, zlast, z_stride, scratch, dlogits,
                     B, NC, mode);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, scratch, B, loss, 0, 1.0f / B);
}

// weighted L1 (functional.py:40-41)
__global__ void l1_kernel(const float* __restrict__ target, int t_stride, const float* __restrict__ recon,
                          float* __restrict__ loss, float* __restrict__ drecon, int B, int Z, float weight) {
  __shared__ float sm[4];
  const int n = B * Z;
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int b = i / Z, j = i % Z;
    const float d = target[(size_t)b * t_stride + j] - recon[i];
    s += fabsf(d);
    if (drecon) drecon[i] = -(weight / n) * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
  }
  const float t = block_sum(s, sm);
  if (threadIdx.x == 0) loss[0] = weight * t / n;
}
void launch_l1(const float* target, int t_stride, const float* recon, float* loss, float* drecon, int B, int Z,
               float weight, hipStream_t s) {
  hipLaunchKernelGGL(l1_kernel, dim3(1), dim3(256), 0, s, target, t_stride, recon, loss, drecon, B, Z, weight);
}

// The encoder's head in a forward-only graph, ONE launch per batch instead of eight (encoder.py:69-79, linear.py:79-92):
//   feat = mean over the map of relu(x) (gmean_fwd_kernel's order);  mu = feat Wmu + bmu,  log sigma = feat Wls + bls;
//   z_r = mu + eps_r exp(log sigma) for the call's one or two scalar draws (reparam_fwd_kernel's expression).
// One block per image.  round_bf16 (compute_dtype bf16: a linear with 512 contracted and 100 produced channels qualifies for
// rounded operands) rounds feat and the weights to bf16 before the fp32 multiply-add; products of bf16 numbers are exact in fp32,
// so what differs from the MFMA path is the order of the 512-term sum only.
template <bool ROUND>
__global__ __launch_bounds__(256) void enc_head_kernel(const float* __restrict__ x, const float* __restrict__ wmu, const float* __restrict__ bmu,
                                                       const float* __restrict__ wls, const float* __restrict__ bls,
                                                       const float* __restrict__ eps, float* __restrict__ mu, float* __restrict__ ls,
                                                       float* __restrict__ z0, float* __restrict__ z1, int HW, int C, int Z) {
  extern __shared__ float sm[];   // feat[C] | part[KS][2 Z]
  constexpr int KS = 4;           // K slices: 4 x (2 Z / 4) column quads = 200 of the 256 threads busy at Z = 100, C / 4 dependent FMAs each
  constexpr int U = 16;           // filter rows in flight per thread (the loop is latency-bound: one block per image, 64 blocks)
  float* feat = sm;
  float* part = sm + C;
  const int n = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f;
#pragma unroll 4
    for (int p = 0; p < HW; ++p) {
      float v = x[((size_t)n * HW + p) * C + c];
      if (v < 0.f) v = 0.f;
      s += v;
    }
    float f = s / HW;
    if (ROUND) f = (float)(__bf16)f;
    feat[c] = f;
  }
  __syncthreads();
  // thread = (K slice, column quad of [mu | log sigma]); each slice's partial sum in ascending k, the slices summed in order below
  const int Zq = Z / 4, nq = 2 * Zq;
  for (int t = threadIdx.x; t < KS * nq; t += 256) {
    const int ks = t / nq, q = t - ks * nq;
    const bool second = q >= Zq;
    const int col = (second ? q - Zq : q) * 4;
    const float* w = (second ? wls : wmu) + col;
    const int per = (C + KS - 1) / KS, k0 = ks * per, k1 = min(C, k0 + per);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int kb = k0; kb < k1; kb += U) {
      float4 wv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) wv[u] = *reinterpret_cast<const float4*>(w + (size_t)min(kb + u, k1 - 1) * Z);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (kb + u < k1) {
          float4 v = wv[u];
          if (ROUND) v = make_float4((float)(__bf16)v.x, (float)(__bf16)v.y, (float)(__bf16)v.z, (float)(__bf16)v.w);
          const float f = feat[kb + u];
          acc.x = fmaf(f, v.x, acc.x); acc.y = fmaf(f, v.y, acc.y); acc.z = fmaf(f, v.z, acc.z); acc.w = fmaf(f, v.w, acc.w);
        }
      }
    }
    *reinterpret_cast<float4*>(part + (size_t)ks * 2 * Z + (second ? Z : 0) + col) = acc;
  }
  __syncthreads();
  const float e0 = eps[0], e1 = z1 ? eps[1] : 0.f;
  for (int j = threadIdx.x; j < Z; j += 256) {
    float m = 0.f, l = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      m += part[ks * 2 * Z + j];
      l += part[ks * 2 * Z + Z + j];
    }
    m += bmu[j];
    l += bls[j];
    const size_t i = (size_t)n * Z + j;
    mu[i] = m;
    ls[i] = l;
    const float ex = expf(l);
    z0[i] = m + e0 * ex;
    if (z1) z1[i] = m + e1 * ex;
  }
}
void launch_enc_head(const float* x, const float* wmu, const float* bmu, const float* wls, const float* bls, const float* eps, float* mu,
                     float* ls, float* z0, float* z1, int NB, int HW, int C, int Z, int round_bf16, hipStream_t s) {
  const size_t lds = (size_t)(C + 4 * 2 * Z) * sizeof(float);
  if (round_bf16)
    hipLaunchKernelGGL(enc_head_kernel<true>, dim3(NB), dim3(256), lds, s, x, wmu, bmu, wls, bls, eps, mu, ls, z0, z1, HW, C, Z);
  else
    hipLaunchKernelGGL(enc_head_kernel<false>, dim3(NB), dim3(256), lds, s, x, wmu, bmu, wls, bls, eps, mu, ls, z0, z1, HW, C, Z);
}

// z = mu + eps*exp(log_sigma) with ONE scalar eps (encoder.py:78-79, quirk N3)
__global__ void reparam_fwd_kernel(const float* mu, const float* ls, const float* eps_p, float* z, size_t n) {
  const float e = *eps_p;
  GRID_STRIDE(i, n) z[i] = mu[i] + e * expf(ls[i]);
}
__global__ void reparam_bwd_kernel(const float* ls, const float* eps_p, const float* dz, float* dmu, float* dls,
                                   size_t n) {
  const float e = *eps_p;
  GRID_STRIDE(i, n) {
    dmu[i] = dz[i];
    dls[i] = dz[i] * e * expf(ls[i]);
  }
}
void launch_reparam_fwd(const float* mu, const float* ls, const float* eps_p, float* z, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(reparam_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, mu, ls, eps_p, z, n);
}
void launch_reparam_bwd(const float* ls, const float* eps_p, const float* dz, float* dmu, float* dls, size_t n,
                        hipStream_t s) {
  hipLaunchKernelGGL(reparam_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, ls, eps_p, dz, dmu, dls, n);
}

// z_onehot (edgegan.py:188-197) / inference concat (edgegan.py:507-514)
__global__ void z_onehot_kernel(const float* z, const int* classes, int z_stride, float* out, int B, int zd, int NC) {
  const int W = zd + NC;
  const size_t n = (size_t)B * W;
  GRID_STRIDE(i, n) {
    const int b = (int)(i / W), j = (int)(i % W);
    if (j < zd) {
      out[i] = z[(size_t)b * z_stride + j];
    } else {
      const int y = classes ? classes[b] : (int)z[(size_t)b * z_stride + zd];
      out[i] = (j - zd == y) ? 1.f : 0.f;
    }
  }
}
void launch_z_onehot(const float* z, float* out, int B, int zd, int NC, hipStream_t s) {
  hipLaunchKernelGGL(z_onehot_kernel, dim3(ew_blocks((size_t)B * (zd + NC))), dim3(256), 0, s, z, (const int*)nullptr,
                     zd + 1, out, B, zd, NC);
}
void launch_class_onehot(const float* zrec, const int* classes, float* out, int B, int zd, int NC, hipStream_t s) {
  hipLaunchKernelGGL(z_onehot_kernel, dim3(ew_blocks((size_t)B * (zd + NC))), dim3(256), 0, s, zrec, classes, zd, out,
                     B, zd, NC);
}

// tf.train.RMSPropOptimizer(lr): decay .9, momentum 0, eps 1e-10
__global__ void rmsprop_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ ms,
                               float* __restrict__ mom, size_t n, float lr, float gscale) {
  GRID_STRIDE(i, n) {
    const float gi = g[i] * gscale;
    const float m = ms[i] + (gi * gi - ms[i]) * (1.0f - 0.9f);
    const float mo = 0.0f * mom[i] + lr * gi / sqrtf(m + 1e-10f);
    ms[i] = m;
    mom[i] = mo;
    w[i] -= mo;
  }
}
void launch_rmsprop(float* w, const float* g, float* ms, float* mom, size_t n, float lr, float gscale,
                    hipStream_t s) {
  hipLaunchKernelGGL(rmsprop_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, w, g, ms, mom, n, lr, gscale);
}


// =============================================================================================
// What the matrix pipe sustains on v_mfma_f32_32x32x16_bf16 with nothing else in the way (bench.py's roofline context):
// one block of 8 waves per CU, each wave `iters` x 16 MFMAs on 4 independent accumulators, no memory traffic.  With
// constant operands the pipe runs at the nominal clock; with random mantissas / signs the power management lowers it
// (MI355X here: 2,482 vs 1,985 TFLOP/s, profiles/r03x_mfma_peak.txt) -- the second figure is what a kernel on real data
// can approach, the guide's 2.5 PFLOP/s what `roofline.peak` is priced against.
// =============================================================================================
typedef __bf16 probe_bf16x8 __attribute__((ext_vector_type(8)));
typedef float probe_f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512, 1) void mfma_probe_kernel(float* out, int iters, int random_operands) {
  probe_f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  probe_bf16x8 a, b;
  unsigned h = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
  for (int i = 0; i < 8; ++i) {
    h = h * 1664525u + 1013904223u;
    const unsigned short ra = (unsigned short)(0x3f00u | ((h >> 9) & 0x80ffu)), rb = (unsigned short)(0x3f00u | ((h >> 17) & 0x80ffu));
    a[i] = random_operands ? __builtin_bit_cast(__bf16, ra) : (__bf16)(float)(threadIdx.x & 3);
    b[i] = random_operands ? __builtin_bit_cast(__bf16, rb) : (__bf16)1.0f;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][threadIdx.x & 15];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
void launch_mfma_probe(float* scratch, int blocks, int iters, int random_operands, hipStream_t s) {
  hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(512), 0, s, scratch, iters, random_operands);
}

}  // namespace eg
