// Normalisation + activation kernels (HBM-bound; NHWC fp32).
//
// InstanceNorm of the reference (edgegan/nn/modules/normalization.py:13-18):
//     y = (x - mean) / (sqrt(var) + 1e-5)     mean/var over H*W per (n, c), biased var, no affine
// fused with the activation that follows it in conv_block / deconv_block / residual
// (conv.py:61-67,70-85,124-130).  Its Jacobian J = r(I - 11^T/HW) - r^2 xc xc^T/(HW sigma),
// r = 1/(sigma+eps), is symmetric, so the same "apply J" code serves the backward pass and the
// forward-mode tangent used for the WGAN-GP term (functional.py:26-29 differentiated once more);
// in_tangent_bwd is the reverse of that tangent (gradient w.r.t. both the tangent and the primal x).
//
// Block = 256 threads = 64 channels x 4 pixel lanes, one block per (sample, 64-channel group):
// a wave reads 256-byte channel rows (coalesced), reductions over the 4 pixel lanes go through LDS.
#include <algorithm>

#include "common.h"
#include "kernels.h"
#include "cq_reduce.h"

namespace eg {

#define IN_EPS 1e-5f

__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case ACT_RELU: return v > 0.f ? v : 0.f;
    case ACT_LRELU: return fmaxf(v, 0.2f * v);
    case ACT_TANH: return tanhf(v);
    default: return v;
  }
}
// derivative expressed through the OUTPUT value (tf tie conventions: SURVEY Appendix B.8)
__device__ __forceinline__ float act_grad_from_out(float o, int act) {
  switch (act) {
    case ACT_RELU: return o > 0.f ? 1.f : 0.f;
    case ACT_LRELU: return o >= 0.f ? 1.f : 0.2f;
    case ACT_TANH: return 1.f - o * o;
    default: return 1.f;
  }
}

// reduce NV per-thread partials over the 4 pixel lanes; result broadcast to all lanes of a channel
template <int NV>
__device__ __forceinline__ void lane4_reduce(float (&v)[NV], float* red /*[NV][4][64]*/, int cx, int py) {
#pragma unroll
  for (int k = 0; k < NV; ++k) red[(k * 4 + py) * 64 + cx] = v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k)
    v[k] = red[(k * 4 + 0) * 64 + cx] + red[(k * 4 + 1) * 64 + cx] + red[(k * 4 + 2) * 64 + cx] +
           red[(k * 4 + 3) * 64 + cx];
  __syncthreads();
}

__global__ __launch_bounds__(256) void in_fwd_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                     float* __restrict__ stats, int HW, int C, int act) {
  __shared__ float red[2 * 4 * 64];
  const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
  const int n = blockIdx.x, c = blockIdx.y * 64 + cx;
  const bool cv = c < C;
  const float* xp = x + (size_t)n * HW * C + c;
  float s[1] = {0.f};
  if (cv)
    for (int p = py; p < HW; p += 4) s[0] += xp[(size_t)p * C];
  lane4_reduce<1>(s, red, cx, py);
  const float mean = s[0] / HW;
  float q[1] = {0.f};
  if (cv)
    for (int p = py; p < HW; p += 4) {
      const float d = xp[(size_t)p * C] - mean;
      q[0] += d * d;
    }
  lane4_reduce<1>(q, red, cx, py);
  const float sigma = sqrtf(q[0] / HW);
  const float r = 1.f / (sigma + IN_EPS);
  if (cv) {
    float* op = out + (size_t)n * HW * C + c;
    for (int p = py; p < HW; p += 4) op[(size_t)p * C] = act_apply((xp[(size_t)p * C] - mean) * r, act);
    if (py == 0) {
      stats[((size_t)n * C + c) * 2 + 0] = mean;
      stats[((size_t)n * C + c) * 2 + 1] = sigma;
    }
  }
}

// MODE 0: backward      res = J (dout * act'(out))                 -> dst (acc)
// MODE 1: tangent fwd   res = act'(out) * (J tx)                   -> dst
template <int MODE>
__global__ __launch_bounds__(256) void in_japply_kernel(const float* __restrict__ x, const float* __restrict__ out,
                                                        const float* __restrict__ v, const float* __restrict__ stats,
                                                        float* __restrict__ dst, int accumulate, int HW, int C,
                                                        int act) {
  __shared__ float red[2 * 4 * 64];
  const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
  const int n = blockIdx.x, c = blockIdx.y * 64 + cx;
  const bool cv = c < C;
  const size_t base = (size_t)n * HW * C + c;
  const float mean = cv ? stats[((size_t)n * C + c) * 2 + 0] : 0.f;
  const float sigma = cv ? stats[((size_t)n * C + c) * 2 + 1] : 1.f;
  const float r = 1.f / (sigma + IN_EPS);
  float s[2] = {0.f, 0.f};  // sum(e), sum(e*xc)
  if (cv)
    for (int p = py; p < HW; p += 4) {
      const size_t i = base + (size_t)p * C;
      float e = v[i];
      if (MODE == 0) e *= act_grad_from_out(out[i], act);
      s[0] += e;
      s[1] += e * (x[i] - mean);
    }
  lane4_reduce<2>(s, red, cx, py);
  const float eb = s[0] / HW;
  const float k2 = r * r * s[1] / (HW * sigma);
  if (cv)
    for (int p = py; p < HW; p += 4) {
      const size_t i = base + (size_t)p * C;
      const float o = out[i];
      float e = v[i];
      if (MODE == 0) e *= act_grad_from_out(o, act);
      float res = r * (e - eb) - k2 * (x[i] - mean);
      if (MODE == 1) res *= act_grad_from_out(o, act);
      dst[i] = accumulate ? dst[i] + res : res;
    }
}

__global__ __launch_bounds__(256) void in_tangent_bwd_kernel(const float* __restrict__ x, const float* __restrict__ out,
                                                             const float* __restrict__ stats,
                                                             const float* __restrict__ tx,
                                                             const float* __restrict__ dtout, float* __restrict__ dtx,
                                                             int acc_dtx, float* __restrict__ dx, int acc_dx, int HW,
                                                             int C, int act) {
  __shared__ float red[5 * 4 * 64];
  const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
  const int n = blockIdx.x, c = blockIdx.y * 64 + cx;
  const bool cv = c < C;
  const size_t base = (size_t)n * HW * C + c;
  const float mean = cv ? stats[((size_t)n * C + c) * 2 + 0] : 0.f;
  const float sigma = cv ? stats[((size_t)n * C + c) * 2 + 1] : 1.f;
  const float r = 1.f / (sigma + IN_EPS);
  // means first, then centred second moments (the uncentred form sum(e t) - HW mean(e) mean(t) cancels
  // catastrophically when the tangent / cotangent carry a large per-channel mean)
  float m[2] = {0.f, 0.f};
  if (cv)
    for (int p = py; p < HW; p += 4) {
      const size_t i = base + (size_t)p * C;
      m[0] += dtout[i] * act_grad_from_out(out[i], act);
      m[1] += tx[i];
    }
  lane4_reduce<2>(m, red, cx, py);
  const float hw = (float)HW;
  const float eb = m[0] / hw, ta = m[1] / hw;
  float s[3] = {0.f, 0.f, 0.f};  // sum (e-eb)(t-ta), sum xc*t, sum e*xc
  if (cv)
    for (int p = py; p < HW; p += 4) {
      const size_t i = base + (size_t)p * C;
      const float e = dtout[i] * act_grad_from_out(out[i], act);
      const float t = tx[i], xc = x[i] - mean;
      s[0] += (e - eb) * (t - ta);
      s[1] += xc * t;
      s[2] += e * xc;
    }
  lane4_reduce<3>(s, red, cx, py);
  const float A = s[0];
  const float S_xt = s[1], S_ex = s[2];
  const float B = S_xt * S_ex / hw;
  const float r2 = r * r;
  const float c1 = (-A * r2 + 2.f * r2 * r * B / sigma + r2 * B / (sigma * sigma)) / (hw * sigma);
  const float c2 = -r2 * S_ex / (hw * sigma);
  const float c3 = -r2 * S_xt / (hw * sigma);
  const float k2 = r2 * S_ex / (hw * sigma);
  if (cv)
    for (int p = py; p < HW; p += 4) {
      const size_t i = base + (size_t)p * C;
      const float e = dtout[i] * act_grad_from_out(out[i], act);
      const float t = tx[i], xc = x[i] - mean;
      const float gt = r * (e - eb) - k2 * xc;
      const float gx = c1 * xc + c2 * (t - ta) + c3 * (e - eb);
      dtx[i] = acc_dtx ? dtx[i] + gt : gt;
      dx[i] = acc_dx ? dx[i] + gx : gx;
    }
}


// =============================================================================================
// Vectorised per-(n,c) kernels: block = CQ channel quads (float4 = 4 channels) x 256/CQ pixel
// lanes; a wave reads CQ*16-byte contiguous channel runs of 64/CQ consecutive pixels; pixel-lane
// reduction by xor-shuffles inside the wave, then across the 4 waves through LDS.  Used when
// C % (4*CQ) == 0 (every IN / BN site of the model); the scalar kernels above remain the fallback.
// =============================================================================================
__device__ __forceinline__ float4 act4(float4 v, int act) {
  return make_float4(act_apply(v.x, act), act_apply(v.y, act), act_apply(v.z, act), act_apply(v.w, act));
}
__device__ __forceinline__ float4 actg4(float4 o, int act) {
  return make_float4(act_grad_from_out(o.x, act), act_grad_from_out(o.y, act), act_grad_from_out(o.z, act),
                     act_grad_from_out(o.w, act));
}

// bf16 plane of the tensor just stored (bf16 mode's second store; k_elem.hip: st_bf4 -- same rounding): element i of the plane
__device__ __forceinline__ void st_bf4n(unsigned short* __restrict__ bf, size_t i, const float4& v) {
  typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  const f32x2_ lo = {v.x, v.y}, hi = {v.z, v.w};
  uint2 o;
  o.x = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2_));
  o.y = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2_));
  *reinterpret_cast<uint2*>(bf + i) = o;
}

template <int CQ>
__global__ __launch_bounds__(256) void in_fwd_v(const float* __restrict__ x, float* __restrict__ out,
                                                float* __restrict__ stats, int HW, int C, int act,
                                                unsigned short* __restrict__ out_bf) {
  constexpr int PL = 256 / CQ;
  __shared__ float4 red[4 * CQ];
  const int cq = threadIdx.x % CQ, pl = threadIdx.x / CQ;
  const int n = blockIdx.x, c = (blockIdx.y * CQ + cq) * 4;
  const float* xp = x + (size_t)n * HW * C + c;
  float4 s[1] = {f4_zero()};
  for (int p = pl; p < HW; p += PL) {
    const float4 v = ld4(xp + (size_t)p * C);
    s[0].x += v.x; s[0].y += v.y; s[0].z += v.z; s[0].w += v.w;
  }
  cq_reduce<1, CQ>(s, red);
  const float inv = 1.f / HW;
  const float4 mean = make_float4(s[0].x * inv, s[0].y * inv, s[0].z * inv, s[0].w * inv);
  float4 q[1] = {f4_zero()};
  for (int p = pl; p < HW; p += PL) {
    const float4 v = ld4(xp + (size_t)p * C);
    const float dx = v.x - mean.x, dy = v.y - mean.y, dz = v.z - mean.z, dw = v.w - mean.w;
    q[0].x += dx * dx; q[0].y += dy * dy; q[0].z += dz * dz; q[0].w += dw * dw;
  }
  cq_reduce<1, CQ>(q, red);
  const float4 sigma = make_float4(sqrtf(q[0].x * inv), sqrtf(q[0].y * inv), sqrtf(q[0].z * inv), sqrtf(q[0].w * inv));
  const float4 r = make_float4(1.f / (sigma.x + IN_EPS), 1.f / (sigma.y + IN_EPS), 1.f / (sigma.z + IN_EPS),
                               1.f / (sigma.w + IN_EPS));
  float* op = out + (size_t)n * HW * C + c;
  for (int p = pl; p < HW; p += PL) {
    const float4 v = ld4(xp + (size_t)p * C);
    const float4 y = act4(make_float4((v.x - mean.x) * r.x, (v.y - mean.y) * r.y, (v.z - mean.z) * r.z, (v.w - mean.w) * r.w), act);
    if (out) st4(op + (size_t)p * C, y);   // (out == nullptr: the output is kept as its bf16 plane only, Tensor::d_plane_only)
    if (out_bf) st_bf4n(out_bf, (size_t)n * HW * C + c + (size_t)p * C, y);
  }
  if (pl == 0) {
    float* sp = stats + ((size_t)n * C + c) * 2;
    sp[0] = mean.x; sp[1] = sigma.x; sp[2] = mean.y; sp[3] = sigma.y;
    sp[4] = mean.z; sp[5] = sigma.z; sp[6] = mean.w; sp[7] = sigma.w;
  }
}

// register-cached form for HW <= 32 * NP: x is read from memory ONCE (the generic kernel re-reads it for the centred
// variance and for the apply pass; at 128 KB per block those re-reads miss L2)
template <int NP, int CQ>
__global__ __launch_bounds__(256) void in_fwd_vr(const float* __restrict__ x, float* __restrict__ out,
                                                 float* __restrict__ stats, int HW, int C, int act, int nslab,
                                                 size_t slab_stride, const float* __restrict__ slab_bias,
                                                 float* __restrict__ xsum, unsigned short* __restrict__ out_bf) {
  constexpr int PL = 256 / CQ;   // CQ = 4 for the 32x32 layers: 16 channels per block -> twice the blocks, half the registers
  __shared__ float4 red[4 * CQ];
  const int cq = threadIdx.x % CQ, pl = threadIdx.x / CQ;
  const int n = blockIdx.x, c = (blockIdx.y * CQ + cq) * 4;
  const float* xp = x + (size_t)n * HW * C + c;
  float4 xv[NP];
  float4 s[1] = {f4_zero()};
  if (nslab > 1) {   // an un-summed K-split convolution output: bias + slabs in conv_split_reduce's order.  Slab by slab with the
                     // block's NP loads of a slab in flight together (the per-pixel form -- load a slab's quad, add, next slab --
                     // was NP x nslab dependent round trips: 25 .. 60 us per block at any batch size)
    const float4 b4 = slab_bias ? ld4(slab_bias + c) : f4_zero();
#pragma unroll
    for (int i = 0; i < NP; ++i) xv[i] = b4;
    for (int k = 0; k < nslab; ++k) {
      const float* sp = xp + (size_t)k * slab_stride;
      float4 q[NP];
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int p = pl + i * PL;
        q[i] = ld4(sp + (size_t)(p < HW ? p : 0) * C);
      }
#pragma unroll
      for (int i = 0; i < NP; ++i) { xv[i].x += q[i].x; xv[i].y += q[i].y; xv[i].z += q[i].z; xv[i].w += q[i].w; }
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int p = pl + i * PL;
      if (p < HW) {
        // training: the summed pre-norm tensor is what the backward (and the gradient penalty's tangent pass) reads
        if (xsum) st4(xsum + (size_t)n * HW * C + c + (size_t)p * C, xv[i]);
      } else {
        xv[i] = f4_zero();
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int p = pl + i * PL;
      xv[i] = p < HW ? ld4(xp + (size_t)p * C) : f4_zero();
    }
  }
#pragma unroll
  for (int i = 0; i < NP; ++i) { s[0].x += xv[i].x; s[0].y += xv[i].y; s[0].z += xv[i].z; s[0].w += xv[i].w; }
  cq_reduce<1, CQ>(s, red);
  const float inv = 1.f / HW;
  const float4 mean = make_float4(s[0].x * inv, s[0].y * inv, s[0].z * inv, s[0].w * inv);
  float4 q[1] = {f4_zero()};
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    if (pl + i * PL < HW) {
      const float dx = xv[i].x - mean.x, dy = xv[i].y - mean.y, dz = xv[i].z - mean.z, dw = xv[i].w - mean.w;
      q[0].x += dx * dx; q[0].y += dy * dy; q[0].z += dz * dz; q[0].w += dw * dw;
    }
  }
  cq_reduce<1, CQ>(q, red);
  const float4 sigma = make_float4(sqrtf(q[0].x * inv), sqrtf(q[0].y * inv), sqrtf(q[0].z * inv), sqrtf(q[0].w * inv));
  const float4 r = make_float4(1.f / (sigma.x + IN_EPS), 1.f / (sigma.y + IN_EPS), 1.f / (sigma.z + IN_EPS),
                               1.f / (sigma.w + IN_EPS));
  float* op = out + (size_t)n * HW * C + c;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int p = pl + i * PL;
    if (p < HW) {
      const float4 y = act4(make_float4((xv[i].x - mean.x) * r.x, (xv[i].y - mean.y) * r.y, (xv[i].z - mean.z) * r.z,
                                        (xv[i].w - mean.w) * r.w), act);
      if (out) st4(op + (size_t)p * C, y);
      if (out_bf) st_bf4n(out_bf, (size_t)n * HW * C + c + (size_t)p * C, y);
    }
  }
  if (pl == 0) {
    float* sp = stats + ((size_t)n * C + c) * 2;
    sp[0] = mean.x; sp[1] = sigma.x; sp[2] = mean.y; sp[3] = sigma.y;
    sp[4] = mean.z; sp[5] = sigma.z; sp[6] = mean.w; sp[7] = sigma.w;
  }
}

struct Stat4 { float4 mean, sigma, r; };
__device__ __forceinline__ Stat4 load_stats4(const float* stats, size_t nc) {
  const float4 a = ld4(stats + nc * 2), b = ld4(stats + nc * 2 + 4);
  Stat4 s;
  s.mean = make_float4(a.x, a.z, b.x, b.z);
  s.sigma = make_float4(a.y, a.w, b.y, b.w);
  s.r = make_float4(1.f / (s.sigma.x + IN_EPS), 1.f / (s.sigma.y + IN_EPS), 1.f / (s.sigma.z + IN_EPS),
                    1.f / (s.sigma.w + IN_EPS));
  return s;
}

// act'(out) of out = act((x - mean) r), recomputed from the pre-norm input with the forward pass's own arithmetic (in_fwd_v /
// in_fwd_vr: the stored mean, r = 1 / (sigma + eps) from the stored sigma) -- the same bits as reading `out`, one tensor read
// less per element in every backward / tangent kernel below
__device__ __forceinline__ float4 actg_from_x(const float4& xv, const Stat4& st, int act) {
  return actg4(act4(make_float4((xv.x - st.mean.x) * st.r.x, (xv.y - st.mean.y) * st.r.y, (xv.z - st.mean.z) * st.r.z,
                                (xv.w - st.mean.w) * st.r.w), act), act);
}

template <int MODE, int CQ>
__global__ __launch_bounds__(256) void in_japply_v(const float* __restrict__ x, const float* __restrict__ out,
                                                   const float* __restrict__ v, const float* __restrict__ stats,
                                                   float* __restrict__ dst, int accumulate, int HW, int C, int act,
                                                   unsigned short* __restrict__ dst_bf) {
  constexpr int PL = 256 / CQ;
  __shared__ float4 red[2 * 4 * CQ];
  const int cq = threadIdx.x % CQ, pl = threadIdx.x / CQ;
  const int n = blockIdx.x, c = (blockIdx.y * CQ + cq) * 4;
  const size_t base = (size_t)n * HW * C + c;
  const Stat4 st = load_stats4(stats, (size_t)n * C + c);
  float4 s[2] = {f4_zero(), f4_zero()};
  // Both passes walk the pixels four at a time with the eight (twelve) loads of a group issued before the first use -- the one-pixel
  // loop was a memory round trip per iteration and pass (HW / PL = 32 .. 128 of them per block); same order of every sum.
  auto reduce1 = [&](float4 e, const float4& xv) __attribute__((always_inline)) {
    if (MODE == 0) {
      const float4 g = actg_from_x(xv, st, act);
      e.x *= g.x; e.y *= g.y; e.z *= g.z; e.w *= g.w;
    }
    s[0].x += e.x; s[0].y += e.y; s[0].z += e.z; s[0].w += e.w;
    s[1].x += e.x * (xv.x - st.mean.x); s[1].y += e.y * (xv.y - st.mean.y);
    s[1].z += e.z * (xv.z - st.mean.z); s[1].w += e.w * (xv.w - st.mean.w);
  };
  {
    int p = pl;
    for (; p + 3 * PL < HW; p += 4 * PL) {
      float4 e[4], xv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const size_t i = base + (size_t)(p + u * PL) * C;
        e[u] = ld4(v + i);
        xv[u] = ld4(x + i);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) reduce1(e[u], xv[u]);
    }
    for (; p < HW; p += PL) {
      const size_t i = base + (size_t)p * C;
      reduce1(ld4(v + i), ld4(x + i));
    }
  }
  cq_reduce<2, CQ>(s, red);
  const float hw = (float)HW;
  const float4 eb = make_float4(s[0].x / hw, s[0].y / hw, s[0].z / hw, s[0].w / hw);
  const float4 k2 = make_float4(st.r.x * st.r.x * s[1].x / (hw * st.sigma.x), st.r.y * st.r.y * s[1].y / (hw * st.sigma.y),
                                st.r.z * st.r.z * s[1].z / (hw * st.sigma.z), st.r.w * st.r.w * s[1].w / (hw * st.sigma.w));
  auto apply1 = [&](size_t i, const float4& xv, float4 e, const float4& d) __attribute__((always_inline)) {
    const float4 g = actg_from_x(xv, st, act);
    if (MODE == 0) { e.x *= g.x; e.y *= g.y; e.z *= g.z; e.w *= g.w; }
    float4 res = make_float4(st.r.x * (e.x - eb.x) - k2.x * (xv.x - st.mean.x), st.r.y * (e.y - eb.y) - k2.y * (xv.y - st.mean.y),
                             st.r.z * (e.z - eb.z) - k2.z * (xv.z - st.mean.z), st.r.w * (e.w - eb.w) - k2.w * (xv.w - st.mean.w));
    if (MODE == 1) { res.x *= g.x; res.y *= g.y; res.z *= g.z; res.w *= g.w; }
    if (accumulate & 1) { res.x += d.x; res.y += d.y; res.z += d.z; res.w += d.w; }
    if (dst && !(accumulate & 2)) st4(dst + i, res);   // (plane-only: the tangent-forward output, or GRAD_PLANE_ONLY)
    if (dst_bf) st_bf4n(dst_bf, i, res);
  };
  {
    int p = pl;
    for (; p + 3 * PL < HW; p += 4 * PL) {
      float4 e[4], xv[4], d[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const size_t i = base + (size_t)(p + u * PL) * C;
        xv[u] = ld4(x + i);
        e[u] = ld4(v + i);
        d[u] = (accumulate & 1) ? ld4(dst + i) : f4_zero();
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) apply1(base + (size_t)(p + u * PL) * C, xv[u], e[u], d[u]);
    }
    for (; p < HW; p += PL) {
      const size_t i = base + (size_t)p * C;
      apply1(i, ld4(x + i), ld4(v + i), (accumulate & 1) ? ld4(dst + i) : f4_zero());
    }
  }
}

// IN+act backward with e = dout * act'(out) and (x - mean) held in registers between the reduction and the apply
// pass (HW <= 32 * NP): out / dout / x are read from memory once
// v (the output gradient) may still be the K-split slabs of the input-gradient convolution that produced it (SlabSrc)
template <int NP, int CQ>
__global__ __launch_bounds__(256, (NP == 8 ? 4 : NP == 16 ? 2 : 1)) void in_bwd_vr(const float* __restrict__ x, const float* __restrict__ out,
                                                 const SlabSrc v, const float* __restrict__ stats,
                                                 float* __restrict__ dst, int accumulate, int HW, int C, int act,
                                                 unsigned short* __restrict__ dst_bf, float* __restrict__ colpart) {
  constexpr int PL = 256 / CQ;
  __shared__ float4 red[2 * 4 * CQ];
  const int cq = threadIdx.x % CQ, pl = threadIdx.x / CQ;
  const int n = blockIdx.x, c = (blockIdx.y * CQ + cq) * 4;
  const size_t base = (size_t)n * HW * C + c;
  const Stat4 st = load_stats4(stats, (size_t)n * C + c);
  float4 ev[NP], xc[NP];
  float4 s[2] = {f4_zero(), f4_zero()};
  // Round 6: ALL of the block's loads are issued before the first value is used.  The per-element form (load e, load x, use, next
  // element) compiled to one s_waitcnt vmcnt(0) per element -- 16 dependent memory round trips per block, ~30 us whatever the batch
  // (the same 31 us at batch 4 and at batch 64).  Out-of-range pixels read pixel 0 and are zeroed afterwards (no branch around a
  // load); a slab input is summed slab by slab (bias, slab 0, slab 1, ..., base: slab_ld4's order) with NP loads in flight per slab.
  auto off = [&](int k) __attribute__((always_inline)) {   // (recomputed at every use: 2 registers of address per element otherwise)
    const int p = pl + k * PL;
    return base + (size_t)(p < HW ? p : 0) * C;
  };
#pragma unroll
  for (int k = 0; k < NP; ++k) xc[k] = ld4(x + off(k));
  if (v.n <= 1) {
#pragma unroll
    for (int k = 0; k < NP; ++k) ev[k] = ld4(v.p + off(k));
  } else {
    const float4 b4 = v.bias ? ld4(v.bias + c) : f4_zero();
#pragma unroll
    for (int k = 0; k < NP; ++k) ev[k] = b4;
    constexpr int NQ = NP > 4 ? NP / 2 : NP;   // loads in flight per batch of the slab sums (registers: the whole set costs a wave of occupancy)
    for (int z = 0; z <= v.n; ++z) {       // (z == v.n: the base, last -- slab_ld4's order)
      if (z == v.n && !v.base) break;
      const float* src = z < v.n ? v.p + (size_t)z * v.stride : v.base;
#pragma unroll
      for (int h = 0; h < NP; h += NQ) {
        float4 q[NQ];
#pragma unroll
        for (int k = 0; k < NQ; ++k) q[k] = ld4(src + off(h + k));
#pragma unroll
        for (int k = 0; k < NQ; ++k) { ev[h + k].x += q[k].x; ev[h + k].y += q[k].y; ev[h + k].z += q[k].z; ev[h + k].w += q[k].w; }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int p = pl + k * PL;
    if (p < HW) {
      const float4 xv = xc[k];
      float4 e = ev[k];
      const float4 g = actg_from_x(xv, st, act);
      e.x *= g.x; e.y *= g.y; e.z *= g.z; e.w *= g.w;
      ev[k] = e;
      xc[k] = make_float4(xv.x - st.mean.x, xv.y - st.mean.y, xv.z - st.mean.z, xv.w - st.mean.w);
      s[0].x += e.x; s[0].y += e.y; s[0].z += e.z; s[0].w += e.w;
      s[1].x += e.x * xc[k].x; s[1].y += e.y * xc[k].y; s[1].z += e.z * xc[k].z; s[1].w += e.w * xc[k].w;
    } else {
      ev[k] = f4_zero();
      xc[k] = f4_zero();
    }
  }
  cq_reduce<2, CQ>(s, red);
  const float hw = (float)HW;
  const float4 eb = make_float4(s[0].x / hw, s[0].y / hw, s[0].z / hw, s[0].w / hw);
  const float4 k2 = make_float4(st.r.x * st.r.x * s[1].x / (hw * st.sigma.x), st.r.y * st.r.y * s[1].y / (hw * st.sigma.y),
                                st.r.z * st.r.z * s[1].z / (hw * st.sigma.z), st.r.w * st.r.w * s[1].w / (hw * st.sigma.w));
  float4 cs[1] = {f4_zero()};
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int p = pl + k * PL;
    if (p < HW) {
      const size_t i = base + (size_t)p * C;
      float4 res = make_float4(st.r.x * (ev[k].x - eb.x) - k2.x * xc[k].x, st.r.y * (ev[k].y - eb.y) - k2.y * xc[k].y,
                               st.r.z * (ev[k].z - eb.z) - k2.z * xc[k].z, st.r.w * (ev[k].w - eb.w) - k2.w * xc[k].w);
      if (accumulate & 1) {
        const float4 d = ld4(dst + i);
        res.x += d.x; res.y += d.y; res.z += d.z; res.w += d.w;
      }
      if (!(accumulate & 2)) st4(dst + i, res);   // (GRAD_PLANE_ONLY)
      if (dst_bf) st_bf4n(dst_bf, i, res);
      cs[0].x += res.x; cs[0].y += res.y; cs[0].z += res.z; cs[0].w += res.w;
    }
  }
  if (colpart) {   // this (n, channel quad)'s sum over H x W of the gradient just written: row n of the bias gradient's first stage
    cq_reduce<1, CQ>(cs, red);
    if (pl == 0) st4(colpart + (size_t)n * C + c, cs[0]);
  }
}

// =============================================================================================
// InstanceNorm on LARGE maps (H W > 1,024: configs[4]'s 256 x 512 geometry -- 64 x 128 .. 128 x 256 pixel maps).  The per-(n, c)
// kernels above give one block a whole map: at batch 16 that is 32 blocks walking 32,768 pixels each, three passes, one memory
// round trip per iteration -- 294 us per launch, 22 % of the iteration's kernel time at 256 x 512 (round 6).  Here the pixels of
// a map are split into chunks of IN_RPC rows, every reduction is two-stage (chunk partials, then a per-(n, c) combine in chunk
// order: fixed order, no atomics; statistics as per-chunk mean + centred second moment merged pairwise), and the apply passes are
// plain element-wise kernels over quads.  Same formulas as in_fwd_v / in_japply_v (eps outside the sqrt: normalization.py:16).
// =============================================================================================
constexpr int IN_RPC = 256;
__global__ __launch_bounds__(256) void in_chunk_stats_kernel(const float* __restrict__ x, float* __restrict__ part, int HW, int C) {
  __shared__ float red[1 * 4 * 64];
  const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + cx, n = blockIdx.z, nch = gridDim.x;
  const bool cv = c < C;
  const int r0 = blockIdx.x * IN_RPC, r1 = min(HW, r0 + IN_RPC);
  const float* xp = x + (size_t)n * HW * C + c;
  // (the chunk's IN_RPC / 4 = 64 rows of a thread stay in registers between the two passes: one load each, all in flight)
  constexpr int NR = IN_RPC / 4;
  float xv[NR];
  float s[1] = {0.f};
  if (cv) {
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int p = r0 + py + 4 * k;
      xv[k] = p < r1 ? xp[(size_t)p * C] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < NR; ++k) s[0] += xv[k];
  }
  lane4_reduce<1>(s, red, cx, py);
  const float mean = s[0] / (float)(r1 - r0);
  float q[1] = {0.f};
  if (cv) {
#pragma unroll
    for (int k = 0; k < NR; ++k)
      if (r0 + py + 4 * k < r1) {
        const float d = xv[k] - mean;
        q[0] += d * d;
      }
  }
  lane4_reduce<1>(q, red, cx, py);
  if (cv && py == 0) {
    float* o = part + (((size_t)n * nch + blockIdx.x) * C + c) * 2;
    o[0] = mean;
    o[1] = q[0];
  }
}
__global__ void in_finalize_stats_kernel(const float* __restrict__ part, int nch, int HW, int C, float* __restrict__ stats) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
  if (c >= C) return;
  float cnt = 0.f, mean = 0.f, m2 = 0.f;
  for (int k0 = 0; k0 < nch; k0 += 8) {   // (eight chunks' partials in flight; the merge itself stays in chunk order)
    float2 pv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      pv[u] = *reinterpret_cast<const float2*>(part + (((size_t)n * nch + min(k0 + u, nch - 1)) * C + c) * 2);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 + u;
      if (k < nch) {
        const float nk = (float)min(IN_RPC, HW - k * IN_RPC);
        const float d = pv[u].x - mean, nt = cnt + nk;
        mean += d * (nk / nt);
        m2 += pv[u].y + d * d * (cnt * nk / nt);
        cnt = nt;
      }
    }
  }
  stats[((size_t)n * C + c) * 2] = mean;
  stats[((size_t)n * C + c) * 2 + 1] = sqrtf(m2 / (float)HW);
}
// y = act((x - mean) r) over the quads of image blockIdx.y
__global__ __launch_bounds__(256) void in_apply_chunked_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                               float* __restrict__ out, unsigned short* __restrict__ out_bf, int HW,
                                                               int C, int act) {
  const int n = blockIdx.y, c4n = C >> 2;
  const size_t per = (size_t)HW * c4n, ib = (size_t)n * HW * C;
  for (size_t q = blockIdx.x * (size_t)256 + threadIdx.x; q < per; q += (size_t)gridDim.x * 256) {
    const int c = (int)(q % c4n) * 4;
    const size_t i = ib + q * 4;
    const Stat4 st = load_stats4(stats, (size_t)n * C + c);
    const float4 v = ld4(x + i);
    const float4 y = act4(make_float4((v.x - st.mean.x) * st.r.x, (v.y - st.mean.y) * st.r.y, (v.z - st.mean.z) * st.r.z,
                                      (v.w - st.mean.w) * st.r.w), act);
    if (out) st4(out + i, y);
    if (out_bf) st_bf4n(out_bf, i, y);
  }
}
// chunk partials of sum e and sum e xc;  MODE 0 (backward): e = v act'(out), MODE 1 (forward tangent): e = v
template <int MODE>
__global__ __launch_bounds__(256) void in_chunk_j_kernel(const float* __restrict__ x, const float* __restrict__ v,
                                                         const float* __restrict__ stats, float* __restrict__ part, int HW, int C,
                                                         int act) {
  __shared__ float red[2 * 4 * 64];
  const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + cx, n = blockIdx.z, nch = gridDim.x;
  const bool cv = c < C;
  const int r0 = blockIdx.x * IN_RPC, r1 = min(HW, r0 + IN_RPC);
  const size_t base = (size_t)n * HW * C + c;
  const float mean = cv ? stats[((size_t)n * C + c) * 2] : 0.f;
  const float r = cv ? 1.f / (stats[((size_t)n * C + c) * 2 + 1] + IN_EPS) : 1.f;
  float s[2] = {0.f, 0.f};
  if (cv)
    for (int p0 = r0 + py; p0 < r1; p0 += 32) {   // eight rows per step, their sixteen loads issued together
      float xr[8], er[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int p = p0 + 4 * u;
        const size_t i = base + (size_t)(p < r1 ? p : r0) * C;
        xr[u] = x[i];
        er[u] = v[i];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (p0 + 4 * u < r1) {
          const float xc = xr[u] - mean;
          float e = er[u];
          if (MODE == 0) e *= act_grad_from_out(act_apply(xc * r, act), act);
          s[0] += e;
          s[1] += e * xc;
        }
    }
  lane4_reduce<2>(s, red, cx, py);
  if (cv && py == 0) {
    float* o = part + (((size_t)n * nch + blockIdx.x) * C + c) * 2;
    o[0] = s[0];
    o[1] = s[1];
  }
}
__global__ void in_finalize_j_kernel(const float* __restrict__ part, int nch, int C, float* __restrict__ sums) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
  if (c >= C) return;
  float a = 0.f, b = 0.f;
  for (int k0 = 0; k0 < nch; k0 += 8) {   // (eight chunks' partials in flight, summed in chunk order)
    float2 pv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      pv[u] = *reinterpret_cast<const float2*>(part + (((size_t)n * nch + min(k0 + u, nch - 1)) * C + c) * 2);
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (k0 + u < nch) {
        a += pv[u].x;
        b += pv[u].y;
      }
  }
  sums[((size_t)n * C + c) * 2] = a;
  sums[((size_t)n * C + c) * 2 + 1] = b;
}
// res = r (e - mean(e)) - r^2 sum(e xc) / (HW sigma) xc;  MODE 0: dst (+)= res;  MODE 1: dst = act'(out) res
template <int MODE>
__global__ __launch_bounds__(256) void in_japply_chunked_kernel(const float* __restrict__ x, const float* __restrict__ v,
                                                                const float* __restrict__ stats, const float* __restrict__ sums,
                                                                float* __restrict__ dst, int accumulate,
                                                                unsigned short* __restrict__ dst_bf, int HW, int C, int act) {
  const int n = blockIdx.y, c4n = C >> 2;
  const size_t per = (size_t)HW * c4n, ib = (size_t)n * HW * C;
  const float hw = (float)HW;
  for (size_t q = blockIdx.x * (size_t)256 + threadIdx.x; q < per; q += (size_t)gridDim.x * 256) {
    const int c = (int)(q % c4n) * 4;
    const size_t i = ib + q * 4;
    const Stat4 st = load_stats4(stats, (size_t)n * C + c);
    const float4 sa = ld4(sums + ((size_t)n * C + c) * 2), sb = ld4(sums + ((size_t)n * C + c) * 2 + 4);
    const float4 eb = make_float4(sa.x / hw, sa.z / hw, sb.x / hw, sb.z / hw);
    const float4 k2 = make_float4(st.r.x * st.r.x * sa.y / (hw * st.sigma.x), st.r.y * st.r.y * sa.w / (hw * st.sigma.y),
                                  st.r.z * st.r.z * sb.y / (hw * st.sigma.z), st.r.w * st.r.w * sb.w / (hw * st.sigma.w));
    const float4 xv = ld4(x + i);
    const float4 g = actg_from_x(xv, st, act);
    float4 e = ld4(v + i);
    if (MODE == 0) { e.x *= g.x; e.y *= g.y; e.z *= g.z; e.w *= g.w; }
    float4 res = make_float4(st.r.x * (e.x - eb.x) - k2.x * (xv.x - st.mean.x), st.r.y * (e.y - eb.y) - k2.y * (xv.y - st.mean.y),
                             st.r.z * (e.z - eb.z) - k2.z * (xv.z - st.mean.z), st.r.w * (e.w - eb.w) - k2.w * (xv.w - st.mean.w));
    if (MODE == 1) { res.x *= g.x; res.y *= g.y; res.z *= g.z; res.w *= g.w; }
    if (accumulate & 1) {
      const float4 d = ld4(dst + i);
      res.x += d.x; res.y += d.y; res.z += d.z; res.w += d.w;
    }
    if (dst && !(accumulate & 2)) st4(dst + i, res);
    if (dst_bf) st_bf4n(dst_bf, i, res);
  }
}
// reverse of the forward tangent on large maps, three stages (as in_tangent_bwd_v: means first, then the centred moments -- the
// uncentred forms cancel): partials of (sum e, sum t), then of (A, S_xt, S_ex), then the element-wise apply
template <int STAGE>
__global__ __launch_bounds__(256) void in_chunk_tb_kernel(const float* __restrict__ x, const float* __restrict__ tx,
                                                          const float* __restrict__ dtout, const float* __restrict__ stats,
                                                          const float* __restrict__ means, float* __restrict__ part, int HW, int C,
                                                          int act) {
  constexpr int NV = STAGE == 0 ? 2 : 3;
  __shared__ float red[NV * 4 * 64];
  const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + cx, n = blockIdx.z, nch = gridDim.x;
  const bool cv = c < C;
  const int r0 = blockIdx.x * IN_RPC, r1 = min(HW, r0 + IN_RPC);
  const size_t base = (size_t)n * HW * C + c;
  const float mean = cv ? stats[((size_t)n * C + c) * 2] : 0.f;
  const float r = cv ? 1.f / (stats[((size_t)n * C + c) * 2 + 1] + IN_EPS) : 1.f;
  const float eb = (STAGE == 1 && cv) ? means[((size_t)n * C + c) * 2] / (float)HW : 0.f;
  const float ta = (STAGE == 1 && cv) ? means[((size_t)n * C + c) * 2 + 1] / (float)HW : 0.f;
  float s[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) s[k] = 0.f;
  if (cv)
    for (int p0 = r0 + py; p0 < r1; p0 += 16) {   // four rows per step, their twelve loads issued together
      float xr[4], tr[4], dr[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int p = p0 + 4 * u;
        const size_t i = base + (size_t)(p < r1 ? p : r0) * C;
        xr[u] = x[i];
        tr[u] = tx[i];
        dr[u] = dtout[i];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (p0 + 4 * u < r1) {
          const float xc = xr[u] - mean, t = tr[u];
          const float e = dr[u] * act_grad_from_out(act_apply(xc * r, act), act);
          if (STAGE == 0) {
            s[0] += e;
            s[1] += t;
          } else {
            s[0] += (e - eb) * (t - ta);
            s[1] += xc * t;
            s[NV - 1] += e * xc;
          }
        }
    }
  lane4_reduce<NV>(s, red, cx, py);
  if (cv && py == 0) {
    float* o = part + (((size_t)n * nch + blockIdx.x) * C + c) * NV;
#pragma unroll
    for (int k = 0; k < NV; ++k) o[k] = s[k];
  }
}
template <int NV>
__global__ void in_finalize_n_kernel(const float* __restrict__ part, int nch, int C, float* __restrict__ sums) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
  if (c >= C) return;
  float a[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) a[k] = 0.f;
  for (int z = 0; z < nch; ++z) {
    const float* o = part + (((size_t)n * nch + z) * C + c) * NV;
#pragma unroll
    for (int k = 0; k < NV; ++k) a[k] += o[k];
  }
#pragma unroll
  for (int k = 0; k < NV; ++k) sums[((size_t)n * C + c) * NV + k] = a[k];
}
__global__ __launch_bounds__(256) void in_tb_apply_chunked_kernel(const float* __restrict__ x, const float* __restrict__ tx,
                                                                  const float* __restrict__ dtout, const float* __restrict__ stats,
                                                                  const float* __restrict__ means, const float* __restrict__ moments,
                                                                  float* __restrict__ dtx, int acc_dtx, float* __restrict__ dx,
                                                                  int acc_dx, unsigned short* __restrict__ dtx_bf, int HW, int C,
                                                                  int act) {
  const int n = blockIdx.y, c4n = C >> 2;
  const size_t per = (size_t)HW * c4n, ib = (size_t)n * HW * C;
  const float hw = (float)HW;
  for (size_t q = blockIdx.x * (size_t)256 + threadIdx.x; q < per; q += (size_t)gridDim.x * 256) {
    const int c = (int)(q % c4n) * 4;
    const size_t i = ib + q * 4;
    const Stat4 st = load_stats4(stats, (size_t)n * C + c);
    const float* mp = means + ((size_t)n * C + c) * 2;
    const float* qp = moments + ((size_t)n * C + c) * 3;
    const float sg[4] = {st.sigma.x, st.sigma.y, st.sigma.z, st.sigma.w}, rr[4] = {st.r.x, st.r.y, st.r.z, st.r.w};
    const float4 xv = ld4(x + i), tv = ld4(tx + i), dv = ld4(dtout + i), g = actg_from_x(xv, st, act);
    const float e[4] = {dv.x * g.x, dv.y * g.y, dv.z * g.z, dv.w * g.w}, t[4] = {tv.x, tv.y, tv.z, tv.w};
    const float xc[4] = {xv.x - st.mean.x, xv.y - st.mean.y, xv.z - st.mean.z, xv.w - st.mean.w};
    float gt[4], gx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float eb = mp[2 * j] / hw, ta = mp[2 * j + 1] / hw;
      const float A = qp[3 * j], Sxt = qp[3 * j + 1], Sex = qp[3 * j + 2];
      const float B = Sxt * Sex / hw, r2 = rr[j] * rr[j];
      const float c1 = (-A * r2 + 2.f * r2 * rr[j] * B / sg[j] + r2 * B / (sg[j] * sg[j])) / (hw * sg[j]);
      const float c2 = -r2 * Sex / (hw * sg[j]), c3 = -r2 * Sxt / (hw * sg[j]), k2 = r2 * Sex / (hw * sg[j]);
      gt[j] = rr[j] * (e[j] - eb) - k2 * xc[j];
      gx[j] = c1 * xc[j] + c2 * (t[j] - ta) + c3 * (e[j] - eb);
    }
    float4 o1 = make_float4(gt[0], gt[1], gt[2], gt[3]), o2 = make_float4(gx[0], gx[1], gx[2], gx[3]);
    if (acc_dtx & 1) { const float4 a = ld4(dtx + i); o1.x += a.x; o1.y += a.y; o1.z += a.z; o1.w += a.w; }
    if (acc_dx) { const float4 a = ld4(dx + i); o2.x += a.x; o2.y += a.y; o2.z += a.z; o2.w += a.w; }
    if (!(acc_dtx & 2)) st4(dtx + i, o1);   // (GRAD_PLANE_ONLY)
    if (dtx_bf) st_bf4n(dtx_bf, i, o1);
    st4(dx + i, o2);
  }
}
static inline int in_chunks(int HW) { return (HW + IN_RPC - 1) / IN_RPC; }
bool in_chunked_shape(int HW, int C) { return HW > 1024 && C % 32 == 0; }
size_t in_chunk_scratch_floats(int NB, int HW, int C) { return (size_t)NB * in_chunks(HW) * C * 3 + (size_t)NB * C * 5; }   // partials (<= 3 sums) + means (2) + moments (3)
static inline int in_ew_blocks(int HW, int C) { return (int)std::min<size_t>(2048, ((size_t)HW * (C >> 2) + 255) / 256); }
static void in_fwd_chunked(const float* x, float* out, float* stats, int NB, int HW, int C, int act, hipStream_t s,
                           unsigned short* out_bf, float* scratch) {
  const int nch = in_chunks(HW);
  hipLaunchKernelGGL(in_chunk_stats_kernel, dim3(nch, (C + 63) / 64, NB), dim3(256), 0, s, x, scratch, HW, C);
  hipLaunchKernelGGL(in_finalize_stats_kernel, dim3((C + 63) / 64, NB), dim3(64), 0, s, scratch, nch, HW, C, stats);
  hipLaunchKernelGGL(in_apply_chunked_kernel, dim3(in_ew_blocks(HW, C), NB), dim3(256), 0, s, x, stats, out, out_bf, HW, C, act);
}
template <int MODE>
static void in_japply_chunked(const float* x, const float* v, const float* stats, float* dst, int accumulate, int NB, int HW, int C,
                              int act, hipStream_t s, unsigned short* dst_bf, float* scratch) {
  const int nch = in_chunks(HW);
  float* sums = scratch + (size_t)NB * nch * C * 3;
  hipLaunchKernelGGL(in_chunk_j_kernel<MODE>, dim3(nch, (C + 63) / 64, NB), dim3(256), 0, s, x, v, stats, scratch, HW, C, act);
  hipLaunchKernelGGL(in_finalize_j_kernel, dim3((C + 63) / 64, NB), dim3(64), 0, s, scratch, nch, C, sums);
  hipLaunchKernelGGL(in_japply_chunked_kernel<MODE>, dim3(in_ew_blocks(HW, C), NB), dim3(256), 0, s, x, v, stats, sums, dst,
                     accumulate, dst_bf, HW, C, act);
}

void launch_in_fwd(const float* x, float* out, float* stats, int NB, int HW, int C, int act, hipStream_t s, int nslab,
                   size_t slab_stride, const float* slab_bias, float* xsum, unsigned short* out_bf, float* chunk_scratch) {
  EG_CHECK(!out_bf || C % 32 == 0, "in_fwd: a bf16 plane was requested with C %% 32 != 0");
  EG_CHECK(out || out_bf, "in_fwd: neither an fp32 output nor a plane");
  if (chunk_scratch && in_chunked_shape(HW, C) && nslab <= 1) {   // large maps: chunked two-stage form
    in_fwd_chunked(x, out, stats, NB, HW, C, act, s, out_bf, chunk_scratch);
    return;
  }
  if (C % 32 == 0) {
    const dim3 grid(NB, C / 32);
#define EG_IN_FWD(NP, CQ, G) hipLaunchKernelGGL((in_fwd_vr<NP, CQ>), G, dim3(256), 0, s, x, out, stats, HW, C, act, nslab, slab_stride, slab_bias, xsum, out_bf)
    if (HW <= 64) EG_IN_FWD(2, 8, grid);
    else if (HW <= 128) EG_IN_FWD(4, 8, grid);
    else if (HW <= 256) EG_IN_FWD(8, 8, grid);
    else if (HW <= 512) EG_IN_FWD(16, 8, grid);
    else if (HW <= 1024) EG_IN_FWD(16, 4, dim3(NB, C / 16));
    else hipLaunchKernelGGL(in_fwd_v<8>, grid, dim3(256), 0, s, x, out, stats, HW, C, act, out_bf);
#undef EG_IN_FWD
    return;
  }

  hipLaunchKernelGGL(in_fwd_kernel, dim3(NB, (C + 63) / 64), dim3(256), 0, s, x, out, stats, HW, C, act);
}
bool in_bwd_takes_slabs(int HW, int C) { return C % 32 == 0 && HW <= 1024; }   // the register-resident forms
int in_bwd_colpart_rows(int NB, int HW, int C) { return in_bwd_takes_slabs(HW, C) ? NB : 0; }
void launch_in_bwd(const float* x, const float* out, const float* dout, const float* stats, float* dx,
                   int accumulate, int NB, int HW, int C, int act, hipStream_t s, unsigned short* dx_bf,
                   const SlabSrc* dout_slabs, float* colpart, float* chunk_scratch) {
  if (chunk_scratch && in_chunked_shape(HW, C) && !dout_slabs && !colpart) {   // large maps: chunked two-stage form
    EG_CHECK(!(accumulate & GRAD_PLANE_ONLY) || dx_bf, "in_bwd: GRAD_PLANE_ONLY without a plane");
    in_japply_chunked<0>(x, dout, stats, dx, accumulate, NB, HW, C, act, s, dx_bf, chunk_scratch);
    return;
  }
  EG_CHECK(!colpart || in_bwd_colpart_rows(NB, HW, C), "in_bwd: column partials need C %% 32 == 0 and HW <= 1024");
  EG_CHECK(!dx_bf || C % 32 == 0, "in_bwd: a bf16 plane was requested with C %% 32 != 0");
  EG_CHECK(!(accumulate & GRAD_PLANE_ONLY) || dx_bf, "in_bwd: GRAD_PLANE_ONLY without a plane");
  EG_CHECK(!dout_slabs || in_bwd_takes_slabs(HW, C), "in_bwd: a slab input needs C %% 32 == 0 and HW <= 1024");
  if (C % 32 == 0) {
    const dim3 grid(NB, C / 32);
    const SlabSrc dsrc = dout_slabs ? *dout_slabs : plain_src(dout);
#define EG_IN_BWD(NP) hipLaunchKernelGGL((in_bwd_vr<NP, 8>), grid, dim3(256), 0, s, x, out, dsrc, stats, dx, accumulate, HW, C, act, dx_bf, colpart)
    if (HW <= 64) EG_IN_BWD(2);
    else if (HW <= 128) EG_IN_BWD(4);
    else if (HW <= 256) EG_IN_BWD(8);
    else if (HW <= 512) EG_IN_BWD(16);
    else if (HW <= 1024)
      hipLaunchKernelGGL((in_bwd_vr<16, 4>), dim3(NB, C / 16), dim3(256), 0, s, x, out, dsrc, stats, dx, accumulate, HW,
                         C, act, dx_bf, colpart);
    else
      hipLaunchKernelGGL((in_japply_v<0, 8>), grid, dim3(256), 0, s, x, out, dout, stats, dx, accumulate, HW, C, act, dx_bf);
#undef EG_IN_BWD
    return;
  }
  hipLaunchKernelGGL(in_japply_kernel<0>, dim3(NB, (C + 63) / 64), dim3(256), 0, s, x, out, dout, stats, dx,
                     accumulate, HW, C, act);
}
void launch_in_tangent_fwd(const float* x, const float* out, const float* stats, const float* tx, float* tout,
                           int NB, int HW, int C, int act, hipStream_t s, unsigned short* tout_bf, float* chunk_scratch) {
  if (chunk_scratch && in_chunked_shape(HW, C)) {   // large maps: chunked two-stage form
    in_japply_chunked<1>(x, tx, stats, tout, 0, NB, HW, C, act, s, tout_bf, chunk_scratch);
    return;
  }
  if (C % 32 == 0) {
    hipLaunchKernelGGL((in_japply_v<1, 8>), dim3(NB, C / 32), dim3(256), 0, s, x, out, tx, stats, tout, 0, HW, C, act, tout_bf);
    return;
  }
  hipLaunchKernelGGL(in_japply_kernel<1>, dim3(NB, (C + 63) / 64), dim3(256), 0, s, x, out, tx, stats, tout, 0, HW,
                     C, act);
}
template <int CQ>
__global__ __launch_bounds__(256) void in_tangent_bwd_v(const float* __restrict__ x, const float* __restrict__ out,
                                                        const float* __restrict__ stats, const float* __restrict__ tx,
                                                        const float* __restrict__ dtout, float* __restrict__ dtx,
                                                        int acc_dtx, float* __restrict__ dx, int acc_dx, int HW, int C,
                                                        int act, unsigned short* __restrict__ dtx_bf) {
  constexpr int PL = 256 / CQ;
  __shared__ float4 red[3 * 4 * CQ];
  const int cq = threadIdx.x % CQ, pl = threadIdx.x / CQ;
  const int n = blockIdx.x, c = (blockIdx.y * CQ + cq) * 4;
  const size_t base = (size_t)n * HW * C + c;
  const Stat4 st = load_stats4(stats, (size_t)n * C + c);
  float4 m[2] = {f4_zero(), f4_zero()};
  for (int p = pl; p < HW; p += PL) {
    const size_t i = base + (size_t)p * C;
    const float4 g = actg_from_x(ld4(x + i), st, act), d = ld4(dtout + i), t = ld4(tx + i);
    m[0].x += d.x * g.x; m[0].y += d.y * g.y; m[0].z += d.z * g.z; m[0].w += d.w * g.w;
    m[1].x += t.x; m[1].y += t.y; m[1].z += t.z; m[1].w += t.w;
  }
  cq_reduce<2, CQ>(m, red);
  const float hw = (float)HW;
  const float4 eb = make_float4(m[0].x / hw, m[0].y / hw, m[0].z / hw, m[0].w / hw);
  const float4 ta = make_float4(m[1].x / hw, m[1].y / hw, m[1].z / hw, m[1].w / hw);
  float4 s[3] = {f4_zero(), f4_zero(), f4_zero()};  // sum (e-eb)(t-ta), sum xc*t, sum e*xc
  for (int p = pl; p < HW; p += PL) {
    const size_t i = base + (size_t)p * C;
    const float4 d = ld4(dtout + i), t = ld4(tx + i), xv = ld4(x + i), g = actg_from_x(xv, st, act);
    const float ex = d.x * g.x, ey = d.y * g.y, ez = d.z * g.z, ew = d.w * g.w;
    const float cx = xv.x - st.mean.x, cy = xv.y - st.mean.y, cz = xv.z - st.mean.z, cw = xv.w - st.mean.w;
    s[0].x += (ex - eb.x) * (t.x - ta.x); s[0].y += (ey - eb.y) * (t.y - ta.y);
    s[0].z += (ez - eb.z) * (t.z - ta.z); s[0].w += (ew - eb.w) * (t.w - ta.w);
    s[1].x += cx * t.x; s[1].y += cy * t.y; s[1].z += cz * t.z; s[1].w += cw * t.w;
    s[2].x += ex * cx; s[2].y += ey * cy; s[2].z += ez * cz; s[2].w += ew * cw;
  }
  cq_reduce<3, CQ>(s, red);
  float c1[4], c2[4], c3[4], k2[4];
  {
    const float A_[4] = {s[0].x, s[0].y, s[0].z, s[0].w}, Sxt[4] = {s[1].x, s[1].y, s[1].z, s[1].w},
                Sex[4] = {s[2].x, s[2].y, s[2].z, s[2].w}, sg[4] = {st.sigma.x, st.sigma.y, st.sigma.z, st.sigma.w},
                rr[4] = {st.r.x, st.r.y, st.r.z, st.r.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float B = Sxt[j] * Sex[j] / hw, r2 = rr[j] * rr[j];
      c1[j] = (-A_[j] * r2 + 2.f * r2 * rr[j] * B / sg[j] + r2 * B / (sg[j] * sg[j])) / (hw * sg[j]);
      c2[j] = -r2 * Sex[j] / (hw * sg[j]);
      c3[j] = -r2 * Sxt[j] / (hw * sg[j]);
      k2[j] = r2 * Sex[j] / (hw * sg[j]);
    }
  }
  for (int p = pl; p < HW; p += PL) {
    const size_t i = base + (size_t)p * C;
    const float4 d = ld4(dtout + i), t = ld4(tx + i), xv = ld4(x + i), g = actg_from_x(xv, st, act);
    const float e[4] = {d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w};
    const float xc[4] = {xv.x - st.mean.x, xv.y - st.mean.y, xv.z - st.mean.z, xv.w - st.mean.w};
    const float tt[4] = {t.x, t.y, t.z, t.w}, ebv[4] = {eb.x, eb.y, eb.z, eb.w}, tav[4] = {ta.x, ta.y, ta.z, ta.w},
                rr[4] = {st.r.x, st.r.y, st.r.z, st.r.w};
    float gt[4], gx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      gt[j] = rr[j] * (e[j] - ebv[j]) - k2[j] * xc[j];
      gx[j] = c1[j] * xc[j] + c2[j] * (tt[j] - tav[j]) + c3[j] * (e[j] - ebv[j]);
    }
    float4 o1 = make_float4(gt[0], gt[1], gt[2], gt[3]), o2 = make_float4(gx[0], gx[1], gx[2], gx[3]);
    if (acc_dtx & 1) { const float4 a = ld4(dtx + i); o1.x += a.x; o1.y += a.y; o1.z += a.z; o1.w += a.w; }
    if (acc_dx) { const float4 a = ld4(dx + i); o2.x += a.x; o2.y += a.y; o2.z += a.z; o2.w += a.w; }
    if (!(acc_dtx & 2)) st4(dtx + i, o1);   // (GRAD_PLANE_ONLY)
    if (dtx_bf) st_bf4n(dtx_bf, i, o1);
    st4(dx + i, o2);
  }
}

// the same with e = dtout * act'(out), (x - mean) and the tangent held in registers between the three passes (HW <= NP * 256 /
// CQ): out / dtout / tx / x are read from memory once instead of 3 + 3 + 3 + 2 times; per-thread summation order as
// in_tangent_bwd_v.  dtx_bf: optional bf16 plane of the tangent's gradient (bf16 mode).
template <int NP, int CQ>
__global__ __launch_bounds__(256) void in_tangent_bwd_vr(const float* __restrict__ x, const float* __restrict__ out,
                                                         const float* __restrict__ stats, const float* __restrict__ tx,
                                                         const float* __restrict__ dtout, float* __restrict__ dtx,
                                                         int acc_dtx, float* __restrict__ dx, int acc_dx, int HW, int C,
                                                         int act, unsigned short* __restrict__ dtx_bf) {
  constexpr int PL = 256 / CQ;
  __shared__ float4 red[3 * 4 * CQ];
  const int cq = threadIdx.x % CQ, pl = threadIdx.x / CQ;
  const int n = blockIdx.x, c = (blockIdx.y * CQ + cq) * 4;
  const size_t base = (size_t)n * HW * C + c;
  const Stat4 st = load_stats4(stats, (size_t)n * C + c);
  float4 ev[NP], xc[NP], tv[NP];
  float4 m[2] = {f4_zero(), f4_zero()};
  // (issuing the 3 NP loads before the first use, as in_bwd_vr does, measured SLOWER here at batch 64 -- 51 -> 57 us -- and equal at
  //  batch 4: 290 registers either way, one block per CU; docs/experiments.md)
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int p = pl + k * PL;
    ev[k] = f4_zero();
    xc[k] = f4_zero();
    tv[k] = f4_zero();
    if (p < HW) {
      const size_t i = base + (size_t)p * C;
      const float4 d = ld4(dtout + i), t = ld4(tx + i), xv = ld4(x + i), g = actg_from_x(xv, st, act);
      ev[k] = make_float4(d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w);
      xc[k] = make_float4(xv.x - st.mean.x, xv.y - st.mean.y, xv.z - st.mean.z, xv.w - st.mean.w);
      tv[k] = t;
      m[0].x += ev[k].x; m[0].y += ev[k].y; m[0].z += ev[k].z; m[0].w += ev[k].w;
      m[1].x += t.x; m[1].y += t.y; m[1].z += t.z; m[1].w += t.w;
    }
  }
  cq_reduce<2, CQ>(m, red);
  const float hw = (float)HW;
  const float4 eb = make_float4(m[0].x / hw, m[0].y / hw, m[0].z / hw, m[0].w / hw);
  const float4 ta = make_float4(m[1].x / hw, m[1].y / hw, m[1].z / hw, m[1].w / hw);
  float4 s[3] = {f4_zero(), f4_zero(), f4_zero()};  // sum (e-eb)(t-ta), sum xc*t, sum e*xc
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    if (pl + k * PL < HW) {
      const float4 e = ev[k], t = tv[k], q = xc[k];
      s[0].x += (e.x - eb.x) * (t.x - ta.x); s[0].y += (e.y - eb.y) * (t.y - ta.y);
      s[0].z += (e.z - eb.z) * (t.z - ta.z); s[0].w += (e.w - eb.w) * (t.w - ta.w);
      s[1].x += q.x * t.x; s[1].y += q.y * t.y; s[1].z += q.z * t.z; s[1].w += q.w * t.w;
      s[2].x += e.x * q.x; s[2].y += e.y * q.y; s[2].z += e.z * q.z; s[2].w += e.w * q.w;
    }
  }
  cq_reduce<3, CQ>(s, red);
  float c1[4], c2[4], c3[4], k2[4];
  {
    const float A_[4] = {s[0].x, s[0].y, s[0].z, s[0].w}, Sxt[4] = {s[1].x, s[1].y, s[1].z, s[1].w},
                Sex[4] = {s[2].x, s[2].y, s[2].z, s[2].w}, sg[4] = {st.sigma.x, st.sigma.y, st.sigma.z, st.sigma.w},
                rr[4] = {st.r.x, st.r.y, st.r.z, st.r.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float B = Sxt[j] * Sex[j] / hw, r2 = rr[j] * rr[j];
      c1[j] = (-A_[j] * r2 + 2.f * r2 * rr[j] * B / sg[j] + r2 * B / (sg[j] * sg[j])) / (hw * sg[j]);
      c2[j] = -r2 * Sex[j] / (hw * sg[j]);
      c3[j] = -r2 * Sxt[j] / (hw * sg[j]);
      k2[j] = r2 * Sex[j] / (hw * sg[j]);
    }
  }
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int p = pl + k * PL;
    if (p < HW) {
      const size_t i = base + (size_t)p * C;
      const float e[4] = {ev[k].x, ev[k].y, ev[k].z, ev[k].w}, q[4] = {xc[k].x, xc[k].y, xc[k].z, xc[k].w},
                  tt[4] = {tv[k].x, tv[k].y, tv[k].z, tv[k].w}, ebv[4] = {eb.x, eb.y, eb.z, eb.w},
                  tav[4] = {ta.x, ta.y, ta.z, ta.w}, rr[4] = {st.r.x, st.r.y, st.r.z, st.r.w};
      float gt[4], gx[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        gt[j] = rr[j] * (e[j] - ebv[j]) - k2[j] * q[j];
        gx[j] = c1[j] * q[j] + c2[j] * (tt[j] - tav[j]) + c3[j] * (e[j] - ebv[j]);
      }
      float4 o1 = make_float4(gt[0], gt[1], gt[2], gt[3]), o2 = make_float4(gx[0], gx[1], gx[2], gx[3]);
      if (acc_dtx & 1) { const float4 a = ld4(dtx + i); o1.x += a.x; o1.y += a.y; o1.z += a.z; o1.w += a.w; }
      if (acc_dx) { const float4 a = ld4(dx + i); o2.x += a.x; o2.y += a.y; o2.z += a.z; o2.w += a.w; }
      if (!(acc_dtx & 2)) st4(dtx + i, o1);   // (GRAD_PLANE_ONLY)
      if (dtx_bf) st_bf4n(dtx_bf, i, o1);
      st4(dx + i, o2);
    }
  }
}

void launch_in_tangent_bwd(const float* x, const float* out, const float* stats, const float* tx,
                           const float* dtout, float* dtx, int acc_dtx, float* dx, int acc_dx, int NB, int HW,
                           int C, int act, hipStream_t s, unsigned short* dtx_bf, float* chunk_scratch) {
  EG_CHECK(!dtx_bf || C % 32 == 0, "in_tangent_bwd: a bf16 plane was requested with C %% 32 != 0");
  EG_CHECK(!(acc_dtx & GRAD_PLANE_ONLY) || dtx_bf, "in_tangent_bwd: GRAD_PLANE_ONLY without a plane");
  if (chunk_scratch && in_chunked_shape(HW, C)) {   // large maps: chunked three-stage form
    const int nch = in_chunks(HW);
    float* means = chunk_scratch + (size_t)NB * nch * C * 3;
    float* moments = means + (size_t)NB * C * 2;
    const dim3 cg(nch, (C + 63) / 64, NB), fg((C + 63) / 64, NB);
    hipLaunchKernelGGL(in_chunk_tb_kernel<0>, cg, dim3(256), 0, s, x, tx, dtout, stats, nullptr, chunk_scratch, HW, C, act);
    hipLaunchKernelGGL(in_finalize_n_kernel<2>, fg, dim3(64), 0, s, chunk_scratch, nch, C, means);
    hipLaunchKernelGGL(in_chunk_tb_kernel<1>, cg, dim3(256), 0, s, x, tx, dtout, stats, means, chunk_scratch, HW, C, act);
    hipLaunchKernelGGL(in_finalize_n_kernel<3>, fg, dim3(64), 0, s, chunk_scratch, nch, C, moments);
    hipLaunchKernelGGL(in_tb_apply_chunked_kernel, dim3(in_ew_blocks(HW, C), NB), dim3(256), 0, s, x, tx, dtout, stats, means, moments,
                       dtx, acc_dtx, dx, acc_dx, dtx_bf, HW, C, act);
    return;
  }
  if (C % 32 == 0) {
    const dim3 grid(NB, C / 32);
#define EG_IN_TBWD(NP, CQ, G) hipLaunchKernelGGL((in_tangent_bwd_vr<NP, CQ>), G, dim3(256), 0, s, x, out, stats, tx, dtout, dtx, acc_dtx, dx, acc_dx, HW, C, act, dtx_bf)
    if (HW <= 64) EG_IN_TBWD(2, 8, grid);
    else if (HW <= 128) EG_IN_TBWD(4, 8, grid);
    else if (HW <= 256) EG_IN_TBWD(8, 8, grid);
    else if (HW <= 512) EG_IN_TBWD(16, 8, grid);
    else if (HW <= 1024) EG_IN_TBWD(16, 4, dim3(NB, C / 16));
    else
      hipLaunchKernelGGL(in_tangent_bwd_v<8>, grid, dim3(256), 0, s, x, out, stats, tx, dtout, dtx, acc_dtx, dx, acc_dx, HW, C,
                         act, dtx_bf);
#undef EG_IN_TBWD
    return;
  }
  hipLaunchKernelGGL(in_tangent_bwd_kernel, dim3(NB, (C + 63) / 64), dim3(256), 0, s, x, out, stats, tx, dtout, dtx,
                     acc_dtx, dx, acc_dx, HW, C, act);
}

// ---------------------------------------------------------------------------------------------
// plain activations
// ---------------------------------------------------------------------------------------------
__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, size_t n, int act) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = act_apply(x[i], act);
}
__global__ void act_bwd_kernel(const float* __restrict__ out, const float* __restrict__ dout, float* __restrict__ dx,
                               int accumulate, size_t n, int act) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float g = dout[i] * act_grad_from_out(out[i], act);
    dx[i] = accumulate ? dx[i] + g : g;
  }
}
static inline int ew_blocks(size_t n) {
  size_t b = (n + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}
void launch_act_fwd(const float* x, float* out, size_t n, int act, hipStream_t s) {
  hipLaunchKernelGGL(act_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, x, out, n, act);
}
void launch_act_bwd(const float* out, const float* dout, float* dx, int accumulate, size_t n, int act,
                    hipStream_t s) {
  hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, out, dout, dx, accumulate, n, act);
}

// ---------------------------------------------------------------------------------------------
// BatchNorm (batch statistics, biased variance inside the sqrt) + ReLU over rows [R][C]
// (generator.py:51-52 via normalization.py:19-25; quirk N1)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_relu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ out,
                                                          float* __restrict__ stats, int R, int C, float eps) {
  __shared__ float red[4 * 64];
  const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  const bool cv = c < C;
  float s[1] = {0.f};
  if (cv)
    for (int p = py; p < R; p += 4) s[0] += x[(size_t)p * C + c];
  lane4_reduce<1>(s, red, cx, py);
  const float mean = s[0] / R;
  float q[1] = {0.f};
  if (cv)
    for (int p = py; p < R; p += 4) {
      const float d = x[(size_t)p * C + c] - mean;
      q[0] += d * d;
    }
  lane4_reduce<1>(q, red, cx, py);
  const float var = q[0] / R;
  if (cv) {
    const float rs = 1.f / sqrtf(var + eps), g = gamma[c], b = beta[c];
    for (int p = py; p < R; p += 4) {
      const float y = (x[(size_t)p * C + c] - mean) * rs * g + b;
      out[(size_t)p * C + c] = y > 0.f ? y : 0.f;
    }
    if (py == 0) {
      stats[c] = mean;
      stats[C + c] = var;
    }
  }
}

__global__ __launch_bounds__(256) void bn_relu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ out,
                                                          const float* __restrict__ dout,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ stats, float* __restrict__ dx,
                                                          int acc_dx, float* __restrict__ dgamma, int acc_dg,
                                                          float* __restrict__ dbeta, int acc_db, int R, int C,
                                                          float eps) {
  __shared__ float red[2 * 4 * 64];
  const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  const bool cv = c < C;
  const float mean = cv ? stats[c] : 0.f, var = cv ? stats[C + c] : 1.f;
  const float rs = 1.f / sqrtf(var + eps);
  float s[2] = {0.f, 0.f};
  if (cv)
    for (int p = py; p < R; p += 4) {
      const size_t i = (size_t)p * C + c;
      const float e = out[i] > 0.f ? dout[i] : 0.f;
      s[0] += e;
      s[1] += e * (x[i] - mean) * rs;
    }
  lane4_reduce<2>(s, red, cx, py);
  if (cv) {
    const float g = gamma[c];
    const float me = s[0] / R, mex = s[1] / R;
    if (dx)
      for (int p = py; p < R; p += 4) {
        const size_t i = (size_t)p * C + c;
        const float e = out[i] > 0.f ? dout[i] : 0.f;
        const float xh = (x[i] - mean) * rs;
        const float v = g * rs * (e - me - xh * mex);
        dx[i] = acc_dx ? dx[i] + v : v;
      }
    if (py == 0) {
      if (dgamma) dgamma[c] = acc_dg ? dgamma[c] + s[1] : s[1];
      if (dbeta) dbeta[c] = acc_db ? dbeta[c] + s[0] : s[0];
    }
  }
}

__global__ void bn_moving_kernel(float* mm, float* mv, const float* stats, int C, int R, float decay, int times) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float mean = stats[c], var = stats[C + c] * ((float)R / (float)(R > 1 ? R - 1 : 1));
  float a = mm[c], b = mv[c];
  for (int t = 0; t < times; ++t) {
    a = a * decay + (1.f - decay) * mean;
    b = b * decay + (1.f - decay) * var;
  }
  mm[c] = a;
  mv[c] = b;
}

// float4 variants (C % 32 == 0): 8 channel quads x 32 row lanes per block
__global__ __launch_bounds__(256) void bn_relu_fwd_v(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ out,
                                                     float* __restrict__ stats, int R, int C, float eps,
                                                     unsigned short* __restrict__ out_bf) {
  __shared__ float4 red[4 * 8];
  const int cq = threadIdx.x % 8, pl = threadIdx.x / 8;
  const int c = (blockIdx.x * 8 + cq) * 4;
  float4 s[1] = {f4_zero()};
  for (int p = pl; p < R; p += 32) {
    const float4 v = ld4(x + (size_t)p * C + c);
    s[0].x += v.x; s[0].y += v.y; s[0].z += v.z; s[0].w += v.w;
  }
  cq_reduce<1, 8>(s, red);
  const float4 mean = make_float4(s[0].x / R, s[0].y / R, s[0].z / R, s[0].w / R);
  float4 q[1] = {f4_zero()};
  for (int p = pl; p < R; p += 32) {
    const float4 v = ld4(x + (size_t)p * C + c);
    const float a = v.x - mean.x, b = v.y - mean.y, d = v.z - mean.z, e = v.w - mean.w;
    q[0].x += a * a; q[0].y += b * b; q[0].z += d * d; q[0].w += e * e;
  }
  cq_reduce<1, 8>(q, red);
  const float4 var = make_float4(q[0].x / R, q[0].y / R, q[0].z / R, q[0].w / R);
  const float4 g = ld4(gamma + c), b = ld4(beta + c);
  const float4 sc = make_float4(g.x / sqrtf(var.x + eps), g.y / sqrtf(var.y + eps), g.z / sqrtf(var.z + eps),
                                g.w / sqrtf(var.w + eps));
  for (int p = pl; p < R; p += 32) {
    const float4 v = ld4(x + (size_t)p * C + c);
    const float4 y = make_float4(fmaxf((v.x - mean.x) * sc.x + b.x, 0.f), fmaxf((v.y - mean.y) * sc.y + b.y, 0.f),
                                 fmaxf((v.z - mean.z) * sc.z + b.z, 0.f), fmaxf((v.w - mean.w) * sc.w + b.w, 0.f));
    st4(out + (size_t)p * C + c, y);
    if (out_bf) st_bf4n(out_bf, (size_t)p * C + c, y);   // (bf16 mode: the plane g_dconv_1 reads, in the same pass)
  }
  if (pl == 0) {
    st4(stats + c, mean);
    st4(stats + C + c, var);
  }
}

// The same kernel with the thread's rows held in registers between the three passes (R <= 32 NI): the generator's
// BatchNorm sits on a chain of small launches (R = 16 B rows of 512 channels, 16 blocks), where the loop form pays the
// global-load latency 3 x R / 32 times in sequence.  Same per-thread summation order => the same bits as bn_relu_fwd_v.
template <int NI>
__global__ __launch_bounds__(256) void bn_relu_fwd_vr(const float* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float* __restrict__ out,
                                                      float* __restrict__ stats, int R, int C, float eps,
                                                      unsigned short* __restrict__ out_bf) {
  __shared__ float4 red[4 * 8];
  const int cq = threadIdx.x % 8, pl = threadIdx.x / 8;
  const int c = (blockIdx.x * 8 + cq) * 4;
  float4 v[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int p = pl + 32 * i;
    v[i] = p < R ? ld4(x + (size_t)p * C + c) : f4_zero();
  }
  float4 s[1] = {f4_zero()};
#pragma unroll
  for (int i = 0; i < NI; ++i)
    if (pl + 32 * i < R) { s[0].x += v[i].x; s[0].y += v[i].y; s[0].z += v[i].z; s[0].w += v[i].w; }
  cq_reduce<1, 8>(s, red);
  const float4 mean = make_float4(s[0].x / R, s[0].y / R, s[0].z / R, s[0].w / R);
  float4 q[1] = {f4_zero()};
#pragma unroll
  for (int i = 0; i < NI; ++i)
    if (pl + 32 * i < R) {
      const float a = v[i].x - mean.x, b = v[i].y - mean.y, d = v[i].z - mean.z, e = v[i].w - mean.w;
      q[0].x += a * a; q[0].y += b * b; q[0].z += d * d; q[0].w += e * e;
    }
  cq_reduce<1, 8>(q, red);
  const float4 var = make_float4(q[0].x / R, q[0].y / R, q[0].z / R, q[0].w / R);
  const float4 g = ld4(gamma + c), b = ld4(beta + c);
  const float4 sc = make_float4(g.x / sqrtf(var.x + eps), g.y / sqrtf(var.y + eps), g.z / sqrtf(var.z + eps),
                                g.w / sqrtf(var.w + eps));
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int p = pl + 32 * i;
    if (p < R) {
      const float4 y = make_float4(fmaxf((v[i].x - mean.x) * sc.x + b.x, 0.f), fmaxf((v[i].y - mean.y) * sc.y + b.y, 0.f),
                                   fmaxf((v[i].z - mean.z) * sc.z + b.z, 0.f), fmaxf((v[i].w - mean.w) * sc.w + b.w, 0.f));
      st4(out + (size_t)p * C + c, y);
      if (out_bf) st_bf4n(out_bf, (size_t)p * C + c, y);
    }
  }
  if (pl == 0) {
    st4(stats + c, mean);
    st4(stats + C + c, var);
  }
}

__global__ __launch_bounds__(256) void bn_relu_bwd_v(const float* __restrict__ x, const float* __restrict__ out,
                                                     const float* __restrict__ dout, const float* __restrict__ gamma,
                                                     const float* __restrict__ stats, float* __restrict__ dx, int acc_dx,
                                                     float* __restrict__ dgamma, int acc_dg, float* __restrict__ dbeta,
                                                     int acc_db, int R, int C, float eps) {
  __shared__ float4 red[2 * 4 * 8];
  const int cq = threadIdx.x % 8, pl = threadIdx.x / 8;
  const int c = (blockIdx.x * 8 + cq) * 4;
  const float4 mean = ld4(stats + c), var = ld4(stats + C + c);
  const float4 rs = make_float4(1.f / sqrtf(var.x + eps), 1.f / sqrtf(var.y + eps), 1.f / sqrtf(var.z + eps),
                                1.f / sqrtf(var.w + eps));
  float4 s[2] = {f4_zero(), f4_zero()};
  constexpr int U = 4;   // rows whose loads are in flight together (the sums keep the row order of the plain loop)
  for (int p0 = pl; p0 < R; p0 += 32 * U) {
    float4 o[U], d[U], v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = p0 + 32 * u;
      if (p < R) {
        const size_t i = (size_t)p * C + c;
        o[u] = ld4(out + i); d[u] = ld4(dout + i); v[u] = ld4(x + i);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (p0 + 32 * u >= R) break;
      const float e0 = o[u].x > 0.f ? d[u].x : 0.f, e1 = o[u].y > 0.f ? d[u].y : 0.f, e2 = o[u].z > 0.f ? d[u].z : 0.f,
                  e3 = o[u].w > 0.f ? d[u].w : 0.f;
      s[0].x += e0; s[0].y += e1; s[0].z += e2; s[0].w += e3;
      s[1].x += e0 * (v[u].x - mean.x) * rs.x; s[1].y += e1 * (v[u].y - mean.y) * rs.y;
      s[1].z += e2 * (v[u].z - mean.z) * rs.z; s[1].w += e3 * (v[u].w - mean.w) * rs.w;
    }
  }
  cq_reduce<2, 8>(s, red);
  const float4 g = ld4(gamma + c);
  const float4 me = make_float4(s[0].x / R, s[0].y / R, s[0].z / R, s[0].w / R);
  const float4 mex = make_float4(s[1].x / R, s[1].y / R, s[1].z / R, s[1].w / R);
  if (dx)
    for (int p0 = pl; p0 < R; p0 += 32 * U) {
      float4 o[U], d[U], v[U], t[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int p = p0 + 32 * u;
        t[u] = f4_zero();
        if (p < R) {
          const size_t i = (size_t)p * C + c;
          o[u] = ld4(out + i); d[u] = ld4(dout + i); v[u] = ld4(x + i);
          if (acc_dx) t[u] = ld4(dx + i);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int p = p0 + 32 * u;
        if (p >= R) break;
        const float e0 = o[u].x > 0.f ? d[u].x : 0.f, e1 = o[u].y > 0.f ? d[u].y : 0.f, e2 = o[u].z > 0.f ? d[u].z : 0.f,
                    e3 = o[u].w > 0.f ? d[u].w : 0.f;
        float4 r = make_float4(g.x * rs.x * (e0 - me.x - (v[u].x - mean.x) * rs.x * mex.x),
                               g.y * rs.y * (e1 - me.y - (v[u].y - mean.y) * rs.y * mex.y),
                               g.z * rs.z * (e2 - me.z - (v[u].z - mean.z) * rs.z * mex.z),
                               g.w * rs.w * (e3 - me.w - (v[u].w - mean.w) * rs.w * mex.w));
        if (acc_dx) { r.x += t[u].x; r.y += t[u].y; r.z += t[u].z; r.w += t[u].w; }
        st4(dx + (size_t)p * C + c, r);
      }
    }
  if (pl == 0) {
    if (dgamma) {
      float4 t = s[1];
      if (acc_dg) { const float4 u = ld4(dgamma + c); t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
      st4(dgamma + c, t);
    }
    if (dbeta) {
      float4 t = s[0];
      if (acc_db) { const float4 u = ld4(dbeta + c); t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
      st4(dbeta + c, t);
    }
  }
}

bool launch_bn_relu_fwd(const float* x, const float* gamma, const float* beta, float* out, float* stats, int R,
                        int C, float eps, hipStream_t s, unsigned short* out_bf) {
  if (C % 32 == 0) {
    if (R <= 32 * 4)
      hipLaunchKernelGGL(bn_relu_fwd_vr<4>, dim3(C / 32), dim3(256), 0, s, x, gamma, beta, out, stats, R, C, eps, out_bf);
    else if (R <= 32 * 32)
      hipLaunchKernelGGL(bn_relu_fwd_vr<32>, dim3(C / 32), dim3(256), 0, s, x, gamma, beta, out, stats, R, C, eps, out_bf);
    else
      hipLaunchKernelGGL(bn_relu_fwd_v, dim3(C / 32), dim3(256), 0, s, x, gamma, beta, out, stats, R, C, eps, out_bf);
    return out_bf != nullptr;
  }
  hipLaunchKernelGGL(bn_relu_fwd_kernel, dim3((C + 63) / 64), dim3(256), 0, s, x, gamma, beta, out, stats, R, C, eps);
  return false;
}
void launch_bn_relu_bwd(const float* x, const float* out, const float* dout, const float* gamma,
                        const float* stats, float* dx, int acc_dx, float* dgamma, int acc_dg, float* dbeta,
                        int acc_db, int R, int C, float eps, hipStream_t s) {
  if (C % 32 == 0) {
    hipLaunchKernelGGL(bn_relu_bwd_v, dim3(C / 32), dim3(256), 0, s, x, out, dout, gamma, stats, dx, acc_dx, dgamma,
                       acc_dg, dbeta, acc_db, R, C, eps);
    return;
  }
  hipLaunchKernelGGL(bn_relu_bwd_kernel, dim3((C + 63) / 64), dim3(256), 0, s, x, out, dout, gamma, stats, dx, acc_dx,
                     dgamma, acc_dg, dbeta, acc_db, R, C, eps);
}
void launch_bn_moving_update(float* mm, float* mv, const float* stats, int C, int R, float decay, int times,
                             hipStream_t s) {
  hipLaunchKernelGGL(bn_moving_kernel, dim3((C + 255) / 256), dim3(256), 0, s, mm, mv, stats, C, R, decay, times);
}

// =============================================================================================
// BatchNorm as the `norm` of a conv_block / deconv_block / residual (normalization.py:19-25 with E_norm / G_norm / D_norm =
// 'batch', train.py:54-59; round 6): tf.contrib.layers.batch_norm(center, scale, is_training=True) over ALL rows of one
// evaluation -- y = act(gamma (x - mean) s + beta), s = (var + eps)^-1/2, biased var -- followed by the block's activation.
// The rows [R][C] of a feature map are far more than one block can walk (R = N H W up to 2^18), so every reduction is two-stage:
// row chunks of BN_RPC rows x 64 channels per block write partials, one small kernel combines them in chunk order (fixed
// order, no atomics), an element-wise kernel applies.  Statistics: per-chunk mean and centred second moment, merged with
// the pairwise update (Chan et al.) -- no E[x^2] - E[x]^2 cancellation.
//   J = a (I - 11^T/m) - (b/m) xc xc^T,  a = gamma s,  b = gamma s^3   (symmetric: backward and forward tangent share bn_japply)
// and the reverse of the tangent (the gradient-penalty term, functional.py:26-29 differentiated once more) follows from
//   da/dx_i = -gamma s^3 xc_i/m,  db/dx_i = -3 gamma s^5 xc_i/m.
// =============================================================================================
constexpr int BN_RPC = 256;
// partials of a chunk: NV sums over its rows of f(row, channel); block = 64 channels x 4 row lanes
__global__ __launch_bounds__(256) void bn_chunk_stats_kernel(const float* __restrict__ x, float* __restrict__ part, int R, int C) {
  __shared__ float red[1 * 4 * 64];
  const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + cx;
  const bool cv = c < C;
  const int r0 = blockIdx.x * BN_RPC, r1 = min(R, r0 + BN_RPC);
  float s[1] = {0.f};
  if (cv)
    for (int p = r0 + py; p < r1; p += 4) s[0] += x[(size_t)p * C + c];
  lane4_reduce<1>(s, red, cx, py);
  const float mean = s[0] / (float)(r1 - r0);
  float q[1] = {0.f};
  if (cv)
    for (int p = r0 + py; p < r1; p += 4) {
      const float d = x[(size_t)p * C + c] - mean;
      q[0] += d * d;
    }
  lane4_reduce<1>(q, red, cx, py);
  if (cv && py == 0) {
    part[((size_t)blockIdx.x * C + c) * 2] = mean;
    part[((size_t)blockIdx.x * C + c) * 2 + 1] = q[0];
  }
}
// stats[c] = mean, stats[C + c] = biased variance; the moving statistics advance `times` times (unbiased variance: fused batch norm)
__global__ void bn_finalize_stats_kernel(const float* __restrict__ part, int nchunks, int R, int C, float* __restrict__ stats,
                                         float* mm, float* mv, float decay, int times) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float n = 0.f, mean = 0.f, m2 = 0.f;
  for (int k = 0; k < nchunks; ++k) {
    const float nk = (float)min(BN_RPC, R - k * BN_RPC);
    const float mk = part[((size_t)k * C + c) * 2], qk = part[((size_t)k * C + c) * 2 + 1];
    const float d = mk - mean, nt = n + nk;
    mean += d * (nk / nt);
    m2 += qk + d * d * (n * nk / nt);
    n = nt;
  }
  const float var = m2 / (float)R;
  stats[c] = mean;
  stats[C + c] = var;
  if (times > 0 && mm && mv) {
    const float uv = var * ((float)R / (float)(R > 1 ? R - 1 : 1));
    float a = mm[c], b = mv[c];
    for (int t = 0; t < times; ++t) {
      a = a * decay + (1.f - decay) * mean;
      b = b * decay + (1.f - decay) * uv;
    }
    mm[c] = a;
    mv[c] = b;
  }
}
__global__ void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
                                const float* __restrict__ beta, float* __restrict__ out, size_t n, int C, float eps, int act) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % (size_t)C);
    const float rs = 1.f / sqrtf(stats[C + c] + eps);
    out[i] = act_apply((x[i] - stats[c]) * rs * gamma[c] + beta[c], act);
  }
}
// MODE 0 (backward): e = v act'(out); MODE 1 (forward tangent): e = v.  Chunk partials of sum e and sum e xh, xh = (x - mean) s
template <int MODE>
__global__ __launch_bounds__(256) void bn_chunk_j_kernel(const float* __restrict__ x, const float* __restrict__ out,
                                                         const float* __restrict__ v, const float* __restrict__ stats,
                                                         float* __restrict__ part, int R, int C, float eps, int act) {
  __shared__ float red[2 * 4 * 64];
  const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + cx;
  const bool cv = c < C;
  const int r0 = blockIdx.x * BN_RPC, r1 = min(R, r0 + BN_RPC);
  const float mean = cv ? stats[c] : 0.f, rs = cv ? 1.f / sqrtf(stats[C + c] + eps) : 1.f;
  float s[2] = {0.f, 0.f};
  if (cv)
    for (int p0 = r0 + py; p0 < r1; p0 += 32) {   // eight rows per step, their loads issued together
      float xr[8], er[8], orr[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int p = p0 + 4 * u;
        const size_t i = (size_t)(p < r1 ? p : r0) * C + c;
        xr[u] = x[i];
        er[u] = v[i];
        orr[u] = MODE == 0 ? out[i] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (p0 + 4 * u < r1) {
          float e = er[u];
          if (MODE == 0) e *= act_grad_from_out(orr[u], act);
          s[0] += e;
          s[1] += e * ((xr[u] - mean) * rs);
        }
    }
  lane4_reduce<2>(s, red, cx, py);
  if (cv && py == 0) {
    part[((size_t)blockIdx.x * C + c) * 2] = s[0];
    part[((size_t)blockIdx.x * C + c) * 2 + 1] = s[1];
  }
}
// sums[c] = sum e, sums[C + c] = sum e xh (chunk order); backward: dgamma (+)= sum e xh, dbeta (+)= sum e
__global__ void bn_finalize_j_kernel(const float* __restrict__ part, int nchunks, int C, float* __restrict__ sums,
                                     float* dgamma, int acc_dg, float* dbeta, int acc_db) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float a = 0.f, b = 0.f;
  for (int k = 0; k < nchunks; ++k) {
    a += part[((size_t)k * C + c) * 2];
    b += part[((size_t)k * C + c) * 2 + 1];
  }
  sums[c] = a;
  sums[C + c] = b;
  if (dgamma) dgamma[c] = acc_dg ? dgamma[c] + b : b;
  if (dbeta) dbeta[c] = acc_db ? dbeta[c] + a : a;
}
// res = gamma s (e - mean(e) - xh mean(e xh));  MODE 0: dx (+)= res;  MODE 1: tout = act'(out) res
template <int MODE>
__global__ void bn_japply_kernel(const float* __restrict__ x, const float* __restrict__ out, const float* __restrict__ v,
                                 const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ sums,
                                 float* __restrict__ dst, int accumulate, size_t n, int R, int C, float eps, int act) {
  const float inv_m = 1.f / (float)R;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % (size_t)C);
    const float rs = 1.f / sqrtf(stats[C + c] + eps);
    const float xh = (x[i] - stats[c]) * rs;
    const float o = out[i];
    float e = v[i];
    if (MODE == 0) e *= act_grad_from_out(o, act);
    float res = gamma[c] * rs * (e - sums[c] * inv_m - xh * (sums[C + c] * inv_m));
    if (MODE == 1) res *= act_grad_from_out(o, act);
    dst[i] = accumulate ? dst[i] + res : res;
  }
}
// reverse of the forward tangent, stage A: chunk partials of sum e (e = dtout act'(out)) and sum t
__global__ __launch_bounds__(256) void bn_chunk_tb_means_kernel(const float* __restrict__ out, const float* __restrict__ tx,
                                                                const float* __restrict__ dtout, float* __restrict__ part, int R,
                                                                int C, int act) {
  __shared__ float red[2 * 4 * 64];
  const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + cx;
  const bool cv = c < C;
  const int r0 = blockIdx.x * BN_RPC, r1 = min(R, r0 + BN_RPC);
  float s[2] = {0.f, 0.f};
  if (cv)
    for (int p = r0 + py; p < r1; p += 4) {
      const size_t i = (size_t)p * C + c;
      s[0] += dtout[i] * act_grad_from_out(out[i], act);
      s[1] += tx[i];
    }
  lane4_reduce<2>(s, red, cx, py);
  if (cv && py == 0) {
    part[((size_t)blockIdx.x * C + c) * 2] = s[0];
    part[((size_t)blockIdx.x * C + c) * 2 + 1] = s[1];
  }
}
// stage B: with the means, chunk partials of A = sum (e - eb)(t - tb), S_xt = sum xc t, S_ex = sum e xc
__global__ __launch_bounds__(256) void bn_chunk_tb_moments_kernel(const float* __restrict__ x, const float* __restrict__ out,
                                                                  const float* __restrict__ tx, const float* __restrict__ dtout,
                                                                  const float* __restrict__ stats, const float* __restrict__ means,
                                                                  float* __restrict__ part, int R, int C, int act) {
  __shared__ float red[3 * 4 * 64];
  const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + cx;
  const bool cv = c < C;
  const int r0 = blockIdx.x * BN_RPC, r1 = min(R, r0 + BN_RPC);
  const float mean = cv ? stats[c] : 0.f;
  const float eb = cv ? means[c] / (float)R : 0.f, tb = cv ? means[C + c] / (float)R : 0.f;
  float s[3] = {0.f, 0.f, 0.f};
  if (cv)
    for (int p = r0 + py; p < r1; p += 4) {
      const size_t i = (size_t)p * C + c;
      const float e = dtout[i] * act_grad_from_out(out[i], act), t = tx[i], xc = x[i] - mean;
      s[0] += (e - eb) * (t - tb);
      s[1] += xc * t;
      s[2] += e * xc;
    }
  lane4_reduce<3>(s, red, cx, py);
  if (cv && py == 0) {
    part[((size_t)blockIdx.x * C + c) * 3] = s[0];
    part[((size_t)blockIdx.x * C + c) * 3 + 1] = s[1];
    part[((size_t)blockIdx.x * C + c) * 3 + 2] = s[2];
  }
}
// moments[c], [C + c], [2C + c] = A, S_xt, S_ex;  dgamma (+)= s A - (s^3/m) S_ex S_xt
__global__ void bn_finalize_tb_kernel(const float* __restrict__ part, int nchunks, int R, int C, const float* __restrict__ stats,
                                      float eps, float* __restrict__ moments, float* dgamma, int acc_dg) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float a = 0.f, b = 0.f, d = 0.f;
  for (int k = 0; k < nchunks; ++k) {
    a += part[((size_t)k * C + c) * 3];
    b += part[((size_t)k * C + c) * 3 + 1];
    d += part[((size_t)k * C + c) * 3 + 2];
  }
  moments[c] = a;
  moments[C + c] = b;
  moments[2 * C + c] = d;
  if (dgamma) {
    const float s = 1.f / sqrtf(stats[C + c] + eps);
    const float g = s * a - s * s * s * d * b / (float)R;
    dgamma[c] = acc_dg ? dgamma[c] + g : g;
  }
}
// stage C:  dtx (+)= J e;   dx (+)= c1 xc - (b/m) [ (e - eb) S_xt + (t - tb) S_ex ],  c1 = (-gamma s^3 A + 3 gamma s^5 S_ex S_xt / m) / m
__global__ void bn_tb_apply_kernel(const float* __restrict__ x, const float* __restrict__ out, const float* __restrict__ tx,
                                   const float* __restrict__ dtout, const float* __restrict__ stats, const float* __restrict__ gamma,
                                   const float* __restrict__ means, const float* __restrict__ moments, float* __restrict__ dtx,
                                   int acc_dtx, float* __restrict__ dx, int acc_dx, size_t n, int R, int C, float eps, int act) {
  const float m = (float)R;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % (size_t)C);
    const float s = 1.f / sqrtf(stats[C + c] + eps), g = gamma[c];
    const float s3 = s * s * s;
    const float A = moments[c], S_xt = moments[C + c], S_ex = moments[2 * C + c];
    const float eb = means[c] / m, tb = means[C + c] / m;
    const float e = dtout[i] * act_grad_from_out(out[i], act), t = tx[i], xc = x[i] - stats[c];
    const float gt = g * s * (e - eb) - g * s3 * S_ex / m * xc;
    const float c1 = (-g * s3 * A + 3.f * g * s3 * s * s * S_ex * S_xt / m) / m;
    const float gx = c1 * xc - g * s3 / m * ((e - eb) * S_xt + (t - tb) * S_ex);
    dtx[i] = acc_dtx ? dtx[i] + gt : gt;
    dx[i] = acc_dx ? dx[i] + gx : gx;
  }
}

static inline int bn_chunks(int R) { return (R + BN_RPC - 1) / BN_RPC; }
size_t bn_scratch_floats(int R, int C) { return (size_t)bn_chunks(R) * C * 3 + (size_t)5 * C; }
static inline int bn_ew_blocks(size_t n) { return (int)std::min<size_t>(4096, (n + 255) / 256); }
void launch_bn_act_fwd(const float* x, const float* gamma, const float* beta, float* out, float* stats, float* scratch, float* mm,
                       float* mv, int times, int R, int C, float eps, int act, hipStream_t s) {
  const int nch = bn_chunks(R);
  hipLaunchKernelGGL(bn_chunk_stats_kernel, dim3(nch, (C + 63) / 64), dim3(256), 0, s, x, scratch, R, C);
  hipLaunchKernelGGL(bn_finalize_stats_kernel, dim3((C + 63) / 64), dim3(64), 0, s, scratch, nch, R, C, stats, mm, mv, 0.9f, times);
  const size_t n = (size_t)R * C;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(bn_ew_blocks(n)), dim3(256), 0, s, x, stats, gamma, beta, out, n, C, eps, act);
}
void launch_bn_act_bwd(const float* x, const float* out, const float* dout, const float* gamma, const float* stats, float* scratch,
                       float* dx, int acc_dx, float* dgamma, int acc_dg, float* dbeta, int acc_db, int R, int C, float eps, int act,
                       hipStream_t s) {
  const int nch = bn_chunks(R);
  float* sums = scratch + (size_t)nch * C * 3;
  hipLaunchKernelGGL(bn_chunk_j_kernel<0>, dim3(nch, (C + 63) / 64), dim3(256), 0, s, x, out, dout, stats, scratch, R, C, eps, act);
  hipLaunchKernelGGL(bn_finalize_j_kernel, dim3((C + 63) / 64), dim3(64), 0, s, scratch, nch, C, sums, dgamma, acc_dg, dbeta, acc_db);
  const size_t n = (size_t)R * C;
  if (dx)
    hipLaunchKernelGGL(bn_japply_kernel<0>, dim3(bn_ew_blocks(n)), dim3(256), 0, s, x, out, dout, stats, gamma, sums, dx, acc_dx, n, R,
                       C, eps, act);
}
void launch_bn_act_tangent_fwd(const float* x, const float* out, const float* gamma, const float* stats, const float* tx, float* tout,
                               float* scratch, int R, int C, float eps, int act, hipStream_t s) {
  const int nch = bn_chunks(R);
  float* sums = scratch + (size_t)nch * C * 3;
  hipLaunchKernelGGL(bn_chunk_j_kernel<1>, dim3(nch, (C + 63) / 64), dim3(256), 0, s, x, out, tx, stats, scratch, R, C, eps, act);
  hipLaunchKernelGGL(bn_finalize_j_kernel, dim3((C + 63) / 64), dim3(64), 0, s, scratch, nch, C, sums, nullptr, 0, nullptr, 0);
  const size_t n = (size_t)R * C;
  hipLaunchKernelGGL(bn_japply_kernel<1>, dim3(bn_ew_blocks(n)), dim3(256), 0, s, x, out, tx, stats, gamma, sums, tout, 0, n, R, C, eps,
                     act);
}
void launch_bn_act_tangent_bwd(const float* x, const float* out, const float* gamma, const float* stats, const float* tx,
                               const float* dtout, float* scratch, float* dtx, int acc_dtx, float* dx, int acc_dx, float* dgamma,
                               int acc_dg, int R, int C, float eps, int act, hipStream_t s) {
  const int nch = bn_chunks(R);
  float* means = scratch + (size_t)nch * C * 3;
  float* moments = means + 2 * C;
  hipLaunchKernelGGL(bn_chunk_tb_means_kernel, dim3(nch, (C + 63) / 64), dim3(256), 0, s, out, tx, dtout, scratch, R, C, act);
  hipLaunchKernelGGL(bn_finalize_j_kernel, dim3((C + 63) / 64), dim3(64), 0, s, scratch, nch, C, means, nullptr, 0, nullptr, 0);
  hipLaunchKernelGGL(bn_chunk_tb_moments_kernel, dim3(nch, (C + 63) / 64), dim3(256), 0, s, x, out, tx, dtout, stats, means, scratch, R,
                     C, act);
  hipLaunchKernelGGL(bn_finalize_tb_kernel, dim3((C + 63) / 64), dim3(64), 0, s, scratch, nch, R, C, stats, eps, moments, dgamma, acc_dg);
  const size_t n = (size_t)R * C;
  hipLaunchKernelGGL(bn_tb_apply_kernel, dim3(bn_ew_blocks(n)), dim3(256), 0, s, x, out, tx, dtout, stats, gamma, means, moments, dtx,
                     acc_dtx, dx, acc_dx, n, R, C, eps, act);
}

}  // namespace eg
