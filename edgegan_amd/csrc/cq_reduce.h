// Channel-quad x pixel-lane block layout helpers shared by the per-(n,c) reduction kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace eg {

__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// quad at float index i (channel c) of a tensor that may still be K-split slabs (kernels.h: SlabSrc): bias + slabs in
// conv_split_reduce's order, then the base
__device__ __forceinline__ float4 slab_ld4(const SlabSrc& s, size_t i, int c) {
  if (s.n <= 1) return ld4(s.p + i);
  float4 v = s.bias ? ld4(s.bias + c) : f4_zero();
  for (int k = 0; k < s.n; ++k) {
    const float4 q = ld4(s.p + (size_t)k * s.stride + i);
    v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
  }
  if (s.base) {
    const float4 q = ld4(s.base + i);
    v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
  }
  return v;
}
// the same for flat element-wise kernels: quad index i4 of an [..][C] tensor
__device__ __forceinline__ float4 slab_ld4_flat(const SlabSrc& s, size_t i4) {
  if (s.n <= 1) return reinterpret_cast<const float4*>(s.p)[i4];
  return slab_ld4(s, i4 * 4, (int)(i4 % (size_t)(s.C >> 2)) * 4);
}
#define F4_OP(dst, expr_x, expr_y, expr_z, expr_w) \
  do { (dst).x = (expr_x); (dst).y = (expr_y); (dst).z = (expr_z); (dst).w = (expr_w); } while (0)

template <int NV, int CQ>
__device__ __forceinline__ void cq_reduce(float4 (&v)[NV], float4* red /*[NV][4][CQ]*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
#pragma unroll
    for (int o = CQ; o < 64; o <<= 1) {
      v[k].x += __shfl_xor(v[k].x, o, 64);
      v[k].y += __shfl_xor(v[k].y, o, 64);
      v[k].z += __shfl_xor(v[k].z, o, 64);
      v[k].w += __shfl_xor(v[k].w, o, 64);
    }
  }
  __syncthreads();
  if (lane < CQ) {
#pragma unroll
    for (int k = 0; k < NV; ++k) red[(k * 4 + wave) * CQ + lane] = v[k];
  }
  __syncthreads();
  const int cq = threadIdx.x % CQ;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const float4 a = red[(k * 4 + 0) * CQ + cq], b = red[(k * 4 + 1) * CQ + cq], c = red[(k * 4 + 2) * CQ + cq],
                 d = red[(k * 4 + 3) * CQ + cq];
    F4_OP(v[k], (a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z),
          (a.w + b.w) + (c.w + d.w));
  }
}


// same layout, min / max instead of sum
template <int CQ>
__device__ __forceinline__ void cq_reduce_minmax(float4& mn, float4& mx, float4* red /*[2][4][CQ]*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = CQ; o < 64; o <<= 1) {
    mn.x = fminf(mn.x, __shfl_xor(mn.x, o, 64)); mn.y = fminf(mn.y, __shfl_xor(mn.y, o, 64));
    mn.z = fminf(mn.z, __shfl_xor(mn.z, o, 64)); mn.w = fminf(mn.w, __shfl_xor(mn.w, o, 64));
    mx.x = fmaxf(mx.x, __shfl_xor(mx.x, o, 64)); mx.y = fmaxf(mx.y, __shfl_xor(mx.y, o, 64));
    mx.z = fmaxf(mx.z, __shfl_xor(mx.z, o, 64)); mx.w = fmaxf(mx.w, __shfl_xor(mx.w, o, 64));
  }
  __syncthreads();
  if (lane < CQ) {
    red[(0 * 4 + wave) * CQ + lane] = mn;
    red[(1 * 4 + wave) * CQ + lane] = mx;
  }
  __syncthreads();
  const int cq = threadIdx.x % CQ;
  float4 a = red[cq], b = red[CQ + cq], c = red[2 * CQ + cq], d = red[3 * CQ + cq];
  mn = make_float4(fminf(fminf(a.x, b.x), fminf(c.x, d.x)), fminf(fminf(a.y, b.y), fminf(c.y, d.y)),
                   fminf(fminf(a.z, b.z), fminf(c.z, d.z)), fminf(fminf(a.w, b.w), fminf(c.w, d.w)));
  a = red[4 * CQ + cq]; b = red[5 * CQ + cq]; c = red[6 * CQ + cq]; d = red[7 * CQ + cq];
  mx = make_float4(fmaxf(fmaxf(a.x, b.x), fmaxf(c.x, d.x)), fmaxf(fmaxf(a.y, b.y), fmaxf(c.y, d.y)),
                   fmaxf(fmaxf(a.z, b.z), fmaxf(c.z, d.z)), fmaxf(fmaxf(a.w, b.w), fmaxf(c.w, d.w)));
}

// the same two reductions for blocks of NW waves (red: [NV][NW][CQ] / [2][NW][CQ]); NW = 4 is the form above, other block
// sizes sum the waves' partials in wave order (fixed)
template <int NV, int CQ, int NW>
__device__ __forceinline__ void cq_reduce_w(float4 (&v)[NV], float4* red) {
  if constexpr (NW == 4) {
    cq_reduce<NV, CQ>(v, red);
  } else {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
#pragma unroll
      for (int o = CQ; o < 64; o <<= 1) {
        v[k].x += __shfl_xor(v[k].x, o, 64);
        v[k].y += __shfl_xor(v[k].y, o, 64);
        v[k].z += __shfl_xor(v[k].z, o, 64);
        v[k].w += __shfl_xor(v[k].w, o, 64);
      }
    }
    __syncthreads();
    if (lane < CQ) {
#pragma unroll
      for (int k = 0; k < NV; ++k) red[(k * NW + wave) * CQ + lane] = v[k];
    }
    __syncthreads();
    const int cq = threadIdx.x % CQ;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      float4 t = red[(k * NW) * CQ + cq];
#pragma unroll
      for (int w = 1; w < NW; ++w) {
        const float4 a = red[(k * NW + w) * CQ + cq];
        t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w;
      }
      v[k] = t;
    }
  }
}
template <int CQ, int NW>
__device__ __forceinline__ void cq_reduce_minmax_w(float4& mn, float4& mx, float4* red) {
  if constexpr (NW == 4) {
    cq_reduce_minmax<CQ>(mn, mx, red);
  } else {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = CQ; o < 64; o <<= 1) {
      mn.x = fminf(mn.x, __shfl_xor(mn.x, o, 64)); mn.y = fminf(mn.y, __shfl_xor(mn.y, o, 64));
      mn.z = fminf(mn.z, __shfl_xor(mn.z, o, 64)); mn.w = fminf(mn.w, __shfl_xor(mn.w, o, 64));
      mx.x = fmaxf(mx.x, __shfl_xor(mx.x, o, 64)); mx.y = fmaxf(mx.y, __shfl_xor(mx.y, o, 64));
      mx.z = fmaxf(mx.z, __shfl_xor(mx.z, o, 64)); mx.w = fmaxf(mx.w, __shfl_xor(mx.w, o, 64));
    }
    __syncthreads();
    if (lane < CQ) {
      red[wave * CQ + lane] = mn;
      red[(NW + wave) * CQ + lane] = mx;
    }
    __syncthreads();
    const int cq = threadIdx.x % CQ;
    mn = red[cq];
    mx = red[NW * CQ + cq];
#pragma unroll
    for (int w = 1; w < NW; ++w) {
      const float4 a = red[w * CQ + cq], b = red[(NW + w) * CQ + cq];
      mn = make_float4(fminf(mn.x, a.x), fminf(mn.y, a.y), fminf(mn.z, a.z), fminf(mn.w, a.w));
      mx = make_float4(fmaxf(mx.x, b.x), fmaxf(mx.y, b.y), fmaxf(mx.z, b.z), fmaxf(mx.w, b.w));
    }
  }
}

}  // namespace eg
