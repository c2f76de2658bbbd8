// bf16-mode filter gradient fed from the bf16 PLANES of both operands -- no VALU staging.
//
//   wgrad_bf16p_kernel : dW[tap][c][n] = sum_m Xbf[pix_in(m, tap)][c] * dYbf[m][n],   m over NB x OH x OW output pixels,
//                        pix_in = (oy S + kh - pad, ox S + kw - pad) (mirrored when reflect): the filter gradients of
//                        conv2d / deconv2d (edgegan/nn/modules/conv.py:13-36, 39-58) for 3x3 stride 1, 4x4 and 5x5 stride 2.
//
// The contraction runs over pixels, the SLOW index of both NHWC operands.  Round 3's bf16 mode gathered fp32 pixels of both
// operands per TAP, rounded them in registers and transposed them in the LDS store (wgrad_x3pp_kernel<1>: 380-550 TFLOP/s).
// Here both operands arrive from their bf16 planes by `buffer_load ... lds` into PIXEL-major LDS images (a halo row of the
// gathered operand: 64 channels = 128 bytes per pixel; the dense operand: 128 channels = 256 bytes per pixel), and the MFMA
// operands -- 8 consecutive k = pixels of one channel per lane -- come out of `ds_read_b64_tr_b16`, gfx950's transposing
// read (semantics: k_gemm_x3.hip, profiles/r03c_ds_read_b64_tr_b16_probe.txt).  The 16-channel blocks of a pixel row are
// XOR-swizzled on the GLOBAL side of the DMA so that a half wave's transpose read covers all 64 banks once.
//
// Unit of work (as the row-halo kernel of round 3): (filter row ky, column phase par < S, 64-channel block of the gathered
// operand) x 128 output channels; the unit's taps kx = par + S d (d < ND) read the SAME staged halo row shifted by d pixels,
// so a K step (32 output pixels = R segments of W = min(OW, 32) pixels of R image rows) stages R x (W + ND - 1) halo pixels
// once and feeds ND taps.  256 threads = 4 waves, wave w owns output channels 32 w .. 32 w + 31 and all 64 rows of the ND
// taps (2 ND accumulators of 32 x 32).  Double-buffered LDS (32 KB): three blocks per CU.  Pixels are split over blockIdx.y
// into slabs, summed by the deterministic second-stage reduction as before.
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace eg {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {
__device__ __forceinline__ int reflect_w(int i, int n) {
  i = i < 0 ? -i : i;
  return i >= n ? 2 * (n - 1) - i : i;
}
template <int N, class F>
__device__ __forceinline__ void static_for_w(F&& f) {
  if constexpr (N > 0) {
    static_for_w<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}
}  // namespace

// CONTIG (4x4 stride 2 on maps at least 16 wide): the unit is (filter row ky, 64-channel block) with ALL KW taps -- the halo
// row holds the S (W - 1) + KW CONSECUTIVE input pixels of the segment and tap kx of output pixel j reads slot S j + kx
// (a transpose read's four pixel addresses are free), so the dense operand's tile is staged once per filter row instead
// of once per (row, column phase): 7.8 instead of 13 KB of LDS-DMA traffic per MFLOP.
template <int S, int KW, int ND, bool CONTIG>
__global__ __launch_bounds__(256, (ND > 3 ? 2 : 3)) void wgrad_bf16p_kernel(const WgradArgs a, float* __restrict__ out, int dense_out) {
  constexpr int XSLOTS = CONTIG ? 80 : 64;
  constexpr int NXJ = (XSLOTS / 8 + 3) / 4;   // X pieces per wave
  constexpr int XB = XSLOTS * 64 * 2;   // bytes of one X halo buffer: pixel slots of 128 bytes
  constexpr int YB = 32 * 128 * 2;    // bytes of one dY buffer: 32 pixels of 256 bytes
  __shared__ __attribute__((aligned(16))) unsigned short Xs[2][XSLOTS * 64];
  __shared__ __attribute__((aligned(16))) unsigned short Ys[2][32 * 128];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int C = a.C, N = a.N, IH = a.IH, IW = a.IW, OW = a.OW, OH = a.OH;
  const int lOW = 31 - __builtin_clz(OW), lOH = 31 - __builtin_clz(OH);
  const int W = OW < 32 ? OW : 32, R = 32 / W, lW = 31 - __builtin_clz(W);
  const int HWc = CONTIG ? S * (W - 1) + KW : W + ND - 1, HWP = (HWc + 7) & ~7;   // halo pixels of a row segment, padded to whole 8-slot DMA pieces
  const int NXP = R * HWP / 8;                          // X pieces of a K step (8 slots x 128 bytes each)
  // unit and output-channel tile of this block
  const int ntn = (N + 127) / 128, cblocks = C / 64;
  const int u = blockIdx.x / ntn, n0 = (blockIdx.x - u * ntn) * 128;
  const int cbk = u % cblocks, urest = u / cblocks;
  const int par = (S == 1 || CONTIG) ? 0 : urest % S, ky = (S == 1 || CONTIG) ? urest : urest / S;
  const int c0 = cbk * 64;
  // pixel range of this slab (whole K steps)
  const int M = a.NB * OH * OW;
  int chunk = (M + a.nsplit - 1) / a.nsplit;
  chunk = (chunk + 31) & ~31;
  const int p_lo = blockIdx.y * chunk, p_hi = min(M, p_lo + chunk);
  const int nsteps = p_hi > p_lo ? (p_hi - p_lo + 31) / 32 : 0;

  const __amdgpu_buffer_rsrc_t xres =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.x_bf), 0, a.xbf_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yres =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.dy_bf), 0, a.dybf_bytes, 0x00020000);

  // ---- DMA of the gathered operand: piece p = wave + 4 j covers halo slots 8 p .. 8 p + 7; lane (slot s, granule q) =
  //      (lane >> 3, lane & 7) brings 8 channels of the slot's source pixel; 16-channel block b of slot s lives at block
  //      b ^ (2 ((s >> 1) & 1)) of the slot's 128 bytes, i.e. granule q ^ (4 ((s >> 1) & 1)) ---------------------------------
  int xr[NXJ], xcol[NXJ];     // the lane's halo row segment / source column per piece; xcol < 0: no pixel
  unsigned xch[NXJ];
#pragma unroll
  for (int j = 0; j < NXJ; ++j) {
    const int s = (wave + 4 * j) * 8 + (lane >> 3);
    const int r = s / HWP, h = s - r * HWP;
    xr[j] = r;
    int ix = -1;
    if (wave + 4 * j < NXP && h < HWc) {
      // tap d of output column ox reads input column ox S + par + S d - pad = (ox + d) S + par - pad: halo pixel h = ox + d
      // (the segment starts at column ox0 = 0 of the image when W == OW, else at a multiple of 32)
      ix = h;   // completed per step with the segment's first column
    }
    xcol[j] = ix;
    xch[j] = (unsigned)((c0 + (((lane & 7) ^ (4 * ((s >> 1) & 1))) * 8)) * 2);
  }
  auto request_x = [&](int t, int buf) __attribute__((always_inline)) {
    const int m = p_lo + t * 32;
#pragma unroll
    for (int j = 0; j < NXJ; ++j)
      if (wave + 4 * j < NXP) {
        const int grow = (m >> lOW) + xr[j];                 // global output row of the lane's segment
        const int nb = grow >> lOH, oy = grow & (OH - 1);
        const int ox0 = m & (OW - 1);                        // first output column of the K step's segments
        int iy = oy * S + ky - a.pad, ix = CONTIG ? ox0 * S + xcol[j] - a.pad : (ox0 + xcol[j]) * S + par - a.pad;
        if (a.reflect) {
          iy = reflect_w(iy, IH);
          ix = reflect_w(ix, IW);
        }
        const bool ok = xcol[j] >= 0 && (unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW && (m + xr[j] * W) < p_hi;
        const unsigned voff = ok ? (unsigned)(((nb * IH + iy) * IW + ix) * C * 2) + xch[j] : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (__attribute__((address_space(3))) void*)&Xs[buf][(wave + 4 * j) * 8 * 64], 16, voff, 0,
                                                 0, 0);
      }
  };
  // ---- DMA of the dense operand: piece p = wave + 4 j covers pixels 4 p .. 4 p + 3 of the K step; lane (pixel, granule q) =
  //      (lane >> 4, lane & 15); 16-channel block b of pixel j lives at block b ^ (2 (j & 3)), i.e. granule q ^ (4 (j & 3)) ----
  unsigned ych[2];
  int ypix[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int px = (wave + 4 * j) * 4 + (lane >> 4);
    const int q = (lane & 15) ^ (4 * (px & 3));
    ypix[j] = px;
    ych[j] = n0 + q * 8 < N ? (unsigned)((n0 + q * 8) * 2) : 0x80000000u;
  }
  auto request_y = [&](int t, int buf) __attribute__((always_inline)) {
    const int m = p_lo + t * 32;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int pm = m + ypix[j];
      const bool in = pm < p_hi;
      if (a.dy_up) {   // WgradArgs::dy_up: pixel (nb, oy, ox) reads the half-resolution plane's (nb, oy / 2, ox / 2)
        const int row = pm >> lOW, nb = row >> lOH, oy = row & (OH - 1), ox = pm & (OW - 1);
        pm = (((nb << (lOH - 1)) + (oy >> 1)) << (lOW - 1)) + (ox >> 1);
      }
      const unsigned voff = in ? (unsigned)(pm * N * 2) + ych[j] : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(yres, (__attribute__((address_space(3))) void*)&Ys[buf][(wave + 4 * j) * 4 * 128], 16, voff, 0, 0,
                                               0);
    }
  };

  // ---- fragment addresses (ds_read_b64_tr_b16: in a 16-lane group, lane 4 r + q supplies the address of 4 consecutive
  //      channels of pixel row r and receives, for channel (its index in the group), the values of the four rows) ------------
  const int tg = lane >> 4, ti = lane & 15, tr = ti >> 2, tq = ti & 3;
  const unsigned xs_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&Xs[0][0];
  const unsigned ys_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&Ys[0][0];
  unsigned xa[ND][2][2];    // [tap d][K half kk][first / second 4 pixels]: channel block 0 of the unit (block 1: ^ 64)
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int j = 16 * kk + 8 * (tg >> 1) + 4 * h + tr;          // pixel of the K step
        const int s = (j >> lW) * HWP + (CONTIG ? S : 1) * (j & (W - 1)) + d;   // its halo slot for tap d
        xa[d][kk][h] = xs_base + (unsigned)(s * 128 + ((((tg & 1)) ^ (2 * ((s >> 1) & 1))) * 32) + tq * 8);
      }
  // dense operand: pixel j = 16 kk + 8 (tg >> 1) + 4 h + tr -> j & 3 == tr: the swizzle is a per-lane constant
  const unsigned ya = ys_base + (unsigned)((8 * (tg >> 1) + tr) * 256 + (((2 * wave + (tg & 1)) ^ (2 * tr)) * 32) + tq * 8);

  f32x16 acc[ND][2];
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[d][cb][r] = 0.f;

#define EG_TRR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
  struct XFrag { u32x2 v[ND][2][2]; };   // [tap][channel block][first / second 4 pixels] of one K half
  struct YFrag { u32x2 v[2]; };
  auto op8 = [](const u32x2& lo, const u32x2& hi) __attribute__((always_inline)) {
    const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, v);
  };
  // one K half (16 pixels) of step-buffer BUF: fragment reads, then 2 ND MFMAs; every fragment register is an in/out operand
  // of the wait that retires its read
  auto half_step = [&](auto BUFc, auto KKc) __attribute__((always_inline)) {
    constexpr int BUF = decltype(BUFc)::value, KK = decltype(KKc)::value;
    XFrag xf;
    YFrag yf;
    const unsigned yq = ya;
    EG_TRR(yf.v[0], yq, BUF * YB + (16 * KK) * 256);
    EG_TRR(yf.v[1], yq, BUF * YB + (16 * KK + 4) * 256);
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const unsigned p0 = xa[d][KK][h], p1 = p0 ^ 64u;
        EG_TRR(xf.v[d][0][h], p0, BUF * XB);
        EG_TRR(xf.v[d][1][h], p1, BUF * XB);
      }
    if constexpr (ND == 4)
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(yf.v[0]), "+v"(yf.v[1]), "+v"(xf.v[0][0][0]), "+v"(xf.v[0][0][1]), "+v"(xf.v[0][1][0]), "+v"(xf.v[0][1][1]),
                     "+v"(xf.v[1][0][0]), "+v"(xf.v[1][0][1]), "+v"(xf.v[1][1][0]), "+v"(xf.v[1][1][1]), "+v"(xf.v[2][0][0]),
                     "+v"(xf.v[2][0][1]), "+v"(xf.v[2][1][0]), "+v"(xf.v[2][1][1]), "+v"(xf.v[ND - 1][0][0]), "+v"(xf.v[ND - 1][0][1]),
                     "+v"(xf.v[ND - 1][1][0]), "+v"(xf.v[ND - 1][1][1])::"memory");
    else if constexpr (ND == 3)
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(yf.v[0]), "+v"(yf.v[1]), "+v"(xf.v[0][0][0]), "+v"(xf.v[0][0][1]), "+v"(xf.v[0][1][0]), "+v"(xf.v[0][1][1]),
                     "+v"(xf.v[1][0][0]), "+v"(xf.v[1][0][1]), "+v"(xf.v[1][1][0]), "+v"(xf.v[1][1][1]), "+v"(xf.v[2][0][0]),
                     "+v"(xf.v[2][0][1]), "+v"(xf.v[2][1][0]), "+v"(xf.v[2][1][1])::"memory");
    else
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(yf.v[0]), "+v"(yf.v[1]), "+v"(xf.v[0][0][0]), "+v"(xf.v[0][0][1]), "+v"(xf.v[0][1][0]), "+v"(xf.v[0][1][1]),
                     "+v"(xf.v[1][0][0]), "+v"(xf.v[1][0][1]), "+v"(xf.v[1][1][0]), "+v"(xf.v[1][1][1])::"memory");
    const bf16x8 yb = op8(yf.v[0], yf.v[1]);
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
        acc[d][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yb, op8(xf.v[d][cb][0], xf.v[d][cb][1]), acc[d][cb], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  constexpr std::integral_constant<int, 0> I0{};
  constexpr std::integral_constant<int, 1> I1{};
  if (nsteps > 0) {
    request_x(0, 0);
    request_y(0, 0);
  }
  auto step = [&](auto BUFc, int t) __attribute__((always_inline)) {
    constexpr int BUF = decltype(BUFc)::value;
    // this wave's pieces of tile t have landed; every wave's are visible, and every wave is done reading tile t - 1
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (t + 1 < nsteps) {
      request_x(t + 1, BUF ^ 1);
      request_y(t + 1, BUF ^ 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    half_step(BUFc, I0);
    half_step(BUFc, I1);
  };
  int t = 0;
  for (; t + 1 < nsteps; t += 2) {
    step(I0, t);
    step(I1, t + 1);
  }
  if (t < nsteps) step(I0, t);
#undef EG_TRR

  // D^T layout: lane l holds row (channel of the gathered operand) l & 31 and, in registers 4q .. 4q+3, output channels
  // 8q + 4 (l >> 5) .. +3 of the wave's 32: one 16-byte store per register quad
  const int l31 = lane & 31, lh = lane >> 5;
  const int T = a.KH * a.KW;
#pragma unroll
  for (int d = 0; d < ND; ++d) {
    const int kx = CONTIG ? d : par + S * d;
    if (kx >= KW) continue;   // (5x5 stride 2, odd phase: the third tap does not exist; computed and dropped)
    const int tap = ky * a.KW + kx;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int c = c0 + cb * 32 + l31;
      float* row = dense_out ? out + (((size_t)blockIdx.y * T + tap) * C + c) * N : out + (size_t)tap * a.w_ts + (size_t)c * a.ldw;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + 32 * wave + q * 8 + lh * 4;
        if (n < N) {
          float4 v = make_float4(acc[d][cb][4 * q], acc[d][cb][4 * q + 1], acc[d][cb][4 * q + 2], acc[d][cb][4 * q + 3]);
          if (a.dy_up) { v.x *= 0.25f; v.y *= 0.25f; v.z *= 0.25f; v.w *= 0.25f; }   // (exact: the pool's 1/4 applied to the sum)
          if (!dense_out && a.accumulate) {
            const float4 o = *reinterpret_cast<const float4*>(row + n);
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
          }
          *reinterpret_cast<float4*>(row + n) = v;
        }
      }
    }
  }
}

static bool wgrad_bf16p_contig(const WgradArgs& a) {   // 4x4 stride 2, output maps at least 16 wide: all four taps of a filter row per unit
  static const bool off = dbg_env("EG_WGRAD_BF16P_NO_CONTIG") != nullptr;
  return !off && a.S == 2 && a.KW == 4 && a.OW >= 16;
}

// units of the plane-fed filter-gradient kernel (0: it does not take the shape)
int wgrad_bf16p_units(const WgradArgs& a) {
  static const bool off = dbg_env("EG_NO_WGRAD_BF16P") != nullptr;
  auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
  const bool shape = (a.S == 1 && a.KW == 3) || (a.S == 2 && (a.KW == 4 || a.KW == 5));
  if (off || !a.x_bf || !a.dy_bf || a.limbs != 1 || !shape || a.KH != a.KW || a.C % 64 != 0 || a.N % 32 != 0 || a.N < 64) return 0;
  if (!pow2(a.OW) || !pow2(a.OH) || a.OW < 8 || a.OH * a.OW < 32 || !a.xbf_bytes || !a.dybf_bytes) return 0;
  if ((((uintptr_t)a.x_bf | (uintptr_t)a.dy_bf) & 15) != 0 || a.ldw % 4 != 0 || a.w_ts % 4 != 0) return 0;
  if (wgrad_bf16p_contig(a)) return a.KH * (a.C / 64);
  return a.KH * std::min(a.S, a.KW) * (a.C / 64);
}

const char* launch_wgrad_bf16p(const WgradArgs& a, float* out, int dense, hipStream_t s) {
  const int units = wgrad_bf16p_units(a);
  dim3 grid(units * ((a.N + 127) / 128), a.nsplit);
  if (a.S == 1) {
    hipLaunchKernelGGL((wgrad_bf16p_kernel<1, 3, 3, false>), grid, dim3(256), 0, s, a, out, dense);
    return "wgrad_bf16p<k3s1;64,128>";
  }
  if (a.KW == 4 && wgrad_bf16p_contig(a)) {
    hipLaunchKernelGGL((wgrad_bf16p_kernel<2, 4, 4, true>), grid, dim3(256), 0, s, a, out, dense);
    return "wgrad_bf16p<k4s2 rows;64,128>";
  }
  if (a.KW == 4) {
    hipLaunchKernelGGL((wgrad_bf16p_kernel<2, 4, 2, false>), grid, dim3(256), 0, s, a, out, dense);
    return "wgrad_bf16p<k4s2;64,128>";
  }
  hipLaunchKernelGGL((wgrad_bf16p_kernel<2, 5, 3, false>), grid, dim3(256), 0, s, a, out, dense);
  return "wgrad_bf16p<k5s2;64,128>";
}

}  // namespace eg
