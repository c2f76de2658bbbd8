// Thin filter gradients (3 or 8 channels in the gathered operand) with (tap, channel) PACKED into the MFMA's row
// dimension -- VERDICT r3 item 4, the filter-gradient half (k_thin_pk.hip is the forward half).
//
//   wgrad_pk_kernel<CP, KH, KW, S, CT, LIMBS> :
//       dW[ky][kx][c][n] = sum_{b,oy,ox} X[b][oy S + ky - pad][ox S + kx - pad][c] dY[b][oy][ox][n]
//       CP = 4: C == 3 -- d_conv_0 of the three discriminators and the encoder stem (4x4 stride 2: edgegan/models/discriminator.py:61-63,
//               encoder.py:61-63), the image-pyramid convolutions of the MRU units (3x3: edgegan/nn/modules/conv.py:199-213), the
//               classifier's 7x7 stem (edgegan/models/classifier.py:39), the generators' 3-channel output layer (5x5 stride 2
//               transposed convolution: the "gathered" operand is the output gradient, conv.py:39-58)
//       CP = 8: C == 8 -- Conv_1 / Conv_3 / the update gate of the classifier's first MRU unit (conv.py:185-224)
//
// wgrad_flatm_kernel (k_gemm.hip) gathers this operand element by element (an index computation and a predicate per fp32
// value, 32-pixel steps behind two block barriers, exact-fp32 16x16x4 MFMAs): 24-44 TFLOP/s, 0.1-1.9 TB/s, on neither
// roofline.  These layers cost what their dense operand dY costs to READ (N = 64 ... 512 channels against 3 or 8): the kernel
// below is built to stream dY once at HBM rate and to do everything else beside that stream.
//
// * The contraction runs over pixels: a K step is 32 output pixels (R rows of W = min(OW, 32) columns).  Every WAVE works
//   alone -- its own K steps of its own 16 CT output channels, its own LDS, no block barrier anywhere: wave w of a block owns
//   channels 16 CT (w mod nwn) .. and pixel sub-range w / nwn of the block's slab.
// * dY of a step (32 pixels x 16 CT channels, whole 64- or 128-byte pieces of a pixel's row) is loaded one step ahead into
//   registers, split into bf16 limbs (LIMBS = 3: the fp32 value is the exact sum of its limbs, k_gemm_x3.hip; LIMBS = 1: the
//   bf16 mode's rounding) and stored PIXEL-major, pitch 96 (32) bytes; the step's input halo ((R-1) S + KH rows of
//   (W-1) S + KW pixels, zero outside the image) likewise, CP channels = 8 (16) bytes per pixel.
// * Both MFMA operands come out of `ds_read_b64_tr_b16` (semantics: profiles/r03c_ds_read_b64_tr_b16_probe.txt): in a
//   16-lane group, source lane 4 r + q supplies the address of 4 consecutive bf16 of "row" r = one of 4 pixels, and result lane
//   i receives element i of the 16 the four quads q hold, for the four pixels.  The dense operand: quad q = channels 4 q .. 4 q + 3
//   of the pixel -> lane i holds channel i.  The gathered operand: a quad's ADDRESS is free, so quad q is the pixel shifted
//   by tap q of the row tile (CP = 4: 16 rows = 4 taps x 4 channels) or half of the pixel shifted by tap q / 2 (CP = 8:
//   2 taps x 8 channels) -- the (tap, channel) rows of the filter gradient are formed by the read itself: no table, no
//   predicate, no per-element work.  v_mfma_f32_16x16x32_bf16 then contracts 32 pixels; 6 limb products (LIMBS = 3) or 1.
// * Pixels are split over blockIdx.y (and the waves' sub-ranges) into slabs [split][tap][c][n], summed by the deterministic
//   second-stage reduction as for every other filter-gradient kernel.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "kernels.h"

namespace eg {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {
__device__ __forceinline__ unsigned pk2w(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_fw(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi_fw(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
// (x, y) -> packed limb pairs p[0 .. L-1]: x = sum of the low halves exactly for L = 3 (k_gemm_x3.hip: split_pair), the
// round-to-nearest-even bf16 for L = 1
template <int L>
__device__ __forceinline__ void split2w(float x, float y, unsigned (&p)[L]) {
  p[0] = pk2w(x, y);
  if constexpr (L == 3) {
    const float rx = x - lo_fw(p[0]), ry = y - hi_fw(p[0]);
    p[1] = pk2w(rx, ry);
    p[2] = pk2w(rx - lo_fw(p[1]), ry - hi_fw(p[1]));
  }
}
__device__ __forceinline__ int reflect_t(int i, int n) {
  i = i < 0 ? -i : i;
  return i >= n ? 2 * (n - 1) - i : i;
}
__device__ __forceinline__ bf16x8 op8w(const u32x2& lo, const u32x2& hi) {
  const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(bf16x8, v);
}
}  // namespace

#define EG_TRW(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

template <int CP, int KH, int KW, int S, int CT, int LIMBS>
__global__ __launch_bounds__(256, 2) void wgrad_pk_kernel(const WgradArgs a, float* __restrict__ out, int dense_out, int nwn) {
  constexpr int T = KH * KW;
  constexpr int TPT = 16 / CP;                        // taps of a 16-row tile
  constexpr int NRT = (T + TPT - 1) / TPT;            // row tiles
  constexpr int HPIX = KH * (31 * S + KW);            // halo pixels of a K step, at most (W = 32, R = 1)
  constexpr int NHP = (HPIX + 63) / 64;               // halo pixels per lane, at most
  constexpr int PB = CP * 2;                          // bytes of a halo pixel in a limb plane
  constexpr int XPL = (HPIX * PB + 15) & ~15;         // bytes of a halo limb plane
  constexpr int PITCH = CT == 2 ? 96 : 32;            // bytes of a dY pixel in a limb plane (8 consecutive pixels: 8 different 32-byte bank groups)
  constexpr int YPL = 32 * PITCH;
  constexpr int WLDS = LIMBS * (XPL + YPL);           // bytes of LDS per wave
  constexpr int QPP = 4 * CT;                         // channel quads of a dY pixel
  constexpr int PPI = 64 / QPP;                       // dY pixels per wave-wide load
  constexpr int NYJ = 32 / PPI;                       // wave-wide dY loads per K step
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * WLDS];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int C = a.C, N = a.N, IH = a.IH, IW = a.IW, OW = a.OW, OH = a.OH;
  const int wn = wave % nwn, ps = wave / nwn, PS = 4 / nwn;
  const int n0 = blockIdx.x * 128 + wn * 16 * CT;
  if (n0 >= N) return;                                // (no block barrier below: a wave without channels just leaves)
  const int split = blockIdx.y * PS + ps;
  const int M = a.NB * OH * OW;
  int chunk = (M + a.nsplit - 1) / a.nsplit;
  chunk = (chunk + 31) & ~31;
  const int p_lo = split * chunk, p_hi = min(M, p_lo + chunk);
  const int nsteps = p_hi > p_lo ? (p_hi - p_lo) >> 5 : 0;
  const int lOW = 31 - __builtin_clz(OW), lOH = 31 - __builtin_clz(OH);
  const int W = OW < 32 ? OW : 32, lW = 31 - __builtin_clz(W), R = 32 >> lW;
  const int HC = (W - 1) * S + KW, HR = (R - 1) * S + KH, npix = HR * HC;

  unsigned char* xs = lds + wave * WLDS;              // [limb][halo pixel][CP] bf16
  unsigned char* ys = xs + LIMBS * XPL;               // [limb][pixel][16 CT (+ pad)] bf16
  const unsigned xs_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)xs;
  const unsigned ys_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)ys;

  // ---- this lane's halo pixels (step-invariant position in the halo) and dY pieces -----------------------------------------
  int hy[NHP], hx[NHP];
#pragma unroll
  for (int i = 0; i < NHP; ++i) {
    const int p = lane + 64 * i;
    hy[i] = p < npix ? p / HC : -1;
    hx[i] = p < npix ? p - (p / HC) * HC : 0;
  }
  const int ypx = lane / QPP, yq = lane - ypx * QPP;   // pixel (+ PPI j) and channel quad of the lane's dY pieces
  const bool yok = n0 + 4 * yq < N;

  float xr[NHP][CP == 8 ? 8 : 3];
  f32x4 yr[NYJ];
  auto load_step = [&](int t) __attribute__((always_inline)) {
    const int m0 = p_lo + 32 * t;
#pragma unroll
    for (int j = 0; j < NYJ; ++j) {
      yr[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (yok) yr[j] = *reinterpret_cast<const f32x4*>(a.dy + (size_t)(m0 + ypx + PPI * j) * a.dy_ps + n0 + 4 * yq);
    }
    const int row0 = m0 >> lOW, nb = row0 >> lOH;
    const int iy0 = (row0 & (OH - 1)) * S - a.pad, ix0 = (m0 & (OW - 1)) * S - a.pad;
#pragma unroll
    for (int i = 0; i < NHP; ++i) {
      int iy = iy0 + hy[i], ix = ix0 + hx[i];
      if (a.reflect) {
        iy = reflect_t(iy, IH);
        ix = reflect_t(ix, IW);
      }
      const bool ok = hy[i] >= 0 && (unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW;
      const float* src = a.x + ((size_t)(nb * IH + (ok ? iy : 0)) * IW + (ok ? ix : 0)) * a.x_ps;
      if constexpr (CP == 8) {
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
        if (ok) {
          v0 = *reinterpret_cast<const f32x4*>(src);
          v1 = *reinterpret_cast<const f32x4*>(src + 4);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xr[i][e] = v0[e];
          xr[i][4 + e] = v1[e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 3; ++e) xr[i][e] = ok ? src[e] : 0.f;
      }
    }
  };
  auto stage_step = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NYJ; ++j) {
      unsigned q0[LIMBS], q1[LIMBS];
      split2w<LIMBS>(yr[j][0], yr[j][1], q0);
      split2w<LIMBS>(yr[j][2], yr[j][3], q1);
#pragma unroll
      for (int l = 0; l < LIMBS; ++l)
        *reinterpret_cast<u32x2*>(ys + l * YPL + (ypx + PPI * j) * PITCH + yq * 8) = (u32x2){q0[l], q1[l]};
    }
#pragma unroll
    for (int i = 0; i < NHP; ++i) {
      if (hy[i] < 0) continue;
      const int p = lane + 64 * i;
      if constexpr (CP == 8) {
        unsigned q[4][LIMBS];
#pragma unroll
        for (int e = 0; e < 4; ++e) split2w<LIMBS>(xr[i][2 * e], xr[i][2 * e + 1], q[e]);
#pragma unroll
        for (int l = 0; l < LIMBS; ++l) *reinterpret_cast<u32x4*>(xs + l * XPL + p * PB) = (u32x4){q[0][l], q[1][l], q[2][l], q[3][l]};
      } else {
        unsigned q0[LIMBS], q1[LIMBS];
        split2w<LIMBS>(xr[i][0], xr[i][1], q0);
        split2w<LIMBS>(xr[i][2], 0.f, q1);
#pragma unroll
        for (int l = 0; l < LIMBS; ++l) *reinterpret_cast<u32x2*>(xs + l * XPL + p * PB) = (u32x2){q0[l], q1[l]};
      }
    }
  };

  // ---- fragment addresses: source lane 4 r + q of a 16-lane group tg; K slot 8 tg + 4 h + r <-> pixel 4 tg + r + 16 h --------
  const int tg = lane >> 4, ti = lane & 15, tr = ti >> 2, tq = ti & 3;
  unsigned pixb[2], tapo[NRT];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int p = 4 * tg + tr + 16 * h;
    pixb[h] = xs_base + (unsigned)((((p >> lW) * S) * HC + (p & (W - 1)) * S) * PB);
  }
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) {
    int tap = rt * TPT + (CP == 8 ? (tq >> 1) : tq);
    if (tap >= T) tap = 0;                              // (a padded row reads a valid pixel; its result is not stored)
    const int ky = tap / KW, kx = tap - ky * KW;
    tapo[rt] = (unsigned)((ky * HC + kx) * PB + (CP == 8 ? (tq & 1) * 8 : 0));
  }
  const unsigned ya = ys_base + (unsigned)((4 * tg + tr) * PITCH + tq * 8);

  f32x4 acc[NRT][CT];
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (nsteps > 0) load_step(0);
  for (int t = 0; t < nsteps; ++t) {
    stage_step();
    if (t + 1 < nsteps) load_step(t + 1);
    // the wave's own LDS stores precede its reads below (one wave: LDS operations execute in issue order; the statement keeps
    // the compiler from sinking the stores past the assembly reads, which it does not know to read memory)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // every fragment register is an in/out operand of the wait that retires its read
    u32x2 yf[CT][LIMBS][2];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int l = 0; l < LIMBS; ++l) {
        if (ct == 0) {
          if (l == 0) { EG_TRW(yf[ct][l][0], ya, 0); EG_TRW(yf[ct][l][1], ya, 16 * PITCH); }
          if (l == 1) { EG_TRW(yf[ct][l][0], ya, YPL); EG_TRW(yf[ct][l][1], ya, YPL + 16 * PITCH); }
          if (l == 2) { EG_TRW(yf[ct][l][0], ya, 2 * YPL); EG_TRW(yf[ct][l][1], ya, 2 * YPL + 16 * PITCH); }
        } else {
          if (l == 0) { EG_TRW(yf[ct][l][0], ya, 32); EG_TRW(yf[ct][l][1], ya, 32 + 16 * PITCH); }
          if (l == 1) { EG_TRW(yf[ct][l][0], ya, 32 + YPL); EG_TRW(yf[ct][l][1], ya, 32 + YPL + 16 * PITCH); }
          if (l == 2) { EG_TRW(yf[ct][l][0], ya, 32 + 2 * YPL); EG_TRW(yf[ct][l][1], ya, 32 + 2 * YPL + 16 * PITCH); }
        }
      }
    if constexpr (CT == 2 && LIMBS == 3)
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(yf[0][0][0]), "+v"(yf[0][0][1]), "+v"(yf[0][1][0]), "+v"(yf[0][1][1]), "+v"(yf[0][2][0]), "+v"(yf[0][2][1]),
                     "+v"(yf[1][0][0]), "+v"(yf[1][0][1]), "+v"(yf[1][1][0]), "+v"(yf[1][1][1]), "+v"(yf[1][2][0]), "+v"(yf[1][2][1])::"memory");
    else if constexpr (CT == 1 && LIMBS == 3)
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(yf[0][0][0]), "+v"(yf[0][0][1]), "+v"(yf[0][1][0]), "+v"(yf[0][1][1]), "+v"(yf[0][2][0]), "+v"(yf[0][2][1])::"memory");
    else if constexpr (CT == 2 && LIMBS == 1)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(yf[0][0][0]), "+v"(yf[0][0][1]), "+v"(yf[1][0][0]), "+v"(yf[1][0][1])::"memory");
    else
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(yf[0][0][0]), "+v"(yf[0][0][1])::"memory");
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
      u32x2 xf[LIMBS][2];
      const unsigned x0 = pixb[0] + tapo[rt], x1 = pixb[1] + tapo[rt];
      EG_TRW(xf[0][0], x0, 0);
      EG_TRW(xf[0][1], x1, 0);
      if constexpr (LIMBS == 3) {
        EG_TRW(xf[1][0], x0, XPL);
        EG_TRW(xf[1][1], x1, XPL);
        EG_TRW(xf[2][0], x0, 2 * XPL);
        EG_TRW(xf[2][1], x1, 2 * XPL);
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(xf[0][0]), "+v"(xf[0][1]), "+v"(xf[1][0]), "+v"(xf[1][1]), "+v"(xf[2][0]), "+v"(xf[2][1])::"memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xf[0][0]), "+v"(xf[0][1])::"memory");
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        if constexpr (LIMBS == 3) {
          // six limb products down to 2^-16 of the leading one, smallest first: (dY limb, X limb)
          const bf16x8 y0 = op8w(yf[ct][0][0], yf[ct][0][1]), y1 = op8w(yf[ct][1][0], yf[ct][1][1]), y2 = op8w(yf[ct][2][0], yf[ct][2][1]);
          const bf16x8 b0 = op8w(xf[0][0], xf[0][1]), b1 = op8w(xf[1][0], xf[1][1]), b2 = op8w(xf[2][0], xf[2][1]);
          acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y2, b0, acc[rt][ct], 0, 0, 0);
          acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y0, b2, acc[rt][ct], 0, 0, 0);
          acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y1, b1, acc[rt][ct], 0, 0, 0);
          acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y1, b0, acc[rt][ct], 0, 0, 0);
          acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y0, b1, acc[rt][ct], 0, 0, 0);
          acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y0, b0, acc[rt][ct], 0, 0, 0);
        } else {
          acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(op8w(yf[ct][0][0], yf[ct][0][1]), op8w(xf[0][0], xf[0][1]), acc[rt][ct], 0, 0, 0);
        }
      }
    }
  }

  // D = (dY columns) x (X rows): lane l holds filter row l & 15 of the tile = (tap, channel) and, in its four registers,
  // output channels 16 ct + 4 (l >> 4) .. + 3: one 16-byte store per tile
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) {
    const int tap = rt * TPT + ti / CP, cch = ti % CP;
    if (tap >= T || cch >= C) continue;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int n = n0 + 16 * ct + 4 * tg;
      if (n >= N) continue;
      float* q = dense_out ? out + (((size_t)split * T + tap) * C + cch) * N + n : out + (size_t)tap * a.w_ts + (size_t)cch * a.ldw + n;
      f32x4 v = acc[rt][ct];
      if (!dense_out && a.accumulate) v += *reinterpret_cast<const f32x4*>(q);
      *reinterpret_cast<f32x4*>(q) = v;
    }
  }
}
#undef EG_TRW

// pixel sub-ranges a block's four waves divide its slab into (the waves not needed along the output channels)
static int wgrad_pk_nwn(const WgradArgs& a) {
  const int per_wave = a.N >= 32 ? 32 : 16;
  return std::min(4, (std::min(a.N, 128) + per_wave - 1) / per_wave);
}

bool wgrad_pk_eligible(const WgradArgs& a) {
  static const bool off = dbg_env("EG_NO_WGRAD_PK") != nullptr;
  auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
  if (off || !(a.limbs == 3 || a.limbs == 1) || !(a.C == 3 || a.C == 8) || a.KH != a.KW) return false;
  const bool shape = a.C == 3 ? ((a.KW == 3 && a.S == 1) || (a.KW == 4 && a.S == 2) || (a.KW == 7 && a.S == 1) || (a.KW == 5 && a.S == 2))
                              : ((a.KW == 3 && a.S == 1) || (a.KW == 1 && a.S == 1));
  if (!shape || !(a.N == 8 || a.N == 16 || a.N % 32 == 0) || a.dy_ps % 4 != 0 || (((uintptr_t)a.dy) & 15) != 0) return false;
  if (a.C == 8 && (a.x_ps % 4 != 0 || (((uintptr_t)a.x) & 15) != 0)) return false;
  if (!pow2(a.OW) || !pow2(a.OH) || a.OW < 4 || a.OH * a.OW < 32 || a.ldw % 4 != 0 || a.w_ts % 4 != 0) return false;
  return (long long)a.NB * a.IH * a.IW * a.x_ps < (1ll << 31) && (long long)a.NB * a.OH * a.OW * a.dy_ps < (1ll << 31);
}

// pixel splits: blocks x sub-ranges; two blocks per CU resident, every wave at least 4 K steps
int wgrad_pk_splits(const WgradArgs& a) {
  static const int target = dbg_env("EG_WGRAD_PK_BLOCKS") ? atoi(dbg_env("EG_WGRAD_PK_BLOCKS")) : 512;
  const int ps = 4 / wgrad_pk_nwn(a);
  const long long M = (long long)a.NB * a.OH * a.OW;
  const int tiles_x = (a.N + 127) / 128;
  long long gy = std::max(1, target / tiles_x);
  gy = std::min(gy, std::max(1LL, M / (128LL * ps)));
  return (int)(gy * ps);
}

template <int CP, int KH, int KW, int S>
static void wgrad_pk_launch(const WgradArgs& a, float* out, int dense, hipStream_t s) {
  const int nwn = wgrad_pk_nwn(a), ps = 4 / nwn;
  dim3 grid((a.N + 127) / 128, a.nsplit / ps);
  // three limbs in BOTH reduced-precision modes: the bf16 mode rounds the operands of its dense contractions only
  // (oracle: bf16_wgrad_rule, c_gathered > 16); these layers keep fp32 results there, as conv_pk does for the forwards
  if (a.N >= 32)
    hipLaunchKernelGGL((wgrad_pk_kernel<CP, KH, KW, S, 2, 3>), grid, dim3(256), 0, s, a, out, dense, nwn);
  else
    hipLaunchKernelGGL((wgrad_pk_kernel<CP, KH, KW, S, 1, 3>), grid, dim3(256), 0, s, a, out, dense, nwn);
}

const char* launch_wgrad_pk(const WgradArgs& a, float* out, int dense, hipStream_t s) {
  if (a.nsplit % (4 / wgrad_pk_nwn(a)) != 0) {
    fprintf(stderr, "wgrad_pk: %d pixel splits for %d sub-ranges per block\n", a.nsplit, 4 / wgrad_pk_nwn(a));
    abort();
  }
  if (a.C == 8) {
    if (a.KW == 3) { wgrad_pk_launch<8, 3, 3, 1>(a, out, dense, s); return "wgrad_pk<c8,k3>"; }
    wgrad_pk_launch<8, 1, 1, 1>(a, out, dense, s);
    return "wgrad_pk<c8,k1>";
  }
  if (a.KW == 3) { wgrad_pk_launch<4, 3, 3, 1>(a, out, dense, s); return "wgrad_pk<c3,k3>"; }
  if (a.KW == 4) { wgrad_pk_launch<4, 4, 4, 2>(a, out, dense, s); return "wgrad_pk<c3,k4s2>"; }
  if (a.KW == 7) { wgrad_pk_launch<4, 7, 7, 1>(a, out, dense, s); return "wgrad_pk<c3,k7>"; }
  wgrad_pk_launch<4, 5, 5, 2>(a, out, dense, s);
  return "wgrad_pk<c3,k5s2>";
}

}  // namespace eg
