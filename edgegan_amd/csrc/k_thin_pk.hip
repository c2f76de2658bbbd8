// Thin forward convolutions (3 or 8 input channels) with (tap, channel) PACKED into the MFMA's K -- VERDICT r3 item 4.
//
//   conv_pk_kernel<CP, KH, KW, S> : Y[oy][ox][n] (+)= b[n] + sum_{ky,kx,c} X[oy S + ky - pad][ox S + kx - pad][c] W[ky][kx][c][n]
//       CP = 8: C == 8, 3x3 stride 1        (the classifier's Conv_1 / update gate of unit 1, edgegan/nn/modules/conv.py:185-224)
//       CP = 4: C == 3, 3x3 stride 1 / 4x4 stride 2  (image-pyramid convolutions of the MRU units, conv.py:199-213; d_conv_0 of
//               the three discriminators and the encoder stem, edgegan/models/discriminator.py:61-63, encoder.py:61-63)
//
// conv_smallc_kernel (k_thin.hip) builds these layers' MFMA operands element by element from a (tap, channel) table with a
// padding predicate per element: ~5,000 issued instructions per 640 fp32 MFMAs, on neither roofline (r03x_smallc_limiter.txt).
// Here the input tile is staged ONCE as a zero-padded halo image of bf16 limb planes [plane][row][col][CP] (the fp32 value is
// the exact sum of its three limbs, k_gemm_x3.hip), and because a pixel's CP channels are 16 (CP = 8) or 8 (CP = 4) contiguous
// bytes, the 8 K-values a lane feeds to v_mfma_f32_32x32x16_bf16 are ONE 16-byte LDS read: one tap's 8 channels (CP = 8) or two
// horizontally adjacent taps' 4 + 4 channels (CP = 4) of the pixel under the lane's output pixel -- no table, no predicate,
// no per-element work.  A K step of 16 is 2 taps (CP = 8: 5 steps for 3x3) or 4 taps (CP = 4: 3 steps for 3x3, 4 for 4x4).
// The filter fragments (K steps x 3 limb planes) live in registers for the whole block; six limb products per step give
// fp32 products (dropped terms < 2^-23), as everywhere in the f32x3 mode.  These layers then cost what their output costs to
// write: the kernel is HBM-bound (algorithmic bytes = input + output + filter).
//
// Block = 256 threads = 4 waves on an output tile of TR = 8 rows x 32 columns x (32 NWN) channels, NWN = min(4, N / 32) waves
// along the channels, the others along the rows.  Used by the f32x3 and bf16 modes (the exact-fp32 mode keeps conv_smallc).
//
// Round 4, second pass -- the rest of what conv_smallc ran in those modes:
//   * input gradients of the stride-1 layers (ConvArgs::D == -1: input pixel = out + off - tap): the same kernel with the halo
//     origin moved by KH - 1 and the filter taps read mirrored (conv.py:13-36 differentiated; 8 -> 3 / 8 channels);
//   * any N (3-channel results are stored element-wise);
//   * 7x7 (the classifier's stem, edgegan/models/classifier.py:39, and its input gradient), 5x5 stride 2 (the input gradient of
//     the generators' 3-channel output layer, conv.py:39-58), 1x1 (Conv_3 of the first MRU unit): with up to 25 K steps the
//     filter fragments no longer fit the registers -- FLDS keeps them in LDS in fragment order (one 16-byte read per step and limb).
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "kernels.h"

namespace eg {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {
__device__ __forceinline__ unsigned pk2(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_f(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi_f(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
// (x, y) -> packed limb pairs p0, p1, p2 with x = lo(p0) + lo(p1) + lo(p2) exactly (k_gemm_x3.hip: split_pair)
__device__ __forceinline__ void split2(float x, float y, unsigned& p0, unsigned& p1, unsigned& p2) {
  p0 = pk2(x, y);
  const float rx = x - lo_f(p0), ry = y - hi_f(p0);
  p1 = pk2(rx, ry);
  p2 = pk2(rx - lo_f(p1), ry - hi_f(p1));
}
}  // namespace

template <int CP, int KH, int KW, int S, bool FLDS>
__global__ __launch_bounds__(256, (FLDS && (CP == 8 || KH == 5)) ? 1 : 2) void conv_pk_kernel(const ConvArgs a, int tiles_x, int tiles_y, int nwn) {
  constexpr int TR = 8;                                   // output rows of a tile
  constexpr int TPS = CP == 8 ? 2 : 4;                    // taps per K step of 16 (8 K-values per lane half)
  constexpr int PR = (KW + 1) / 2;                        // CP = 4: horizontal tap pairs per filter row
  constexpr int NS = CP == 8 ? (KH * KW + 1) / 2 : (KH * PR + 1) / 2;   // K steps
  constexpr int HR = (TR - 1) * S + KH;                   // halo rows
  constexpr int HC = ((31 * S + KW + 1) + 3) & ~3;        // halo columns (+1: the second tap of a pair past the last column)
  constexpr int PB = CP * 2;                              // bytes of a pixel in a limb plane
  constexpr int PLANE = HR * HC * PB;                     // bytes of a limb plane
  constexpr bool B128 = (CP == 8) || (S == 2);            // a lane's 16 bytes are 16-byte aligned (else two 8-byte reads)
  constexpr int FNW = KH == 5 ? 2 : 1;                    // FLDS: 32-channel groups the filter image holds (conv_pk_eligible)
  static_assert(TPS * CP == 16, "a K step of 16 = TPS taps x CP channels");
  __shared__ __attribute__((aligned(16))) unsigned char Hs[3 * PLANE];
  __shared__ __attribute__((aligned(16))) unsigned char Fs[FLDS ? NS * 3 * FNW * 1024 : 16];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int C = a.C, N = a.N, IH = a.IH, IW = a.IW;
  const ConvPhase ph = a.ph[0];
  const int OH = ph.OHp, OW = ph.OWp;
  const int tpi = tiles_x * tiles_y;
  const int nb = blockIdx.x / tpi, trem = blockIdx.x - nb * tpi;
  const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
  const int oy0 = ty * TR, ox0 = tx * 32;
  const bool mirror = a.D < 0;                            // input gradient (S == 1): pixel = out + off - tap = out + off - (K - 1) + (K - 1 - tap)
  const int iy0 = oy0 * S + ph.offy - (mirror ? KH - 1 : 0), ix0 = ox0 * S + ph.offx - (mirror ? KW - 1 : 0);   // input pixel of halo (0, 0)
  const int wn = wave % nwn, wr = wave / nwn, nwr = 4 / nwn;
  const int n0 = (blockIdx.y * nwn + wn) * 32;

  // ---- filter fragments: lane (n = n0 + l31, K half lh) holds, per K step, the 8 K-values of its half as three limbs --------
  auto filter_frag = [&](int s, bf16x8 (&f)[3]) __attribute__((always_inline)) {
    const int n = n0 + l31;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int ky = 0, kx = 0, c = 0;
      bool ok;
      if constexpr (CP == 8) {
        const int tap = 2 * s + lh;
        c = e;
        ok = tap < KH * KW;
        ky = tap / KW;
        kx = tap - ky * KW;
      } else {
        const int pi = 2 * s + lh;
        ky = pi / PR;
        kx = 2 * (pi - ky * PR) + (e >> 2);
        c = e & 3;
        ok = ky < KH && kx < KW;
      }
      const int wt = mirror ? (KH - 1 - ky) * KW + (KW - 1 - kx) : ky * KW + kx;
      v[e] = (ok && c < C && n < N) ? a.wgt[(size_t)wt * a.w_ts + (size_t)c * a.ldw + n] : 0.f;
    }
    unsigned q0[4], q1[4], q2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) split2(v[2 * q], v[2 * q + 1], q0[q], q1[q], q2[q]);
    f[0] = __builtin_bit_cast(bf16x8, (u32x4){q0[0], q0[1], q0[2], q0[3]});
    f[1] = __builtin_bit_cast(bf16x8, (u32x4){q1[0], q1[1], q1[2], q1[3]});
    f[2] = __builtin_bit_cast(bf16x8, (u32x4){q2[0], q2[1], q2[2], q2[3]});
  };
  bf16x8 bfr[FLDS ? 1 : NS][3];
  if constexpr (FLDS) {
    for (int s = wr; s < NS; s += nwr) {                  // the waves of a 32-channel group share the group's image
      bf16x8 f[3];
      filter_frag(s, f);
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x8*>(&Fs[(((s * 3 + p) * FNW + wn) * 64 + lane) * 16]) = f[p];
    }
  } else {
#pragma unroll
    for (int s = 0; s < NS; ++s) filter_frag(s, bfr[s]);
  }

  // ---- halo image: every pixel (zero outside the image) as CP channels x 3 limb planes ------------------------------------
  for (int p = tid; p < HR * HC; p += 256) {
    const int hy = p / HC, hx = p - hy * HC;
    const int iy = iy0 + hy, ix = ix0 + hx;
    const bool in = (unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW;
    const float* src = a.x + ((size_t)(nb * IH + (in ? iy : 0)) * IW + (in ? ix : 0)) * a.x_ps;
    if constexpr (CP == 8) {
      float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
      if (in) {
        lo = *reinterpret_cast<const float4*>(src);
        hi = *reinterpret_cast<const float4*>(src + 4);
      }
      unsigned q0[4], q1[4], q2[4];
      split2(lo.x, lo.y, q0[0], q1[0], q2[0]);
      split2(lo.z, lo.w, q0[1], q1[1], q2[1]);
      split2(hi.x, hi.y, q0[2], q1[2], q2[2]);
      split2(hi.z, hi.w, q0[3], q1[3], q2[3]);
      *reinterpret_cast<u32x4*>(&Hs[p * PB]) = (u32x4){q0[0], q0[1], q0[2], q0[3]};
      *reinterpret_cast<u32x4*>(&Hs[PLANE + p * PB]) = (u32x4){q1[0], q1[1], q1[2], q1[3]};
      *reinterpret_cast<u32x4*>(&Hs[2 * PLANE + p * PB]) = (u32x4){q2[0], q2[1], q2[2], q2[3]};
    } else {
      float v0 = 0.f, v1 = 0.f, v2 = 0.f;
      if (in) {
        v0 = src[0];
        v1 = src[1];
        v2 = src[2];
      }
      unsigned q0[2], q1[2], q2[2];
      split2(v0, v1, q0[0], q1[0], q2[0]);
      split2(v2, 0.f, q0[1], q1[1], q2[1]);
      *reinterpret_cast<u32x2*>(&Hs[p * PB]) = (u32x2){q0[0], q0[1]};
      *reinterpret_cast<u32x2*>(&Hs[PLANE + p * PB]) = (u32x2){q1[0], q1[1]};
      *reinterpret_cast<u32x2*>(&Hs[2 * PLANE + p * PB]) = (u32x2){q2[0], q2[1]};
    }
  }
  __syncthreads();

  // ---- per-lane fragment offsets: K step s, half lh -> the tap (pair) the lane's 16 bytes start at ---------------------------
  const unsigned hs_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&Hs[0];
  unsigned aoff[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    int ky, kx;
    if constexpr (CP == 8) {
      const int tap = min(2 * s + lh, KH * KW - 1);     // (a padded tap reads a valid pixel; its filter rows are zero)
      ky = tap / KW;
      kx = tap - ky * KW;
    } else {
      const int pi = 2 * s + lh;
      ky = min(pi / PR, KH - 1);
      kx = 2 * (pi % PR);
    }
    aoff[s] = hs_base + (unsigned)((ky * HC + l31 * S + kx) * PB);
  }
  const bool vec_out = (N & 3) == 0 && (a.y_ps & 3) == 0;

  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  auto a_frag = [&](int s, int p, unsigned rowoff) __attribute__((always_inline)) -> bf16x8 {
    const __attribute__((address_space(3))) unsigned char* q =
        (const __attribute__((address_space(3))) unsigned char*)(uintptr_t)(aoff[s] + rowoff + p * PLANE);
    if constexpr (B128) {
      return *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(q);
    } else {
      const u32x2 lo = *reinterpret_cast<const __attribute__((address_space(3))) u32x2*>(q);
      const u32x2 hi = *reinterpret_cast<const __attribute__((address_space(3))) u32x2*>(q + 8);
      const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
      return __builtin_bit_cast(bf16x8, v);
    }
  };
  // six limb products down to 2^-16 of the leading one: (0,0) (0,1) (0,2) (1,0) (1,1) (2,0), smallest first
  auto products = [&](f32x16& acc, const bf16x8 (&b)[3], const bf16x8 (&af)[3]) __attribute__((always_inline)) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[2], af[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0], af[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[1], af[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[1], af[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0], af[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0], af[0], acc, 0, 0, 0);
  };
  // D^T layout: lane l holds output pixel ox0 + (l & 31) and, in registers 4q .. 4q+3, channels 8q + 4 (l >> 5) .. +3
  auto store_row = [&](const f32x16& acc, int oy) __attribute__((always_inline)) {
    const int ox = ox0 + l31;
    if (ox >= OW) return;
    float* yrow = a.y + ((size_t)(nb * a.OHf + oy) * a.OWf + ox) * a.y_ps;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + q * 8 + lh * 4;
      if (n >= N) continue;
      float4 v = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
      if (vec_out) {
        const float4 b4 = a.bias ? *reinterpret_cast<const float4*>(a.bias + n) : zero4;
        v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
        if (a.accumulate) {
          const float4 o = *reinterpret_cast<const float4*>(yrow + n);
          v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        if (a.act == ACT_RELU) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        } else if (a.act == ACT_LRELU) {
          v.x = fmaxf(v.x, 0.2f * v.x); v.y = fmaxf(v.y, 0.2f * v.y); v.z = fmaxf(v.z, 0.2f * v.z); v.w = fmaxf(v.w, 0.2f * v.w);
        }
        if (!a.y_skip) *reinterpret_cast<float4*>(yrow + n) = v;
        if (a.y_bf) {   // the plane of the value just stored, rounded to nearest even (k_elem.hip: st_bf4 -- what cvt_bf16 would make of it)
          const f32x2 lo2 = {v.x, v.y}, hi2 = {v.z, v.w};
          const u32x2 o = {__builtin_bit_cast(unsigned, __builtin_convertvector(lo2, bf16x2)),
                           __builtin_bit_cast(unsigned, __builtin_convertvector(hi2, bf16x2))};
          *reinterpret_cast<u32x2*>(a.y_bf + (yrow - a.y) + n) = o;
        }
      } else {
        const float e4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (n + j >= N) continue;
          float r1 = e4[j] + (a.bias ? a.bias[n + j] : 0.f);
          if (a.accumulate) r1 += yrow[n + j];
          if (a.act == ACT_RELU) r1 = fmaxf(r1, 0.f);
          else if (a.act == ACT_LRELU) r1 = fmaxf(r1, 0.2f * r1);
          yrow[n + j] = r1;
        }
      }
    }
  };
  if constexpr (FLDS) {
    // two output rows per pass share a step's filter fragments (nwn <= 2: a wave owns at least two rows of the tile); the
    // fragments of step s + 1 are requested before the products of step s
    for (int r = wr; r < TR; r += 2 * nwr) {
      const int oy = oy0 + r, oy2 = oy + nwr;
      if (oy >= OH) break;
      const bool two = r + nwr < TR && oy2 < OH;
      f32x16 acc0, acc1;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc0[q] = acc1[q] = 0.f;
      const unsigned ro0 = (unsigned)(r * S * HC * PB), ro1 = two ? (unsigned)((r + nwr) * S * HC * PB) : ro0;
      bf16x8 af0[2][3], af1[2][3], bq[2][3];
      auto load = [&](int sl, int s) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          bq[sl][p] = *reinterpret_cast<const bf16x8*>(&Fs[(((s * 3 + p) * FNW + wn) * 64 + lane) * 16]);
          af0[sl][p] = a_frag(s, p, ro0);
          af1[sl][p] = a_frag(s, p, ro1);
        }
      };
      load(0, 0);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if (s + 1 < NS) load((s + 1) & 1, s + 1);
        products(acc0, bq[s & 1], af0[s & 1]);
        products(acc1, bq[s & 1], af1[s & 1]);
      }
      store_row(acc0, oy);
      if (two) store_row(acc1, oy2);
    }
  } else {
    for (int r = wr; r < TR; r += nwr) {
      const int oy = oy0 + r;
      if (oy >= OH) break;
      f32x16 acc;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = 0.f;
      const unsigned rowoff = (unsigned)(r * S * HC * PB);
      bf16x8 af[NS][3];
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int p = 0; p < 3; ++p) af[s][p] = a_frag(s, p, rowoff);
#pragma unroll
      for (int s = 0; s < NS; ++s) products(acc, bfr[s], af[s]);
      store_row(acc, oy);
    }
  }
}

// shape class of a launch: 0 = none, else the template instance (CP, K, S)
static int conv_pk_class(const ConvArgs& a) {
  const ConvPhase& p = a.ph[0];
  if (p.Ty != p.Tx || p.Ty != a.KW) return 0;
  const int k = a.KW;
  if (a.C == 3) {
    if (a.S == 1 && k == 3) return 1;
    if (a.S == 2 && k == 4 && a.D == 1) return 2;
    if (a.S == 1 && k == 7 && a.N <= 32) return 3;
    if (a.S == 2 && k == 5 && a.D == 1 && a.N <= 64) return 4;
    return 0;
  }
  if (a.C == 8 && a.S == 1) {
    if (k == 3) return 5;
    if (k == 7 && a.N <= 32) return 6;
    if (k == 1) return 7;
  }
  return 0;
}

bool conv_pk_eligible(const ConvArgs& a) {
  static const bool off = dbg_env("EG_NO_PK") != nullptr;
  static const bool old_only = dbg_env("EG_PK_FWD_ONLY") != nullptr;   // A/B: the round's first version (forward 3x3 / 4x4 only)
  if (off || !a.pk || !a.wgt || a.nphase != 1 || a.reflect || a.shuf || a.ksplit > 1 || a.os != 1 || a.kst != 1) return false;
  const ConvPhase& p = a.ph[0];
  if (!(a.D == 1 || (a.D == -1 && a.S == 1)) || p.kh0 != 0 || p.kw0 != 0 || p.ophy != 0 || p.ophx != 0) return false;
  const int cls = conv_pk_class(a);
  if (!cls) return false;
  if (old_only && (a.D != 1 || !(cls == 1 || cls == 2 || cls == 5) || a.N < 8 || a.N % 4 != 0)) return false;
  if (a.C == 8 && (a.x_ps % 4 != 0 || (((uintptr_t)a.x) & 15) != 0)) return false;
  if (a.N % 4 == 0 && a.y_ps % 4 == 0 && ((((uintptr_t)a.y) & 15) != 0 || (a.bias && (((uintptr_t)a.bias) & 15) != 0))) return false;
  return (long long)a.NB * a.IH * a.IW * a.x_ps < (1ll << 31);
}

const char* launch_conv_pk(const ConvArgs& a, hipStream_t s) {
  const ConvPhase& p = a.ph[0];
  const int tiles_x = (p.OWp + 31) / 32, tiles_y = (p.OHp + 7) / 8;
  const int nb32 = (a.N + 31) / 32;
  const int nwn = nb32 >= 4 ? 4 : nb32 >= 2 ? 2 : 1;
  dim3 grid(a.NB * tiles_x * tiles_y, (nb32 + nwn - 1) / nwn);
  switch (conv_pk_class(a)) {
    case 1: hipLaunchKernelGGL((conv_pk_kernel<4, 3, 3, 1, false>), grid, dim3(256), 0, s, a, tiles_x, tiles_y, nwn); return a.D > 0 ? "conv_pk<c3,k3>" : "conv_pk<c3,k3,dgrad>";
    case 2: hipLaunchKernelGGL((conv_pk_kernel<4, 4, 4, 2, false>), grid, dim3(256), 0, s, a, tiles_x, tiles_y, nwn); return "conv_pk<c3,k4s2>";
    case 3: hipLaunchKernelGGL((conv_pk_kernel<4, 7, 7, 1, true>), grid, dim3(256), 0, s, a, tiles_x, tiles_y, nwn); return a.D > 0 ? "conv_pk<c3,k7>" : "conv_pk<c3,k7,dgrad>";
    case 4: hipLaunchKernelGGL((conv_pk_kernel<4, 5, 5, 2, true>), grid, dim3(256), 0, s, a, tiles_x, tiles_y, nwn); return "conv_pk<c3,k5s2>";
    case 5: hipLaunchKernelGGL((conv_pk_kernel<8, 3, 3, 1, false>), grid, dim3(256), 0, s, a, tiles_x, tiles_y, nwn); return a.D > 0 ? "conv_pk<c8,k3>" : "conv_pk<c8,k3,dgrad>";
    case 6: hipLaunchKernelGGL((conv_pk_kernel<8, 7, 7, 1, true>), grid, dim3(256), 0, s, a, tiles_x, tiles_y, nwn); return a.D > 0 ? "conv_pk<c8,k7>" : "conv_pk<c8,k7,dgrad>";
    default: hipLaunchKernelGGL((conv_pk_kernel<8, 1, 1, 1, false>), grid, dim3(256), 0, s, a, tiles_x, tiles_y, nwn); return "conv_pk<c8,k1>";
  }
}

}  // namespace eg
