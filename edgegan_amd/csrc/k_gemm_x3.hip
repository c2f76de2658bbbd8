// fp32 contractions on the bf16 matrix pipe: three-limb operand split (eg_config.compute_dtype == 2).
//
// An fp32 number a is EXACTLY the sum of three bf16 numbers: a0 = rne_bf16(a), a1 = rne_bf16(a - a0),
// a2 = a - a0 - a1 (each residual is exactly representable in fp32 and the last one fits bf16's 8 significant bits),
// with |a1| <= 2^-8 |a| and |a2| <= 2^-16 |a|.  A product of two fp32 numbers is therefore the exact sum of the nine
// limb products a_i * b_j, every one of which the bf16 MFMA forms exactly (8 x 8 significant bits) and accumulates in
// fp32.  The kernels issue the limb products down to relative magnitude 2^-16 -- (0,0) (0,1) (1,0) (1,1) (0,2) (2,0),
// "NPROD = 6" -- or all nine (NPROD = 9); the three terms NPROD = 6 drops are bounded by 2^-23 |a b| together, i.e.
// below one fp32 unit in the last place of the product they belong to, and enter a sum whose own fp32 accumulation
// rounds at 2^-24 of the running total per step.  Measured against the fp64 oracle the results are as close as the
// exact-fp32 MFMA's (tests/test_x3_gpu.py); v_mfma_f32_16x16x32_bf16 runs 16x the rate of v_mfma_f32_16x16x4_f32,
// so six limb products cost 6/16 of the fp32 instruction's time.
//
//   conv_gemm_x3_kernel : Y[pix_out(m)][n] (+)= b[n] + sum_taps sum_c X[pix_in(m,tap)][c] * W[tap][c][n]
//                         X is gathered as fp32 and split on its way into LDS; W comes as three bf16 planes with the
//                         contracted index contiguous (launch_split_bf16x3 / launch_tap_transpose_bf16x3)
//   wgrad_x3_kernel     : dW[tap][c][n] = sum_m X[pix_in(m,tap)][c] * dY[m][n]      (both operands split in the kernel)
//
// LDS images are the bf16 kernels' ([row][32 k], 64-byte rows, XOR chunk swizzle: every fragment is one conflict-free
// ds_read_b128), three planes per operand.  One LDS stage (48 KB for the 128x128 tile) + register-staged prefetch of
// the next K tile: two blocks per CU, so one block's split / LDS traffic runs under the other block's MFMAs.
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace eg {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int BKX = 32;    // contraction elements per K step (one MFMA)
constexpr int ROWX = BKX;  // LDS row pitch in bf16 elements (64 bytes)

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_f(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi_f(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// (x, y) -> packed limb pairs p0, p1, p2 with x = lo(p0) + lo(p1) + lo(p2) exactly (same for y in the high halves)
__device__ __forceinline__ void split_pair(float x, float y, unsigned& p0, unsigned& p1, unsigned& p2) {
  p0 = pack2(x, y);
  const float rx = x - lo_f(p0), ry = y - hi_f(p0);
  p1 = pack2(rx, ry);
  p2 = pack2(rx - lo_f(p1), ry - hi_f(p1));
}
__device__ __forceinline__ void split8(const float4& a, const float4& b, uint4& l0, uint4& l1, uint4& l2) {
  split_pair(a.x, a.y, l0.x, l1.x, l2.x);
  split_pair(a.z, a.w, l0.y, l1.y, l2.y);
  split_pair(b.x, b.y, l0.z, l1.z, l2.z);
  split_pair(b.z, b.w, l0.w, l1.w, l2.w);
}
__device__ __forceinline__ void split8v(const f32x4& a, const f32x4& b, u32x4& l0, u32x4& l1, u32x4& l2) {
  unsigned p[4][3];
  split_pair(a[0], a[1], p[0][0], p[0][1], p[0][2]);
  split_pair(a[2], a[3], p[1][0], p[1][1], p[1][2]);
  split_pair(b[0], b[1], p[2][0], p[2][1], p[2][2]);
  split_pair(b[2], b[3], p[3][0], p[3][1], p[3][2]);
  l0 = (u32x4){p[0][0], p[1][0], p[2][0], p[3][0]};
  l1 = (u32x4){p[0][1], p[1][1], p[2][1], p[3][1]};
  l2 = (u32x4){p[0][2], p[1][2], p[2][2], p[3][2]};
}
__device__ __forceinline__ int reflect_x(int i, int n) {
  i = i < 0 ? -i : i;
  return i >= n ? 2 * (n - 1) - i : i;
}
// physical 16-byte chunk = logical chunk ^ swz(row)   (k_gemm_bf16.hip: lds_swz)
__device__ __forceinline__ int swz(int row) { return ((-((row >> 2) & 3)) & 3) ^ ((row >> 4) & 1); }

// the limb products of one K step for a wave's MI x NJ accumulators; the fragment loads are ordered so that at most
// four limb fragment sets are live at a time
template <int MI, int NJ, int NPROD, typename LA, typename LB>
__device__ __forceinline__ void limb_products(f32x4 (&acc)[MI][NJ], LA load_a, LB load_b) {
  auto mm = [&](const bf16x8 (&af)[MI], const bf16x8 (&bf)[NJ]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
  };
  bf16x8 b0[NJ], b1[NJ];
  {
    bf16x8 a0[MI];
    load_a(0, a0);
    load_b(0, b0);
    mm(a0, b0);
    load_b(1, b1);
    mm(a0, b1);
    bf16x8 b2[NJ];
    load_b(2, b2);
    mm(a0, b2);
    if constexpr (NPROD >= 8) {
      bf16x8 a1[MI];
      load_a(1, a1);
      mm(a1, b0);
      mm(a1, b1);
      mm(a1, b2);
      bf16x8 a2[MI];
      load_a(2, a2);
      mm(a2, b0);
      mm(a2, b1);
      if constexpr (NPROD >= 9) mm(a2, b2);
      return;
    }
  }
  {
    bf16x8 a1[MI];
    load_a(1, a1);
    mm(a1, b0);
    mm(a1, b1);
  }
  bf16x8 a2[MI];
  load_a(2, a2);
  mm(a2, b0);
}

// The same with v_mfma_f32_32x32x16_bf16 (wave tile 64 x 64 = 2 x 2 accumulators of 32 x 32, two K halves of 16 per K
// step): half as many, twice as long matrix instructions -- a 32-cycle MFMA leaves room for ~5 other instructions of the
// SIMD's second wave per MFMA, a 16-cycle one for ~1, and the limb split needs ~2.6 per 32-cycle MFMA.
// A fragment set is [block 0..1][K half 0..1]; lane l supplies row l & 31 and the 8 k of K-half chunk l >> 5.
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NPROD, typename LA, typename LB>
__device__ __forceinline__ void limb_products32(f32x16 (&acc)[2][2], LA load_a, LB load_b) {
  auto mm = [&](const bf16x8 (&af)[2][2], const bf16x8 (&bf)[2][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j][kk], af[i][kk], acc[i][j], 0, 0, 0);
  };
  bf16x8 b0[2][2], b1[2][2];
  if constexpr (NPROD == 1) {   // one plane: the operands ROUNDED to bf16 (compute_dtype bf16)
    bf16x8 a0[2][2];
    load_a(0, a0);
    load_b(0, b0);
    mm(a0, b0);
    return;
  }
  {
    bf16x8 a0[2][2];
    load_a(0, a0);
    load_b(0, b0);
    mm(a0, b0);
    load_b(1, b1);
    mm(a0, b1);
    bf16x8 b2[2][2];
    load_b(2, b2);
    mm(a0, b2);
    if constexpr (NPROD >= 8) {
      bf16x8 a1[2][2];
      load_a(1, a1);
      mm(a1, b0);
      mm(a1, b1);
      mm(a1, b2);
      bf16x8 a2[2][2];
      load_a(2, a2);
      mm(a2, b0);
      mm(a2, b1);
      if constexpr (NPROD >= 9) mm(a2, b2);
      return;
    }
  }
  {
    bf16x8 a1[2][2];
    load_a(1, a1);
    mm(a1, b0);
    mm(a1, b1);
  }
  bf16x8 a2[2][2];
  load_a(2, a2);
  mm(a2, b0);
}


}  // namespace

// =============================================================================================
// conv_gemm_x3_kernel
// =============================================================================================
template <int BM, int BN, int WTM, int WTN, bool SPLITK, int NPROD>
__global__ __launch_bounds__(256, 2) void conv_gemm_x3_kernel(const ConvArgs a) {
  constexpr int WGN = BN / WTN;
  constexpr int MI = WTM / 16, NJ = WTN / 16;
  static_assert((BM / WTM) * (BN / WTN) == 4, "4 waves");
  __shared__ __attribute__((aligned(16))) unsigned short As[3][BM * ROWX];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[3][BN * ROWX];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int zphase = SPLITK ? (int)blockIdx.z / a.ksplit : (int)blockIdx.z;
  const int zsplit = SPLITK ? (int)blockIdx.z - zphase * a.ksplit : 0;
  const ConvPhase ph = a.ph[zphase];
  const int M = a.NB * ph.OHp * ph.OWp;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  if (m0 >= M) return;
  const int IH = a.IH, IW = a.IW, C = a.C, N = a.N;

  constexpr int A_V = BM / 64, B_V = BN / 64;  // 8-element chunks per thread: row = tid/4 + 64*i, chunk = tid%4
  const int q8 = (tid & 3) * 8;
  const int st8 = ((tid & 3) ^ swz(tid >> 2)) * 8;   // rows tid/4 + 64*i all share this swizzle
  int r_iy0[A_V], r_ix0[A_V], r_base[A_V];
#pragma unroll
  for (int i = 0; i < A_V; ++i) {
    const int m = m0 + (tid >> 2) + 64 * i;
    if (m < M) {
      const int ox = m % ph.OWp;
      const int t2 = m / ph.OWp;
      const int oy = t2 % ph.OHp;
      const int nb = t2 / ph.OHp;
      r_iy0[i] = oy * a.S + ph.offy;
      r_ix0[i] = ox * a.S + ph.offx;
      r_base[i] = nb * IH * IW;
    } else {
      r_iy0[i] = 0;
      r_ix0[i] = 0;
      r_base[i] = -1;
    }
  }

  const int ntaps = ph.Ty * ph.Tx;
  const int cpt = C / BKX;
  const int nk = ntaps * cpt;
  int nx_ty = 0, nx_tx = 0, nx_c0 = 0;
  int kt_lo = 0, kt_hi = nk;
  if constexpr (SPLITK) {
    const int per = (nk + a.ksplit - 1) / a.ksplit;
    kt_lo = zsplit * per;
    kt_hi = min(nk, kt_lo + per);
    const int tap0 = kt_lo / cpt;
    nx_c0 = (kt_lo - tap0 * cpt) * BKX;
    nx_ty = tap0 / ph.Tx;
    nx_tx = tap0 - nx_ty * ph.Tx;
  }

  float4 ra[A_V][2];
  uint4 rb[B_V][3];
  auto load_tile = [&]() {
    const int ty = nx_ty, tx = nx_tx, c0 = nx_c0;
    const long long wt = ((long long)((ph.kh0 + ty * a.kst) * a.KW + (ph.kw0 + tx * a.kst)) * cpt + c0 / BKX) * N * BKX;
#pragma unroll
    for (int i = 0; i < A_V; ++i) {
      int iy = r_iy0[i] + ty * a.D, ix = r_ix0[i] + tx * a.D;
      if (a.reflect) {
        iy = reflect_x(iy, IH);
        ix = reflect_x(ix, IW);
      }
      float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
      if (r_base[i] >= 0 && iy >= 0 && iy < IH && ix >= 0 && ix < IW) {
        const float* px = a.x + (size_t)(r_base[i] + iy * IW + ix) * a.x_ps + c0 + q8;
        v0 = *reinterpret_cast<const float4*>(px);
        v1 = *reinterpret_cast<const float4*>(px + 4);
      }
      ra[i][0] = v0;
      ra[i][1] = v1;
    }
#pragma unroll
    for (int i = 0; i < B_V; ++i) {
      const int n = n0 + (tid >> 2) + 64 * i;
      const unsigned short* pw = a.wgt_bf + wt + (size_t)n * BKX + q8;
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (n < N) v = *reinterpret_cast<const uint4*>(pw + (size_t)p * a.wbf_plane);
        rb[i][p] = v;
      }
    }
    nx_c0 += BKX;
    if (nx_c0 >= C) {
      nx_c0 = 0;
      if (++nx_tx == ph.Tx) {
        nx_tx = 0;
        ++nx_ty;
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < A_V; ++i) {
      uint4 l0, l1, l2;
      split8(ra[i][0], ra[i][1], l0, l1, l2);
      const int o = ((tid >> 2) + 64 * i) * ROWX + st8;
      *reinterpret_cast<uint4*>(&As[0][o]) = l0;
      *reinterpret_cast<uint4*>(&As[1][o]) = l1;
      *reinterpret_cast<uint4*>(&As[2][o]) = l2;
    }
#pragma unroll
    for (int i = 0; i < B_V; ++i) {
      const int o = ((tid >> 2) + 64 * i) * ROWX + st8;
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<uint4*>(&Bs[p][o]) = rb[i][p];
    }
  };

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int l15 = lane & 15, lq = lane >> 4;
  const int fsw = (-(l15 >> 2)) & 3;   // swz of a fragment row, without the (row >> 4) & 1 bit (= i & 1 below)
  static_assert((WTM / 16) % 2 == 0 && (WTN / 16) % 2 == 0, "wave sub-tiles start at even 16-row blocks");
  auto load_a = [&](int p, bf16x8 (&f)[MI]) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
      f[i] = *reinterpret_cast<const bf16x8*>(&As[p][(wm * WTM + i * 16 + l15) * ROWX + (lq ^ fsw ^ (i & 1)) * 8]);
  };
  auto load_b = [&](int p, bf16x8 (&f)[NJ]) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      f[j] = *reinterpret_cast<const bf16x8*>(&Bs[p][(wn * WTN + j * 16 + l15) * ROWX + (lq ^ fsw ^ (j & 1)) * 8]);
  };

  if (kt_lo < kt_hi) load_tile();
  for (int kt = kt_lo; kt < kt_hi; ++kt) {
    store_tile();
    __syncthreads();
    if (kt + 1 < kt_hi) load_tile();
    limb_products<MI, NJ, NPROD>(acc, load_a, load_b);
    __syncthreads();
  }

  // operands swapped as in the fp32 kernel: lane l holds pixel row (l & 15) and output channels (l >> 4) * 4 + reg
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m0 + wm * WTM + i * 16 + l15;
    if (m >= M) continue;
    const int ox = m % ph.OWp;
    const int t2 = m / ph.OWp;
    const int oy = t2 % ph.OHp;
    const int nb = t2 / ph.OHp;
    float* yrow = (SPLITK ? a.ws + (size_t)zsplit * a.y_total : a.y) +
                  ((size_t)(nb * a.OHf + oy * a.os + ph.ophy) * a.OWf + ox * a.os + ph.ophx) * a.y_ps;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = n0 + wn * WTN + j * 16 + lq * 4;
      if (n < N) {
        float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        if constexpr (!SPLITK) {
          if (a.bias) {
            const float4 b4 = *reinterpret_cast<const float4*>(a.bias + n);
            v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
          }
          if (a.accumulate) {
            const float4 o = *reinterpret_cast<const float4*>(yrow + n);
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
          }
        }
        *reinterpret_cast<float4*>(yrow + n) = v;
      }
    }
  }
}

// =============================================================================================
// conv_gemm_x3pp_kernel: two 128x128 tiles per 512-thread block in explicit anti-phase ("ping-pong").  Waves 0-3 (group
// X) and 4-7 (group Y) own the tiles of rows m0 and m0 + 128 of the SAME 128 output channels; a SIMD hosts one wave of
// each group.  A K step of a group is a VALU / LDS-store phase (limb split of the tile fetched during its previous
// MFMA phase, LDS stores, buffer loads of the next tile) followed by an MFMA phase (24 fragment reads + 96 MFMAs); the
// block-wide barrier between phases keeps the groups exactly half a step apart, so the MFMA pipe of a SIMD always has
// one wave feeding it while the other wave's VALU / DS / VMEM instructions issue in the gaps.  The filter tile is shared:
// group X stages it (double-buffered, no arithmetic), both groups multiply with it.
//   * both operands come through buffer loads: the per-thread byte offset sits in a VGPR (an out-of-range offset returns
//     zeros -- that is how padding taps and rows beyond M are read: no branches, no selects on the data), the K-step
//     position in the uniform soffset;
//   * every wave-level load covers whole 128-byte lines: 8 lanes x 16 bytes = the 32 fp32 channels of one gathered pixel,
//     and the limb planes are stored K-step-major ([plane][tap][C/32][N][32]) so a filter tile is one contiguous 8 KB run;
//   * zero padding: a row's offset is linear in the tap (off0 + a uniform per-tap delta) and "tap inside the image" is
//     one bit of a per-row mask -- a handful of VALU instructions per K step; REFLECT keeps the index arithmetic.
// =============================================================================================
template <bool SPLITK, int NPROD, bool REFLECT, bool TIMING = false>
__global__ __launch_bounds__(512, 1) void conv_gemm_x3pp_kernel(const ConvArgs a) {
  constexpr int BM = 128, BN = 128, WTM = 64, WTN = 64;
  constexpr int NL = NPROD == 1 ? 1 : 3;   // limb planes (1: operands rounded to bf16, the compute_dtype bf16 arithmetic)
  __shared__ __attribute__((aligned(16))) unsigned short AsG[2][NL][BM * ROWX];   // per group
  __shared__ __attribute__((aligned(16))) unsigned short BsG[3][NL][BN * ROWX];   // per K step mod 3

  const int tid = threadIdx.x & 255, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);   // uniform
  const int wm = wave >> 1, wn = wave & 1;
  unsigned short (&As)[NL][BM * ROWX] = AsG[grp];
  const int zphase = SPLITK ? (int)blockIdx.z / a.ksplit : (int)blockIdx.z;
  const int zsplit = SPLITK ? (int)blockIdx.z - zphase * a.ksplit : 0;
  const ConvPhase ph = a.ph[zphase];
  const int M = a.NB * ph.OHp * ph.OWp;
  const int m0 = (blockIdx.x * 2 + grp) * BM, n0 = blockIdx.y * BN;   // m0 >= M: the group only keeps the barriers
  const int IH = a.IH, IW = a.IW, C = a.C, N = a.N;

  // A: row j*32 + tid/8, floats 4*(tid%8) .. +3 of the K step
  constexpr int A_R = BM / 32;
  const int c4 = tid & 7, ar = tid >> 3;
  const int ast = ((c4 >> 1) ^ swz(ar)) * 8 + (c4 & 1) * 4;   // swz(j*32 + ar) == swz(ar)
  int r_iy0[REFLECT ? A_R : 1], r_ix0[REFLECT ? A_R : 1], r_base[REFLECT ? A_R : 1];
  unsigned r_off0[A_R], r_mask[A_R];
#pragma unroll
  for (int j = 0; j < A_R; ++j) {
    const int m = m0 + j * 32 + ar;
    int iy0 = 0, ix0 = 0, base = -1;
    r_mask[j] = 0u;
    if (m < M) {
      const int ox = m % ph.OWp;
      const int t2 = m / ph.OWp;
      const int oy = t2 % ph.OHp;
      const int nb = t2 / ph.OHp;
      iy0 = oy * a.S + ph.offy;
      ix0 = ox * a.S + ph.offx;
      base = nb * IH * IW;
      if constexpr (!REFLECT) {
        for (int ty = 0; ty < ph.Ty; ++ty)
          for (int tx = 0; tx < ph.Tx; ++tx) {
            const int iy = iy0 + ty * a.D, ix = ix0 + tx * a.D;
            if ((unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW) r_mask[j] |= 1u << (ty * ph.Tx + tx);
          }
      }
    }
    r_off0[j] = (unsigned)(((base + iy0 * IW + ix0) * a.x_ps + c4 * 4) * 4);
    if constexpr (REFLECT) {
      r_iy0[j] = iy0;
      r_ix0[j] = ix0;
      r_base[j] = base;
    }
  }
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wres =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.wgt_bf), 0, a.wbf_bytes, 0x00020000);
  // B (group X) goes global -> LDS directly (buffer_load ... lds, no registers, no LDS store instructions): one wave
  // instruction fills 16 consecutive 64-byte rows of a plane (lane l -> LDS bytes 16*l .. +15 behind M0), so the XOR
  // chunk swizzle is applied on the GLOBAL side: lane l fetches logical chunk (l & 3) ^ swz(row) of row l / 4.
  constexpr int B_V = BN / 64;
  unsigned woff[B_V];
  {
    const int lsw = (((-((lane >> 4) & 3)) & 3) ^ (wave & 1));   // swz(i*64 + wave*16 + lane/4)
#pragma unroll
    for (int i = 0; i < B_V; ++i)
      woff[i] = (unsigned)(((size_t)min(n0 + i * 64 + wave * 16 + (lane >> 2), N - 1) * BKX + ((lane & 3) ^ lsw) * 8) * 2);
  }
  const int plane_bytes = (int)(a.wbf_plane * 2);

  const int ntaps = ph.Ty * ph.Tx;
  const int cpt = C / BKX;
  const int nk = ntaps * cpt;
  int nx_ty = 0, nx_tx = 0, nx_cc = 0;
  int kt_lo = 0, kt_hi = nk;
  if constexpr (SPLITK) {
    const int per = (nk + a.ksplit - 1) / a.ksplit;
    kt_lo = zsplit * per;
    kt_hi = min(nk, kt_lo + per);
    const int tap0 = kt_lo / cpt;
    nx_cc = kt_lo - tap0 * cpt;
    nx_ty = tap0 / ph.Tx;
    nx_tx = tap0 - nx_ty * ph.Tx;
  }
  const int wtile = N * BKX * 2;   // bytes of one (tap, channel chunk) filter tile of a plane

  // The gathered operand has two register staging sets: tile k+1 is requested BEFORE tile k is split, so every load has a
  // whole VALU + MFMA phase to land and its address processing overlaps the split arithmetic.
  f32x4 ra0[A_R], ra1[A_R];
  int kx_ty = nx_ty, kx_tx = nx_tx, kx_cc = nx_cc;   // position of the next FILTER tile (steps separately from A's)
  auto step = [&](int& ty, int& tx, int& cc) __attribute__((always_inline)) {
    cc += 1;
    const bool wc = cc == cpt;
    cc = wc ? 0 : cc;
    tx += wc ? 1 : 0;
    const bool wx = tx == ph.Tx;
    tx = wx ? 0 : tx;
    ty += wx ? 1 : 0;
  };
  auto load_a_tile = [&](f32x4 (&ra)[A_R]) __attribute__((always_inline)) {
    const int ty = nx_ty, tx = nx_tx, cc = nx_cc;
    const int tapbit = ty * ph.Tx + tx;
    const unsigned delta = (unsigned)(((ty * IW + tx) * a.D * a.x_ps) * 4);
#pragma unroll
    for (int j = 0; j < A_R; ++j) {
      unsigned off;
      if constexpr (REFLECT) {
        const int iy = reflect_x(r_iy0[j] + ty * a.D, IH), ix = reflect_x(r_ix0[j] + tx * a.D, IW);
        const bool ok = r_base[j] >= 0 && (unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW;
        off = ok ? (unsigned)(((r_base[j] + iy * IW + ix) * a.x_ps + c4 * 4) * 4) : 0x80000000u;
      } else {
        off = ((r_mask[j] >> tapbit) & 1u) ? r_off0[j] + delta : 0x80000000u;
      }
      ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, off, cc * (BKX * 4), 0));
    }
    step(nx_ty, nx_tx, nx_cc);
  };
  auto dma_b_tile = [&](int par) __attribute__((always_inline)) {   // filter tile -> BsG[par], 6 x 1 KB per wave
    const int wsoff = (((ph.kh0 + kx_ty * a.kst) * a.KW + (ph.kw0 + kx_tx * a.kst)) * cpt + kx_cc) * wtile;
#pragma unroll
    for (int p = 0; p < NL; ++p)
#pragma unroll
      for (int i = 0; i < B_V; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wres, (__attribute__((address_space(3))) void*)&BsG[par][p][(i * 64 + wave * 16) * ROWX],
                                                 16, woff[i], wsoff + p * plane_bytes, 0, 0);
    step(kx_ty, kx_tx, kx_cc);
  };
  auto store_a_tile = [&](const f32x4 (&ra)[A_R]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < A_R; ++j) {
      const int o = (j * 32 + ar) * ROWX + ast;
      if constexpr (NL == 1) {
        *reinterpret_cast<u32x2*>(&As[0][o]) = (u32x2){pack2(ra[j][0], ra[j][1]), pack2(ra[j][2], ra[j][3])};
      } else {
        unsigned p0a, p1a, p2a, p0b, p1b, p2b;
        split_pair(ra[j][0], ra[j][1], p0a, p1a, p2a);
        split_pair(ra[j][2], ra[j][3], p0b, p1b, p2b);
        *reinterpret_cast<u32x2*>(&As[0][o]) = (u32x2){p0a, p0b};
        *reinterpret_cast<u32x2*>(&As[NL > 1 ? 1 : 0][o]) = (u32x2){p1a, p1b};
        *reinterpret_cast<u32x2*>(&As[NL > 1 ? 2 : 0][o]) = (u32x2){p2a, p2b};
      }
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int l31 = lane & 31, lh = lane >> 5;
  const int fsw = swz(l31);   // == swz(row) for every fragment row: the block offsets are multiples of 32
  const unsigned short* Bcur = &BsG[0][0][0];
  auto load_a = [&](int p, bf16x8 (&f)[2][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        f[i][kk] = *reinterpret_cast<const bf16x8*>(&As[p][(wm * WTM + i * 32 + l31) * ROWX + ((kk * 2 + lh) ^ fsw) * 8]);
  };
  auto load_b = [&](int p, bf16x8 (&f)[2][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        f[j][kk] = *reinterpret_cast<const bf16x8*>(Bcur + p * (BN * ROWX) + (wn * WTN + j * 32 + l31) * ROWX + ((kk * 2 + lh) ^ fsw) * 8);
  };

  // barrier slots: X stores in even slots and multiplies in odd ones, Y the other way round (one idle slot each);
  // filter tile k lives in BsG[k & 1]: written by X in slot 2k, read by X in slot 2k+1 and by Y in slot 2k+2
  const int nt = kt_hi - kt_lo;
  const bool tim = TIMING && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0 && wave == 0;
  long long* tbuf = reinterpret_cast<long long*>(a.ws) + grp * 64;
  // Filter tile k lives in BsG[k % 3].  X requests tile k+2 at the END of its MFMA phase k (slot 2k+1: the buffer's
  // previous tenant, tile k-1, was last read by Y in slot 2k; no LDS read of this phase follows the request, so the
  // compiler's "DMA in flight may alias this LDS read" wait only appears in MFMA phase k+1, ~2 phases later), and it is
  // read by X in slot 2k+5 and by Y in slot 2k+6.  X's VMEM stream completes in order: ... A(k+1) x4, DMA B(k+2) x6,
  // A(k+2) x4 ...  The two groups run separate instantiations of the loop and its steady state has no run-time
  // conditions (the compiler's outstanding-load bookkeeping merges control-flow paths pessimistically).
  // Raw barrier, no fences: a fence would drain the DMA in flight before every barrier; what the barrier orders here is
  // LDS traffic only.
  auto lds_barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  auto run = [&](auto GRP) __attribute__((always_inline)) {
    constexpr bool X = decltype(GRP)::value == 0;
    auto valu_phase = [&](auto HASNEXT, int k, const f32x4 (&ra)[A_R], f32x4 (&ran)[A_R]) __attribute__((always_inline)) {
      if (tim && k >= 8 && k < 12) tbuf[(k - 8) * 8 + 0] = __builtin_amdgcn_s_memtime();
      if constexpr (decltype(HASNEXT)::value) {   // (a run-time test here would make the compiler wait for the NEW loads)
        load_a_tile(ran);
        __builtin_amdgcn_sched_barrier(0);   // the requests go out BEFORE the split, not wherever the scheduler likes
      }
      store_a_tile(ra);
      if (tim && k >= 8 && k < 12) tbuf[(k - 8) * 8 + 1] = __builtin_amdgcn_s_memtime();
    };
    auto mfma_phase = [&](auto DMA, int k, int bpar, int bpar_dma) __attribute__((always_inline)) {
      lds_barrier();
      if (tim && k >= 8 && k < 12) tbuf[(k - 8) * 8 + 2] = __builtin_amdgcn_s_memtime();
      Bcur = &BsG[bpar][0][0];
      limb_products32<NPROD>(acc, load_a, load_b);
      if constexpr (X && decltype(DMA)::value) {
        __builtin_amdgcn_sched_barrier(0);
        dma_b_tile(bpar_dma);
      }
      if (tim && k >= 8 && k < 12) tbuf[(k - 8) * 8 + 3] = __builtin_amdgcn_s_memtime();
      lds_barrier();
      if (tim && k >= 8 && k < 12) tbuf[(k - 8) * 8 + 4] = __builtin_amdgcn_s_memtime();
    };
    constexpr std::true_type yes{};
    if (nt > 0) {
      if constexpr (X) {
        dma_b_tile(0);
        if (nt > 1) dma_b_tile(1);
      }
      load_a_tile(ra0);
    }
    if constexpr (!X) lds_barrier();
    int bp = 0, k = 0;   // bp = k % 3
    for (; k + 3 < nt; k += 2) {   // steady state: steps k, k+1 have successors and filter tiles k+2, k+3 exist
      const int bp1 = bp == 2 ? 0 : bp + 1, bp2 = bp1 == 2 ? 0 : bp1 + 1;
      valu_phase(yes, k, ra0, ra1);
      mfma_phase(yes, k, bp, bp2);
      valu_phase(yes, k + 1, ra1, ra0);
      mfma_phase(yes, k + 1, bp1, bp);
      bp = bp2;
    }
    // up to 3 steps left (k is even here); run-time conditions are fine now.  Straight-line on purpose: a loop with the
    // staging sets swapped per parity makes the optimiser select between POINTERS to them and keeps both in scratch.
    auto tail_step = [&](int kk, int bpar, int bpar_dma, const f32x4 (&ra)[A_R], f32x4 (&ran)[A_R]) __attribute__((always_inline)) {
      if (kk + 1 < nt) load_a_tile(ran);
      store_a_tile(ra);
      lds_barrier();
      Bcur = &BsG[bpar][0][0];
      limb_products32<NPROD>(acc, load_a, load_b);
      if constexpr (X)
        if (kk + 2 < nt) dma_b_tile(bpar_dma);
      lds_barrier();
    };
    if (k < nt) {
      const int bp1 = bp == 2 ? 0 : bp + 1, bp2 = bp1 == 2 ? 0 : bp1 + 1;
      tail_step(k, bp, bp2, ra0, ra1);
      if (k + 1 < nt) {
        tail_step(k + 1, bp1, bp, ra1, ra0);
        if (k + 2 < nt) tail_step(k + 2, bp2, bp1, ra0, ra1);
      }
    }
    if constexpr (X) lds_barrier();
  };
  if (grp == 0) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});

  // D^T layout of the 32x32 accumulator: lane l holds pixel row l & 31 and, in registers 4g .. 4g+3, the four consecutive
  // output channels 8g + 4 (l >> 5) .. +3 of its 32-channel block: one 16-byte store per register quad
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wm * WTM + i * 32 + l31;
    if (m >= M) continue;
    const int ox = m % ph.OWp;
    const int t2 = m / ph.OWp;
    const int oy = t2 % ph.OHp;
    const int nb = t2 / ph.OHp;
    float* yrow = (SPLITK ? a.ws + (size_t)zsplit * a.y_total : a.y) +
                  ((size_t)(nb * a.OHf + oy * a.os + ph.ophy) * a.OWf + ox * a.os + ph.ophx) * a.y_ps;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * WTN + j * 32 + g * 8 + lh * 4;
        if (n < N) {
          float4 v = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
          if constexpr (!SPLITK) {
            if (a.bias) {
              const float4 b4 = *reinterpret_cast<const float4*>(a.bias + n);
              v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
            }
            if (a.accumulate) {
              const float4 o = *reinterpret_cast<const float4*>(yrow + n);
              v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
          }
          *reinterpret_cast<float4*>(yrow + n) = v;
        }
      }
  }
}

bool conv_x3_eligible(const ConvArgs& a) {
  if (a.limbs == 1)   // one plane (bf16-rounded operands): only the ping-pong kernel has that form
    return a.wgt_bf && a.x3_tile == 1 && a.C % BKX == 0 && a.N >= 128 && a.N % 4 == 0 && a.x_ps % 4 == 0 && a.y_ps % 4 == 0 &&
           a.x_bytes && a.wbf_bytes && (((uintptr_t)a.x & 15) == 0) && (((uintptr_t)a.y & 15) == 0) &&
           (((uintptr_t)a.wgt_bf & 15) == 0) && (!a.bias || (((uintptr_t)a.bias & 15) == 0));
  return a.wgt_bf && a.limbs == 3 && a.C % BKX == 0 && a.N > 16 && a.N % 4 == 0 && a.x_ps % 4 == 0 && a.y_ps % 4 == 0 &&
         a.wbf_plane % 8 == 0 && (((uintptr_t)a.x & 15) == 0) &&
         (((uintptr_t)a.y & 15) == 0) && (((uintptr_t)a.wgt_bf & 15) == 0) && (!a.bias || (((uintptr_t)a.bias & 15) == 0));
}

static int x3_nprod() {
  static const int n = [] {
    const char* e = dbg_env("EG_X3_PRODUCTS");
    const int v = e ? atoi(e) : 6;
    return (v == 8 || v == 9) ? v : 6;
  }();
  return n;
}

template <int BM, int BN, int WTM, int WTN, int NPROD>
static void conv_x3_launch(const ConvArgs& a, int maxM, hipStream_t s) {
  if (a.ksplit > 1) {
    dim3 grid((maxM + BM - 1) / BM, (a.N + BN - 1) / BN, a.nphase * a.ksplit);
    hipLaunchKernelGGL((conv_gemm_x3_kernel<BM, BN, WTM, WTN, true, NPROD>), grid, dim3(256), 0, s, a);
  } else {
    dim3 grid((maxM + BM - 1) / BM, (a.N + BN - 1) / BN, a.nphase);
    hipLaunchKernelGGL((conv_gemm_x3_kernel<BM, BN, WTM, WTN, false, NPROD>), grid, dim3(256), 0, s, a);
  }
}

// does launch_conv_gemm_x3 hand the launch to the halo kernel (k_gemm_x3h.hip)?
bool conv_x3_takes_halo(const ConvArgs& a, bool big_tile) {
  return (a.limbs == 1 || (big_tile && x3_nprod() == 6)) && conv_x3h_eligible(a);
}

const char* launch_conv_gemm_x3(const ConvArgs& a, int maxM, bool big_tile, hipStream_t s, int* caps) {
  const int np = x3_nprod();
  static const int pipe = dbg_env("EG_X3_PIPE") ? atoi(dbg_env("EG_X3_PIPE")) : 2;
  if (conv_x3_takes_halo(a, big_tile)) {
    if (caps && a.ksplit <= 1) *caps |= CONV_CAP_EPI_ACT | CONV_CAP_Y2;   // the un-split halo kernel's epilogue (k_gemm_x3h.hip)
    if (caps && conv_x3h_pool_ok(a)) *caps |= CONV_CAP_POOL;
    return launch_conv_gemm_x3h(a, s);
  }
  if (a.limbs == 1) {
    const int zs = a.ksplit > 1 ? a.ksplit : 1;
    dim3 grid((maxM + 255) / 256, (a.N + 127) / 128, a.nphase * zs);
    if (a.reflect) {
      if (a.ksplit > 1) hipLaunchKernelGGL((conv_gemm_x3pp_kernel<true, 1, true>), grid, dim3(512), 0, s, a);
      else hipLaunchKernelGGL((conv_gemm_x3pp_kernel<false, 1, true>), grid, dim3(512), 0, s, a);
    } else {
      if (a.ksplit > 1) hipLaunchKernelGGL((conv_gemm_x3pp_kernel<true, 1, false>), grid, dim3(512), 0, s, a);
      else hipLaunchKernelGGL((conv_gemm_x3pp_kernel<false, 1, false>), grid, dim3(512), 0, s, a);
    }
    return a.ksplit > 1 ? "conv_gemm_bf16pp<2x128,128,splitk>" : "conv_gemm_bf16pp<2x128,128>";
  }
  if (big_tile && pipe == 2 && np == 6 && a.x_bytes && a.wbf_bytes) {
    const int zs = a.ksplit > 1 ? a.ksplit : 1;
    dim3 grid((maxM + 255) / 256, (a.N + 127) / 128, a.nphase * zs);
    static const bool timing = dbg_env("EG_X3_TIMING") != nullptr;
    if (a.reflect) {
      if (a.ksplit > 1) hipLaunchKernelGGL((conv_gemm_x3pp_kernel<true, 6, true>), grid, dim3(512), 0, s, a);
      else hipLaunchKernelGGL((conv_gemm_x3pp_kernel<false, 6, true>), grid, dim3(512), 0, s, a);
    } else {
      if (a.ksplit > 1) hipLaunchKernelGGL((conv_gemm_x3pp_kernel<true, 6, false>), grid, dim3(512), 0, s, a);
      else if (timing && a.ws) hipLaunchKernelGGL((conv_gemm_x3pp_kernel<false, 6, false, true>), grid, dim3(512), 0, s, a);
      else hipLaunchKernelGGL((conv_gemm_x3pp_kernel<false, 6, false>), grid, dim3(512), 0, s, a);
    }
    return a.ksplit > 1 ? "conv_gemm_x3pp<2x128,128,splitk>" : "conv_gemm_x3pp<2x128,128>";
  }
  if (big_tile) {
    if (np == 9) conv_x3_launch<128, 128, 64, 64, 9>(a, maxM, s);
    else if (np == 8) conv_x3_launch<128, 128, 64, 64, 8>(a, maxM, s);
    else conv_x3_launch<128, 128, 64, 64, 6>(a, maxM, s);
    return a.ksplit > 1 ? "conv_gemm_x3<128,128,splitk>" : "conv_gemm_x3<128,128>";
  }
  if (np == 9) conv_x3_launch<64, 64, 32, 32, 9>(a, maxM, s);
  else if (np == 8) conv_x3_launch<64, 64, 32, 32, 8>(a, maxM, s);
  else conv_x3_launch<64, 64, 32, 32, 6>(a, maxM, s);
  return a.ksplit > 1 ? "conv_gemm_x3<64,64,splitk>" : "conv_gemm_x3<64,64>";
}

// =============================================================================================
// wgrad_x3pp_kernel: dW[tap][c][n] = sum_m X[pix_in(m,tap)][c] * dY[m][n] with both operands split into limbs in the
// kernel.  The contraction runs over pixels, the SLOW index of both NHWC operands, so the transpose happens in the
// register -> LDS store: a thread owns 4 channels x 2 consecutive pixels of a K step (32 pixels), splits the two
// pixels of each channel as one packed pair (split_pair: limb dword = {pixel 2q, pixel 2q+1}) and writes 3 x 4 dwords
// to the channel rows of the three limb planes (lane / store order as in k_gemm_bf16.hip: every ds_write_b32 hits 32
// distinct banks; global loads stay whole 128-byte lines: 8 lanes x 16 bytes of one pixel).
// Ping-pong as in conv_gemm_x3pp_kernel: 512 threads = groups X and Y, each owning one (tap, 128-channel block) unit of
// the gathered operand with its own LDS image; the dY tile (same pixels, same 128 output channels for both units) is
// shared and double-buffered by K-step parity, each group splitting one half of its rows -- X the half of step k in its
// VALU phase k (slot 2k), Y the half of step k+1 in its VALU phase k (slot 2k+1), i.e. always at least one barrier
// before the first reader (X, slot 2k+1 / 2k+3) and after the last reader of the buffer's previous tenant.
// =============================================================================================
template <int NPROD>
__global__ __launch_bounds__(512, 1) void wgrad_x3pp_kernel(const WgradArgs a, float* __restrict__ out, int dense_out) {
  constexpr int BM = 128, BN = 128, WTM = 64, WTN = 64;
  constexpr int NL = NPROD == 1 ? 1 : 3;
  __shared__ __attribute__((aligned(16))) unsigned short AsG[2][NL][BM * ROWX];   // per group: [c][pixel]
  __shared__ __attribute__((aligned(16))) unsigned short BsG[2][NL][BN * ROWX];   // per K-step parity: [n][pixel]
  // per group, per 256-pixel block parity (8 K steps), per row half of the unit: gathered-pixel byte offsets.  The two
  // halves (rows 0-63 / 64-127 of the A tile) belong to the same tap when C >= 128 and to two consecutive taps when
  // C == 64 (a unit is then taps 2u, 2u+1 x all 64 channels: no half-empty tiles for the 64-channel layers)
  __shared__ unsigned poffG[2][2][2][256];

  const int tid = threadIdx.x & 255, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
  const int wm = wave >> 1, wn = wave & 1;
  unsigned short (&As)[NL][BM * ROWX] = AsG[grp];
  unsigned (&poff)[2][2][256] = poffG[grp];
  const int C = a.C, N = a.N, IH = a.IH, IW = a.IW;
  const int T = a.KH * a.KW;
  const bool two_taps = C == 64;
  const int ctiles = (C + BM - 1) / BM, ntn = (N + BN - 1) / BN, units = two_taps ? (T + 1) / 2 : T * ctiles;
  // XCD-aware block map: the blocks of one pixel slab (same split index, all units and output tiles) read the same x
  // and dY; the dispatcher places block b on XCD b % 8, so without a remap a slab is fetched into several XCDs' L2s
  // (measured 4.5x the algorithmic HBM reads).  xcd_remap gives every XCD a contiguous range of virtual ids.
  const unsigned vblk = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y, a.dbg & 16);
  const int bx = __builtin_amdgcn_readfirstlane(vblk % gridDim.x), by = __builtin_amdgcn_readfirstlane(vblk / gridDim.x);
  const int unit = (bx / ntn) * 2 + grp;
  const bool uvalid = unit < units;                      // an idle group still splits its half of dY and keeps the barriers
  const int tap = uvalid ? (two_taps ? 2 * unit : unit / ctiles) : 0;
  const int tap1 = two_taps ? tap + 1 : tap;             // tap of the tile's upper row half (may be == T: an empty half)
  const int c0 = (uvalid && !two_taps ? unit % ctiles : 0) * BM, n0 = (bx % ntn) * BN;
  const int khh[2] = {tap / a.KW, tap1 / a.KW};
  const int kwh[2] = {tap - khh[0] * a.KW, tap1 - khh[1] * a.KW};
  const int M = a.NB * a.OH * a.OW;
  int chunk = (M + a.nsplit - 1) / a.nsplit;
  chunk = (chunk + BKX - 1) / BKX * BKX;
  const int p_lo = by * chunk;
  const int p_hi = min(M, p_lo + chunk);
  const int nt = p_hi > p_lo ? (p_hi - p_lo + BKX - 1) / BKX : 0;

  // unit = (pixel pair mp of the K step, channel quad cq): a half wave is 8 consecutive quads (one 128-byte line of
  // a pixel) x 4 pixel pairs of ONE 8-pixel chunk
  const int cq8 = tid & 7, mpl = (tid >> 3) & 3, rest = tid >> 5;
  const int kch = rest & 3;
  const int mp = kch * 4 + mpl;
  const int cqb = (rest >> 2) * 8 + cq8;          // A: quads cqb, cqb + 16; B: quad cqb + 16 * grp
  const int ex = cq8 & 1;                          // odd quads store their 4 rows in the order 1,0,3,2
  const unsigned exmask = 0u - (unsigned)ex;
  const int st_col = (kch ^ (((-(cq8 & 3)) & 3) ^ (cq8 >> 2))) * 4 + mpl;   // dword column, swizzled (swz(4*cq))

  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, a.dy_bytes, 0x00020000);
  unsigned a_coff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = two_taps ? cqb * 4 : c0 + (cqb + 16 * i) * 4;
    const bool ok = two_taps ? (i == 0 ? tap : tap1) < T : c < C;
    a_coff[i] = (uvalid && ok) ? (unsigned)(c * 4) : 0x40000000u;   // (extents are < 1 GB: launch_wgrad)
  }
  const int bn = n0 + (cqb + 16 * grp) * 4;
  const unsigned b_noff = bn < N ? (unsigned)(bn * 4) : 0x40000000u;

  auto decode_block = [&](int blk) __attribute__((always_inline)) {
    const int m = p_lo + blk * 256 + tid;
    unsigned off[2] = {0x80000000u, 0x80000000u};
    if (m < p_hi) {
      const int ox = m % a.OW;
      const int t2 = m / a.OW;
      const int oy = t2 % a.OH;
      const int nb = t2 / a.OH;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int iy = oy * a.S + khh[h] - a.pad, ix = ox * a.S + kwh[h] - a.pad;
        if (a.reflect) {
          iy = reflect_x(iy, IH);
          ix = reflect_x(ix, IW);
        }
        if (iy >= 0 && iy < IH && ix >= 0 && ix < IW) off[h] = (unsigned)(((nb * IH + iy) * IW + ix) * a.x_ps * 4);
      }
    }
    poff[blk & 1][0][tid] = off[0];
    poff[blk & 1][1][tid] = off[1];
  };
  f32x4 ra0[2][2], ra1[2][2], rb0[2], rb1[2];
  auto load_a = [&](int t, f32x4 (&ra)[2][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned* po = &poff[(t >> 3) & 1][i][(t & 7) * 32 + 2 * mp];
      const unsigned o0 = po[0], o1 = po[1];
      ra[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, o0 + a_coff[i], 0, 0));
      ra[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, o1 + a_coff[i], 0, 0));
    }
  };
  auto load_b = [&](int t, f32x4 (&rb)[2]) __attribute__((always_inline)) {
    const int m = p_lo + t * BKX + 2 * mp;
    const unsigned r0 = m < p_hi ? (unsigned)(m * a.dy_ps * 4) : 0x80000000u;
    const unsigned r1 = m + 1 < p_hi ? (unsigned)((m + 1) * a.dy_ps * 4) : 0x80000000u;
    rb[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(yres, r0 + b_noff, 0, 0));
    rb[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(yres, r1 + b_noff, 0, 0));
  };
  // 4 channels x 2 pixels -> 3 planes x 4 packed dwords into rows 4*cq + (j ^ ex), j = 0..3
  auto store_unit = [&](unsigned short (&P)[NL][BM * ROWX], int cq, const f32x4& lo, const f32x4& hi) __attribute__((always_inline)) {
    unsigned l0[4], l1[4], l2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (NL == 1) l0[j] = pack2(lo[j], hi[j]);
      else split_pair(lo[j], hi[j], l0[j], l1[j], l2[j]);
    }
    const int o = (4 * cq) * (ROWX / 2) + st_col;
    auto put = [&](unsigned short* plane, const unsigned (&p)[4]) __attribute__((always_inline)) {
      // instruction j writes channel j ^ ex to row 4*cq + (j ^ ex): branch-free swap of the packed pairs
      const unsigned s01 = (p[0] ^ p[1]) & exmask, s23 = (p[2] ^ p[3]) & exmask;
      unsigned* q = reinterpret_cast<unsigned*>(plane) + o;
      q[(0 ^ ex) * (ROWX / 2)] = p[0] ^ s01;
      q[(1 ^ ex) * (ROWX / 2)] = p[1] ^ s01;
      q[(2 ^ ex) * (ROWX / 2)] = p[2] ^ s23;
      q[(3 ^ ex) * (ROWX / 2)] = p[3] ^ s23;
    };
    put(P[0], l0);
    if constexpr (NL > 1) {
      put(P[NL > 1 ? 1 : 0], l1);
      put(P[NL > 1 ? 2 : 0], l2);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int l31 = lane & 31, lh = lane >> 5;
  const int fsw = swz(l31);
  const unsigned short* Bcur = &BsG[0][0][0];
  auto frag_a = [&](int p, bf16x8 (&f)[2][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        f[i][kk] = *reinterpret_cast<const bf16x8*>(&As[p][(wm * WTM + i * 32 + l31) * ROWX + ((kk * 2 + lh) ^ fsw) * 8]);
  };
  auto frag_b = [&](int p, bf16x8 (&f)[2][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        f[j][kk] = *reinterpret_cast<const bf16x8*>(Bcur + p * (BN * ROWX) + (wn * WTN + j * 32 + l31) * ROWX + ((kk * 2 + lh) ^ fsw) * 8);
  };
  auto lds_barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  const int nblk = (nt + 7) / 8;

  auto run = [&](auto GRP) __attribute__((always_inline)) {
    constexpr int LEAD = decltype(GRP)::value;   // X stages its dY half of step k in VALU phase k, Y that of step k + 1
    // one VALU phase: requests for the next tiles first, then split + store; RT = run-time guards (tail only)
    auto valu_phase = [&](auto RT, int k, const f32x4 (&ra)[2][2], f32x4 (&ran)[2][2], const f32x4 (&rb)[2], f32x4 (&rbn)[2])
                          __attribute__((always_inline)) {
      constexpr bool rt = decltype(RT)::value;
      if (!rt || k + 1 < nt) load_a(k + 1, ran);
      if (!rt || k + 1 + LEAD < nt) load_b(k + 1 + LEAD, rbn);
      if constexpr (!rt) __builtin_amdgcn_sched_barrier(0);
      if ((k & 7) == 2 && (k >> 3) + 1 < nblk) decode_block((k >> 3) + 1);
#pragma unroll
      for (int i = 0; i < 2; ++i) store_unit(As, cqb + 16 * i, ra[i][0], ra[i][1]);
      if (!rt || k + LEAD < nt) store_unit(BsG[(k + LEAD) & 1], cqb + 16 * LEAD, rb[0], rb[1]);
    };
    auto mfma_phase = [&](int k) __attribute__((always_inline)) {
      lds_barrier();
      Bcur = &BsG[k & 1][0][0];
      limb_products32<NPROD>(acc, frag_a, frag_b);
      lds_barrier();
    };
    constexpr std::true_type guarded{};
    constexpr std::false_type steady{};
    // prologue: X holds A(0), B(0); Y holds A(0), B(1) and has already stored its half of B(0) (in its idle slot 0)
    if (nt > 0) {
      decode_block(0);
      lds_barrier();           // both groups: the offset tables are per group, but the barrier is block-wide
      load_a(0, ra0);
      if constexpr (LEAD == 0) {
        load_b(0, rb0);
      } else {
        load_b(0, rb0);
        if (nt > 1) load_b(1, rb1);
        store_unit(BsG[0], cqb + 16, rb0[0], rb0[1]);
        lds_barrier();         // Y's idle slot 0
      }
    }
    // staging sets: A(k) in ra[k & 1]; X: B(k) in rb[k & 1]; Y: B(k + 1) in rb[(k + 1) & 1]
    int k = 0;
    for (; k + 3 < nt; k += 2) {
      if constexpr (LEAD == 0) {
        valu_phase(steady, k, ra0, ra1, rb0, rb1);
        mfma_phase(k);
        valu_phase(steady, k + 1, ra1, ra0, rb1, rb0);
        mfma_phase(k + 1);
      } else {
        valu_phase(steady, k, ra0, ra1, rb1, rb0);
        mfma_phase(k);
        valu_phase(steady, k + 1, ra1, ra0, rb0, rb1);
        mfma_phase(k + 1);
      }
    }
    if (k < nt) {   // up to 3 steps left (k even), straight-line with run-time guards
      if constexpr (LEAD == 0) valu_phase(guarded, k, ra0, ra1, rb0, rb1);
      else valu_phase(guarded, k, ra0, ra1, rb1, rb0);
      mfma_phase(k);
      if (k + 1 < nt) {
        if constexpr (LEAD == 0) valu_phase(guarded, k + 1, ra1, ra0, rb1, rb0);
        else valu_phase(guarded, k + 1, ra1, ra0, rb0, rb1);
        mfma_phase(k + 1);
        if (k + 2 < nt) {
          if constexpr (LEAD == 0) valu_phase(guarded, k + 2, ra0, ra1, rb0, rb1);
          else valu_phase(guarded, k + 2, ra0, ra1, rb1, rb0);
          mfma_phase(k + 2);
        }
      }
    }
    if constexpr (LEAD == 0)
      if (nt > 0) lds_barrier();   // X's idle last slot
  };
  if (grp == 0) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});

  // lane l holds filter row c = .. + (l & 31) and, in registers 4g .. 4g+3, output channels 8g + 4 (l >> 5) .. +3
  if (!uvalid) return;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    // the wave's 64 rows are one half of the tile: in the two-tap form that half is a tap of its own
    const int c = two_taps ? i * 32 + l31 : c0 + wm * WTM + i * 32 + l31;
    const int otap = two_taps ? tap + wm : tap;
    if (c >= C || otap >= T) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * WTN + j * 32 + g * 8 + lh * 4;
        if (n >= N) continue;
        float* q = dense_out ? out + (((size_t)by * T + otap) * C + c) * N + n
                             : out + (size_t)otap * a.w_ts + (size_t)c * a.ldw + n;
        float4 v = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
        if (!dense_out && a.accumulate) {
          const float4 o = *reinterpret_cast<const float4*>(q);
          v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        *reinterpret_cast<float4*>(q) = v;
      }
  }
}

template <int NPROD>
__global__ __launch_bounds__(512, 1) void wgrad_x3t_kernel(const WgradArgs a, float* __restrict__ out, int dense_out) {
  constexpr int BM = 128, BN = 128, WTM = 64, WTN = 64;
  constexpr int NL = NPROD == 1 ? 1 : 3;
  static_assert(NPROD == 6, "the transpose-read form exists for the three-limb mode");
  __shared__ __attribute__((aligned(16))) unsigned short AsG[2][NL][BKX * BM];   // per group: [pixel][c] (pixel-major, 256-byte rows)
  __shared__ __attribute__((aligned(16))) unsigned short BsG[2][NL][BKX * BN];   // per K-step parity: [pixel][n]
  // per group, per 256-pixel block parity (8 K steps), per row half of the unit: gathered-pixel byte offsets.  The two
  // halves (rows 0-63 / 64-127 of the A tile) belong to the same tap when C >= 128 and to two consecutive taps when
  // C == 64 (a unit is then taps 2u, 2u+1 x all 64 channels: no half-empty tiles for the 64-channel layers)
  __shared__ unsigned poffG[2][2][2][256];

  const int tid = threadIdx.x & 255, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
  const int wm = wave >> 1, wn = wave & 1;
#ifdef EG_EXPERIMENTS
  const int dbg = a.dbg;     // limiter probes: 1 no limb split, 2 no fragment reads, 4 no LDS stores, 8 no global loads
#else
  constexpr int dbg = 0;
#endif
  unsigned short (&As)[NL][BKX * BM] = AsG[grp];
  unsigned (&poff)[2][2][256] = poffG[grp];
  const int C = a.C, N = a.N, IH = a.IH, IW = a.IW;
  const int T = a.KH * a.KW;
  const bool two_taps = C == 64;
  const int ctiles = (C + BM - 1) / BM, ntn = (N + BN - 1) / BN, units = two_taps ? (T + 1) / 2 : T * ctiles;
  // XCD-aware block map: the blocks of one pixel slab (same split index, all units and output tiles) read the same x
  // and dY; the dispatcher places block b on XCD b % 8, so without a remap a slab is fetched into several XCDs' L2s
  // (measured 4.5x the algorithmic HBM reads).  xcd_remap gives every XCD a contiguous range of virtual ids.
  const unsigned vblk = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y, a.dbg & 16);
  const int bx = __builtin_amdgcn_readfirstlane(vblk % gridDim.x), by = __builtin_amdgcn_readfirstlane(vblk / gridDim.x);
  const int unit = (bx / ntn) * 2 + grp;
  const bool uvalid = unit < units;                      // an idle group still splits its half of dY and keeps the barriers
  const int tap = uvalid ? (two_taps ? 2 * unit : unit / ctiles) : 0;
  const int tap1 = two_taps ? tap + 1 : tap;             // tap of the tile's upper row half (may be == T: an empty half)
  const int c0 = (uvalid && !two_taps ? unit % ctiles : 0) * BM, n0 = (bx % ntn) * BN;
  const int khh[2] = {tap / a.KW, tap1 / a.KW};
  const int kwh[2] = {tap - khh[0] * a.KW, tap1 - khh[1] * a.KW};
  const int M = a.NB * a.OH * a.OW;
  int chunk = (M + a.nsplit - 1) / a.nsplit;
  chunk = (chunk + BKX - 1) / BKX * BKX;
  const int p_lo = by * chunk;
  const int p_hi = min(M, p_lo + chunk);
  const int nt = p_hi > p_lo ? (p_hi - p_lo + BKX - 1) / BKX : 0;

  // unit = (pixel pair mp of the K step, channel quad cq): a half wave is 8 consecutive quads (one 128-byte line of
  // a pixel) x 4 pixel pairs of ONE 8-pixel chunk
  const int cq8 = tid & 7, mpl = (tid >> 3) & 3, rest = tid >> 5;
  const int kch = rest & 3;
  const int mp = kch * 4 + mpl;
  const int cqb = (rest >> 2) * 8 + cq8;          // A: quads cqb, cqb + 16; B: quad cqb + 16 * grp
  // pixel-major image: row p = 128 channels = 32 chunks of 4 channels (8 bytes); chunk k4 = c / 4 of pixel p sits at
  // chunk position ((k4 >> 2) ^ fsw(p)) * 4 + (k4 & 3) -- the XOR moves 16-channel blocks only, so the four chunks a
  // transpose read needs stay one 32-byte run -- with fsw(p) bits {p2, p0 ^ p1, p1}: the (pixel 0..3) x (two 16-channel
  // blocks) x (4 chunks) addresses of a half wave's ds_read_b64_tr_b16 cover all 64 banks once, and the (8 quads) x (4 pixel
  // pairs) ds_write_b64 of a half wave hit every bank pair exactly twice (the 2-cycle minimum of a 256-byte store)
  auto fsw = [](int p) __attribute__((always_inline)) { return ((p >> 2) & 1) | (((p ^ (p >> 1)) & 1) << 1) | (((p >> 1) & 1) << 2); };

  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, a.dy_bytes, 0x00020000);
  unsigned a_coff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = two_taps ? cqb * 4 : c0 + (cqb + 16 * i) * 4;
    const bool ok = two_taps ? (i == 0 ? tap : tap1) < T : c < C;
    a_coff[i] = (uvalid && ok) ? (unsigned)(c * 4) : 0x40000000u;   // (extents are < 1 GB: launch_wgrad)
  }
  const int bn = n0 + (cqb + 16 * grp) * 4;
  const unsigned b_noff = bn < N ? (unsigned)(bn * 4) : 0x40000000u;

  auto decode_block = [&](int blk) __attribute__((always_inline)) {
    const int m = p_lo + blk * 256 + tid;
    unsigned off[2] = {0x80000000u, 0x80000000u};
    if (m < p_hi) {
      const int ox = m % a.OW;
      const int t2 = m / a.OW;
      const int oy = t2 % a.OH;
      const int nb = t2 / a.OH;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int iy = oy * a.S + khh[h] - a.pad, ix = ox * a.S + kwh[h] - a.pad;
        if (a.reflect) {
          iy = reflect_x(iy, IH);
          ix = reflect_x(ix, IW);
        }
        if (iy >= 0 && iy < IH && ix >= 0 && ix < IW) off[h] = (unsigned)(((nb * IH + iy) * IW + ix) * a.x_ps * 4);
      }
    }
    poff[blk & 1][0][tid] = off[0];
    poff[blk & 1][1][tid] = off[1];
  };
  f32x4 ra0[2][2], ra1[2][2], rb0[2], rb1[2];
  auto load_a = [&](int t, f32x4 (&ra)[2][2]) __attribute__((always_inline)) {
    if (dbg & 8) return;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned* po = &poff[(t >> 3) & 1][i][(t & 7) * 32 + 2 * mp];
      unsigned o0 = po[0], o1 = po[1];
      if (dbg & 32) { o0 &= 0x80000fffu; o1 &= 0x80000fffu; }   // probe: every gather inside one 4 KB window (cache-resident)
      ra[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, o0 + a_coff[i], 0, 0));
      ra[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, o1 + a_coff[i], 0, 0));
    }
  };
  auto load_b = [&](int t, f32x4 (&rb)[2]) __attribute__((always_inline)) {
    if (dbg & 8) return;
    const int m = p_lo + t * BKX + 2 * mp;
    unsigned r0 = m < p_hi ? (unsigned)(m * a.dy_ps * 4) : 0x80000000u;
    unsigned r1 = m + 1 < p_hi ? (unsigned)((m + 1) * a.dy_ps * 4) : 0x80000000u;
    if (dbg & 32) { r0 &= 0x80000fffu; r1 &= 0x80000fffu; }
    rb[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(yres, r0 + b_noff, 0, 0));
    rb[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(yres, r1 + b_noff, 0, 0));
  };
  // 4 channels x 2 pixels -> per plane and pixel ONE 8-byte store of the pixel's 4 channels (pairs split along channels)
  auto store_unit = [&](unsigned short (&P)[NL][BKX * BM], int cq, const f32x4& lo, const f32x4& hi) __attribute__((always_inline)) {
    const int p0 = 2 * mp, p1 = 2 * mp + 1;
    const int o0 = p0 * BM + ((((cq >> 2) ^ fsw(p0)) << 2) | (cq & 3)) * 4;
    const int o1 = p1 * BM + ((((cq >> 2) ^ fsw(p1)) << 2) | (cq & 3)) * 4;
    unsigned a0, a1, a2, b0, b1, b2;
    if (dbg & 4) return;
    if (dbg & 1) {
      const u32x4 l = __builtin_bit_cast(u32x4, lo), h = __builtin_bit_cast(u32x4, hi);
      *reinterpret_cast<u32x2*>(&P[0][o0]) = (u32x2){l[0], l[1]};
      *reinterpret_cast<u32x2*>(&P[1][o0]) = (u32x2){l[2], l[3]};
      *reinterpret_cast<u32x2*>(&P[2][o0]) = (u32x2){l[0], l[2]};
      *reinterpret_cast<u32x2*>(&P[0][o1]) = (u32x2){h[0], h[1]};
      *reinterpret_cast<u32x2*>(&P[1][o1]) = (u32x2){h[2], h[3]};
      *reinterpret_cast<u32x2*>(&P[2][o1]) = (u32x2){h[0], h[2]};
      return;
    }
    split_pair(lo[0], lo[1], a0, a1, a2);
    split_pair(lo[2], lo[3], b0, b1, b2);
    *reinterpret_cast<u32x2*>(&P[0][o0]) = (u32x2){a0, b0};
    *reinterpret_cast<u32x2*>(&P[1][o0]) = (u32x2){a1, b1};
    *reinterpret_cast<u32x2*>(&P[2][o0]) = (u32x2){a2, b2};
    split_pair(hi[0], hi[1], a0, a1, a2);
    split_pair(hi[2], hi[3], b0, b1, b2);
    *reinterpret_cast<u32x2*>(&P[0][o1]) = (u32x2){a0, b0};
    *reinterpret_cast<u32x2*>(&P[1][o1]) = (u32x2){a1, b1};
    *reinterpret_cast<u32x2*>(&P[2][o1]) = (u32x2){a2, b2};
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int l31 = lane & 31, lh = lane >> 5;
  // ds_read_b64_tr_b16 (gfx950): in a 16-lane group, lane 4r + q supplies the address of 4 consecutive channels (chunk q)
  // of pixel row r and RECEIVES, for channel column (its index in the group), the values of the four rows -- i.e. the
  // k-contiguous MFMA operand out of a pixel-major image (semantics probed on the hardware: profiles/r03c_ds_read_b64_tr_b16_probe.txt).
  // A fragment (32 rows x 16 k, lane = row l & 31, k = 8 (l >> 5) .. + 7) = two reads: pixels +0..3 and +4..7.
  const int tg = lane >> 4, ti = lane & 15, tr = ti >> 2, tq = ti & 3;
  unsigned ea[2][2], eb[2][2];   // byte addresses: [32-row block][first / second 4 pixels] of K half 0, plane 0
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int pix = 8 * (tg >> 1) + 4 * h + tr;
      const int cba = 2 * (wm * 2 + i) + (tg & 1), cbb = 2 * (wn * 2 + i) + (tg & 1);
      ea[i][h] = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&As[0][0] +
                 (unsigned)((pix * BM + (((cba ^ fsw(pix)) << 2) | tq) * 4) * 2);
      eb[i][h] = (unsigned)((pix * BN + (((cbb ^ fsw(pix)) << 2) | tq) * 4) * 2);
    }
  unsigned bcur = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&BsG[0][0][0];
  constexpr int PLB = BKX * BM * 2;      // bytes of one plane
  constexpr int KHB = 16 * BM * 2;       // bytes between the two K halves (16 pixel rows)
#define EG_TRR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
  typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
  struct Frag { u32x2 h[2][2][2]; };    // [block][K half][first / second 4 pixels]
  // (macros, not lambdas: the plane offset must reach the instruction as an immediate)
#define EG_RDA(f, pl)                                                                                     \
  do {                                                                                                    \
    if (dbg & 2) break;                                                                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                    \
      EG_TRR(f.h[i_][0][0], ea[i_][0], (pl) * PLB); EG_TRR(f.h[i_][0][1], ea[i_][1], (pl) * PLB);           \
      EG_TRR(f.h[i_][1][0], ea[i_][0], (pl) * PLB + KHB); EG_TRR(f.h[i_][1][1], ea[i_][1], (pl) * PLB + KHB); \
    }                                                                                                     \
  } while (0)
#define EG_RDB(f, pl)                                                                                     \
  do {                                                                                                    \
    if (dbg & 2) break;                                                                                   \
    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                                    \
      const unsigned b0_ = bcur + eb[j_][0], b1_ = bcur + eb[j_][1];                                      \
      EG_TRR(f.h[j_][0][0], b0_, (pl) * PLB); EG_TRR(f.h[j_][0][1], b1_, (pl) * PLB);                       \
      EG_TRR(f.h[j_][1][0], b0_, (pl) * PLB + KHB); EG_TRR(f.h[j_][1][1], b1_, (pl) * PLB + KHB);           \
    }                                                                                                     \
  } while (0)
  auto op8 = [](const Frag& f, int i, int kk) __attribute__((always_inline)) {
    const u32x4_ v = {f.h[i][kk][0][0], f.h[i][kk][0][1], f.h[i][kk][1][0], f.h[i][kk][1][1]};
    return __builtin_bit_cast(bf16x8, v);
  };
  auto mm = [&](const Frag& af, const Frag& bf) __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op8(bf, j, kk), op8(af, i, kk), acc[i][j], 0, 0, 0);
  };
  // lgkmcnt waits with the fragments they release as operands (the compiler cannot see the asm reads' latency)
#define EG_TRW(n, f) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(f.h[0][0][0]), "+v"(f.h[0][0][1]), "+v"(f.h[0][1][0]), "+v"(f.h[0][1][1]), \
                                  "+v"(f.h[1][0][0]), "+v"(f.h[1][0][1]), "+v"(f.h[1][1][0]), "+v"(f.h[1][1][1]))
  auto lds_barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  const int nblk = (nt + 7) / 8;

  auto run = [&](auto GRP) __attribute__((always_inline)) {
    constexpr int LEAD = decltype(GRP)::value;   // X stages its dY half of step k in VALU phase k, Y that of step k + 1
    // one VALU phase: requests for the next tiles first, then split + store; RT = run-time guards (tail only)
    auto valu_phase = [&](auto RT, int k, const f32x4 (&ra)[2][2], f32x4 (&ran)[2][2], const f32x4 (&rb)[2], f32x4 (&rbn)[2])
                          __attribute__((always_inline)) {
      constexpr bool rt = decltype(RT)::value;
      if (!rt || k + 1 < nt) load_a(k + 1, ran);
      if (!rt || k + 1 + LEAD < nt) load_b(k + 1 + LEAD, rbn);
      if constexpr (!rt) __builtin_amdgcn_sched_barrier(0);
      if ((k & 7) == 2 && (k >> 3) + 1 < nblk) decode_block((k >> 3) + 1);
#pragma unroll
      for (int i = 0; i < 2; ++i) store_unit(As, cqb + 16 * i, ra[i][0], ra[i][1]);
      if (!rt || k + LEAD < nt) store_unit(BsG[(k + LEAD) & 1], cqb + 16 * LEAD, rb[0], rb[1]);
    };
    auto mfma_phase = [&](int k) __attribute__((always_inline)) {
      lds_barrier();
      bcur = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&BsG[k & 1][0][0];
      // six limb products a0b0 a0b1 a0b2 a1b0 a1b1 a2b0; every fragment set is 8 reads, issued one set ahead of its use
      // (LDS reads return in order; at most 15 are tracked, the hardware holds back the rest)
      Frag a0, b0, b1, b2, a1, a2;
      EG_RDA(a0, 0);
      EG_RDB(b0, 0);
      EG_RDB(b1, 1);
      __builtin_amdgcn_sched_barrier(0);
      EG_TRW(8, a0); EG_TRW(8, b0);
      mm(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      EG_RDB(b2, 2);
      EG_TRW(8, b1);
      mm(a0, b1);
      __builtin_amdgcn_sched_barrier(0);
      EG_RDA(a1, 1);
      EG_TRW(8, b2);
      mm(a0, b2);
      __builtin_amdgcn_sched_barrier(0);
      EG_RDA(a2, 2);
      EG_TRW(8, a1);
      mm(a1, b0);
      mm(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      EG_TRW(0, a2);
      mm(a2, b0);
      lds_barrier();
    };
    constexpr std::true_type guarded{};
    constexpr std::false_type steady{};
    // prologue: X holds A(0), B(0); Y holds A(0), B(1) and has already stored its half of B(0) (in its idle slot 0)
    if (nt > 0) {
      decode_block(0);
      lds_barrier();           // both groups: the offset tables are per group, but the barrier is block-wide
      load_a(0, ra0);
      if constexpr (LEAD == 0) {
        load_b(0, rb0);
      } else {
        load_b(0, rb0);
        if (nt > 1) load_b(1, rb1);
        store_unit(BsG[0], cqb + 16, rb0[0], rb0[1]);
        lds_barrier();         // Y's idle slot 0
      }
    }
    // staging sets: A(k) in ra[k & 1]; X: B(k) in rb[k & 1]; Y: B(k + 1) in rb[(k + 1) & 1]
    int k = 0;
    for (; k + 3 < nt; k += 2) {
      if constexpr (LEAD == 0) {
        valu_phase(steady, k, ra0, ra1, rb0, rb1);
        mfma_phase(k);
        valu_phase(steady, k + 1, ra1, ra0, rb1, rb0);
        mfma_phase(k + 1);
      } else {
        valu_phase(steady, k, ra0, ra1, rb1, rb0);
        mfma_phase(k);
        valu_phase(steady, k + 1, ra1, ra0, rb0, rb1);
        mfma_phase(k + 1);
      }
    }
    if (k < nt) {   // up to 3 steps left (k even), straight-line with run-time guards
      if constexpr (LEAD == 0) valu_phase(guarded, k, ra0, ra1, rb0, rb1);
      else valu_phase(guarded, k, ra0, ra1, rb1, rb0);
      mfma_phase(k);
      if (k + 1 < nt) {
        if constexpr (LEAD == 0) valu_phase(guarded, k + 1, ra1, ra0, rb1, rb0);
        else valu_phase(guarded, k + 1, ra1, ra0, rb0, rb1);
        mfma_phase(k + 1);
        if (k + 2 < nt) {
          if constexpr (LEAD == 0) valu_phase(guarded, k + 2, ra0, ra1, rb0, rb1);
          else valu_phase(guarded, k + 2, ra0, ra1, rb1, rb0);
          mfma_phase(k + 2);
        }
      }
    }
    if constexpr (LEAD == 0)
      if (nt > 0) lds_barrier();   // X's idle last slot
  };
  if (grp == 0) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});

  // lane l holds filter row c = .. + (l & 31) and, in registers 4g .. 4g+3, output channels 8g + 4 (l >> 5) .. +3
  if (!uvalid) return;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    // the wave's 64 rows are one half of the tile: in the two-tap form that half is a tap of its own
    const int c = two_taps ? i * 32 + l31 : c0 + wm * WTM + i * 32 + l31;
    const int otap = two_taps ? tap + wm : tap;
    if (c >= C || otap >= T) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * WTN + j * 32 + g * 8 + lh * 4;
        if (n >= N) continue;
        float* q = dense_out ? out + (((size_t)by * T + otap) * C + c) * N + n
                             : out + (size_t)otap * a.w_ts + (size_t)c * a.ldw + n;
        float4 v = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
        if (!dense_out && a.accumulate) {
          const float4 o = *reinterpret_cast<const float4*>(q);
          v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        *reinterpret_cast<float4*>(q) = v;
      }
  }
}
#undef EG_TRR
#undef EG_TRW
#undef EG_RDA
#undef EG_RDB


// =============================================================================================
// filter gradient, ROW-HALO form (three-limb mode): 3x3 stride 1, 4x4 stride 2, 5x5 stride 2 (KW <= 3 S).
// wgrad_x3t above stages, per (tap, 128-channel) unit and K step, the 32 gathered pixels of ONE tap: every staged element
// feeds one tap's products and the staging (gather latency, limb split, LDS stores) is what bounds it (limiter probes:
// profiles/r03w_wgrad_limiter.txt).  Here a unit is (filter ROW ky, column phase par < S, 64-channel block): a K step's 32
// output pixels are R = 32 / W image-row segments of W = min(OW, 32) pixels, and the unit's taps kx = par + S d (d < 3)
// read the same R x (W + ntaps - 1) input pixels -- those of columns S (ox + hx) + par - pad -- shifted by one halo column
// per tap: staged ONCE, pixel-major, and read ntaps times with ds_read_b64_tr_b16 (each lane addresses its own pixel row,
// so a column shift is free).  Staged elements per MFMA at 3x3: 14.7 against 32.
// Block = two groups (units 2b, 2b+1) x four waves, the 128-wide dY tile shared by the groups as in wgrad_x3t; a wave owns
// 32 filter rows x 64 output channels for the unit's taps (6 accumulators at three taps).
// =============================================================================================
template <int S, int KW, int NL>   // NL: limb planes -- 3 (six limb products, fp32 result) or 1 (bf16-rounded operands)
__global__ __launch_bounds__(512, 1) void wgrad_x3r_kernel(const WgradArgs a, float* __restrict__ out, int dense_out) {
  constexpr int BC = 64, BN = 128, HPMAX = 48, NU = 3;
  __shared__ __attribute__((aligned(16))) unsigned short AsG[2][NL][HPMAX * BC];   // per group: [halo pixel][c], 128-byte rows
  __shared__ __attribute__((aligned(16))) unsigned short BsG[2][NL][BKX * BN];     // per K-step parity: [pixel][n]

  const int tid = threadIdx.x & 255, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
  const int C = a.C, N = a.N, OH = a.OH, OW = a.OW;
  constexpr int nph = S < KW ? S : KW, NT = (KW + S - 1) / S;   // NT: most taps a unit has (compile time: register budget)
  static_assert(NT <= 3, "at most three taps per (filter row, column phase)");
  const int cblocks = C / BC, ntn = (N + BN - 1) / BN, upr = nph * cblocks, units = a.KH * upr;   // upr: units per filter row
  const unsigned vblk = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y, a.dbg & 16);
  const int bx = __builtin_amdgcn_readfirstlane(vblk % gridDim.x), by = __builtin_amdgcn_readfirstlane(vblk / gridDim.x);
  const int unit = (bx / ntn) * 2 + grp;
  const bool uvalid = unit < units;                      // an idle group still splits its half of dY and keeps the barriers
  const int ky = uvalid ? unit / upr : 0, urem = uvalid ? unit - ky * upr : 0;
  const int par = urem / cblocks;                         // column phase: the unit's taps are kx = par + S d
  constexpr int ntaps = NT;    // (5x5 stride 2: the odd phase has two taps; its third is computed and dropped -- no run-time tap count in the loop)
  const int c0 = (urem - par * cblocks) * BC, n0 = (bx % ntn) * BN;
  const int IH = a.IH, IW = a.IW;
  const int M = a.NB * OH * OW;
  int chunk = (M + a.nsplit - 1) / a.nsplit;
  chunk = (chunk + BKX - 1) / BKX * BKX;
  const int p_lo = by * chunk;
  const int p_hi = min(M, p_lo + chunk);
  const int nt = p_hi > p_lo ? (p_hi - p_lo + BKX - 1) / BKX : 0;
  // OW, OH are powers of two (launch_wgrad_x3): a K step = R rows x W columns, halo = R x (W + 2) pixels
  const int lOW = __builtin_ctz(OW), lOH = __builtin_ctz(OH);
  const int lW = min(lOW, 5), W = 1 << lW, HWP = W + ntaps - 1, HP = (32 >> lW) * HWP;

  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, a.dy_bytes, 0x00020000);

  // ---- A staging: unit u = tid + 256 j = (halo pixel h = u >> 4, channel quad u & 15); the 16 lanes of a halo pixel store
  // its 128-byte row.  Row q's 16-channel blocks are XORed with fa(q) = 2 ((q >> 1) & 1): the (4 consecutive rows) x (two
  // 16-channel blocks) x (4 chunks) addresses of a half wave's transpose read cover all 64 banks once for any row
  // alignment (so for every tap shift), and the two rows a half wave stores are one 256-byte run.
  auto fa = [](int q) __attribute__((always_inline)) { return q & 2; };
  const unsigned as_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&AsG[grp][0][0];
  int hrow[NU], hcol[NU];
  unsigned sa[NU], a_coff[NU];
  bool hv[NU];
#pragma unroll
  for (int j = 0; j < NU; ++j) {
    const int u = tid + 256 * j, h = u >> 4, cq = u & 15;
    hv[j] = uvalid && h < HP;
    hrow[j] = h / HWP;
    hcol[j] = h - hrow[j] * HWP;          // halo column: input column S (ox0 + hcol) + par - pad
    sa[j] = as_base + (unsigned)(h * (BC * 2) + ((((cq >> 2) ^ fa(h)) << 2) | (cq & 3)) * 8);
    a_coff[j] = (unsigned)((c0 + cq * 4) * 4);
  }
  // (wave-uniform) does unit j of this wave hold any halo pixel?
  bool wact[NU];
#pragma unroll
  for (int j = 0; j < NU; ++j) wact[j] = ((wave * 64 + 256 * j) >> 4) < HP;
  auto a_off = [&](int j, int t) __attribute__((always_inline)) {
    const int m = p_lo + t * BKX;
    const int grow = (m >> lOW) + hrow[j];
    const int nb = grow >> lOH, oy = grow & (OH - 1);
    int iy = oy * S + ky - a.pad, ix = ((m & (OW - 1)) + hcol[j]) * S + par - a.pad;
    if (a.reflect) {
      iy = reflect_x(iy, IH);
      ix = reflect_x(ix, IW);
    }
    const bool ok = hv[j] && (unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW;
#ifdef EG_EXPERIMENTS
    if (a.dbg & 32) return ok ? ((unsigned)(((nb * IH + iy) * IW + ix) * a.x_ps * 4) & 0xfffu) + a_coff[j] : 0x80000000u;   // probe: cache-resident gathers
#endif
    return ok ? (unsigned)(((nb * IH + iy) * IW + ix) * a.x_ps * 4) + a_coff[j] : 0x80000000u;
  };
  // ---- B staging (as wgrad_x3t): unit = (pixel pair mp, channel quad), this group stages quads cqb + 16 grp
  const int cq8 = tid & 7, mpl = (tid >> 3) & 3, rest = tid >> 5;
  const int mp = (rest & 3) * 4 + mpl;
  const int cqb = (rest >> 2) * 8 + cq8;
  auto fsw = [](int p) __attribute__((always_inline)) { return ((p >> 2) & 1) | (((p ^ (p >> 1)) & 1) << 1) | (((p >> 1) & 1) << 2); };
  const int bn = n0 + (cqb + 16 * grp) * 4;
  const unsigned b_noff = bn < N ? (unsigned)(bn * 4) : 0x40000000u;

  f32x4 ra0[NU], ra1[NU], rb0[2], rb1[2];
  auto load_a = [&](int t, f32x4 (&ra)[NU]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NU; ++j)
      if (wact[j]) ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, a_off(j, t), 0, 0));
  };
  auto load_b = [&](int t, f32x4 (&rb)[2]) __attribute__((always_inline)) {
    const int m = p_lo + t * BKX + 2 * mp;
    unsigned r0 = m < p_hi ? (unsigned)(m * a.dy_ps * 4) : 0x80000000u;
    unsigned r1 = m + 1 < p_hi ? (unsigned)((m + 1) * a.dy_ps * 4) : 0x80000000u;
    if (a.dy_up) {   // WgradArgs::dy_up: pixel (nb, oy, ox) reads the half-resolution tensor's (nb, oy / 2, ox / 2) -- m is even
      const int row = m >> lOW, nb = row >> lOH, oy = row & (OH - 1), ox = m & (OW - 1);
      const int hm = (((nb << (lOH - 1)) + (oy >> 1)) << (lOW - 1)) + (ox >> 1);
      r0 = r1 = m < p_hi ? (unsigned)(hm * a.dy_ps * 4) : 0x80000000u;
    }
#ifdef EG_EXPERIMENTS
    if (a.dbg & 32) { r0 &= 0x80000fffu; r1 &= 0x80000fffu; }
#endif
    rb[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(yres, r0 + b_noff, 0, 0));
    rb[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(yres, r1 + b_noff, 0, 0));
  };
  constexpr int PLA = HPMAX * BC * 2;    // bytes of one A plane
  auto store_a = [&](int j, const f32x4& v) __attribute__((always_inline)) {
    __attribute__((address_space(3))) char* q = (__attribute__((address_space(3))) char*)(uintptr_t)sa[j];
#ifdef EG_EXPERIMENTS
    if (a.dbg & 1) {   // probe: no limb split
      const u32x4 l = __builtin_bit_cast(u32x4, v);
      *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(q) = (u32x2){l[0], l[1]};
      *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(q + (NL > 1 ? PLA : 0)) = (u32x2){l[2], l[3]};
      *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(q + (NL > 1 ? 2 * PLA : 0)) = (u32x2){l[0], l[2]};
      return;
    }
#endif
    if constexpr (NL == 1) {
      *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(q) = (u32x2){pack2(v[0], v[1]), pack2(v[2], v[3])};
      return;
    }
    unsigned x0, x1, x2, y0, y1, y2;
    split_pair(v[0], v[1], x0, x1, x2);
    split_pair(v[2], v[3], y0, y1, y2);
    *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(q) = (u32x2){x0, y0};
    *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(q + (NL > 1 ? PLA : 0)) = (u32x2){x1, y1};
    *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(q + (NL > 1 ? 2 * PLA : 0)) = (u32x2){x2, y2};
  };
  auto store_b = [&](unsigned short (&P)[NL][BKX * BN], int cq, const f32x4& lo, const f32x4& hi) __attribute__((always_inline)) {
    const int p0 = 2 * mp, p1 = 2 * mp + 1;
    const int o0 = p0 * BN + ((((cq >> 2) ^ fsw(p0)) << 2) | (cq & 3)) * 4;
    const int o1 = p1 * BN + ((((cq >> 2) ^ fsw(p1)) << 2) | (cq & 3)) * 4;
    if constexpr (NL == 1) {
      *reinterpret_cast<u32x2*>(&P[0][o0]) = (u32x2){pack2(lo[0], lo[1]), pack2(lo[2], lo[3])};
      *reinterpret_cast<u32x2*>(&P[0][o1]) = (u32x2){pack2(hi[0], hi[1]), pack2(hi[2], hi[3])};
      return;
    }
    unsigned x0, x1, x2, y0, y1, y2;
    split_pair(lo[0], lo[1], x0, x1, x2);
    split_pair(lo[2], lo[3], y0, y1, y2);
    *reinterpret_cast<u32x2*>(&P[0][o0]) = (u32x2){x0, y0};
    *reinterpret_cast<u32x2*>(&P[NL > 1 ? 1 : 0][o0]) = (u32x2){x1, y1};
    *reinterpret_cast<u32x2*>(&P[NL > 1 ? 2 : 0][o0]) = (u32x2){x2, y2};
    split_pair(hi[0], hi[1], x0, x1, x2);
    split_pair(hi[2], hi[3], y0, y1, y2);
    *reinterpret_cast<u32x2*>(&P[0][o1]) = (u32x2){x0, y0};
    *reinterpret_cast<u32x2*>(&P[NL > 1 ? 1 : 0][o1]) = (u32x2){x1, y1};
    *reinterpret_cast<u32x2*>(&P[NL > 1 ? 2 : 0][o1]) = (u32x2){x2, y2};
  };

  // wave (wc, wn2) owns filter rows 32 wc .. +31 of the unit and output channels 64 wn2 .. +63 of the dY tile: the dY
  // fragments (two 32-channel blocks) are read once per K step and serve every tap, the per-tap x fragment is one block --
  // 60 transpose reads per 72 MFMAs (the 64 x 32 split needs 84)
  const int wc = wave >> 1, wn2 = wave & 1;
  f32x16 acc[NT][2];    // [tap d][32-channel block j of the wave's 64 output channels]
#pragma unroll
  for (int d = 0; d < NT; ++d)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[d][i][r] = 0.f;
  const int l31 = lane & 31, lh = lane >> 5;
  // transpose reads (see wgrad_x3t): lane 4r + q of a 16-lane group addresses chunk q of pixel row r
  const int tg = lane >> 4, ti = lane & 15, tr = ti >> 2, tq = ti & 3;
  unsigned ea[2][2][NT];   // A: [K half][first / second 4 pixels][tap column], plane 0
  unsigned eb[2][2];      // B: [32-channel block][first / second 4 pixels] of K half 0, plane 0, relative to the parity's base
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int p = 16 * kk + 8 * (tg >> 1) + 4 * h + tr;
      const int q0 = (p >> lW) * HWP + (p & (W - 1));
#pragma unroll
      for (int d = 0; d < NT; ++d) {
        const int q = q0 + d;
        ea[kk][h][d] = as_base + (unsigned)(q * (BC * 2) + ((((2 * wc + (tg & 1)) ^ fa(q)) << 2) | tq) * 8);
      }
    }
#pragma unroll
  for (int jb = 0; jb < 2; ++jb)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int pix = 8 * (tg >> 1) + 4 * h + tr;
      const int cbb = 2 * (2 * wn2 + jb) + (tg & 1);
      eb[jb][h] = (unsigned)((pix * BN + (((cbb ^ fsw(pix)) << 2) | tq) * 4) * 2);
    }
  unsigned bcur = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&BsG[0][0][0];
  constexpr int PLB = BKX * BN * 2;      // bytes of one B plane
  constexpr int KHB = 16 * BN * 2;       // bytes between the two K halves of the B tile (16 pixel rows)
#define EG_TRR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
  struct FragA { u32x2 h[2][2]; };       // [K half][first / second 4 pixels]                      (4 reads)
  struct FragB { u32x2 h[2][2][2]; };    // [32-channel block][K half][first / second 4 pixels]  (8 reads)
#define EG_RDA(f, pl, d)                                                                                  \
  do {                                                                                                    \
    EG_TRR(f.h[0][0], ea[0][0][d], (pl) * PLA); EG_TRR(f.h[0][1], ea[0][1][d], (pl) * PLA);                 \
    EG_TRR(f.h[1][0], ea[1][0][d], (pl) * PLA); EG_TRR(f.h[1][1], ea[1][1][d], (pl) * PLA);                 \
  } while (0)
#define EG_RDB(f, pl)                                                                                     \
  do {                                                                                                    \
    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                                    \
      const unsigned b0_ = bcur + eb[j_][0], b1_ = bcur + eb[j_][1];                                      \
      EG_TRR(f.h[j_][0][0], b0_, (pl) * PLB); EG_TRR(f.h[j_][0][1], b1_, (pl) * PLB);                       \
      EG_TRR(f.h[j_][1][0], b0_, (pl) * PLB + KHB); EG_TRR(f.h[j_][1][1], b1_, (pl) * PLB + KHB);           \
    }                                                                                                     \
  } while (0)
#define EG_TWB(n, f) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(f.h[0][0][0]), "+v"(f.h[0][0][1]), "+v"(f.h[0][1][0]), "+v"(f.h[0][1][1]), \
                                  "+v"(f.h[1][0][0]), "+v"(f.h[1][0][1]), "+v"(f.h[1][1][0]), "+v"(f.h[1][1][1]))
#define EG_TWA(n, f) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(f.h[0][0]), "+v"(f.h[0][1]), "+v"(f.h[1][0]), "+v"(f.h[1][1]))
  auto opa = [](const FragA& f, int kk) __attribute__((always_inline)) {
    const u32x4 v = {f.h[kk][0][0], f.h[kk][0][1], f.h[kk][1][0], f.h[kk][1][1]};
    return __builtin_bit_cast(bf16x8, v);
  };
  auto opb = [](const FragB& f, int jb, int kk) __attribute__((always_inline)) {
    const u32x4 v = {f.h[jb][kk][0][0], f.h[jb][kk][0][1], f.h[jb][kk][1][0], f.h[jb][kk][1][1]};
    return __builtin_bit_cast(bf16x8, v);
  };
  auto lds_barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  auto run = [&](auto GRP) __attribute__((always_inline)) {
    constexpr int LEAD = decltype(GRP)::value;   // X stages its dY half of step k in VALU phase k, Y that of step k + 1
    auto valu_phase = [&](auto RT, int k, const f32x4 (&ra)[NU], f32x4 (&ran)[NU], const f32x4 (&rb)[2], f32x4 (&rbn)[2])
                          __attribute__((always_inline)) {
      constexpr bool rt = decltype(RT)::value;
      if (!rt || k + 1 < nt) load_a(k + 1, ran);
      if (!rt || k + 1 + LEAD < nt) load_b(k + 1 + LEAD, rbn);
      if constexpr (!rt) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NU; ++j)
        if (wact[j]) store_a(j, ra[j]);
      if (!rt || k + LEAD < nt) store_b(BsG[(k + LEAD) & 1], cqb + 16 * LEAD, rb[0], rb[1]);
    };
    auto mfma_phase = [&](int k) __attribute__((always_inline)) {
      lds_barrier();
      bcur = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&BsG[k & 1][0][0];
      // per tap column the six limb products a0b0 a0b1 a0b2 a1b0 a1b1 a2b0 (4 MFMAs each); the dY fragments are read once
      // and serve the three taps; every fragment set is requested at least one product ahead of its use
      FragB b0, b1, b2;
      FragA a0, a1, a2;
      auto mm = [&](int d, const FragA& af, const FragB& bf) __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int jb = 0; jb < 2; ++jb)
            acc[d][jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(opb(bf, jb, kk), opa(af, kk), acc[d][jb], 0, 0, 0);
      };
      if constexpr (NL == 1) {   // one product per tap: the three taps' fragments are a0, a1, a2
        EG_RDB(b0, 0);
        EG_RDA(a0, 0, 0);
        if constexpr (NT > 1) EG_RDA(a1, 0, NT > 1 ? 1 : 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NT > 1) { EG_TWB(4, b0); EG_TWA(4, a0); } else { EG_TWB(0, b0); EG_TWA(0, a0); }
        mm(0, a0, b0);
        if constexpr (NT > 1) {
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (NT > 2) { EG_RDA(a2, 0, NT > 2 ? 2 : 0); EG_TWA(4, a1); } else { EG_TWA(0, a1); }
          mm(1, a1, b0);
          if constexpr (NT > 2) {
            __builtin_amdgcn_sched_barrier(0);
            EG_TWA(0, a2);
            mm(2, a2, b0);
          }
        }
        lds_barrier();
        return;
      }
      // (outstanding reads in brackets; LDS reads return in order)
      EG_RDB(b0, 0);
      EG_RDA(a0, 0, 0);
      EG_RDB(b1, 1);                  // [b0 8, a0 4, b1 8]
      __builtin_amdgcn_sched_barrier(0);
      EG_TWB(8, b0); EG_TWA(8, a0);
      mm(0, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      EG_RDB(b2, 2);                  // [b1 8, b2 8]
      EG_TWB(8, b1);
      mm(0, a0, b1);
      __builtin_amdgcn_sched_barrier(0);
      EG_RDA(a1, 1, 0);               // [b2 8, a1 4]
      EG_TWB(4, b2);
      mm(0, a0, b2);
      __builtin_amdgcn_sched_barrier(0);
      EG_RDA(a2, 2, 0);               // [a1 4, a2 4]
      EG_TWA(4, a1);
      mm(0, a1, b0);
      mm(0, a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      if (NT > 1 && ntaps > 1) {      // (compile time)
        EG_RDA(a0, 0, NT > 1 ? 1 : 0);   // [a2 4, a0 4]
        EG_TWA(4, a2);
      } else {
        EG_TWA(0, a2);
      }
      mm(0, a2, b0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int d = 1; d < NT; ++d) {
        if (d >= ntaps) break;
        EG_RDA(a1, 1, d);
        EG_TWA(4, a0);
        mm(d, a0, b0);
        mm(d, a0, b1);
        mm(d, a0, b2);
        __builtin_amdgcn_sched_barrier(0);
        EG_RDA(a2, 2, d);
        EG_TWA(4, a1);
        mm(d, a1, b0);
        mm(d, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        if (d == 1 && NT > 2 && ntaps > 2) {
          EG_RDA(a0, 0, NT > 2 ? 2 : 0);
          EG_TWA(4, a2);
        } else {
          EG_TWA(0, a2);
        }
        mm(d, a2, b0);
        __builtin_amdgcn_sched_barrier(0);
      }
      lds_barrier();
    };
    constexpr std::true_type guarded{};
    constexpr std::false_type steady{};
    // prologue: X holds A(0), B(0); Y holds A(0), B(1) and has already stored its half of B(0) (in its idle slot 0)
    if (nt > 0) {
      load_a(0, ra0);
      if constexpr (LEAD == 0) {
        load_b(0, rb0);
      } else {
        load_b(0, rb0);
        if (nt > 1) load_b(1, rb1);
        store_b(BsG[0], cqb + 16, rb0[0], rb0[1]);
        lds_barrier();         // Y's idle slot 0
      }
    }
    // staging sets: A(k) in ra[k & 1]; X: B(k) in rb[k & 1]; Y: B(k + 1) in rb[(k + 1) & 1]
    int k = 0;
    for (; k + 3 < nt; k += 2) {
      if constexpr (LEAD == 0) {
        valu_phase(steady, k, ra0, ra1, rb0, rb1);
        mfma_phase(k);
        valu_phase(steady, k + 1, ra1, ra0, rb1, rb0);
        mfma_phase(k + 1);
      } else {
        valu_phase(steady, k, ra0, ra1, rb1, rb0);
        mfma_phase(k);
        valu_phase(steady, k + 1, ra1, ra0, rb0, rb1);
        mfma_phase(k + 1);
      }
    }
    if (k < nt) {   // up to 3 steps left (k even), straight-line with run-time guards
      if constexpr (LEAD == 0) valu_phase(guarded, k, ra0, ra1, rb0, rb1);
      else valu_phase(guarded, k, ra0, ra1, rb1, rb0);
      mfma_phase(k);
      if (k + 1 < nt) {
        if constexpr (LEAD == 0) valu_phase(guarded, k + 1, ra1, ra0, rb1, rb0);
        else valu_phase(guarded, k + 1, ra1, ra0, rb0, rb1);
        mfma_phase(k + 1);
        if (k + 2 < nt) {
          if constexpr (LEAD == 0) valu_phase(guarded, k + 2, ra0, ra1, rb0, rb1);
          else valu_phase(guarded, k + 2, ra0, ra1, rb1, rb0);
          mfma_phase(k + 2);
        }
      }
    }
    if constexpr (LEAD == 0)
      if (nt > 0) lds_barrier();   // X's idle last slot
  };
  if (grp == 0) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});

  // lane l holds filter row c = .. + (l & 31) and, in registers 4g .. 4g+3, output channels 8g + 4 (l >> 5) .. +3
  if (!uvalid) return;
  const int T = a.KH * KW;
#pragma unroll
  for (int d = 0; d < NT; ++d)
#pragma unroll
    for (int i = 0; i < 2; ++i) {     // (i: the wave's 32-channel block of output channels)
      if (par + S * d >= KW) continue;
      const int c = c0 + wc * 32 + l31, otap = ky * KW + par + S * d;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn2 * 64 + i * 32 + g * 8 + lh * 4;
        if (n >= N) continue;
        float* q = dense_out ? out + (((size_t)by * T + otap) * C + c) * N + n
                             : out + (size_t)otap * a.w_ts + (size_t)c * a.ldw + n;
        float4 v = make_float4(acc[d][i][4 * g], acc[d][i][4 * g + 1], acc[d][i][4 * g + 2], acc[d][i][4 * g + 3]);
        if (a.dy_up) { v.x *= 0.25f; v.y *= 0.25f; v.z *= 0.25f; v.w *= 0.25f; }   // (exact: the pool's 1/4 applied to the sum)
        if (!dense_out && a.accumulate) {
          const float4 o = *reinterpret_cast<const float4*>(q);
          v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        *reinterpret_cast<float4*>(q) = v;
      }
    }
}
#undef EG_TRR
#undef EG_RDA
#undef EG_RDB
#undef EG_TWA
#undef EG_TWB

// the row-halo form applies to: at most three taps per (filter row, column phase) -- 3x3 stride 1, 4x4 and 5x5 stride 2 --
// 64-channel blocks, power-of-two output maps at least 4 wide; returns its unit count (0: does not apply)
int wgrad_x3_rowhalo_units(const WgradArgs& a) {
  static const int mode = dbg_env("EG_WGRAD_ROWHALO") ? atoi(dbg_env("EG_WGRAD_ROWHALO")) : 3;   // A/B: bit 0 stride 1, bit 1 stride 2, bit 2 the bf16 mode
  // (bf16 mode stays on the per-tap kernel: with one product per tap the row-halo form's per-step address arithmetic and
  //  three short MFMA bursts lose to it, 350-385 against 446-495 TFLOP/s, iteration -3.7 %: profiles/r03w_wgrad_rowhalo_check_bf16.txt)
  auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
  const bool shape = (a.S == 1 && a.KW == 3) || (a.S == 2 && (a.KW == 4 || a.KW == 5));
  const bool ok = (a.limbs == 3 || (a.limbs == 1 && (mode & 4))) && shape && a.KH == a.KW && ((mode >> (a.S - 1)) & 1) && a.C % 64 == 0 &&
                  pow2(a.OW) && pow2(a.OH) && a.OW >= 4;
  return ok ? a.KH * std::min(a.S, a.KW) * (a.C / 64) : 0;
}

const char* launch_wgrad_x3(const WgradArgs& a, float* out, int dense, bool big_tile, hipStream_t s) {
  (void)big_tile;
#ifdef EG_EXPERIMENTS
  const_cast<WgradArgs&>(a).dbg = dbg_env("EG_WG_DBG") ? atoi(dbg_env("EG_WG_DBG")) : 0;
#endif
  const int ntn = (a.N + 127) / 128;
  if (const int ru = wgrad_x3_rowhalo_units(a)) {
    dim3 grid(((ru + 1) / 2) * ntn, a.nsplit);
    if (a.limbs == 1) {
      if (a.S == 1) hipLaunchKernelGGL((wgrad_x3r_kernel<1, 3, 1>), grid, dim3(512), 0, s, a, out, dense);
      else if (a.KW == 4) hipLaunchKernelGGL((wgrad_x3r_kernel<2, 4, 1>), grid, dim3(512), 0, s, a, out, dense);
      else hipLaunchKernelGGL((wgrad_x3r_kernel<2, 5, 1>), grid, dim3(512), 0, s, a, out, dense);
      return a.S == 1 ? "wgrad_bf16r<3x3,2x64,128>" : a.KW == 4 ? "wgrad_bf16r<4x4s2,2x64,128>" : "wgrad_bf16r<5x5s2,2x64,128>";
    }
    if (a.S == 1) hipLaunchKernelGGL((wgrad_x3r_kernel<1, 3, 3>), grid, dim3(512), 0, s, a, out, dense);
    else if (a.KW == 4) hipLaunchKernelGGL((wgrad_x3r_kernel<2, 4, 3>), grid, dim3(512), 0, s, a, out, dense);
    else hipLaunchKernelGGL((wgrad_x3r_kernel<2, 5, 3>), grid, dim3(512), 0, s, a, out, dense);
    return a.S == 1 ? "wgrad_x3r<3x3,2x64,128>" : a.KW == 4 ? "wgrad_x3r<4x4s2,2x64,128>" : "wgrad_x3r<5x5s2,2x64,128>";
  }
  const int units = a.C == 64 ? (a.KH * a.KW + 1) / 2 : a.KH * a.KW * ((a.C + 127) / 128);
  dim3 grid(((units + 1) / 2) * ntn, a.nsplit);
  if (a.limbs == 1) {
    hipLaunchKernelGGL((wgrad_x3pp_kernel<1>), grid, dim3(512), 0, s, a, out, dense);
    return "wgrad_bf16pp<2x128,128>";
  }
  static const bool no_tr = dbg_env("EG_NO_WGRAD_TR") != nullptr;   // A/B: the channel-major image with plain 16-byte fragment reads
  if (!no_tr) {
    hipLaunchKernelGGL((wgrad_x3t_kernel<6>), grid, dim3(512), 0, s, a, out, dense);
    return "wgrad_x3t<2x128,128>";
  }
  hipLaunchKernelGGL((wgrad_x3pp_kernel<6>), grid, dim3(512), 0, s, a, out, dense);
  return "wgrad_x3pp<2x128,128>";
}

// =============================================================================================
// three-limb bf16 filter planes, K-step-major: src [T][R][Cc] fp32 -> out[plane][T][K/32][Nrows][32] bf16 with
// (Nrows, K) = (R, Cc) ("as stored": the contracted index is the source's fast one) or (Cc, R) (transposed)
// =============================================================================================
__global__ void split_bf16x3_kernel(const float* __restrict__ in, unsigned short* __restrict__ out, int R, int Cc,
                                    int transposed, size_t plane, int nplanes) {
  __shared__ float tile[32][33];
  const int t = blockIdx.z;
  const float* src = in + (size_t)t * R * Cc;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + threadIdx.x;
    tile[i][threadIdx.x] = (r < R && c < Cc) ? src[(size_t)r * Cc + c] : 0.f;
  }
  __syncthreads();
  // output rows n, contracted k; this block holds a 32 x 32 patch
  const int Nrows = transposed ? Cc : R, K = transposed ? R : Cc;
  const int nb0 = transposed ? c0 : r0, kb0 = transposed ? r0 : c0;   // kb0 is a multiple of 32: one K chunk
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int n = nb0 + i, k = kb0 + threadIdx.x;
    if (n < Nrows && k < K) {
      const float v = transposed ? tile[threadIdx.x][i] : tile[i][threadIdx.x];
      unsigned p0, p1, p2;
      split_pair(v, 0.f, p0, p1, p2);
      const size_t o = (((size_t)t * (K / 32) + kb0 / 32) * Nrows + n) * 32 + threadIdx.x;
      out[o] = (unsigned short)(p0 & 0xffffu);
      if (nplanes == 3) {
        out[o + plane] = (unsigned short)(p1 & 0xffffu);
        out[o + 2 * plane] = (unsigned short)(p2 & 0xffffu);
      }
    }
  }
}
// Round 6: a thread owns FOUR consecutive contracted indices of one output row -- the block's 32 x 32 patch is 2 KB contiguous
// per plane, written as 8-byte stores (512 B per wave instruction; the form above stores 2 bytes per lane), and the patch
// is read with 16-byte loads where the source allows it.  The classifier's 17 M parameters (one launch after its update,
// on the path into sub-step 5): 105 -> 60 us in f32x3 mode.
__global__ __launch_bounds__(256) void split_bf16x3_batch_kernel(const SplitBatch b) {
  __shared__ float tile[32][33];
  int e = 0;
  while (e + 1 < b.count && blockIdx.x >= b.start[e + 1]) ++e;
  const SplitDesc& d = b.d[e];
  const int local = (int)(blockIdx.x - b.start[e]);
  const int bx = local % d.gx, by = (local / d.gx) % d.gy, t = local / (d.gx * d.gy);
  const int R = d.R, Cc = d.Cc, transposed = d.transposed;
  const float* src = d.in + (size_t)t * R * Cc;
  const int r0 = by * 32, c0 = bx * 32;
  const int tid = threadIdx.y * 32 + threadIdx.x;
  if ((Cc & 3) == 0 && ((((uintptr_t)src)) & 15) == 0) {
    const int r = r0 + (tid >> 3), c = c0 + (tid & 7) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < R && c < Cc) v = *reinterpret_cast<const float4*>(src + (size_t)r * Cc + c);   // (Cc % 4 == 0: the quad is inside the row)
    float* tr = &tile[tid >> 3][(tid & 7) * 4];
    tr[0] = v.x; tr[1] = v.y; tr[2] = v.z; tr[3] = v.w;
  } else {
    for (int i = threadIdx.y; i < 32; i += 8) {
      const int r = r0 + i, c = c0 + threadIdx.x;
      tile[i][threadIdx.x] = (r < R && c < Cc) ? src[(size_t)r * Cc + c] : 0.f;
    }
  }
  __syncthreads();
  const int Nrows = transposed ? Cc : R, K = transposed ? R : Cc;
  const int nb0 = transposed ? c0 : r0, kb0 = transposed ? r0 : c0;
  const int ni = tid >> 3, k4 = (tid & 7) * 4;
  const int n = nb0 + ni, k = kb0 + k4;
  if (n < Nrows && k < K) {
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = transposed ? tile[k4 + j][ni] : tile[ni][k4 + j];
    unsigned p0[2], p1[2], p2[2];
    split_pair(v[0], v[1], p0[0], p1[0], p2[0]);
    split_pair(v[2], v[3], p0[1], p1[1], p2[1]);
    const size_t o = (((size_t)t * (K / 32) + kb0 / 32) * Nrows + n) * 32 + k4;
    if (k + 3 < K && (d.plane & 3) == 0) {
      *reinterpret_cast<uint2*>(d.out + o) = make_uint2(p0[0], p0[1]);
      if (d.nplanes == 3) {
        *reinterpret_cast<uint2*>(d.out + o + d.plane) = make_uint2(p1[0], p1[1]);
        *reinterpret_cast<uint2*>(d.out + o + 2 * d.plane) = make_uint2(p2[0], p2[1]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (k + j < K) {
          const int sh = (j & 1) * 16;
          d.out[o + j] = (unsigned short)((p0[j >> 1] >> sh) & 0xffffu);
          if (d.nplanes == 3) {
            d.out[o + j + d.plane] = (unsigned short)((p1[j >> 1] >> sh) & 0xffffu);
            d.out[o + j + 2 * d.plane] = (unsigned short)((p2[j >> 1] >> sh) & 0xffffu);
          }
        }
    }
  }
}
void launch_split_bf16x3_batch(const SplitBatch& b, hipStream_t s) {
  if (b.count == 0) return;
  hipLaunchKernelGGL(split_bf16x3_batch_kernel, dim3(b.start[b.count]), dim3(32, 8), 0, s, b);
}
void launch_split_bf16x3(const float* in, unsigned short* out, int T, int R, int Cc, int transposed, size_t plane,
                         int nplanes, hipStream_t s) {
  dim3 grid((Cc + 31) / 32, (R + 31) / 32, T);
  hipLaunchKernelGGL(split_bf16x3_kernel, grid, dim3(32, 8), 0, s, in, out, R, Cc, transposed, plane, nplanes);
}

}  // namespace eg
