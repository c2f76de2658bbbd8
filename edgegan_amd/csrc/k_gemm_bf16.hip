// bf16-operand variants of the tap-gather implicit GEMMs (eg_config.compute_dtype == 1; BASELINE configs[1],[3],[4]).
//
// Mixed precision in the usual sense: activations, gradients, weights and optimizer state stay fp32 in HBM; the two
// operands of every dense contraction are rounded to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) on their way
// into LDS and multiplied on v_mfma_f32_16x16x32_bf16 (fp32 accumulate, 16x the fp32 MFMA rate).  Everything that
// is not a dense contraction with K-extent % 32 == 0 (norms, activations, losses, RMSProp, the thin 3/8-channel
// layers, the linears) runs the fp32 kernels unchanged.
//
//   conv_gemm_bf16_kernel : Y[pix_out(m)][n] (+)= b[n] + sum_taps sum_c bf16(X[pix_in(m,tap)][c]) * Wb[tap][n][c]
//                           Wb = bf16 copy of the filter with the CONTRACTED index contiguous (made once per
//                           sub-step by launch_cvt_bf16 / launch_tap_transpose_bf16)
//   wgrad_bf16_kernel     : dW[tap][c][n] = sum_m bf16(X[pix_in(m,tap)][c]) * bf16(dY[m][n])   (split over m)
//
// MFMA operand layout (16x16x32): lane l supplies row/col (l & 15) and the 8 consecutive k = (l >> 4) * 8 .. + 7, so
// both LDS images are [row][32 k] with k contiguous (64-byte rows, no padding) and every fragment is ONE 16-byte
// ds_read.  The four 16-byte chunks of a row are XOR-swizzled (lds_swz): with the gfx950 ds_read_b128 lane groups
// {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... each group's 16 reads then tile all 64 banks.
// The wgrad contracts over pixels, which are the SLOW index of both NHWC operands; the transpose happens in the
// register -> LDS store: a thread owns 4 channels x 2 pixels and writes 4 dwords (2 pixels packed) to 4 rows.  A
// half wave is 8 channel quads (one 128-byte line per pixel, so the global loads stay coalesced) x 4 pixel pairs;
// the swizzle plus a per-lane order of the 4 row stores makes every ds_write_b32 hit 32 distinct banks.
#include "common.h"
#include "kernels.h"

namespace eg {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ uint4 pack_bf16x8(const float4& a, const float4& b) {
  return make_uint4(pack_bf16(a.x, a.y), pack_bf16(a.z, a.w), pack_bf16(b.x, b.y), pack_bf16(b.z, b.w));
}
__device__ __forceinline__ int reflect_idx_b(int i, int n) {
  i = i < 0 ? -i : i;
  return i >= n ? 2 * (n - 1) - i : i;
}

constexpr int BKB = 32;   // contraction elements per K step (one MFMA)
constexpr int ROWB = BKB; // LDS row pitch in bf16 elements (64 bytes)

// physical 16-byte chunk = logical chunk ^ lds_swz(row)
__device__ __forceinline__ int lds_swz(int row) { return ((-((row >> 2) & 3)) & 3) ^ ((row >> 4) & 1); }

// =============================================================================================
// conv_gemm_bf16_kernel
// =============================================================================================
// ABF: the gathered operand is read from a bf16 copy of the activation (a.x_bf, same [pixel][x_ps] layout): half the
// bytes per K step through L2 (the kernel's limiter) and no conversion on the way into LDS.
template <int BM, int BN, int WTM, int WTN, bool SPLITK, bool ABF>
__global__ __launch_bounds__(256, 2) void conv_gemm_bf16_kernel(const ConvArgs a) {
  constexpr int WGN = BN / WTN;
  constexpr int MI = WTM / 16, NJ = WTN / 16;
  static_assert((BM / WTM) * (BN / WTN) == 4, "4 waves");
  __shared__ __attribute__((aligned(16))) unsigned short As[2][BM * ROWB];   // double-buffered: one barrier per K step
  __shared__ __attribute__((aligned(16))) unsigned short Bs[2][BN * ROWB];

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
  const int st8 = ((tid & 3) ^ lds_swz(tid >> 2)) * 8;   // rows tid/4 + 64*i all share this swizzle
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
  const int cpt = C / BKB;
  const int nk = ntaps * cpt;
  int nx_ty = 0, nx_tx = 0, nx_c0 = 0;
  int kt_lo = 0, kt_hi = nk;
  if constexpr (SPLITK) {
    const int per = (nk + a.ksplit - 1) / a.ksplit;
    kt_lo = zsplit * per;
    kt_hi = min(nk, kt_lo + per);
    const int tap0 = kt_lo / cpt;
    nx_c0 = (kt_lo - tap0 * cpt) * BKB;
    nx_ty = tap0 / ph.Tx;
    nx_tx = tap0 - nx_ty * ph.Tx;
  }

  // two register staging sets: the global loads of K tiles t+1 and t+2 are in flight while tile t is multiplied
  // (the kernel is bound by bytes in flight per CU, not by the MFMA pipe)
  constexpr int AW = ABF ? 1 : 2;   // 16-byte registers per gathered 8-element chunk
  float4 ra0[A_V][AW], ra1[A_V][AW];
  uint4 rb0[B_V], rb1[B_V];
  auto load_tile = [&](float4 (&ra)[A_V][AW], uint4 (&rb)[B_V]) {
    const int ty = nx_ty, tx = nx_tx, c0 = nx_c0;
    const long long wt = (long long)((ph.kh0 + ty * a.kst) * a.KW + (ph.kw0 + tx * a.kst)) * a.wbf_ts;
#pragma unroll
    for (int i = 0; i < A_V; ++i) {
      int iy = r_iy0[i] + ty * a.D, ix = r_ix0[i] + tx * a.D;
      if (a.reflect) {
        iy = reflect_idx_b(iy, IH);
        ix = reflect_idx_b(ix, IW);
      }
      float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
      if (r_base[i] >= 0 && iy >= 0 && iy < IH && ix >= 0 && ix < IW) {
        if constexpr (ABF) {
          v0 = *reinterpret_cast<const float4*>(a.x_bf + (size_t)(r_base[i] + iy * IW + ix) * a.x_ps + c0 + q8);
        } else {
          const float* px = a.x + (size_t)(r_base[i] + iy * IW + ix) * a.x_ps + c0 + q8;
          v0 = *reinterpret_cast<const float4*>(px);
          v1 = *reinterpret_cast<const float4*>(px + 4);
        }
      }
      ra[i][0] = v0;
      if constexpr (!ABF) ra[i][1] = v1;
    }
#pragma unroll
    for (int i = 0; i < B_V; ++i) {
      const int n = n0 + (tid >> 2) + 64 * i;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (n < N) v = *reinterpret_cast<const uint4*>(a.wgt_bf + wt + (size_t)n * a.ldwbf + c0 + q8);
      rb[i] = v;
    }
    nx_c0 += BKB;
    if (nx_c0 >= C) {
      nx_c0 = 0;
      if (++nx_tx == ph.Tx) {
        nx_tx = 0;
        ++nx_ty;
      }
    }
  };
  auto store_tile = [&](const float4 (&ra)[A_V][AW], const uint4 (&rb)[B_V], int buf) {
#pragma unroll
    for (int i = 0; i < A_V; ++i) {
      if constexpr (ABF)
        *reinterpret_cast<float4*>(&As[buf][((tid >> 2) + 64 * i) * ROWB + st8]) = ra[i][0];
      else
        *reinterpret_cast<uint4*>(&As[buf][((tid >> 2) + 64 * i) * ROWB + st8]) = pack_bf16x8(ra[i][0], ra[i][AW - 1]);
    }
#pragma unroll
    for (int i = 0; i < B_V; ++i) *reinterpret_cast<uint4*>(&Bs[buf][((tid >> 2) + 64 * i) * ROWB + st8]) = rb[i];
  };

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int l15 = lane & 15, lq = lane >> 4;
  const int fsw = (-(l15 >> 2)) & 3;   // lds_swz of a fragment row, without the (row >> 4) & 1 bit (= i & 1 below)
  static_assert((WTM / 16) % 2 == 0 && (WTN / 16) % 2 == 0, "wave sub-tiles start at even 16-row blocks");
  auto compute = [&](int buf) {
    bf16x8 af[MI], bf[NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
      af[i] = *reinterpret_cast<const bf16x8*>(&As[buf][(wm * WTM + i * 16 + l15) * ROWB + (lq ^ fsw ^ (i & 1)) * 8]);
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      bf[j] = *reinterpret_cast<const bf16x8*>(&Bs[buf][(wn * WTN + j * 16 + l15) * ROWB + (lq ^ fsw ^ (j & 1)) * 8]);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
  };

  // step t: LDS buffer t&1 holds tile t; tile t+1 goes from its staging set into the other buffer (whose readers
  // passed the previous barrier), that set is refilled with tile t+3
  const int nt = kt_hi - kt_lo;
  if (nt > 0) load_tile(ra0, rb0);
  if (nt > 1) load_tile(ra1, rb1);
  if (nt > 0) {
    store_tile(ra0, rb0, 0);
    if (nt > 2) load_tile(ra0, rb0);
  }
  __syncthreads();
  for (int t = 0; t < nt; t += 2) {
    if (t + 1 < nt) {
      store_tile(ra1, rb1, 1);
      if (t + 3 < nt) load_tile(ra1, rb1);
    }
    compute(0);
    __syncthreads();
    if (t + 1 >= nt) break;
    if (t + 2 < nt) {
      store_tile(ra0, rb0, 0);
      if (t + 4 < nt) load_tile(ra0, rb0);
    }
    compute(1);
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

bool conv_bf16_eligible(const ConvArgs& a) {
  return a.wgt_bf && a.limbs == 0 && a.C % BKB == 0 && a.N > 16 && a.N % 4 == 0 && a.x_ps % 4 == 0 && a.y_ps % 4 == 0 &&
         a.ldwbf % 8 == 0 && a.wbf_ts % 8 == 0 && (((uintptr_t)a.x & 15) == 0) && (((uintptr_t)a.y & 15) == 0) &&
         (((uintptr_t)a.wgt_bf & 15) == 0) && (!a.bias || (((uintptr_t)a.bias & 15) == 0));
}

template <int BM, int BN, int WTM, int WTN>
static void conv_bf16_launch(const ConvArgs& a, int maxM, hipStream_t s) {
  const bool abf = a.x_bf != nullptr && (((uintptr_t)a.x_bf & 15) == 0) && a.x_ps % 8 == 0;
  if (a.ksplit > 1) {
    dim3 grid((maxM + BM - 1) / BM, (a.N + BN - 1) / BN, a.nphase * a.ksplit);
    if (abf) hipLaunchKernelGGL((conv_gemm_bf16_kernel<BM, BN, WTM, WTN, true, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv_gemm_bf16_kernel<BM, BN, WTM, WTN, true, false>), grid, dim3(256), 0, s, a);
  } else {
    dim3 grid((maxM + BM - 1) / BM, (a.N + BN - 1) / BN, a.nphase);
    if (abf) hipLaunchKernelGGL((conv_gemm_bf16_kernel<BM, BN, WTM, WTN, false, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv_gemm_bf16_kernel<BM, BN, WTM, WTN, false, false>), grid, dim3(256), 0, s, a);
  }
}

const char* launch_conv_gemm_bf16(const ConvArgs& a, int maxM, bool big_tile, hipStream_t s) {
  if (big_tile) {
    conv_bf16_launch<128, 128, 64, 64>(a, maxM, s);
    return "conv_gemm_bf16<128,128>";
  }
  conv_bf16_launch<64, 64, 32, 32>(a, maxM, s);
  return "conv_gemm_bf16<64,64>";
}

// =============================================================================================
// wgrad_bf16_kernel
// =============================================================================================
template <int BM, int BN, int WTM, int WTN>
__global__ __launch_bounds__(256, 2) void wgrad_bf16_kernel(const WgradArgs a, float* __restrict__ out, int dense_out) {
  constexpr int WGN = BN / WTN;
  constexpr int MI = WTM / 16, NJ = WTN / 16;
  __shared__ __attribute__((aligned(16))) unsigned short At[2][BM * ROWB];  // [c][pixel], double-buffered
  __shared__ __attribute__((aligned(16))) unsigned short Bt[2][BN * ROWB];  // [n][pixel]
  __shared__ long long poff[2][256];   // gathered-pixel offsets, decoded 256 pixels (8 K steps) at a time

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int C = a.C, N = a.N, IH = a.IH, IW = a.IW;
  const int ntn = (N + BN - 1) / BN;
  const int c0 = (blockIdx.x / ntn) * BM, n0 = (blockIdx.x % ntn) * BN;
  const int tap = blockIdx.y;
  const int kh = tap / a.KW, kw = tap - kh * a.KW;
  const int M = a.NB * a.OH * a.OW;
  int chunk = (M + a.nsplit - 1) / a.nsplit;
  chunk = (chunk + BKB - 1) / BKB * BKB;
  const int p_lo = blockIdx.z * chunk;
  const int p_hi = min(M, p_lo + chunk);

  // unit = (pixel pair mp, channel quad cq).  A half wave is 8 consecutive quads (128 contiguous bytes of one
  // pixel) x 4 pixel pairs of ONE 8-pixel chunk; unit i of a thread adds 16 quads.
  constexpr int A_U = BM / 64, B_U = BN / 64;
  float4 ra0[A_U][2], rb0[B_U][2], ra1[A_U][2], rb1[B_U][2];   // two staging sets (tiles t+1, t+2 in flight)
  const int cq8 = tid & 7, mpl = (tid >> 3) & 3, rest = tid >> 5;
  const int kch = rest & 3;                        // 8-pixel chunk of the K step
  const int mp = kch * 4 + mpl;                    // pixel pair 0..15 of the 32-pixel K step
  const int cqb = (rest >> 2) * 8 + cq8;          // + 16 * i
  const int ex = cq8 & 1;                          // odd quads store their 4 rows in the order 1,0,3,2
  const unsigned exmask = 0u - (unsigned)ex;
  const int st_col = (kch ^ (((-(cq8 & 3)) & 3) ^ (cq8 >> 2))) * 4 + mpl;   // dword column, swizzled (lds_swz(4*cq))

  // every thread decodes one pixel of a 256-pixel block (8 K steps): the integer divisions leave the per-step path
  auto decode_block = [&](int blk) {
    const int m = p_lo + blk * 256 + tid;
    long long off = -1;
    if (m < p_hi) {
      const int ox = m % a.OW;
      const int t2 = m / a.OW;
      const int oy = t2 % a.OH;
      const int nb = t2 / a.OH;
      int iy = oy * a.S + kh - a.pad, ix = ox * a.S + kw - a.pad;
      if (a.reflect) {
        iy = reflect_idx_b(iy, IH);
        ix = reflect_idx_b(ix, IW);
      }
      if (iy >= 0 && iy < IH && ix >= 0 && ix < IW) off = (long long)((nb * IH + iy) * IW + ix) * a.x_ps;
    }
    poff[blk & 1][tid] = off;
  };
  auto load_tile = [&](int t, float4 (&ra)[A_U][2], float4 (&rb)[B_U][2]) {
    const long long* po = &poff[(t >> 3) & 1][(t & 7) * 32 + 2 * mp];
    const long long o0 = po[0], o1 = po[1];
    const int m = p_lo + t * BKB + 2 * mp;
    // (no `cond ? *ptr : zero`: that selects between a global and a stack ADDRESS and loads through a flat pointer)
#pragma unroll
    for (int i = 0; i < A_U; ++i) {
      const int c = c0 + (cqb + 16 * i) * 4;
      float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
      if (c < C) {
        if (o0 >= 0) v0 = *reinterpret_cast<const float4*>(a.x + o0 + c);
        if (o1 >= 0) v1 = *reinterpret_cast<const float4*>(a.x + o1 + c);
      }
      ra[i][0] = v0;
      ra[i][1] = v1;
    }
#pragma unroll
    for (int i = 0; i < B_U; ++i) {
      const int n = n0 + (cqb + 16 * i) * 4;
      float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
      if (n < N) {
        if (m < p_hi) v0 = *reinterpret_cast<const float4*>(a.dy + (size_t)m * a.dy_ps + n);
        if (m + 1 < p_hi) v1 = *reinterpret_cast<const float4*>(a.dy + (size_t)(m + 1) * a.dy_ps + n);
      }
      rb[i][0] = v0;
      rb[i][1] = v1;
    }
  };
  // 4 channels x 2 pixels -> 4 packed dwords into rows 4*cq + (j ^ ex), j = 0..3
  auto store_unit = [&](unsigned short* base, int cq, const float4& lo, const float4& hi) {
    // instruction j writes channel j ^ ex to row 4*cq + (j ^ ex); branch-free swap of the packed pairs (a select
    // on the float4 components makes the compiler index the staging registers through scratch memory)
    const unsigned p0 = pack_bf16(lo.x, hi.x), p1 = pack_bf16(lo.y, hi.y);
    const unsigned p2 = pack_bf16(lo.z, hi.z), p3 = pack_bf16(lo.w, hi.w);
    const unsigned s01 = (p0 ^ p1) & exmask, s23 = (p2 ^ p3) & exmask;
    const unsigned d0 = p0 ^ s01, d1 = p1 ^ s01, d2 = p2 ^ s23, d3 = p3 ^ s23;
    unsigned* q = reinterpret_cast<unsigned*>(base) + (size_t)(4 * cq) * (ROWB / 2) + st_col;
    q[(0 ^ ex) * (ROWB / 2)] = d0;
    q[(1 ^ ex) * (ROWB / 2)] = d1;
    q[(2 ^ ex) * (ROWB / 2)] = d2;
    q[(3 ^ ex) * (ROWB / 2)] = d3;
  };
  auto store_tile = [&](const float4 (&ra)[A_U][2], const float4 (&rb)[B_U][2], int buf) {
#pragma unroll
    for (int i = 0; i < A_U; ++i) store_unit(At[buf], cqb + 16 * i, ra[i][0], ra[i][1]);
#pragma unroll
    for (int i = 0; i < B_U; ++i) store_unit(Bt[buf], cqb + 16 * i, rb[i][0], rb[i][1]);
  };

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int l15 = lane & 15, lq = lane >> 4;
  const int fsw = (-(l15 >> 2)) & 3;
  static_assert((WTM / 16) % 2 == 0 && (WTN / 16) % 2 == 0, "wave sub-tiles start at even 16-row blocks");
  auto compute = [&](int buf) {
    bf16x8 af[MI], bf[NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
      af[i] = *reinterpret_cast<const bf16x8*>(&At[buf][(wm * WTM + i * 16 + l15) * ROWB + (lq ^ fsw ^ (i & 1)) * 8]);
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      bf[j] = *reinterpret_cast<const bf16x8*>(&Bt[buf][(wn * WTN + j * 16 + l15) * ROWB + (lq ^ fsw ^ (j & 1)) * 8]);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
  };

  // step t: multiply tile t (LDS buffer t&1), move tile t+1 from its staging set to the other buffer, refill that
  // set with tile t+3 (even part) / t+4 (odd part).  Offset block b (tiles 8b..8b+7) lives in half b&1, whose previous
  // tenant b-2 is last read by the load of tile 8b-9 in the even part of step 8b-12; block b is decoded at step 8b-10
  // (after that step's barriers) and first read by the load of tile 8b in the odd part of step 8b-4.
  const int nt = (p_hi - p_lo + BKB - 1) / BKB;
  if (nt > 0) {
    decode_block(0);
    if (nt > 8) decode_block(1);
    __syncthreads();
    load_tile(0, ra0, rb0);
    if (nt > 1) load_tile(1, ra1, rb1);
    store_tile(ra0, rb0, 0);
    if (nt > 2) load_tile(2, ra0, rb0);
    __syncthreads();
    for (int t = 0; t < nt; t += 2) {
      if (t + 1 < nt) {
        store_tile(ra1, rb1, 1);
        if (t + 3 < nt) load_tile(t + 3, ra1, rb1);
      }
      if ((t & 7) == 6 && (t >> 3) + 2 < (nt + 7) / 8) decode_block((t >> 3) + 2);
      compute(0);
      __syncthreads();
      if (t + 1 >= nt) break;
      if (t + 2 < nt) {
        store_tile(ra0, rb0, 0);
        if (t + 4 < nt) load_tile(t + 4, ra0, rb0);
      }
      compute(1);
      __syncthreads();
    }
  }

  const int T = a.KH * a.KW;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int c = c0 + wm * WTM + i * 16 + l15;
    if (c >= C) continue;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = n0 + wn * WTN + j * 16 + lq * 4;
      if (n >= N) continue;
      float* q = dense_out ? out + (((size_t)blockIdx.z * T + tap) * C + c) * N + n
                           : out + (size_t)tap * a.w_ts + (size_t)c * a.ldw + n;
      float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      if (!dense_out && a.accumulate) {
        const float4 o = *reinterpret_cast<const float4*>(q);
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
      }
      *reinterpret_cast<float4*>(q) = v;
    }
  }
}

const char* launch_wgrad_bf16(const WgradArgs& a, float* out, int dense, bool big_tile, hipStream_t s) {
  if (big_tile) {
    dim3 grid(((a.C + 127) / 128) * ((a.N + 127) / 128), a.KH * a.KW, a.nsplit);
    hipLaunchKernelGGL((wgrad_bf16_kernel<128, 128, 64, 64>), grid, dim3(256), 0, s, a, out, dense);
    return "wgrad_bf16<128,128>";
  }
  dim3 grid(((a.C + 63) / 64) * ((a.N + 63) / 64), a.KH * a.KW, a.nsplit);
  hipLaunchKernelGGL((wgrad_bf16_kernel<64, 64, 32, 32>), grid, dim3(256), 0, s, a, out, dense);
  return "wgrad_bf16<64,64>";
}

// =============================================================================================
// bf16 filter copies: [T][R][Cc] fp32 -> same layout bf16, or per-tap transposed [T][Cc][R] bf16
// =============================================================================================
__global__ void cvt_bf16_kernel(const float* __restrict__ in, unsigned short* __restrict__ out, size_t n4) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(in)[i];
    reinterpret_cast<uint2*>(out)[i] = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
  }
}
// the inverse (exact): a bf16 plane widened to fp32 -- for the rare reader of a tensor that was only ever stored as its plane
__global__ void cvt_from_bf16_kernel(const unsigned short* __restrict__ in, float* __restrict__ out, size_t n4) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const uint2 v = reinterpret_cast<const uint2*>(in)[i];
    reinterpret_cast<float4*>(out)[i] = make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u),
                                                     __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
  }
}
__global__ void tap_transpose_bf16_kernel(const float* __restrict__ in, unsigned short* __restrict__ out, int R, int Cc) {
  __shared__ float tile[32][33];
  const int t = blockIdx.z;
  const float* src = in + (size_t)t * R * Cc;
  unsigned short* dst = out + (size_t)t * R * Cc;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + threadIdx.x;
    tile[i][threadIdx.x] = (r < R && c < Cc) ? src[(size_t)r * Cc + c] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + threadIdx.x;
    if (r < R && c < Cc) dst[(size_t)c * R + r] = (unsigned short)(pack_bf16(tile[threadIdx.x][i], 0.f) & 0xffffu);
  }
}
void launch_cvt_bf16(const float* in, unsigned short* out, size_t n, hipStream_t s) {
  const size_t n4 = n / 4;  // callers guarantee n % 4 == 0 (K extent % 32 == 0)
  size_t blocks = (n4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(cvt_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, in, out, n4);
}
__global__ void act_bwd_bf_kernel(const unsigned short* __restrict__ out_bf, const float* __restrict__ dout, float* __restrict__ dx,
                                  unsigned short* __restrict__ dx_bf, size_t n4, int act) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const uint2 q = reinterpret_cast<const uint2*>(out_bf)[i];
    const float o[4] = {__builtin_bit_cast(float, q.x << 16), __builtin_bit_cast(float, q.x & 0xffff0000u),
                        __builtin_bit_cast(float, q.y << 16), __builtin_bit_cast(float, q.y & 0xffff0000u)};
    const float4 d = reinterpret_cast<const float4*>(dout)[i];
    const float dv[4] = {d.x, d.y, d.z, d.w};
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)   // slopes as act_grad_from_out (k_norm.hip): ReLU o > 0 ? 1 : 0, LeakyReLU o >= 0 ? 1 : 0.2
      r[j] = dv[j] * (act == ACT_LRELU ? (o[j] >= 0.f ? 1.f : 0.2f) : (o[j] > 0.f ? 1.f : 0.f));
    if (dx) reinterpret_cast<float4*>(dx)[i] = make_float4(r[0], r[1], r[2], r[3]);
    if (dx_bf) {
      typedef __bf16 bf16x2_a __attribute__((ext_vector_type(2)));
      typedef float f32x2_a __attribute__((ext_vector_type(2)));
      const f32x2_a lo = {r[0], r[1]}, hi = {r[2], r[3]};
      uint2 w;
      w.x = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2_a));
      w.y = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2_a));
      reinterpret_cast<uint2*>(dx_bf)[i] = w;
    }
  }
}
void launch_act_bwd_bf(const unsigned short* out_bf, const float* dout, float* dx, unsigned short* dx_bf, size_t n, int act, hipStream_t s) {
  EG_CHECK(n % 4 == 0 && (act == ACT_RELU || act == ACT_LRELU), "act_bwd_bf: n %% 4 == 0, ReLU / LeakyReLU");
  const size_t n4 = n / 4;
  size_t blocks = (n4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(act_bwd_bf_kernel, dim3((unsigned)blocks), dim3(256), 0, s, out_bf, dout, dx, dx_bf, n4, act);
}
void launch_cvt_from_bf16(const unsigned short* in, float* out, size_t n, hipStream_t s) {   // n % 4 == 0
  const size_t n4 = n / 4;
  size_t blocks = (n4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(cvt_from_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, in, out, n4);
}
void launch_tap_transpose_bf16(const float* in, unsigned short* out, int T, int R, int Cc, hipStream_t s) {
  dim3 grid((Cc + 31) / 32, (R + 31) / 32, T);
  hipLaunchKernelGGL(tap_transpose_bf16_kernel, grid, dim3(32, 8), 0, s, in, out, R, Cc);
}

}  // namespace eg
