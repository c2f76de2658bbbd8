// GEMM-shaped kernels for the EdgeGAN conv / transposed-conv stack on gfx950 (MI355X).
//
// All dense contractions of the reference (tf.nn.conv2d, tf.nn.conv2d_transpose and their
// gradients: edgegan/nn/modules/conv.py:29,49-53,278-285 and the tf.gradients of those ops) are
// expressed as ONE tap-gather implicit GEMM on the exact-fp32 matrix instruction
// v_mfma_f32_16x16x4_f32 (157.3 TFLOP/s peak, bitwise an fmaf chain): no im2col buffer, the input
// tile is gathered straight from NHWC into LDS, stride-2 transposed convolutions are decomposed into
// their 4 sub-pixel phases so no multiplications by structural zeros are issued.
//
//   conv_gemm_kernel : Y[pix_out(m)][n] (+)= b[n] + sum_taps sum_c X[pix_in(m,tap)][c] Wt[tap][c][n]
//   wgrad_kernel     : dW[tap][c][n]      = sum_m X[pix_in(m,tap)][c] dY[m][n]   (split over m)
//
// Tiles: 256 threads = 4 waves; block tile BMxBN, K-step 16; each wave owns a WTMxWTN sub-tile as
// (WTM/16)x(WTN/16) 16x16 accumulators.  Global->register prefetch of the next K tile overlaps the
// MFMA work on the current one (register-staged, write-after-barrier).
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace eg {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int reflect_idx(int i, int n) {
  i = i < 0 ? -i : i;
  return i >= n ? 2 * (n - 1) - i : i;
}

// =============================================================================================
// conv_gemm_kernel
// =============================================================================================
template <int BM, int BN, int WTM, int WTN, int BK, bool VEC, bool BVEC, bool SPLITK = false>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const ConvArgs a) {
  constexpr int LDA = BK + 4;
  constexpr int LDB = BN + ((BN % 32 == 0) ? 16 : 32);
  constexpr int WGN = BN / WTN;
  constexpr int MI = WTM / 16, NJ = WTN / 16;
  static_assert((BM / WTM) * (BN / WTN) == 4, "4 waves");
  static_assert(VEC || (!BVEC && BK == 16), "flat-K path: scalar loads, K step 16");
  __shared__ __attribute__((aligned(16))) float As[BM * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[BK * LDB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int zphase = SPLITK ? (int)blockIdx.z / a.ksplit : (int)blockIdx.z;
  const int zsplit = SPLITK ? (int)blockIdx.z - zphase * a.ksplit : 0;
  const ConvPhase ph = a.ph[zphase];
  const int M = a.NB * ph.OHp * ph.OWp;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  if (m0 >= M) return;
  const int IH = a.IH, IW = a.IW, C = a.C, N = a.N;

  // ---- per-thread A-row bookkeeping --------------------------------------------------------
  constexpr int Q = BK / 4;                        // float4 per A row (VEC)
  constexpr int A_V = (BM * Q) / 256;              // float4 per thread (VEC)
  constexpr int B_V = (BK * BN / 4 + 255) / 256;   // float4 per thread (BVEC)
  constexpr int A_S = BM / 16;                     // scalars per thread (FLAT)
  constexpr int B_S = (BK / 16) * (BN / 16);       // scalars per thread (!BVEC)
  constexpr int A_ROWS = VEC ? A_V : A_S;
  int r_iy0[A_ROWS], r_ix0[A_ROWS], r_base[A_ROWS];
#pragma unroll
  for (int i = 0; i < A_ROWS; ++i) {
    const int r = VEC ? (tid / Q + (256 / Q) * i) : ((tid >> 4) + 16 * i);
    const int m = m0 + r;
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
  const int Kflat = ntaps * C;
  const int nk = VEC ? ntaps * (C / BK) : (Kflat + BK - 1) / BK;

  float4 ra4[VEC ? A_V : 1], rb4[BVEC ? B_V : 1];
  float ras[VEC ? 1 : A_S], rbs[BVEC ? 1 : B_S];

  auto gather_ok = [&](int base, int& iy, int& ix) -> bool {
    if (a.reflect) {
      iy = reflect_idx(iy, IH);
      ix = reflect_idx(ix, IW);
    }
    return base >= 0 && iy >= 0 && iy < IH && ix >= 0 && ix < IW;
  };

  // position of the NEXT tile to load in the tap-tiled K order (no divisions in the loop)
  int nx_ty = 0, nx_tx = 0, nx_c0 = 0;
  int kt_lo = 0, kt_hi = nk;
  if constexpr (SPLITK) {  // this block owns K tiles [kt_lo, kt_hi) and writes a partial slab
    const int per = (nk + a.ksplit - 1) / a.ksplit;
    kt_lo = zsplit * per;
    kt_hi = min(nk, kt_lo + per);
    const int cpt = C / BK;
    const int tap0 = kt_lo / cpt;
    nx_c0 = (kt_lo - tap0 * cpt) * BK;
    nx_ty = tap0 / ph.Tx;
    nx_tx = tap0 - nx_ty * ph.Tx;
  }

  auto load_tile = [&](int kt) {
    if constexpr (VEC) {
      const int ty = nx_ty, tx = nx_tx, c0 = nx_c0;
      const long long wt = (long long)((ph.kh0 + ty * a.kst) * a.KW + (ph.kw0 + tx * a.kst)) * a.w_ts;
      const int kq = tid % Q;
#pragma unroll
      for (int i = 0; i < A_V; ++i) {
        int iy = r_iy0[i] + ty * a.D, ix = r_ix0[i] + tx * a.D;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gather_ok(r_base[i], iy, ix))
          v = *reinterpret_cast<const float4*>(a.x + (size_t)(r_base[i] + iy * IW + ix) * a.x_ps + c0 + kq * 4);
        ra4[i] = v;
      }
      if constexpr (BVEC) {
#pragma unroll
        for (int i = 0; i < B_V; ++i) {
          const int idx = tid + 256 * i;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (idx < BK * BN / 4) {
            const int k = idx / (BN / 4), n = n0 + (idx % (BN / 4)) * 4;
            if (n < N) v = *reinterpret_cast<const float4*>(a.wgt + wt + (size_t)(c0 + k) * a.ldw + n);
          }
          rb4[i] = v;
        }
      } else {  // 16-byte A gathers, scalar filter loads (Cout not a multiple of 4, e.g. the 3-channel image)
#pragma unroll
        for (int j = 0; j < B_S; ++j) {
          const int kk = (tid >> 4) + 16 * (j / (BN / 16));
          const int n = n0 + (tid & 15) + 16 * (j % (BN / 16));
          rbs[j] = n < N ? a.wgt[wt + (size_t)(c0 + kk) * a.ldw + n] : 0.f;
        }
      }
      nx_c0 += BK;
      if (nx_c0 >= C) {
        nx_c0 = 0;
        if (++nx_tx == ph.Tx) {
          nx_tx = 0;
          ++nx_ty;
        }
      }
    } else {
      {  // A: this thread's k column is fixed inside the tile
        const int k = kt * BK + (tid & 15);
        const bool kv = k < Kflat;
        const int tap = kv ? k / C : 0;
        const int c = k - tap * C;
        const int ty = tap / ph.Tx, tx = tap - ty * ph.Tx;
#pragma unroll
        for (int i = 0; i < A_S; ++i) {
          int iy = r_iy0[i] + ty * a.D, ix = r_ix0[i] + tx * a.D;
          float v = 0.f;
          if (kv && gather_ok(r_base[i], iy, ix)) v = a.x[(size_t)(r_base[i] + iy * IW + ix) * a.x_ps + c];
          ras[i] = v;
        }
      }
      {  // B: this thread's k row is fixed inside the tile
        const int k = kt * BK + (tid >> 4);
        const bool kv = k < Kflat;
        const int tap = kv ? k / C : 0;
        const int c = k - tap * C;
        const int ty = tap / ph.Tx, tx = tap - ty * ph.Tx;
        const long long wt = (long long)((ph.kh0 + ty * a.kst) * a.KW + (ph.kw0 + tx * a.kst)) * a.w_ts;
#pragma unroll
        for (int j = 0; j < B_S; ++j) {
          const int n = n0 + (tid & 15) + 16 * j;
          rbs[j] = (kv && n < N) ? a.wgt[wt + (size_t)c * a.ldw + n] : 0.f;
        }
      }
    }
  };

  auto store_tile = [&]() {
    if constexpr (VEC) {
#pragma unroll
      for (int i = 0; i < A_V; ++i)
        *reinterpret_cast<float4*>(&As[(tid / Q + (256 / Q) * i) * LDA + (tid % Q) * 4]) = ra4[i];
    } else {
#pragma unroll
      for (int i = 0; i < A_S; ++i) As[((tid >> 4) + 16 * i) * LDA + (tid & 15)] = ras[i];
    }
    if constexpr (BVEC) {
#pragma unroll
      for (int i = 0; i < B_V; ++i) {
        const int idx = tid + 256 * i;
        if (idx < BK * BN / 4) *reinterpret_cast<float4*>(&Bs[(idx / (BN / 4)) * LDB + (idx % (BN / 4)) * 4]) = rb4[i];
      }
    } else {
#pragma unroll
      for (int j = 0; j < B_S; ++j)
        Bs[((tid >> 4) + 16 * (j / (BN / 16))) * LDB + (tid & 15) + 16 * (j % (BN / 16))] = rbs[j];
    }
  };

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int l15 = lane & 15, lq = lane >> 4;
  if (kt_lo < kt_hi) load_tile(kt_lo);
  for (int kt = kt_lo; kt < kt_hi; ++kt) {
    store_tile();
    __syncthreads();
    if (kt + 1 < kt_hi) load_tile(kt + 1);
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      float af[MI], bf[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = As[(wm * WTM + i * 16 + l15) * LDA + kk * 4 + lq];
#pragma unroll
      for (int j = 0; j < NJ; ++j) bf[j] = Bs[(kk * 4 + lq) * LDB + wn * WTN + j * 16 + l15];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j], af[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue: the MFMA operands are swapped (D^T = B^T A^T), so lane l holds, for pixel row lane&15, the 4
  // consecutive output channels (lane>>4)*4 + reg: one 16-byte store per accumulator instead of four scalars
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m0 + wm * WTM + i * 16 + l15;
    if (m >= M) continue;
    const int ox = m % ph.OWp;
    const int t2 = m / ph.OWp;
    const int oy = t2 % ph.OHp;
    const int nb = t2 / ph.OHp;
    if constexpr (BN == 16) {
      if (a.shuf) {   // phase-merged stride-2 layer: 12 channels -> a 2x2 block of 3-channel output pixels
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = lq * 4 + r;
          if (n < 12) {
            const int p = n / 3, ch = n - 3 * p;
            float* q = a.y + ((size_t)(nb * a.OHf + 2 * oy + (p >> 1)) * a.OWf + 2 * ox + (p & 1)) * a.y_ps + ch;
            float v = acc[i][0][r];
            if (a.bias) v += a.bias[ch];
            if (a.accumulate) v += *q;
            *q = v;
          }
        }
        continue;
      }
    }
    float* yrow = (SPLITK ? a.ws + (size_t)zsplit * a.y_total : a.y) +
                  ((size_t)(nb * a.OHf + oy * a.os + ph.ophy) * a.OWf + ox * a.os + ph.ophx) * a.y_ps;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = n0 + wn * WTN + j * 16 + lq * 4;
      if constexpr (BVEC) {   // N % 4 == 0: whole float4 in range or not at all
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
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (n + r < N) {
            float v = acc[i][j][r];
            if (a.bias) v += a.bias[n + r];
            if (a.accumulate) v += yrow[n + r];
            yrow[n + r] = v;
          }
        }
      }
    }
  }
}

template <int BM, int BN, int WTM, int WTN, int BK, bool VEC, bool BVEC>
static void conv_launch(const ConvArgs& a, int maxM, hipStream_t s) {
  if constexpr (VEC && BVEC) {
    if (a.ksplit > 1) {
      dim3 grid((maxM + BM - 1) / BM, (a.N + BN - 1) / BN, a.nphase * a.ksplit);
      hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WTM, WTN, BK, true, true, true>), grid, dim3(256), 0, s, a);
      return;
    }
  }
  dim3 grid((maxM + BM - 1) / BM, (a.N + BN - 1) / BN, a.nphase);
  hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WTM, WTN, BK, VEC, BVEC, false>), grid, dim3(256), 0, s, a);
}

// y (+)= bias + sum over the K-split slabs (fixed order).  float4 form: the split-K kernels only exist for N % 4 == 0,
// so a quad never straddles a pixel row; y_ps4 / N4 in float4 units, 32-bit index arithmetic.
__global__ void conv_split_reduce_kernel(const float* __restrict__ ws, float* __restrict__ y, const float* __restrict__ bias,
                                         unsigned total4, int y_ps4, int N4, int ksplit, int accumulate) {
  const float4* w4 = reinterpret_cast<const float4*>(ws);
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += gridDim.x * blockDim.x) {
    const int n4 = y_ps4 == N4 ? (int)(i % (unsigned)N4) : (int)(i % (unsigned)y_ps4);
    if (n4 >= N4) continue;
    float4 v = bias ? reinterpret_cast<const float4*>(bias)[n4] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < ksplit; ++k) {
      const float4 p = w4[(size_t)k * total4 + i];
      v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    float4* o = reinterpret_cast<float4*>(y) + i;
    if (accumulate) {
      const float4 q = *o;
      v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
    }
    *o = v;
  }
}

void launch_conv_split_reduce(const float* ws, float* y, const float* bias, size_t total, int y_ps, int N, int ksplit,
                              int accumulate, hipStream_t s) {
  const size_t total4 = total / 4;   // split-K requires N % 4 == 0 and a 16-byte aligned, y_ps % 4 == 0 output
  size_t blocks = (total4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(conv_split_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, ws, y, bias, (unsigned)total4, y_ps / 4,
                     N / 4, ksplit, accumulate);
}

// K-split factor for launches that cannot fill the 256 CUs with output tiles (small M x N, long K)
// Limb-mode (k_gemm_x3.hip) launches: tile family and K split from a small cost model.  Unit of time = one K step
// (32 channels of one tap) of a 128x128 tile at the ping-pong kernel's rate (~1.3 us per CU).  The ping-pong block
// (two 128x128 tiles, one block per CU) advances 2 tiles per 2 units; two 64x64 blocks share a CU at ~0.66 of that
// kernel's per-flop rate; a K split adds the slab reduction (HBM-bound) and needs >= 4 K steps per slab.
static int pick_x3(ConvArgs& a) {
  int maxM = 0, nk = 1 << 30, nk_max = 0;   // nk: K steps of the shortest phase (split bound); nk_max: of the longest (time)
  for (int p = 0; p < a.nphase; ++p) {
    maxM = std::max(maxM, a.NB * a.ph[p].OHp * a.ph[p].OWp);
    nk = std::min(nk, a.ph[p].Ty * a.ph[p].Tx * (a.C / 32));
    nk_max = std::max(nk_max, a.ph[p].Ty * a.ph[p].Tx * (a.C / 32));
  }
  static const bool by_longest = dbg_env("EG_X3_COST_MIN_PHASE") == nullptr;   // A/B: the model's former step count
  const int nk_t = by_longest ? nk_max : nk;
  static const bool chunk_split_on = dbg_env("EG_X3_COST_STEP_SPLIT") == nullptr;   // A/B: the model's former step-granular split
  const int chunks = (a.S == 2 ? 4 : 1) * (a.C / 32);
  const bool chunk_split = chunk_split_on && chunks > 0 && nk_t % chunks == 0;
  const long long tpp = (long long)((maxM + 255) / 256) * ((a.N + 127) / 128) * a.nphase;
  const long long t64 = (long long)((maxM + 63) / 64) * ((a.N + 63) / 64) * a.nphase;
  const bool pp_ok = a.N >= 128 && a.x_bytes != 0 && a.wbf_bytes != 0;
  const double out_bytes = 4.0 * (double)maxM * a.nphase * a.N;
  double best = 1e300;
  int best_ks = 1, best_tile = 2;
  for (int tile = pp_ok ? 1 : 2; tile <= (a.limbs == 1 ? 1 : 2); ++tile)
    for (int ks = 1; ks <= 8; ++ks) {
      if (ks > 1 && nk / ks < 4) break;
      const long long blocks = (tile == 1 ? tpp : t64) * ks;
      const long long slots = tile == 1 ? 256 : 512;
      double steps = (double)((nk_t + ks - 1) / ks);   // the launch lasts as long as its longest phase
      if (tile == 1 && chunk_split) {
        // the halo kernel splits K by whole 32-channel chunks (S2D: by the 4 C / 32 phase chunks): 16 chunks over 5 slabs
        // are 4 + 4 + 4 + 4 + 0 -- price what will run, and never an empty slab
        const int per = (chunks + ks - 1) / ks;
        if (per * (ks - 1) >= chunks && ks > 1) continue;
        steps = (double)per * (nk_t / chunks);
      }
      double t = (double)((blocks + slots - 1) / slots) * steps * (tile == 1 ? 2.0 : 0.76) + (tile == 1 ? 3.0 : 1.0);
      static const double slab_w = dbg_env("EG_X3_SLAB_COST") ? atof(dbg_env("EG_X3_SLAB_COST")) : 1.0;
      if (ks > 1) t += slab_w * ((ks + 1) * out_bytes / 3.0e12 / 1.3e-6 + 4.0);   // slab reduction + its launch
      if (t < best) {
        best = t;
        best_ks = ks;
        best_tile = tile;
      }
    }
  a.x3_tile = best_tile;
  return best_ks;
}

// Measured (tile family, K split) of the f32x3 convolution launches whose best choice the cost model misses: one row per launch
// shape of the iteration (BASELINE configs[2], B = 64), generated by tools/tune_x3.py from per-layer HIP-event timings of every
// candidate on MI355X (profiles/r05*_tune_*).  Shapes that are not listed -- other batch sizes, other resolutions -- use the model.
struct X3Tuned {
  int M, N, C, taps, nphase, S, tile, ks;
};
static const X3Tuned kX3Tuned[] = {
#include "x3_tuned.h"
    {0, 0, 0, 0, 0, 0, 0, 0}};

int conv_pick_ksplit(ConvArgs& a) {
  if (a.N <= 16 || a.C % 32 != 0 || a.N % 4 != 0) return 1;
  if (a.limbs == 3 && a.wgt_bf) {
    int maxM = 0, taps = 0;
    for (int p = 0; p < a.nphase; ++p) {
      maxM = std::max(maxM, a.NB * a.ph[p].OHp * a.ph[p].OWp);
      taps += a.ph[p].Ty * a.ph[p].Tx;
    }
#ifdef EG_EXPERIMENTS
    // tools/tune_x3.py: "tile,ks" forces every limb-mode launch onto one candidate (clipped to what the shape allows)
    static const char* force = dbg_env("EG_X3_FORCE");
    if (force) {
      int ft = 0, fk = 1;
      if (sscanf(force, "%d,%d", &ft, &fk) == 2 && (ft == 1 || ft == 2)) {
        int nk = 1 << 30;
        for (int p = 0; p < a.nphase; ++p) nk = std::min(nk, a.ph[p].Ty * a.ph[p].Tx * (a.C / 32));
        while (fk > 1 && nk / fk < 4) --fk;
        if (ft == 1 && !(a.N >= 64 && a.x_bytes && a.wbf_bytes)) ft = 2;
        if (ft == 1 && a.N < 128) {   // (below 128 output channels only the halo kernel has a 256-pixel form)
          ConvArgs b = a;
          b.x3_tile = 1;
          b.ksplit = fk;
          if (!conv_x3h_eligible(b)) ft = 2;
        }
        a.x3_tile = ft;
        a.tuned = 1;
        return fk;
      }
    }
    static const bool no_table = dbg_env("EG_X3_NO_TABLE") != nullptr;
    if (!no_table)
#endif
    for (const X3Tuned* t = kX3Tuned; t->M; ++t)
      if (t->M == maxM && t->N == a.N && t->C == a.C && t->taps == taps && t->nphase == a.nphase && t->S == a.S) {
        a.x3_tile = t->tile;
        a.tuned = 1;
        return t->ks;
      }
  }
  if ((a.limbs == 3 || a.limbs == 1) && a.wgt_bf) return pick_x3(a);
  int maxM = 0, minTiles = 1 << 30;
  for (int p = 0; p < a.nphase; ++p) {
    maxM = std::max(maxM, a.NB * a.ph[p].OHp * a.ph[p].OWp);
    minTiles = std::min(minTiles, a.ph[p].Ty * a.ph[p].Tx * (a.C / 32));
  }
  const long long t64 = (long long)((maxM + 63) / 64) * ((a.N + 63) / 64) * a.nphase;
  // Split K whenever the output tiles alone would leave CUs short of resident blocks: the target, 3072 64x64-tile
  // equivalents = 768 128x128 blocks, is 3 resident blocks on each of the 256 CUs -- what the register budget of the
  // 128x128 kernel allows -- so a wave waiting on its loads always has MFMA work from another block next to it.
  // (Measured: threshold/target 384/512 -> 2048/3072 = +2.5 % on the iteration; the 4096-pixel layers of the
  // classifier's last unit and the discriminators' conv4 move from 64x64 tiles to 128x128 tiles x 3-4 K slabs.)
  // The bf16 kernels multiply 16x faster, the slab reduction does not: they keep the conservative rule (measured).
  const int thr = a.wgt_bf ? 384 : 2048, tgt = a.wgt_bf ? 512 : 3072;
  if (t64 >= thr) return 1;
  int ks = (int)((tgt + t64 - 1) / t64);
  ks = std::min(ks, 8);
  ks = std::min(ks, minTiles / 4);   // at least 4 K tiles (128 of K) per split
  return ks < 2 ? 1 : ks;
}

// k32: C % 32 == 0 -> K step 32 (half the barriers / address arithmetic per MFMA)
template <int BM, int BN, int WTM, int WTN>
static const char* conv_dispatch(const ConvArgs& a, int maxM, bool avec, bool bvec, hipStream_t s, const char* nv,
                                 const char* nav, const char* nf) {
  const bool k32 = (a.C % 32 == 0);
  if (avec && bvec) {
    if (k32) conv_launch<BM, BN, WTM, WTN, 32, true, true>(a, maxM, s);
    else conv_launch<BM, BN, WTM, WTN, 16, true, true>(a, maxM, s);
    return nv;
  }
  if (avec) {
    if (k32) conv_launch<BM, BN, WTM, WTN, 32, true, false>(a, maxM, s);
    else conv_launch<BM, BN, WTM, WTN, 16, true, false>(a, maxM, s);
    return nav;
  }
  conv_launch<BM, BN, WTM, WTN, 16, false, false>(a, maxM, s);
  return nf;
}

// the tile choice of the 17+-output-channel kernels: by CU load balance -- the MFMA pipe is the bottleneck, so what matters is
// how evenly the tiles divide over the 256 CUs (measured: 4 vs 3 resident blocks per CU changes nothing), times the per-tile
// efficiency of each configuration (64x64 runs at ~0.8 of the 128x128 tile's rate)
static bool conv_big_tile(const ConvArgs& a, int maxM) {
  const long long blocks128 = (long long)((maxM + 127) / 128) * ((a.N + 127) / 128) * a.nphase;
  const int ksp = a.ksplit > 1 ? a.ksplit : 1;
  const long long t128 = blocks128 * ksp;
  const long long t64 = (long long)((maxM + 63) / 64) * ((a.N + 63) / 64) * a.nphase * ksp;
  auto balance = [](long long tiles) {
    const long long per_cu = (tiles + 255) / 256;
    return (double)tiles / (double)(per_cu * 256);
  };
  const double e128 = a.N >= 128 ? 0.95 * balance(t128) : 0.0;
  const double e64 = 0.80 * balance(t64);
  return e128 >= e64;
}
static int conv_max_m(const ConvArgs& a) {
  int maxM = 0;
  for (int p = 0; p < a.nphase; ++p) maxM = std::max(maxM, a.NB * a.ph[p].OHp * a.ph[p].OWp);
  return maxM;
}
// only the fully vectorised kernels have a split-K form
static void conv_fix_ksplit(ConvArgs& a) {
  const bool avec = (a.C % 16 == 0) && (a.x_ps % 4 == 0) && (((uintptr_t)a.x & 15) == 0);
  const bool bvec = (a.N % 4 == 0) && (a.ldw % 4 == 0) && (a.w_ts % 4 == 0) && (((uintptr_t)a.wgt & 15) == 0);
  if (!(avec && bvec) || a.N <= 16 || !a.ws || a.y_ps % 4 != 0 || (((uintptr_t)a.y | (uintptr_t)a.bias) & 15) != 0 ||
      a.y_total / 4 >= (1ull << 32))
    a.ksplit = 1;
}
// ---------------------------------------------------------------------------------------------------------------------
// rows_linear_kernel (round 6): y[m][n] (+)= b[n] + sum_k x[m][k] w[k][n] for a HANDFUL of rows (m <= 64: the batch) with a
// short contraction (k <= 128) and many columns -- the generators' first linear z -> 8,192 (generator.py:45-52), the input
// gradients of FC8_mu / FC8_sigma and of the classifier's class head (linear.py:10-31,34-76).  The 64 x 64-tile gather kernel
// spends 15 - 22 us on them: 2 K steps of a pipeline built for thousands, one M tile.  Here a block owns 32 rows x 64 columns;
// x's tile sits in LDS transposed ([k][row], read as broadcast quads), a thread owns one column and the k = kq (mod 4) quarter
// of the contraction for all 32 rows (32 fp32 FMA chains, the filter row loads four deep in flight), and the four quarters
// are summed in LDS in the order kq = 0 .. 3.  fp32 FMAs: exact products, one rounding per step -- the fp32 MFMA path's
// arithmetic in another order (parity: the op tests' linear shapes, tests/parity.py).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kRowsLinC = 128, kRowsLinPitch = 36;
__global__ __launch_bounds__(256) void rows_linear_kernel(const float* __restrict__ x, int x_ps, const float* __restrict__ w, int ldw,
                                                          const float* __restrict__ bias, float* __restrict__ y, int y_ps, int M, int C,
                                                          int N, int accumulate) {
  __shared__ __attribute__((aligned(16))) float xs[kRowsLinC * kRowsLinPitch];
  __shared__ float red[4 * 32 * 64];
  const int tid = threadIdx.x, lane = tid & 63, kq = tid >> 6;
  const int n0 = blockIdx.x * 64, m0 = blockIdx.y * 32;
  const int col = n0 + lane;
  const bool cv = col < N;
  const float* wc = w + (cv ? col : 0);
  // every global load of the block is issued before the first use: the thread's <= 32 filter elements (k = kq, kq + 4, ...)
  // and its 16 elements of x's tile -- two memory latencies in the whole kernel instead of one per K step
  float wv[kRowsLinC / 4];
#pragma unroll
  for (int j = 0; j < kRowsLinC / 4; ++j) {
    const int k = kq + 4 * j;
    wv[j] = (k < C && cv) ? wc[(size_t)k * ldw] : 0.f;
  }
  {
    const int r = tid >> 3, k8 = tid & 7;
    const float* xr = x + (size_t)(m0 + r) * x_ps;
    const bool rv = m0 + r < M;
    float xv[kRowsLinC / 8];
#pragma unroll
    for (int j = 0; j < kRowsLinC / 8; ++j) {
      const int k = k8 + 8 * j;
      xv[j] = (rv && k < C) ? xr[k] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < kRowsLinC / 8; ++j) {
      const int k = k8 + 8 * j;
      if (k < C) xs[k * kRowsLinPitch + r] = xv[j];
    }
  }
  __syncthreads();
  float acc[32];
#pragma unroll
  for (int r = 0; r < 32; ++r) acc[r] = 0.f;
#pragma unroll
  for (int j = 0; j < kRowsLinC / 4; ++j) {
    const int k = kq + 4 * j;
    if (k < C) {   // (uniform over the wave)
      const float4* xr = reinterpret_cast<const float4*>(xs + k * kRowsLinPitch);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 v = xr[q];
        acc[4 * q] += v.x * wv[j]; acc[4 * q + 1] += v.y * wv[j]; acc[4 * q + 2] += v.z * wv[j]; acc[4 * q + 3] += v.w * wv[j];
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 32; ++r) red[(kq * 32 + r) * 64 + lane] = acc[r];
  __syncthreads();
  for (int o = tid; o < 32 * 64; o += 256) {
    const int r = o >> 6, c = o & 63;
    if (m0 + r < M && n0 + c < N) {
      float v = ((red[r * 64 + c] + red[(32 + r) * 64 + c]) + red[(64 + r) * 64 + c]) + red[(96 + r) * 64 + c];
      if (bias) v += bias[n0 + c];
      float* yp = y + (size_t)(m0 + r) * y_ps + n0 + c;
      *yp = accumulate ? *yp + v : v;
    }
  }
}
static bool rows_linear_eligible(const ConvArgs& a, int maxM) {
  static const bool off = dbg_env("EG_NO_ROWS_LINEAR") != nullptr;
  if (off || !a.wgt || a.nphase != 1 || maxM > 64 || a.C > kRowsLinC || a.N < 256) return false;
  const ConvPhase& p = a.ph[0];
  if (p.Ty != 1 || p.Tx != 1 || a.S != 1 || a.os != 1 || a.reflect || a.shuf || a.x_up || a.ksplit > 1) return false;
  if (a.IH != 1 || a.IW != 1 || a.OHf != 1 || a.OWf != 1 || p.OHp != 1 || p.OWp != 1 || p.offy || p.offx || p.ophy || p.ophx) return false;
  if (a.epi_act || a.y2_leak || a.pool_y || a.y_bf || a.epi_bf) return false;
  return true;
}
static const char* launch_rows_linear(const ConvArgs& a, hipStream_t s) {
  const ConvPhase& p = a.ph[0];
  const float* w = a.wgt + (long long)(p.kh0 * a.KW + p.kw0) * a.w_ts;
  hipLaunchKernelGGL(rows_linear_kernel, dim3((a.N + 63) / 64, (a.NB + 31) / 32), dim3(256), 0, s, a.x, a.x_ps, w, a.ldw, a.bias, a.y,
                     a.y_ps, a.NB, a.C, a.N, a.accumulate);
  return "rows_linear<32,64>";
}

// The kernel family launch_conv_gemm runs for these arguments (a: after conv_fix_ksplit).  ONE function for the launcher and for
// the predicates that must agree with it (conv_up_kernel below; ADVICE r5: the predicate used to repeat the dispatch's tests by hand).
enum ConvFamily { CONV_PK, CONV_X3Q, CONV_SCATTER, CONV_SMALLC, CONV_NARROW, CONV_BF16P, CONV_X3, CONV_BF16, CONV_ROWS_LINEAR, CONV_GENERIC };
static ConvFamily conv_family(const ConvArgs& a, int maxM) {
  if (conv_pk_eligible(a)) return CONV_PK;
#ifdef EG_EXPERIMENTS
  if (a.x3q && a.N > 16 && conv_x3q_eligible(a)) return CONV_X3Q;
#endif
  if (conv_scatter_eligible(a)) return CONV_SCATTER;
  if (conv_smallc_eligible(a)) return CONV_SMALLC;
  if (a.N <= 16) return CONV_NARROW;
  if (conv_bf16p_eligible(a)) return CONV_BF16P;   // bf16 mode, the operand's bf16 plane is there: staged by LDS-DMA (k_conv_bf16p.hip)
  if (conv_x3_eligible(a)) return CONV_X3;
  if (conv_bf16_eligible(a)) return CONV_BF16;
  if (rows_linear_eligible(a, maxM)) return CONV_ROWS_LINEAR;
  return CONV_GENERIC;
}
// ConvArgs::x_up is honoured by the plane-fed kernel and the halo kernel (their stride-1 forms): would launch_conv_gemm run one
// of them for these arguments?
bool conv_up_kernel(const ConvArgs& a_in) {
  ConvArgs a = a_in;
  conv_fix_ksplit(a);
  if (a.S != 1 || a.nphase != 1 || (a.IH & 1) || (a.IW & 1)) return false;
  const int maxM = conv_max_m(a);
  switch (conv_family(a, maxM)) {
    case CONV_BF16P: return true;
    case CONV_X3: return conv_x3_takes_halo(a, a.x3_tile ? a.x3_tile == 1 : conv_big_tile(a, maxM));
    default: return false;
  }
}

double launch_conv_gemm(ConvArgs& a_in, hipStream_t s, const char** kernel_name) {
  ConvArgs a = a_in;
  int caps = 0;   // what the kernel that runs honours of the optional epilogue work (ConvArgs::epi_act, ::y2): set where it is chosen
  int maxM = 0;
  double flops = 0;
  for (int p = 0; p < a.nphase; ++p) {
    const int M = a.NB * a.ph[p].OHp * a.ph[p].OWp;
    if (M > maxM) maxM = M;
    flops += 2.0 * M * a.N * (double)(a.ph[p].Ty * a.ph[p].Tx) * a.C;
  }
  EG_CHECK(!a.x_up || conv_up_kernel(a), "conv launch: x_up with a kernel that does not read the half-resolution operand (N=%d C=%d)", a.N, a.C);
  const bool avec = (a.C % 16 == 0) && (a.x_ps % 4 == 0) && (((uintptr_t)a.x & 15) == 0);
  const bool bvec = (a.N % 4 == 0) && (a.ldw % 4 == 0) && (a.w_ts % 4 == 0) && (((uintptr_t)a.wgt & 15) == 0);
  conv_fix_ksplit(a);
  const char* nm;
  EG_CHECK(a.wgt || (a.N > 16 && (conv_bf16p_eligible(a) || conv_x3_eligible(a) || conv_bf16_eligible(a))),
           "conv launch without an fp32 filter that no limb / bf16 kernel accepts (N=%d C=%d)", a.N, a.C);
  // (ops.cpp: conv2d asked for the output's bf16 plane, which only conv_pk writes)
  EG_CHECK(!a.y_bf || conv_pk_eligible(a), "output plane requested for a launch the packed-K kernel does not take (N=%d C=%d)", a.N, a.C);
  switch (conv_family(a, maxM)) {
    case CONV_PK:
      a.y2_leak = nullptr;
      nm = launch_conv_pk(a, s);
      // (ConvArgs::y2 is NOT taken here although the float4 epilogue could store it: this kernel is bound by its 32-byte
      //  partial-line output writes, and a second output stream of the same shape cost 97 us on the classifier's 8 -> 128
      //  layer at 64 x 64 where the separate PReLU pass costs 49 -- profiles/r05b_*)
      break;
#ifdef EG_EXPERIMENTS
    case CONV_X3Q:
      if (a.epi_act == ACT_PRELU_X) a.epi_act = 0;   // (that kernel's epilogue has the ReLU / LeakyReLU forms only)
      nm = launch_conv_x3q(a, s);
      if (a.ksplit <= 1 && a.epi_act) caps |= CONV_CAP_EPI_ACT;
      break;
#endif
    case CONV_SCATTER:
      if (a.act == ACT_TANH && !(a.shuf && !a.accumulate)) a.act = 0;
      nm = launch_conv_scatter(a, s);
      if (a.act == ACT_TANH) caps |= CONV_CAP_TANH;
      break;
    case CONV_SMALLC:
      nm = launch_conv_smallc(a, s);
      break;
    case CONV_NARROW:
      if ((long long)((maxM + 255) / 256) * a.nphase >= 256)
        nm = conv_dispatch<256, 16, 64, 16>(a, maxM, avec, bvec, s, "conv_gemm<256,16,vec>", "conv_gemm<256,16,avec>",
                                            "conv_gemm<256,16,flat>");
      else
        nm = conv_dispatch<64, 16, 16, 16>(a, maxM, avec, bvec, s, "conv_gemm<64,16,vec>", "conv_gemm<64,16,avec>",
                                           "conv_gemm<64,16,flat>");
      break;
    case CONV_BF16P:
      nm = launch_conv_bf16p(a, s);
      if (a.ksplit <= 1) caps |= CONV_CAP_EPI_ACT | CONV_CAP_Y2;
      if (conv_bf16p_pool_ok(a)) caps |= CONV_CAP_POOL;
      break;
    case CONV_X3:
      nm = launch_conv_gemm_x3(a, maxM, a.x3_tile ? a.x3_tile == 1 : conv_big_tile(a, maxM), s, &caps);
      break;
    case CONV_BF16:
      nm = launch_conv_gemm_bf16(a, maxM, conv_big_tile(a, maxM), s);
      break;
    case CONV_ROWS_LINEAR:
      nm = launch_rows_linear(a, s);
      break;
    default:
      if (conv_big_tile(a, maxM))   // (tile by CU load balance, above)
        nm = conv_dispatch<128, 128, 64, 64>(a, maxM, avec, bvec, s,
                                             a.ksplit > 1 ? "conv_gemm<128,128,vec,splitk>" : "conv_gemm<128,128,vec>",
                                             "conv_gemm<128,128,avec>", "conv_gemm<128,128,flat>");
      else
        nm = conv_dispatch<64, 64, 32, 32>(a, maxM, avec, bvec, s,
                                           a.ksplit > 1 ? "conv_gemm<64,64,vec,splitk>" : "conv_gemm<64,64,vec>",
                                           "conv_gemm<64,64,avec>", "conv_gemm<64,64,flat>");
      break;
  }
  if (a.ksplit > 1 && !a.keep_slabs) launch_conv_split_reduce(a.ws, a.y, a.bias, a.y_total, a.y_ps, a.N, a.ksplit, a.accumulate, s);
  static const bool trace = dbg_env("EG_TRACE_GEMM") != nullptr;   // shape log for offline tile/split tuning
  if (trace)
    fprintf(stderr, "EG_GEMM conv maxM=%d N=%d C=%d nphase=%d taps0=%d ksplit=%d keep=%d acc=%d kernel=%s gflop=%.3f\n", maxM, a.N, a.C,
            a.nphase, a.ph[0].Ty * a.ph[0].Tx, a.ksplit, a.keep_slabs, a.accumulate, nm, flops * 1e-9);
  if (kernel_name) *kernel_name = nm;
  a_in.caps = caps;
  a_in.ksplit = a.ksplit;   // (the split that ran: the caller's slab bookkeeping must not assume the one it asked for)
  return flops;
}

// =============================================================================================
// wgrad_kernel
// =============================================================================================
template <int BM, int BN, int WTM, int WTN, int BK, bool VEC>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs a, float* __restrict__ out, int dense_out) {
  constexpr int LDA = BM + ((BM % 32 == 0) ? 16 : 32);
  constexpr int LDB = BN + ((BN % 32 == 0) ? 16 : 32);
  constexpr int WGN = BN / WTN;
  constexpr int MI = WTM / 16, NJ = WTN / 16;
  static_assert(VEC || BK == 16, "scalar path uses K step 16");
  __shared__ __attribute__((aligned(16))) float At[BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[BK * LDB];
  __shared__ long long poff[2][BK];   // gathered-pixel offset (floats) of each K row, -1 if outside

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int C = a.C, N = a.N, IH = a.IH, IW = a.IW;
  const int ntn = (N + BN - 1) / BN;
  const int c0 = (blockIdx.x / ntn) * BM, n0 = (blockIdx.x % ntn) * BN;
  const int tap = blockIdx.y;
  const int kh = tap / a.KW, kw = tap - kh * a.KW;
  const int M = a.NB * a.OH * a.OW;
  int chunk = (M + a.nsplit - 1) / a.nsplit;
  chunk = (chunk + BK - 1) / BK * BK;
  const int p_lo = blockIdx.z * chunk;
  const int p_hi = min(M, p_lo + chunk);

  constexpr int A_V = (BK * BM / 4 + 255) / 256, B_V = (BK * BN / 4 + 255) / 256;
  constexpr int A_S = BM / 16, B_S = BN / 16;
  float4 ra4[VEC ? A_V : 1], rb4[VEC ? B_V : 1];
  float ras[VEC ? 1 : A_S], rbs[VEC ? 1 : B_S];

  // one thread per K row decodes the pixel once for the whole block
  auto decode = [&](int p, int buf) {
    if (tid < BK) {
      const int m = p + tid;
      long long off = -1;
      if (m < p_hi) {
        const int ox = m % a.OW;
        const int t2 = m / a.OW;
        const int oy = t2 % a.OH;
        const int nb = t2 / a.OH;
        int iy = oy * a.S + kh - a.pad, ix = ox * a.S + kw - a.pad;
        if (a.reflect) {
          iy = reflect_idx(iy, IH);
          ix = reflect_idx(ix, IW);
        }
        if (iy >= 0 && iy < IH && ix >= 0 && ix < IW) off = (long long)((nb * IH + iy) * IW + ix) * a.x_ps;
      }
      poff[buf][tid] = off;
    }
  };

  auto load_tile = [&](int p, int buf) {
    if constexpr (VEC) {
#pragma unroll
      for (int i = 0; i < A_V; ++i) {
        const int idx = tid + 256 * i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < BK * BM / 4) {
          const int k = idx / (BM / 4), c = c0 + (idx % (BM / 4)) * 4;
          const long long off = poff[buf][k];
          if (off >= 0 && c < C) v = *reinterpret_cast<const float4*>(a.x + off + c);
        }
        ra4[i] = v;
      }
#pragma unroll
      for (int i = 0; i < B_V; ++i) {
        const int idx = tid + 256 * i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < BK * BN / 4) {
          const int k = idx / (BN / 4), n = n0 + (idx % (BN / 4)) * 4;
          const int m = p + k;
          if (m < p_hi && n < N) v = *reinterpret_cast<const float4*>(a.dy + (size_t)m * a.dy_ps + n);
        }
        rb4[i] = v;
      }
    } else {
      const int k = tid >> 4;
      const int m = p + k;
      const long long off = poff[buf][k];
#pragma unroll
      for (int j = 0; j < A_S; ++j) {
        const int c = c0 + (tid & 15) + 16 * j;
        ras[j] = (off >= 0 && c < C) ? a.x[off + c] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < B_S; ++j) {
        const int n = n0 + (tid & 15) + 16 * j;
        rbs[j] = (m < p_hi && n < N) ? a.dy[(size_t)m * a.dy_ps + n] : 0.f;
      }
    }
  };
  auto store_tile = [&]() {
    if constexpr (VEC) {
#pragma unroll
      for (int i = 0; i < A_V; ++i) {
        const int idx = tid + 256 * i;
        if (idx < BK * BM / 4) *reinterpret_cast<float4*>(&At[(idx / (BM / 4)) * LDA + (idx % (BM / 4)) * 4]) = ra4[i];
      }
#pragma unroll
      for (int i = 0; i < B_V; ++i) {
        const int idx = tid + 256 * i;
        if (idx < BK * BN / 4) *reinterpret_cast<float4*>(&Bs[(idx / (BN / 4)) * LDB + (idx % (BN / 4)) * 4]) = rb4[i];
      }
    } else {
#pragma unroll
      for (int j = 0; j < A_S; ++j) At[(tid >> 4) * LDA + (tid & 15) + 16 * j] = ras[j];
#pragma unroll
      for (int j = 0; j < B_S; ++j) Bs[(tid >> 4) * LDB + (tid & 15) + 16 * j] = rbs[j];
    }
  };

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int l15 = lane & 15, lq = lane >> 4;
  if (p_lo < p_hi) {
    decode(p_lo, 0);
    __syncthreads();
    load_tile(p_lo, 0);
    int it = 0;
    for (int p = p_lo; p < p_hi; p += BK, ++it) {
      store_tile();
      const bool more = p + BK < p_hi;
      if (more) decode(p + BK, (it + 1) & 1);
      __syncthreads();
      if (more) load_tile(p + BK, (it + 1) & 1);
#pragma unroll
      for (int kk = 0; kk < BK / 4; ++kk) {
        float af[MI], bf[NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = At[(kk * 4 + lq) * LDA + wm * WTM + i * 16 + l15];
#pragma unroll
        for (int j = 0; j < NJ; ++j) bf[j] = Bs[(kk * 4 + lq) * LDB + wn * WTN + j * 16 + l15];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j], af[i], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
    }
  }

  const int T = a.KH * a.KW;
  // operands swapped (see conv_gemm_kernel): lane l owns filter row c = ..+(lane&15) and the 4 consecutive output
  // channels (lane>>4)*4 + reg
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int c = c0 + wm * WTM + i * 16 + l15;
    if (c >= C) continue;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = n0 + wn * WTN + j * 16 + lq * 4;
      float* q = dense_out ? out + (((size_t)blockIdx.z * T + tap) * C + c) * N + n  // partial slab [split][T][C][N]
                           : out + (size_t)tap * a.w_ts + (size_t)c * a.ldw + n;
      if constexpr (VEC) {
        if (n < N) {
          float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
          if (!dense_out && a.accumulate) {
            const float4 o = *reinterpret_cast<const float4*>(q);
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
          }
          *reinterpret_cast<float4*>(q) = v;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < N) q[r] = (!dense_out && a.accumulate) ? q[r] + acc[i][j][r] : acc[i][j][r];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// wgrad for small gathered-channel counts (C <= 16: the 3-channel image layers, the 8-channel first
// MRU unit, the generator's 3-channel output): the GEMM M dimension is the FLATTENED (tap, c) index,
// so dY is streamed once instead of once per tap.
// ---------------------------------------------------------------------------------------------
template <int BK>
__global__ __launch_bounds__(256) void wgrad_flatm_kernel(const WgradArgs a, float* __restrict__ out, int dense_out) {
  constexpr int BM = 64, BN = 64, LD = 64 + 16;
  constexpr int AR = BK / 4, BR = BK / 16;   // A rows / B row groups per thread
  __shared__ __attribute__((aligned(16))) float At[BK * LD];
  __shared__ __attribute__((aligned(16))) float Bs[BK * LD];
  __shared__ int pinfo[2][BK][3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int C = a.C, N = a.N, IH = a.IH, IW = a.IW;
  const int TC = a.KH * a.KW * C;
  const int ntn = (N + BN - 1) / BN;
  const int m0 = (blockIdx.x / ntn) * BM, n0 = (blockIdx.x % ntn) * BN;
  const int M = a.NB * a.OH * a.OW;
  int chunk = (M + a.nsplit - 1) / a.nsplit;
  chunk = (chunk + BK - 1) / BK * BK;
  const int p_lo = blockIdx.y * chunk;
  const int p_hi = min(M, p_lo + chunk);
  // this thread's flattened (tap, c) column of A
  const int mcol = m0 + (tid & 63);
  const bool mv = mcol < TC;
  const int tap = mv ? mcol / C : 0;
  const int cch = mcol - tap * C;
  const int kh = tap / a.KW, kw = tap - kh * a.KW;
  const bool nvec = (N % 4 == 0) && (a.dy_ps % 4 == 0) && (((uintptr_t)a.dy & 15) == 0);

  auto decode = [&](int p, int buf) {
    if (tid < BK) {
      const int m = p + tid;
      int b = -1, iy0 = 0, ix0 = 0;
      if (m < p_hi) {
        const int ox = m % a.OW;
        const int t2 = m / a.OW;
        const int oy = t2 % a.OH;
        b = (t2 / a.OH) * IH * IW;
        iy0 = oy * a.S - a.pad;
        ix0 = ox * a.S - a.pad;
      }
      pinfo[buf][tid][0] = b;
      pinfo[buf][tid][1] = iy0;
      pinfo[buf][tid][2] = ix0;
    }
  };
  float ra[AR];
  float4 rb[BR];
  auto load_tile = [&](int p, int buf) {
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const int k = (tid >> 6) + 4 * i;
      const int b = pinfo[buf][k][0];
      int iy = pinfo[buf][k][1] + kh, ix = pinfo[buf][k][2] + kw;
      if (a.reflect) {
        iy = reflect_idx(iy, IH);
        ix = reflect_idx(ix, IW);
      }
      float v = 0.f;
      if (mv && b >= 0 && iy >= 0 && iy < IH && ix >= 0 && ix < IW) v = a.x[(size_t)(b + iy * IW + ix) * a.x_ps + cch];
      ra[i] = v;
    }
#pragma unroll
    for (int j = 0; j < BR; ++j) {
      const int k = (tid >> 4) + 16 * j, n = n0 + (tid & 15) * 4, m = p + k;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < p_hi) {
        const float* row = a.dy + (size_t)m * a.dy_ps;
        if (nvec && n + 3 < N) {
          v = *reinterpret_cast<const float4*>(row + n);
        } else {
          if (n < N) v.x = row[n];
          if (n + 1 < N) v.y = row[n + 1];
          if (n + 2 < N) v.z = row[n + 2];
          if (n + 3 < N) v.w = row[n + 3];
        }
      }
      rb[j] = v;
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < AR; ++i) At[((tid >> 6) + 4 * i) * LD + (tid & 63)] = ra[i];
#pragma unroll
    for (int j = 0; j < BR; ++j) *reinterpret_cast<float4*>(&Bs[((tid >> 4) + 16 * j) * LD + (tid & 15) * 4]) = rb[j];
  };

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int l15 = lane & 15, lq = lane >> 4;
  if (p_lo < p_hi) {
    decode(p_lo, 0);
    __syncthreads();
    load_tile(p_lo, 0);
    int it = 0;
    for (int p = p_lo; p < p_hi; p += BK, ++it) {
      store_tile();
      const bool more = p + BK < p_hi;
      if (more) decode(p + BK, (it + 1) & 1);
      __syncthreads();
      if (more) load_tile(p + BK, (it + 1) & 1);
#pragma unroll
      for (int kk = 0; kk < BK / 4; ++kk) {
        float af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = At[(kk * 4 + lq) * LD + wm * 32 + i * 16 + l15];
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[j] = Bs[(kk * 4 + lq) * LD + wn * 32 + j * 16 + l15];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wm * 32 + i * 16 + lq * 4 + r;
      if (m >= TC) continue;
      const int t = m / C, c = m - t * C;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 32 + j * 16 + l15;
        if (n >= N) continue;
        if (dense_out) {
          out[((size_t)blockIdx.y * TC + m) * N + n] = acc[i][j][r];
        } else {
          float* q = out + (size_t)t * a.w_ts + (size_t)c * a.ldw + n;
          *q = a.accumulate ? *q + acc[i][j][r] : acc[i][j][r];
        }
      }
    }
}

// second stage of the split reduction: 16 elements x 16 split-lanes per block, fixed summation order
// (lane l sums splits l, l+16, ...; the 16 lane totals are added in lane order) -> deterministic.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                           int T, int C, int N, long long w_ts, int ldw, int nsplit,
                                                           int accumulate) {
  __shared__ float red[16][17];
  const size_t total = (size_t)T * C * N;
  const int e = threadIdx.x & 15, l = threadIdx.x >> 4;
  const size_t i = (size_t)blockIdx.x * 16 + e;
  float s = 0.f;
  if (i < total)
    for (int z = l; z < nsplit; z += 16) s += ws[(size_t)z * total + i];
  red[l][e] = s;
  __syncthreads();
  if (l == 0 && i < total) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][e];
    const int n = (int)(i % N);
    const size_t t2 = i / N;
    const int c = (int)(t2 % C);
    const int tp = (int)(t2 / C);
    float* p = dw + (size_t)tp * w_ts + (size_t)c * ldw + n;
    *p = accumulate ? (*p + t) : t;
  }
}

// float4 form (N % 4 == 0): 64 element quads x 4 split lanes per block; lane l sums splits l, l+4, ... with 4
// independent loads in flight, the 4 lane totals are added in lane order -> deterministic, 16-byte coalesced.
__device__ __forceinline__ void wgrad_reduce_v_body(float4 (&red)[4][64], unsigned block, const float* __restrict__ ws,
                                                    float* __restrict__ dw, int T, int C, int N, long long w_ts, int ldw,
                                                    int nsplit, int accumulate) {
  const size_t total4 = (size_t)T * C * N / 4;
  const int e = threadIdx.x & 63, l = threadIdx.x >> 6;
  const size_t i4 = (size_t)block * 64 + e;
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
  if (i4 < total4) {
    const float4* p = reinterpret_cast<const float4*>(ws) + i4;
    int z = l;
    for (; z + 12 < nsplit; z += 16) {
      const float4 a0 = p[(size_t)z * total4], a1 = p[(size_t)(z + 4) * total4], a2 = p[(size_t)(z + 8) * total4],
                   a3 = p[(size_t)(z + 12) * total4];
      s0.x += a0.x; s0.y += a0.y; s0.z += a0.z; s0.w += a0.w;
      s1.x += a1.x; s1.y += a1.y; s1.z += a1.z; s1.w += a1.w;
      s2.x += a2.x; s2.y += a2.y; s2.z += a2.z; s2.w += a2.w;
      s3.x += a3.x; s3.y += a3.y; s3.z += a3.z; s3.w += a3.w;
    }
    for (; z < nsplit; z += 4) {
      const float4 a0 = p[(size_t)z * total4];
      s0.x += a0.x; s0.y += a0.y; s0.z += a0.z; s0.w += a0.w;
    }
  }
  red[l][e] = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z),
                          (s0.w + s1.w) + (s2.w + s3.w));
  __syncthreads();
  if (l == 0 && i4 < total4) {
    const float4 r0 = red[0][e], r1 = red[1][e], r2 = red[2][e], r3 = red[3][e];
    float4 t = make_float4((r0.x + r1.x) + (r2.x + r3.x), (r0.y + r1.y) + (r2.y + r3.y), (r0.z + r1.z) + (r2.z + r3.z),
                           (r0.w + r1.w) + (r2.w + r3.w));
    const size_t i = i4 * 4;
    const int n = (int)(i % N);
    const size_t t2 = i / N;
    const int c = (int)(t2 % C);
    const int tp = (int)(t2 / C);
    float4* q = reinterpret_cast<float4*>(dw + (size_t)tp * w_ts + (size_t)c * ldw + n);
    if (accumulate) {
      const float4 o = *q;
      t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
    }
    *q = t;
  }
}

__global__ __launch_bounds__(256) void wgrad_reduce_v_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                             int T, int C, int N, long long w_ts, int ldw, int nsplit,
                                                             int accumulate) {
  __shared__ float4 red[4][64];
  wgrad_reduce_v_body(red, blockIdx.x, ws, dw, T, C, N, w_ts, ldw, nsplit, accumulate);
}

constexpr int kWideSplits = 64;
// One launch for a batch of queued second-stage reductions (kernels.h: RedBatch).  A block finds its descriptor from the
// block-offset table and runs that reduction's ordinary block body.
__global__ __launch_bounds__(256) void reduce_batch_kernel(const RedBatch b) {
  __shared__ float4 red[4][64];
  int i = 0;
  while (i + 1 < b.count && blockIdx.x >= b.start[i + 1]) ++i;   // uniform
  const RedDesc& d = b.d[i];
  const unsigned block = blockIdx.x - b.start[i];
  if (d.kind == 0 && d.n >= kWideSplits) {
    // many slabs, few elements (the 3- / 8-channel layers: up to 757 slabs of a 48 x 64 filter): 16 split lanes x 16
    // element quads per block, so a lane's chain of dependent loads is nsplit / 16 long instead of nsplit / 4
    const int e = threadIdx.x & 15, l = threadIdx.x >> 4;
    const size_t total4 = (size_t)d.T * d.C * d.N / 4;
    const size_t i4 = (size_t)block * 16 + e;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    if (i4 < total4) {
      const float4* p = reinterpret_cast<const float4*>(d.src) + i4;
      int z = l;
      for (; z + 16 < d.n; z += 32) {
        const float4 a0 = p[(size_t)z * total4], a1 = p[(size_t)(z + 16) * total4];
        s0.x += a0.x; s0.y += a0.y; s0.z += a0.z; s0.w += a0.w;
        s1.x += a1.x; s1.y += a1.y; s1.z += a1.z; s1.w += a1.w;
      }
      if (z < d.n) {
        const float4 a0 = p[(size_t)z * total4];
        s0.x += a0.x; s0.y += a0.y; s0.z += a0.z; s0.w += a0.w;
      }
    }
    float4* r16 = &red[0][0];   // [16 lanes][16 quads]
    r16[l * 16 + e] = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
    __syncthreads();
    if (l == 0 && i4 < total4) {
      float4 t = r16[e];
#pragma unroll
      for (int q = 1; q < 16; ++q) {
        const float4 v = r16[q * 16 + e];
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
      }
      const size_t i = i4 * 4;
      const int n = (int)(i % d.N);
      const size_t t2 = i / d.N;
      const int c = (int)(t2 % d.C);
      const int tp = (int)(t2 / d.C);
      float4* q = reinterpret_cast<float4*>(d.dst + (size_t)tp * d.w_ts + (size_t)c * d.ldw + n);
      if (d.accumulate) {
        const float4 o = *q;
        t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
      }
      *q = t;
    }
  } else if (d.kind == 0) {
#ifdef EG_REDUCE_LANES
    wgrad_reduce_v_body(red, block, d.src, d.dst, d.T, d.C, d.N, d.w_ts, d.ldw, d.n, d.accumulate);
#else
    // one element quad per thread, all slabs: 8 independent loads in flight, the partial sums combined in a fixed tree --
    // no LDS stage, 4 KB contiguous per block and slab
    const size_t total4 = (size_t)d.T * d.C * d.N / 4;
    const size_t i4 = (size_t)block * 256 + threadIdx.x;
    if (i4 < total4) {
      const float4* p = reinterpret_cast<const float4*>(d.src) + i4;
      float4 s[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      int z = 0;
      for (; z + 8 <= d.n; z += 8) {
        float4 a[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = p[(size_t)(z + k) * total4];
#pragma unroll
        for (int k = 0; k < 8; ++k) { s[k].x += a[k].x; s[k].y += a[k].y; s[k].z += a[k].z; s[k].w += a[k].w; }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (z + k < d.n) {
          const float4 a = p[(size_t)(z + k) * total4];
          s[k].x += a.x; s[k].y += a.y; s[k].z += a.z; s[k].w += a.w;
        }
      float4 t = make_float4(((s[0].x + s[1].x) + (s[2].x + s[3].x)) + ((s[4].x + s[5].x) + (s[6].x + s[7].x)),
                             ((s[0].y + s[1].y) + (s[2].y + s[3].y)) + ((s[4].y + s[5].y) + (s[6].y + s[7].y)),
                             ((s[0].z + s[1].z) + (s[2].z + s[3].z)) + ((s[4].z + s[5].z) + (s[6].z + s[7].z)),
                             ((s[0].w + s[1].w) + (s[2].w + s[3].w)) + ((s[4].w + s[5].w) + (s[6].w + s[7].w)));
      const size_t i = i4 * 4;
      const int n = (int)(i % d.N);
      const size_t t2 = i / d.N;
      const int c = (int)(t2 % d.C);
      const int tp = (int)(t2 / d.C);
      float4* q = reinterpret_cast<float4*>(d.dst + (size_t)tp * d.w_ts + (size_t)c * d.ldw + n);
      if (d.accumulate) {
        const float4 o = *q;
        t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
      }
      *q = t;
    }
#endif
  } else {   // col_sum_stage2 (k_elem.hip): 64 columns per block, 4 row lanes
    float* r = reinterpret_cast<float*>(&red[0][0]);
    const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
    const int c = (int)block * 64 + cx;
    float s = 0.f;
    if (c < d.C)
      for (int q = py; q < d.n; q += 4) s += d.src[(size_t)q * d.C + c];
    r[py * 64 + cx] = s;
    __syncthreads();
    if (py == 0 && c < d.C) {
      const float t = (r[cx] + r[64 + cx]) + (r[128 + cx] + r[192 + cx]);
      d.dst[c] = d.accumulate ? d.dst[c] + t : t;
    }
  }
}
unsigned red_desc_blocks(const RedDesc& d) {
  if (d.kind == 0 && d.n >= kWideSplits) return (unsigned)(((size_t)d.T * d.C * d.N / 4 + 15) / 16);
#ifdef EG_REDUCE_LANES
  if (d.kind == 0) return (unsigned)(((size_t)d.T * d.C * d.N / 4 + 63) / 64);
#else
  if (d.kind == 0) return (unsigned)(((size_t)d.T * d.C * d.N / 4 + 255) / 256);
#endif
  return (unsigned)((d.C + 63) / 64);
}
void launch_reduce_batch(const RedBatch& b, hipStream_t s) {
  if (b.count == 0) return;
  hipLaunchKernelGGL(reduce_batch_kernel, dim3(b.start[b.count]), dim3(256), 0, s, b);
}

static bool wgrad_big(const WgradArgs& a) { return a.C >= 128 && a.N >= 128; }
static bool wgrad_flatm(const WgradArgs& a) { return a.C <= 16; }

// Split count over the pixel dimension.  The kernels are MFMA-bound, so a CU retires blocks at a fixed rate however
// many are resident; what costs time is the TAIL: with tiles*s = 1026 blocks two CUs get a 5th block and the launch
// takes 5 block-times instead of 4.008 (measured: 88 vs 106 TFLOP/s on the 3x3 128->128 layer).  Pick the smallest
// s near the occupancy target whose block count divides evenly over the 256 CUs; fewer splits also mean fewer
// partial slabs for the second-stage reduction.
static int pick_balanced_splits(long long tiles, long long target_blocks, long long smax) {
  if (smax < 1) smax = 1;
  long long s0 = (target_blocks + tiles - 1) / tiles;
  if (s0 > smax) s0 = smax;
  if (s0 <= 1) return 1;
  auto eff = [&](long long s) {
    const long long blocks = tiles * s;
    return (double)blocks / (double)(((blocks + 255) / 256) * 256);
  };
  const long long lo = std::max(1LL, (s0 * 5) / 8), hi = std::min(smax, s0 + s0 / 8);
  double best = 0.0;
  for (long long s = lo; s <= hi; ++s) best = std::max(best, eff(s));
  for (long long s = lo; s <= hi; ++s)
    if (eff(s) >= best - 0.015) return (int)s;
  return (int)s0;
}

static bool wgrad_x3(const WgradArgs& a) {   // (limbs == 1: the same kernel with one plane = bf16-rounded operands)
  static const bool bf16_pp = !(dbg_env("EG_BF16_PP") && atoi(dbg_env("EG_BF16_PP")) == 0);
  // C == 64: two taps per 128-row tile (k_gemm_x3.hip), worth it when there are at least two taps
  const bool c_ok = a.C >= 128 || (a.C == 64 && a.KH * a.KW >= 2);
  return (a.limbs == 3 || (a.limbs == 1 && bf16_pp)) && c_ok && a.N >= 128 && a.C % 4 == 0 && a.N % 4 == 0 && a.x_ps % 4 == 0 && a.dy_ps % 4 == 0 &&
         a.x_bytes != 0 && a.dy_bytes != 0 && a.x_bytes <= (1u << 30) && a.dy_bytes <= (1u << 30) &&
         (((uintptr_t)a.x | (uintptr_t)a.dy) & 15) == 0 && a.ldw % 4 == 0 && a.w_ts % 4 == 0;
}

int wgrad_pick_splits(const WgradArgs& a) {
  const long long M = (long long)a.NB * a.OH * a.OW;
  if (const int pu = wgrad_bf16p_units(a)) {   // plane-fed kernel: 256-thread blocks, up to three per CU; a slab is >= 8 K steps of 32 pixels.  Target 384 blocks: with 768
                                               // (three full waves) the slabs cost more in the second-stage reduction than the occupancy gained --
                                               // in-box 2,870 / 2,885 / 2,944 / 2,943 images/s at 768 / 512 / 384 / 256 (profiles/r04u_ab_splits.txt)
    const long long tiles = (long long)pu * ((a.N + 127) / 128);
    static const int target = dbg_env("EG_WGRAD_BF16P_BLOCKS") ? atoi(dbg_env("EG_WGRAD_BF16P_BLOCKS")) : 384;
    return pick_balanced_splits(tiles, target, std::min(256LL, (M + 255) / 256));
  }
  if (wgrad_x3(a)) {   // ping-pong blocks: two (tap, 128-channel) units x 128 output channels, one block per CU
    const long long units = wgrad_x3_rowhalo_units(a) ? wgrad_x3_rowhalo_units(a)     // (filter row, column phase, 64-channel block)
                            : a.C == 64 ? (a.KH * a.KW + 1) / 2 : (long long)a.KH * a.KW * ((a.C + 127) / 128);
    const long long tiles = ((units + 1) / 2) * ((a.N + 127) / 128);
    // one block per CU (512 threads, 100 KB LDS): 256 blocks = one full wave of blocks.  Measured against 512 (two
    // waves): filter-gradient kernels 7.05 -> 6.70 ms per iteration and half the partial slabs to write and reduce
    static const int x3_blocks = dbg_env("EG_WGRAD_X3_BLOCKS") ? atoi(dbg_env("EG_WGRAD_X3_BLOCKS")) : 256;   // (tools/tune_x3.py sweeps it)
    return pick_balanced_splits(tiles, x3_blocks, std::min(256LL, (M + 255) / 256));
  }
  if (wgrad_pk_eligible(a)) return wgrad_pk_splits(a);
  if (wgrad_flatm(a)) {
    const long long tiles = (long long)((a.KH * a.KW * a.C + 63) / 64) * ((a.N + 63) / 64);
    // latency-bound (one gather round trip per 32-pixel step, prefetch distance 1): eight resident blocks per CU hide it
    static const int fl_chunk = dbg_env("EG_FLATM_CHUNK") ? atoi(dbg_env("EG_FLATM_CHUNK")) : 128;
    static const int fl_blocks = dbg_env("EG_FLATM_BLOCKS") ? atoi(dbg_env("EG_FLATM_BLOCKS")) : 2048;
    return pick_balanced_splits(tiles, fl_blocks, std::min(2048LL, (M + fl_chunk - 1) / fl_chunk));
  }
  const int bm = wgrad_big(a) ? 128 : 64;
  const long long tiles = (long long)((a.C + bm - 1) / bm) * ((a.N + bm - 1) / bm) * a.KH * a.KW;
  return pick_balanced_splits(tiles, 1024, std::min(256LL, (M + 255) / 256));
}

size_t wgrad_workspace_floats(const WgradArgs& a) {
  if (a.nsplit == 1) return 0;
  return (size_t)a.nsplit * a.KH * a.KW * a.C * a.N;
}

template <int BM, int BN, int WTM, int WTN, bool VEC>
static void wgrad_launch(const WgradArgs& a, float* out, int dense, hipStream_t s) {
  dim3 grid(((a.C + BM - 1) / BM) * ((a.N + BN - 1) / BN), a.KH * a.KW, a.nsplit);
  if constexpr (VEC)
    hipLaunchKernelGGL((wgrad_kernel<BM, BN, WTM, WTN, 32, true>), grid, dim3(256), 0, s, a, out, dense);
  else
    hipLaunchKernelGGL((wgrad_kernel<BM, BN, WTM, WTN, 16, false>), grid, dim3(256), 0, s, a, out, dense);
}

// The kernel family launch_wgrad runs for these arguments (`out`: where the launch writes -- dw, or the slab workspace): one
// function for the launcher and for wgrad_up_kernel
enum WgradFamily { WG_BF16P, WG_X3, WG_PK, WG_BF16, WG_FLATM, WG_GENERIC };
static bool wgrad_vec(const WgradArgs& a, const float* out) {
  return (a.C % 4 == 0) && (a.N % 4 == 0) && (a.x_ps % 4 == 0) && (a.dy_ps % 4 == 0) && (((uintptr_t)a.x & 15) == 0) &&
         (((uintptr_t)a.dy & 15) == 0) && (a.ldw % 4 == 0) && (a.w_ts % 4 == 0) && (((uintptr_t)out & 15) == 0);
}
static WgradFamily wgrad_family(const WgradArgs& a, const float* out) {
  const bool out16 = (((uintptr_t)out) & 15) == 0;
  if (wgrad_bf16p_units(a) && out16) return WG_BF16P;
  if (wgrad_x3(a) && out16) return WG_X3;
  if (wgrad_pk_eligible(a) && out16) return WG_PK;
  if (a.bf16 && wgrad_vec(a, out) && !wgrad_flatm(a)) return WG_BF16;
  if (wgrad_flatm(a)) return WG_FLATM;
  return WG_GENERIC;
}
// WgradArgs::dy_up is honoured by the plane-fed kernel and the row-halo limb kernel: would launch_wgrad run one of them?
bool wgrad_up_kernel(const WgradArgs& a) {
  const float* out = a.nsplit == 1 ? a.dw : a.ws;
  if ((a.OH & 1) || (a.OW & 1)) return false;
  switch (wgrad_family(a, out)) {
    case WG_BF16P: return true;
    case WG_X3: return wgrad_x3_rowhalo_units(a) > 0;
    default: return false;
  }
}

double launch_wgrad(const WgradArgs& a, hipStream_t s, const char** kernel_name) {
  const bool direct = (a.nsplit == 1);
  float* out = direct ? a.dw : a.ws;
  const int dense = direct ? 0 : 1;
  EG_CHECK(!a.dy_up || wgrad_up_kernel(a), "wgrad launch: dy_up with a kernel that does not read the half-resolution operand (N=%d C=%d)", a.N, a.C);
  const bool vec = wgrad_vec(a, out);
  const char* nm;
  switch (wgrad_family(a, out)) {
    case WG_BF16P: nm = launch_wgrad_bf16p(a, out, dense, s); break;
    case WG_X3: nm = launch_wgrad_x3(a, out, dense, true, s); break;
    case WG_PK: nm = launch_wgrad_pk(a, out, dense, s); break;
    case WG_BF16: nm = launch_wgrad_bf16(a, out, dense, wgrad_big(a), s); break;
    case WG_FLATM: {
      dim3 grid(((a.KH * a.KW * a.C + 63) / 64) * ((a.N + 63) / 64), a.nsplit);
      hipLaunchKernelGGL(wgrad_flatm_kernel<32>, grid, dim3(256), 0, s, a, out, dense);
      nm = "wgrad_flatm<64,64>";
      break;
    }
    default:
      if (wgrad_big(a)) {
        if (vec) { wgrad_launch<128, 128, 64, 64, true>(a, out, dense, s); nm = "wgrad<128,128,vec>"; }
        else { wgrad_launch<128, 128, 64, 64, false>(a, out, dense, s); nm = "wgrad<128,128,flat>"; }
      } else {
        if (vec) { wgrad_launch<64, 64, 32, 32, true>(a, out, dense, s); nm = "wgrad<64,64,vec>"; }
        else { wgrad_launch<64, 64, 32, 32, false>(a, out, dense, s); nm = "wgrad<64,64,flat>"; }
      }
      break;
  }
  if (!direct) {
    const size_t total = (size_t)a.KH * a.KW * a.C * a.N;
    const bool rv = (a.N % 4 == 0) && (a.ldw % 4 == 0) && (a.w_ts % 4 == 0) && (((uintptr_t)a.dw & 15) == 0) &&
                    (((uintptr_t)a.ws & 15) == 0);
    if (rv && a.defer_reduce)
      const_cast<WgradArgs&>(a).reduce_deferred = 1;
    else if (rv)
      hipLaunchKernelGGL(wgrad_reduce_v_kernel, dim3((unsigned)((total / 4 + 63) / 64)), dim3(256), 0, s, a.ws, a.dw,
                         a.KH * a.KW, a.C, a.N, a.w_ts, a.ldw, a.nsplit, a.accumulate);
    else
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, s, a.ws, a.dw,
                         a.KH * a.KW, a.C, a.N, a.w_ts, a.ldw, a.nsplit, a.accumulate);
  }
  static const bool trace = dbg_env("EG_TRACE_GEMM") != nullptr;
  if (trace)
    fprintf(stderr, "EG_GEMM wgrad M=%d N=%d C=%d taps=%d nsplit=%d kernel=%s gflop=%.3f\n", a.NB * a.OH * a.OW, a.N, a.C,
            a.KH * a.KW, a.nsplit, nm, 2e-9 * a.NB * a.OH * a.OW * (double)a.N * a.C * a.KH * a.KW);
  if (kernel_name) *kernel_name = nm;
  return 2.0 * a.NB * a.OH * a.OW * (double)a.N * a.C * a.KH * a.KW;
}

// =============================================================================================
// per-tap transpose [T][R][Cc] -> [T][Cc][R], optional scale 1/(*inv_of)
// =============================================================================================
__global__ void tap_transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int Cc,
                                     const float* __restrict__ inv_of) {
  __shared__ float tile[32][33];
  const int t = blockIdx.z;
  const float sc = inv_of ? 1.0f / *inv_of : 1.0f;
  const float* src = in + (size_t)t * R * Cc;
  float* dst = out + (size_t)t * R * Cc;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + threadIdx.x;
    tile[i][threadIdx.x] = (r < R && c < Cc) ? src[(size_t)r * Cc + c] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + threadIdx.x;
    if (r < R && c < Cc) dst[(size_t)c * R + r] = tile[threadIdx.x][i] * sc;
  }
}

__global__ void merge_phase_filter_kernel(const float* __restrict__ w, float* __restrict__ wm, int k, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // over 9 * C * 12
  if (i >= 9 * C * 12) return;
  const int n = i % 12, c = (i / 12) % C, tap = i / (12 * C);
  const int dy = tap / 3 - 1, dx = tap % 3 - 1;
  const int p = n / 3, ch = n - 3 * p;
  const int kh = (p >> 1) + 1 - 2 * dy, kw = (p & 1) + 1 - 2 * dx;
  float v = 0.f;
  if (kh >= 0 && kh < k && kw >= 0 && kw < k) v = w[((size_t)(kh * k + kw) * 3 + ch) * C + c];
  wm[i] = v;
}
void launch_merge_phase_filter(const float* w, float* wm, int k, int C, hipStream_t s) {
  const int n = 9 * C * 12;
  hipLaunchKernelGGL(merge_phase_filter_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w, wm, k, C);
}

void launch_tap_transpose(const float* in, float* out, int T, int R, int Cc, const float* inv_of, hipStream_t s) {
  dim3 grid((Cc + 31) / 32, (R + 31) / 32, T);
  hipLaunchKernelGGL(tap_transpose_kernel, grid, dim3(32, 8), 0, s, in, out, R, Cc, inv_of);
}

}  // namespace eg
