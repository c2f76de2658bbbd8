// Kernels for the thin layers of the EdgeGAN stack on gfx950: contractions with <= 12 output channels (input gradients
// towards the 3-channel image / the classifier's 8-channel stem, the phase-merged 3-channel transposed convolution) or
// <= 8 input channels.  These layers are HBM-class: a few FLOPs per byte, so what matters is that every
// operand byte is read from HBM/L2 ONCE, in whole 128-byte lines, and that no MFMA work is spent on structural padding.
//
//   conv_scatter_kernel : 3x3 stride-1 convolution with N <= 12 output channels and C % 32 == 0 input channels,
//                         written as ONE dense product over the INPUT pixels followed by a col2im in LDS:
//                             P[pix][tap*N + n] = sum_c X[pix][c] * W[tap][c][n]          (M = pixels, N' = 9N, K = C)
//                             Y[oy][ox][n]      = b[n] + sum_tap P[(oy,ox) + d(tap)][tap*N + n]
//                         X is read exactly once, straight from global memory into the MFMA operand registers (whole
//                         128-byte lines, no LDS staging); the gather form re-reads it 9 times through L2 and pads N to
//                         16 per tap (N = 3: 81 % padding; here 27 -> 32 columns: 16 %).
//                         conv.py:13-36 / 278-285 input gradients, conv.py:49-53 (deconv2d, phase-merged) forward.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "kernels.h"

namespace eg {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------------
// conv_scatter_kernel<N>: N output channels, NT = ceil(9 N / 16) 16-column tiles of the product (NP = 16 NT).
//
// Block = one image x a strip of S output rows; 512 threads = 4 product waves + 4 output waves.  The strip's input rows
// [lo, hi) (the strip plus one halo row each side, clipped to the image) are processed in rounds of R rows (R W >= 64
// pixels, a multiple of 64).  A product wave owns the 16-pixel tiles wave, wave + 4, ... of the round, runs the K loop over
// all C channels for its tile with the 16x16x4 fp32 MFMA (exact fp32 products, fp32 accumulation) and writes the tile's NP
// columns to a ring of 2R + 2 product rows in LDS.  One barrier per round.  During round k the output waves sum the
// output rows that round k - 1 completed from the ring (9 taps x N per pixel, float4 reads when N % 4 == 0) and store
// them: the col2im runs beside the MFMAs on the same SIMDs instead of between them (ring rows written in round k are
// never the ones read for round k - 1: see the index argument at emit_rows).
//
// MFMA operands: lane (p = l & 15, g = l >> 4) loads channels [32 ks + 8 g, + 8) of pixel p as two float4 (a wave-level
// load pair = 16 pixels x one whole 128-byte line); MFMA j = 0..7 of the K step takes element j, i.e. contracts channels
// 32 ks + 8 g' + j over g' = 0..3.  The filter image in LDS is laid out to match: Wl[c / 8][col][c % 8], so the same lane
// reads its 8 B-operands of a column tile with two ds_read_b128.  Operands are swapped (D^T = B^T A^T): lane l ends up
// with columns 16 ct + 4 g .. + 3 of pixel p -> one ds_write_b128 per tile and column tile.
// ---------------------------------------------------------------------------------------------------------------------
//
// X3 (the f32x3 and bf16 modes; round 4): the same product on the bf16 MFMA with three-limb operands -- the lane's 8
// channels ARE the 8 K-values v_mfma_f32_16x16x32_bf16 takes from lane (p, g), so the operand registers are split in place
// (k_gemm_x3.hip: split_pair) and one K step of 32 channels is 6 limb MFMAs of 16 cycles per column tile instead of 8 fp32
// MFMAs of 32; the filter image holds the three limb planes [limb][C / 8][NP][8] bf16 (one ds_read_b128 per plane).
constexpr int kScatStripW = 62;   // output columns of a column strip (its 64 local columns = these + one halo column each side)
template <int N, bool X3>
__global__ __launch_bounds__(512) void conv_scatter_kernel(const ConvArgs a, int S, int R, int lw, int xstrips) {
  constexpr int NT = (9 * N + 15) / 16;
  constexpr int NP = NT * 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // Column strips (round 6: images wider than the product ring can hold -- 128 / 256 columns at configs[4]'s geometry): a block
  // then owns kScatStripW output columns; its LOCAL width W = 64 is those plus one halo column each side, local column lc is image
  // column gx0 + lc, columns outside the image load zeros (their products are zero), and only the inner columns are emitted.
  // xstrips == 1: the whole row, W = the image width, gx0 = 0.
  const int C = a.C, H = a.IH, WG = a.IW;
  const int W = xstrips > 1 ? 64 : WG;
  float* Wl = smem;                        // [C / 8][NP][8] fp32, or X3: [3][C / 8][NP][8] bf16
  unsigned short* Wb = reinterpret_cast<unsigned short*>(smem);
  float* Pr = smem + (X3 ? (size_t)C * NP * 3 / 2 : (size_t)C * NP);       // [(2R + 2) rows][W][NP]
  const int ring = 2 * R + 2;
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const bool producer = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) == 0;
  const int strips = (H + S - 1) / S;
  const int bxy = (int)blockIdx.x / xstrips, xs = (int)blockIdx.x - bxy * xstrips;
  const int nb = bxy / strips, ya = (bxy % strips) * S;
  const int gx0 = xstrips > 1 ? xs * kScatStripW - 1 : 0;                 // image column of local column 0
  const int ol0 = xstrips > 1 ? 1 : 0;                                    // local columns [ol0, ol1) are this block's outputs
  const int ol1 = xstrips > 1 ? min(1 + kScatStripW, WG - gx0) : WG;
  const int yb = min(H, ya + S);
  const ConvPhase ph = a.ph[0];
  const int D = a.D;                       // +1: input pixel = out + tap - 1; -1: out - tap + 1 (offy == offx == -D)

  // ---- filter -> LDS, columns (tap, n), zero padded to NP: linear (coalesced) reads, 8 per thread in flight -----------
  {
    const int CN = C * N, total = 9 * CN;
    for (int base = 0; base < total; base += 512 * 8) {
      float v[8];
      int dst[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * 512 + (int)threadIdx.x;
        v[u] = 0.f;
        dst[u] = -1;
        if (i < total) {
          const int tap = i / CN, r = i - tap * CN;
          const int c = r / N, n = r - c * N;
          const int ty = tap / 3, tx = tap - 3 * ty;
          const long long wt = (long long)((ph.kh0 + ty * a.kst) * a.KW + (ph.kw0 + tx * a.kst)) * a.w_ts;
          v[u] = a.wgt[wt + (size_t)c * a.ldw + n];
          dst[u] = ((c >> 3) * NP + tap * N + n) * 8 + (c & 7);
        }
      }
      const int plane = C * NP;             // elements of a limb plane
      auto bf = [](float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); };
      auto bff = [](unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); };
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (dst[u] >= 0) {
          if constexpr (X3) {
            const unsigned short h0 = bf(v[u]);
            const float r1 = v[u] - bff(h0);
            const unsigned short h1 = bf(r1);
            Wb[dst[u]] = h0;
            Wb[plane + dst[u]] = h1;
            Wb[2 * plane + dst[u]] = bf(r1 - bff(h1));
          } else {
            Wl[dst[u]] = v[u];
          }
        }
    }
    constexpr int PADC = NP - 9 * N;
    if constexpr (PADC > 0)
      for (int i = threadIdx.x; i < C * PADC; i += 512) {
        const int c = i / PADC, col = 9 * N + (i - c * PADC);
        const int d = ((c >> 3) * NP + col) * 8 + (c & 7);
        if constexpr (X3) {
          Wb[d] = 0;
          Wb[C * NP + d] = 0;
          Wb[2 * C * NP + d] = 0;
        } else {
          Wl[d] = 0.f;
        }
      }
  }

  const int lo = max(0, ya - 1), hi = min(H, yb + 1);
  const int rounds = (hi - lo + R - 1) / R;

  // Output rows [from, to) from the ring.  Called for round k - 1 while round k is being multiplied: it reads product
  // rows >= lo + (k-1) R - 2; round k writes rows [lo + kR, lo + (k+1) R), whose ring slots last held rows
  // [lo + (k-1) R - R - 2, lo + (k-1) R - 3]: disjoint.
  auto emit_rows = [&](int from, int to) {
    for (int oy = from; oy < to; ++oy) {
      int slot[3];
#pragma unroll
      for (int ty = 0; ty < 3; ++ty) {
        const int iy = oy + (ty - 1) * D;
        slot[ty] = (iy < 0 || iy >= H) ? -1 : iy % ring;
      }
      if constexpr (N % 4 == 0) {
        constexpr int NQ = N / 4;
        for (int t = tid + ol0 * NQ; t < ol1 * NQ; t += 256) {
          const int oxl = t / NQ, q4 = t - oxl * NQ, ox = gx0 + oxl;   // (oxl: local column of the ring, ox: image column)
          // branch-free: out-of-image taps read a clamped address and are multiplied by 0 (all 9 reads in flight at once)
          f32x4 pv[9];
          bool keep[9];
#pragma unroll
          for (int ty = 0; ty < 3; ++ty) {
            const float* prow = Pr + (size_t)(slot[ty] < 0 ? 0 : slot[ty]) * W * NP + 4 * q4;
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) {
              const int ix = oxl + (tx - 1) * D;
              const bool ok = slot[ty] >= 0 && ix >= 0 && ix < W;
              const int ixc = min(max(ix, 0), W - 1);
              pv[ty * 3 + tx] = *reinterpret_cast<const f32x4*>(prow + ixc * NP + (ty * 3 + tx) * N);
              keep[ty * 3 + tx] = ok;
            }
          }
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int q = 0; q < 9; ++q) {   // (select, not multiply: an unwritten ring row may hold NaN bit patterns)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += keep[q] ? pv[q][j] : 0.f;
          }
          if (a.shuf) {   // 12 channels -> 2x2 output pixels of 3 channels (phase-merged stride-2 layer)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int n = 4 * q4 + j, p = n / 3, ch = n - 3 * p;
              float* qy = a.y + ((size_t)(nb * a.OHf + 2 * oy + (p >> 1)) * a.OWf + 2 * ox + (p & 1)) * a.y_ps + ch;
              float r = v[j];
              if (a.bias) r += a.bias[ch];
              if (a.accumulate) r += *qy;
              if (a.act == ACT_TANH) r = tanhf(r);   // (generator.py:70-74: the generator's output activation, forward-only graphs; act_fwd_kernel's expression)
              *qy = r;
            }
          } else {
            float* qy = a.y + ((size_t)(nb * a.OHf + oy) * a.OWf + ox) * a.y_ps + 4 * q4;
            if (a.bias) {
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] += a.bias[4 * q4 + j];
            }
            if ((a.y_ps & 3) == 0 && (((uintptr_t)a.y) & 15) == 0) {
              if (a.accumulate) v += *reinterpret_cast<const f32x4*>(qy);
              *reinterpret_cast<f32x4*>(qy) = v;
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j) qy[j] = a.accumulate ? qy[j] + v[j] : v[j];
            }
          }
        }
      } else {
        for (int oxl = tid + ol0; oxl < ol1; oxl += 256) {
          const int ox = gx0 + oxl;
          float v[N];
#pragma unroll
          for (int n = 0; n < N; ++n) v[n] = 0.f;
#pragma unroll
          for (int ty = 0; ty < 3; ++ty) {
            if (slot[ty] < 0) continue;
            const float* prow = Pr + (size_t)slot[ty] * W * NP;
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) {
              const int ix = oxl + (tx - 1) * D;
              if (ix < 0 || ix >= W) continue;
#pragma unroll
              for (int n = 0; n < N; ++n) v[n] += prow[ix * NP + (ty * 3 + tx) * N + n];
            }
          }
          float* qy = a.y + ((size_t)(nb * a.OHf + oy) * a.OWf + ox) * a.y_ps;
#pragma unroll
          for (int n = 0; n < N; ++n) {
            float r = v[n];
            if (a.bias) r += a.bias[n];
            if (a.accumulate) r += qy[n];
            qy[n] = r;
          }
        }
      }
    }
  };
  // output rows complete once product rows < pr are: all of the strip after the last round, else rows < pr - 1
  auto rows_done = [&](int k) { const int pr = min(hi, lo + (k + 1) * R); return pr >= hi ? yb : max(ya, pr - 1); };

  // round barrier: LDS traffic only -- the operand prefetch (global loads several items ahead) stays in flight across it,
  // which __syncthreads() would drain (s_waitcnt vmcnt(0)) once per round
  auto lds_barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  if (!producer) {
    // ---- output waves: round k's barrier releases the rows round k completed -------------------------------------------
    __syncthreads();   // filter image (not read here, but the barrier count must match the product waves')
    int emitted = ya;
    for (int k = 0; k < rounds; ++k) {
      lds_barrier();
      const int e_now = rows_done(k);
      emit_rows(emitted, e_now);
      emitted = e_now;
    }
    return;
  }

  // ---- product waves --------------------------------------------------------------------------------------------------
  const int tiles = (R * W) >> 4;          // 16-pixel tiles per round (R W % 64 == 0)
  const int tpw = tiles >> 2;              // tiles per wave and round
  const int KS = C >> 5;                   // K steps of 32 channels
  const int p15 = lane & 15, g = lane >> 4;
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);

  // this lane's pixel of tile j in round k: byte offset of its channel group, or out of range (-> zeros)
  auto pix_off = [&](int k, int j) -> unsigned {
    const int idx = ((wave + 4 * j) << 4) + p15;
    const int row = lo + k * R + (idx >> lw), col = gx0 + (idx & (W - 1));
    if (k >= rounds || row >= hi || col < 0 || col >= WG) return 0x80000000u;
    return (unsigned)((((nb * H + row) * WG + col) * a.x_ps + 8 * g) * 4);
  };
  // flat sequence of this wave's work items (round, tile, K step); the operand loads run PD items ahead
#ifndef EG_SCATTER_PD
#define EG_SCATTER_PD 4
#endif
  constexpr int PD = EG_SCATTER_PD;   // (even: the filter fragments are double buffered by item parity)
  f32x4 ra[PD][2];
  const int items = rounds * tpw * KS;
  int ld_k = 0, ld_j = 0, ld_ks = 0;
  unsigned ld_off = pix_off(0, 0);
  auto issue = [&](f32x4 (&r)[2]) __attribute__((always_inline)) {
    r[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, ld_off, ld_ks * 128, 0));
    r[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, ld_off, ld_ks * 128 + 16, 0));
    if (++ld_ks == KS) {
      ld_ks = 0;
      if (++ld_j == tpw) {
        ld_j = 0;
        ++ld_k;
      }
      ld_off = pix_off(ld_k, ld_j);
    }
  };
#pragma unroll
  for (int i = 0; i < PD; ++i) issue(ra[i]);   // (items past the end resolve to out-of-range offsets: zeros, no faults)

  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();   // filter image complete

  const float* wl_lane = Wl + ((size_t)g * NP + p15) * 8;
  const unsigned short* wb_lane = Wb + ((size_t)g * NP + p15) * 8;
  int cur_k = 0, cur_j = 0, cur_ks = 0;
  typedef __bf16 bf16x8s __attribute__((ext_vector_type(8)));
  constexpr int NBR = X3 ? 3 : 2;   // filter registers per column tile: three limb planes / two fp32 quads
  f32x4 bb[2][NT][NBR];   // filter fragments of the current / the next item (double buffered by item parity; PD is even)
  auto load_b = [&](f32x4 (&b)[NT][NBR], int ks) __attribute__((always_inline)) {
    if constexpr (X3) {
      const unsigned short* wk = wb_lane + (size_t)ks * 4 * NP * 8;
#pragma unroll
      for (int l = 0; l < 3; ++l)
#pragma unroll
        for (int t = 0; t < NT; ++t) b[t][l] = *reinterpret_cast<const f32x4*>(wk + (size_t)l * C * NP + t * 16 * 8);
    } else {
      const float* wk = wl_lane + (size_t)ks * 4 * NP * 8;
#pragma unroll
      for (int t = 0; t < NT; ++t) b[t][0] = *reinterpret_cast<const f32x4*>(wk + t * 16 * 8);
#pragma unroll
      for (int t = 0; t < NT; ++t) b[t][1] = *reinterpret_cast<const f32x4*>(wk + t * 16 * 8 + 4);
    }
  };
  auto pk2s = [](float lo, float hi) __attribute__((always_inline)) {
    typedef float f32x2s __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2s __attribute__((ext_vector_type(2)));
    const f32x2s v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2s));
  };
  load_b(bb[0], 0);
  for (int it = 0; it < items; it += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      if (it + u < items) {
        // consume the staged operand of item it + u, then re-issue its registers for item it + u + PD
        const f32x4 a0 = ra[u][0], a1 = ra[u][1];
        issue(ra[u]);
        // this item's filter fragments were requested one item ago; request the next item's before multiplying
        const int nks = cur_ks + 1 == KS ? 0 : cur_ks + 1;
        load_b(bb[(u + 1) & 1], nks);
        if constexpr (X3) {
          // the lane's 8 channels as three limb octets: x = l0 + l1 + l2 exactly
          u32x4 x0, x1, x2;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float lo = q < 2 ? a0[2 * q] : a1[2 * q - 4], hi = q < 2 ? a0[2 * q + 1] : a1[2 * q - 3];
            const unsigned p0 = pk2s(lo, hi);
            const float rl = lo - __builtin_bit_cast(float, p0 << 16), rh = hi - __builtin_bit_cast(float, p0 & 0xffff0000u);
            const unsigned p1 = pk2s(rl, rh);
            x0[q] = p0;
            x1[q] = p1;
            x2[q] = pk2s(rl - __builtin_bit_cast(float, p1 << 16), rh - __builtin_bit_cast(float, p1 & 0xffff0000u));
          }
          const bf16x8s xa = __builtin_bit_cast(bf16x8s, x0), xb = __builtin_bit_cast(bf16x8s, x1), xc = __builtin_bit_cast(bf16x8s, x2);
#pragma unroll
          for (int t = 0; t < NT; ++t) {   // six limb products down to 2^-16 of the leading one, smallest first
            const bf16x8s w0 = __builtin_bit_cast(bf16x8s, bb[u & 1][t][0]), w1 = __builtin_bit_cast(bf16x8s, bb[u & 1][t][1]),
                          w2 = __builtin_bit_cast(bf16x8s, bb[u & 1][t][2]);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, xa, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, xc, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, xb, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, xa, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, xb, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, xa, acc[t], 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t)
              acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bb[u & 1][t][0][j], a0[j], acc[t], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t)
              acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bb[u & 1][t][1][j], a1[j], acc[t], 0, 0, 0);
        }
        if (++cur_ks == KS) {   // tile complete: its NP columns -> ring
          cur_ks = 0;
          const int idx = ((wave + 4 * cur_j) << 4) + p15;
          const int row = lo + cur_k * R + (idx >> lw), col = idx & (W - 1);
          if (row < hi) {
            float* dst = Pr + ((size_t)(row % ring) * W + col) * NP + 4 * g;
#pragma unroll
            for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4*>(dst + 16 * t) = acc[t];
          }
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (++cur_j == tpw) {
            cur_j = 0;
            lds_barrier();
            ++cur_k;
          }
        }
      }
    }
  }
}

static int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

static bool scatter_x3(const ConvArgs& a) {   // f32x3 / bf16 modes: limb products on the bf16 MFMA (fp32 results)
  static const bool off = dbg_env("EG_SCATTER_F32") != nullptr;
  return a.pk && !off;
}

// xstrips: 1 = a block takes whole image rows (the image's width must be a power of two and its product ring must fit the LDS);
// > 1 = column strips of kScatStripW outputs with a 64-column local ring (any width >= 64)
static bool scatter_geometry(const ConvArgs& a, int* S, int* R, size_t* lds, int* xstrips) {
  const int H = a.IH;
  const int NP = ((9 * a.N + 15) / 16) * 16;
  const bool x3 = scatter_x3(a);
  const size_t cap = (size_t)(x3 ? 160 : 150) * 1024;
  for (int strip = 0; strip < 2; ++strip) {
    const int W = strip ? 64 : a.IW;
    if (strip && a.IW < 64) return false;
    if (W < 4 || (W & (W - 1)) != 0) continue;
    const int r = std::max(1, 64 / W);
    if (r > H || (H % r) != 0 || (r * W) % 64 != 0) continue;
    const int xs = strip ? (a.IW + kScatStripW - 1) / kScatStripW : 1;
    // row strips: at least ~256 blocks when the batch (and the column strips) alone do not give them, height a multiple of R
    int strips = std::max(1, (256 + a.NB * xs - 1) / (a.NB * xs));
    strips = std::min(strips, H / r);
    int s = (H + strips - 1) / strips;
    s = (s + r - 1) / r * r;
    const size_t bytes = ((x3 ? (size_t)a.C * NP * 3 / 2 : (size_t)a.C * NP) + (size_t)(2 * r + 2) * W * NP) * sizeof(float);
    if (bytes > cap) continue;
    *S = s;
    *R = r;
    *lds = bytes;
    *xstrips = xs;
    return true;
  }
  return false;
}

bool conv_scatter_eligible(const ConvArgs& a) {
  static const bool off = dbg_env("EG_NO_SCATTER") != nullptr;
  if (off || !a.wgt || a.nphase != 1 || !(a.N == 3 || a.N == 4 || a.N == 8 || a.N == 12) || a.C % 32 != 0 || a.C < 32) return false;
  const ConvPhase& p = a.ph[0];
  if (p.Ty != 3 || p.Tx != 3 || a.kst != 1 || a.S != 1 || a.reflect) return false;
  if (!(a.D == 1 || a.D == -1) || p.offy != -a.D || p.offx != -a.D) return false;
  if (p.OHp != a.IH || p.OWp != a.IW || p.ophy != 0 || p.ophx != 0) return false;
  if (a.shuf ? (a.N != 12 || a.OHf != 2 * a.IH || a.OWf != 2 * a.IW) : (a.os != 1 || a.OHf != a.IH || a.OWf != a.IW)) return false;
  if (a.x_ps % 4 != 0 || (((uintptr_t)a.x) & 15) != 0 || a.x_bytes == 0) return false;
  int S, R, xs;
  size_t lds;
  return scatter_geometry(a, &S, &R, &lds, &xs);
}

template <int N, bool X3>
static void scatter_launch(const ConvArgs& a, int S, int R, size_t lds, int xstrips, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_scatter_kernel<N, X3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    attr_set = true;
  }
  const int strips = (a.IH + S - 1) / S;
  hipLaunchKernelGGL((conv_scatter_kernel<N, X3>), dim3(a.NB * strips * xstrips), dim3(512), lds, s, a, S, R,
                     ilog2(xstrips > 1 ? 64 : a.IW), xstrips);
}

const char* launch_conv_scatter(const ConvArgs& a, hipStream_t s) {
  int S, R, xs;
  size_t lds;
  scatter_geometry(a, &S, &R, &lds, &xs);
  if (scatter_x3(a)) {
    switch (a.N) {
      case 3: scatter_launch<3, true>(a, S, R, lds, xs, s); break;
      case 4: scatter_launch<4, true>(a, S, R, lds, xs, s); break;
      case 8: scatter_launch<8, true>(a, S, R, lds, xs, s); break;
      default: scatter_launch<12, true>(a, S, R, lds, xs, s); break;
    }
    return xs > 1 ? "conv_scatter<3x3,N<=12,limbs,column strips>" : "conv_scatter<3x3,N<=12,limbs>";
  }
  switch (a.N) {
    case 3: scatter_launch<3, false>(a, S, R, lds, xs, s); break;
    case 4: scatter_launch<4, false>(a, S, R, lds, xs, s); break;
    case 8: scatter_launch<8, false>(a, S, R, lds, xs, s); break;
    default: scatter_launch<12, false>(a, S, R, lds, xs, s); break;
  }
  return xs > 1 ? "conv_scatter<3x3,N<=12,column strips>" : "conv_scatter<3x3,N<=12>";
}

// ---------------------------------------------------------------------------------------------------------------------
// conv_smallc_kernel<NT>: convolution with few input channels (the 3-channel image, the classifier's 8-channel stem):
// K = taps x C is tiny (27 ... 392), the layer is bound by writing its output.  The gather form's flat-K path loads every
// operand element with its own scalar global load; here a block stages the input ROWS its 512 output pixels need in LDS
// with linear 16-byte loads (north_star: "LDS-staged input tiles and coalesced HBM reads"), keeps the whole filter
// [K][16 NT] in LDS, and builds the MFMA operands from LDS: no im2col buffer, every global byte read once per block.
//
// Block tile = one image x 512 (or 128) consecutive output pixels (sub-tiles of 128; a wave owns 2 x 16 pixels) x 16 NT
// output channels.  K is padded to a multiple of 16 and contracted in the order of the fragment layout: lane (p, q) of
// MFMA j in K group kg multiplies element k = 16 kg + 4 q + j, so the filter image [k / 4][n][k % 4] gives a lane its 4
// B operands of a column tile as one ds_read_b128; A operands are single LDS reads at staged[(iy - ylo) IW + ix][c]
// through a per-k offset table (taps outside the image: predicated to zero = SAME padding).
// conv.py:13-36 (conv2d forward, SAME), conv.py:278-285 (conv2d2), and the stride-2 convolution that is the input
// gradient of g_dconv_4 (conv.py:49-53).
// ---------------------------------------------------------------------------------------------------------------------
template <int NT, int PT>
__global__ __launch_bounds__(256, 2) void conv_smallc_kernel(const ConvArgs a, int K, int K16, int tiles_per_img, int subs, int dbg_) {
  constexpr int BN = NT * 16;
#ifdef EG_EXPERIMENTS
  const int dbg = dbg_;    // limiter probes: 1 no output stores, 2 no MFMAs, 4 no A-operand LDS reads, 8 no input staging
#else
  constexpr int dbg = 0;
#endif
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Bl = smem;                                          // [K16 / 4][BN][4]
  int2* kt = reinterpret_cast<int2*>(Bl + (size_t)K16 * BN); // [K16] {packed (dy, dx) or -1, LDS offset (dy IW + dx) C + c}
  int* wsrc = reinterpret_cast<int*>(kt + K16);              // [K16] filter element offset of (k, n = 0)
  float* Bias = reinterpret_cast<float*>(wsrc + K16);        // [BN] (zeros without a bias)
  float* Xs = Bias + BN;                                     // staged input rows [ylo, yhi] of the image
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p15 = lane & 15, kq = lane >> 4;
  const ConvPhase ph = a.ph[0];
  const int C = a.C, N = a.N, IH = a.IH, IW = a.IW, OWp = ph.OWp, npix = ph.OHp * ph.OWp;
  const int n0 = (int)blockIdx.y * BN;

  const int dy_lo = min(ph.offy, (ph.Ty - 1) * a.D + ph.offy), dy_hi = max(ph.offy, (ph.Ty - 1) * a.D + ph.offy);
  const float* bl_lane = Bl + ((size_t)kq * BN + p15) * 4;
  const int total_tiles = a.NB * tiles_per_img;

  // The rows of the NEXT tile are fetched into registers before the current tile is multiplied and stored, and written to
  // LDS after it: the fetch latency runs under the MFMAs, and (vmcnt is one in-order counter for loads and stores on
  // gfx9) waiting for those loads does not wait for the output stores issued after them.  Round barriers are LDS-only.
  constexpr int PF = 6;   // 16-byte registers per thread: tiles stage <= 24 KB of rows (smallc_geometry)
  auto lds_barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  f32x4 pf[PF];
  int pf_n4 = 0;
  auto tile_rows = [&](int tile, int* nb_, int* q_lo_, int* q_hi_, int* ylo_) __attribute__((always_inline)) {
    const int nb = tile / tiles_per_img, q_lo = (tile - nb * tiles_per_img) * subs * (64 * PT);
    const int q_hi = min(q_lo + subs * (64 * PT), npix) - 1;
    const int ylo = max(0, (q_lo / OWp) * a.S + dy_lo), yhi = min(IH - 1, (q_hi / OWp) * a.S + dy_hi);
    *nb_ = nb; *q_lo_ = q_lo; *q_hi_ = q_hi; *ylo_ = ylo;
    return max(0, yhi - ylo + 1) * IW * C >> 2;   // 16-byte units
  };
  auto prefetch = [&](int tile) __attribute__((always_inline)) {
    pf_n4 = 0;
    if (tile >= total_tiles || (dbg & 8)) return;
    int nb, q_lo, q_hi, ylo;
    pf_n4 = tile_rows(tile, &nb, &q_lo, &q_hi, &ylo);
    const f32x4* src = reinterpret_cast<const f32x4*>(a.x + ((size_t)nb * IH + ylo) * IW * C);
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int i = u * 256 + tid;
      if (i < pf_n4) pf[u] = src[i];
    }
  };
  if (dbg & 16) return;    // probe: launch + LDS allocation only
  // the first tile's input rows are requested before anything else: their round trip runs under the table / filter staging
  prefetch(blockIdx.x);

  // ---- tap table ---------------------------------------------------------------------------------------------------
  for (int k = tid; k < K16; k += 256) {
    int2 e = {-1, 0};
    int ws = 0;
    if (k < K) {
      const int tap = k / C, c = k - tap * C;
      const int ty = tap / ph.Tx, tx = tap - ty * ph.Tx;
      const int dy = ty * a.D + ph.offy, dx = tx * a.D + ph.offx;
      e.x = ((dy + 512) << 16) | (dx + 512);
      e.y = (dy * IW + dx) * C + c;
      ws = (int)(((ph.kh0 + ty * a.kst) * a.KW + (ph.kw0 + tx * a.kst)) * a.w_ts + (long long)c * a.ldw);
    }
    kt[k] = e;
    wsrc[k] = ws;
  }
  for (int n = tid; n < BN; n += 256) Bias[n] = (a.bias && n0 + n < N) ? a.bias[n0 + n] : 0.f;
  __syncthreads();
  // ---- filter -> LDS, once per block (the block then walks over pixel tiles): n fastest (coalesced), 8 loads per thread
  // in flight; K padding rows are zero -----------------------------------------------------------------------------------
  if (BN >= 64 && n0 + BN <= N && (a.ldw & 3) == 0 && (a.w_ts & 3) == 0 && (((uintptr_t)a.wgt) & 15) == 0) {
    // whole 16-byte loads along n (a filter row of BN floats is contiguous), scattered to the [k / 4][n][k % 4] image
    constexpr int Q = BN / 4;
    const int total = K16 * Q;
    for (int base = 0; base < total; base += 256 * 4) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = base + u * 256 + tid;
        const int k = i / Q, n4 = i - k * Q;
        v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (i < total && k < K) v[u] = *reinterpret_cast<const f32x4*>(a.wgt + (size_t)wsrc[k] + n0 + 4 * n4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = base + u * 256 + tid;
        const int k = i / Q, n4 = i - k * Q;
        if (i < total) {
          float* d = Bl + ((k >> 2) * BN + 4 * n4) * 4 + (k & 3);
#pragma unroll
          for (int r = 0; r < 4; ++r) d[4 * r] = v[u][r];
        }
      }
    }
  } else {
    const int total = K16 * BN;
    for (int base = 0; base < total; base += 256 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * 256 + tid;
        const int k = i / BN, n = i - k * BN;   // BN is a power of two
        v[u] = (i < total && k < K && n0 + n < N) ? a.wgt[(size_t)wsrc[k] + n0 + n] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * 256 + tid;
        const int k = i / BN, n = i - k * BN;
        if (i < total) Bl[((k >> 2) * BN + n) * 4 + (k & 3)] = v[u];
      }
    }
  }
  __syncthreads();   // filter image complete
  if (dbg & 32) return;    // probe: ... + table and filter staging

  for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    int nb, q_lo, q_hi, ylo;
    tile_rows(tile, &nb, &q_lo, &q_hi, &ylo);
    lds_barrier();   // the previous tile's rows are no longer read
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int i = u * 256 + tid;
      if (i < pf_n4) reinterpret_cast<f32x4*>(Xs)[i] = pf[u];
    }
    lds_barrier();
    prefetch(tile + gridDim.x);

    for (int sub = 0; sub < subs; ++sub) {
      const int qb = q_lo + sub * (64 * PT) + wave * (16 * PT);
      if (qb > q_hi || (dbg & 64)) break;
      int oyS[PT], oxS[PT], base[PT];
      bool pv[PT];
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        const int q = qb + pt * 16 + p15;
        pv[pt] = q <= q_hi;
        const int oy = q / OWp, ox = q - oy * OWp;
        oyS[pt] = oy * a.S;
        oxS[pt] = ox * a.S;
        base[pt] = ((oyS[pt] - ylo) * IW + oxS[pt]) * C;
      }
      f32x4 acc[PT][NT];
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[pt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // A operands of K group kg: table entry -> predicate + staged-row address -> LDS read.  They are requested one group
      // ahead (av0 / av1 alternate), so the two dependent LDS round trips run under the previous group's MFMAs.
      auto load_av = [&](int kg, float (&av)[PT][4]) __attribute__((always_inline)) {
        if (dbg & 128) {   // probe: no table reads, predicates or operand reads at all
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) av[pt][j] = 1.f;
          return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int2 e = kt[kg * 16 + kq * 4 + j];
          const int dy = (e.x >> 16) - 512, dx = (e.x & 0xffff) - 512;
#pragma unroll
          for (int pt = 0; pt < PT; ++pt) {
            const bool ok = pv[pt] && e.x >= 0 && (unsigned)(oyS[pt] + dy) < (unsigned)IH && (unsigned)(oxS[pt] + dx) < (unsigned)IW;
            if (dbg & 4) { av[pt][j] = ok ? 1.f : 0.f; continue; }
            const float v = Xs[ok ? base[pt] + e.y : 0];
            av[pt][j] = ok ? v : 0.f;
          }
        }
      };
      auto mfma_group = [&](int kg, const float (&av)[PT][4]) __attribute__((always_inline)) {
        const float* bk = bl_lane + (size_t)kg * 4 * BN * 4;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const f32x4 b = (dbg & 256) ? (f32x4){1.f, 2.f, 3.f, 4.f} : *reinterpret_cast<const f32x4*>(bk + t * 16 * 4);   // probe 256: no filter reads
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
              if (dbg & 2) asm volatile("" : "+v"(acc[pt][t]) : "v"(b[j]), "v"(av[pt][j]));
              else acc[pt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j], av[pt][j], acc[pt][t], 0, 0, 0);
            }
        }
      };
      const int nkg = K16 / 16;
      float av0[PT][4], av1[PT][4];
      load_av(0, av0);
      int kg = 0;
      for (; kg + 1 < nkg; kg += 2) {
        load_av(kg + 1, av1);
        mfma_group(kg, av0);
        if (kg + 2 < nkg) load_av(kg + 2, av0);
        mfma_group(kg + 1, av1);
      }
      if (kg < nkg) mfma_group(kg, av0);
      // epilogue: lane holds pixel p15, channels 16 t + 4 kq .. + 3.  All loads (the bias from LDS, the old output when
      // accumulating) are issued before the first store: vmcnt is one in-order counter, a load issued after a store would
      // wait for that store's round trip.
      const bool vec_out = (a.y_ps & 3) == 0 && (((uintptr_t)a.y) & 15) == 0 && (N & 3) == 0;
      float* yrow[PT];
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        const int q = qb + pt * 16 + p15;
        const int oy = q / OWp, ox = q - oy * OWp;
        yrow[pt] = a.y + ((size_t)(nb * a.OHf + oy * a.os + ph.ophy) * a.OWf + ox * a.os + ph.ophx) * a.y_ps;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int nl = t * 16 + kq * 4;
          acc[pt][t] += *reinterpret_cast<const f32x4*>(Bias + nl);
          if (a.act == 1) {   // ReLU / LeakyReLU(0.2) fused behind the bias (conv_block of d_conv_0, the encoder stem)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[pt][t][r] = fmaxf(acc[pt][t][r], 0.f);
          } else if (a.act == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[pt][t][r] = fmaxf(acc[pt][t][r], 0.2f * acc[pt][t][r]);
          }
          if (a.accumulate && pv[pt]) {
            if (vec_out) {
              if (n0 + nl < N) acc[pt][t] += *reinterpret_cast<const f32x4*>(yrow[pt] + n0 + nl);
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (n0 + nl + r < N) acc[pt][t][r] += yrow[pt][n0 + nl + r];
            }
          }
        }
      }
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        if (!pv[pt] || ((dbg & 1) && acc[pt][0][0] != 12345.678f)) continue;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int n = n0 + t * 16 + kq * 4;
          if (vec_out) {
            if (n < N) *reinterpret_cast<f32x4*>(yrow[pt] + n) = acc[pt][t];
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (n + r < N) yrow[pt][n + r] = acc[pt][t][r];
          }
        }
      }
    }
  }
}

static bool smallc_geometry(const ConvArgs& a, int* K, int* K16, int* nt, int* tiles, int* subs, size_t* lds) {
  const ConvPhase& p = a.ph[0];
  *K = p.Ty * p.Tx * a.C;
  *K16 = (*K + 15) / 16 * 16;
  const int n16 = (a.N + 15) / 16;
  *nt = n16 >= 8 ? 8 : (n16 >= 4 ? 4 : (n16 >= 2 ? 2 : 1));
  const int npix = p.OHp * p.OWp;
  // a block's tile = subs x 128 output pixels of one image: 512 or 256 (a wave owns 4 x 16 pixels of a 256-pixel sub-tile) whose input rows fit
  // the 24 KB register prefetch (6 x 16 bytes per thread) and that leaves the ~512 resident blocks >= 4 tiles each
  const int ny = (a.N + *nt * 16 - 1) / (*nt * 16);
  int xs_floats = 0;
  *subs = 0;
  for (int sb = 2; sb >= 1; sb >>= 1) {
    const int tpx = sb * 256;
    const int orows = std::min(p.OHp, (tpx + p.OWp - 1) / p.OWp + 1);
    const int rows = std::min(a.IH, (orows - 1) * a.S + p.Ty + 1);
    xs_floats = rows * a.IW * a.C;
    const bool enough_tiles = (long long)a.NB * ((npix + tpx - 1) / tpx) * ny >= 2048;
    if (xs_floats <= 6 * 1024 && (enough_tiles || sb == 1)) {
      *subs = sb;
      break;
    }
  }
  if (*subs == 0) return false;
  *tiles = (npix + *subs * 256 - 1) / (*subs * 256);
  *lds = ((size_t)*K16 * (*nt * 16) + 3 * (size_t)*K16 + (size_t)(*nt * 16) + (size_t)xs_floats) * sizeof(float);
  return *lds <= 80 * 1024;
}

bool conv_smallc_eligible(const ConvArgs& a) {
  static const bool off = dbg_env("EG_NO_SMALLC") != nullptr;
  if (off || !a.wgt || a.nphase != 1 || a.C > 8 || a.x_ps != a.C || a.reflect || a.shuf || a.ksplit > 1) return false;
  const ConvPhase& p = a.ph[0];
  if (!(a.D == 1 || a.D == -1) || a.S < 1 || a.S > 2) return false;
  if (p.Ty > 8 || p.Tx > 8) return false;
  if ((a.IW * a.C) % 4 != 0 || (((uintptr_t)a.x) & 15) != 0) return false;   // rows are staged in 16-byte units
  if ((long long)a.NB * a.IH * a.IW * a.C >= (1ll << 31) || (long long)p.Ty * p.Tx * a.w_ts >= (1ll << 31)) return false;
  int K, K16, nt, tiles, subs;
  size_t lds;
  return smallc_geometry(a, &K, &K16, &nt, &tiles, &subs, &lds);
}

template <int NT>
static void smallc_launch(const ConvArgs& a, int K, int K16, int tiles, int subs, size_t lds, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_smallc_kernel<NT, 4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              80 * 1024);
    attr_set = true;
  }
  // persistent blocks: the filter is staged once per block, which then walks over (image, 512-pixel tile) pairs
  const int ny = (a.N + NT * 16 - 1) / (NT * 16);
  dim3 grid(std::min(a.NB * tiles, std::max(1, 512 / ny)), ny);
  static const int dbg = dbg_env("EG_SMALLC_DBG") ? atoi(dbg_env("EG_SMALLC_DBG")) : 0;
  hipLaunchKernelGGL((conv_smallc_kernel<NT, 4>), grid, dim3(256), lds, s, a, K, K16, tiles, subs, dbg);
}

const char* launch_conv_smallc(const ConvArgs& a, hipStream_t s) {
  int K, K16, nt, tiles, subs;
  size_t lds;
  smallc_geometry(a, &K, &K16, &nt, &tiles, &subs, &lds);
  switch (nt) {
    case 1: smallc_launch<1>(a, K, K16, tiles, subs, lds, s); break;
    case 2: smallc_launch<2>(a, K, K16, tiles, subs, lds, s); break;
    case 4: smallc_launch<4>(a, K, K16, tiles, subs, lds, s); break;
    default: smallc_launch<8>(a, K, K16, tiles, subs, lds, s); break;
  }
  return "conv_smallc<C<=8>";
}

}  // namespace eg
