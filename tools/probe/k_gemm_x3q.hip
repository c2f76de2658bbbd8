// f32x3 halo convolution, TWO blocks per CU: 128 pixels x 128 channels per block, K chunks of 16 channels.
// EXPERIMENTS BUILD ONLY (make EXPERIMENTS=1, EG_X3Q=1): built in round 4 to answer "what would independent blocks on a CU buy the
// f32x3 mode?" -- parity-green (tests/test_x3_gpu.py::test_x3_ops_at_fp32_tolerances with EG_X3Q=1), and EQUAL to the one-block
// kernel: unit 1 / 2 / 3 Conv_2 0.345 / 0.341 / 0.343 ms against 0.355 / 0.336 / 0.326 (profiles/r04ad_microbench_x3q.txt), the
// iteration within noise.  Its K loop pays twice the barriers, DMA requests and waits per MFMA (16-channel steps), which is what
// the neighbour block then hides; the shipped kernel stays k_gemm_x3h.hip.  docs/experiments.md has the account.
//
//   conv_x3q_kernel : Y[pix_out][n] (+)= b[n] + sum_taps sum_c X[pix_out + d(tap)][c] * W[tap][c][n]   -- fp32 results from exact
//                     three-limb bf16 operand splits (k_gemm_x3.hip), for the stride-1 layers with 2 .. 9 taps per phase and at
//                     least 128 output channels on maps at least 16 wide: the layers k_gemm_x3h.hip serves with ONE 8-wave block
//                     per CU (edgegan/nn/modules/conv.py:13-36,39-58,133-243).
//
// Why.  The halo kernel keeps 150 KB of LDS (a 256-pixel halo image of three limb planes at 32 channels per chunk + a 3-slot
// filter ring), so nothing runs beside a block: its prologue, its output burst, the limb split of every chunk's halo between
// two barriers and the barrier bubbles of eight lock-stepped waves all leave the matrix pipe idle -- 69 % busy
// (profiles/r04z_pmc_utilisation.csv; the tile boundary alone costs ~9 us, profiles/r04ab_microbench_spread_reads.txt).  The
// plane-fed bf16 kernel (k_conv_bf16p.hip) showed what independent blocks on a CU are worth.  Three limb planes fit twice
// into a CU when the K chunk is 16 channels instead of 32:
//     halo   2 buffers x 3 planes x 240 rows x 32 B = 46 KB      (rows = TI x (TH + taps - 1) x pitch 20, 16-wide tiles)
//     filter 2 slots   x 3 planes x 128 columns x 32 B = 24.6 KB
//   = 70.7 KB per block.  256 threads = 4 waves (2 x 2, wave tile 64 x 64: the halo kernel's accumulators), two blocks = the same
//   eight waves per CU -- but each SIMD now holds one wave of EITHER block, and a wave alone can keep the pipe busy.
// * K step = one tap x 16 channels = 6 limb products x 4 MFMAs (v_mfma_f32_32x32x16_bf16, one per accumulator).  The whole
//   fragment set of step k + 1 (6 groups of two 16-byte reads) is requested under step k's products, one group per product
//   (k_gemm_x3h.hip, round 4), into the second register set; a step's top is ONE wait + barrier.
// * With two halo buffers the next chunk's staging leaves the critical path: its fp32 quads are loaded at the chunk's first
//   step and split + stored into the OTHER buffer two steps before the chunk ends -- beside the neighbour block's MFMAs.
// * Taps and chunk pairs are unrolled at compile time (k_conv_bf16p.hip): buffer, slot, register set, row shift and
//   filter offset of a step are immediates.  LDS rows are 32 bytes; the 16-byte K half of row r sits at half (kh ^ (hx >> 3) & 1)
//   (hx = the pixel's halo column): the 16 rows a quarter wave reads are 8 bank-eighths twice, in different halves.
// * Filter tiles arrive by `buffer_load ... lds` from the K-step-major limb planes ([tap][C / 32][N][32] bf16: a 16-channel
//   half of a row is 32 contiguous bytes), the half swizzle applied on the global side.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace eg {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {
__device__ __forceinline__ unsigned pk2q(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_q(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi_q(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
// (x, y) -> packed limb pairs: x = lo(p0) + lo(p1) + lo(p2) exactly (k_gemm_x3.hip: split_pair)
__device__ __forceinline__ void split2q(float x, float y, unsigned& p0, unsigned& p1, unsigned& p2) {
  p0 = pk2q(x, y);
  const float rx = x - lo_q(p0), ry = y - hi_q(p0);
  p1 = pk2q(rx, ry);
  p2 = pk2q(rx - lo_q(p1), ry - hi_q(p1));
}
__device__ __forceinline__ int reflect_q(int i, int n) {
  i = i < 0 ? -i : i;
  return i >= n ? 2 * (n - 1) - i : i;
}
template <int N, class F>
__device__ __forceinline__ void static_for_q(F&& f) {
  if constexpr (N > 0) {
    static_for_q<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}
}  // namespace

struct QGeom {
  int lTH, TI;             // tile = TI images x 2^lTH rows x 16 columns = 128 output pixels
  int tiles_x, tiles_y;    // tiles per image
};

constexpr int kHRQ = 200;  // halo rows (pitch 20): 1 image x (8 + 2) rows x 20

// PH: the tap grids of the launch's phases -- 0: 3x3, 1: 2x2 (the four phases of a 4x4 stride-2 input gradient), 2: any of
// 3x3 / 3x2 / 2x3 / 2x2 per phase (5x5 stride-2 transposed convolutions)
template <bool SPLITK, int PH>
__global__ __launch_bounds__(256, 2) void conv_x3q_kernel(const ConvArgs a, const QGeom g) {
  constexpr int BN = 128, NA = (kHRQ * 4 + 255) / 256;   // NA: staged quads (4 channels of a halo row) per thread and chunk
  constexpr int APLANE = kHRQ * 32, ABUF = 3 * APLANE;  // bytes of a halo limb plane / of a halo buffer
  constexpr int BPLANE = BN * 32, BSLOT = 3 * BPLANE;   // bytes of a filter limb plane / of a ring slot
  __shared__ __attribute__((aligned(16))) unsigned char Ah[2 * ABUF];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[3 * BSLOT];   // 3-slot ring where the taps of a chunk are a multiple of 3 (the slot of a step is then a compile-time constant), else 2 slots

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const unsigned gxy = gridDim.x * gridDim.y;
  const unsigned vblk = xcd_remap((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, gxy * gridDim.z, 0);
  const int bz = __builtin_amdgcn_readfirstlane(vblk / gxy), bxy = __builtin_amdgcn_readfirstlane(vblk % gxy);
  const int bx = bxy / (int)gridDim.y, by = bxy - bx * (int)gridDim.y;
  const int zphase = SPLITK ? bz / a.ksplit : bz;
  const int zsplit = SPLITK ? bz - zphase * a.ksplit : 0;
  const ConvPhase ph = a.ph[zphase];
  const int IH = a.IH, IW = a.IW, C = a.C, N = a.N, D = a.D;
  const int nTy = ph.Ty, nTx = ph.Tx;
  const int TH = 1 << g.lTH;
  const int tpi = g.tiles_x * g.tiles_y;
  const int tn = bx / tpi, trem = bx - tn * tpi;
  const int tyy = trem / g.tiles_x, txx = trem - tyy * g.tiles_x;
  const int img0 = tn * g.TI, oy0 = tyy * TH, ox0 = txx * 16;
  const int n0 = by * BN;
  const int offy = ph.offy, offx = ph.offx;
  const int dminy = min(offy, (nTy - 1) * D + offy), dminx = min(offx, (nTx - 1) * D + offx);
  const int HH = TH + nTy - 1, HW = 16 + nTx - 1, HPI = HH * 20;

  // ---- halo staging: unit u = tid + 256 j = (LDS row u >> 2, channel quad u & 3); rows without a pixel read zeros ------------
  unsigned goff[NA];
  int sdst[NA];      // byte offset of the quad inside a limb plane, -1: no row
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int u = tid + 256 * j, row = u >> 2, q = u & 3;
    unsigned off = 0x80000000u;
    sdst[j] = -1;
    const int ti = row / HPI, rem = row - ti * HPI;
    const int hy = rem / 20, hx = rem - hy * 20;
    if (row < kHRQ && ti < g.TI && hx < HW) {
      int iy = oy0 + dminy + hy, ix = ox0 + dminx + hx;
      if (a.reflect) {
        iy = reflect_q(iy, IH);
        ix = reflect_q(ix, IW);
      }
      if ((unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW)
        off = (unsigned)(((((img0 + ti) * IH + iy) * IW + ix) * a.x_ps + q * 4) * 4);
      sdst[j] = row * 32 + (((q >> 1) ^ ((hx >> 3) & 1)) * 16) + (q & 1) * 8;
    }
    goff[j] = off;
  }
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wres =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.wgt_bf), 0, a.wbf_bytes, 0x00020000);
  f32x4 ra[NA];
  auto load_halo = [&](int cc) __attribute__((always_inline)) {   // cc: 16-channel chunk
#pragma unroll
    for (int j = 0; j < NA; ++j) ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, goff[j], cc * 64, 0));
  };
  auto store_halo = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NA; ++j)
      if (sdst[j] >= 0) {
        unsigned p0a, p1a, p2a, p0b, p1b, p2b;
        split2q(ra[j][0], ra[j][1], p0a, p1a, p2a);
        split2q(ra[j][2], ra[j][3], p0b, p1b, p2b);
        unsigned char* q = &Ah[buf * ABUF + sdst[j]];
        *reinterpret_cast<u32x2*>(q) = (u32x2){p0a, p0b};
        *reinterpret_cast<u32x2*>(q + APLANE) = (u32x2){p1a, p1b};
        *reinterpret_cast<u32x2*>(q + 2 * APLANE) = (u32x2){p2a, p2b};
      }
  };

  // ---- filter ring: wave w's instruction for plane l fills columns 32 w .. 32 w + 31 of the slot's plane; lane = (column, half) --
  const int fcol = 32 * wave + (lane >> 1);
  const unsigned woff = (unsigned)(((size_t)min(n0 + fcol, N - 1) * 32 + (((lane & 1) ^ ((fcol >> 3) & 1)) * 8)) * 2);
  const int cpt = C / 32, wtile = N * 32 * 2;
  const int plane_bytes = (int)(a.wbf_plane * 2);
  const int vcpt = C / 16;                  // K chunks the loop walks
  int cc_lo = 0, cc_hi = vcpt;
  if constexpr (SPLITK) {
    const int per = (vcpt + a.ksplit - 1) / a.ksplit;
    cc_lo = min(vcpt, zsplit * per);
    cc_hi = min(vcpt, cc_lo + per);
  }
  const int ncc = cc_hi - cc_lo;

  // ---- fragments ---------------------------------------------------------------------------------------------------------------
  const int l31 = lane & 31, lh = lane >> 5;
  const unsigned ah_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&Ah[0];
  const unsigned bs_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&Bs[0];
  // the wave's two 32-row blocks are two tile rows = 40 halo rows apart and share the pixel column: an immediate
  unsigned abase[3];
  {
    const int r = wm * 64 + l31;
    const int ti = r >> (4 + g.lTH), py = (r >> 4) & (TH - 1), px = r & 15;
    const int row0 = ti * HPI + py * 20 + px;
#pragma unroll
    for (int e = 0; e < 3; ++e) abase[e] = ah_base + (row0 + e) * 32 + ((lh ^ (((px + e) >> 3) & 1)) * 16);
  }
  const unsigned bb0 = bs_base + (wn * 64 + l31) * 32 + ((lh ^ ((l31 >> 3) & 1)) * 16);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#define EG_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
  struct Frags {
    bf16x8 a[3][2], b[3][2];   // [limb][32-row block] / [limb][32-column block]
  };
  // s_waitcnt lgkmcnt(0) (+ barrier) with every register of the fragment set as in/out operand
  auto land = [&](Frags& f, auto BAR) __attribute__((always_inline)) {
    if constexpr (decltype(BAR)::value)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier"
                   : "+v"(f.a[0][0]), "+v"(f.a[0][1]), "+v"(f.a[1][0]), "+v"(f.a[1][1]), "+v"(f.a[2][0]), "+v"(f.a[2][1]), "+v"(f.b[0][0]),
                     "+v"(f.b[0][1]), "+v"(f.b[1][0]), "+v"(f.b[1][1]), "+v"(f.b[2][0]), "+v"(f.b[2][1])::"memory");
    else
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(f.a[0][0]), "+v"(f.a[0][1]), "+v"(f.a[1][0]), "+v"(f.a[1][1]), "+v"(f.a[2][0]), "+v"(f.a[2][1]), "+v"(f.b[0][0]),
                     "+v"(f.b[0][1]), "+v"(f.b[1][0]), "+v"(f.b[1][1]), "+v"(f.b[2][0]), "+v"(f.b[2][1])::"memory");
  };
  constexpr std::true_type WITH_BARRIER{};
  constexpr std::false_type NO_BARRIER{};

  auto kloop = [&](auto TYc, auto TXc) __attribute__((always_inline)) {
    constexpr int TY = decltype(TYc)::value, TX = decltype(TXc)::value, NT = TY * TX;
    // byte offset of tap t's filter tiles inside a limb plane ([tap][C / 32][N][32] bf16)
    int tapoff[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int f = t / TX, e = t - f * TX;
      const int ty = D > 0 ? f : TY - 1 - f, tx = D > 0 ? e : TX - 1 - e;
      tapoff[t] = ((ph.kh0 + ty * a.kst) * a.KW + (ph.kw0 + tx * a.kst)) * cpt * wtile;
    }
    auto wrow_of = [&](int cc) __attribute__((always_inline)) { return (cc >> 1) * wtile + (cc & 1) * 32; };
    Frags fs[2] = {};
    // one fragment group of step (halo buffer BUF, tap T, ring slot SLOT): G = 0 / 2 / 4 the halo limbs, 1 / 3 / 5 the filter limbs
    auto read_group = [&](Frags& fr, auto BUFc, auto Tc, auto SLOTc, auto Gc) __attribute__((always_inline)) {
      constexpr int BUF = decltype(BUFc)::value, T = decltype(Tc)::value, SLOT = decltype(SLOTc)::value, G = decltype(Gc)::value;
      constexpr int f = T / TX, e = T - f * TX, L = G >> 1;
      if constexpr ((G & 1) == 0) {
        // (named copies: a variable that a generic lambda uses ONLY as an asm operand is not captured by hipcc 7.2)
        const unsigned p = abase[e];
        EG_DSR(fr.a[L][0], p, BUF * ABUF + L * APLANE + f * 20 * 32);
        EG_DSR(fr.a[L][1], p, BUF * ABUF + L * APLANE + (f + 2) * 20 * 32);
      } else {
        const unsigned p = bb0;
        EG_DSR(fr.b[L][0], p, SLOT * BSLOT + L * BPLANE);
        EG_DSR(fr.b[L][1], p, SLOT * BSLOT + L * BPLANE + 32 * 32);
      }
    };
    auto read_step = [&](Frags& fr, auto BUFc, auto Tc, auto SLOTc) __attribute__((always_inline)) {
      static_for_q<6>([&](auto Gc) __attribute__((always_inline)) { read_group(fr, BUFc, Tc, SLOTc, Gc); });
    };
    auto request_b = [&](auto SLOTc, int wsoff) __attribute__((always_inline)) {
      constexpr int SLOT = decltype(SLOTc)::value;
      const unsigned voff = woff;
#pragma unroll
      for (int l = 0; l < 3; ++l)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wres, (__attribute__((address_space(3))) void*)&Bs[SLOT * BSLOT + l * BPLANE + wave * 1024], 16, voff,
                                                 wsoff + l * plane_bytes, 0, 0);
    };
    // limb product (AL, BL) of fragment set f: four MFMAs; RD: the reads issued behind the first of them
#define EG_QPROD(f, AL, BL, RD)                                                                              \
  do {                                                                                                       \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b[BL][0], f.a[AL][0], acc[0][0], 0, 0, 0);         \
    RD;                                                                                                      \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b[BL][1], f.a[AL][0], acc[0][1], 0, 0, 0);         \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b[BL][0], f.a[AL][1], acc[1][0], 0, 0, 0);         \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b[BL][1], f.a[AL][1], acc[1][1], 0, 0, 0);         \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
  } while (0)
    constexpr std::integral_constant<int, 0> I0{};
    constexpr std::integral_constant<int, 1> I1{};
    constexpr bool RING3 = NT % 3 == 0;     // filter tile k + 3 requested in step k (two steps of lead) instead of tile k + 2 (one)
    constexpr int LEAD = RING3 ? 3 : 2;
    constexpr std::integral_constant<int, 2> I2{};
    if (ncc > 0) {
      load_halo(cc_lo);
      request_b(I0, tapoff[0] + wrow_of(cc_lo));
      request_b(I1, tapoff[1] + wrow_of(cc_lo));     // (NT >= 2)
      if constexpr (RING3) request_b(I2, tapoff[2] + wrow_of(cc_lo));
      store_halo(0);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      read_step(fs[0], I0, I0, I0);
    }
    // one chunk, the H-th of a pair (H = 0: halo buffer 0, H = 1: buffer 1); global step k = (2 p + H) NT + T, so the
    // register set of a step is (H NT + T) & 1 and its ring slot T % 3 (three slots: NT is a multiple of 3) or (H NT + T) & 1
    auto chunk = [&](auto Hc, int cc, bool has_next) __attribute__((always_inline)) {
      constexpr int H = decltype(Hc)::value;
      const int wrow = wrow_of(cc), wrow_next = has_next ? wrow_of(cc + 1) : 0;
      static_for_q<NT>([&](auto Tc) __attribute__((always_inline)) {
        constexpr int T = decltype(Tc)::value;
        constexpr int SET = (H * NT + T) & 1;
        constexpr int SLOT = RING3 ? T % 3 : SET;                  // ring slot of tile k (and of tile k + LEAD)
        constexpr int NSLOT = RING3 ? (T + 1) % 3 : SET ^ 1;       // ... of tile k + 1
        constexpr std::integral_constant<int, SLOT> SLOTc{};
        constexpr std::integral_constant<int, NSLOT> NSLOTc{};
        // tile k + 1 has landed.  Newer than it in this wave's (in-order) queue: with three slots the 3 pieces of tile k + 2
        // where that tile exists; in the two steps behind a chunk's first (one step with two slots) the next chunk's halo loads
        {
          const bool next_tile = RING3 && (T + 2 < NT || has_next);
          const bool halo_new = has_next && (T == 1 || (RING3 && T == 2));
          if (next_tile && halo_new) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 + NA) : "memory");
          else if (next_tile) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
          else if (halo_new) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        land(fs[SET], WITH_BARRIER);
        // the next chunk's halo: split + stored into the other buffer two steps before the chunk ends (its loads are older than
        // anything the wait above left in flight; the barrier of step NT - 1 orders the stores before that step's look-ahead reads)
        if constexpr (T == NT - 2) {
          if (has_next) store_halo(H ^ 1);
          __builtin_amdgcn_sched_barrier(0);
        }
        // filter tile k + LEAD -> the slot tile k was read from
        if constexpr (T + LEAD < NT) request_b(SLOTc, tapoff[T + LEAD] + wrow);
        else if (has_next) request_b(SLOTc, tapoff[T + LEAD - NT] + wrow_next);
        __builtin_amdgcn_sched_barrier(0);   // (the vmcnt waits count on the issue order: filter tile, then halo loads)
        if constexpr (T == 0)
          if (has_next) load_halo(cc + 1);
        __builtin_amdgcn_sched_barrier(0);
        // the six limb products of step k; the fragment groups of step k + 1 return under them into the other register set
        constexpr bool same = T + 1 < NT;
        constexpr std::integral_constant<int, same ? H : (H ^ 1)> NBUFc{};
        constexpr std::integral_constant<int, same ? T + 1 : 0> NTAPc{};
        const bool rd = same || has_next;
        Frags& f = fs[SET];
        Frags& n = fs[SET ^ 1];
        EG_QPROD(f, 0, 0, if (rd) read_group(n, NBUFc, NTAPc, NSLOTc, std::integral_constant<int, 0>{}));
        EG_QPROD(f, 1, 0, if (rd) read_group(n, NBUFc, NTAPc, NSLOTc, std::integral_constant<int, 1>{}));
        EG_QPROD(f, 2, 0, if (rd) read_group(n, NBUFc, NTAPc, NSLOTc, std::integral_constant<int, 2>{}));
        EG_QPROD(f, 0, 1, if (rd) read_group(n, NBUFc, NTAPc, NSLOTc, std::integral_constant<int, 3>{}));
        EG_QPROD(f, 0, 2, if (rd) read_group(n, NBUFc, NTAPc, NSLOTc, std::integral_constant<int, 4>{}));
        EG_QPROD(f, 1, 1, if (rd) read_group(n, NBUFc, NTAPc, NSLOTc, std::integral_constant<int, 5>{}));
      });
    };
    int c = 0;
    for (; c + 1 < ncc; c += 2) {
      chunk(I0, cc_lo + c, true);
      chunk(I1, cc_lo + c + 1, c + 2 < ncc);
    }
    if (c < ncc) chunk(I0, cc_lo + c, false);
    // nothing is in flight here (the last step issued no reads); sets whose reads were never issued hold zeros
    land(fs[0], NO_BARRIER);
    land(fs[1], NO_BARRIER);
    __builtin_amdgcn_sched_barrier(0);
#undef EG_QPROD
  };
  constexpr std::integral_constant<int, 2> K2{};
  constexpr std::integral_constant<int, 3> K3{};
  if constexpr (PH == 0) {
    kloop(K3, K3);
  } else if constexpr (PH == 1) {
    kloop(K2, K2);
  } else {
    if (nTy == 3 && nTx == 3) kloop(K3, K3);
    else if (nTy == 3) kloop(K3, K2);
    else if (nTx == 3) kloop(K2, K3);
    else kloop(K2, K2);
  }
#undef EG_DSR

  // D^T layout of the 32x32 accumulator: lane l holds pixel row l & 31 and, in registers 4q .. 4q+3, the four consecutive
  // output channels 8q + 4 (l >> 5) .. +3 of its 32-channel block: one 16-byte store per register quad
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = wm * 64 + i * 32 + l31;
    const int ti = r >> (4 + g.lTH), py = (r >> 4) & (TH - 1), px = r & 15;
    const int nb = img0 + ti, oy = oy0 + py, ox = ox0 + px;
    float* yrow = (SPLITK ? a.ws + (size_t)zsplit * a.y_total : a.y) +
                  ((size_t)(nb * a.OHf + oy * a.os + ph.ophy) * a.OWf + ox * a.os + ph.ophx) * a.y_ps;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn * 64 + j * 32 + q * 8 + lh * 4;
        if (n < N) {
          float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
          if constexpr (!SPLITK) {
            if (a.bias) {
              const float4 b4 = *reinterpret_cast<const float4*>(a.bias + n);
              v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
            }
            if (a.accumulate) {
              const float4 o = *reinterpret_cast<const float4*>(yrow + n);
              v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            if (a.epi_act) {   // dL/d(pre-activation) = dL/d(out) * act'(out), slopes as act_grad_from_out (k_norm.hip)
              size_t ei = (size_t)(yrow - a.y) + n;
              if (a.epi_wrap && ei >= a.epi_wrap) ei -= a.epi_wrap;
              const float4 o = *reinterpret_cast<const float4*>(a.epi_ref + ei);
              const float neg = a.epi_act == ACT_LRELU ? 0.2f : 0.f;
              if (a.epi_act == ACT_LRELU) {
                v.x *= o.x >= 0.f ? 1.f : neg; v.y *= o.y >= 0.f ? 1.f : neg; v.z *= o.z >= 0.f ? 1.f : neg; v.w *= o.w >= 0.f ? 1.f : neg;
              } else {
                v.x *= o.x > 0.f ? 1.f : neg; v.y *= o.y > 0.f ? 1.f : neg; v.z *= o.z > 0.f ? 1.f : neg; v.w *= o.w > 0.f ? 1.f : neg;
              }
            }
          }
          *reinterpret_cast<float4*>(yrow + n) = v;
        }
      }
  }
}

static int ilog2q(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

// tile geometry: 16-wide tiles of 128 output pixels = 8 rows of one image; false when the kernel does not take the shape
static bool x3q_geometry(const ConvArgs& a, QGeom* g) {
  const int OH = a.ph[0].OHp, OW = a.ph[0].OWp;
  for (int p = 0; p < a.nphase; ++p)
    if (a.ph[p].OHp != OH || a.ph[p].OWp != OW) return false;
  if (OW < 16 || OW % 16 != 0 || OH < 4) return false;
  const int TH = std::min(8, OH);
  if ((TH & (TH - 1)) != 0 || OH % TH != 0) return false;
  const int TI = 128 / (16 * TH);
  if (TI != 1) return false;
  for (int p = 0; p < a.nphase; ++p) {
    const ConvPhase& ph = a.ph[p];
    if (ph.Ty < 2 || ph.Ty > 3 || ph.Tx < 2 || ph.Tx > 3) return false;
    if (TI * (TH + ph.Ty - 1) * 20 > kHRQ) return false;
  }
  g->lTH = ilog2q(TH);
  g->TI = TI;
  g->tiles_x = OW / 16;
  g->tiles_y = OH / TH;
  return true;
}

static long long x3q_tiles(const ConvArgs& a, const QGeom& g) {
  return (long long)(a.NB / g.TI) * g.tiles_x * g.tiles_y * ((a.N + 127) / 128) * a.nphase;
}

bool conv_x3q_eligible(const ConvArgs& a) {
  static const bool on = dbg_env("EG_X3Q") != nullptr;   // (experiments build only; off unless asked for)
  if (!on || !a.wgt_bf || a.limbs != 3 || a.shuf || a.S != 1 || !(a.D == 1 || a.D == -1) || a.C % 32 != 0 || a.N < 128 || a.N % 4 != 0) return false;
  if (a.x_ps % 4 != 0 || a.y_ps % 4 != 0 || !a.x_bytes || !a.wbf_bytes || a.wbf_plane % 8 != 0) return false;
  if ((((uintptr_t)a.x | (uintptr_t)a.y | (uintptr_t)a.wgt_bf) & 15) != 0 || (a.bias && (((uintptr_t)a.bias) & 15) != 0)) return false;
  if (a.ksplit > 1 && a.ksplit > a.C / 16) return false;
  QGeom g;
  return x3q_geometry(a, &g);
}

// K split: the two resident blocks per CU want 512 blocks; slabs cost what they cost the plane-fed bf16 kernel
int conv_x3q_pick_ksplit(const ConvArgs& a) {
  QGeom g;
  if (!x3q_geometry(a, &g)) return 1;
  static const int target = dbg_env("EG_X3Q_KS_BLOCKS") ? atoi(dbg_env("EG_X3Q_KS_BLOCKS")) : 512;
  const long long tiles = x3q_tiles(a, g);
  const int chunks = a.C / 16;
  if (tiles >= target * 7 / 8 || chunks < 8) return 1;
  int ks = (int)((target + tiles - 1) / tiles);
  ks = std::min(ks, std::min(8, chunks / 4));
  return ks < 2 ? 1 : ks;
}

template <int PH>
static void x3q_launch_ph(const ConvArgs& a, const QGeom& g, dim3 grid, hipStream_t s) {
  if (a.ksplit > 1) hipLaunchKernelGGL((conv_x3q_kernel<true, PH>), grid, dim3(256), 0, s, a, g);
  else hipLaunchKernelGGL((conv_x3q_kernel<false, PH>), grid, dim3(256), 0, s, a, g);
}

const char* launch_conv_x3q(const ConvArgs& a, hipStream_t s) {
  QGeom g;
  x3q_geometry(a, &g);
  const int zs = a.ksplit > 1 ? a.ksplit : 1;
  dim3 grid((a.NB / g.TI) * g.tiles_x * g.tiles_y, (a.N + 127) / 128, a.nphase * zs);
  bool all33 = true, all22 = true;
  for (int p = 0; p < a.nphase; ++p) {
    all33 = all33 && a.ph[p].Ty == 3 && a.ph[p].Tx == 3;
    all22 = all22 && a.ph[p].Ty == 2 && a.ph[p].Tx == 2;
  }
  if (all33) x3q_launch_ph<0>(a, g, grid, s);
  else if (all22) x3q_launch_ph<1>(a, g, grid, s);
  else x3q_launch_ph<2>(a, g, grid, s);
  return a.ksplit > 1 ? "conv_gemm_x3q<128,128,splitk>" : "conv_gemm_x3q<128,128>";
}

}  // namespace eg
