// Limb-mode / bf16-mode convolution with the gathered operand staged ONCE per channel chunk as an LDS-resident halo tile
// (north_star: "im2col-free direct conv with LDS-staged input tiles").
//
//   conv_gemm_x3h_kernel : Y[pix_out][n] (+)= b[n] + sum_taps sum_c X[pix_out + d(tap)][c] * W[tap][c][n]
//                          stride-1 layers with 2 .. 9 taps (3x3 convolutions and their input gradients, the sub-pixel
//                          phases of the stride-2 transposed convolutions / stride-2 input gradients)
//
// The tap-gather kernels (k_gemm_x3.hip) fetch, split and store the A tile of every (tap, 32-channel chunk) K step: each
// activation element travels L2 -> registers -> limb split -> LDS once per TAP.  Measured there: with the gathered
// operand served from a cache-resident dummy the kernel runs 12-19 % faster, i.e. the tap re-reads through L2 cost that
// much, and the per-step split shares the SIMD with the MFMAs.  Here a block owns a 2-D tile of 256 output pixels
// (16 x 16, or 4 images of 8 x 8) x 128 output channels; per 32-channel chunk it loads the tile's input halo
// ((16 + 2)^2 = 324 pixels for 3x3) ONCE, splits it ONCE into its bf16 limb planes in LDS, and all taps of the chunk
// read their MFMA fragments from that image at a per-tap row offset: L2 traffic, split arithmetic and LDS stores drop by
// the tap count (9 -> 1.27 halo overhead).  Zero padding = out-of-range buffer offsets (zeros), REFLECT = mirrored source
// pixels at staging time; no per-tap masks.
//
// 512 threads = 8 waves (4 x 2: two per SIMD, so one wave's barrier / fragment-read bubbles are filled by its sibling's
// MFMAs), wave tile 64 x 64 (2 x 2 accumulators of v_mfma_f32_32x32x16_bf16): per K step a wave reads 24 fragments for 48
// MFMAs.  The filter tiles ([plane][tap][C/32][N][32], launch_split_bf16x3) come
// global -> LDS by buffer_load ... lds into a 3-slot ring, requested two K steps ahead; the NEXT chunk's halo is fetched
// into registers at the start of the current chunk (9 K steps to land).  One barrier per K step, two at a chunk start.
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

#ifndef EG_X3H_DBG
#define EG_X3H_DBG 0   // timing experiments (wrong results): 1 no per-step barrier, 2 no halo store, 4 no filter DMA, 8 no MFMA, 16 A reads from row 0
#endif
namespace {
constexpr int HRMAX = 400;   // halo rows (pixels) per block: 18 x 18 = 324, 4 x (10 x 10) = 400
constexpr int NA = 7;        // 16-byte staging loads per thread and chunk: 7 x 512 / 8 = 448 >= HRMAX rows x 8 quads

__device__ __forceinline__ unsigned pack2h(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_fh(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi_fh(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
// (x, y) -> packed limb pairs p0, p1, p2 with x = lo(p0) + lo(p1) + lo(p2) exactly (k_gemm_x3.hip: split_pair)
__device__ __forceinline__ void split_pair_h(float x, float y, unsigned& p0, unsigned& p1, unsigned& p2) {
  p0 = pack2h(x, y);
  const float rx = x - lo_fh(p0), ry = y - hi_fh(p0);
  p1 = pack2h(rx, ry);
  p2 = pack2h(rx - lo_fh(p1), ry - hi_fh(p1));
}
__device__ __forceinline__ int swz_h(int row) { return ((-((row >> 2) & 3)) & 3) ^ ((row >> 4) & 1); }
__device__ __forceinline__ int reflect_h(int i, int n) {
  i = i < 0 ? -i : i;
  return i >= n ? 2 * (n - 1) - i : i;
}
}  // namespace

struct HaloGeom {
  int lTW, lTH, TI;        // tile = TI images x 2^lTH rows x 2^lTW columns = 256 output pixels
  int tiles_x, tiles_y;    // tiles per image
  int pitch;               // LDS row pitch of a halo row: 20 for 16-wide tiles (conflict-free layout), 0 = dense rows
};

template <int NPROD, bool SPLITK, int BN, bool SW16>
__global__ __launch_bounds__(512, 1) void conv_gemm_x3h_kernel(const ConvArgs a, const HaloGeom g) {
  constexpr int NL = NPROD == 1 ? 1 : 3;
  constexpr int NSB = 3;
  constexpr int MI = BN / 64;                 // 32-row accumulator blocks per wave: wave tile 64 x 64 (BN = 128: waves 4 x 2)
                                              // or 32 x 64 (BN = 64, layers with 64 .. 127 output channels: waves 8 x 1)
  constexpr int WROWS = 32 * MI;
  constexpr int NBI = NL;                     // filter DMA instructions per wave and K step (one per plane)
  constexpr int APL = HRMAX * 64;             // bytes of one halo plane
  constexpr int BPL = BN * 64;                // bytes of one filter plane of a ring slot
  __shared__ __attribute__((aligned(16))) unsigned short Ah[NL][HRMAX * 32];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[NSB][NL][BN * 32];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7: two waves per SIMD
  const int wm = BN == 128 ? wave >> 1 : wave, wn = BN == 128 ? wave & 1 : 0;
  const bool issuer = wave < BN / 16;         // waves that carry a share of the filter DMA (16 rows per instruction)
  const int zphase = SPLITK ? (int)blockIdx.z / a.ksplit : (int)blockIdx.z;
  const int zsplit = SPLITK ? (int)blockIdx.z - zphase * a.ksplit : 0;
  const ConvPhase ph = a.ph[zphase];
  const int IH = a.IH, IW = a.IW, C = a.C, N = a.N, D = a.D;
  const int TW = 1 << g.lTW, TH = 1 << g.lTH;
  // tile position
  const int tpi = g.tiles_x * g.tiles_y;
  const int tn = (int)blockIdx.x / tpi, trem = (int)blockIdx.x - tn * tpi;
  const int tyy = trem / g.tiles_x, txx = trem - tyy * g.tiles_x;
  const int img0 = tn * g.TI, oy0 = tyy * TH, ox0 = txx * TW;
  const int n0 = blockIdx.y * BN;
  // halo geometry of this phase (stride 1): input pixel of output (oy, ox), tap (ty, tx) = (oy + ty D + offy, ox + tx D + offx)
  const int dminy = min(ph.offy, (ph.Ty - 1) * D + ph.offy), dminx = min(ph.offx, (ph.Tx - 1) * D + ph.offx);
  const int HH = TH + ph.Ty - 1, HW = TW + ph.Tx - 1, HU = g.TI * HH * HW;   // HU staged halo pixels
  // LDS image: pixel (ti, hy, hx) lives in row ti HPI + hy HWP + hx.  16-wide tiles: pitch 20 (a multiple of 4, so a row's
  // bank quarter is hx mod 4) with the 16-byte chunk swizzled by (hx >> 2) & 3 -- the four rows of a ds_read_b128 lane
  // group that share a bank quarter are hx = h, h + 12 (one tile row) and h + 4, h + 8 (the next one): four different
  // swizzles for every tap offset, i.e. conflict-free fragment reads.  Narrower tiles keep the dense rows / row swizzle.
  constexpr bool sw16 = SW16;
  const int HWP = sw16 ? 20 : HW, HPI = HH * HWP;

  // ---- staging: thread -> (halo row, channel quad) units u = tid + 512 j; global byte offsets are per block constants ---
  const int c4 = tid & 7;
  unsigned goff[NA];
  int srow[NA];    // LDS row of the unit (-1: none), bits 16..17: its chunk swizzle
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int u = (tid + 512 * j) >> 3;
    unsigned off = 0x80000000u;
    srow[j] = -1;
    if (u < HU) {
      const int ti = u / (HH * HW), rem = u - ti * (HH * HW);
      const int hy = rem / HW, hx = rem - hy * HW;
      int iy = oy0 + dminy + hy, ix = ox0 + dminx + hx;
      if (a.reflect) {
        iy = reflect_h(iy, IH);
        ix = reflect_h(ix, IW);
      }
      if ((unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW)
        off = (unsigned)(((((img0 + ti) * IH + iy) * IW + ix) * a.x_ps + c4 * 4) * 4);
      const int row = ti * HPI + hy * HWP + hx;
      srow[j] = row | ((sw16 ? (hx >> 2) & 3 : swz_h(row)) << 16);
    }
    goff[j] = off;
  }
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wres =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.wgt_bf), 0, a.wbf_bytes, 0x00020000);
  f32x4 ra[NA];
  auto load_halo = [&](int cc) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NA; ++j)
      ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, goff[j], cc * 128, 0));
  };
  auto store_halo = [&]() __attribute__((always_inline)) {
    if constexpr ((EG_X3H_DBG & 2) != 0) return;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      if (srow[j] >= 0) {
        const int o = (srow[j] & 0xffff) * 32 + (((c4 >> 1) ^ (srow[j] >> 16)) * 8) + (c4 & 1) * 4;
        if constexpr (NL == 1) {
          *reinterpret_cast<u32x2*>(&Ah[0][o]) = (u32x2){pack2h(ra[j][0], ra[j][1]), pack2h(ra[j][2], ra[j][3])};
        } else {
          unsigned p0a, p1a, p2a, p0b, p1b, p2b;
          split_pair_h(ra[j][0], ra[j][1], p0a, p1a, p2a);
          split_pair_h(ra[j][2], ra[j][3], p0b, p1b, p2b);
          *reinterpret_cast<u32x2*>(&Ah[0][o]) = (u32x2){p0a, p0b};
          *reinterpret_cast<u32x2*>(&Ah[NL > 1 ? 1 : 0][o]) = (u32x2){p1a, p1b};
          *reinterpret_cast<u32x2*>(&Ah[NL > 1 ? 2 : 0][o]) = (u32x2){p2a, p2b};
        }
      }
    }
  };

  // ---- filter ring: a wave's instruction fills rows wave*16 + lane/4 of a plane of the slot ----------------------------
  const int lsw = (((-((lane >> 4) & 3)) & 3) ^ (wave & 1));   // swz(wave*16 + lane/4)
  const unsigned woff = (unsigned)(((size_t)min(n0 + wave * 16 + (lane >> 2), N - 1) * 32 + ((lane & 3) ^ lsw) * 8) * 2);
  const int plane_bytes = (int)(a.wbf_plane * 2);
  const int cpt = C / 32, ntaps = ph.Ty * ph.Tx;
  const int wtile = N * 32 * 2;
  int cc_lo = 0, cc_hi = cpt;
  if constexpr (SPLITK) {
    const int per = (cpt + a.ksplit - 1) / a.ksplit;
    cc_lo = min(cpt, zsplit * per);
    cc_hi = min(cpt, cc_lo + per);
  }
  const int ncc = cc_hi - cc_lo, nt = ncc * ntaps;
  int rq_cc = cc_lo, rq_ty = 0, rq_tx = 0, rq_slot = 0;   // next filter tile to request
  auto request_b = [&]() __attribute__((always_inline)) {
    const int wsoff = (((ph.kh0 + rq_ty * a.kst) * a.KW + (ph.kw0 + rq_tx * a.kst)) * cpt + rq_cc) * wtile;
    if (issuer && (EG_X3H_DBG & 4) == 0) {
#pragma unroll
      for (int p = 0; p < NL; ++p)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wres, (__attribute__((address_space(3))) void*)&Bs[rq_slot][p][(wave * 16) * 32], 16, woff,
                                                 wsoff + p * plane_bytes, 0, 0);
    }
    rq_slot = rq_slot == NSB - 1 ? 0 : rq_slot + 1;
    rq_tx += 1;
    const bool wx = rq_tx == ph.Tx;
    rq_tx = wx ? 0 : rq_tx;
    rq_ty += wx ? 1 : 0;
    const bool wy = rq_ty == ph.Ty;
    rq_ty = wy ? 0 : rq_ty;
    rq_cc += wy ? 1 : 0;
  };

  // ---- fragments --------------------------------------------------------------------------------------------------------
  const int l31 = lane & 31, lh = lane >> 5;
  int base_row[MI], base_px[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int r = wm * WROWS + i * 32 + l31;
    const int ti = r >> (g.lTW + g.lTH), py = (r >> g.lTW) & (TH - 1), px = r & (TW - 1);
    base_row[i] = ti * HPI + py * HWP + px;
    base_px[i] = px;
  }
  const unsigned ah_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&Ah[0][0];
  const unsigned bs_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&Bs[0][0][0];
  const unsigned b_frag = ((wn * 64 + l31) * 32 + ((lh ^ swz_h(l31)) * 8)) * 2;

  f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Fragment reads as inline assembly: the compiler guards ordinary LDS loads against the filter DMA still in flight
  // ("may alias") with s_waitcnt vmcnt(0), which would wait for the requests issued a moment ago and undo the ring; the
  // ordering that matters is the explicit vmcnt wait + barrier of every K step.
#define EG_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define EG_WAIT4(n, f) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[1][0]), "+v"(f[1][1]))
#define EG_WAIT2(n, f) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(f[0][0]), "+v"(f[0][1]))
  // one limb product: 8 MFMAs of the wave's 2 x 2 accumulators over the K step's two 16-wide halves
  auto mm = [&](const bf16x8 (&af)[MI][2], const bf16x8 (&bf)[2][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr ((EG_X3H_DBG & 8) != 0) asm volatile("" : "+v"(acc[i][j]) : "v"(bf[j][kk]), "v"(af[i][kk]));
          else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j][kk], af[i][kk], acc[i][j], 0, 0, 0);
        }
  };
  auto read_a = [&](bf16x8 (&f)[MI][2], const unsigned (&aa)[MI], auto PL) __attribute__((always_inline)) {
    constexpr int off = decltype(PL)::value * APL;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      EG_DSR(f[i][0], aa[i], off);
      EG_DSR(f[i][1], aa[i] ^ 32u, off);
    }
  };
  auto read_b = [&](bf16x8 (&f)[2][2], unsigned bb, auto PL) __attribute__((always_inline)) {
    constexpr int off = decltype(PL)::value * BPL;
    EG_DSR(f[0][0], bb, off); EG_DSR(f[0][1], bb ^ 32u, off); EG_DSR(f[1][0], bb, off + 2048); EG_DSR(f[1][1], bb ^ 32u, off + 2048);
  };
  constexpr std::integral_constant<int, 0> P0{};
  constexpr std::integral_constant<int, 1> P1{};
  constexpr std::integral_constant<int, 2> P2{};
  // fragment addresses of a tap: halo row of the lane's output pixel + the tap's row offset, chunk swizzled by that row
  auto tap_addr = [&](int ty, int tx, unsigned (&aa)[MI]) __attribute__((always_inline)) {
    const int dxx = tx * D + ph.offx - dminx;
    const int toff = (ty * D + ph.offy - dminy) * HWP + dxx;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int row = (EG_X3H_DBG & 16) ? (lane & 31) : base_row[i] + toff;
      const int sw = sw16 ? ((base_px[i] + dxx) >> 2) & 3 : swz_h(row);
      aa[i] = ah_base + row * 64 + ((lh ^ sw) * 16);
    }
  };

  if (nt > 0) {
    load_halo(cc_lo);
    request_b();
    if (nt > 1) request_b();
  }
  int k = 0, slot = 0;
  for (int c = 0; c < ncc; ++c) {
    const bool has_next = c + 1 < ncc;
    // ---- chunk start: the previous chunk's fragment reads are done (barrier), its successor's halo is in registers --------
    asm volatile("s_barrier" ::: "memory");
    // everything older than the two filter requests in flight has landed (the halo loads were issued >= 2 K steps ago)
    // (a wave without a share of the filter DMA has only halo loads outstanding: it waits for all of them)
    if (!issuer) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (nt - k >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NBI) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBI) : "memory");
    store_halo();
    if (has_next) load_halo(cc_lo + c + 1);
    unsigned aa[MI];
    bf16x8 a0[MI][2];
    tap_addr(0, 0, aa);
    int ty = 0, tx = 0;
    for (int t = 0; t < ntaps; ++t, ++k) {
      // filter tile k has landed when only the requests issued after it are outstanding: tile k+1, and for the first two
      // taps of a chunk the next chunk's halo loads (issued at the chunk start, after the requests of tiles k, k+1)
      if (!issuer) {
      } else if (k + 1 >= nt) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (t < 2 && has_next) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBI + NA) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBI) : "memory");
      if constexpr ((EG_X3H_DBG & 1) != 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // halo stores + every wave's share of tile k visible
      const unsigned bb = bs_base + slot * (NL * BPL) + b_frag;
      bf16x8 b0[2][2];
      // the first plane of the gathered operand: read here for the first tap of a chunk (the halo image was just stored),
      // otherwise it was requested under the previous tap's last products (the image does not change within a chunk) and
      // has landed (the lgkmcnt(0) of the barrier above)
      if (t == 0) read_a(a0, aa, P0);
      read_b(b0, bb, P0);
      // next tap's position: its address arithmetic and the filter request issue under the first product's MFMAs
      tx += 1;
      const bool wx = tx == ph.Tx;
      tx = wx ? 0 : tx;
      ty += wx ? 1 : 0;
      if constexpr (NPROD == 1) {
        if constexpr (MI == 2) EG_WAIT4(0, a0);
        else EG_WAIT2(0, a0);
        EG_WAIT4(0, b0);
        mm(a0, b0);
        if (k + 2 < nt) request_b();
        tap_addr(ty, tx, aa);
        read_a(a0, aa, P0);
      } else {
        // LDS reads return in order: [a0 (2 MI)], b0 (4), b1 (4), b2 (4), then a1 (2 MI), then a2 (2 MI), then the next a0
        bf16x8 b1[2][2], b2[2][2], a1[MI][2], a2[MI][2];
        read_b(b1, bb, P1);
        read_b(b2, bb, P2);
        if constexpr (MI == 2) EG_WAIT4(8, a0);
        else EG_WAIT2(8, a0);
        EG_WAIT4(8, b0);
        read_a(a1, aa, P1);
        __builtin_amdgcn_sched_barrier(0);
        mm(a0, b0);
        if (k + 2 < nt) request_b();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MI == 2) EG_WAIT4(8, b1);
        else EG_WAIT4(6, b1);
        mm(a0, b1);
        if constexpr (MI == 2) EG_WAIT4(4, b2);
        else EG_WAIT4(2, b2);
        mm(a0, b2);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MI == 2) EG_WAIT4(0, a1);
        else EG_WAIT2(0, a1);
        read_a(a2, aa, P2);
        tap_addr(ty, tx, aa);   // (all reads through the current addresses are issued)
        read_a(a0, aa, P0);     // next tap's first plane (a0's last use was the third product); after the last tap of a
                                // chunk it fetches stale rows that the re-read at t == 0 replaces
        __builtin_amdgcn_sched_barrier(0);
        mm(a1, b0);
        mm(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MI == 2) EG_WAIT4(4, a2);
        else EG_WAIT2(2, a2);
        mm(a2, b0);
      }
      slot = slot == NSB - 1 ? 0 : slot + 1;
    }
  }
#undef EG_DSR
#undef EG_WAIT4
#undef EG_WAIT2

  // D^T layout of the 32x32 accumulator: lane l holds pixel row l & 31 and, in registers 4q .. 4q+3, the four consecutive
  // output channels 8q + 4 (l >> 5) .. +3 of its 32-channel block: one 16-byte store per register quad
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int r = wm * WROWS + i * 32 + l31;
    const int ti = r >> (g.lTW + g.lTH), py = (r >> g.lTW) & (TH - 1), px = r & (TW - 1);
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
          }
          *reinterpret_cast<float4*>(yrow + n) = v;
        }
      }
  }
}

static int ilog2h(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

static bool x3h_geometry(const ConvArgs& a, HaloGeom* g) {
  const int OH = a.ph[0].OHp, OW = a.ph[0].OWp;
  for (int p = 0; p < a.nphase; ++p)
    if (a.ph[p].OHp != OH || a.ph[p].OWp != OW) return false;
  if (OW < 4 || OH < 4 || (OW & (OW - 1)) != 0) return false;
  const int TW = std::min(16, OW);
  int TH = std::min(256 / TW, OH);
  if ((TH & (TH - 1)) != 0 || OH % TH != 0 || OW % TW != 0) return false;
  const int TI = 256 / (TW * TH);
  if (TI * TW * TH != 256 || a.NB % TI != 0) return false;
  static const bool dense = getenv("EG_X3H_DENSE") != nullptr;   // A/B switch: dense halo rows / row swizzle everywhere
  g->pitch = TW == 16 && !dense ? 20 : 0;
  for (int p = 0; p < a.nphase; ++p)
    if (TI * (TH + a.ph[p].Ty - 1) * 20 > HRMAX) g->pitch = 0;
  for (int p = 0; p < a.nphase; ++p) {
    const ConvPhase& ph = a.ph[p];
    const int nt = ph.Ty * ph.Tx;
    if (nt < 2 || ph.Ty > 3 || ph.Tx > 3) return false;
    if (TI * (TH + ph.Ty - 1) * (TW + ph.Tx - 1) > HRMAX) return false;
  }
  g->lTW = ilog2h(TW);
  g->lTH = ilog2h(TH);
  g->TI = TI;
  g->tiles_x = OW / TW;
  g->tiles_y = OH / TH;
  return true;
}

bool conv_x3h_eligible(const ConvArgs& a) {
  static const bool off = getenv("EG_NO_HALO") != nullptr;
  if (off || !a.wgt_bf || !(a.limbs == 3 || a.limbs == 1) || a.shuf) return false;
  if (a.S != 1 || !(a.D == 1 || a.D == -1) || a.C % 32 != 0 || a.N < 64 || a.N % 4 != 0) return false;
  if (a.x_ps % 4 != 0 || a.y_ps % 4 != 0 || !a.x_bytes || !a.wbf_bytes || a.wbf_plane % 8 != 0) return false;
  if ((((uintptr_t)a.x | (uintptr_t)a.y | (uintptr_t)a.wgt_bf) & 15) != 0 || (a.bias && (((uintptr_t)a.bias) & 15) != 0)) return false;
  if (a.ksplit > 1 && a.ksplit > a.C / 32) return false;
  HaloGeom g;
  return x3h_geometry(a, &g);
}

template <int BN>
static const char* x3h_launch(const ConvArgs& a, const HaloGeom& g, hipStream_t s) {
  const int zs = a.ksplit > 1 ? a.ksplit : 1;
  dim3 grid((a.NB / g.TI) * g.tiles_x * g.tiles_y, (a.N + BN - 1) / BN, a.nphase * zs);
  if (a.limbs == 1) {
    if (g.pitch) {
      if (a.ksplit > 1) hipLaunchKernelGGL((conv_gemm_x3h_kernel<1, true, BN, true>), grid, dim3(512), 0, s, a, g);
      else hipLaunchKernelGGL((conv_gemm_x3h_kernel<1, false, BN, true>), grid, dim3(512), 0, s, a, g);
    } else {
      if (a.ksplit > 1) hipLaunchKernelGGL((conv_gemm_x3h_kernel<1, true, BN, false>), grid, dim3(512), 0, s, a, g);
      else hipLaunchKernelGGL((conv_gemm_x3h_kernel<1, false, BN, false>), grid, dim3(512), 0, s, a, g);
    }
    return BN == 128 ? (a.ksplit > 1 ? "conv_gemm_bf16h<256,128,splitk>" : "conv_gemm_bf16h<256,128>")
                     : (a.ksplit > 1 ? "conv_gemm_bf16h<256,64,splitk>" : "conv_gemm_bf16h<256,64>");
  }
  if (g.pitch) {
    if (a.ksplit > 1) hipLaunchKernelGGL((conv_gemm_x3h_kernel<6, true, BN, true>), grid, dim3(512), 0, s, a, g);
    else hipLaunchKernelGGL((conv_gemm_x3h_kernel<6, false, BN, true>), grid, dim3(512), 0, s, a, g);
  } else {
    if (a.ksplit > 1) hipLaunchKernelGGL((conv_gemm_x3h_kernel<6, true, BN, false>), grid, dim3(512), 0, s, a, g);
    else hipLaunchKernelGGL((conv_gemm_x3h_kernel<6, false, BN, false>), grid, dim3(512), 0, s, a, g);
  }
  return BN == 128 ? (a.ksplit > 1 ? "conv_gemm_x3h<256,128,splitk>" : "conv_gemm_x3h<256,128>")
                   : (a.ksplit > 1 ? "conv_gemm_x3h<256,64,splitk>" : "conv_gemm_x3h<256,64>");
}

const char* launch_conv_gemm_x3h(const ConvArgs& a, hipStream_t s) {
  HaloGeom g;
  x3h_geometry(a, &g);
  return a.N >= 128 ? x3h_launch<128>(a, g, s) : x3h_launch<64>(a, g, s);
}

}  // namespace eg
