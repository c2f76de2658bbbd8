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
#define EG_X3H_DBG 0   // timing experiments (wrong results): 1 no per-step barrier, 2 no halo store, 4 no filter DMA, 8 no MFMA, 16 A reads from row 0, 32 no fragment reads, 128 no output stores
#endif
namespace {
typedef __bf16 bf16x8_ __attribute__((ext_vector_type(8)));
typedef float f32x16_ __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void mfma_acc(f32x16_& c, const bf16x8_& x, const bf16x8_& y) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0);
}
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
  int dispatch_order;      // experiments: 1 = blockIdx as dispatched (no XCD-aware block map)
  int pair;                // 1: the PAIR form runs the launch (kernel comment)
};

// S2D: a 4x4 stride-2 SAME convolution (the discriminators' forward layers) as a 2x2-tap STRIDE-1 convolution over the
// space-to-depth view V[Y][X][(py, px, c)] = x[2Y + py - 1][2X + px - 1][c] of its input: y[oy][ox] = sum over dY, dX in
// {0, 1}, (py, px, c) of V[oy + dY][ox + dX][(py, px, c)] w[2 dY + py][2 dX + px][c].  V is never materialised: a K chunk
// of 32 channels lies inside ONE phase (py, px), so staging chunk (q = 2 py + px, cq) reads the halo pixels' phase-(py, px)
// source pixels straight from x (per-row base offset + a uniform phase offset, per-row validity of the 4 phases), and
// the chunk's filter tiles are those of taps (2 dY + py, 2 dX + px).  The halo image is shared by 4 taps (9 at 3x3).
// PAIR (64-output-channel layers with phases in pairs that differ in the column phase only -- the 4x4 stride-2 input gradients
// towards a 64-channel tensor): the two 64-column halves of the 128-wide block tile are the two PHASES (ph[2z], ph[2z + 1]) of one
// pixel tile instead of two channel blocks.  Both phases read the same staged halo (one column wider), wave column wn takes the
// filter rows, tap offsets and output pixels of phase wn: every wave keeps the 64 x 64 tile of the 128-channel form where the
// 64-channel form (BN = 64) runs 32 x 64 tiles at 1.5 fragment reads per MFMA.
template <int NPROD, bool SPLITK, int BN, bool SW16, bool S2D = false, bool PAIR = false>
__global__ __launch_bounds__(512, 1) void conv_gemm_x3h_kernel(const ConvArgs a, const HaloGeom g) {
  static_assert(!PAIR || (BN == 128 && !S2D), "PAIR is a form of the 128-wide stride-1 tile");
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
  // XCD-aware block map: the output tiles (blockIdx.y) of one pixel tile stage the same halo; the dispatcher places block b
  // on XCD b % 8, so the remap makes them neighbours in ONE XCD's contiguous range (halo fetched into one L2, not gridDim.y)
  const unsigned gxy = gridDim.x * gridDim.y;
  const unsigned vblk = xcd_remap((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, gxy * gridDim.z, g.dispatch_order);
  const int bz = __builtin_amdgcn_readfirstlane(vblk / gxy), bxy = __builtin_amdgcn_readfirstlane(vblk % gxy);
  const int bx = bxy / (int)gridDim.y, by = bxy - bx * (int)gridDim.y;
  const int zphase = SPLITK ? bz / a.ksplit : bz;
  const int zsplit = SPLITK ? bz - zphase * a.ksplit : 0;
  // (PAIR: zphase counts phase pairs; ph = the phase of this wave's accumulators, dkh0 / dkw0 = first tap of the phase whose filter rows its DMA brings)
  const ConvPhase ph = a.ph[PAIR ? 2 * zphase + wn : zphase];
  const int dkh0 = PAIR ? a.ph[2 * zphase + (wave >> 2)].kh0 : ph.kh0, dkw0 = PAIR ? a.ph[2 * zphase + (wave >> 2)].kw0 : ph.kw0;
  const int IH = a.IH, IW = a.IW, C = a.C, N = a.N, D = S2D ? 1 : a.D;
  const int nTy = S2D ? 2 : ph.Ty, nTx = S2D ? 2 : ph.Tx;   // taps the K loop walks
  const int TW = 1 << g.lTW, TH = 1 << g.lTH;
  // tile position
  const int tpi = g.tiles_x * g.tiles_y;
  const int tn = bx / tpi, trem = bx - tn * tpi;
  const int tyy = trem / g.tiles_x, txx = trem - tyy * g.tiles_x;
  const int img0 = tn * g.TI, oy0 = tyy * TH, ox0 = txx * TW;
  const int n0 = PAIR ? 0 : by * BN;
  // halo geometry of this phase (stride 1): input pixel of output (oy, ox), tap (ty, tx) = (oy + ty D + offy, ox + tx D + offx)
  const int offy = S2D ? 0 : ph.offy, offx = S2D ? 0 : ph.offx;   // (S2D: ph.off* is the source offset of phase 0, below)
  const int dminy = min(offy, (nTy - 1) * D + offy);
  int dminx = min(offx, (nTx - 1) * D + offx), xspan = 0;
  if constexpr (PAIR) {   // the halo covers both phases' taps (their column offsets differ by one)
    const int ox0p = a.ph[2 * zphase].offx, ox1p = a.ph[2 * zphase + 1].offx;
    dminx = min(min(ox0p, (nTx - 1) * D + ox0p), min(ox1p, (nTx - 1) * D + ox1p));
    xspan = ox0p > ox1p ? ox0p - ox1p : ox1p - ox0p;
  }
  const int HH = TH + nTy - 1, HW = TW + nTx - 1 + xspan, HU = g.TI * HH * HW;   // HU staged halo pixels
  // LDS image: pixel (ti, hy, hx) lives in row ti HPI + hy HWP + hx.  16-wide tiles: pitch 20 (a multiple of 4, so a row's
  // bank quarter is hx mod 4) with the 16-byte chunk swizzled by (hx >> 2) & 3 -- the four rows of a ds_read_b128 lane
  // group that share a bank quarter are hx = h, h + 12 (one tile row) and h + 4, h + 8 (the next one): four different
  // swizzles for every tap offset, i.e. conflict-free fragment reads.  Narrower tiles keep the dense rows / row swizzle.
  constexpr bool sw16 = SW16;
  const int HWP = sw16 ? 20 : HW, HPI = HH * HWP;

  // ---- staging: thread -> (halo row, channel quad) units u = tid + 512 j; global byte offsets are per block constants ---
  const int c4 = tid & 7;
  unsigned goff[NA];
  int srow[NA];    // LDS row of the unit (-1: none), bits 16..17: its chunk swizzle; S2D: bits 20..23 = source row / column of phase py = 0, 1 / px = 0, 1 inside the image
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int u = (tid + 512 * j) >> 3;
    unsigned off = 0x80000000u;
    srow[j] = -1;
    if (u < HU) {
      const int ti = u / (HH * HW), rem = u - ti * (HH * HW);
      const int hy = rem / HW, hx = rem - hy * HW;
      int iy = oy0 + dminy + hy, ix = ox0 + dminx + hx;
      int vbits = 0;
      if constexpr (S2D) {   // source pixel of phase (0, 0); the offset may be "negative" (wraps; only used where valid)
        iy = 2 * iy + ph.offy;
        ix = 2 * ix + ph.offx;
        off = (unsigned)(((((img0 + ti) * IH + iy) * IW + ix) * a.x_ps + c4 * 4) * 4);
        vbits = ((unsigned)iy < (unsigned)IH ? 1 : 0) | ((unsigned)(iy + 1) < (unsigned)IH ? 2 : 0) |
                ((unsigned)ix < (unsigned)IW ? 4 : 0) | ((unsigned)(ix + 1) < (unsigned)IW ? 8 : 0);
      } else {
        if (a.reflect) {
          iy = reflect_h(iy, IH);
          ix = reflect_h(ix, IW);
        }
        if ((unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW) {
          if (a.x_up)   // (the source is the half-resolution tensor: four output-side pixels read one of its pixels)
            off = (unsigned)(((((img0 + ti) * (IH >> 1) + (iy >> 1)) * (IW >> 1) + (ix >> 1)) * a.x_ps + c4 * 4) * 4);
          else
            off = (unsigned)(((((img0 + ti) * IH + iy) * IW + ix) * a.x_ps + c4 * 4) * 4);
        }
      }
      const int row = ti * HPI + hy * HWP + hx;
      srow[j] = row | ((sw16 ? (hx >> 2) & 3 : swz_h(row)) << 16) | (vbits << 20);
    }
    goff[j] = off;
  }
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wres =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.wgt_bf), 0, a.wbf_bytes, 0x00020000);
  f32x4 ra[NA];
  const int cpt = C / 32;
  auto load_halo = [&](int cc) __attribute__((always_inline)) {
    if constexpr (S2D) {
      const int q = cc / cpt, cq = cc - q * cpt, py = q >> 1, px = q & 1;   // (uniform: once per chunk = per 4 K steps)
      const unsigned soff = (unsigned)((((py * IW + px) * a.x_ps) + cq * 32) * 4);
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        const bool ok = ((srow[j] >> (20 + py)) & (srow[j] >> (22 + px)) & 1) != 0 && srow[j] >= 0;
        ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, ok ? goff[j] + soff : 0x80000000u, 0, 0));
      }
    } else {
#pragma unroll
      for (int j = 0; j < NA; ++j)
        ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, goff[j], cc * 128, 0));
    }
  };
  auto store_halo = [&]() __attribute__((always_inline)) {
    if constexpr ((EG_X3H_DBG & 2) != 0) return;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      if (srow[j] >= 0) {
        const int o = (srow[j] & 0xffff) * 32 + (((c4 >> 1) ^ ((srow[j] >> 16) & 3)) * 8) + (c4 & 1) * 4;
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
  const unsigned woff =
      (unsigned)(((size_t)min(n0 + (PAIR ? wave & 3 : wave) * 16 + (lane >> 2), N - 1) * 32 + ((lane & 3) ^ lsw) * 8) * 2);
  const int plane_bytes = (int)(a.wbf_plane * 2);
  const int ntaps = nTy * nTx;
  const int vcpt = S2D ? 4 * cpt : cpt;   // K chunks the loop walks (S2D: 4 phases x C / 32)
  const int wtile = N * 32 * 2;
  int cc_lo = 0, cc_hi = vcpt;
  if constexpr (SPLITK) {
    const int per = (vcpt + a.ksplit - 1) / a.ksplit;
    cc_lo = min(vcpt, zsplit * per);
    cc_hi = min(vcpt, cc_lo + per);
  }
  const int ncc = cc_hi - cc_lo, nt = ncc * ntaps;
  int rq_cc = cc_lo, rq_ty = 0, rq_tx = 0, rq_slot = 0;   // next filter tile to request
  auto request_b = [&]() __attribute__((always_inline)) {
    int wsoff;
    if constexpr (S2D) {   // K chunk rq_cc = phase q, channel chunk cq (uniform arithmetic, once per K step)
      const int q = rq_cc / cpt, cq = rq_cc - q * cpt;
      wsoff = (((2 * rq_ty + (q >> 1)) * a.KW + (2 * rq_tx + (q & 1))) * cpt + cq) * wtile;
    } else {
      wsoff = (((dkh0 + rq_ty * a.kst) * a.KW + (dkw0 + rq_tx * a.kst)) * cpt + rq_cc) * wtile;
    }
    if (issuer && (EG_X3H_DBG & 4) == 0) {
#pragma unroll
      for (int p = 0; p < NL; ++p)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wres, (__attribute__((address_space(3))) void*)&Bs[rq_slot][p][(wave * 16) * 32], 16, woff,
                                                 wsoff + p * plane_bytes, 0, 0);
    }
    rq_slot = rq_slot == NSB - 1 ? 0 : rq_slot + 1;
    rq_tx += 1;
    const bool wx = rq_tx == nTx;
    rq_tx = wx ? 0 : rq_tx;
    rq_ty += wx ? 1 : 0;
    const bool wy = rq_ty == nTy;
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
#if (EG_X3H_DBG & 32)   // no fragment reads: the registers keep whatever they hold ("+v": no new value, ordering kept)
#define EG_DSR(dst, addr, off) asm volatile("" : "+v"(dst) : "v"(addr), "n"(off))
#elif (EG_X3H_DBG & 64)   // half of the fragment reads (every instruction reading the odd 16-byte chunk) dropped
#define EG_DSR(dst, addr, off) do { if (__builtin_constant_p(off) && ((off) % 4096) >= 2048) asm volatile("" : "+v"(dst) : "v"(addr), "n"(off)); else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off)); } while (0)
#else
#define EG_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#endif
#ifndef EG_X3H_BURST
#define EG_X3H_BURST 0
#endif
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
          else mfma_acc(acc[i][j], bf[j][kk], af[i][kk]);
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
  [[maybe_unused]] constexpr std::integral_constant<int, 1> P1{};
  [[maybe_unused]] constexpr std::integral_constant<int, 2> P2{};
  // fragment addresses of a tap: halo row of the lane's output pixel + the tap's row offset, chunk swizzled by that row
  auto tap_addr = [&](int ty, int tx, unsigned (&aa)[MI]) __attribute__((always_inline)) {
    const int dxx = tx * D + offx - dminx;
    const int toff = (ty * D + offy - dminy) * HWP + dxx;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int row = (EG_X3H_DBG & 16) ? (lane & 31) : base_row[i] + toff;
      const int sw = sw16 ? ((base_px[i] + dxx) >> 2) & 3 : swz_h(row);
      aa[i] = ah_base + row * 64 + ((lh ^ sw) * 16);
    }
  };

  // K step k = (chunk c, tap t).  Filter tile k+2 is requested in step k; tile k+1 has landed and is visible to every wave
  // at the barrier of step k (each wave waits for its share before it), so step k also fetches the first filter plane of
  // step k+1, and -- the halo image being constant within a chunk -- the first halo plane of the next tap: a step
  // opens with its first product's operands already in registers, and the other five products' planes return under it.
  // Product order a0b0, a1b0, a2b0 | (b0 free: next b0) | a0b1, a0b2 | (a0 free: next a0) | a1b1.
  // Every fragment register an inline-assembly read targets is an in/out operand of the wait that retires the read: the
  // compiler believes an asm read's result exists when the statement ends, so a destination that is DEAD in the program
  // (the look-ahead a0 read after a chunk's last tap, whose value the re-read at t == 0 replaces) is free for its register
  // allocator at once -- round 3's one-plane S2D split-K instantiations kept the NEXT b0 fragment address in such a
  // register and the stale data landed on it (tools/isa_async_lint.py finds the pattern in the ISA).
  // Listing the registers here keeps them allocated until the data has landed.
#define EG_TOP_ASM(text, a, b)                                                                                                  \
  do {                                                                                                                          \
    if constexpr (MI == 2)                                                                                                      \
      asm volatile(text : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[MI - 1][0]), "+v"(a[MI - 1][1]), "+v"(b[0][0]), "+v"(b[0][1]),   \
                   "+v"(b[1][0]), "+v"(b[1][1])::"memory");                                                                     \
    else                                                                                                                        \
      asm volatile(text : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1])::"memory");  \
  } while (0)
#define EG_TOP(a, b) EG_TOP_ASM("s_waitcnt lgkmcnt(0)\n\ts_barrier", a, b)
#define EG_TOP_NB(a, b) EG_TOP_ASM("s_waitcnt lgkmcnt(0)", a, b)
  bf16x8 a0[MI][2] = {}, b0[2][2] = {};   // (defined values: both are in/out operands of the first wait)
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);   // the later-dispatched half loses every issue arbitration otherwise (+0.5 %)
  if (nt > 0) {
    load_halo(cc_lo);
    // (S2D: the vmcnt waits below count on the issue ORDER halo loads, tile 0, tile 1; the phase arithmetic in front of
    //  the S2D loads gives the scheduler room to hoist the address-free DMA requests above them -- pin the order)
    if constexpr (S2D) __builtin_amdgcn_sched_barrier(0);
    request_b();
    if (nt > 1) request_b();
    if constexpr (S2D) __builtin_amdgcn_sched_barrier(0);
    // filter tile 0 (outstanding, oldest first: halo, tile 0, tile 1)
    if (!issuer) {
    } else if (nt > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    read_b(b0, bs_base + b_frag, P0);
  }
  int k = 0, slot = 0;
  for (int c = 0; c < ncc; ++c) {
    const bool has_next = c + 1 < ncc;
    // ---- chunk start: the previous chunk's fragment reads are done (barrier), its successor's halo is in registers --------
    // (a0 holds the stale look-ahead read of the previous chunk's last tap: in flight until this wait, see EG_TOP_ASM)
    EG_TOP(a0, b0);
    // the halo loads are older than the newest filter request (tile k+1, issued in the previous chunk's last step)
    if (!issuer || k + 1 >= nt) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBI) : "memory");
    store_halo();
    if constexpr (S2D) __builtin_amdgcn_sched_barrier(0);
    if (has_next) load_halo(cc_lo + c + 1);
    if constexpr (S2D) __builtin_amdgcn_sched_barrier(0);
    unsigned aa[MI];
    tap_addr(0, 0, aa);
    int ty = 0, tx = 0;
    for (int t = 0; t < ntaps; ++t, ++k) {
      // tile k+1 has landed: it is the newest request except, in a chunk's first step, for the next chunk's halo loads
      if (!issuer) {
      } else if (t == 0 && has_next) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // halo stores + every wave's share of tile k+1 visible, all reads of tile k-1 done; a0 / b0 have landed
      if constexpr ((EG_X3H_DBG & 1) != 0) EG_TOP_NB(a0, b0);
      else EG_TOP(a0, b0);
      const unsigned bb = bs_base + slot * (NL * BPL) + b_frag;
      const int nslot = slot == NSB - 1 ? 0 : slot + 1;
      const unsigned bbn = bs_base + nslot * (NL * BPL) + b_frag;
      if (t == 0) {   // first tap of a chunk: the halo image was just stored
        read_a(a0, aa, P0);
        if constexpr (MI == 2) EG_WAIT4(0, a0);
        else EG_WAIT2(0, a0);
      }
      // next tap's position
      tx += 1;
      const bool wx = tx == nTx;
      tx = wx ? 0 : tx;
      ty += wx ? 1 : 0;
      if constexpr (NPROD == 1) {
        mm(a0, b0);
        if (k + 2 < nt) request_b();
        __builtin_amdgcn_sched_barrier(0);
        tap_addr(ty, tx, aa);
        read_a(a0, aa, P0);      // (after a chunk's last tap: stale rows, replaced by the re-read at t == 0)
        read_b(b0, bbn, P0);
      } else {
        bf16x8 b1[2][2], b2[2][2], a1[MI][2], a2[MI][2];
#if EG_X3H_BURST   // round 2-3 schedule, kept for the A/B: the step's 12 / 16 reads dealt out between the FIRST product's MFMAs
        // LDS reads return in order: a1 (2 MI), a2 (2 MI), b1 (4), b2 (4), then the next b0 (4), then the next a0 (2 MI)
        {
          constexpr int NM = 4 * MI;   // MFMAs of a product
#pragma unroll
          for (int n = 0; n < NM; ++n) {
            const int kk = n / (2 * MI), i = (n / 2) % MI, j = n % 2;
            mfma_acc(acc[i][j], b0[j][kk], a0[i][kk]);
            if constexpr (MI == 2) {
              if (n == 0) { EG_DSR(a1[0][0], aa[0], 1 * APL); EG_DSR(a1[0][1], aa[0] ^ 32u, 1 * APL); }
              if (n == 1) { EG_DSR(a1[1][0], aa[1], 1 * APL); EG_DSR(a1[1][1], aa[1] ^ 32u, 1 * APL); }
              if (n == 2) { EG_DSR(a2[0][0], aa[0], 2 * APL); EG_DSR(a2[0][1], aa[0] ^ 32u, 2 * APL); }
              if (n == 3) { EG_DSR(a2[1][0], aa[1], 2 * APL); EG_DSR(a2[1][1], aa[1] ^ 32u, 2 * APL); }
              if (n == 4) { EG_DSR(b1[0][0], bb, 1 * BPL); EG_DSR(b1[0][1], bb ^ 32u, 1 * BPL); }
              if (n == 5) { EG_DSR(b1[1][0], bb, 1 * BPL + 2048); EG_DSR(b1[1][1], bb ^ 32u, 1 * BPL + 2048); }
              if (n == 6) { EG_DSR(b2[0][0], bb, 2 * BPL); EG_DSR(b2[0][1], bb ^ 32u, 2 * BPL); }
              if (n == 7) { EG_DSR(b2[1][0], bb, 2 * BPL + 2048); EG_DSR(b2[1][1], bb ^ 32u, 2 * BPL + 2048); }
            } else {
              if (n == 0) { EG_DSR(a1[0][0], aa[0], 1 * APL); EG_DSR(a1[0][1], aa[0] ^ 32u, 1 * APL); EG_DSR(a2[0][0], aa[0], 2 * APL); EG_DSR(a2[0][1], aa[0] ^ 32u, 2 * APL); }
              if (n == 1) { EG_DSR(b1[0][0], bb, 1 * BPL); EG_DSR(b1[0][1], bb ^ 32u, 1 * BPL); EG_DSR(b1[1][0], bb, 1 * BPL + 2048); EG_DSR(b1[1][1], bb ^ 32u, 1 * BPL + 2048); }
              if (n == 2) { EG_DSR(b2[0][0], bb, 2 * BPL); EG_DSR(b2[0][1], bb ^ 32u, 2 * BPL); }
              if (n == 3) { EG_DSR(b2[1][0], bb, 2 * BPL + 2048); EG_DSR(b2[1][1], bb ^ 32u, 2 * BPL + 2048); }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (k + 2 < nt) request_b();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MI == 2) EG_WAIT4(12, a1);
        else EG_WAIT2(10, a1);
        mm(a1, b0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MI == 2) EG_WAIT4(8, a2);
        else EG_WAIT2(8, a2);
        mm(a2, b0);
        __builtin_amdgcn_sched_barrier(0);
        read_b(b0, bbn, P0);     // tile k+1 (after the last step: a stale slot, never used)
        __builtin_amdgcn_sched_barrier(0);
        EG_WAIT4(8, b1);
        mm(a0, b1);
        __builtin_amdgcn_sched_barrier(0);
        EG_WAIT4(4, b2);
        mm(a0, b2);
        __builtin_amdgcn_sched_barrier(0);
        tap_addr(ty, tx, aa);
        read_a(a0, aa, P0);      // (after a chunk's last tap: stale rows, replaced by the re-read at t == 0)
        __builtin_amdgcn_sched_barrier(0);
        mm(a1, b1);
#else
        // Round 4: ONE fragment group per product, requested under that product's first two MFMAs and consumed by the NEXT
        // product -- a1 | a2 | b1 | b2 | next b0 | next a0 under a0b0 | a1b0 | a2b0 | a0b1 | a0b2 | a1b1.  The earlier schedule
        // put the step's 16 reads under the first product: eight waves then ask the LDS for 128 KB in the ~512 cycles that
        // product lasts on a SIMD, twice what it delivers, and an in-order wave cannot issue the MFMA behind a read the full
        // queue refuses -- the matrix pipe starved for ~500 of a step's ~3,100 cycles ("fragment reads 20 %").
        // Spread, the LDS runs at an even ~50 %, and at most five fragment groups are live instead of six.
        // `R0` / `R1`: the reads issued behind the product's first / second MFMA.
#define EG_PROD(af, bf, R0, R1)                                                   \
  do {                                                                            \
    constexpr int NM_ = 4 * MI;                                                   \
    _Pragma("unroll") for (int n_ = 0; n_ < NM_; ++n_) {                          \
      const int kk_ = n_ / (2 * MI), i_ = (n_ / 2) % MI, j_ = n_ % 2;             \
      mfma_acc(acc[i_][j_], bf[j_][kk_], af[i_][kk_]);                            \
      if (n_ == 0) { R0 }                                                         \
      if (n_ == 1) { R1 }                                                         \
      __builtin_amdgcn_sched_barrier(0);                                          \
    }                                                                             \
  } while (0)
        const unsigned aa0 = aa[0], aa1 = aa[MI - 1];
        // a0 b0, a1 arriving
        EG_PROD(a0, b0, EG_DSR(a1[0][0], aa0, 1 * APL); EG_DSR(a1[0][1], aa0 ^ 32u, 1 * APL);,
                if constexpr (MI == 2) { EG_DSR(a1[MI - 1][0], aa1, 1 * APL); EG_DSR(a1[MI - 1][1], aa1 ^ 32u, 1 * APL); });
        if (k + 2 < nt) request_b();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MI == 2) EG_WAIT4(0, a1);
        else EG_WAIT2(0, a1);
        // a1 b0, a2 arriving
        EG_PROD(a1, b0, EG_DSR(a2[0][0], aa0, 2 * APL); EG_DSR(a2[0][1], aa0 ^ 32u, 2 * APL);,
                if constexpr (MI == 2) { EG_DSR(a2[MI - 1][0], aa1, 2 * APL); EG_DSR(a2[MI - 1][1], aa1 ^ 32u, 2 * APL); });
        if constexpr (MI == 2) EG_WAIT4(0, a2);
        else EG_WAIT2(0, a2);
        // a2 b0, b1 arriving
        EG_PROD(a2, b0, EG_DSR(b1[0][0], bb, 1 * BPL); EG_DSR(b1[0][1], bb ^ 32u, 1 * BPL);,
                EG_DSR(b1[1][0], bb, 1 * BPL + 2048); EG_DSR(b1[1][1], bb ^ 32u, 1 * BPL + 2048););
        EG_WAIT4(0, b1);
        // a0 b1, b2 arriving
        EG_PROD(a0, b1, EG_DSR(b2[0][0], bb, 2 * BPL); EG_DSR(b2[0][1], bb ^ 32u, 2 * BPL);,
                EG_DSR(b2[1][0], bb, 2 * BPL + 2048); EG_DSR(b2[1][1], bb ^ 32u, 2 * BPL + 2048););
        EG_WAIT4(0, b2);
        // a0 b2, the next step's b0 arriving (tile k+1; after the last step: a stale slot, never used)
        EG_PROD(a0, b2, EG_DSR(b0[0][0], bbn, 0); EG_DSR(b0[0][1], bbn ^ 32u, 0);,
                EG_DSR(b0[1][0], bbn, 2048); EG_DSR(b0[1][1], bbn ^ 32u, 2048););
        // a1 b1, the next tap's a0 arriving (after a chunk's last tap: stale rows, replaced by the re-read at t == 0)
        tap_addr(ty, tx, aa);
        const unsigned an0 = aa[0], an1 = aa[MI - 1];
        EG_PROD(a1, b1, EG_DSR(a0[0][0], an0, 0); EG_DSR(a0[0][1], an0 ^ 32u, 0);,
                if constexpr (MI == 2) { EG_DSR(a0[MI - 1][0], an1, 0); EG_DSR(a0[MI - 1][1], an1 ^ 32u, 0); });
#undef EG_PROD
#endif
      }
      slot = nslot;
    }
  }
  // the last step's look-ahead reads still target a0 / b0: they land before anything else may use those registers
  EG_TOP_NB(a0, b0);
  __builtin_amdgcn_sched_barrier(0);
#undef EG_TOP_ASM
#undef EG_TOP
#undef EG_TOP_NB
#undef EG_DSR
#undef EG_WAIT4
#undef EG_WAIT2

  // D^T layout of the 32x32 accumulator: lane l holds pixel row l & 31 and, in registers 4q .. 4q+3, the four consecutive
  // output channels 8q + 4 (l >> 5) .. +3 of its 32-channel block: one 16-byte store per register quad
  if constexpr (!SPLITK) {
    if (a.pool_y) {   // pool2_add in the epilogue (kernels.h: ConvArgs::pool_y): the full-resolution output is never stored
      const float plk = a.pool_leak ? *a.pool_leak : 0.f;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int r = wm * WROWS + i * 32 + l31;
        const int ti = r >> (g.lTW + g.lTH), py = (r >> g.lTW) & (TH - 1), px = r & (TW - 1);
        const int nb = img0 + ti, oy = oy0 + py, ox = ox0 + px;
        const bool owner = ((py | px) & 1) == 0;
        const size_t prow = ((size_t)(nb * (a.OHf >> 1) + (oy >> 1)) * (a.OWf >> 1) + (ox >> 1)) * a.y_ps;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int n = n0 + (PAIR ? 0 : wn * 64) + j * 32 + q * 8 + lh * 4;
            const float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
            conv_epi_pool(a, v, TW, owner && n < N, prow + n, n, plk);
          }
      }
      return;
    }
  }
  const float elk = (!SPLITK && a.epi_act == ACT_PRELU_X) ? *a.epi_leak : 0.f;
  const float y2lk = (!SPLITK && a.y2_leak) ? *a.y2_leak : 0.f;   // second output: PReLU of the stored value (kernels.h: conv_epi_y2)
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
        const int n = n0 + (PAIR ? 0 : wn * 64) + j * 32 + q * 8 + lh * 4;
        if (n < N) {
          float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
          if (a.x_up) { v.x *= 0.25f; v.y *= 0.25f; v.z *= 0.25f; v.w *= 0.25f; }   // (exact: the pool's 1/4 applied to the sum)
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
              float4 o;
              if (a.epi_ref_bf) {   // (the output as its plane: exact widening, the sign is what the ReLU / LeakyReLU forms read)
                const uint2 q = *reinterpret_cast<const uint2*>(a.epi_ref_bf + ei);
                o = make_float4(__builtin_bit_cast(float, q.x << 16), __builtin_bit_cast(float, q.x & 0xffff0000u),
                                __builtin_bit_cast(float, q.y << 16), __builtin_bit_cast(float, q.y & 0xffff0000u));
              } else {
                o = *reinterpret_cast<const float4*>(a.epi_ref + ei);
              }
              const float neg = a.epi_act == ACT_LRELU ? 0.2f : 0.f;
              if (a.epi_act == ACT_PRELU_X) {   // (o = the PReLU's input; prelu_bwd's expression, k_elem.hip)
                v.x = elk * o.x >= o.x ? elk * v.x : v.x; v.y = elk * o.y >= o.y ? elk * v.y : v.y;
                v.z = elk * o.z >= o.z ? elk * v.z : v.z; v.w = elk * o.w >= o.w ? elk * v.w : v.w;
              } else if (a.epi_act == ACT_LRELU) {
                v.x *= o.x >= 0.f ? 1.f : neg; v.y *= o.y >= 0.f ? 1.f : neg; v.z *= o.z >= 0.f ? 1.f : neg; v.w *= o.w >= 0.f ? 1.f : neg;
              } else {
                v.x *= o.x > 0.f ? 1.f : neg; v.y *= o.y > 0.f ? 1.f : neg; v.z *= o.z > 0.f ? 1.f : neg; v.w *= o.w > 0.f ? 1.f : neg;
              }
            }
          }
          if (((EG_X3H_DBG & 128) == 0 || v.x == 123.456f) && (SPLITK || !a.epi_skip_f32))
            *reinterpret_cast<float4*>(yrow + n) = v;   // (bit 7: timing without the output stores)
          if constexpr (!SPLITK) {
            if (a.y2_leak) conv_epi_y2(a, (size_t)(yrow - a.y) + n, v, y2lk);
            if (a.epi_bf) conv_epi_plane(a.epi_bf + (size_t)(yrow - a.y) + n, v);
          }
        }
      }
  }
}

static int ilog2h(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

static bool x3h_s2d_shape(const ConvArgs& a) {   // 4x4 stride-2 SAME forward convolution (one phase, taps as stored)
  const ConvPhase& p = a.ph[0];
  return a.S == 2 && a.nphase == 1 && a.KW == 4 && p.Ty == 4 && p.Tx == 4 && a.kst == 1 && a.D == 1 && a.os == 1 && p.kh0 == 0 &&
         p.kw0 == 0 && p.offy == -1 && p.offx == -1 && p.ophy == 0 && p.ophx == 0 && !a.reflect && a.IH == 2 * p.OHp &&
         a.IW == 2 * p.OWp;
}

// PAIR form (kernel comment): 64 output channels, four phases in pairs (0, 1), (2, 3) that differ in their column phase only
static bool x3h_pair_shape(const ConvArgs& a) {
  static const bool off = dbg_env("EG_NO_HALO_PAIR") != nullptr;
  if (off || a.nphase != 4 || a.N != 64 || a.S != 1 || a.pool_y || a.y2_leak) return false;
  for (int z = 0; z < 2; ++z) {
    const ConvPhase &p = a.ph[2 * z], &q = a.ph[2 * z + 1];
    const int d = p.offx - q.offx;
    if (p.OHp != q.OHp || p.OWp != q.OWp || p.Ty != q.Ty || p.Tx != q.Tx || p.kh0 != q.kh0 || p.offy != q.offy || p.ophy != q.ophy ||
        d < -1 || d > 1)
      return false;
  }
  return true;
}

static bool x3h_geometry(const ConvArgs& a_in, HaloGeom* g) {
  ConvArgs a = a_in;
  if (x3h_s2d_shape(a)) a.ph[0].Ty = a.ph[0].Tx = 2;   // the halo is that of the 2x2-tap space-to-depth form
  const int OH = a.ph[0].OHp, OW = a.ph[0].OWp;
  for (int p = 0; p < a.nphase; ++p)
    if (a.ph[p].OHp != OH || a.ph[p].OWp != OW) return false;
  if (OW < 4 || OH < 4 || (OW & (OW - 1)) != 0) return false;
  const int TW = std::min(16, OW);
  int TH = std::min(256 / TW, OH);
  if ((TH & (TH - 1)) != 0 || OH % TH != 0 || OW % TW != 0) return false;
  const int TI = 256 / (TW * TH);
  if (TI * TW * TH != 256 || a.NB % TI != 0) return false;
  static const bool dense = dbg_env("EG_X3H_DENSE") != nullptr;   // A/B switch: dense halo rows / row swizzle everywhere
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
  g->pair = 0;
  if (x3h_pair_shape(a)) {   // the shared halo is up to one column wider
    bool fits = true;
    for (int z = 0; z < 2; ++z) {
      const ConvPhase &p = a.ph[2 * z], &q = a.ph[2 * z + 1];
      const int span = p.offx > q.offx ? p.offx - q.offx : q.offx - p.offx, HW = TW + p.Tx - 1 + span;
      if (TI * (TH + p.Ty - 1) * HW > HRMAX || (g->pitch && HW > 20)) fits = false;
    }
    g->pair = fits ? 1 : 0;
  }
  return true;
}

// The pooled epilogue needs the 2x2 partners of every output pixel inside one 32-row MFMA tile of the block: even tile extents
// (TW >= 2 columns, TH >= 2 rows, 32 / TW rows per MFMA tile is then even), one phase, stride-1 output grid, a plain store
bool conv_x3h_pool_ok(const ConvArgs& a) {
  if (!a.pool_y || a.ksplit > 1 || a.nphase != 1 || a.os != 1 || a.S != 1 || a.accumulate || a.epi_act || a.y2_leak) return false;
  if ((a.OHf | a.OWf) & 1 || a.ph[0].ophy || a.ph[0].ophx || a.N % 4 != 0 || a.y_ps % 4 != 0) return false;
  if (((uintptr_t)a.pool_y | (uintptr_t)a.pool_b | (uintptr_t)a.pool_y2 | (uintptr_t)a.pool_y2_bf) & 15) return false;
  HaloGeom g;
  if (!x3h_geometry(a, &g)) return false;
  return g.lTW >= 1 && g.lTW <= 4 && g.lTH >= 1;
}

bool conv_x3h_eligible(const ConvArgs& a) {
  static const bool off = dbg_env("EG_NO_HALO") != nullptr;
  if (off || !a.wgt_bf || !(a.limbs == 3 || a.limbs == 1) || a.shuf) return false;
  static const bool no_s2d = dbg_env("EG_NO_S2D") != nullptr;
  const bool s2d = x3h_s2d_shape(a) && !no_s2d;
  if (!(a.S == 1 || s2d) || !(a.D == 1 || a.D == -1) || a.C % 32 != 0 || a.N < 64 || a.N % 4 != 0) return false;
  if (a.x_ps % 4 != 0 || a.y_ps % 4 != 0 || !a.x_bytes || !a.wbf_bytes || a.wbf_plane % 8 != 0) return false;
  if ((((uintptr_t)a.x | (uintptr_t)a.y | (uintptr_t)a.wgt_bf) & 15) != 0 || (a.bias && (((uintptr_t)a.bias) & 15) != 0)) return false;
  if (a.ksplit > 1 && a.ksplit > (s2d ? 4 : 1) * (a.C / 32)) return false;
  HaloGeom g;
  return x3h_geometry(a, &g);
}

template <int NPROD, int BN, bool S2D>
static void x3h_launch_variant(const ConvArgs& a, const HaloGeom& g, dim3 grid, hipStream_t s) {
  if (g.pitch) {
    if (a.ksplit > 1) hipLaunchKernelGGL((conv_gemm_x3h_kernel<NPROD, true, BN, true, S2D>), grid, dim3(512), 0, s, a, g);
    else hipLaunchKernelGGL((conv_gemm_x3h_kernel<NPROD, false, BN, true, S2D>), grid, dim3(512), 0, s, a, g);
  } else {
    if (a.ksplit > 1) hipLaunchKernelGGL((conv_gemm_x3h_kernel<NPROD, true, BN, false, S2D>), grid, dim3(512), 0, s, a, g);
    else hipLaunchKernelGGL((conv_gemm_x3h_kernel<NPROD, false, BN, false, S2D>), grid, dim3(512), 0, s, a, g);
  }
}

template <int BN>
static const char* x3h_launch(const ConvArgs& a, const HaloGeom& g, hipStream_t s) {
  const int zs = a.ksplit > 1 ? a.ksplit : 1;
  dim3 grid((a.NB / g.TI) * g.tiles_x * g.tiles_y, (a.N + BN - 1) / BN, a.nphase * zs);
  const bool s2d = x3h_s2d_shape(a);
  if (a.limbs == 1) {
    if (s2d) x3h_launch_variant<1, BN, true>(a, g, grid, s);
    else x3h_launch_variant<1, BN, false>(a, g, grid, s);
    if (s2d) return BN == 128 ? (a.ksplit > 1 ? "conv_gemm_bf16h<256,128,s2d,splitk>" : "conv_gemm_bf16h<256,128,s2d>")
                              : (a.ksplit > 1 ? "conv_gemm_bf16h<256,64,s2d,splitk>" : "conv_gemm_bf16h<256,64,s2d>");
    return BN == 128 ? (a.ksplit > 1 ? "conv_gemm_bf16h<256,128,splitk>" : "conv_gemm_bf16h<256,128>")
                     : (a.ksplit > 1 ? "conv_gemm_bf16h<256,64,splitk>" : "conv_gemm_bf16h<256,64>");
  }
  if (s2d) x3h_launch_variant<6, BN, true>(a, g, grid, s);
  else x3h_launch_variant<6, BN, false>(a, g, grid, s);
  if (s2d) return BN == 128 ? (a.ksplit > 1 ? "conv_gemm_x3h<256,128,s2d,splitk>" : "conv_gemm_x3h<256,128,s2d>")
                            : (a.ksplit > 1 ? "conv_gemm_x3h<256,64,s2d,splitk>" : "conv_gemm_x3h<256,64,s2d>");
  return BN == 128 ? (a.ksplit > 1 ? "conv_gemm_x3h<256,128,splitk>" : "conv_gemm_x3h<256,128>")
                   : (a.ksplit > 1 ? "conv_gemm_x3h<256,64,splitk>" : "conv_gemm_x3h<256,64>");
}

template <int NPROD>
static void x3h_launch_pair(const ConvArgs& a, const HaloGeom& g, hipStream_t s) {
  const int zs = a.ksplit > 1 ? a.ksplit : 1;
  dim3 grid((a.NB / g.TI) * g.tiles_x * g.tiles_y, 1, 2 * zs);
  if (g.pitch) {
    if (a.ksplit > 1) hipLaunchKernelGGL((conv_gemm_x3h_kernel<NPROD, true, 128, true, false, true>), grid, dim3(512), 0, s, a, g);
    else hipLaunchKernelGGL((conv_gemm_x3h_kernel<NPROD, false, 128, true, false, true>), grid, dim3(512), 0, s, a, g);
  } else {
    if (a.ksplit > 1) hipLaunchKernelGGL((conv_gemm_x3h_kernel<NPROD, true, 128, false, false, true>), grid, dim3(512), 0, s, a, g);
    else hipLaunchKernelGGL((conv_gemm_x3h_kernel<NPROD, false, 128, false, false, true>), grid, dim3(512), 0, s, a, g);
  }
}

const char* launch_conv_gemm_x3h(const ConvArgs& a_in, hipStream_t s) {
  ConvArgs a = a_in;
  if (!conv_x3h_pool_ok(a)) a.pool_y = nullptr;   // (the caller sees CONV_CAP_POOL missing and runs pool2_add itself)
  HaloGeom g;
  x3h_geometry(a, &g);
  static const bool dispatch_order = dbg_env("EG_NO_XCD_MAP") != nullptr;
  g.dispatch_order = dispatch_order;
  if (g.pair) {
    if (a.limbs == 1) x3h_launch_pair<1>(a, g, s);
    else x3h_launch_pair<6>(a, g, s);
    if (a.limbs == 1) return a.ksplit > 1 ? "conv_gemm_bf16h<256,2x64,splitk>" : "conv_gemm_bf16h<256,2x64>";
    return a.ksplit > 1 ? "conv_gemm_x3h<256,2x64,splitk>" : "conv_gemm_x3h<256,2x64>";
  }
  return a.N >= 128 ? x3h_launch<128>(a, g, s) : x3h_launch<64>(a, g, s);
}

}  // namespace eg
