// bf16-mode convolution fed from a bf16 PLANE of its gathered operand -- no VALU staging at all.
//
//   conv_bf16p_kernel : Y[pix_out][n] (+)= b[n] + sum_taps sum_c Xbf[pix_out + d(tap)][c] * Wbf[tap][c][n]
//                       stride-1 layers with 2 .. 9 taps per phase (3x3 convolutions and their input gradients, the
//                       sub-pixel phases of the stride-2 transposed convolutions / stride-2 input gradients) -- the
//                       layers of edgegan/nn/modules/conv.py:13-36,39-58,133-243 that k_gemm_x3h.hip serves in limb mode.
//
// Round 3's bf16 mode kept fp32 activations in HBM and rounded them on their way into LDS (k_gemm_x3h.hip, NPROD = 1): per
// 32-channel chunk 7 x 16-byte fp32 loads per thread into registers, a v_cvt_pk_bf16_f32 pass, ds_write_b64 stores, one block
// per CU (8 waves, 150 KB of LDS in the shared template) -- its K loop is 6x shorter than the limb kernel's, so the un-overlapped
// prologue / VALU phase / epilogue dominated: 520-650 TFLOP/s = 0.21-0.26 of the dense bf16 peak (profiles/r04b_layers_bf16.csv).
// Here the PRODUCER of the operand (InstanceNorm / PReLU / gate / pool kernels, or launch_cvt_bf16 for tensors whose producer
// has no plane store yet) leaves a bf16 copy [pixel][C] next to the fp32 tensor, and this kernel's halo image arrives by
// `buffer_load ... lds` (LDS-DMA): 16 halo rows x 64 bytes per wave-instruction, the XOR chunk swizzle applied on the
// GLOBAL side (the LDS destination of a DMA is lane-linear), zero padding = out-of-range buffer offsets, REFLECT = mirrored
// source pixels.  No staging registers, no conversion, no LDS stores.
//
// Shape: 256 threads = 4 waves, (WM x WN) wave grid, wave tile (32 MI) pixels x 64 channels of v_mfma_f32_32x32x16_bf16;
// block tile PX = 32 MI WM pixels x BN = 64 WN channels.  LDS = two halo buffers (chunk c + 1 lands while chunk c is
// multiplied) + a 2-slot filter ring = 43 KB at PX = 128, BN = 128: THREE blocks per CU, so one block's prologue, epilogue
// store burst and barrier bubbles are covered by its neighbours' MFMAs.  Per K step (one tap x 32 channels):
//   top:  s_waitcnt vmcnt(0)  (this wave's DMA pieces of step k-1: filter tile k+1, possibly the next halo)
//         s_waitcnt lgkmcnt(0) + s_barrier     (tile k+1 / the halo visible to every wave; every read of tile k done)
//         DMA: filter tile k+2 -> ring slot k & 1;  at a chunk's first tap: the NEXT chunk's halo -> the other buffer
//         look-ahead fragment reads of step k+1 (second register set)
//         MFMAs of step k
// The first version of this loop walked (chunk, tap) with run-time state: ~70 SALU + ~20 VALU instructions per step of 8
// MFMAs, and a CU has ONE scalar unit for its 12 resident waves -- with the DMA, the reads and the MFMAs all removed the
// launch still took 0.030 of its 0.096 ms (profiles/r04d_bf16p_limiter.txt).  Now the taps of a chunk AND pairs of chunks are
// unrolled at compile time (the phase's tap grid is a template argument of the loop): halo buffer, ring slot, register
// set, the tap's row / column shift and its filter offset are all immediates or loop-invariant registers -- a step of the
// 16-wide-tile form is 8 ds_read_b128 with immediate offsets, 8 MFMAs, 2 DMA requests, one wait + barrier, ~8 SALU, 0 VALU.
// Fragment reads are inline assembly (the compiler would guard ordinary LDS reads against the DMA in flight with
// vmcnt(0)); every fragment register is an in/out operand of the wait that retires its read.
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace eg {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef EG_BF16P_DBG
#define EG_BF16P_DBG 0   // timing experiments (wrong results): 1 no per-step barrier, 2 no halo DMA, 4 no filter DMA, 8 no MFMA, 16 no fragment reads, 32 no output stores, 64 no vmcnt wait at the step top
#endif
namespace {
__device__ __forceinline__ int swz_p(int row) { return ((-((row >> 2) & 3)) & 3) ^ ((row >> 4) & 1); }
__device__ __forceinline__ int reflect_p(int i, int n) {
  i = i < 0 ? -i : i;
  return i >= n ? 2 * (n - 1) - i : i;
}
}  // namespace

struct PlaneGeom {
  int lTW, lTH, TI;        // tile = TI images x 2^lTH rows x 2^lTW columns = PX output pixels
  int tiles_x, tiles_y;    // tiles per image
  int pitch;               // LDS row pitch of a halo row: 20 for 16-wide tiles (conflict-free layout), 0 = dense rows
};

template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// PH: the tap grids of the launch's phases -- 0: 3x3 (stride-1 convolutions and their input gradients), 1: 2x2 (the four
// phases of a 4x4 stride-2 input gradient), 2: any of 3x3 / 3x2 / 2x3 / 2x2 per phase (5x5 stride-2 transposed convolutions),
// 3: a 4x4 stride-2 SAME forward convolution (the discriminators' layers, edgegan/models/discriminator.py:61-72) in
//    space-to-depth form (k_gemm_x3h.hip, S2D): a 2x2-tap stride-1 convolution over the view V[Y][X][(py, px, c)] =
//    x[2Y + py - 1][2X + px - 1][c]; K chunk (q = 2 py + px, cq) stages phase (py, px) of the halo pixels' source pixels
//    straight from x's plane (a uniform phase offset + per-row validity of the 4 phases) and pairs with the filter taps
//    (2 dY + py, 2 dX + px)
template <int MI, int WN, bool SPLITK, bool SW16, int HR, int PH>
__global__ __launch_bounds__(256, ((HR <= 288 && SW16) ? 3 : 2)) void conv_bf16p_kernel(const ConvArgs a, const PlaneGeom g) {
  constexpr int BN = 64 * WN;                // block tile: (4 / WN) waves x 32 MI pixels by WN waves x 64 channels
  constexpr int WROWS = 32 * MI;
  constexpr int NSB = 2;                     // ring slots: tile k + 1 is being read while tile k + 2 lands in tile k's slot
  constexpr int APL = HR * 64;             // bytes of one halo buffer
  constexpr int BPL = BN * 64;             // bytes of one ring slot
  constexpr int NPC = HR / 16;             // DMA pieces (16 rows x 64 B) of one halo image
  constexpr int NAJ = (NPC + 3) / 4;       // ... per wave
  constexpr int NBJ = BN / 64;             // filter DMA pieces per wave and K step (16 rows each; 4 waves)
  static_assert(HR % 16 == 0, "halo rows come in 16-row DMA pieces");
  __shared__ __attribute__((aligned(16))) unsigned short Ah[2][HR * 32];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[NSB][BN * 32];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = WN == 2 ? wave >> 1 : wave, wn = WN == 2 ? wave & 1 : 0;
  // XCD-aware block map (as k_gemm_x3h.hip): the output-channel tiles of one pixel tile are neighbours inside one XCD
  const unsigned gxy = gridDim.x * gridDim.y;
  const unsigned vblk = xcd_remap((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, gxy * gridDim.z, 0);
  const int bz = __builtin_amdgcn_readfirstlane(vblk / gxy), bxy = __builtin_amdgcn_readfirstlane(vblk % gxy);
  const int bx = bxy / (int)gridDim.y, by = bxy - bx * (int)gridDim.y;
  const int zphase = SPLITK ? bz / a.ksplit : bz;
  const int zsplit = SPLITK ? bz - zphase * a.ksplit : 0;
  const ConvPhase ph = a.ph[zphase];
  constexpr bool S2D = PH == 3;
  const int IH = a.IH, IW = a.IW, C = a.C, N = a.N, D = S2D ? 1 : a.D;
  const int nTy = S2D ? 2 : ph.Ty, nTx = S2D ? 2 : ph.Tx;
  const int TW = 1 << g.lTW, TH = 1 << g.lTH;
  const int tpi = g.tiles_x * g.tiles_y;
  const int tn = bx / tpi, trem = bx - tn * tpi;
  const int tyy = trem / g.tiles_x, txx = trem - tyy * g.tiles_x;
  const int img0 = tn * g.TI, oy0 = tyy * TH, ox0 = txx * TW;
  const int n0 = by * BN;
  const int offy = S2D ? 0 : ph.offy, offx = S2D ? 0 : ph.offx;   // (S2D: ph.off* is the source offset of phase (0, 0), below)
  const int dminy = min(offy, (nTy - 1) * D + offy), dminx = min(offx, (nTx - 1) * D + offx);
  const int HH = TH + nTy - 1, HW = TW + nTx - 1;
  constexpr bool sw16 = SW16;
  const int HWP = sw16 ? 20 : HW, HPI = HH * HWP;

  // ---- halo DMA: piece p = wave + 4 j covers LDS rows 16 p .. 16 p + 15; lane (r, q) = (lane >> 2, lane & 3) brings the
  //      16-byte channel group (q ^ swizzle(row)) of the row's source pixel; rows without a pixel / padding read zeros -----
  unsigned goffA[NAJ];
  [[maybe_unused]] int vbA[NAJ];   // S2D: bits 0 / 1 = source row of phase py = 0 / 1 inside the image, bits 2 / 3 = column of px = 0 / 1
#pragma unroll
  for (int j = 0; j < NAJ; ++j) {
    const int row = (wave + 4 * j) * 16 + (lane >> 2);
    unsigned off = 0x80000000u;
    vbA[j] = 0;
    if (wave + 4 * j < NPC) {
      const int ti = row / HPI, rem = row - ti * HPI;
      const int hy = rem / HWP, hx = rem - hy * HWP;
      if (ti < g.TI && hx < HW) {
        int iy = oy0 + dminy + hy, ix = ox0 + dminx + hx;
        if constexpr (S2D) {   // source pixel of phase (0, 0); the offset may be "negative" (wraps; only used where valid)
          iy = 2 * iy + ph.offy;
          ix = 2 * ix + ph.offx;
          const int sw = sw16 ? (hx >> 2) & 3 : swz_p(row);
          off = (unsigned)(((((img0 + ti) * IH + iy) * IW + ix) * a.xbf_ps + (((lane & 3) ^ sw) * 8)) * 2);
          vbA[j] = ((unsigned)iy < (unsigned)IH ? 1 : 0) | ((unsigned)(iy + 1) < (unsigned)IH ? 2 : 0) |
                   ((unsigned)ix < (unsigned)IW ? 4 : 0) | ((unsigned)(ix + 1) < (unsigned)IW ? 8 : 0);
          iy = ix = -1;        // (the generic path below does not apply)
        }
        if (a.reflect) {
          iy = reflect_p(iy, IH);
          ix = reflect_p(ix, IW);
        }
        if ((unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW) {
          const int sw = sw16 ? (hx >> 2) & 3 : swz_p(row);
          if (a.x_up)   // (the source is the half-resolution plane: four output-side pixels read one of its pixels)
            off = (unsigned)(((((img0 + ti) * (IH >> 1) + (iy >> 1)) * (IW >> 1) + (ix >> 1)) * a.xbf_ps + (((lane & 3) ^ sw) * 8)) * 2);
          else
            off = (unsigned)(((((img0 + ti) * IH + iy) * IW + ix) * a.xbf_ps + (((lane & 3) ^ sw) * 8)) * 2);
        }
      }
    }
    goffA[j] = off;
  }
  const __amdgpu_buffer_rsrc_t xres =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.x_bf), 0, a.xbf_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wres =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.wgt_bf), 0, a.wbf_bytes, 0x00020000);
  // (hipcc 7.2 quirk: with the DMA builtin's voffset argument written as an element of a template-sized local array the HOST
  //  pass silently drops the kernel's launch stub -- "undefined symbol __device_stub__..." at load time; a scalar copy is fine)
  const int cpt = C / 32;
  auto request_halo = [&](int cc, int buf) __attribute__((always_inline)) {
    [[maybe_unused]] int py = 0, px = 0;
    [[maybe_unused]] unsigned soff = 0;
    if constexpr (S2D) {   // K chunk cc = phase q, channel chunk cq (uniform arithmetic, once per chunk)
      const int q = cc / cpt, cq = cc - q * cpt;
      py = q >> 1;
      px = q & 1;
      soff = (unsigned)((((py * IW + px) * a.xbf_ps) + cq * 32) * 2);
    }
#pragma unroll
    for (int j = 0; j < NAJ; ++j)
      if (wave + 4 * j < NPC && (EG_BF16P_DBG & 2) == 0)
      {
        if constexpr (S2D) {   // (the sum is formed here, mod 2^32: a "negative" phase-(0,0) offset plus the phase offset is in range again)
          const bool ok = ((vbA[j] >> py) & (vbA[j] >> (2 + px)) & 1) != 0;
          const unsigned voff = ok ? goffA[j] + soff : 0x80000000u;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (__attribute__((address_space(3))) void*)&Ah[buf][(wave + 4 * j) * 16 * 32], 16,
                                                   voff, 0, 0, 0);
        } else {
          const unsigned voff = goffA[j];
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (__attribute__((address_space(3))) void*)&Ah[buf][(wave + 4 * j) * 16 * 32], 16,
                                                   voff, cc * 64, 0, 0);
        }
      }
  };

  // ---- filter ring: piece (wave, j) fills rows (wave NBJ + j) 16 + lane / 4 of the slot -----------------------------------
  unsigned woff[NBJ];
#pragma unroll
  for (int j = 0; j < NBJ; ++j) {
    const int row = (wave * NBJ + j) * 16 + (lane >> 2);
    woff[j] = (unsigned)(((size_t)min(n0 + row, N - 1) * 32 + ((lane & 3) ^ swz_p(row)) * 8) * 2);
  }
  const int wtile = N * 32 * 2;
  const int vcpt = S2D ? 4 * cpt : cpt;   // K chunks the loop walks (S2D: 4 phases x C / 32)
  int cc_lo = 0, cc_hi = vcpt;
  if constexpr (SPLITK) {
    const int per = (vcpt + a.ksplit - 1) / a.ksplit;
    cc_lo = min(vcpt, zsplit * per);
    cc_hi = min(vcpt, cc_lo + per);
  }
  const int ncc = cc_hi - cc_lo;

  // ---- fragments --------------------------------------------------------------------------------------------------------
  const int l31 = lane & 31, lh = lane >> 5;
  const unsigned ah_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&Ah[0][0];
  const unsigned bs_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&Bs[0][0];
  int base_row[MI], base_px[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int r = wm * WROWS + i * 32 + l31;
    const int ti = r >> (g.lTW + g.lTH), py = (r >> g.lTW) & (TH - 1), px = r & (TW - 1);
    base_row[i] = ti * HPI + py * HWP + px;
    base_px[i] = px;
  }
  // 16-wide tiles: the fragment address of (32-row block i, column shift e, K half kk) is a per-lane constant -- the row
  // shift of a tap and the halo buffer are immediate offsets (pitch 20 rows of 64 bytes)
  // (the wave's 32-row blocks i are 2 tile rows = 40 halo rows apart and share the pixel column: another immediate)
  unsigned abase[3][2];
#pragma unroll
  for (int e = 0; e < 3; ++e)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
      abase[e][kk] = ah_base + (base_row[0] + e) * 64 + ((((kk << 1) | lh) ^ (((base_px[0] + e) >> 2) & 3)) * 16);
  const unsigned bb0 = bs_base + ((wn * 64 + l31) * 32 + ((lh ^ swz_p(l31)) * 8)) * 2, bb1 = bb0 ^ 32u;

  f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#if (EG_BF16P_DBG & 16)
#define EG_DSR(dst, addr, off) asm volatile("" : "+v"(dst) : "v"(addr), "n"(off))
#else
#define EG_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#endif
  struct Frags {
    bf16x8 a[MI][2], b[2][2];
  };
  // s_waitcnt lgkmcnt(0) (+ barrier) with every register of the fragment set as in/out operand
  auto land = [&](Frags& f, auto BAR) __attribute__((always_inline)) {
    if constexpr (MI == 2) {
      if constexpr (decltype(BAR)::value)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+v"(f.a[0][0]), "+v"(f.a[0][1]), "+v"(f.a[1][0]), "+v"(f.a[1][1]), "+v"(f.b[0][0]), "+v"(f.b[0][1]), "+v"(f.b[1][0]), "+v"(f.b[1][1])::"memory");
      else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.a[0][0]), "+v"(f.a[0][1]), "+v"(f.a[1][0]), "+v"(f.a[1][1]), "+v"(f.b[0][0]), "+v"(f.b[0][1]), "+v"(f.b[1][0]), "+v"(f.b[1][1])::"memory");
    } else if constexpr (MI == 1) {
      if constexpr (decltype(BAR)::value)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+v"(f.a[0][0]), "+v"(f.a[0][1]), "+v"(f.b[0][0]), "+v"(f.b[0][1]), "+v"(f.b[1][0]), "+v"(f.b[1][1])::"memory");
      else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.a[0][0]), "+v"(f.a[0][1]), "+v"(f.b[0][0]), "+v"(f.b[0][1]), "+v"(f.b[1][0]), "+v"(f.b[1][1])::"memory");
    } else {
      static_assert(MI == 4, "wave tiles of 32, 64 or 128 pixels");
      if constexpr (decltype(BAR)::value)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+v"(f.a[0][0]), "+v"(f.a[0][1]), "+v"(f.a[1][0]), "+v"(f.a[1][1]), "+v"(f.a[2][0]), "+v"(f.a[2][1]), "+v"(f.a[3][0]), "+v"(f.a[3][1]), "+v"(f.b[0][0]), "+v"(f.b[0][1]), "+v"(f.b[1][0]), "+v"(f.b[1][1])::"memory");
      else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.a[0][0]), "+v"(f.a[0][1]), "+v"(f.a[1][0]), "+v"(f.a[1][1]), "+v"(f.a[2][0]), "+v"(f.a[2][1]), "+v"(f.a[3][0]), "+v"(f.a[3][1]), "+v"(f.b[0][0]), "+v"(f.b[0][1]), "+v"(f.b[1][0]), "+v"(f.b[1][1])::"memory");
    }
  };
  constexpr std::true_type WITH_BARRIER{};
  constexpr std::false_type NO_BARRIER{};
  auto mm = [&](const Frags& f) __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr ((EG_BF16P_DBG & 8) != 0) asm volatile("" : "+v"(acc[i][j]) : "v"(f.b[j][kk]), "v"(f.a[i][kk]));
          else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b[j][kk], f.a[i][kk], acc[i][j], 0, 0, 0);
        }
  };

  // ---- the K loop of one phase: TY x TX taps (row shift f, column shift e of the halo; the filter tap they pair with
  //      depends on the sign of D), walked chunk by chunk; everything about a tap is a compile-time constant ----------------
  auto kloop = [&](auto TYc, auto TXc) __attribute__((always_inline)) {
    constexpr int TY = decltype(TYc)::value, TX = decltype(TXc)::value, NT = TY * TX;
    // byte offset of tap t's filter tile inside a chunk row of the K-step-major plane ([tap][C/32][N][32] bf16)
    int tapoff[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int f = t / TX, e = t - f * TX;
      const int ty = D > 0 ? f : TY - 1 - f, tx = D > 0 ? e : TX - 1 - e;
      if constexpr (S2D) tapoff[t] = ((2 * f) * a.KW + 2 * e) * cpt * wtile;   // + the chunk's phase (py, px): wrow_of
      else tapoff[t] = ((ph.kh0 + ty * a.kst) * a.KW + (ph.kw0 + tx * a.kst)) * cpt * wtile;
    }
    // byte offset of K chunk cc inside a tap's filter tiles (S2D: + the tap shift of the chunk's phase)
    auto wrow_of = [&](int cc) __attribute__((always_inline)) {
      if constexpr (S2D) {
        const int q = cc / cpt, cq = cc - q * cpt;
        return (((q >> 1) * a.KW + (q & 1)) * cpt + cq) * wtile;
      } else {
        return cc * wtile;
      }
    };
    // two fragment sets (the look-ahead reads of step k + 1 are issued before the MFMAs of step k) where the registers
    // allow it; the 128 x 64 wave tile (128 accumulator registers) keeps ONE set: reads after the MFMAs that consumed it
    constexpr int NSETS = MI == 4 ? 1 : 2;
    Frags fs[NSETS] = {};
    // fragment reads of step (halo buffer BUF, tap T, ring slot SLOT) -- all three compile-time
    auto read_step = [&](Frags& fr, auto BUFc, auto Tc, auto SLOTc) __attribute__((always_inline)) {
      constexpr int BUF = decltype(BUFc)::value, T = decltype(Tc)::value, SLOT = decltype(SLOTc)::value;
      constexpr int f = T / TX, e = T - f * TX;
      if constexpr (SW16) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          // (named copies: a variable that a generic lambda uses ONLY as an asm operand is not captured by hipcc 7.2)
          const unsigned p0 = abase[e][0], p1 = abase[e][1];
          EG_DSR(fr.a[i][0], p0, BUF * APL + (f + 2 * i) * 20 * 64);
          EG_DSR(fr.a[i][1], p1, BUF * APL + (f + 2 * i) * 20 * 64);
        }
      } else {
        const int toff = f * HWP + e;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int row = base_row[i] + toff;
          const unsigned aa = ah_base + row * 64 + ((lh ^ swz_p(row)) * 16);
          EG_DSR(fr.a[i][0], aa, BUF * APL);
          EG_DSR(fr.a[i][1], aa ^ 32u, BUF * APL);
        }
      }
      const unsigned q0 = bb0, q1 = bb1;
      EG_DSR(fr.b[0][0], q0, SLOT * BPL);
      EG_DSR(fr.b[0][1], q1, SLOT * BPL);
      EG_DSR(fr.b[1][0], q0, SLOT * BPL + 2048);
      EG_DSR(fr.b[1][1], q1, SLOT * BPL + 2048);
    };
    auto request_b = [&](auto SLOTc, int wsoff) __attribute__((always_inline)) {
      constexpr int SLOT = decltype(SLOTc)::value;
#pragma unroll
      for (int j = 0; j < NBJ; ++j)
        if constexpr ((EG_BF16P_DBG & 4) == 0) {
          const unsigned voff = woff[j];
          __builtin_amdgcn_raw_ptr_buffer_load_lds(wres, (__attribute__((address_space(3))) void*)&Bs[SLOT][(wave * NBJ + j) * 16 * 32], 16,
                                                   voff, wsoff, 0, 0);
        }
    };
    constexpr std::integral_constant<int, 0> I0{};
    constexpr std::integral_constant<int, 1> I1{};
    if (ncc > 0) {
      request_halo(cc_lo, 0);
      request_b(I0, tapoff[0] + wrow_of(cc_lo));
      request_b(I1, tapoff[1] + wrow_of(cc_lo));     // (NT >= 2)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
      read_step(fs[0], I0, I0, I0);
    }
    // one chunk, the H-th of a pair (H = 0: halo buffer 0, H = 1: buffer 1); global step k = (2 p + H) NT + T, so the
    // register set and ring slot of a step are (H NT + T) & 1
    auto chunk = [&](auto Hc, int cc, bool has_next) __attribute__((always_inline)) {
      constexpr int H = decltype(Hc)::value;
      const int wrow = wrow_of(cc), wrow_next = has_next ? wrow_of(cc + 1) : 0;
      static_for<NT>([&](auto Tc) __attribute__((always_inline)) {
        constexpr int T = decltype(Tc)::value;
        constexpr int SLOT = (H * NT + T) & 1;                   // ring slot of tile k (and of tile k + 2)
        constexpr int SET = NSETS == 2 ? SLOT : 0, NXT = NSETS == 2 ? SLOT ^ 1 : 0;
        constexpr std::integral_constant<int, SLOT> SLOTc{};
        constexpr std::integral_constant<int, SLOT ^ 1> NSLOTc{};
        if constexpr ((EG_BF16P_DBG & 64) == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr ((EG_BF16P_DBG & 1) != 0) land(fs[SET], NO_BARRIER);
        else land(fs[SET], WITH_BARRIER);
        // filter tile k + 2 -> the slot tile k was read from
        if constexpr (T + 2 < NT) request_b(SLOTc, tapoff[T + 2] + wrow);
        else if (has_next) request_b(SLOTc, tapoff[T + 2 - NT] + wrow_next);
        if constexpr (T == 0)
          if (has_next) request_halo(cc + 1, H ^ 1);
        if constexpr (NSETS == 1) {
          mm(fs[0]);
          __builtin_amdgcn_sched_barrier(0);
        }
        // the fragments of step k + 1
        if constexpr (T + 1 < NT) read_step(fs[NXT], Hc, std::integral_constant<int, T + 1>{}, NSLOTc);
        else if (has_next) read_step(fs[NXT], std::integral_constant<int, H ^ 1>{}, I0, NSLOTc);
        if constexpr (NSETS == 2) mm(fs[SET]);
        __builtin_amdgcn_sched_barrier(0);
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
    if constexpr (NSETS == 2) land(fs[1], NO_BARRIER);
    __builtin_amdgcn_sched_barrier(0);
  };
  constexpr std::integral_constant<int, 2> K2{};
  constexpr std::integral_constant<int, 3> K3{};
  if constexpr (PH == 0) {
    kloop(K3, K3);
  } else if constexpr (PH == 1 || PH == 3) {
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
            const int n = n0 + wn * 64 + j * 32 + q * 8 + lh * 4;
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
        const int n = n0 + wn * 64 + j * 32 + q * 8 + lh * 4;
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
          if (((EG_BF16P_DBG & 32) == 0 || v.x == 123.456f) && (SPLITK || !a.epi_skip_f32)) *reinterpret_cast<float4*>(yrow + n) = v;
          if constexpr (!SPLITK) {
            if (a.y2_leak) conv_epi_y2(a, (size_t)(yrow - a.y) + n, v, y2lk);
            if (a.epi_bf) conv_epi_plane(a.epi_bf + (size_t)(yrow - a.y) + n, v);
          }
        }
      }
  }
}

static int ilog2p(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

static bool bf16p_s2d_shape(const ConvArgs& a) {   // 4x4 stride-2 SAME forward convolution (one phase, taps as stored)
  const ConvPhase& p = a.ph[0];
  return a.S == 2 && a.nphase == 1 && a.KW == 4 && p.Ty == 4 && p.Tx == 4 && a.kst == 1 && a.D == 1 && a.os == 1 && p.kh0 == 0 &&
         p.kw0 == 0 && p.offy == -1 && p.offx == -1 && p.ophy == 0 && p.ophx == 0 && !a.reflect && a.IH == 2 * p.OHp &&
         a.IW == 2 * p.OWp;
}

// tile geometry for PX output pixels per block and at most HR halo rows; false when the kernel does not take the shape
static bool bf16p_geometry(const ConvArgs& a_in, int PX, int HR, PlaneGeom* g) {
  ConvArgs a = a_in;
  if (bf16p_s2d_shape(a)) a.ph[0].Ty = a.ph[0].Tx = 2;   // the halo is that of the 2x2-tap space-to-depth form
  const int OH = a.ph[0].OHp, OW = a.ph[0].OWp;
  for (int p = 0; p < a.nphase; ++p)
    if (a.ph[p].OHp != OH || a.ph[p].OWp != OW) return false;
  if (OW < 4 || OH < 4 || (OW & (OW - 1)) != 0) return false;
  const int TW = std::min(16, OW);
  int TH = std::min(PX / TW, OH);
  if ((TH & (TH - 1)) != 0 || OH % TH != 0 || OW % TW != 0) return false;
  const int TI = PX / (TW * TH);
  if (TI * TW * TH != PX || a.NB % TI != 0) return false;
  g->pitch = TW == 16 ? 20 : 0;
  for (int p = 0; p < a.nphase; ++p)
    if (TI * (TH + a.ph[p].Ty - 1) * 20 > HR) g->pitch = 0;
  for (int p = 0; p < a.nphase; ++p) {
    const ConvPhase& ph = a.ph[p];
    if (ph.Ty < 2 || ph.Ty > 3 || ph.Tx < 2 || ph.Tx > 3) return false;   // (the K loop's tap grids)
    if (TI * (TH + ph.Ty - 1) * (TW + ph.Tx - 1) > HR) return false;
  }
  g->lTW = ilog2p(TW);
  g->lTH = ilog2p(TH);
  g->TI = TI;
  g->tiles_x = OW / TW;
  g->tiles_y = OH / TH;
  return true;
}

// halo rows of the 128-pixel form: 288 = 8 images of 4x4 outputs (6 x 6 halo each); 53 KB of LDS, still three blocks per CU
constexpr int kHR128 = 288, kHR256 = 400;

// K split of a plane-fed launch.  The kernel's time is (blocks x K steps) / 256 CUs x ~0.27 us whatever the split (measured:
// profiles/r04h_layers_bf16.csv), so the split only has to give every CU two blocks -- each further slab is 4 bytes per output
// element written and read again by the reduction (round 3's cost model, tuned for 256-pixel tiles at one block per CU,
// chose 8 slabs for the classifier's last unit: 100 MB of slab traffic around a 43 GFLOP launch).
struct X3Tuned {   // (as in k_gemm.hip)
  int M, N, C, taps, nphase, S, tile, ks;
};
static const X3Tuned kBf16pTuned[] = {
#include "bf16p_tuned.h"
    {0, 0, 0, 0, 0, 0, 0, 0}};
int conv_bf16p_pick_ksplit(const ConvArgs& a) {
  PlaneGeom g;
  const bool n128 = a.N >= 128;
  if (!bf16p_geometry(a, n128 ? 128 : 256, n128 ? kHR128 : kHR256, &g)) return 1;
  const long long tiles = (long long)(a.NB / g.TI) * g.tiles_x * g.tiles_y * ((a.N + (n128 ? 127 : 63)) / (n128 ? 128 : 64)) * a.nphase;
  const int chunks = (bf16p_s2d_shape(a) ? 4 : 1) * (a.C / 32);
  static const int target = dbg_env("EG_BF16P_KS_BLOCKS") ? atoi(dbg_env("EG_BF16P_KS_BLOCKS")) : 256;   // in-box 2,961 / 2,963 / 2,997 / 2,971 / 2,948 images/s at 512 / 384 / 256 / 192 / 128 (profiles/r04v_ab_ksplit.txt)
#ifdef EG_EXPERIMENTS
  if (const char* f = dbg_env("EG_BF16P_FORCE_KS")) {   // tools/tune_x3.py bf16: every plane-fed launch on one K split (clipped)
    const int fk = atoi(f);
    return std::max(1, std::min(fk, std::min(8, chunks / 2)));
  }
#endif
  for (const X3Tuned* t = kBf16pTuned; t->M; ++t) {   // measured per-shape splits where the rule below misses (tools/tune_x3.py)
    int maxM = 0, taps = 0;
    for (int p = 0; p < a.nphase; ++p) {
      maxM = std::max(maxM, a.NB * a.ph[p].OHp * a.ph[p].OWp);
      taps += a.ph[p].Ty * a.ph[p].Tx;
    }
    if (t->M == maxM && t->N == a.N && t->C == a.C && t->taps == taps && t->nphase == a.nphase && t->S == a.S)
      return std::min(t->ks, std::max(1, chunks / 2));
  }
  if (tiles >= target * 7 / 8 || chunks < 4) return 1;
  int ks = (int)((target + tiles - 1) / tiles);
  ks = std::min(ks, std::min(8, chunks / 2));
  return ks < 2 ? 1 : ks;
}

bool conv_bf16p_eligible(const ConvArgs& a) {
  static const bool off = dbg_env("EG_NO_BF16P") != nullptr;
  if (off || !a.x_bf || !a.wgt_bf || a.limbs != 1 || a.shuf) return false;
  static const bool no_s2d = dbg_env("EG_NO_S2D") != nullptr;
  const bool s2d = bf16p_s2d_shape(a) && !no_s2d;
  if (!(a.S == 1 || s2d) || !(a.D == 1 || a.D == -1) || a.C % 32 != 0 || a.N < 64 || a.N % 4 != 0) return false;
  if (a.xbf_ps % 8 != 0 || a.y_ps % 4 != 0 || !a.xbf_bytes || !a.wbf_bytes) return false;
  if ((((uintptr_t)a.x_bf | (uintptr_t)a.y | (uintptr_t)a.wgt_bf) & 15) != 0 || (a.bias && (((uintptr_t)a.bias) & 15) != 0)) return false;
  if (a.ksplit > 1 && a.ksplit > (s2d ? 4 : 1) * (a.C / 32)) return false;
  PlaneGeom g;
  return bf16p_geometry(a, a.N >= 128 ? 128 : 256, a.N >= 128 ? kHR128 : kHR256, &g);
}

// tap-grid class of a launch (the kernel's PH): 0 all phases 3x3, 1 all 2x2, 2 mixed (each of Ty, Tx in {2, 3}); -1: none
static int bf16p_phase_class(const ConvArgs& a) {
  bool all33 = true, all22 = true;
  for (int p = 0; p < a.nphase; ++p) {
    const int ty = a.ph[p].Ty, tx = a.ph[p].Tx;
    if (ty < 2 || ty > 3 || tx < 2 || tx > 3) return -1;
    all33 = all33 && ty == 3 && tx == 3;
    all22 = all22 && ty == 2 && tx == 2;
  }
  return all33 ? 0 : all22 ? 1 : 2;
}

template <int MI, int WN, int HR, int PH>
static void bf16p_launch_ph(const ConvArgs& a, const PlaneGeom& g, dim3 grid, hipStream_t s) {
  if (g.pitch) {
    if (a.ksplit > 1) hipLaunchKernelGGL((conv_bf16p_kernel<MI, WN, true, true, HR, PH>), grid, dim3(256), 0, s, a, g);
    else hipLaunchKernelGGL((conv_bf16p_kernel<MI, WN, false, true, HR, PH>), grid, dim3(256), 0, s, a, g);
  } else {
    if (a.ksplit > 1) hipLaunchKernelGGL((conv_bf16p_kernel<MI, WN, true, false, HR, PH>), grid, dim3(256), 0, s, a, g);
    else hipLaunchKernelGGL((conv_bf16p_kernel<MI, WN, false, false, HR, PH>), grid, dim3(256), 0, s, a, g);
  }
}

template <int MI, int WN, int HR>
static void bf16p_launch_variant(const ConvArgs& a, const PlaneGeom& g, dim3 grid, hipStream_t s) {
  if (bf16p_s2d_shape(a)) return bf16p_launch_ph<MI, WN, HR, 3>(a, g, grid, s);
  const int pc = bf16p_phase_class(a);
  if (pc == 0) bf16p_launch_ph<MI, WN, HR, 0>(a, g, grid, s);
  else if (pc == 1) bf16p_launch_ph<MI, WN, HR, 1>(a, g, grid, s);
  else bf16p_launch_ph<MI, WN, HR, 2>(a, g, grid, s);
}

// the pooled epilogue (kernels.h: ConvArgs::pool_y): see conv_x3h_pool_ok -- every tile form here has TW >= 4, TH >= 2
bool conv_bf16p_pool_ok(const ConvArgs& a) {
  if (!a.pool_y || a.ksplit > 1 || a.nphase != 1 || a.os != 1 || a.S != 1 || a.accumulate || a.epi_act || a.y2_leak) return false;
  if ((a.OHf | a.OWf) & 1 || a.ph[0].ophy || a.ph[0].ophx || a.N % 4 != 0 || a.y_ps % 4 != 0) return false;
  if (((uintptr_t)a.pool_y | (uintptr_t)a.pool_b | (uintptr_t)a.pool_y2 | (uintptr_t)a.pool_y2_bf) & 15) return false;
  PlaneGeom g;
  if (!bf16p_geometry(a, a.N >= 128 ? 128 : 256, a.N >= 128 ? kHR128 : kHR256, &g)) return false;
  return g.lTW >= 1 && g.lTW <= 4 && g.lTH >= 1;
}

const char* launch_conv_bf16p(const ConvArgs& a_in, hipStream_t s) {
  ConvArgs a = a_in;
  if (!conv_bf16p_pool_ok(a)) a.pool_y = nullptr;   // (the caller sees CONV_CAP_POOL missing and runs pool2_add itself)
  PlaneGeom g;
  const int zs = a.ksplit > 1 ? a.ksplit : 1;
  // 256 pixels x 128 channels, waves 2 x 2 with 128 x 64 wave tiles (16 MFMAs per 12 fragment reads, half the filter DMA
  // per MFMA of the 128-pixel form; 2 blocks per CU): when there are at least two such blocks for every CU
  static const bool big_on = dbg_env("EG_BF16P_NO_BIG") == nullptr;
  if (big_on && a.N >= 128 && bf16p_geometry(a, 256, kHR256, &g) && g.pitch) {
    const long long blocks = (long long)(a.NB / g.TI) * g.tiles_x * g.tiles_y * ((a.N + 127) / 128) * a.nphase * zs;
    if (blocks >= 512) {
      dim3 grid((a.NB / g.TI) * g.tiles_x * g.tiles_y, (a.N + 127) / 128, a.nphase * zs);
      bf16p_launch_variant<4, 2, kHR256>(a, g, grid, s);
      return a.ksplit > 1 ? "conv_bf16p<256,128,splitk>" : "conv_bf16p<256,128>";
    }
  }
  if (a.N >= 128) {   // 128 pixels x 128 channels, waves 2 x 2
    bf16p_geometry(a, 128, kHR128, &g);
    dim3 grid((a.NB / g.TI) * g.tiles_x * g.tiles_y, (a.N + 127) / 128, a.nphase * zs);
    bf16p_launch_variant<2, 2, kHR128>(a, g, grid, s);
    return a.ksplit > 1 ? "conv_bf16p<128,128,splitk>" : "conv_bf16p<128,128>";
  }
  // 64 .. 127 output channels: 256 pixels x 64 channels, waves 4 x 1
  bf16p_geometry(a, 256, kHR256, &g);
  dim3 grid((a.NB / g.TI) * g.tiles_x * g.tiles_y, (a.N + 63) / 64, a.nphase * zs);
  bf16p_launch_variant<2, 1, kHR256>(a, g, grid, s);
  return a.ksplit > 1 ? "conv_bf16p<256,64,splitk>" : "conv_bf16p<256,64>";
}

}  // namespace eg
