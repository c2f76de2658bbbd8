// bf16 contractions with BOTH operands already stored as bf16 (compute_dtype bf16: the activation's bf16 copy and the
// filter's K-step-major bf16 plane): a kernel whose waves issue nothing but LDS-DMA requests, fragment reads and MFMAs.
//
//   conv_gemm_dma_kernel : Y[pix_out(m)][n] (+)= b[n] + sum_taps sum_c Xb[pix_in(m,tap)][c] * Wb[tap][c/32][n][c%32]
//
// Block = 256 threads = 4 waves (2 x 2), tile 256 output pixels x 128 output channels, wave tile 128 x 64 = 4 x 2
// accumulators of v_mfma_f32_32x32x16_bf16 (128 VGPRs): per K step of 32 channels a wave reads 12 KB of fragments for
// 16 MFMAs, i.e. the block draws 96 bytes/clk from LDS against its 128 bytes/clk -- a 64 x 64 wave tile (16 KB per 8
// MFMAs) would be LDS-bound at half the matrix rate, which is where the register-staged bf16 kernels sit.
// Both tiles go global -> LDS by buffer_load ... lds (no registers, no ds_write): a wave instruction fills 16 rows x
// 64 bytes; lane l owns LDS bytes 16 l .. +15 behind the instruction's base, so the XOR chunk swizzle that makes the
// fragment reads conflict-free is applied on the GLOBAL side (lane l fetches logical chunk (l & 3) ^ swz(row)).
// Out-of-range offsets return zeros: padding taps and rows beyond M need no branches.  NS-stage ring, requests NS - 1
// K steps ahead, one barrier per K step.
#include <cstdlib>

#include "kernels.h"

namespace eg {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int BKD = 32;    // K step
constexpr int ROWD = BKD;  // LDS row pitch in bf16 elements (64 bytes)
__device__ __forceinline__ int swz_d(int row) { return ((-((row >> 2) & 3)) & 3) ^ ((row >> 4) & 1); }
}  // namespace

template <bool SPLITK>
__global__ __launch_bounds__(256, 1) void conv_gemm_dma_kernel(const ConvArgs a) {
  constexpr int BM = 256, BN = 128, WTM = 128, WTN = 64, NS = 5;
  constexpr int NDMA = BM / 64 + BN / 64;   // DMA instructions per wave and K step
  __shared__ __attribute__((aligned(16))) unsigned short As[NS][BM * ROWD];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[NS][BN * ROWD];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int zphase = SPLITK ? (int)blockIdx.z / a.ksplit : (int)blockIdx.z;
  const int zsplit = SPLITK ? (int)blockIdx.z - zphase * a.ksplit : 0;
  const ConvPhase ph = a.ph[zphase];
  const int M = a.NB * ph.OHp * ph.OWp;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  if (m0 >= M) return;
  const int IH = a.IH, IW = a.IW, C = a.C, N = a.N;

  // ---- per-lane rows of the gathered tile: instruction i fills rows i*64 + wave*16 + lane/4 ---------------------------
  const int lsw = (((-((lane >> 4) & 3)) & 3) ^ (wave & 1));   // swz(i*64 + wave*16 + lane/4)
  const int chunk = (lane & 3) ^ lsw;
  unsigned r_off0[BM / 64], r_mask[BM / 64];
#pragma unroll
  for (int i = 0; i < BM / 64; ++i) {
    const int m = m0 + i * 64 + wave * 16 + (lane >> 2);
    r_mask[i] = 0u;
    r_off0[i] = 0u;
    if (m < M) {
      const int ox = m % ph.OWp;
      const int t2 = m / ph.OWp;
      const int oy = t2 % ph.OHp;
      const int nb = t2 / ph.OHp;
      const int iy0 = oy * a.S + ph.offy, ix0 = ox * a.S + ph.offx;
      for (int ty = 0; ty < ph.Ty; ++ty)
        for (int tx = 0; tx < ph.Tx; ++tx) {
          const int iy = iy0 + ty * a.D, ix = ix0 + tx * a.D;
          if ((unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW) r_mask[i] |= 1u << (ty * ph.Tx + tx);
        }
      r_off0[i] = (unsigned)((((nb * IH + iy0) * IW + ix0) * a.x_ps + chunk * 8) * 2);
    }
  }
  unsigned woff[BN / 64];
#pragma unroll
  for (int i = 0; i < BN / 64; ++i)
    woff[i] = (unsigned)(((size_t)min(n0 + i * 64 + wave * 16 + (lane >> 2), N - 1) * BKD + chunk * 8) * 2);
  const __amdgpu_buffer_rsrc_t xres =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.x_bf), 0, a.x_bytes / 2, 0x00020000);   // (x_bytes: fp32 extent)
  const __amdgpu_buffer_rsrc_t wres =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.wgt_bf), 0, a.wbf_bytes, 0x00020000);

  const int ntaps = ph.Ty * ph.Tx;
  const int cpt = C / BKD;
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
  const int nt = kt_hi - kt_lo;
  const int wtile = N * BKD * 2;   // bytes of one (tap, channel chunk) filter tile

  // request K step (nx_ty, nx_tx, nx_cc) into ring slot `st`; steps past the end request nothing (the vmcnt bookkeeping
  // below counts instructions, so the tail waits for everything)
  auto request = [&](int st) __attribute__((always_inline)) {
    const int ty = nx_ty, tx = nx_tx, cc = nx_cc;
    const int tapbit = ty * ph.Tx + tx;
    const unsigned delta = (unsigned)(((ty * IW + tx) * a.D * a.x_ps) * 2);
#pragma unroll
    for (int i = 0; i < BM / 64; ++i) {
      const unsigned off = ((r_mask[i] >> tapbit) & 1u) ? r_off0[i] + delta : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (__attribute__((address_space(3))) void*)&As[st][(i * 64 + wave * 16) * ROWD], 16,
                                               off, cc * (BKD * 2), 0, 0);
    }
    const int wsoff = (((ph.kh0 + ty * a.kst) * a.KW + (ph.kw0 + tx * a.kst)) * cpt + cc) * wtile;
#pragma unroll
    for (int i = 0; i < BN / 64; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wres, (__attribute__((address_space(3))) void*)&Bs[st][(i * 64 + wave * 16) * ROWD], 16,
                                               woff[i], wsoff, 0, 0);
    nx_cc += 1;
    const bool wc = nx_cc == cpt;
    nx_cc = wc ? 0 : nx_cc;
    nx_tx += wc ? 1 : 0;
    const bool wx = nx_tx == ph.Tx;
    nx_tx = wx ? 0 : nx_tx;
    nx_ty += wx ? 1 : 0;
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int l31 = lane & 31, lh = lane >> 5;
  const int fsw = swz_d(l31);   // == swz(row) for every fragment row: the block offsets are multiples of 32
  // fragment byte offsets inside a ring slot (kk = 0; kk = 1 is chunk ^ 2, i.e. byte offset ^ 32)
  const unsigned a_frag = ((wm * WTM + l31) * ROWD + ((lh ^ fsw) * 8)) * 2;
  const unsigned b_frag = ((wn * WTN + l31) * ROWD + ((lh ^ fsw) * 8)) * 2;
  const unsigned as_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&As[0][0];
  const unsigned bs_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&Bs[0][0];

  // prologue: NS - 1 steps in flight
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nt) request(s);

  int st = 0;   // ring slot of step k
  for (int k = 0; k < nt; ++k) {
    // the oldest request (step k) has landed when at most the NDMA * (steps still in flight behind it) newer
    // instructions are outstanding; near the end fewer are in flight: wait for all
    if (k + NS - 1 <= nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * (NS - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");   // every wave's share of step k is in LDS; slot (k - 1) % NS is free
    if (k + NS - 1 < nt) request(st == 0 ? NS - 1 : st - 1);
    // Fragment reads as inline assembly: with ordinary LDS loads the compiler protects them against the DMA writes still
    // in flight ("may alias") with s_waitcnt vmcnt(0), i.e. it would wait for the requests issued a moment ago and undo
    // the ring.  The ordering that matters is the explicit vmcnt wait + barrier above.
    const unsigned aa0 = as_base + st * (BM * ROWD * 2) + a_frag, aa1 = aa0 ^ 32u;
    const unsigned ba0 = bs_base + st * (BN * ROWD * 2) + b_frag, ba1 = ba0 ^ 32u;
    bf16x8 a0[4], a1[4], b0[2], b1[2];
#define EG_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
    EG_DSR(a0[0], aa0, 0); EG_DSR(b0[0], ba0, 0); EG_DSR(a0[1], aa0, 2048); EG_DSR(b0[1], ba0, 2048);
    EG_DSR(a0[2], aa0, 4096); EG_DSR(a0[3], aa0, 6144);
    EG_DSR(a1[0], aa1, 0); EG_DSR(b1[0], ba1, 0); EG_DSR(a1[1], aa1, 2048); EG_DSR(b1[1], ba1, 2048);
    EG_DSR(a1[2], aa1, 4096); EG_DSR(a1[3], aa1, 6144);
#undef EG_DSR
    asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a0[0]), "+v"(a0[1]), "+v"(a0[2]), "+v"(a0[3]), "+v"(b0[0]), "+v"(b0[1]));
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0[j], a0[i], acc[i][j], 0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a1[0]), "+v"(a1[1]), "+v"(a1[2]), "+v"(a1[3]), "+v"(b1[0]), "+v"(b1[1]));
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1[j], a1[i], acc[i][j], 0, 0, 0);
    st = st + 1 == NS ? 0 : st + 1;
  }

  // D^T layout of the 32x32 accumulator: lane l holds pixel row l & 31 and, in registers 4g .. 4g+3, the four consecutive
  // output channels 8g + 4 (l >> 5) .. +3 of its 32-channel block: one 16-byte store per register quad
#pragma unroll
  for (int i = 0; i < 4; ++i) {
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

bool conv_dma_eligible(const ConvArgs& a) {
  static const bool off = getenv("EG_NO_DMA") != nullptr;
  if (off || a.limbs != 1 || !a.wgt_bf || !a.x_bf || a.reflect || a.shuf) return false;
  if (a.C % BKD != 0 || a.N < 128 || a.N % 4 != 0 || a.x_ps % 8 != 0 || a.y_ps % 4 != 0) return false;
  if (!a.x_bytes || !a.wbf_bytes) return false;
  if ((((uintptr_t)a.x_bf | (uintptr_t)a.wgt_bf | (uintptr_t)a.y) & 15) != 0 || (a.bias && (((uintptr_t)a.bias) & 15) != 0)) return false;
  for (int p = 0; p < a.nphase; ++p)
    if (a.ph[p].Ty * a.ph[p].Tx > 32) return false;
  return true;
}

const char* launch_conv_gemm_dma(const ConvArgs& a, int maxM, hipStream_t s) {
  const int zs = a.ksplit > 1 ? a.ksplit : 1;
  dim3 grid((maxM + 255) / 256, (a.N + 127) / 128, a.nphase * zs);
  if (a.ksplit > 1) hipLaunchKernelGGL((conv_gemm_dma_kernel<true>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((conv_gemm_dma_kernel<false>), grid, dim3(256), 0, s, a);
  return a.ksplit > 1 ? "conv_gemm_bf16dma<256,128,splitk>" : "conv_gemm_bf16dma<256,128>";
}

}  // namespace eg
