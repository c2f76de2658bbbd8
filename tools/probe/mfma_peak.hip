// What the matrix pipe of THIS machine sustains on v_mfma_f32_32x32x16_bf16 with nothing else in the way: W waves per SIMD
// each issuing ITER x 16 MFMAs on 4 independent accumulators, one block per CU, no memory traffic.  Prints TFLOP/s and the
// shader clock implied by s_memtime (100 MHz) against clock64()... the rate is what the roofline's "peak" means in practice.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe/mfma_peak.hip -o gpurun_out/mfma_peak ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void spin(float* out, int iters, unsigned long long* cyc) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  // operands: benign constants (iters > 0) or pseudo-random normal-range bf16 values with random signs and mantissas
  // (iters < 0): the data toggling sets the matrix pipe's power, and with it the clock the pipe is allowed
  const bool rnd = iters < 0;
  iters = rnd ? -iters : iters;
  unsigned h = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
  for (int i = 0; i < 8; ++i) {
    h = h * 1664525u + 1013904223u;
    const unsigned short ra = (unsigned short)(0x3f00u | ((h >> 9) & 0x80ffu)), rb = (unsigned short)(0x3f00u | ((h >> 17) & 0x80ffu));
    a[i] = rnd ? __builtin_bit_cast(__bf16, ra) : (__bf16)(float)(threadIdx.x & 3);
    b[i] = rnd ? __builtin_bit_cast(__bf16, rb) : (__bf16)1.0f;
  }
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][threadIdx.x & 15];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC>
static void run(int threads, int blocks, int iters, bool rnd) {
  float* d; unsigned long long* c;
  hipMalloc(&d, (size_t)blocks * threads * 4); hipMalloc(&c, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(spin<NACC>, dim3(blocks), dim3(threads), 0, 0, d, 1000, c);
  hipDeviceSynchronize();
  float best = 1e30f; unsigned long long cy = 0;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(spin<NACC>, dim3(blocks), dim3(threads), 0, 0, d, rnd ? -iters : iters, c);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) { best = ms; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost); }
  }
  const double flops = 2.0 * 32 * 32 * 16 * 16.0 * iters * (threads / 64) * blocks;
  printf("%s waves/CU %2d blocks %4d acc %d: %8.3f ms  %8.1f TFLOP/s   (wave 0: %llu counter ticks for %d MFMAs -> %.2f ticks per MFMA)\n", rnd ? "random  " : "constant", threads / 64, blocks, NACC,
         best, flops / best / 1e9, cy, iters * 16, (double)cy / (iters * 16.0));
  hipFree(d); hipFree(c);
}
int main() {
  for (int pass = 0; pass < 2; ++pass) {
    const bool rnd = pass == 1;
    run<4>(256, 256, 40000, rnd);    // 1 wave per SIMD
    run<4>(512, 256, 40000, rnd);    // 2 waves per SIMD
    run<4>(512, 256, 400000, rnd);   // ... for 170 ms: long enough for the power management to settle
  }
  return 0;
}
