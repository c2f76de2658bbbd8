// Probe of gfx950's ds_read_b64_tr_b16: which LDS elements does lane l receive, given per-lane addresses?
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe/tr_probe.hip -o gpurun_out/tr_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(unsigned* out, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;   // element value = its index
  __syncthreads();
  const int l = threadIdx.x;
  // mode 0: a dense row-major [4][16] block per 16-lane group at base g*64 elements: lane (r = i/4, q = i%4) -> &M[r][4q]
  // mode 1: rows 64 elements (128 B) apart: lane (r, q) -> base + r*64 + 4q elements, group g at + g*16 elements
  // mode 2: like mode 1 but rows permuted per lane (row = 3 - r) to see whether the ROW comes from the lane's own address
  const int g = l >> 4, i = l & 15, r = i >> 2, q = i & 3;
  int elem;
  if (mode == 0) elem = g * 64 + r * 16 + q * 4;
  else if (mode == 1) elem = r * 64 + g * 16 + q * 4;
  else elem = (3 - r) * 64 + g * 16 + q * 4;
  const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&lds[elem];
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[l * 4 + 0] = v[0] & 0xffff;
  out[l * 4 + 1] = v[0] >> 16;
  out[l * 4 + 2] = v[1] & 0xffff;
  out[l * 4 + 3] = v[1] >> 16;
}
int main() {
  unsigned* d;
  hipMalloc(&d, 256 * 4);
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
    unsigned h[256];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4u %4u %4u %4u\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  }
  return 0;
}
