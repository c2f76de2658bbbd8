// Host-side helpers of the checkpoint path: CRC-32C (Castagnoli), the checksum of TensorFlow's tensor
// bundle format (tensorflow/core/lib/hash/crc32c.h), hardware-accelerated through SSE4.2.
#include <nmmintrin.h>

#include <cstddef>
#include <cstdint>

extern "C" unsigned int eg_crc32c(unsigned int crc, const void* data, unsigned long long n) {
  const unsigned char* p = static_cast<const unsigned char*>(data);
  uint64_t c = crc ^ 0xffffffffu;
  while (n && (reinterpret_cast<uintptr_t>(p) & 7)) {
    c = _mm_crc32_u8(static_cast<uint32_t>(c), *p++);
    --n;
  }
  while (n >= 8) {
    c = _mm_crc32_u64(c, *reinterpret_cast<const uint64_t*>(p));
    p += 8;
    n -= 8;
  }
  while (n--) c = _mm_crc32_u8(static_cast<uint32_t>(c), *p++);
  return static_cast<uint32_t>(c) ^ 0xffffffffu;
}
