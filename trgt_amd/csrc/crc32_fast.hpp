// trgt_amd/csrc/crc32_fast.hpp -- CRC-32 (IEEE 802.3, the BGZF footer's) of a host buffer by carry-less multiplication: four 128-bit lanes
// folded per 64 bytes, then Barrett reduction (the published PCLMULQDQ scheme of Gopal et al., "Fast CRC Computation for Generic
// Polynomials Using PCLMULQDQ Instruction"); zlib's crc32() takes the tail and the whole job on a CPU without the instruction.  htslib
// checks this CRC on every block it reads and writes one on every block it writes (bgzf_read_block / bgzf_write behind
// bam::IndexedReader and bam::Writer, src/trgt/workflows/tr.rs:268-305, src/trgt/writers/write_bam.rs:72-144): at 1 GB/s per core,
// zlib's table-driven loop was a third of the host ingestion's time once the check was there (round 6); this runs at 6 GB/s.
#pragma once
#include <immintrin.h>
#include <zlib.h>

#include <cstddef>
#include <cstdint>

namespace trgt {

__attribute__((target("pclmul,sse4.1"))) inline uint32_t crc32_clmul_bulk(const uint8_t* buf, size_t len, uint32_t crc) {  // len >= 64 and a multiple of 16; crc: the register as it stands (not inverted)
  alignas(16) static const uint64_t k1k2[] = {0x0154442bd4ull, 0x01c6e41596ull};
  alignas(16) static const uint64_t k3k4[] = {0x01751997d0ull, 0x00ccaa009eull};
  alignas(16) static const uint64_t k5k0[] = {0x0163cd6124ull, 0x0000000000ull};
  alignas(16) static const uint64_t poly[] = {0x01db710641ull, 0x01f7011641ull};
  __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
  x1 = _mm_loadu_si128((const __m128i*)(buf + 0x00)); x2 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
  x3 = _mm_loadu_si128((const __m128i*)(buf + 0x20)); x4 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
  x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
  x0 = _mm_load_si128((const __m128i*)k1k2);
  buf += 64; len -= 64;
  while (len >= 64) {
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x6 = _mm_clmulepi64_si128(x2, x0, 0x00); x7 = _mm_clmulepi64_si128(x3, x0, 0x00); x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11); x3 = _mm_clmulepi64_si128(x3, x0, 0x11); x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
    y5 = _mm_loadu_si128((const __m128i*)(buf + 0x00)); y6 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
    y7 = _mm_loadu_si128((const __m128i*)(buf + 0x20)); y8 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5); x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
    x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7); x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
    buf += 64; len -= 64;
  }
  x0 = _mm_load_si128((const __m128i*)k3k4);  // the four lanes into one
  x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
  x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
  x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
  while (len >= 16) {
    x2 = _mm_loadu_si128((const __m128i*)buf);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    buf += 16; len -= 16;
  }
  x2 = _mm_clmulepi64_si128(x1, x0, 0x10);  // 128 -> 64 bits
  x3 = _mm_setr_epi32(~0, 0, ~0, 0);
  x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), x2);
  x0 = _mm_loadl_epi64((const __m128i*)k5k0);
  x2 = _mm_srli_si128(x1, 4); x1 = _mm_and_si128(x1, x3); x1 = _mm_xor_si128(_mm_clmulepi64_si128(x1, x0, 0x00), x2);
  x0 = _mm_load_si128((const __m128i*)poly);  // Barrett reduction to 32 bits
  x2 = _mm_and_si128(x1, x3); x2 = _mm_clmulepi64_si128(x2, x0, 0x10); x2 = _mm_and_si128(x2, x3); x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
  x1 = _mm_xor_si128(x1, x2);
  return (uint32_t)_mm_extract_epi32(x1, 1);
}

// crc32(0, p, n) of zlib
inline uint32_t crc32_fast(const uint8_t* p, size_t n) {
  static const bool have = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
  if (!have || n < 64) return (uint32_t)crc32(crc32(0L, Z_NULL, 0), p, (uInt)n);
  const size_t bulk = n & ~(size_t)15;
  uint32_t c = ~crc32_clmul_bulk(p, bulk, 0xFFFFFFFFu);
  if (n > bulk) c = (uint32_t)crc32(c, p + bulk, (uInt)(n - bulk));
  return c;
}

}  // namespace trgt
