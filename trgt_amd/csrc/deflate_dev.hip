// trgt_amd/csrc/deflate_dev.hip -- raw DEFLATE (RFC 1951) ENCODING of many independent blocks on the device: the BGZF blocks of the
// spanning-reads BAM (round 5).
//
// Replaces, for the writers of SURVEY.md 8(f) row 4, what the reference does inside htslib's bgzf_write (rust-htslib bam::Writer ->
// htslib -> zlib deflate; src/trgt/writers/write_bam.rs:72-144 reaches it record by record): every BGZF block (at most 0xFF00 bytes of
// payload here) is a DEFLATE stream of its own, so the blocks of a chunk of loci are a few thousand independent streams -- and on the
// 16-CPU quota of the GPU boxes zlib at level 6 is what bounds the writer (and with the inflate of the ingestion the whole BAM -> VCF
// pipeline).
//
// One wave per block, every LANE encodes its own slice (1/64 of the block, about 1 KB): LZ77 with a private 256-entry hash table in LDS
// (matches inside the lane's slice only: what a spanning read repeats -- its tandem repeat, runs of equal qualities -- repeats within
// tens of bytes), greedy, fixed Huffman codes (BTYPE 01: no tree to build), bits gathered in a 64-bit accumulator and written as
// 32-bit words to the wave's scratch.  Then the lanes' bit strings are laid end to end: a prefix sum of their lengths gives every lane
// its bit offset, and it ORs its words, shifted, into the cleared output (atomicOr: neighbours share a word at the seams).  The result
// is ONE final fixed-Huffman block per BGZF block: any inflate reads it.  It is deterministic (no lane sees another lane's table) and it
// is NOT zlib's byte stream -- the file is a valid BAM with the same records, as with another zlib level; the ratio is about that of
// zlib's level 1.  A block whose encoding does not fit the room the caller gives (data that does not compress) is DECLINED (length 0):
// the caller deflates it with zlib.  The BGZF header, the CRC-32 and ISIZE stay with the host.
#include "common.hpp"

namespace trgt {
namespace defl {

constexpr int TAB = 256;                    // hash entries per lane
constexpr uint32_t WINDOW_SLICES = 1;       // slices before its own that a lane's window takes in (hashed first, see the kernel; 3 gave 13.0 MB against 13.1)
constexpr uint32_t SLICE_MAX = 1024;        // bytes per lane at most (64 KB per block)
constexpr uint32_t WORDS_MAX = 296;         // 32-bit words a lane may produce: 1024 literals of 9 bits = 288 words, + the flush

struct BlockDesc { uint64_t src_off, dst_off; uint32_t src_len, dst_cap; };

__device__ __forceinline__ uint32_t rev(uint32_t v, int n) { return __builtin_bitreverse32(v) >> (32 - n); }

struct BitOut {
  uint64_t acc = 0; uint32_t n = 0, words = 0;
  uint32_t* out; uint32_t stride;
  __device__ __forceinline__ void put(uint32_t v, uint32_t bits) {  // LSB-first packing (RFC 1951 3.1.1); bits <= 24
    acc |= (uint64_t)v << n; n += bits;
    if (n >= 32) { out[(size_t)words * stride] = (uint32_t)acc; ++words; acc >>= 32; n -= 32; }
  }
  __device__ __forceinline__ uint32_t total_bits() const { return 32u * words + n; }
  __device__ __forceinline__ void flush() { if (n) { out[(size_t)words * stride] = (uint32_t)acc; } }
};

// fixed Huffman code of a literal / length symbol (RFC 1951 3.2.6), already bit-reversed for LSB-first packing
__device__ __forceinline__ void put_litlen(BitOut& b, uint32_t sym) {
  if (sym < 144) b.put(rev(0x30 + sym, 8), 8);
  else if (sym < 256) b.put(rev(0x190 + (sym - 144), 9), 9);
  else if (sym < 280) b.put(rev(sym - 256, 7), 7);
  else b.put(rev(0xC0 + (sym - 280), 8), 8);
}
__device__ __forceinline__ void put_match(BitOut& b, uint32_t len, uint32_t dist) {  // 3 <= len <= 258, 1 <= dist <= 32768
  // length code (3.2.5): 257..264 one length each, then groups of 4 codes with 1, 2, ... 5 extra bits, 285 = 258
  uint32_t sym, extra, ebits;
  if (len == 258) { sym = 285; extra = 0; ebits = 0; }
  else if (len <= 10) { sym = 254 + len; extra = 0; ebits = 0; }
  else {
    const uint32_t l = len - 3;                       // 8 .. 254
    const uint32_t e = 29u - (uint32_t)__builtin_clz(l);  // extra bits: floor(log2 l) - 2
    sym = 261 + 4 * e + ((l >> e) - 4);              // (l >> e) is 4 .. 7: the code inside its group of four
    extra = l & ((1u << e) - 1u); ebits = e;
  }
  put_litlen(b, sym);
  if (ebits) b.put(extra, ebits);
  // distance code: 0..3 one distance each, then pairs of codes with 1, 2, ... 13 extra bits; 5-bit fixed codes
  const uint32_t d = dist - 1;
  uint32_t dc, dextra, dbits;
  if (d < 4) { dc = d; dextra = 0; dbits = 0; }
  else {
    const uint32_t e = 30u - (uint32_t)__builtin_clz(d);  // extra bits: floor(log2 d) - 1
    dc = 2 * e + 2 + ((d >> e) & 1u);
    dextra = d & ((1u << e) - 1u); dbits = e;
  }
  b.put(rev(dc, 5), 5);
  if (dbits) b.put(dextra, dbits);
}

__device__ __forceinline__ uint32_t load32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

__global__ void __launch_bounds__(64) deflate_blocks_kernel(const uint8_t* __restrict__ src, const BlockDesc* __restrict__ descs, uint32_t n_blocks, uint8_t* __restrict__ dst,
                                                            uint32_t* __restrict__ dst_len, uint32_t* __restrict__ scratch, unsigned int* __restrict__ counter) {
  __shared__ uint16_t tab[64 * TAB];
  __shared__ uint32_t s_block;
  const int lane = (int)threadIdx.x;
  uint32_t* const my_words = scratch + (size_t)blockIdx.x * 64 * WORDS_MAX + lane;  // word k of lane l at [k * 64 + l]
  for (;;) {
    if (lane == 0) s_block = atomicAdd(counter, 1u);
    __syncthreads();
    const uint32_t b = s_block;
    __syncthreads();
    if (b >= n_blocks) break;
    const BlockDesc d = descs[b];
    const uint8_t* __restrict__ in = src + d.src_off;
    const uint32_t n = d.src_len;
    const uint32_t slice = (n + 63u) / 64u;
    const uint32_t s0 = min(n, (uint32_t)lane * slice), s1 = min(n, s0 + slice);
    bool fits = n <= 64u * SLICE_MAX;
    uint16_t* const t = tab + lane * TAB;
    for (int i = 0; i < TAB; ++i) t[i] = 0xFFFFu;
    BitOut bo; bo.out = my_words; bo.stride = 64;
    // the lane's window opens one slice before its own: the positions of the slice before are hashed first (nothing is emitted for
    // them), so that a match may reach back into bytes another lane encodes -- they precede this lane's in the stream all the same
    const uint32_t win0 = s0 > WINDOW_SLICES * slice ? s0 - WINDOW_SLICES * slice : 0u;
    if (fits) {
      for (uint32_t p = win0; p < s0 && p + 4 <= n; ++p) t[(load32(in + p) * 2654435761u) >> 24] = (uint16_t)(p - win0);
      uint32_t p = s0;
      while (p < s1) {
        uint32_t len = 0, dist = 0;
        if (p + 4 <= s1) {
          const uint32_t w = load32(in + p);
          const uint32_t h = (w * 2654435761u) >> 24;
          const uint32_t c = t[h];
          t[h] = (uint16_t)(p - win0);
          if (c != 0xFFFFu) {
            const uint32_t q = win0 + c;
            if (load32(in + q) == w) {
              const uint32_t lim = min(258u, s1 - p);
              len = 4;
              while (len + 4 <= lim && load32(in + q + len) == load32(in + p + len)) len += 4;
              while (len < lim && in[q + len] == in[p + len]) ++len;
              dist = p - q;
            }
          }
        }
        if (len >= 4) { put_match(bo, len, dist); p += len; }
        else { put_litlen(bo, in[p]); ++p; }
      }
    }
    bo.flush();
    // ---- the lanes' bit strings end to end: header (3 bits) | lane 0 | lane 1 | ... | end of block (7 bits)
    const uint32_t my_bits = fits ? bo.total_bits() : 0u;
    uint32_t off = my_bits;  // inclusive prefix sum over the lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)off, o); if (lane >= o) off += v; }
    const uint32_t total_bits = 3u + (uint32_t)__shfl((int)off, 63) + 7u;
    const uint32_t my_off = 3u + off - my_bits;
    const uint32_t out_bytes = (total_bits + 7u) / 8u;
    uint8_t* const out = dst + d.dst_off;
    const bool ok = fits && out_bytes <= d.dst_cap && (d.dst_off & 3u) == 0u;
    if (ok) {
      uint32_t* const ow = reinterpret_cast<uint32_t*>(out);
      const uint32_t n_ow = (out_bytes + 3u) / 4u + 1u;  // (the caller leaves room for the word the last OR may touch: dst_cap + 8 bytes are its)
      for (uint32_t i = (uint32_t)lane; i < n_ow; i += 64) ow[i] = 0u;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the zeros are at the L2 the atomics below work in: one wave, no other reader -- not an agent-scope fence, which writes back and invalidates the XCD's L2)
      __syncthreads();
      if (lane == 0) atomicOr(ow, 3u);  // BFINAL = 1, BTYPE = 01 (fixed Huffman codes)
      const uint32_t n_words = (my_bits + 31u) / 32u;
      const uint32_t sh = my_off & 31u, w0 = my_off >> 5;
      for (uint32_t k = 0; k < n_words; ++k) {
        uint32_t v = my_words[(size_t)k * 64];
        if (k == n_words - 1 && (my_bits & 31u)) v &= (1u << (my_bits & 31u)) - 1u;
        if (v) {
          atomicOr(ow + w0 + k, v << sh);
          if (sh) { const uint32_t hi = v >> (32u - sh); if (hi) atomicOr(ow + w0 + k + 1, hi); }
        }
      }
    }
    if (lane == 0) dst_len[b] = ok ? out_bytes : 0u;
    __syncthreads();
  }
}

}  // namespace defl

int deflate_blocks_device(trgt_hip_ctx* c, int64_t n, const uint8_t* src, uint64_t src_bytes, const defl::BlockDesc* descs, uint8_t* dst, uint64_t dst_bytes, uint32_t* dst_len) {
  if (n <= 0) return TRGT_OK;
  TRGT_HIP_TRY(c, hipSetDevice(c->device));
  void *d_src = nullptr, *d_desc = nullptr, *d_dst = nullptr, *d_len = nullptr, *d_counter = nullptr, *d_scratch = nullptr;
  int rc;
  const unsigned grid = (unsigned)std::min<int64_t>(n, (int64_t)c->num_cus * 4);
  if ((rc = dev_get(c, S_INF_SRC, (size_t)src_bytes + 64, &d_src)) || (rc = dev_get(c, S_INF_DESC, (size_t)n * sizeof(defl::BlockDesc), &d_desc)) ||
      (rc = dev_get(c, S_INF_DST, (size_t)dst_bytes + 64, &d_dst)) || (rc = dev_get(c, S_INF_STATUS, (size_t)n * 4 + 16, &d_len)) ||
      (rc = dev_get(c, S_INF_COUNTER, 16, &d_counter)) || (rc = dev_get(c, S_DEFL_SCRATCH, (size_t)grid * 64 * defl::WORDS_MAX * 4, &d_scratch)))
    return rc;
  TRGT_HIP_TRY(c, hipMemcpyAsync(d_src, src, (size_t)src_bytes, hipMemcpyHostToDevice, c->stream));
  TRGT_HIP_TRY(c, hipMemcpyAsync(d_desc, descs, (size_t)n * sizeof(defl::BlockDesc), hipMemcpyHostToDevice, c->stream));
  TRGT_HIP_TRY(c, hipMemsetAsync(d_counter, 0, 16, c->stream));
  hipLaunchKernelGGL(defl::deflate_blocks_kernel, dim3(grid), dim3(64), 0, c->stream, (const uint8_t*)d_src, (const defl::BlockDesc*)d_desc, (uint32_t)n, (uint8_t*)d_dst,
                     (uint32_t*)d_len, (uint32_t*)d_scratch, (unsigned int*)d_counter);
  TRGT_HIP_TRY(c, hipGetLastError());
  // the lengths first, then only what the streams fill of their slots (the slots are 64 KB apart and a third full at most --
  // copying them whole moved three to four times the compressed bytes back over PCIe): one strided copy when the slots are evenly
  // spaced (the writer's are), the whole region otherwise
  TRGT_HIP_TRY(c, hipMemcpyAsync(dst_len, d_len, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  TRGT_HIP_TRY(c, trgt::stream_wait(c, c->stream));
  uint32_t widest = 0;
  for (int64_t b = 0; b < n; ++b) widest = std::max(widest, dst_len[b]);
  const uint64_t pitch = n > 1 ? descs[1].dst_off - descs[0].dst_off : 0;
  bool even = n > 1 && descs[0].dst_off == 0 && pitch >= widest;
  for (int64_t b = 1; even && b < n; ++b) even = descs[b].dst_off == (uint64_t)b * pitch;
  if (even && widest > 0) TRGT_HIP_TRY(c, hipMemcpy2DAsync(dst, (size_t)pitch, d_dst, (size_t)pitch, ((size_t)widest + 3) & ~(size_t)3, (size_t)n, hipMemcpyDeviceToHost, c->stream));
  else if (widest > 0) TRGT_HIP_TRY(c, hipMemcpyAsync(dst, d_dst, (size_t)dst_bytes, hipMemcpyDeviceToHost, c->stream));
  TRGT_HIP_TRY(c, trgt::stream_wait(c, c->stream));
  return TRGT_OK;
}

}  // namespace trgt

// include/trgt_hip.h: "device-side BGZF deflate"
extern "C" int trgt_deflate_blocks(trgt_hip_ctx* c, int64_t n_blocks, const uint8_t* src, const uint64_t* src_off, const uint32_t* src_len, uint8_t* dst,
                                   const uint64_t* dst_off, const uint32_t* dst_cap, uint32_t* dst_len) {
  if (!c) return TRGT_ERR_INVALID;
  if (n_blocks < 0 || (n_blocks > 0 && (!src || !src_off || !src_len || !dst || !dst_off || !dst_cap || !dst_len))) return trgt::fail(c, TRGT_ERR_INVALID, "trgt_deflate_blocks: null argument");
  try {
    std::vector<trgt::defl::BlockDesc> d((size_t)n_blocks);
    uint64_t sb = 0, db = 0;
    for (int64_t b = 0; b < n_blocks; ++b) {
      if (src_len[b] > 65536u) return trgt::fail(c, TRGT_ERR_INVALID, "trgt_deflate_blocks: block %lld has %u bytes (a BGZF block holds at most 65536)", (long long)b, src_len[b]);
      if (dst_off[b] & 3u) return trgt::fail(c, TRGT_ERR_INVALID, "trgt_deflate_blocks: dst_off[%lld] is not a multiple of 4", (long long)b);
      d[(size_t)b] = trgt::defl::BlockDesc{src_off[b], dst_off[b], src_len[b], dst_cap[b]};
      sb = std::max<uint64_t>(sb, src_off[b] + src_len[b]); db = std::max<uint64_t>(db, dst_off[b] + dst_cap[b] + 8);
    }
    return trgt::deflate_blocks_device(c, n_blocks, src, sb, d.data(), dst, db, dst_len);
  } catch (const std::bad_alloc&) { return trgt::fail(c, TRGT_ERR_NOMEM, "out of host memory"); }
}
