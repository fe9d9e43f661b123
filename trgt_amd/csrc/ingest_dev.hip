// trgt_amd/csrc/ingest_dev.hip -- device-side read ingestion (round 6): behind the device inflate (inflate_dev.hip) the
// inflated BAM bytes of a batch of loci stay in HBM and are turned into the arrays of trgt_locus_batch_in by kernels:
//   crc32_blocks_kernel   CRC-32 of every inflated BGZF block against its footer (what htslib's bgzf_read_block checks for
//                         bam::IndexedReader, src/trgt/workflows/tr.rs:268-305)
//   walk_kernel           extract_reads (tr.rs:268-361): per locus the record chain of its .bai chunks, the stop at the first record
//                         beyond the window, the secondary / supplementary and rq filters, the list of the kept records -- with the
//                         reservoir of 3 * max_depth reads and its random replacements (StdRng::seed_from_u64(42)) for deeper loci
//   read_sizes_kernel     HiFiRead::from_hts_rec (reads/read.rs:98-141) + clip_to_region (reads/clip_region.rs:19-184) as sizes: the
//                         CIGAR of a read as prefix sums over a wave (where the clip window cuts it, which operation is split), the
//                         mismatch offsets of snp.rs:51-79 counted, the aux tags located
//   read_meth_kernel      get_meth (read.rs:55-96): the MM / ML tags against the CpGs of the stored sequence (bit masks + ranks in LDS)
//   scan_kernel           exclusive sums of the sizes -> where every read's pieces go
//   read_fill_kernel      the clipped bases (4-bit codes -> ASCII, and the 4-bit form once more), qualities, names, clipped CIGAR,
//                         mismatch offsets, methylation values and the per-read scalars
// One wave per locus / per read; a wave's lanes share the serial work of a record by ballots and scans, not by diverging.  Anything a
// kernel does not take (a block it cannot inflate, a record that runs out of its range, MM strings beyond the LDS caps) sends the WHOLE
// call back to the host path of ingest.hip, which then produces the data or the error message.
#include <zlib.h>

#include <chrono>
#include <cstring>

#include "common.hpp"
#include "crc32_fast.hpp"
#include "ingest_dev.hpp"

namespace trgt {
namespace ingd {

// ------------------------------------------------------------------------------------------------ small device helpers
__device__ __forceinline__ uint32_t ld32(const uint8_t* p) {  // little-endian word at any alignment (two aligned loads; the buffers carry slack behind their end)
  const uintptr_t a = (uintptr_t)p;
  const uint32_t* q = (const uint32_t*)(a & ~(uintptr_t)3);
  const uint32_t sh = (uint32_t)(a & 3u) * 8u;
  const uint32_t lo = q[0];
  if (!sh) return lo;
  return (lo >> sh) | (q[1] << (32u - sh));
}
__device__ __forceinline__ uint32_t ld16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
__device__ __forceinline__ int lane_id() { return (int)threadIdx.x & 63; }
__device__ __forceinline__ uint64_t ballot64(bool p) { return __ballot(p); }
__device__ __forceinline__ uint64_t lanes_below(int lane) { return lane ? (~0ull >> (64 - lane)) : 0ull; }
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t rfl64(uint64_t v) { return ((uint64_t)rfl((uint32_t)(v >> 32)) << 32) | rfl((uint32_t)v); }
__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src) { return ((uint64_t)(uint32_t)__shfl((int)(v >> 32), src) << 32) | (uint32_t)__shfl((int)(uint32_t)v, src); }
// inclusive sum over the wave
__device__ __forceinline__ uint64_t wave_incl_sum(uint64_t v) {
  const int lane = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const uint64_t o = shfl64(v, lane - d < 0 ? lane : lane - d); if (lane >= d) v += o; }
  return v;
}
__device__ __forceinline__ uint64_t wave_sum(uint64_t v) { return shfl64(wave_incl_sum(v), 63); }

constexpr int OP_M = 0, OP_I = 1, OP_D = 2, OP_N = 3, OP_S = 4, OP_EQ = 7, OP_X = 8;
__device__ __forceinline__ uint32_t ref_len(uint32_t op) { const uint32_t c = op & 0xFu; return (c == OP_M || c == OP_D || c == OP_N || c == OP_EQ || c == OP_X) ? (op >> 4) : 0u; }
__device__ __forceinline__ uint32_t qry_len(uint32_t op) { const uint32_t c = op & 0xFu; return (c == OP_M || c == OP_I || c == OP_S || c == OP_EQ || c == OP_X) ? (op >> 4) : 0u; }

// The fields of a BAM record body (SAM spec 4.2), `rec` pointing behind block_size
struct RecHead { int32_t ref_id, pos, l_seq; uint32_t l_rn, mapq, n_cig, flag; uint64_t o_cig, o_seq, o_qual, o_aux; };
__device__ __forceinline__ RecHead rec_head(const uint8_t* rec) {
  RecHead h;
  h.ref_id = (int32_t)ld32(rec); h.pos = (int32_t)ld32(rec + 4);
  const uint32_t w2 = ld32(rec + 8), w3 = ld32(rec + 12);
  h.l_rn = w2 & 0xFFu; h.mapq = (w2 >> 8) & 0xFFu; h.n_cig = w3 & 0xFFFFu; h.flag = w3 >> 16; h.l_seq = (int32_t)ld32(rec + 16);
  h.o_cig = 32ull + h.l_rn; h.o_seq = h.o_cig + 4ull * h.n_cig; h.o_qual = h.o_seq + ((uint64_t)(uint32_t)h.l_seq + 1) / 2; h.o_aux = h.o_qual + (uint64_t)(uint32_t)h.l_seq;
  return h;
}

// find_aux of ingest.hip for several tags in one pass: offsets (of the type byte, from `rec`) of the first "rq", "HP", "MM", "Mm", "ML", "Ml"
// fields; 0 = absent.  The pass ends where find_aux would give up (an unknown type, a truncated B array).
struct AuxAt { uint32_t rq, hp, mm, mm2, ml, ml2; };
__device__ __forceinline__ int aux_size(uint32_t t) { return (t == 'A' || t == 'c' || t == 'C') ? 1 : (t == 's' || t == 'S') ? 2 : (t == 'i' || t == 'I' || t == 'f') ? 4 : 0; }
__device__ inline void aux_scan(const uint8_t* rec, uint64_t o_aux, uint64_t size, AuxAt& at, bool only_rq) {
  at.rq = at.hp = at.mm = at.mm2 = at.ml = at.ml2 = 0;
  uint64_t p = o_aux;
  while (p + 3 <= size) {
    const uint32_t t0 = rec[p], t1 = rec[p + 1], ty = rec[p + 2];
    const uint32_t tag = t0 | (t1 << 8);
    if (tag == ('r' | ('q' << 8))) { if (!at.rq) at.rq = (uint32_t)(p + 2); if (only_rq) return; }
    else if (tag == ('H' | ('P' << 8))) { if (!at.hp) at.hp = (uint32_t)(p + 2); }
    else if (tag == ('M' | ('M' << 8))) { if (!at.mm) at.mm = (uint32_t)(p + 2); }
    else if (tag == ('M' | ('m' << 8))) { if (!at.mm2) at.mm2 = (uint32_t)(p + 2); }
    else if (tag == ('M' | ('L' << 8))) { if (!at.ml) at.ml = (uint32_t)(p + 2); }
    else if (tag == ('M' | ('l' << 8))) { if (!at.ml2) at.ml2 = (uint32_t)(p + 2); }
    p += 3;
    if (const int s = aux_size(ty)) p += (uint64_t)s;
    else if (ty == 'Z' || ty == 'H') {
      while (p < size) {  // to the NUL, a word at a time where the bytes are there
        if (p + 4 <= size) { const uint32_t w = ld32(rec + p); if (!((w - 0x01010101u) & ~w & 0x80808080u)) { p += 4; continue; } }
        if (!rec[p]) break;
        ++p;
      }
      ++p;
    } else if (ty == 'B') {
      if (p + 5 > size) return;
      const int s2 = aux_size(rec[p]); const uint32_t n = ld32(rec + p + 1);
      p += 5ull + (uint64_t)s2 * n;
    } else return;
  }
}

// ------------------------------------------------------------------------------------------------ CRC-32 of the inflated blocks
struct CrcTables { uint32_t t[4][256]; uint32_t shift[6][32]; };  // slicing-by-4 tables; operators "append 1024 * 2^j zero bytes"

__device__ __forceinline__ uint32_t gf2_apply(const uint32_t* m, uint32_t v) {
  uint32_t r = 0;
#pragma unroll 8
  for (int b = 0; b < 32; ++b) r ^= m[b] & (0u - ((v >> b) & 1u));
  return r;
}

// One wave per block: lane l takes the 1024 bytes that end 1024 * l bytes before the block's end (the first, shorter slice falls to the
// highest lane at work and starts from the CRC's initial register), then the lanes' remainders are combined pairwise: a remainder that
// 1024 * 2^j more bytes follow is multiplied by x^(8 * 1024 * 2^j) mod P (a 32 x 32 bit matrix).  status[b]: 1 stays 1 when the CRC
// matches, becomes 2 when it does not; 0 (declined by the inflate kernel) is left alone.
__global__ void __launch_bounds__(64) crc32_blocks_kernel(const uint8_t* __restrict__ data, const infl::BlockDesc* __restrict__ blocks, const uint32_t* __restrict__ want,
                                                          uint32_t n_blocks, const CrcTables* __restrict__ tab, uint8_t* __restrict__ status) {
  __shared__ CrcTables sh;
  const int lane = lane_id();
  for (int i = lane; i < (int)(sizeof(CrcTables) / 4); i += 64) ((uint32_t*)&sh)[i] = ((const uint32_t*)tab)[i];
  __syncthreads();
  const uint32_t b = blockIdx.x;
  if (b >= n_blocks) return;
  if (status[b] != 1) return;
  const infl::BlockDesc bd = blocks[b];
  const uint8_t* p0 = data + bd.dst_off;
  const uint32_t len = bd.dst_len;
  const int64_t end = (int64_t)len - 1024ll * lane;
  uint32_t r = 0;
  if (end > 0) {
    const int64_t beg = end > 1024 ? end - 1024 : 0;
    if (beg == 0) r = 0xFFFFFFFFu;
    const uint8_t* p = p0 + beg;
    uint32_t n = (uint32_t)(end - beg);
    while (n & 3u) { r = sh.t[0][(r ^ *p++) & 0xFFu] ^ (r >> 8); --n; }
    for (; n; n -= 4, p += 4) {
      r ^= ld32(p);
      r = sh.t[3][r & 0xFFu] ^ sh.t[2][(r >> 8) & 0xFFu] ^ sh.t[1][(r >> 16) & 0xFFu] ^ sh.t[0][r >> 24];
    }
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const uint32_t up = (uint32_t)__shfl((int)r, (lane + (1 << j)) & 63);
    if ((lane & ((2 << j) - 1)) == 0) r ^= gf2_apply(sh.shift[j], up);
  }
  if (lane == 0 && (r ^ 0xFFFFFFFFu) != want[b]) status[b] = 2;
}

// ------------------------------------------------------------------------------------------------ the reservoir's random stream
// extract_reads (tr.rs:311-335) keeps 3 * max_depth reads; every further read replaces a random one of them with probability
// reservoir / (n + 1), drawn from rand 0.9's StdRng::seed_from_u64(42): ChaCha12 keyed by a PCG32 stream over the seed, its 32-bit words
// read in order, ranges by Canon's widening multiplication.  The restatement of ingest.hip's StdRng for one lane (the words of the
// 64-word buffer there are four consecutive blocks: the same stream block by block); unpinned like the host's (no fixture reaches it).
struct DevRng {
  uint32_t key[8]; uint32_t buf[16]; uint32_t at; uint64_t counter;
  __device__ void seed(uint64_t state) {
    const uint64_t MUL = 6364136223846793005ull, INC = 11634580027462260723ull;
    for (int i = 0; i < 8; ++i) {
      state = state * MUL + INC;
      const uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27), rot = (uint32_t)(state >> 59);
      key[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
    }
    at = 16; counter = 0;
  }
  __device__ void block() {
    const uint32_t s0[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                             (uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
    uint32_t x[16];
    for (int i = 0; i < 16; ++i) x[i] = s0[i];
    auto rotl = [](uint32_t v, int n) { return (v << n) | (v >> (32 - n)); };
    auto qr = [&](int a, int b, int c, int d) {
      x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16); x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12);
      x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8); x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7);
    };
    for (int r = 0; r < 6; ++r) { qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15); qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14); }
    for (int i = 0; i < 16; ++i) buf[i] = x[i] + s0[i];
    ++counter; at = 0;
  }
  __device__ uint32_t next_u32() { if (at >= 16) block(); return buf[at++]; }
  __device__ uint32_t range(uint32_t n) {  // 0 .. n (exclusive), n >= 1
    const uint64_t m = (uint64_t)next_u32() * n;
    uint32_t hi = (uint32_t)(m >> 32); const uint32_t lo = (uint32_t)m;
    if (lo > (uint32_t)(0u - n)) { const uint32_t hi2 = (uint32_t)(((uint64_t)next_u32() * n) >> 32); if ((uint64_t)lo + hi2 > 0xFFFFFFFFull) ++hi; }
    return hi;
  }
};

// ------------------------------------------------------------------------------------------------ the record walk of a locus
struct WalkOut { uint32_t n_kept, n_filt, status, pad; };
enum : uint32_t { WS_OK = 0, WS_ERROR = 1, WS_OVERFLOW = 2 };

__global__ void __launch_bounds__(64) walk_kernel(const uint8_t* __restrict__ infl, const LocusDesc* __restrict__ loci, const ChunkDesc* __restrict__ chunks, uint32_t n_loci,
                                                  uint32_t reservoir, double min_rq, uint64_t* __restrict__ rec_list, WalkOut* __restrict__ out) {
  __shared__ uint64_t tile[64];
  __shared__ uint32_t s_cnt, s_flags;
  __shared__ uint64_t s_next;
  __shared__ DevRng rng;  // (lane 0's; seeded when the reservoir first overflows)
  bool rng_ready = false;
  const uint32_t li = blockIdx.x;
  if (li >= n_loci) return;
  const int lane = lane_id();
  const LocusDesc L = loci[li];
  uint32_t n_reads = 0, n_filt = 0, status = WS_OK;
  bool stop = false;
  for (int c = L.chunk_begin; c < L.chunk_end && !stop && status == WS_OK; ++c) {
    const ChunkDesc ch = chunks[c];
    uint64_t p = ch.lin0;
    if (p > ch.lin_limit) { status = WS_ERROR; break; }
    bool chunk_done = false;
    while (!chunk_done && !stop && status == WS_OK) {
      // ---- lane 0 follows the chain of block_size fields: up to 64 record starts, the stop test of extract_reads on the way
      if (lane == 0) {
        uint32_t cnt = 0, flags = 0;  // bit 0: chunk done, bit 1: stop (a record of another contig or beyond the window), bit 2: error
        uint64_t q = p;
        while (cnt < 64) {
          if (q >= ch.lin1 || q >= ch.lin_limit) { flags |= 1u; break; }
          if (q + 4 > ch.lin_limit) { flags |= 4u; break; }
          const uint32_t bs = ld32(infl + q);
          if (bs < 32u || bs > (1u << 29) || q + 4 + bs > ch.lin_limit) { flags |= 4u; break; }
          const RecHead h = rec_head(infl + q + 4);
          if (h.l_seq < 0 || h.o_aux > bs) { flags |= 4u; break; }
          if (h.ref_id != L.tid || (int64_t)h.pos >= L.end) { flags |= 2u; break; }
          tile[cnt++] = q;
          q += 4ull + bs;
        }
        s_cnt = cnt; s_flags = flags; s_next = q;
      }
      __syncthreads();
      const uint32_t cnt = s_cnt, flags = s_flags;
      p = s_next;
      // ---- a lane per record: reference end from the CIGAR, flags, rq tag
      bool keep = false, filt = false;
      uint64_t q = 0;
      if ((uint32_t)lane < cnt) {
        q = tile[lane];
        const uint32_t bs = ld32(infl + q);
        const uint8_t* rec = infl + q + 4;
        const RecHead h = rec_head(rec);
        int64_t e = h.pos;
        for (uint32_t i = 0; i < h.n_cig; ++i) e += ref_len(ld32(rec + h.o_cig + 4ull * i));
        if (e == h.pos) e = (int64_t)h.pos + 1;
        if (e > L.beg && !(h.flag & (0x800u | 0x100u))) {
          AuxAt at; aux_scan(rec, h.o_aux, bs, at, true);
          double v = 1.0;
          if (at.rq && rec[at.rq] == 'f') v = (double)__uint_as_float(ld32(rec + at.rq + 1));
          if (v < min_rq) filt = true; else keep = true;
        }
      }
      const uint64_t km = ballot64(keep), fm = ballot64(filt);
      n_filt += (uint32_t)__popcll(fm);
      if (keep) { const uint32_t idx = n_reads + (uint32_t)__popcll(km & lanes_below(lane)); if (idx < reservoir) rec_list[(uint64_t)li * reservoir + idx] = q; }
      if (n_reads + (uint32_t)__popcll(km) > reservoir) {
        // the reservoir is full: every further read, in file order, replaces slot j = range(reads so far) when j < reservoir (lane 0)
        __syncthreads();
        if (!rng_ready) { if (lane == 0) rng.seed(42); rng_ready = true; }
        if (lane == 0) {
          uint32_t seen = n_reads;
          for (uint64_t rest = km; rest; rest &= rest - 1, ++seen) {
            if (seen < reservoir) continue;
            const uint32_t j = rng.range(seen);
            if (j < reservoir) rec_list[(uint64_t)li * reservoir + j] = tile[__builtin_ctzll(rest)];
          }
        }
      }
      n_reads += (uint32_t)__popcll(km);
      if (flags & 4u) status = WS_ERROR;
      if (flags & 2u) stop = true;
      if (flags & 1u) chunk_done = true;
      __syncthreads();
    }
  }
  if (lane == 0) { WalkOut w; w.n_kept = n_reads; w.n_filt = n_filt; w.status = status; w.pad = 0; out[li] = w; }
}

// ------------------------------------------------------------------------------------------------ per read: sizes
struct ReadInfo {
  uint64_t rec;            // position of the record body in the inflated bytes
  int64_t c_ref;           // reference position of the clipped alignment
  uint32_t bs, locus;
  uint32_t ok;             // clip_to_region gave a read
  uint32_t n_bases, c_qry, n_name, n_snp, n_cig;
  uint32_t i_first, n_whole, part_op, tail_op;   // clipped CIGAR = [part_op] + cigar[i_first .. i_first + n_whole) + [tail_op] (0 = none)
  uint32_t o_rq, o_hp, o_mm, o_ml;               // aux fields (type byte), 0 = absent
  int32_t start_offset, end_offset;
  uint32_t has_meth, n_meth; uint64_t meth_scratch;  // read_meth_kernel
};
struct ReadOff { uint64_t idx, bytes, name, snp, meth, cig, bam4; };
struct Counters { uint32_t n_tagged, meth_flag; uint64_t meth_scratch; uint64_t tot[7]; };

__device__ __forceinline__ uint32_t locus_of(const uint64_t* __restrict__ first, uint32_t n_loci, uint64_t w) {  // largest l with first[l] <= w
  uint32_t lo = 0, hi = n_loci;
  while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (first[mid] <= w) lo = mid; else hi = mid; }
  return lo;
}

__global__ void __launch_bounds__(64) read_sizes_kernel(const uint8_t* __restrict__ infl, const LocusDesc* __restrict__ loci, uint32_t n_loci, const uint64_t* __restrict__ first,
                                                        uint32_t reservoir, const uint64_t* __restrict__ rec_list, uint64_t n_pre, ReadInfo* __restrict__ info, Counters* __restrict__ counters) {
  const uint64_t w = blockIdx.x;
  if (w >= n_pre) return;
  const int lane = lane_id();
  const uint32_t li = locus_of(first, n_loci, w);
  const LocusDesc L = loci[li];
  const uint64_t q = rec_list[(uint64_t)li * reservoir + (w - first[li])];
  const uint32_t bs = ld32(infl + q);
  const uint8_t* rec = infl + q + 4;
  const RecHead h = rec_head(rec);
  const int64_t rs = L.clip_start, re = L.clip_end;
  // ---- the CIGAR, 64 operations at a time: reference / query position in front of every operation by prefix sums
  int64_t carry_r = h.pos; uint64_t carry_q = 0;
  bool found_a = false, found_b = false;
  int64_t a_r = 0, b_r = 0; uint64_t a_q = 0, b_q = 0; uint32_t a_op = 0, b_op = 0, i0 = h.n_cig, j1 = h.n_cig;
  uint64_t n_snp = 0;
  for (uint32_t base = 0; base < h.n_cig; base += 64) {
    const uint32_t k = base + (uint32_t)lane;
    const bool valid = k < h.n_cig;
    const uint32_t op = valid ? ld32(rec + h.o_cig + 4ull * k) : 0u;
    const uint32_t rl = ref_len(op), ql = qry_len(op);
    const uint64_t ir = wave_incl_sum(rl), iq = wave_incl_sum(ql);
    const int64_t r_after = carry_r + (int64_t)ir, r_before = r_after - rl;
    const uint64_t q_before = carry_q + iq - ql;
    const uint32_t n_valid = h.n_cig - base < 64u ? h.n_cig - base : 64u;
    if (!found_a) {
      const uint32_t ca = (uint32_t)__popcll(ballot64(valid && r_after <= rs));
      if (ca < n_valid) { found_a = true; i0 = base + ca; a_r = (int64_t)shfl64((uint64_t)r_before, (int)ca); a_q = shfl64(q_before, (int)ca); a_op = (uint32_t)__shfl((int)op, (int)ca); }
    }
    if (!found_b) {
      const uint32_t cb = (uint32_t)__popcll(ballot64(valid && r_after <= re));
      if (cb < n_valid) { found_b = true; j1 = base + cb; b_r = (int64_t)shfl64((uint64_t)r_before, (int)cb); b_q = shfl64(q_before, (int)cb); b_op = (uint32_t)__shfl((int)op, (int)cb); }
    }
    // extract_snps_offset (snp.rs:51-79): the X runs that start outside [region_start, region_end]; positions in u32 as there
    const uint32_t start_ref = (uint32_t)r_before;
    const bool inside = (int64_t)start_ref >= L.region_start && (int64_t)start_ref <= L.region_end;
    n_snp += wave_sum((valid && (op & 0xFu) == OP_X && !inside) ? (uint64_t)(op >> 4) : 0ull);
    carry_r = (int64_t)shfl64((uint64_t)r_after, 63); carry_q = shfl64(carry_q + iq, 63);
  }
  const int64_t read_end = carry_r; const uint64_t total_q = carry_q;
  ReadInfo R;
  memset(&R, 0, sizeof R);
  R.rec = q + 4; R.bs = bs; R.locus = li;
  R.start_offset = (int32_t)((int64_t)h.pos - L.region_start); R.end_offset = (int32_t)(read_end - L.region_end);
  bool ok = !(h.flag & 0x4u) && !(read_end <= rs || re <= (int64_t)h.pos);
  if (ok) {
    int64_t c_ref = found_a ? a_r : read_end; uint64_t c_qry = found_a ? a_q : total_q;
    uint32_t i = i0, part = 0, tail = 0; uint64_t q_len = 0; bool beyond = false;
    uint64_t q_i = c_qry;  // query position in front of operation i
    if (found_a && a_r < rs) {  // the operation across the start of the window is split
      const int64_t outside = rs - a_r, rl0 = ref_len(a_op);
      const int64_t keep = a_r + rl0 <= re ? rl0 - outside : re - rs;
      part = ((uint32_t)keep << 4) | (a_op & 0xFu);
      c_ref += outside;
      if (qry_len(part) != 0) c_qry += (uint64_t)outside;
      q_len += qry_len(part);
      q_i = a_q + qry_len(a_op);
      i = i0 + 1;
      beyond = a_r + rl0 > re;
    }
    uint32_t n_whole = 0;
    if (!beyond) {
      const uint64_t q_j1 = found_b ? b_q : total_q;
      if (j1 > i) { n_whole = j1 - i; q_len += q_j1 - q_i; }
      if (found_b && j1 < h.n_cig && b_r < re) { tail = ((uint32_t)(re - b_r) << 4) | (b_op & 0xFu); q_len += qry_len(tail); }
    }
    const uint64_t q_end = c_qry + q_len;
    if (q_end > (uint64_t)(uint32_t)h.l_seq) ok = false;
    else {
      R.c_ref = c_ref; R.c_qry = (uint32_t)c_qry; R.n_bases = (uint32_t)q_len; R.i_first = i; R.n_whole = n_whole; R.part_op = part; R.tail_op = tail;
      R.n_cig = (part ? 1u : 0u) + n_whole + (tail ? 1u : 0u);
      R.n_name = h.l_rn ? h.l_rn - 1 : 0; R.n_snp = (uint32_t)n_snp;
    }
  }
  R.ok = ok ? 1u : 0u;
  if (ok) {
    AuxAt at; aux_scan(rec, h.o_aux, bs, at, false);
    R.o_rq = at.rq; R.o_hp = at.hp; R.o_mm = at.mm ? at.mm : at.mm2; R.o_ml = at.ml ? at.ml : at.ml2;
    if (R.o_mm && R.o_ml && lane == 0) {
      atomicAdd(&counters->n_tagged, 1u);
      R.meth_scratch = atomicAdd((unsigned long long*)&counters->meth_scratch, (unsigned long long)(R.n_bases / 2 + 1));
    }
  }
  if (lane == 0) info[w] = R;
}

// ------------------------------------------------------------------------------------------------ get_meth on the device
constexpr uint32_t METH_MAX_SEQ = 65535, METH_WORDS = 1024, METH_MAX_MODS = 4096, METH_TEXT = 2048;
struct MethShared {
  uint64_t cpg[METH_WORDS], want[METH_WORDS];    // bit per stored C of a CpG; bit per base the MM entry counts over, in ORIGINAL-strand order
  uint16_t cpg_rank[METH_WORDS + 1], want_rank[METH_WORDS + 1];   // set bits in front of every word
  uint16_t pos[METH_MAX_MODS];                   // stored positions of the calls of one MM entry
  uint8_t text[METH_TEXT + 16];
};
__device__ __forceinline__ uint32_t base_code(const uint8_t* seq, uint32_t i) { const uint32_t b = seq[i >> 1]; return (i & 1u) ? (b & 0xFu) : (b >> 4); }
__device__ __forceinline__ uint32_t mask_rank(const uint64_t* m, const uint16_t* rk, uint32_t i) {  // set bits at positions < i
  const uint32_t w = i >> 6, b = i & 63u;
  return (uint32_t)rk[w] + (b ? (uint32_t)__popcll(m[w] & (~0ull >> (64 - b))) : 0u);
}
__device__ __forceinline__ uint32_t mask_select(const uint64_t* m, const uint16_t* rk, uint32_t n_words, uint32_t t) {  // position of set bit number t (0-based); t < total
  uint32_t lo = 0, hi = n_words;  // largest w with rk[w] <= t
  while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if ((uint32_t)rk[mid] <= t) lo = mid; else hi = mid; }
  uint64_t v = m[lo]; uint32_t k = t - rk[lo];
  while (k--) v &= v - 1;
  return (lo << 6) + (uint32_t)__builtin_ctzll(v);
}

// One wave per read that carries MM + ML: basemods_5mc_tags / meth_per_cpg of ingest.hip (htslib's bam_parse_basemod as get_meth uses it)
// on wave-uniform values; the values of the CpGs inside the clipped part go to scratch, has_meth / n_meth into the read's info.
__global__ void __launch_bounds__(64) read_meth_kernel(const uint8_t* __restrict__ infl, uint64_t n_pre, ReadInfo* __restrict__ info, uint8_t* __restrict__ scratch, Counters* __restrict__ counters) {
  __shared__ MethShared sh;
  const uint64_t w = blockIdx.x;
  if (w >= n_pre) return;
  const int lane = lane_id();
  ReadInfo R = info[w];
  if (!R.ok || !R.o_mm || !R.o_ml) return;
  const uint8_t* rec = infl + R.rec;
  const RecHead h = rec_head(rec);
  const uint8_t* mm = rec + R.o_mm; const uint8_t* ml = rec + R.o_ml;
  if (mm[0] != 'Z' || ml[0] != 'B' || (ml[1] != 'C' && ml[1] != 'c')) return;  // has_meth stays 0
  const uint32_t n = (uint32_t)h.l_seq;
  if (n > METH_MAX_SEQ) { if (lane == 0) counters->meth_flag = 1; return; }
  const uint32_t n_ml = ld32(ml + 2); const uint8_t* mlv = ml + 6;
  const uint8_t* seq = rec + h.o_seq;
  const bool rev = (h.flag & 0x10u) != 0;
  const uint32_t want_code = rev ? 4u : 2u;  // "=ACMGRSVTWYHKDBN": C = 2, G = 4
  const uint32_t n_words = (n + 63) >> 6;
  for (uint32_t wd = 0; wd < n_words; ++wd) {
    const uint32_t i = (wd << 6) + (uint32_t)lane;
    const bool c = i + 1 < n && base_code(seq, i) == 2u && base_code(seq, i + 1) == 4u;
    const uint32_t at = rev ? n - 1 - i : i;  // logical (original-strand) index i -> stored index
    const bool wn = i < n && base_code(seq, at) == want_code;
    const uint64_t mc = ballot64(c), mw = ballot64(wn);
    if (lane == 0) { sh.cpg[wd] = mc; sh.want[wd] = mw; }
  }
  __syncthreads();
  if (lane == 0) { uint32_t a = 0, b = 0; for (uint32_t wd = 0; wd < n_words; ++wd) { sh.cpg_rank[wd] = (uint16_t)a; sh.want_rank[wd] = (uint16_t)b; a += (uint32_t)__popcll(sh.cpg[wd]); b += (uint32_t)__popcll(sh.want[wd]); } sh.cpg_rank[n_words] = (uint16_t)a; sh.want_rank[n_words] = (uint16_t)b; }
  __syncthreads();
  const uint32_t ncpg = sh.cpg_rank[n_words], n_want = sh.want_rank[n_words];
  const uint32_t mi0 = mask_rank(sh.cpg, sh.cpg_rank, R.c_qry), mi1 = mask_rank(sh.cpg, sh.cpg_rank, R.c_qry + R.n_bases);
  uint8_t* dst = scratch + R.meth_scratch;
  for (uint32_t i = (uint32_t)lane; i < mi1 - mi0; i += 64) dst[i] = 0;
  __syncthreads();
  // ---- the MM text through a window in LDS; the parser runs on uniform values
  const uint8_t* text = mm + 1;
  const uint64_t text_room = (uint64_t)R.bs - (R.o_mm + 1);  // bytes of the record behind the type byte
  uint32_t wb = 0;  // text offset of sh.text[0]
  auto load_window = [&](uint32_t from) {
    __syncthreads();
    for (uint32_t i = (uint32_t)lane; i < METH_TEXT + 16; i += 64) sh.text[i] = (uint64_t)from + i < text_room ? text[from + i] : (uint8_t)0;
    wb = from;
    __syncthreads();
  };
  load_window(0);
  uint32_t s = 0;
  auto T = [&](uint32_t at) -> uint32_t { return sh.text[at - wb]; };
  auto ensure = [&](uint32_t at) { if (at + 16 > wb + METH_TEXT) load_window(at); };
  uint32_t ml_at = 0, ind = 0;
  bool failed = false, overflow = false;
  for (;;) {
    ensure(s);
    const uint32_t base = T(s);
    if (!base) break;
    const uint32_t strand = T(s + 1);
    const bool is_base = base == 'A' || base == 'C' || base == 'G' || base == 'T' || base == 'U' || base == 'N';
    if (!is_base || (strand != '+' && strand != '-')) { failed = true; break; }
    uint32_t q = s + 2, n_code_chars = 0, m_idx = ~0u; bool all_digits = true;
    for (;;) {
      ensure(q);
      const uint32_t ch = T(q);
      if (!ch || ch == ',' || ch == ';' || ch == '.' || ch == '?') break;
      if (ch == 'm' && m_idx == ~0u) m_idx = n_code_chars;
      if (ch < '0' || ch > '9') all_digits = false;
      ++n_code_chars; ++q;
    }
    if (!n_code_chars) { failed = true; break; }
    uint32_t n_codes = n_code_chars;
    if (all_digits) { n_codes = 1; m_idx = ~0u; }  // a numeric ChEBI code is ONE modification
    { const uint32_t ch = T(q); if (ch == '.' || ch == '?') ++q; }
    const bool takes = base == 'C' && strand == '+' && m_idx != ~0u;
    uint32_t n_deltas = 0, n_found = 0; uint64_t t = 0; bool run = true;
    for (;;) {
      ensure(q);
      if (T(q) != ',') break;
      ++q;
      uint32_t v = 0;
      for (;;) { ensure(q); const uint32_t ch = T(q); if (ch < '0' || ch > '9') break; v = v * 10u + (ch - '0'); ++q; }
      if (takes && run) {
        t += v;
        if (t >= n_want) run = false;
        else {
          const uint32_t li = mask_select(sh.want, sh.want_rank, n_words, (uint32_t)t);  // logical index of the call
          const uint32_t at = rev ? n - 1 - li : li;
          const uint64_t mi = (uint64_t)ml_at + (uint64_t)n_deltas * n_codes + m_idx;
          if (mi < n_ml) { if (n_found < METH_MAX_MODS) { if (lane == 0) sh.pos[n_found] = (uint16_t)at; } else overflow = true; ++n_found; }
          t += 1;
        }
      }
      ++n_deltas;
    }
    ensure(q);
    if (T(q) == ';') ++q;
    s = q;
    if (takes && n_found && !overflow) {
      __syncthreads();
      // meth_per_cpg: the calls of this entry in stored order against the CpGs (ind never goes back)
      for (uint32_t j = 0; j < n_found; ++j) {
        const uint32_t k = rev ? n_found - 1 - j : j;
        const uint32_t P = sh.pos[k];
        const uint32_t off = rev ? 1u : 0u;
        const uint32_t c = P >= off ? P - off : 0u;
        const uint32_t r = P >= off ? mask_rank(sh.cpg, sh.cpg_rank, c) : 0u;
        const bool is_cpg = P >= off && c < n && ((sh.cpg[c >> 6] >> (c & 63u)) & 1ull);
        if (r >= ind) {
          ind = r;
          if (is_cpg) {
            const uint32_t m = rev ? ncpg - 1 - r : r;
            if (m >= mi0 && m < mi1 && lane == 0) dst[m - mi0] = mlv[(uint64_t)ml_at + (uint64_t)k * n_codes + m_idx];
            ind = r + 1;
          }
        }
      }
      __syncthreads();
    }
    if (overflow) break;
    ml_at += n_deltas * n_codes;
  }
  if (overflow) { if (lane == 0) counters->meth_flag = 1; return; }
  if (lane == 0 && !failed && ind != 0) { info[w].has_meth = 1; info[w].n_meth = mi1 - mi0; }
}

// ------------------------------------------------------------------------------------------------ where every read's pieces go
// One workgroup: thread t sums a contiguous run of reads, the runs' sums are scanned, the run is walked again.
__global__ void __launch_bounds__(1024) scan_kernel(const ReadInfo* __restrict__ info, uint64_t n_pre, ReadOff* __restrict__ off, const uint64_t* __restrict__ first, uint32_t n_loci,
                                                    uint64_t* __restrict__ lrb, Counters* __restrict__ counters) {
  __shared__ uint64_t part[1024][7];
  const uint32_t t = threadIdx.x;
  const uint64_t run = (n_pre + 1023) / 1024, a = (uint64_t)t * run, b = a + run < n_pre ? a + run : n_pre;
  auto sizes = [&](const ReadInfo& R, uint64_t* v) {
    if (!R.ok) { for (int k = 0; k < 7; ++k) v[k] = 0; return; }
    v[0] = 1; v[1] = R.n_bases; v[2] = R.n_name; v[3] = R.n_snp; v[4] = R.has_meth ? R.n_meth : 0; v[5] = R.n_cig; v[6] = ((uint64_t)R.n_bases + 1) / 2;
  };
  uint64_t sum[7] = {0, 0, 0, 0, 0, 0, 0};
  for (uint64_t i = a; i < b; ++i) { uint64_t v[7]; sizes(info[i], v); for (int k = 0; k < 7; ++k) sum[k] += v[k]; }
  for (int k = 0; k < 7; ++k) part[t][k] = sum[k];
  __syncthreads();
  if (t < 7) { uint64_t acc = 0; for (int i = 0; i < 1024; ++i) { const uint64_t v = part[i][t]; part[i][t] = acc; acc += v; } counters->tot[t] = acc; }
  __syncthreads();
  uint64_t acc[7];
  for (int k = 0; k < 7; ++k) acc[k] = part[t][k];
  for (uint64_t i = a; i < b; ++i) {
    uint64_t v[7]; sizes(info[i], v);
    ReadOff o; o.idx = acc[0]; o.bytes = acc[1]; o.name = acc[2]; o.snp = acc[3]; o.meth = acc[4]; o.cig = acc[5]; o.bam4 = acc[6];
    off[i] = o;
    for (int k = 0; k < 7; ++k) acc[k] += v[k];
  }
  __syncthreads();
  __threadfence();
  for (uint32_t l = t; l <= n_loci; l += 1024) { const uint64_t f = first[l]; lrb[l] = f < n_pre ? off[f].idx : counters->tot[0]; }
}

// ------------------------------------------------------------------------------------------------ per read: the pieces
struct OutPtrs {
  uint64_t* read_off; uint32_t* read_len; uint8_t* reads; uint8_t* quals; char* names; uint64_t* name_off; double* rq; uint8_t* is_reverse; uint8_t* mapq; int16_t* hp;
  int32_t* start_offset; int32_t* end_offset; int32_t* snp; uint64_t* snp_off; uint8_t* meth; uint64_t* meth_off; uint8_t* has_meth; uint32_t* cig; uint64_t* cig_off;
  int64_t* cig_ref_pos; uint8_t* bam4; uint64_t* bam4_off;
};

__global__ void __launch_bounds__(64) read_fill_kernel(const uint8_t* __restrict__ infl, const LocusDesc* __restrict__ loci, const ReadInfo* __restrict__ info, const ReadOff* __restrict__ off,
                                                       uint64_t n_pre, const uint8_t* __restrict__ scratch, OutPtrs o) {
  const uint64_t w = blockIdx.x;
  if (w >= n_pre) return;
  const ReadInfo R = info[w];
  if (!R.ok) return;
  const int lane = lane_id();
  const ReadOff F = off[w];
  const uint8_t* rec = infl + R.rec;
  const RecHead h = rec_head(rec);
  const uint64_t k = F.idx;
  if (lane == 0) {
    o.read_off[k] = F.bytes; o.read_len[k] = R.n_bases;
    double rq = __longlong_as_double(0x7FF8000000000000ll);
    if (R.o_rq && rec[R.o_rq] == 'f') rq = (double)__uint_as_float(ld32(rec + R.o_rq + 1));
    o.rq[k] = rq;
    o.is_reverse[k] = (h.flag & 0x10u) ? 1 : 0; o.mapq[k] = (uint8_t)h.mapq;
    o.hp[k] = (R.o_hp && rec[R.o_hp] == 'C') ? (int16_t)rec[R.o_hp + 1] : (int16_t)-1;
    o.start_offset[k] = R.start_offset; o.end_offset[k] = R.end_offset;
    o.has_meth[k] = R.has_meth ? 1 : 0;
    o.cig_ref_pos[k] = R.c_ref;
    o.name_off[k + 1] = F.name + R.n_name; o.snp_off[k + 1] = F.snp + R.n_snp; o.meth_off[k + 1] = F.meth + (R.has_meth ? R.n_meth : 0); o.cig_off[k + 1] = F.cig + R.n_cig;
    if (k == 0) { o.name_off[0] = 0; o.snp_off[0] = 0; o.meth_off[0] = 0; o.cig_off[0] = 0; }
    if (o.bam4) o.bam4_off[k] = F.bam4;
  }
  // bases ("=ACMGRSVTWYHKDBN"[code]) and qualities of [c_qry, c_qry + n_bases)
  {
    const uint8_t* seq = rec + h.o_seq; const uint8_t* qual = rec + h.o_qual;
    uint8_t* db = o.reads + F.bytes; uint8_t* dq = o.quals + F.bytes;
    for (uint32_t j = (uint32_t)lane; j < R.n_bases; j += 64) {
      const uint32_t code = base_code(seq, R.c_qry + j);
      db[j] = (uint8_t)"=ACMGRSVTWYHKDBN"[code];
      dq[j] = qual[R.c_qry + j];
    }
    if (o.bam4) {
      uint8_t* d4 = o.bam4 + F.bam4;
      for (uint32_t j = (uint32_t)lane; j < (R.n_bases + 1) / 2; j += 64) {
        const uint32_t hi = base_code(seq, R.c_qry + 2 * j), lo = 2 * j + 1 < R.n_bases ? base_code(seq, R.c_qry + 2 * j + 1) : 0u;
        d4[j] = (uint8_t)((hi << 4) | lo);
      }
    }
  }
  for (uint32_t j = (uint32_t)lane; j < R.n_name; j += 64) o.names[F.name + j] = (char)rec[32 + j];
  // clipped CIGAR
  {
    uint32_t* dc = o.cig + F.cig;
    uint32_t at = 0;
    if (R.part_op) { if (lane == 0) dc[0] = R.part_op; at = 1; }
    for (uint32_t j = (uint32_t)lane; j < R.n_whole; j += 64) dc[at + j] = ld32(rec + h.o_cig + 4ull * (R.i_first + j));
    if (R.tail_op && lane == 0) dc[at + R.n_whole] = R.tail_op;
  }
  // mismatch offsets: the pass of read_sizes_kernel once more, now with the positions
  if (R.n_snp) {
    const LocusDesc L = loci[R.locus];
    int32_t* ds = o.snp + F.snp;
    int64_t carry_r = h.pos; uint64_t carry_n = 0;
    for (uint32_t base = 0; base < h.n_cig; base += 64) {
      const uint32_t kk = base + (uint32_t)lane;
      const bool valid = kk < h.n_cig;
      const uint32_t op = valid ? ld32(rec + h.o_cig + 4ull * kk) : 0u;
      const uint32_t rl = ref_len(op);
      const uint64_t ir = wave_incl_sum(rl);
      const int64_t r_before = carry_r + (int64_t)ir - rl;
      const uint32_t start_ref = (uint32_t)r_before;
      const bool inside = (int64_t)start_ref >= L.region_start && (int64_t)start_ref <= L.region_end;
      const uint64_t nx = (valid && (op & 0xFu) == OP_X && !inside) ? (uint64_t)(op >> 4) : 0ull;
      const uint64_t in = wave_incl_sum(nx);
      if (nx) {
        const int32_t diff = (int64_t)start_ref < L.region_start ? (int32_t)start_ref - (int32_t)L.region_start : (int32_t)start_ref - (int32_t)L.region_end;
        const uint64_t at = carry_n + in - nx;
        for (uint32_t i = 0; i < (uint32_t)nx; ++i) ds[at + i] = diff + (int32_t)i;
      }
      carry_r += (int64_t)shfl64(ir, 63); carry_n += shfl64(in, 63);
    }
  }
  if (R.has_meth) for (uint32_t j = (uint32_t)lane; j < R.n_meth; j += 64) o.meth[F.meth + j] = scratch[R.meth_scratch + j];
}

// ================================================================================================ host side
SlabPool::~SlabPool() {
  for (auto& s : idle) { if (s.dev) { (void)hipSetDevice(s.device); (void)hipFree(s.dev); } if (s.pin) (void)hipHostFree(s.pin); }
}
bool SlabPool::take(int device, size_t bytes, Slab& out) {
  {
    std::lock_guard<std::mutex> g(mu);
    size_t best = idle.size();
    for (size_t i = 0; i < idle.size(); ++i) if (idle[i].device == device && idle[i].cap >= bytes && (best == idle.size() || idle[i].cap < idle[best].cap)) best = i;
    if (best < idle.size()) { out = idle[best]; idle.erase(idle.begin() + (ptrdiff_t)best); return true; }
    // nothing fits: the smallest idle slab of this device makes room
    for (size_t i = 0; i < idle.size(); ++i) if (idle[i].device == device) { Slab s = idle[i]; idle.erase(idle.begin() + (ptrdiff_t)i); (void)hipSetDevice(device); (void)hipFree(s.dev); (void)hipHostFree(s.pin); break; }
  }
  Slab s; s.device = device; s.cap = bytes + bytes / 4 + (1u << 20);
  (void)hipSetDevice(device);
  if (hipMalloc(&s.dev, s.cap) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (hipHostMalloc(&s.pin, s.cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(s.dev); return false; }
  out = s;
  return true;
}
void SlabPool::give(Slab& s) {
  if (!s.dev && !s.pin) return;
  std::lock_guard<std::mutex> g(mu);
  idle.push_back(s);
  s = Slab();
}

namespace {
struct DevBuf {
  void* p = nullptr; size_t cap = 0;
  bool need(size_t bytes) {
    if (cap >= bytes) return true;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    if (hipMalloc(&p, want) != hipSuccess) { (void)hipGetLastError(); p = nullptr; return false; }
    cap = want;
    return true;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};
struct PinBuf {
  void* p = nullptr; size_t cap = 0;
  bool need(size_t bytes) {
    if (cap >= bytes) return true;
    if (p) (void)hipHostFree(p);
    p = nullptr; cap = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); p = nullptr; return false; }
    cap = want;
    return true;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

void make_crc_tables(CrcTables& T) {
  for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1; T.t[0][i] = c; }
  for (uint32_t i = 0; i < 256; ++i) for (int s = 1; s < 4; ++s) T.t[s][i] = T.t[0][T.t[s - 1][i] & 0xFFu] ^ (T.t[s - 1][i] >> 8);
  // operator "one zero bit follows" of the reflected CRC, squared up to 1024 bytes = 2^13 bits, then doubled five times more
  uint32_t m[32], sq[32];
  m[0] = 0xEDB88320u;
  for (int b = 1; b < 32; ++b) m[b] = 1u << (b - 1);
  auto apply = [](const uint32_t* mat, uint32_t v) { uint32_t r = 0; for (int b = 0; v; v >>= 1, ++b) if (v & 1u) r ^= mat[b]; return r; };
  auto square = [&](uint32_t* dst, const uint32_t* src) { for (int b = 0; b < 32; ++b) dst[b] = apply(src, src[b]); };
  for (int k = 0; k < 13; ++k) { square(sq, m); std::memcpy(m, sq, sizeof m); }
  for (int j = 0; j < 6; ++j) { std::memcpy(T.shift[j], m, sizeof m); square(sq, m); std::memcpy(m, sq, sizeof m); }
}
}  // namespace

class Slot {
 public:
  int device = -1;
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;  // blocking-sync event: a caller SLEEPS while its slot's kernels run (a spinning wait per caller would eat three of the
                              // host's CPUs -- sixteen per process on the GPU boxes, shared with the file reads, the locus stage's host threads and the writer)
  PinBuf src_pin, small_pin;
  DevBuf d_src, d_blocks, d_crc, d_infl, d_status, d_counter, d_loci, d_chunks, d_list, d_walk, d_first, d_info, d_off, d_counters, d_scratch, d_lrb, d_tab;
  bool tab_ready = false;
  ~Slot() {
    if (device >= 0) (void)hipSetDevice(device);
    for (DevBuf* b : {&d_src, &d_blocks, &d_crc, &d_infl, &d_status, &d_counter, &d_loci, &d_chunks, &d_list, &d_walk, &d_first, &d_info, &d_off, &d_counters, &d_scratch, &d_lrb, &d_tab}) b->release();
    src_pin.release(); small_pin.release();
    if (done) (void)hipEventDestroy(done);
    if (stream) (void)hipStreamDestroy(stream);
  }
};
static inline hipError_t slot_wait(Slot* s) {
  const hipError_t e = hipEventRecord(s->done, s->stream);
  return e != hipSuccess ? e : hipEventSynchronize(s->done);
}

Slot* slot_create(int device, std::string& err) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) { (void)hipGetLastError(); err = "trgt_ingest: ingest_device " + std::to_string(device) + ": no such GPU"; return nullptr; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess || std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) { (void)hipGetLastError(); err = "trgt_ingest: ingest_device " + std::to_string(device) + " is not a gfx950 GPU"; return nullptr; }
  (void)hipSetDevice(device);
  std::unique_ptr<Slot> s(new Slot());
  s->device = device;
  // (the least urgent priority: the launches of trgt_locus_batch on the same GPU go first wherever a CU has room for both)
  int prio_lo = 0, prio_hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
  if (hipStreamCreateWithPriority(&s->stream, hipStreamNonBlocking, prio_lo) != hipSuccess) { (void)hipGetLastError(); err = "trgt_ingest: hipStreamCreate failed"; return nullptr; }
  if (hipEventCreateWithFlags(&s->done, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); err = "trgt_ingest: hipEventCreate failed"; return nullptr; }
  return s.release();
}
void slot_destroy(Slot* s) { delete s; }

uint8_t* slot_src(Slot* s, size_t bytes, std::string& err) {
  (void)hipSetDevice(s->device);
  if (!s->src_pin.need(bytes + 64)) { err = "trgt_ingest: no pinned memory for the compressed blocks"; return nullptr; }
  return (uint8_t*)s->src_pin.p;
}

#define ING_TRY(expr)                                                                                                          \
  do {                                                                                                                          \
    hipError_t e__ = (expr);                                                                                                    \
    if (e__ != hipSuccess) { err = std::string("trgt_ingest (device): ") + #expr + " failed: " + hipGetErrorString(e__); return TRGT_ERR_HIP; } \
  } while (0)
#define ING_NEED(buf, bytes) do { if (!(buf).need(bytes)) { err = "trgt_ingest (device): out of device memory"; return TRGT_ERR_NOMEM; } } while (0)

int slot_run(Slot* s, const RunIn& in, SlabPool& pool, RunOut& out, std::string& err) {
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  out = RunOut();
  ING_TRY(hipSetDevice(s->device));
  hipStream_t st = s->stream;
  const double t0 = now();
  const uint32_t nb = (uint32_t)in.n_blocks, nl = (uint32_t)in.n_loci;
  if (!s->tab_ready) {
    ING_NEED(s->d_tab, sizeof(CrcTables));
    CrcTables T; make_crc_tables(T);
    ING_TRY(hipMemcpy(s->d_tab.p, &T, sizeof T, hipMemcpyHostToDevice));
    s->tab_ready = true;
  }
  // ---- upload: compressed bytes (staged by the caller in slot_src()), block table, footer CRCs, locus and chunk tables
  ING_NEED(s->d_src, in.src_bytes + 64); ING_NEED(s->d_blocks, (size_t)nb * sizeof(infl::BlockDesc) + 64); ING_NEED(s->d_crc, (size_t)nb * 4 + 64);
  ING_NEED(s->d_infl, in.infl_bytes + 256); ING_NEED(s->d_status, (size_t)nb + 64); ING_NEED(s->d_counter, 64);
  ING_NEED(s->d_loci, (size_t)nl * sizeof(LocusDesc) + 64); ING_NEED(s->d_chunks, (size_t)in.n_chunks * sizeof(ChunkDesc) + 64);
  ING_NEED(s->d_list, (size_t)nl * in.reservoir * 8 + 64); ING_NEED(s->d_walk, (size_t)nl * sizeof(WalkOut) + 64); ING_NEED(s->d_first, ((size_t)nl + 1) * 8 + 64);
  ING_NEED(s->d_counters, sizeof(Counters)); ING_NEED(s->d_lrb, ((size_t)nl + 1) * 8 + 64);
  const size_t small_bytes = (size_t)nb * (sizeof(infl::BlockDesc) + 4 + 1) + (size_t)nl * (sizeof(LocusDesc) + sizeof(WalkOut) + 16) + (size_t)in.n_chunks * sizeof(ChunkDesc) + sizeof(Counters) + 1024;
  if (!s->small_pin.need(small_bytes)) { err = "trgt_ingest (device): no pinned memory"; return TRGT_ERR_NOMEM; }
  uint8_t* sp = (uint8_t*)s->small_pin.p;
  auto carve = [&](size_t bytes) { uint8_t* p = sp; sp += (bytes + 63) & ~(size_t)63; return p; };
  uint8_t* h_blocks = carve((size_t)nb * sizeof(infl::BlockDesc)); uint8_t* h_crc = carve((size_t)nb * 4); uint8_t* h_loci = carve((size_t)nl * sizeof(LocusDesc));
  uint8_t* h_chunks = carve((size_t)in.n_chunks * sizeof(ChunkDesc)); uint8_t* h_status = carve(nb); WalkOut* h_walk = (WalkOut*)carve((size_t)nl * sizeof(WalkOut));
  uint64_t* h_first = (uint64_t*)carve(((size_t)nl + 1) * 8); Counters* h_counters = (Counters*)carve(sizeof(Counters));
  std::memcpy(h_blocks, in.blocks, (size_t)nb * sizeof(infl::BlockDesc)); std::memcpy(h_crc, in.crc, (size_t)nb * 4);
  std::memcpy(h_loci, in.loci, (size_t)nl * sizeof(LocusDesc)); std::memcpy(h_chunks, in.chunks, (size_t)in.n_chunks * sizeof(ChunkDesc));
  ING_TRY(hipMemcpyAsync(s->d_src.p, s->src_pin.p, in.src_bytes, hipMemcpyHostToDevice, st));
  ING_TRY(hipMemcpyAsync(s->d_blocks.p, h_blocks, (size_t)nb * sizeof(infl::BlockDesc), hipMemcpyHostToDevice, st));
  ING_TRY(hipMemcpyAsync(s->d_crc.p, h_crc, (size_t)nb * 4, hipMemcpyHostToDevice, st));
  ING_TRY(hipMemcpyAsync(s->d_loci.p, h_loci, (size_t)nl * sizeof(LocusDesc), hipMemcpyHostToDevice, st));
  ING_TRY(hipMemcpyAsync(s->d_chunks.p, h_chunks, (size_t)in.n_chunks * sizeof(ChunkDesc), hipMemcpyHostToDevice, st));
  ING_TRY(hipMemsetAsync(s->d_counter.p, 0, 64, st));
  ING_TRY(hipMemsetAsync(s->d_counters.p, 0, sizeof(Counters), st));
  ING_TRY(hipMemsetAsync(s->d_status.p, 0, (size_t)nb + 64, st));
  // ---- inflate + CRC-32
  const unsigned waves_per_cu = in.waves_per_cu > 0 ? (unsigned)std::min(in.waves_per_cu, 16) : 12u;
  int cus = 256; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, s->device);
  trgt::inflate_launch((void*)st, (const uint8_t*)s->d_src.p, (const infl::BlockDesc*)s->d_blocks.p, nb, (uint8_t*)s->d_infl.p, (uint8_t*)s->d_status.p, (unsigned*)s->d_counter.p,
                       (unsigned)cus * waves_per_cu);
  if (nb) hipLaunchKernelGGL(crc32_blocks_kernel, dim3(nb), dim3(64), 0, st, (const uint8_t*)s->d_infl.p, (const infl::BlockDesc*)s->d_blocks.p, (const uint32_t*)s->d_crc.p, nb,
                             (const CrcTables*)s->d_tab.p, (uint8_t*)s->d_status.p);
  ING_TRY(hipGetLastError());
  ING_TRY(hipMemcpyAsync(h_status, s->d_status.p, nb, hipMemcpyDeviceToHost, st));
  ING_TRY(slot_wait(s));
  const double t1 = now();
  // a block the device declined goes through zlib here (rare: the kernel takes every stream zlib level 1-9 writes); a block that does not
  // inflate to its ISIZE or whose CRC-32 differs sends the call to the host path, which reports it
  for (uint32_t b = 0; b < nb; ++b) {
    if (h_status[b] == 1) continue;
    if (h_status[b] == 2) { out.fallback = FB_BLOCK; return TRGT_OK; }
    const infl::BlockDesc& bd = in.blocks[b];
    std::vector<uint8_t> tmp(bd.dst_len);
    z_stream zs; std::memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) { out.fallback = FB_BLOCK; return TRGT_OK; }
    zs.next_in = (Bytef*)s->src_pin.p + bd.src_off; zs.avail_in = bd.src_len; zs.next_out = tmp.data(); zs.avail_out = bd.dst_len;
    const int rc = inflate(&zs, Z_FINISH);
    const bool good = rc == Z_STREAM_END && zs.avail_out == 0 && trgt::crc32_fast(tmp.data(), bd.dst_len) == in.crc[b];
    inflateEnd(&zs);
    if (!good) { out.fallback = FB_BLOCK; return TRGT_OK; }
    ING_TRY(hipMemcpyAsync((uint8_t*)s->d_infl.p + bd.dst_off, tmp.data(), bd.dst_len, hipMemcpyHostToDevice, st));
    ING_TRY(slot_wait(s));
    ++out.blocks_host_inflated;
  }
  // ---- the record walk
  if (nl) hipLaunchKernelGGL(walk_kernel, dim3(nl), dim3(64), 0, st, (const uint8_t*)s->d_infl.p, (const LocusDesc*)s->d_loci.p, (const ChunkDesc*)s->d_chunks.p, nl, in.reservoir, in.min_rq,
                             (uint64_t*)s->d_list.p, (WalkOut*)s->d_walk.p);
  ING_TRY(hipGetLastError());
  ING_TRY(hipMemcpyAsync(h_walk, s->d_walk.p, (size_t)nl * sizeof(WalkOut), hipMemcpyDeviceToHost, st));
  ING_TRY(slot_wait(s));
  const double t2 = now();
  uint64_t n_pre = 0;
  for (uint32_t l = 0; l < nl; ++l) {
    if (h_walk[l].status == WS_ERROR) { out.fallback = FB_WALK; return TRGT_OK; }
    h_first[l] = n_pre; n_pre += std::min(h_walk[l].n_kept, in.reservoir);  // (n_kept counts every read that passed the filters: beyond the reservoir they replaced others)
  }
  h_first[nl] = n_pre;
  if (n_pre >= (1ull << 31)) { out.fallback = FB_WALK; return TRGT_OK; }
  ING_NEED(s->d_info, (size_t)(n_pre + 1) * sizeof(ReadInfo)); ING_NEED(s->d_off, (size_t)(n_pre + 1) * sizeof(ReadOff));
  ING_TRY(hipMemcpyAsync(s->d_first.p, h_first, ((size_t)nl + 1) * 8, hipMemcpyHostToDevice, st));
  if (n_pre) hipLaunchKernelGGL(read_sizes_kernel, dim3((unsigned)n_pre), dim3(64), 0, st, (const uint8_t*)s->d_infl.p, (const LocusDesc*)s->d_loci.p, nl, (const uint64_t*)s->d_first.p, in.reservoir,
                                (const uint64_t*)s->d_list.p, n_pre, (ReadInfo*)s->d_info.p, (Counters*)s->d_counters.p);
  auto scan = [&]() {
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, (const ReadInfo*)s->d_info.p, n_pre, (ReadOff*)s->d_off.p, (const uint64_t*)s->d_first.p, nl, (uint64_t*)s->d_lrb.p, (Counters*)s->d_counters.p);
  };
  scan();
  ING_TRY(hipGetLastError());
  ING_TRY(hipMemcpyAsync(h_counters, s->d_counters.p, sizeof(Counters), hipMemcpyDeviceToHost, st));
  ING_TRY(slot_wait(s));
  if (h_counters->n_tagged) {  // reads with MM + ML: get_meth, then the sizes once more
    ING_NEED(s->d_scratch, (size_t)h_counters->meth_scratch + 64);
    hipLaunchKernelGGL(read_meth_kernel, dim3((unsigned)n_pre), dim3(64), 0, st, (const uint8_t*)s->d_infl.p, n_pre, (ReadInfo*)s->d_info.p, (uint8_t*)s->d_scratch.p, (Counters*)s->d_counters.p);
    scan();
    ING_TRY(hipGetLastError());
    ING_TRY(hipMemcpyAsync(h_counters, s->d_counters.p, sizeof(Counters), hipMemcpyDeviceToHost, st));
    ING_TRY(slot_wait(s));
    if (h_counters->meth_flag) { out.fallback = FB_METH; return TRGT_OK; }
  }
  const double t3 = now();
  // ---- the slab of the batch: every per-read array a 64-byte aligned piece, device copy + pinned mirror
  const uint64_t nr = h_counters->tot[0], n_bytes = h_counters->tot[1], n_name = h_counters->tot[2], n_snp = h_counters->tot[3], n_meth = h_counters->tot[4], n_cig = h_counters->tot[5],
                 n_bam4 = in.keep_bam4 ? h_counters->tot[6] : 0;
  size_t at = 0;
  auto piece = [&](size_t bytes) { const size_t a = at; at += (bytes + 63) & ~(size_t)63; return a; };
  const size_t a_lrb = piece(((size_t)nl + 1) * 8), a_nfilt = piece((size_t)nl * 4), a_nseen = piece((size_t)nl * 8), a_roff = piece((nr + 1) * 8), a_rlen = piece((nr + 1) * 4),
               a_reads = piece(n_bytes + 1), a_quals = piece(n_bytes + 1), a_names = piece(n_name + 1), a_noff = piece((nr + 1) * 8), a_rq = piece((nr + 1) * 8), a_rev = piece(nr + 1), a_mapq = piece(nr + 1),
               a_hp = piece((nr + 1) * 2), a_so = piece((nr + 1) * 4), a_eo = piece((nr + 1) * 4), a_snp = piece((n_snp + 1) * 4), a_soff = piece((nr + 1) * 8), a_meth = piece(n_meth + 1),
               a_moff = piece((nr + 1) * 8), a_hm = piece(nr + 1), a_cig = piece((n_cig + 1) * 4), a_coff = piece((nr + 1) * 8), a_cpos = piece((nr + 1) * 8),
               a_b4 = piece(in.keep_bam4 ? n_bam4 + 1 : 0), a_b4off = piece(in.keep_bam4 ? (nr + 1) * 8 : 0);
  if (!pool.take(s->device, at + 64, out.slab)) { err = "trgt_ingest (device): no memory for the batch's arrays"; return TRGT_ERR_NOMEM; }
  uint8_t* D = (uint8_t*)out.slab.dev; uint8_t* H = (uint8_t*)out.slab.pin;
  OutPtrs o;
  o.read_off = (uint64_t*)(D + a_roff); o.read_len = (uint32_t*)(D + a_rlen); o.reads = D + a_reads; o.quals = D + a_quals; o.names = (char*)(D + a_names); o.name_off = (uint64_t*)(D + a_noff);
  o.rq = (double*)(D + a_rq); o.is_reverse = D + a_rev; o.mapq = D + a_mapq; o.hp = (int16_t*)(D + a_hp); o.start_offset = (int32_t*)(D + a_so); o.end_offset = (int32_t*)(D + a_eo);
  o.snp = (int32_t*)(D + a_snp); o.snp_off = (uint64_t*)(D + a_soff); o.meth = D + a_meth; o.meth_off = (uint64_t*)(D + a_moff); o.has_meth = D + a_hm; o.cig = (uint32_t*)(D + a_cig);
  o.cig_off = (uint64_t*)(D + a_coff); o.cig_ref_pos = (int64_t*)(D + a_cpos); o.bam4 = in.keep_bam4 ? D + a_b4 : nullptr; o.bam4_off = in.keep_bam4 ? (uint64_t*)(D + a_b4off) : nullptr;
  if (n_bytes == 0) ING_TRY(hipMemsetAsync(D + a_reads, 0, 64, st));  // (an empty blob is handed out as one NUL byte, like the host path's)
  if (nr == 0) ING_TRY(hipMemsetAsync(D + a_noff, 0, 64, st));  // (the [n_reads + 1] offset arrays hold a 0 at least)
  if (nr == 0) { ING_TRY(hipMemsetAsync(D + a_soff, 0, 64, st)); ING_TRY(hipMemsetAsync(D + a_moff, 0, 64, st)); ING_TRY(hipMemsetAsync(D + a_coff, 0, 64, st)); }
  if (n_pre) hipLaunchKernelGGL(read_fill_kernel, dim3((unsigned)n_pre), dim3(64), 0, st, (const uint8_t*)s->d_infl.p, (const LocusDesc*)s->d_loci.p, (const ReadInfo*)s->d_info.p, (const ReadOff*)s->d_off.p, n_pre,
                                (const uint8_t*)s->d_scratch.p, o);
  ING_TRY(hipGetLastError());
  ING_TRY(hipMemcpyAsync(D + a_lrb, s->d_lrb.p, ((size_t)nl + 1) * 8, hipMemcpyDeviceToDevice, st));
  const double t4 = now();
  ING_TRY(hipMemcpyAsync(H, D, at, hipMemcpyDeviceToHost, st));
  ING_TRY(slot_wait(s));
  {  // per locus counters: known to the host since the walk
    int32_t* nf = (int32_t*)(H + a_nfilt); int64_t* ns = (int64_t*)(H + a_nseen);
    for (uint32_t l = 0; l < nl; ++l) { nf[l] = (int32_t)h_walk[l].n_filt; ns[l] = (int64_t)h_walk[l].n_kept; }
  }
  HostOut& R = out.out;
  R.n_reads = (int64_t)nr; R.read_bytes = n_bytes; R.name_bytes = n_name; R.snp_n = n_snp; R.meth_n = n_meth; R.cig_n = n_cig; R.bam4_bytes = n_bam4;
  R.lrb = (const uint64_t*)(H + a_lrb); R.n_filt = (const int32_t*)(H + a_nfilt); R.n_seen = (const int64_t*)(H + a_nseen);
  R.read_off = (const uint64_t*)(H + a_roff); R.read_len = (const uint32_t*)(H + a_rlen); R.reads = H + a_reads; R.quals = H + a_quals; R.names = (const char*)(H + a_names);
  R.name_off = (const uint64_t*)(H + a_noff); R.rq = (const double*)(H + a_rq); R.is_reverse = H + a_rev; R.mapq = H + a_mapq; R.hp = (const int16_t*)(H + a_hp);
  R.start_offset = (const int32_t*)(H + a_so); R.end_offset = (const int32_t*)(H + a_eo); R.snp = (const int32_t*)(H + a_snp); R.snp_off = (const uint64_t*)(H + a_soff);
  R.meth = H + a_meth; R.meth_off = (const uint64_t*)(H + a_moff); R.has_meth = H + a_hm; R.cig = (const uint32_t*)(H + a_cig); R.cig_off = (const uint64_t*)(H + a_coff);
  R.cig_ref_pos = (const int64_t*)(H + a_cpos);
  if (in.keep_bam4) { R.bam4 = H + a_b4; R.bam4_off = (const uint64_t*)(H + a_b4off); }
  R.dev_reads = D + a_reads;
  const double t5 = now();
  out.ms_upload = t4 - t3; out.ms_inflate = t1 - t0; out.ms_walk = t2 - t1; out.ms_reads = t3 - t2; out.ms_download = t5 - t4;
  return TRGT_OK;
}

}  // namespace ingd
}  // namespace trgt
