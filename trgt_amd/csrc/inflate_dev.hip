// trgt_amd/csrc/inflate_dev.hip -- raw DEFLATE (RFC 1951) of many independent blocks on the device: the BGZF blocks of a BAM.
//
// Replaces, for the read ingestion of SURVEY.md 8(f) row 3, what the reference does inside htslib's bgzf_read_block (rust-htslib ->
// htslib -> zlib inflate; src/trgt/workflows/tr.rs:268-305 reaches it through bam::IndexedReader::fetch / records()): every BGZF block
// is a DEFLATE stream of its own (at most 64 KB inflated), so a chunk of loci is a few thousand independent streams.
//
// One wave per block, blocks claimed from a counter.  The Huffman decoding itself is a serial dependent chain (the position of a code
// depends on the length of the one before): every lane runs it on the same values, which are kept wave-uniform on purpose (what comes
// back from LDS goes through v_readfirstlane) so that the decoder's state lives in SGPRs, its arithmetic is scalar and its branches are
// scalar branches -- no exec-mask bookkeeping around a one-lane loop; what parallelises is done by the whole wave at rendezvous points of a
// uniform loop: loading the next window of compressed bytes into LDS, filling the lookup tables of a dynamic block, copying stored
// blocks, and flushing finished 4-KB segments of the output.  The last 8 KB of output live in an LDS ring (matches are byte copies by
// the whole wave: from the ring, or -- the few that reach further back, up to 32 KB -- from the output already flushed to HBM), so a block
// needs 16 KB of LDS and ten blocks are in flight per CU: the decode loop is latency-bound (a table look-up per symbol), and the
// number of streams in flight is what sets the rate (32 KB of ring: four per CU, 7.7 GB/s).
// A stream this decoder does not like (bad code lengths, distance too far back, output not exactly the announced size) is DECLINED
// (status 0): the caller hands that block to zlib, which produces the data or the error message.
#include <cstdlib>

#include "common.hpp"
#include "inflate_dev.hpp"

namespace trgt {
namespace infl {

#ifndef TRGT_INFL_RING
#define TRGT_INFL_RING 2048
#endif
#if TRGT_INFL_RING == 4096
#define INFL_RMASK "0xfff"
#define INFL_RNEAR "0xfc0"   /* RING - 64 */
#define INFL_RSIZE "0x1000"
#elif TRGT_INFL_RING == 2048
#define INFL_RMASK "0x7ff"
#define INFL_RNEAR "0x7c0"
#define INFL_RSIZE "0x800"
#else
#error "TRGT_INFL_RING: 2048 or 4096"
#endif
constexpr uint32_t RING = TRGT_INFL_RING, RING_MASK = RING - 1, SEG = RING / 2;  // the last 4 KB of output in LDS (2 KB until the hand-written loop: a fifth of a BAM block's matches reach further back than that, an eighth further than 4 KB); older bytes are read back from the flushed output
constexpr uint32_t IN_WIN = RING == 2048 ? 896 : 1024;  // compressed bytes staged in LDS (with the 2-KB ring a block's LDS is 10 240 bytes: sixteen blocks per CU)
constexpr uint32_t HDR_ROOM = 576;       // a dynamic block header (<= 14 + 19 * 3 + 320 * 14 bits = 569 bytes) is parsed without a reload in between
constexpr int LIT_BITS = 10, DIST_BITS = 8;

enum : uint32_t { EV_NONE = 0, EV_RELOAD = 1, EV_FLUSH = 2, EV_BUILD = 3, EV_COPY = 4, EV_DONE = 5, EV_ERROR = 6 };

struct Shared {
  alignas(16) uint8_t out[RING];  // first: at LDS address 0 (the kernel's only LDS object), which the hand-written loop relies on (checked at the kernel's start)
  uint32_t ev;            // what lane 0 asks the wave for
  uint32_t win_base;      // source offset of in_win[0]
  uint32_t op;            // bytes produced so far
  uint32_t copy_src, copy_dst, copy_len;  // EV_COPY: stored bytes src[copy_src ..) -> out[copy_dst ..)
  uint32_t nlit, ndist;   // EV_BUILD
  uint32_t block;         // the claimed block
  uint32_t lit32[1 << LIT_BITS], dist32[1 << DIST_BITS];      // what the symbol loop reads: see lit_entry / dist_entry
  uint16_t lit_cnt[16], dist_cnt[16];     // codes per length (canonical decoding of the long codes, and the checks)
  uint16_t lit_sym[288], dist_sym[32];    // symbols in code order
  uint16_t code[320];                     // canonical code of every symbol (table fill)
  uint8_t lens[320], lens2[320];
  uint16_t offs[16], next[16];            // prepare_codes
  alignas(16) uint8_t in_win[IN_WIN + 32];  // (+ what the refills of one symbol may read past a window that already reaches the end of the input)
};

#define RFL(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))

__device__ const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};  // RFC 1951 3.2.7

__device__ __forceinline__ uint32_t rev_bits(uint32_t v, int n) { return __builtin_bitreverse32(v) >> (32 - n); }

// Table entries of the symbol loop (round 6: the loop is bound by scalar instruction issue -- one SALU instruction per cycle and CU -- so
// everything a symbol implies is worked out once per table, not once per symbol).  Bits 0-3: bits of the code (0: longer than the table's
// index, decoded canonically); literal / length table: F_LIT with the byte in 16-23, F_PAIR when a second literal's code fits the index as
// well (its byte in 24-31, bits 0-3 then count both codes), F_EOB, else a length: extra bits in 4-7, base length in 16-31; distance table:
// extra bits in 4-7, base distance in 16-31.  F_BAD: a symbol DEFLATE does not define.
constexpr uint32_t F_LIT = 1u << 8, F_EOB = 1u << 9, F_PAIR = 1u << 10, F_BAD = 1u << 11;
__device__ __forceinline__ uint32_t lit_entry(uint32_t sym, uint32_t l) {
  if (sym < 256u) return l | F_LIT | (sym << 16);
  if (sym == 256u) return l | F_EOB;
  if (sym > 285u) return l | F_BAD;
  if (sym < 265u) return l | ((sym - 254u) << 16);
  if (sym == 285u) return l | (258u << 16);
  const uint32_t eb = (sym - 261u) >> 2;
  return l | (eb << 4) | ((3u + ((4u + ((sym - 265u) & 3u)) << eb)) << 16);
}
__device__ __forceinline__ uint32_t dist_entry(uint32_t ds, uint32_t l) {
  if (ds > 29u) return l | F_BAD;
  if (ds < 4u) return l | ((ds + 1u) << 16);
  const uint32_t eb = (ds >> 1) - 1u;
  return l | (eb << 4) | ((((2u + (ds & 1u)) << eb) + 1u) << 16);
}

// canonical decoding bit by bit (codes longer than the table's bits: rare symbols by construction)
__device__ __forceinline__ int slow_decode(const uint16_t* cnt, const uint16_t* sym, uint64_t bits, int& len_out) {
  int code = 0, first = 0, index = 0;
#pragma unroll 1
  for (int len = 1; len <= 15; ++len) {
    code |= (int)(bits & 1); bits >>= 1;
    const int c = (int)RFL(cnt[len]);
    if (code - c < first) { len_out = len; return (int)RFL(sym[index + (code - first)]); }
    index += c; first += c; first <<= 1; code <<= 1;
  }
  len_out = 0;
  return -1;
}

// code lengths -> counts, symbols in code order, canonical codes; false: over-subscribed or incomplete (a single distance code is
// allowed to be incomplete, RFC 1951 3.2.7)
__device__ inline bool prepare_codes(const uint8_t* lens, int n, uint16_t* cnt, uint16_t* sym, uint16_t* code, bool dist, uint16_t* offs, uint16_t* next) {
  // (every lane of the wave runs this on the same values: the read-modify-writes of LDS below happen in lockstep)
  for (int l = 0; l < 16; ++l) cnt[l] = 0;
  for (int s = 0; s < n; ++s) { const uint32_t l = RFL(lens[s]); cnt[l] = (uint16_t)(RFL(cnt[l]) + 1u); }
  if ((int)RFL(cnt[0]) == n) return dist;  // no codes at all: fine for distances of an all-literal block
  int left = 1;
  for (int l = 1; l <= 15; ++l) { left <<= 1; left -= (int)RFL(cnt[l]); if (left < 0) return false; }
  if (left > 0 && !(dist && n - (int)RFL(cnt[0]) == 1)) return false;
  uint32_t o = 0, c = 0;
  for (int l = 1; l <= 15; ++l) { offs[l] = (uint16_t)o; next[l] = (uint16_t)c; const uint32_t k = RFL(cnt[l]); o += k; c = (c + k) << 1; }
  for (int s = 0; s < n; ++s) {
    const uint32_t l = RFL(lens[s]);
    if (l) { const uint32_t at = RFL(offs[l]), cd = RFL(next[l]); sym[at] = (uint16_t)s; code[s] = (uint16_t)cd; offs[l] = (uint16_t)(at + 1); next[l] = (uint16_t)(cd + 1); }
  }
  return true;
}

// The symbol loop by hand (round 6, second half).  The compiler's scalar code for the loop costs 67 SALU instructions per symbol (a CU
// issues ONE per cycle) -- flag registers for every merge of the nested branches, 64-bit window arithmetic for bit-buffer refills at
// arbitrary byte positions, five separate range checks per match -- and every match that reaches beyond the ring (a fifth of the matches
// of a BAM block: distances are spread evenly up to 32 KB) went through the general copy with its modulo arithmetic.  This loop keeps
// the decoder's state in fixed scalar registers and handles what almost every symbol is: a literal (or a pair), a match of at most 64
// bytes from the LDS ring (overlapping its destination or not), a match of at most 64 bytes from the flushed output -- ~ 18 scalar
// instructions per literal iteration, ~ 50 per match.  Everything else (codes longer than the table's index, end of block, undefined
// symbols, copies of more than 64 bytes, sources that straddle the ring's edge, a window that runs low) LEAVES the loop with the state
// of the reference decoder intact and says where it stopped; the C++ code does that one symbol.
//   Preconditions (the caller's): in_pos is a multiple of 4 (refills are aligned dword reads of the window), op < stop <= out_len - 258
//   (no symbol can pass the end of the output), stop at most the next segment boundary (the ring is flushed segment by segment).
//   Returns 0: op reached stop (at a symbol boundary); 1: at a symbol boundary, the next symbol is the caller's (or the window is low);
//   2: a length has been decoded into len, its distance code is the next thing in the bit buffer; 3: len and dist decoded, copy not done.
__device__ __forceinline__ uint32_t fast_symbols(uint64_t& bitbuf, uint32_t& bitcnt, uint32_t& ip_lds, uint32_t& op, uint32_t& len, uint32_t& dist, uint32_t ip_limit,
                                                 uint32_t op_stop, uint32_t lit_base, uint32_t dist_base, uint32_t lane, const uint8_t* outp) {
  uint32_t code, e, n, t0, t1, pd, v0, v1, v2, vn, vpd, vpa;
  static_assert(LIT_BITS == 10 && DIST_BITS == 8, "constants of the hand-written loop");
  // COMPLETE: the copy from the flushed output that is still in flight (pd != -1: its destination is [pd, pd + its length), the loaded
  // bytes arrive in vpd, their ring addresses are in vpa, its lanes in s[68:69]) is written to the ring
#define INFL_COMPLETE                      \
  "s_waitcnt vmcnt(0)\n\t"                 \
  "s_mov_b64 s[66:67], exec\n\t"           \
  "s_mov_b64 exec, s[68:69]\n\t"           \
  "ds_write_b8 %[vpa], %[vpd]\n\t"         \
  "s_mov_b64 exec, s[66:67]\n\t"           \
  "s_mov_b32 %[pd], -1\n\t"
  // REFILL: 32 more bits.  The dword was asked for when the one before it was consumed (vn), so the LDS round trip is over by now; the
  // next one is asked for right away
#define INFL_REFILL                                  \
  "s_waitcnt lgkmcnt(0)\n\t"                         \
  "v_readfirstlane_b32 s64, %[vn]\n\t"               \
  "s_add_i32 %[ip], %[ip], 4\n\t"                    \
  "v_mov_b32 %[v1], %[ip]\n\t"                       \
  "ds_read_b32 %[vn], %[v1]\n\t"                     \
  "s_lshl_b64 s[62:63], s[64:65], %[bc]\n\t"         \
  "s_or_b32 %[bc], %[bc], 32\n\t"                    \
  "s_or_b64 s[60:61], s[60:61], s[62:63]\n"
  asm volatile(
      "s_mov_b64 s[60:61], %[bb]\n\t"
      "s_mov_b32 s65, 0\n\t"
      "s_mov_b32 %[code], 0\n\t"
      "s_mov_b32 %[pd], -1\n\t"
      "s_mov_b32 s70, 0\n\t"                  // end of the destination of the copy in flight
      "v_mov_b32 %[v1], %[ip]\n\t"
      "ds_read_b32 %[vn], %[v1]\n"
      ".Ltop%=:\n\t"
      "s_cmp_gt_u32 %[bc], 31\n\t"
      "s_cbranch_scc1 .Llook%=\n"
      // ---- refill at a symbol boundary (the window must not be low)
      ".Lfill%=:\n\t"
      "s_cmp_gt_u32 %[ip], %[iplim]\n\t"
      "s_cbranch_scc1 .Lexit1%=\n\t"
      INFL_REFILL
      ".Llook%=:\n\t"
      "s_and_b32 %[t0], s60, 0x3ff\n\t"
      "s_lshl2_add_u32 %[t0], %[t0], %[litb]\n\t"
      "v_mov_b32 %[v0], %[t0]\n\t"
      "ds_read_b32 %[v0], %[v0]\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_readfirstlane_b32 %[e], %[v0]\n\t"
      "s_and_b32 %[n], %[e], 15\n\t"
      "s_cbranch_scc0 .Lexit1%=\n\t"          // a code longer than the index: the caller's
      "s_bitcmp1_b32 %[e], 8\n\t"             // F_LIT
      "s_cbranch_scc0 .Lnolit%=\n\t"
      // ---- one literal (bits 16-23 of the entry, still in v0) or two (F_PAIR: the second in bits 24-31)
      "s_and_b32 %[t0], %[op], " INFL_RMASK "\n\t"
      "v_mov_b32 %[v1], %[t0]\n\t"
      "ds_write_b8_d16_hi %[v1], %[v0]\n\t"
      "s_add_i32 %[op], %[op], 1\n\t"
      "s_bitcmp1_b32 %[e], 10\n\t"            // F_PAIR
      "s_cbranch_scc0 .Llitdone%=\n\t"
      "s_and_b32 %[t0], %[op], " INFL_RMASK "\n\t"
      "v_mov_b32 %[v1], %[t0]\n\t"
      "v_lshrrev_b32 %[v2], 24, %[v0]\n\t"
      "ds_write_b8 %[v1], %[v2]\n\t"
      "s_add_i32 %[op], %[op], 1\n"
      ".Llitdone%=:\n\t"
      "s_lshr_b64 s[60:61], s[60:61], %[n]\n\t"
      "s_sub_i32 %[bc], %[bc], %[n]\n\t"
      "s_cmp_lt_u32 %[op], %[opstop]\n\t"
      "s_cbranch_scc0 .Lout%=\n\t"
      "s_cmp_gt_u32 %[bc], 31\n\t"
      "s_cbranch_scc1 .Llook%=\n\t"
      "s_branch .Lfill%=\n"
      ".Lnolit%=:\n\t"                        // (end of block, undefined symbols and lengths that can pass 64 have empty entries: they left above)
      // ---- a length: base in bits 16-31, extra bits in 4-7
      "s_lshr_b64 s[60:61], s[60:61], %[n]\n\t"
      "s_sub_i32 %[bc], %[bc], %[n]\n\t"
      "s_lshr_b32 %[len], %[e], 16\n\t"
      "s_bfe_u32 %[t0], %[e], 0x40004\n\t"
      "s_cbranch_scc0 .Llenok%=\n\t"         // (lengths 3 .. 10 have no extra bits: most matches)
      "s_bfm_b32 %[t1], %[t0], 0\n\t"
      "s_and_b32 %[t1], s60, %[t1]\n\t"
      "s_add_i32 %[len], %[len], %[t1]\n\t"
      "s_lshr_b64 s[60:61], s[60:61], %[t0]\n\t"
      "s_sub_i32 %[bc], %[bc], %[t0]\n"
      ".Llenok%=:\n\t"
      // a distance code and its extra bits take at most 15 + 13 bits
      "s_cmp_gt_u32 %[bc], 27\n\t"
      "s_cbranch_scc1 .Ldist%=\n\t"
      "s_cmp_gt_u32 %[ip], %[iplim]\n\t"     // (EVERY refill asks whether the window is low: a run of matches of ~ 32 bits each refills here symbol
      "s_cbranch_scc1 .Lexit2%=\n\t"         //  after symbol and seldom at a boundary -- unchecked, such a run can walk out of the window)
      INFL_REFILL
      ".Ldist%=:\n\t"
      "s_and_b32 %[t0], s60, 0xff\n\t"
      "s_lshl2_add_u32 %[t0], %[t0], %[distb]\n\t"
      "v_mov_b32 %[v0], %[t0]\n\t"
      "ds_read_b32 %[v0], %[v0]\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_readfirstlane_b32 %[e], %[v0]\n\t"
      "s_and_b32 %[n], %[e], 15\n\t"
      "s_cbranch_scc0 .Lexit2%=\n\t"          // a long distance code
      "s_lshr_b64 s[60:61], s[60:61], %[n]\n\t"
      "s_sub_i32 %[bc], %[bc], %[n]\n\t"
      "s_bfe_u32 %[t0], %[e], 0x40004\n\t"
      "s_bfm_b32 %[t1], %[t0], 0\n\t"
      "s_and_b32 %[t1], s60, %[t1]\n\t"
      "s_lshr_b32 %[dist], %[e], 16\n\t"
      "s_add_i32 %[dist], %[dist], %[t1]\n\t"
      "s_lshr_b64 s[60:61], s[60:61], %[t0]\n\t"
      "s_sub_i32 %[bc], %[bc], %[t0]\n\t"
      "s_cmp_gt_u32 %[dist], %[op]\n\t"       // before the start of the output: the caller reports it
      "s_cbranch_scc1 .Lexit3%=\n\t"
      "s_cmp_lt_u32 %[dist], %[len]\n\t"
      "s_cbranch_scc1 .Lslow%=\n\t"           // a source that overlaps its destination (len <= 58 here: longer ones never enter)
      "s_cmpk_gt_u32 %[dist], " INFL_RNEAR "\n\t"  // RING - 64
      "s_cbranch_scc1 .Lfar%=\n\t"
      // ---- the copy in one round by the whole wave: len <= 64, len <= dist (no overlap), dist + len <= RING (the source is in the ring)
      "s_sub_i32 %[t0], %[op], %[dist]\n\t"
      // (a source that touches the destination of the copy in flight waits for it: [t0, t0 + len) against [pd, s70); pd = -1: none)
      "s_add_i32 %[t1], %[t0], %[len]\n\t"
      "s_cmp_gt_u32 %[t1], %[pd]\n\t"
      "s_cbranch_scc0 .Lfree%=\n\t"
      "s_cmp_lt_u32 %[t0], s70\n\t"
      "s_cbranch_scc0 .Lfree%=\n\t"
      INFL_COMPLETE
      ".Lfree%=:\n\t"
      "v_cmp_gt_u32 vcc, %[len], %[lane]\n\t"
      "s_and_saveexec_b64 s[66:67], vcc\n\t"
      "v_add_u32 %[v0], %[t0], %[lane]\n\t"
      "v_and_b32 %[v0], " INFL_RMASK ", %[v0]\n\t"
      "ds_read_u8 %[v2], %[v0]\n\t"
      "v_add_u32 %[v1], %[op], %[lane]\n\t"
      "v_and_b32 %[v1], " INFL_RMASK ", %[v1]\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "ds_write_b8 %[v1], %[v2]\n\t"
      "s_mov_b64 exec, s[66:67]\n"
      ".Lcopied%=:\n\t"
      "s_add_i32 %[op], %[op], %[len]\n\t"
      "s_cmp_lt_u32 %[op], %[opstop]\n\t"
      "s_cbranch_scc0 .Lout%=\n\t"
      "s_cmp_gt_u32 %[bc], 31\n\t"
      "s_cbranch_scc1 .Llook%=\n\t"
      "s_branch .Lfill%=\n"
      // ---- a source beyond the ring: dist >= RING means every byte of it has been flushed (segments go out as they fill, and the flush is
      //      fenced), so the wave reads the output it wrote itself; RING - 64 < dist < RING may straddle the ring's edge: the caller's.
      //      The load is NOT waited for (a round trip to L2 / HBM is worth several symbols): the bytes are written to the ring when a
      //      later copy reads their destination, when the next such load is issued, or when the loop is left
      ".Lfar%=:\n\t"
      "s_cmpk_lt_u32 %[dist], " INFL_RSIZE "\n\t"
      "s_cbranch_scc1 .Lexit3%=\n\t"
      "s_cmp_eq_u32 %[pd], -1\n\t"
      "s_cbranch_scc1 .Lfargo%=\n\t"
      INFL_COMPLETE
      ".Lfargo%=:\n\t"
      "s_sub_i32 %[t0], %[op], %[dist]\n\t"
      "v_cmp_gt_u32 vcc, %[len], %[lane]\n\t"
      "s_and_saveexec_b64 s[66:67], vcc\n\t"
      "s_mov_b64 s[68:69], exec\n\t"
      "v_add_u32 %[v0], %[t0], %[lane]\n\t"
      "global_load_ubyte %[vpd], %[v0], %[outp] sc1\n\t"
      "v_add_u32 %[vpa], %[op], %[lane]\n\t"
      "v_and_b32 %[vpa], " INFL_RMASK ", %[vpa]\n\t"
      "s_mov_b64 exec, s[66:67]\n\t"
      "s_mov_b32 %[pd], %[op]\n\t"
      "s_add_i32 s70, %[op], %[len]\n\t"
      "s_branch .Lcopied%=\n"
      // ---- dist < len <= 64: the dist bytes in front of op repeat.  Round r copies the bytes [dist (2^r - 1), dist (2^(r+1) - 1)) from
      //      dist 2^r places before them -- a multiple of the period, and everything that far back is written: log2(len / dist) rounds
      ".Lslow%=:\n\t"
      "s_cmp_eq_u32 %[pd], -1\n\t"            // (dist < 64: the source may well be what the copy in flight writes)
      "s_cbranch_scc1 .Lrounds%=\n\t"
      INFL_COMPLETE
      ".Lrounds%=:\n\t"
      "s_mov_b32 %[t0], 0\n\t"                // done
      "s_mov_b32 %[t1], %[dist]\n"            // step
      ".Lround%=:\n\t"
      "s_add_i32 %[n], %[t0], %[t1]\n\t"
      "s_min_u32 %[n], %[n], %[len]\n\t"
      "v_cmp_le_u32 vcc, %[t0], %[lane]\n\t"
      "s_and_saveexec_b64 s[66:67], vcc\n\t"
      "v_cmp_gt_u32 vcc, %[n], %[lane]\n\t"
      "s_and_b64 exec, exec, vcc\n\t"
      "v_add_u32 %[v1], %[op], %[lane]\n\t"
      "v_subrev_u32 %[v0], %[t1], %[v1]\n\t"
      "v_and_b32 %[v0], " INFL_RMASK ", %[v0]\n\t"
      "ds_read_u8 %[v2], %[v0]\n\t"
      "v_and_b32 %[v1], " INFL_RMASK ", %[v1]\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "ds_write_b8 %[v1], %[v2]\n\t"
      "s_mov_b64 exec, s[66:67]\n\t"
      "s_mov_b32 %[t0], %[n]\n\t"
      "s_lshl_b32 %[t1], %[t1], 1\n\t"
      "s_cmp_lt_u32 %[t0], %[len]\n\t"
      "s_cbranch_scc1 .Lround%=\n\t"
      "s_branch .Lcopied%=\n"
      ".Lexit3%=:\n\t"
      "s_mov_b32 %[code], 3\n\t"
      "s_branch .Lout%=\n"
      ".Lexit2%=:\n\t"
      "s_mov_b32 %[code], 2\n\t"
      "s_branch .Lout%=\n"
      ".Lexit1%=:\n\t"
      "s_mov_b32 %[code], 1\n"
      ".Lout%=:\n\t"
      "s_cmp_eq_u32 %[pd], -1\n\t"
      "s_cbranch_scc1 .Lleave%=\n\t"
      INFL_COMPLETE
      ".Lleave%=:\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"              // (the dword asked for ahead of time: nobody may find it in flight)
      "s_mov_b64 %[bb], s[60:61]\n\t"
      : [bb] "+s"(bitbuf), [bc] "+s"(bitcnt), [ip] "+s"(ip_lds), [op] "+s"(op), [len] "+s"(len), [dist] "+s"(dist), [code] "=&s"(code), [e] "=&s"(e), [n] "=&s"(n),
        [t0] "=&s"(t0), [t1] "=&s"(t1), [pd] "=&s"(pd), [v0] "=&v"(v0), [v1] "=&v"(v1), [v2] "=&v"(v2), [vn] "=&v"(vn), [vpd] "=&v"(vpd), [vpa] "=&v"(vpa)
      : [iplim] "s"(ip_limit), [opstop] "s"(op_stop), [litb] "s"(lit_base), [distb] "s"(dist_base), [lane] "v"(lane), [outp] "s"(outp)
      : "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "vcc", "scc", "memory");
#undef INFL_COMPLETE
#undef INFL_REFILL
  return code;
}

// (Tried and not kept, profiles/r06h_ab_inflate_gather.txt: the table look-ups of SEVERAL symbols in one LDS round trip -- lane 32 + j reads the
//  literal / length entry for the bits behind the first j of the buffer, lane j the distance entry, the entries of a round of symbols come
//  out of that register by v_readlane, the buffer is shifted once per round.  Same instruction count per symbol, 7.46 against 7.53 ms per
//  launch: what a wave pays per symbol is instruction issue next to its neighbours on the SIMD, not the look-ups' round trips.)
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) inflate_blocks_kernel(const uint8_t* __restrict__ src, const BlockDesc* __restrict__ blocks, uint32_t n_blocks,
                                                            uint8_t* __restrict__ dst, uint8_t* __restrict__ status, unsigned int* __restrict__ counter, uint32_t fast) {
  __shared__ Shared sh;
  const int lane = threadIdx.x;
  typedef __attribute__((address_space(3))) void* lds_ptr;  // LDS byte addresses for the hand-written loop
  const uint32_t win_lds = (uint32_t)(uintptr_t)(lds_ptr)sh.in_win, lit_lds = (uint32_t)(uintptr_t)(lds_ptr)sh.lit32, dist_lds = (uint32_t)(uintptr_t)(lds_ptr)sh.dist32,
                 ring_lds = (uint32_t)(uintptr_t)(lds_ptr)sh.out;
  if (ring_lds != 0u) fast = 0u;  // (the hand-written loop addresses the ring without a base)
  for (;;) {
    __syncthreads();
    if (lane == 0) sh.block = atomicAdd(counter, 1u);
    __syncthreads();
    const uint32_t bi = RFL(sh.block);
    if (bi >= n_blocks) return;
    const BlockDesc bd = blocks[bi];
    const uint8_t* __restrict__ in = src + bd.src_off;
    uint8_t* __restrict__ outp = dst + bd.dst_off;
    const uint32_t in_len = bd.src_len, out_len = bd.dst_len;
    // lane 0's decoder state (registers; the other lanes carry copies they never use)
    uint64_t bitbuf = 0; uint32_t bitcnt = 0, in_pos = 0, op = 0, wb = 0, pending_skip = 0, stored_left = 0;
    int phase = 0;           // 0: at a block header, 1: inside a Huffman block, 2: inside a stored block
    bool final_block = false, have_window = false;
    uint32_t flushed = 0;    // uniform
    bool ok = out_len <= 65536u;
    for (;;) {
      // ---- the decoder runs (on every lane, on uniform values) until it needs the wave
      {
        uint32_t want = ok ? (uint32_t)EV_NONE : (uint32_t)EV_ERROR;
        auto refill = [&]() {
          if (bitcnt < 32) {
            const uint32_t o = in_pos - wb;
            const uint32_t a = RFL(*reinterpret_cast<const uint32_t*>(sh.in_win + (o & ~3u))), b = RFL(*reinterpret_cast<const uint32_t*>(sh.in_win + (o & ~3u) + 4));
            const uint32_t w = (uint32_t)((((uint64_t)b << 32) | a) >> (8 * (o & 3u)));
            bitbuf |= (uint64_t)w << bitcnt; bitcnt += 32; in_pos += 4;
          }
        };
        auto take = [&](uint32_t n) { const uint32_t v = (uint32_t)(bitbuf & ((1ull << n) - 1ull)); bitbuf >>= n; bitcnt -= n; return v; };
        // the window must hold `room` more bytes (unless it already reaches the end of the input: zeros follow)
        auto window_low = [&](uint32_t room) { return in_pos + room > wb + IN_WIN && wb + IN_WIN < in_len + 8; };
        if (!have_window) { have_window = true; if (want == EV_NONE) want = EV_RELOAD; }
        else if (pending_skip) {  // behind a reload: the consumed bits of the first byte are dropped again
          const uint32_t byte = RFL(sh.in_win[in_pos - wb]);
          bitbuf = (uint64_t)(byte >> pending_skip); bitcnt = 8u - pending_skip; in_pos += 1; pending_skip = 0;
        }
        while (want == EV_NONE) {
          if (in_pos > in_len + 8) { want = EV_ERROR; break; }
          if (phase == 0) {
            // ---- block header (a dynamic one is parsed without a reload in between)
            if (window_low(HDR_ROOM)) { want = EV_RELOAD; break; }
            refill();
            final_block = take(1) != 0;
            const uint32_t type = take(2);
            if (type == 0) {
              take(bitcnt & 7u);  // to the byte boundary
              refill();
              const uint32_t len = take(16), nlen = take(16);
              if ((len ^ nlen) != 0xFFFFu) { want = EV_ERROR; break; }
              in_pos -= bitcnt >> 3; bitbuf = 0; bitcnt = 0;  // first byte of the data
              if (in_pos + len > in_len || op + len > out_len) { want = EV_ERROR; break; }
              stored_left = len; phase = 2;
            } else if (type == 1 || type == 2) {
              uint32_t nlit, ndist;
              if (type == 1) {
                nlit = 288; ndist = 32;
                for (int s = 0; s < 144; ++s) sh.lens[s] = 8;
                for (int s = 144; s < 256; ++s) sh.lens[s] = 9;
                for (int s = 256; s < 280; ++s) sh.lens[s] = 7;
                for (int s = 280; s < 288; ++s) sh.lens[s] = 8;
                for (int s = 0; s < 32; ++s) sh.lens[288 + s] = 5;
              } else {
                refill();
                nlit = take(5) + 257; ndist = take(5) + 1; const uint32_t ncode = take(4) + 4;
                if (nlit > 286 || ndist > 30) { want = EV_ERROR; break; }
                for (int i = 0; i < 19; ++i) sh.lens[i] = 0;
                for (uint32_t i = 0; i < ncode; ++i) { refill(); sh.lens[RFL(kClOrder[i])] = (uint8_t)take(3); }
                if (in_pos > in_len + 8) { want = EV_ERROR; break; }  // (a truncated stream: the header ran past the input -- ADVICE r4)
                // the code-length code (19 symbols, at most 7 bits), decoded canonically from counts kept in the distance arrays
                if (!prepare_codes(sh.lens, 19, sh.dist_cnt, sh.dist_sym, sh.code, false, sh.offs, sh.next)) { want = EV_ERROR; break; }
                uint32_t idx = 0; bool bad = false;
                while (idx < nlit + ndist) {
                  if (in_pos > in_len + 8) { bad = true; break; }  // (the window's room covers an intact header only: stop at the end of the input)
                  refill();
                  int l; const int sym = slow_decode(sh.dist_cnt, sh.dist_sym, bitbuf, l);
                  if (sym < 0) { bad = true; break; }
                  take((uint32_t)l);
                  if (sym < 16) sh.lens2[idx++] = (uint8_t)sym;
                  else {
                    uint32_t rep, val = 0;
                    if (sym == 16) { if (idx == 0) { bad = true; break; } val = RFL(sh.lens2[idx - 1]); rep = 3 + take(2); }
                    else if (sym == 17) rep = 3 + take(3);
                    else rep = 11 + take(7);
                    if (idx + rep > nlit + ndist) { bad = true; break; }
                    while (rep--) sh.lens2[idx++] = (uint8_t)val;
                  }
                }
                if (bad || RFL(sh.lens2[256]) == 0) { want = EV_ERROR; break; }
                for (uint32_t i = 0; i < nlit + ndist; ++i) sh.lens[i] = (uint8_t)RFL(sh.lens2[i]);
              }
              if (!prepare_codes(sh.lens, (int)nlit, sh.lit_cnt, sh.lit_sym, sh.code, false, sh.offs, sh.next) ||
                  !prepare_codes(sh.lens + nlit, (int)ndist, sh.dist_cnt, sh.dist_sym, sh.code + nlit, true, sh.offs, sh.next)) { want = EV_ERROR; break; }
              sh.nlit = nlit; sh.ndist = ndist;
              phase = 1;
              want = EV_BUILD;
            } else { want = EV_ERROR; break; }
          } else if (phase == 2) {
            // ---- stored bytes, one piece (up to the next segment boundary) per rendezvous
            if (stored_left == 0) { phase = 0; if (final_block) want = EV_DONE; continue; }
            const uint32_t room = SEG - (op & (SEG - 1));
            const uint32_t piece = stored_left < room ? stored_left : room;
            sh.copy_src = in_pos; sh.copy_dst = op; sh.copy_len = piece;
            in_pos += piece; op += piece; stored_left -= piece;
            want = EV_COPY;
          } else {
            // ---- symbols of a Huffman block: a loop of its own that is left only for an event (its iterations carry no state-machine
            // bookkeeping: the symbol loop is bound by scalar instruction issue).  One bound covers "the window runs low" and "the input is over"
            // (written with one back edge and one exit: breaks out of nested branches make the compiler carry flag registers through
            //  every iteration)
            const bool to_end = wb + IN_WIN >= in_len + 8;
            const uint32_t safe_end = to_end ? in_len + 8 : wb + IN_WIN - 16;
            const uint32_t low_ev = to_end ? (uint32_t)EV_ERROR : (uint32_t)EV_RELOAD;
            uint32_t ev = in_pos > safe_end ? low_ev : (uint32_t)EV_NONE;
            while (ev == EV_NONE) {
              const uint32_t before = op;
              uint32_t stage = 1, len = 0, dist = 0;  // 1: at a symbol boundary; 2: a length is decoded, its distance code comes next; 3: length and distance decoded
              if (fast && op + 258u < out_len) {
                // the hand-written loop (fast_symbols) wants in_pos on a dword boundary: whole bytes go into the bit buffer until it is
                while ((in_pos & 3u) && bitcnt <= 56u) { bitbuf |= (uint64_t)RFL(sh.in_win[in_pos - wb]) << bitcnt; bitcnt += 8; ++in_pos; }
                if (!(in_pos & 3u)) {
                  uint32_t ip = win_lds + (in_pos - wb);
                  const uint32_t seg_end = (op | (SEG - 1u)) + 1u, tail = out_len - 258u;
                  stage = fast_symbols(bitbuf, bitcnt, ip, op, len, dist, win_lds + (safe_end - wb), seg_end < tail ? seg_end : tail, lit_lds, dist_lds, (uint32_t)lane, outp);
                  in_pos = wb + (ip - win_lds);
                }
              }
              if (stage == 1) {
                if (in_pos > safe_end) ev = low_ev;  // (the loop above stopped for the window)
                else {
                  refill();  // (>= 32 bits: a literal / length code and its extra bits take at most 15 + 5)
                  uint32_t e = RFL(sh.lit32[bitbuf & ((1u << LIT_BITS) - 1u)]);
                  if (!(e & 15u)) { int l; const int sym = slow_decode(sh.lit_cnt, sh.lit_sym, bitbuf, l); e = sym < 0 ? (F_BAD | 1u) : lit_entry((uint32_t)sym, (uint32_t)l); }
                  take(e & 15u);
                  if (e & F_LIT) {
                    sh.out[op & RING_MASK] = (uint8_t)(e >> 16); ++op;
                    if (e & F_PAIR) { sh.out[op & RING_MASK] = (uint8_t)(e >> 24); ++op; }
                  } else if (e & (F_EOB | F_BAD)) {
                    if (e & F_BAD) ev = EV_ERROR;
                    else { phase = 0; ev = final_block ? (uint32_t)EV_DONE : (uint32_t)EV_BUILD; }  // (EV_BUILD stands for "leave the loop, nothing to do": reset below)
                  } else {
                    len = (e >> 16) + take((e >> 4) & 15u);
                    stage = 2;
                  }
                }
              }
              if (stage == 2) {
                refill();  // (a distance code and its extra bits: at most 15 + 13)
                uint32_t d = RFL(sh.dist32[bitbuf & ((1u << DIST_BITS) - 1u)]);
                if (!(d & 15u)) { int dl; const int ds = slow_decode(sh.dist_cnt, sh.dist_sym, bitbuf, dl); d = ds < 0 ? (F_BAD | 1u) : dist_entry((uint32_t)ds, (uint32_t)dl); }
                take(d & 15u);
                dist = (d >> 16) + take((d >> 4) & 15u);
                if (d & F_BAD) ev = EV_ERROR;
                else stage = 3;
              }
              if (stage == 3) {
                if (dist > op || op + len > out_len) ev = EV_ERROR;
                else {
                  // the copy by the whole wave: lane i takes byte i of a round of 64.  A source that overlaps its destination (dist < len)
                  // repeats the dist bytes in front of op: byte k comes from op - dist + k mod dist, all of them written already
                  // A source byte still in the ring (not overwritten before this copy ends: its position + RING >= op + len) comes from LDS;
                  // an older one was flushed (every finished segment goes out before decoding continues) and is read back from the output.
                  if (len <= 64u && dist >= len && op - dist + RING >= op + len) {  // the common case in one round: no overlap, the source still in the ring
                    if ((uint32_t)lane < len) sh.out[(op + (uint32_t)lane) & RING_MASK] = sh.out[(op - dist + (uint32_t)lane) & RING_MASK];
                  } else {
                    for (uint32_t k0 = 0; k0 < len; k0 += 64) {  // (dist >= 64: a round only reads what earlier rounds or earlier symbols wrote)
                      const uint32_t k = k0 + (uint32_t)lane;
                      if (k < len) {
                        const uint32_t sp = dist >= 64u ? op + k - dist : dist == 1u ? op - 1u : op - dist + k % dist;
                        uint8_t v;
                        if (sp + RING >= op + len) v = sh.out[sp & RING_MASK];
                        else v = __hip_atomic_load(outp + sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        sh.out[(op + k) & RING_MASK] = v;
                      }
                    }
                  }
                  op += len;
                }
              }
              // (a different segment: the hand-written loop stops right behind a boundary, one more symbol of this loop may follow it)
              if (ev == EV_NONE) ev = op > out_len ? (uint32_t)EV_ERROR : ((before ^ op) >= SEG) ? (uint32_t)EV_FLUSH : in_pos > safe_end ? low_ev : (uint32_t)EV_NONE;
            }
            want = ev == EV_BUILD ? (uint32_t)EV_NONE : ev;
          }
        }
        if (want == EV_RELOAD) {
          // the window restarts at the byte that holds the next unconsumed bit; its consumed bits are dropped again behind the load
          const uint32_t whole = bitcnt >> 3, frac = bitcnt & 7u;
          in_pos = in_pos - whole - (frac ? 1u : 0u);
          pending_skip = frac ? 8u - frac : 0u;
          bitbuf = 0; bitcnt = 0;
          wb = in_pos & ~3u;
          sh.win_base = wb;
        }
        sh.ev = want; sh.op = op;
      }
      __syncthreads();
      const uint32_t ev = RFL(sh.ev);
      if (ev == EV_ERROR) { ok = false; break; }
      // ---- what the wave does together
      if (ev == EV_RELOAD) {
        const uint32_t w0 = sh.win_base;  // (a multiple of 4)
        for (uint32_t i = 4u * (uint32_t)lane; i < IN_WIN + 32; i += 256) {
          uint32_t w = 0;
          const uint32_t p = w0 + i;
          if (p + 4 <= in_len) __builtin_memcpy(&w, in + p, 4);
          else for (uint32_t q = 0; q < 4; ++q) if (p + q < in_len) w |= (uint32_t)in[p + q] << (8 * q);
          *reinterpret_cast<uint32_t*>(sh.in_win + i) = w;
        }
      } else if (ev == EV_COPY) {
        const uint32_t cs = sh.copy_src, cl = sh.copy_len, o = sh.copy_dst;
        for (uint32_t i = (uint32_t)lane; i < cl; i += 64) sh.out[(o + i) & RING_MASK] = in[cs + i];
      } else if (ev == EV_BUILD) {
        const uint32_t nlit = sh.nlit, ndist = sh.ndist;
        for (uint32_t i = (uint32_t)lane; i < (1u << LIT_BITS); i += 64) sh.lit32[i] = 0;
        for (uint32_t i = (uint32_t)lane; i < (1u << DIST_BITS); i += 64) sh.dist32[i] = 0;
        __syncthreads();
        for (uint32_t s = (uint32_t)lane; s < nlit + ndist; s += 64) {
          const uint32_t l = sh.lens[s];
          if (!l) continue;
          const bool is_dist = s >= nlit;
          const uint32_t bits = is_dist ? (uint32_t)DIST_BITS : (uint32_t)LIT_BITS;
          if (l > bits) continue;
          uint32_t* tab = is_dist ? sh.dist32 : sh.lit32;
          const uint32_t entry = (is_dist ? dist_entry(s - nlit, l) : lit_entry(s, l)) | (l << 12);  // (bits 12-15: the code's own length, which pairing leaves alone)
          // end of block, undefined symbols and lengths that can pass 64 bytes (symbols 275 ..: 51 + 15) keep EMPTY entries: both loops then
          // decode them canonically (slow_decode), and the hand-written loop needs no test for them
          if ((entry & (F_EOB | F_BAD)) || (!is_dist && s >= 275u)) continue;
          for (uint32_t i = rev_bits(sh.code[s], (int)l); i < (1u << bits); i += 1u << l) tab[i] = entry;
        }
        __syncthreads();
        // two literals in one entry where the second one's code lies inside the index as well.  In place: a lane reads the byte, the
        // literal flag and bits 12-15 of another entry, which no lane changes
        for (uint32_t i = (uint32_t)lane; i < (1u << LIT_BITS); i += 64) {
          const uint32_t e = sh.lit32[i], l = (e >> 12) & 15u;
          if ((e & F_LIT) && l < (uint32_t)LIT_BITS) {
            const uint32_t e2 = sh.lit32[i >> l], l2 = (e2 >> 12) & 15u;
            if ((e2 & F_LIT) && l + l2 <= (uint32_t)LIT_BITS) sh.lit32[i] = (e & 0x00FFF000u) | (l + l2) | F_LIT | F_PAIR | ((e2 & 0x00FF0000u) << 8);
          }
        }
      }
      __syncthreads();
      {  // finished segments of the ring go out; at the end, the tail
        const uint32_t opn = sh.op;
        const uint32_t upto = ev == EV_DONE ? opn : (opn & ~(SEG - 1));
        if (upto > flushed) {
          for (uint32_t i = flushed + 16u * (uint32_t)lane; i < upto; i += 1024) {
            if (i + 16 <= upto) { const uint4 v = *reinterpret_cast<const uint4*>(sh.out + (i & RING_MASK)); __builtin_memcpy(outp + i, &v, 16); }
            else for (uint32_t q = i; q < upto; ++q) outp[q] = sh.out[q & RING_MASK];
          }
          flushed = upto;
          // Far matches read these bytes back -- THIS wave does, through the L2 its own stores went to (loads marked sc1: past the CU's
          // L1), so all it takes is that the stores have completed.  __threadfence() here was an agent-scope release: buffer_wbl2 +
          // buffer_inv, a write-back and an invalidation of the XCD's whole L2 by every wave at every segment (~ 200 000 per launch) --
          // every copy from the flushed output then missed the L2 it had just been written to
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
      if (ev == EV_DONE) break;
      __syncthreads();
    }
    if (lane == 0) status[bi] = (uint8_t)((ok && flushed == out_len) ? 1 : 0);
    // (lane 0's store: every lane holds the same verdict)
  }
}

}  // namespace infl

// n raw DEFLATE streams src[src_off .. + src_len) -> dst[dst_off .. + dst_len) (host or device memory), status[b] = 1 inflated, 0 declined.
// Synchronous: returns when dst and status are complete.
// TRGT_INFLATE_COMPILER_LOOP=1: every symbol through the compiler's loop (the A/B of the hand-written one, and a mode of the tests)
// (read per launch: a test runs both loops in one process)
static uint32_t inflate_fast_flag() { return std::getenv("TRGT_INFLATE_COMPILER_LOOP") ? 0u : 1u; }

int inflate_blocks_device(trgt_hip_ctx* c, int64_t n, const uint8_t* src, uint64_t src_bytes, const infl::BlockDesc* descs, uint8_t* dst, uint64_t dst_bytes,
                          uint8_t* status, bool preserve_dst) {
  if (n <= 0) return TRGT_OK;
  TRGT_HIP_TRY(c, hipSetDevice(c->device));
  void *d_src = nullptr, *d_desc = nullptr, *d_dst = nullptr, *d_status = nullptr, *d_counter = nullptr;
  int rc;
  const bool src_dev = is_device_ptr(src), dst_dev = is_device_ptr(dst);
  if ((!src_dev && (rc = dev_get(c, S_INF_SRC, (size_t)src_bytes + 64, &d_src))) || (rc = dev_get(c, S_INF_DESC, (size_t)n * sizeof(infl::BlockDesc), &d_desc)) ||
      (!dst_dev && (rc = dev_get(c, S_INF_DST, (size_t)dst_bytes + 64, &d_dst))) || (rc = dev_get(c, S_INF_STATUS, (size_t)n + 16, &d_status)) ||
      (rc = dev_get(c, S_INF_COUNTER, 16, &d_counter)))
    return rc;
  if (src_dev) d_src = const_cast<uint8_t*>(src);
  else TRGT_HIP_TRY(c, hipMemcpyAsync(d_src, src, (size_t)src_bytes, hipMemcpyHostToDevice, c->stream));
  if (dst_dev) d_dst = dst;
  else if (preserve_dst) TRGT_HIP_TRY(c, hipMemcpyAsync(d_dst, dst, (size_t)dst_bytes, hipMemcpyHostToDevice, c->stream));  // (the bytes between the blocks come back as they were)
  TRGT_HIP_TRY(c, hipMemcpyAsync(d_desc, descs, (size_t)n * sizeof(infl::BlockDesc), hipMemcpyHostToDevice, c->stream));
  TRGT_HIP_TRY(c, hipMemsetAsync(d_counter, 0, 16, c->stream));
  const unsigned grid = (unsigned)std::min<int64_t>(n, (int64_t)c->num_cus * 16);
  hipLaunchKernelGGL(infl::inflate_blocks_kernel, dim3(grid), dim3(64), 0, c->stream, (const uint8_t*)d_src, (const infl::BlockDesc*)d_desc, (uint32_t)n, (uint8_t*)d_dst,
                     (uint8_t*)d_status, (unsigned int*)d_counter, inflate_fast_flag());
  TRGT_HIP_TRY(c, hipGetLastError());
  if (!dst_dev) TRGT_HIP_TRY(c, hipMemcpyAsync(dst, d_dst, (size_t)dst_bytes, hipMemcpyDeviceToHost, c->stream));
  TRGT_HIP_TRY(c, hipMemcpyAsync(status, d_status, (size_t)n, hipMemcpyDeviceToHost, c->stream));
  TRGT_HIP_TRY(c, trgt::stream_wait(c, c->stream));
  return TRGT_OK;
}

void inflate_launch(void* hip_stream, const uint8_t* d_src, const infl::BlockDesc* d_blocks, uint32_t n, uint8_t* d_dst, uint8_t* d_status, unsigned* d_counter,
                    unsigned waves) {
  if (!n) return;
  hipLaunchKernelGGL(infl::inflate_blocks_kernel, dim3(std::min<unsigned>(n, std::max(1u, waves))), dim3(64), 0, (hipStream_t)hip_stream, d_src, d_blocks, n, d_dst, d_status, d_counter, inflate_fast_flag());
}

}  // namespace trgt

// include/trgt_hip.h: "device-side BGZF inflate"
extern "C" int trgt_inflate_blocks(trgt_hip_ctx* c, int64_t n_blocks, const uint8_t* src, const uint64_t* src_off, const uint32_t* src_len, uint8_t* dst,
                                   const uint64_t* dst_off, const uint32_t* dst_len, uint8_t* status) {
  if (!c) return TRGT_ERR_INVALID;
  if (n_blocks < 0 || (n_blocks > 0 && (!src || !src_off || !src_len || !dst || !dst_off || !dst_len || !status))) return trgt::fail(c, TRGT_ERR_INVALID, "trgt_inflate_blocks: null argument");
  try {
    std::vector<trgt::infl::BlockDesc> d((size_t)n_blocks);
    uint64_t sb = 0, db = 0;
    for (int64_t b = 0; b < n_blocks; ++b) {
      if (dst_len[b] > 65536u) return trgt::fail(c, TRGT_ERR_INVALID, "trgt_inflate_blocks: block %lld inflates to %u bytes (a BGZF block holds at most 65536)", (long long)b, dst_len[b]);
      d[(size_t)b] = trgt::infl::BlockDesc{src_off[b], dst_off[b], src_len[b], dst_len[b]};
      sb = std::max<uint64_t>(sb, src_off[b] + src_len[b]); db = std::max<uint64_t>(db, dst_off[b] + dst_len[b]);
    }
    return trgt::inflate_blocks_device(c, n_blocks, src, sb, d.data(), dst, db, status, true);
  } catch (const std::bad_alloc&) { return trgt::fail(c, TRGT_ERR_NOMEM, "out of host memory"); }
}
