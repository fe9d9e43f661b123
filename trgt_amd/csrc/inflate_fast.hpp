// trgt_amd/csrc/inflate_fast.hpp -- raw DEFLATE (RFC 1951) decoder for whole BGZF blocks, host code.
//
// Read ingestion (ingest.hip) spends five sixths of its time inflating BGZF blocks; zlib 1.2.11 -- what the image has -- does
// about 230-360 MB/s per core there.  This decoder takes what is particular to the case -- the whole compressed block and the whole
// output buffer are in memory, the output size is known -- and does the usual things with it: a 64-bit bit buffer refilled with one
// unaligned load, one table look-up per symbol (11-bit root table for literals / lengths, 8-bit for distances, sub-tables behind
// them), matches copied eight bytes at a time.  It decodes or it fails: on ANY irregularity (invalid or incomplete code, distance too
// far back, input or output exhausted, output size not met) it returns false and the caller runs zlib on the same block, so that
// what is accepted and what is reported for a damaged file stay zlib's.  tests/test_inflate.py compares it with zlib on the blocks of
// BAM files and on streams of every block type and compression level.
#pragma once
#include <cstdint>
#include <cstring>

namespace trgt {
namespace inflate_fast {

constexpr int LIT_ROOT = 11, DIST_ROOT = 8, PRE_ROOT = 7;
constexpr int LIT_TABLE = (1 << LIT_ROOT) + 1024, DIST_TABLE = (1 << DIST_ROOT) + 512, PRE_TABLE = 1 << PRE_ROOT;  // (sub-tables: bounded below, checked while building)
// table entry: bits 0-3 code bits to consume | 4-6 kind | 8-12 extra bits (or sub-table index bits) | 16-31 value (literal, base, sub-table offset)
enum : uint32_t { K_LIT = 0, K_EOB = 1, K_BASE = 2, K_SUB = 3, K_BAD = 7 };
constexpr uint32_t BAD_ENTRY = K_BAD << 4;
inline uint32_t entry(uint32_t bits, uint32_t kind, uint32_t extra, uint32_t value) { return bits | (kind << 4) | (extra << 8) | (value << 16); }

struct Tables {
  uint32_t lit[LIT_TABLE];
  uint32_t dist[DIST_TABLE];
  uint32_t pre[PRE_TABLE];
};

inline uint32_t reverse_bits(uint32_t code, int len) {
  uint32_t r = 0;
  for (int i = 0; i < len; ++i) { r = (r << 1) | (code & 1u); code >>= 1; }
  return r;
}

// Canonical Huffman decoding table of `n` symbols with code lengths lens[] (0 = unused, <= 15).  sym_entry(s, len): the entry of symbol
// s without its code length.  Returns false for an over-subscribed or incomplete set, or one that does not fit the table: the caller
// falls back to zlib, which knows the exceptions RFC 1951 allows.
template <typename F>
inline bool build_table(const uint8_t* lens, int n, int root, uint32_t* table, int table_cap, F sym_entry) {
  int count[16] = {0};
  for (int s = 0; s < n; ++s) ++count[lens[s]];
  if (count[0] == n) return false;
  int left = 1;
  for (int l = 1; l <= 15; ++l) { left = (left << 1) - count[l]; if (left < 0) return false; }
  if (left != 0) return false;  // incomplete
  uint32_t next_code[16]; uint32_t code = 0;
  for (int l = 1; l <= 15; ++l) { code = (code + (uint32_t)count[l - 1]) << 1; next_code[l] = code; }
  next_code[0] = 0;
  const int root_size = 1 << root;
  for (int i = 0; i < root_size; ++i) table[i] = BAD_ENTRY;
  // first pass: the longest code behind every root prefix (-> size of its sub-table)
  uint8_t sub_len[1 << LIT_ROOT];
  bool any_long = false;
  uint32_t codes[320];
  if (n > 320) return false;
  std::memset(sub_len, 0, (size_t)root_size);
  for (int s = 0; s < n; ++s) {
    const int l = lens[s];
    if (!l) continue;
    const uint32_t r = reverse_bits(next_code[l]++, l);
    codes[s] = r;
    if (l > root) { any_long = true; uint8_t& m = sub_len[r & (uint32_t)(root_size - 1)]; if (l - root > m) m = (uint8_t)(l - root); }
  }
  int used = root_size;
  if (any_long) {
    for (int p = 0; p < root_size; ++p) {
      if (!sub_len[p]) continue;
      const int sz = 1 << sub_len[p];
      if (used + sz > table_cap) return false;
      table[p] = entry((uint32_t)root, K_SUB, sub_len[p], (uint32_t)used);
      for (int i = 0; i < sz; ++i) table[used + i] = BAD_ENTRY;
      used += sz;
    }
  }
  for (int s = 0; s < n; ++s) {
    const int l = lens[s];
    if (!l) continue;
    const uint32_t r = codes[s];
    if (l <= root) {
      const uint32_t e = sym_entry(s) | (uint32_t)l;
      for (uint32_t i = r; i < (uint32_t)root_size; i += 1u << l) table[i] = e;
    } else {
      const uint32_t head = table[r & (uint32_t)(root_size - 1)];
      const int sb = (int)((head >> 8) & 31u);
      const uint32_t off = head >> 16;
      const uint32_t e = sym_entry(s) | (uint32_t)(l - root);
      for (uint32_t i = r >> root; i < (1u << sb); i += 1u << (l - root)) table[off + i] = e;
    }
  }
  return true;
}

inline uint64_t load_u64(const uint8_t* p) { uint64_t v; std::memcpy(&v, p, 8); return v; }

// in[0, n_in): a complete raw deflate stream; out[0, n_out): exactly what it must inflate to.
inline bool inflate_block(const uint8_t* in, size_t n_in, uint8_t* out, size_t n_out, Tables& T) {
  static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
  static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
  static const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
  static const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
  static const uint8_t pre_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  const uint8_t* ip = in; const uint8_t* const in_end = in + n_in;
  uint8_t* op = out; uint8_t* const out_end = out + n_out;
  uint64_t buf = 0; unsigned cnt = 0;  // cnt valid bits in buf
  size_t fake = 0;                     // zero bytes shifted in beyond the input (more than the buffer can hold means the stream was cut short)
  auto refill = [&]() {
    if (in_end - ip >= 8) { buf |= load_u64(ip) << cnt; ip += (63 - cnt) >> 3; cnt |= 56; }
    else { while (cnt < 56) { if (ip < in_end) buf |= (uint64_t)*ip++ << cnt; else ++fake; cnt += 8; } }  // (56 .. 63 valid bits either way)
  };
  auto lit_sym = [&](int s) -> uint32_t {
    if (s < 256) return entry(0, K_LIT, 0, (uint32_t)s);
    if (s == 256) return entry(0, K_EOB, 0, 0);
    if (s <= 285) return entry(0, K_BASE, len_extra[s - 257], len_base[s - 257]);
    return BAD_ENTRY;  // 286, 287: never valid in the data
  };
  auto dist_sym = [&](int s) -> uint32_t { return s < 30 ? entry(0, K_BASE, dist_extra[s], dist_base[s]) : BAD_ENTRY; };
  auto pre_sym = [&](int s) -> uint32_t { return entry(0, K_LIT, 0, (uint32_t)s); };
  bool last = false;
  while (!last) {
    refill();
    last = buf & 1u;
    const unsigned type = (unsigned)(buf >> 1) & 3u;
    buf >>= 3; cnt -= 3;
    if (type == 0) {  // stored
      const unsigned drop = cnt & 7u;
      buf >>= drop; cnt -= drop;
      // the bytes still in the bit buffer go back to the input (the fake ones, shifted in last, are simply dropped)
      if (fake > (cnt >> 3)) return false;
      ip -= (cnt >> 3) - fake; fake = 0; buf = 0; cnt = 0;
      if (in_end - ip < 4) return false;
      const unsigned len = ip[0] | (ip[1] << 8), nlen = ip[2] | (ip[3] << 8);
      ip += 4;
      if ((len ^ 0xFFFFu) != nlen || (size_t)(in_end - ip) < len || (size_t)(out_end - op) < len) return false;
      std::memcpy(op, ip, len); ip += len; op += len;
      continue;
    }
    if (type == 3) return false;
    if (type == 1) {  // fixed codes
      uint8_t lens[288 + 32];
      for (int i = 0; i < 144; ++i) lens[i] = 8;
      for (int i = 144; i < 256; ++i) lens[i] = 9;
      for (int i = 256; i < 280; ++i) lens[i] = 7;
      for (int i = 280; i < 288; ++i) lens[i] = 8;
      for (int i = 0; i < 32; ++i) lens[288 + i] = 5;
      if (!build_table(lens, 288, LIT_ROOT, T.lit, LIT_TABLE, lit_sym) || !build_table(lens + 288, 32, DIST_ROOT, T.dist, DIST_TABLE, dist_sym)) return false;
    } else {  // dynamic codes
      const int hlit = (int)(buf & 31u) + 257, hdist = (int)((buf >> 5) & 31u) + 1, hclen = (int)((buf >> 10) & 15u) + 4;
      buf >>= 14; cnt -= 14;
      if (hlit > 286 || hdist > 30) return false;
      uint8_t plens[19] = {0};
      refill();
      for (int i = 0; i < hclen; ++i) {
        if (i == 12) refill();  // (19 x 3 = 57 bits: one more than a refill guarantees)
        plens[pre_order[i]] = (uint8_t)(buf & 7u); buf >>= 3; cnt -= 3;
      }
      if (!build_table(plens, 19, PRE_ROOT, T.pre, PRE_TABLE, pre_sym)) return false;
      uint8_t lens[286 + 30 + 140];
      int i = 0;
      const int total = hlit + hdist;
      while (i < total) {
        refill();
        const uint32_t e = T.pre[buf & (PRE_TABLE - 1)];
        if (((e >> 4) & 7u) != K_LIT) return false;
        buf >>= (e & 15u); cnt -= (e & 15u);
        const unsigned s = e >> 16;
        if (s < 16) { lens[i++] = (uint8_t)s; continue; }
        unsigned rep, val = 0;
        if (s == 16) { if (i == 0) return false; val = lens[i - 1]; rep = 3 + (unsigned)(buf & 3u); buf >>= 2; cnt -= 2; }
        else if (s == 17) { rep = 3 + (unsigned)(buf & 7u); buf >>= 3; cnt -= 3; }
        else { rep = 11 + (unsigned)(buf & 127u); buf >>= 7; cnt -= 7; }
        if (i + (int)rep > total) return false;
        std::memset(lens + i, (int)val, rep);
        i += (int)rep;
      }
      if (lens[256] == 0) return false;  // no end-of-block code
      if (!build_table(lens, hlit, LIT_ROOT, T.lit, LIT_TABLE, lit_sym)) return false;
      if (!build_table(lens + hlit, hdist, DIST_ROOT, T.dist, DIST_TABLE, dist_sym)) {
        // a block of literals only may come with no usable distance code: then any distance symbol is an error
        for (int k = 0; k < (1 << DIST_ROOT); ++k) T.dist[k] = BAD_ENTRY;
        int nz = 0; for (int k = 0; k < hdist; ++k) nz += lens[hlit + k] != 0;
        if (nz > 1) return false;  // (incomplete sets with several codes: zlib decides)
        if (nz == 1) return false; // (the one-code set RFC 1951 allows: rare, left to zlib as well)
      }
    }
    // ---- the symbols of the block
    const uint32_t lit_mask = (1u << LIT_ROOT) - 1;
    uint8_t* const out_fast_end = n_out > 320 ? out_end - 320 : out;         // a match writes at most 258 + 7 bytes
    const uint8_t* const in_fast_end = n_in > 32 ? in_end - 32 : in;          // two refills read at most 16 bytes
    for (;;) {
      if (op < out_fast_end && ip < in_fast_end) {
        // ---- away from both ends: no bounds tests, up to three literals per refill
        buf |= load_u64(ip) << cnt; ip += (63 - cnt) >> 3; cnt |= 56;
        uint32_t e = T.lit[buf & lit_mask];
        if ((e & 0x70u) == (K_LIT << 4)) {
          buf >>= (e & 15u); cnt -= (e & 15u); *op++ = (uint8_t)(e >> 16);
          e = T.lit[buf & lit_mask];
          if ((e & 0x70u) == (K_LIT << 4)) {
            buf >>= (e & 15u); cnt -= (e & 15u); *op++ = (uint8_t)(e >> 16);
            e = T.lit[buf & lit_mask];
            if ((e & 0x70u) == (K_LIT << 4)) { buf >>= (e & 15u); cnt -= (e & 15u); *op++ = (uint8_t)(e >> 16); continue; }
          }
          buf |= load_u64(ip) << cnt; ip += (63 - cnt) >> 3; cnt |= 56;  // (the entry at hand was read from bits that stay where they are)
        }
        if ((e & 0x70u) == (K_SUB << 4)) {
          buf >>= LIT_ROOT; cnt -= LIT_ROOT;
          e = T.lit[(e >> 16) + (uint32_t)(buf & ((1u << ((e >> 8) & 31u)) - 1))];
        }
        const uint32_t kind = (e >> 4) & 7u;
        buf >>= (e & 15u); cnt -= (e & 15u);
        if (kind == K_LIT) { *op++ = (uint8_t)(e >> 16); continue; }
        if (kind == K_EOB) break;
        if (kind != K_BASE) return false;
        const unsigned xb = (e >> 8) & 31u;
        const size_t len = (e >> 16) + (size_t)(buf & ((1u << xb) - 1));
        buf >>= xb; cnt -= xb;
        uint32_t d = T.dist[buf & ((1u << DIST_ROOT) - 1)];
        if ((d & 0x70u) == (K_SUB << 4)) {
          buf >>= DIST_ROOT; cnt -= DIST_ROOT;
          d = T.dist[(d >> 16) + (uint32_t)(buf & ((1u << ((d >> 8) & 31u)) - 1))];
        }
        if ((d & 0x70u) != (K_BASE << 4)) return false;
        buf >>= (d & 15u); cnt -= (d & 15u);
        const unsigned db = (d >> 8) & 31u;
        const size_t dist = (d >> 16) + (size_t)(buf & ((1u << db) - 1));
        buf >>= db; cnt -= db;
        if (dist > (size_t)(op - out)) return false;
        const uint8_t* src = op - dist;
        uint8_t* dst = op;
        op += len;
        if (dist >= 16) { do { std::memcpy(dst, src, 16); dst += 16; src += 16; } while (dst < op); }  // (up to 15 bytes beyond the match: inside the 320-byte margin)
        else if (dist >= 8) { do { std::memcpy(dst, src, 8); dst += 8; src += 8; } while (dst < op); }
        else if (dist == 1) std::memset(dst, *src, len);
        else { do { *dst++ = *src++; } while (dst < op); }
        continue;
      }
      refill();
      if (fake > 8) return false;
      uint32_t e = T.lit[buf & ((1u << LIT_ROOT) - 1)];
      if (((e >> 4) & 7u) == K_SUB) {
        buf >>= LIT_ROOT; cnt -= LIT_ROOT;
        e = T.lit[(e >> 16) + (uint32_t)(buf & ((1u << ((e >> 8) & 31u)) - 1))];
      }
      const uint32_t kind = (e >> 4) & 7u;
      buf >>= (e & 15u); cnt -= (e & 15u);
      if (kind == K_LIT) {
        if (op >= out_end) return false;
        *op++ = (uint8_t)(e >> 16);
        // a second literal from the same refill (at most 2 x 15 bits used so far)
        uint32_t e2 = T.lit[buf & ((1u << LIT_ROOT) - 1)];
        if (((e2 >> 4) & 7u) == K_LIT && op < out_end) { buf >>= (e2 & 15u); cnt -= (e2 & 15u); *op++ = (uint8_t)(e2 >> 16); }
        continue;
      }
      if (kind == K_EOB) break;
      if (kind != K_BASE) return false;
      const unsigned xb = (e >> 8) & 31u;
      size_t len = (e >> 16) + (size_t)(buf & ((1u << xb) - 1));
      buf >>= xb; cnt -= xb;
      // (bits used since the refill: <= 15 + 5 = 20; the distance needs <= 15 + 13 = 28 more; >= 56 were there)
      uint32_t d = T.dist[buf & ((1u << DIST_ROOT) - 1)];
      if (((d >> 4) & 7u) == K_SUB) {
        buf >>= DIST_ROOT; cnt -= DIST_ROOT;
        d = T.dist[(d >> 16) + (uint32_t)(buf & ((1u << ((d >> 8) & 31u)) - 1))];
      }
      if (((d >> 4) & 7u) != K_BASE) return false;
      buf >>= (d & 15u); cnt -= (d & 15u);
      const unsigned db = (d >> 8) & 31u;
      const size_t dist = (d >> 16) + (size_t)(buf & ((1u << db) - 1));
      buf >>= db; cnt -= db;
      if (dist > (size_t)(op - out) || len > (size_t)(out_end - op)) return false;
      const uint8_t* src = op - dist;
      if (dist >= 8 && (size_t)(out_end - op) >= len + 8) {  // eight bytes at a time (may write up to 7 bytes beyond the match: inside the buffer)
        uint8_t* dst = op;
        const uint8_t* const stop = op + len;
        do { std::memcpy(dst, src, 8); dst += 8; src += 8; } while (dst < stop);
      } else if (dist == 1) {
        std::memset(op, *src, len);
      } else {
        for (size_t k = 0; k < len; ++k) op[k] = src[k];
      }
      op += len;
    }
  }
  // everything consumed must have been real input, and the output must be exactly as long as announced
  if (op != out_end) return false;
  const size_t unread_bytes = cnt >> 3;  // whole bytes still in the bit buffer
  if (fake > unread_bytes) return false;
  return true;
}

}  // namespace inflate_fast
}  // namespace trgt
