// trgt_amd/csrc/ingest.hip -- read ingestion in front of the GPU path, host C++ (no device code):
//   repeat catalog + indexed FASTA -> Locus                     src/trgt/locus.rs:13-23, 168-215
//   indexed BAM -> the reads of a locus                          src/trgt/workflows/tr.rs:268-361 (extract_reads)
//   HiFiRead::from_hts_rec                                       src/trgt/reads/read.rs:55-141 (bases, quals, rq / HP tags, MM / ML)
//   extract_snps_offset                                          src/trgt/reads/snp.rs:51-79
//   HiFiRead::clip_to_region, clip_reads                         src/trgt/reads/clip_region.rs:19-184, tr.rs:186-196
// and the assembly of the result as the arrays of trgt_locus_batch_in, so that a BAM goes straight into trgt_locus_batch.
// The reference does the file work through htslib; here BGZF blocks are inflated with zlib, regions are looked up in the .bai
// (bins + linear index) and in the .fai, and the loci of a batch are read by a pool of threads, each with its own file handle.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <sstream>
#include <string>
#include <chrono>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/trgt_hip.h"
#include "inflate_dev.hpp"
#include "inflate_fast.hpp"
#include "crc32_fast.hpp"
#include "ingest_dev.hpp"
#include <condition_variable>

#ifndef TRGT_DEV_ENV   // (common.hpp: developer switches are read in `make DEV=1` builds only)
#ifdef TRGT_DEV_BUILD
#define TRGT_DEV_ENV(name) getenv(name)
#else
#define TRGT_DEV_ENV(name) ((const char*)nullptr)
#endif
#endif

namespace {

// ---------------------------------------------------------------------------------------------- BGZF
struct Bgzf {
  // A few inflated blocks are kept (least recently used one replaced): the .bai sends the query of a locus back to the first record of
  // the 16 kb window its region starts in, so neighbouring loci of a dense catalog walk over the same blocks, and a worker that takes
  // a run of consecutive loci inflates each of them once.
  struct Block { uint64_t coff = ~0ull; uint32_t csize = 0; uint64_t stamp = 0; std::vector<uint8_t> data; };
  int fd = -1;
  std::vector<uint8_t> raw;
  std::vector<Block> cache;
  size_t cur = 0;               // the block being read
  uint64_t clock = 0, n_inflated = 0, n_hits = 0, n_declined = 0;
  std::unique_ptr<trgt::inflate_fast::Tables> tables;
  uint64_t block_coff = ~0ull;  // compressed offset of the current block
  uint32_t block_csize = 0;     // its size in the file (0: end of file)
  size_t pos = 0;               // read position inside it
  z_stream zs; bool zs_ready = false;
  std::string err;
  explicit Bgzf(size_t n_cached = 1) : cache(std::max<size_t>(1, n_cached)) { std::memset(&zs, 0, sizeof zs); }
  Bgzf(const Bgzf&) = delete;
  Bgzf& operator=(const Bgzf&) = delete;
  ~Bgzf() { if (fd >= 0) ::close(fd); if (zs_ready) inflateEnd(&zs); }
  const uint8_t* cur_data = nullptr; size_t cur_size = 0;  // the current block's bytes: an entry of `cache`
  bool open(const char* path) { fd = ::open(path, O_RDONLY); if (fd < 0) { err = std::string("cannot open ") + path; return false; } return true; }
  bool load(uint64_t coff) {
    size_t lru = 0;
    for (size_t i = 0; i < cache.size(); ++i) {
      if (cache[i].coff == coff) { cur = i; cache[i].stamp = ++clock; block_coff = coff; block_csize = cache[i].csize; pos = 0; ++n_hits; cur_data = cache[i].data.data(); cur_size = cache[i].data.size(); return true; }
      if (cache[i].stamp < cache[lru].stamp) lru = i;
    }
    uint8_t h[18];
    const ssize_t got = ::pread(fd, h, 18, (off_t)coff);
    Block& B = cache[lru];
    if (got == 0) { B.data.clear(); B.coff = ~0ull; cur = lru; block_coff = coff; block_csize = 0; pos = 0; cur_data = nullptr; cur_size = 0; return true; }  // end of file (not kept)
    if (got != 18 || h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) { err = "not a BGZF block"; return false; }
    const uint32_t xlen = h[10] | (h[11] << 8);
    // the BC subfield is the first one in every file bgzip / htslib / pbmm2 writes; look it up properly all the same
    uint32_t bsize = 0; bool found = false;
    if (xlen == 6 && h[12] == 'B' && h[13] == 'C' && h[14] == 2 && h[15] == 0) { bsize = h[16] | (h[17] << 8); found = true; }
    else {
      std::vector<uint8_t> extra(xlen);
      if (::pread(fd, extra.data(), xlen, (off_t)coff + 12) != (ssize_t)xlen) { err = "truncated BGZF header"; return false; }
      for (uint32_t i = 0; i + 4 <= xlen;) {
        const uint32_t slen = extra[i + 2] | (extra[i + 3] << 8);
        if (extra[i] == 'B' && extra[i + 1] == 'C' && slen == 2 && i + 6 <= xlen) { bsize = extra[i + 4] | (extra[i + 5] << 8); found = true; break; }
        i += 4 + slen;
      }
    }
    if (!found) { err = "BGZF block without BC field"; return false; }
    const uint32_t total = bsize + 1, hdr = 12 + xlen;
    if (total < hdr + 8) { err = "bad BGZF block size"; return false; }
    raw.resize(total);
    if (::pread(fd, raw.data(), total, (off_t)coff) != (ssize_t)total) { err = "truncated BGZF block"; return false; }
    const uint32_t isize = raw[total - 4] | (raw[total - 3] << 8) | (raw[total - 2] << 16) | ((uint32_t)raw[total - 1] << 24);
    B.coff = ~0ull;  // (not a valid entry while it is being overwritten)
    B.data.resize(isize);
    bool isize_done = false;
    static const bool zlib_only = TRGT_DEV_ENV("TRGT_ZLIB_INFLATE") != nullptr;  // (the decoder of inflate_fast.hpp is tried first; zlib takes every block it declines)
    if (isize && !zlib_only) {
      if (!tables) tables.reset(new trgt::inflate_fast::Tables());
      if (!trgt::inflate_fast::inflate_block(raw.data() + hdr, total - hdr - 8, B.data.data(), isize, *tables)) ++n_declined;
      else isize_done = true;
    }
    if (isize && !isize_done) {
      if (!zs_ready) { if (inflateInit2(&zs, -15) != Z_OK) { err = "inflateInit2 failed"; return false; } zs_ready = true; }
      else if (inflateReset(&zs) != Z_OK) { err = "inflateReset failed"; return false; }
      zs.next_in = raw.data() + hdr; zs.avail_in = total - hdr - 8;
      zs.next_out = B.data.data(); zs.avail_out = isize;
      const int rc = inflate(&zs, Z_FINISH);
      if (rc != Z_STREAM_END || zs.avail_out != 0) { err = "corrupt BGZF block"; return false; }
    }
    // the footer's CRC-32 of the inflated bytes (htslib's bgzf_read_block refuses a block whose CRC does not match: a damaged block whose
    // length happens to fit must not be read as records)
    if (trgt::crc32_fast(B.data.data(), isize) != (raw[total - 8] | (raw[total - 7] << 8) | (raw[total - 6] << 16) | ((uint32_t)raw[total - 5] << 24))) { err = "corrupt BGZF block (CRC-32 mismatch)"; return false; }
    ++n_inflated;
    B.coff = coff; B.csize = total; B.stamp = ++clock; cur = lru;
    block_coff = coff; block_csize = total; pos = 0; cur_data = B.data.data(); cur_size = B.data.size();
    return true;
  }
  bool seek(uint64_t voff) {
    const uint64_t coff = voff >> 16;
    if (coff != block_coff && !load(coff)) return false;
    pos = (size_t)(voff & 0xFFFF);
    return pos <= cur_size;
  }
  uint64_t tell() const { return pos < cur_size || block_csize == 0 ? (block_coff << 16) | pos : ((block_coff + block_csize) << 16); }
  // n bytes; returns 1 ok, 0 clean end of file before the first byte, -1 error
  int read(void* dst, size_t n) {
    uint8_t* d = (uint8_t*)dst;
    size_t done = 0;
    while (done < n) {
      if (pos >= cur_size) {
        if (block_csize == 0 && block_coff != ~0ull) { if (done == 0) return 0; err = "unexpected end of BAM"; return -1; }
        if (!load(block_coff + block_csize)) return -1;
        if (cur_size == 0 && block_csize == 0) { if (done == 0) return 0; err = "unexpected end of BAM"; return -1; }
        continue;
      }
      const size_t k = std::min(n - done, cur_size - pos);
      std::memcpy(d + done, cur_data + pos, k);
      done += k; pos += k;
    }
    return 1;
  }
};

inline uint32_t le32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint64_t le64(const uint8_t* p) { return (uint64_t)le32(p) | ((uint64_t)le32(p + 4) << 32); }

// ---------------------------------------------------------------------------------------------- BAI
struct BaiRef {
  std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins;
  std::vector<uint64_t> linear;
};
struct Bai {
  std::vector<BaiRef> refs;
  bool load(const std::string& path, std::string& err) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { err = "cannot open " + path; return false; }
    std::vector<uint8_t> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (d.size() < 8 || std::memcmp(d.data(), "BAI\1", 4) != 0) { err = "not a BAI index: " + path; return false; }
    size_t p = 4;
    auto need = [&](size_t n) { return p + n <= d.size(); };
    const uint32_t n_ref = le32(&d[p]); p += 4;
    if ((uint64_t)n_ref * 8 > d.size()) { err = "corrupt BAI"; return false; }  // (every reference has at least two counters)
    refs.resize(n_ref);
    for (uint32_t r = 0; r < n_ref; ++r) {
      if (!need(4)) { err = "truncated BAI"; return false; }
      const uint32_t n_bin = le32(&d[p]); p += 4;
      for (uint32_t b = 0; b < n_bin; ++b) {
        if (!need(8)) { err = "truncated BAI"; return false; }
        const uint32_t bin = le32(&d[p]), n_chunk = le32(&d[p + 4]); p += 8;
        if (!need(16ull * n_chunk)) { err = "truncated BAI"; return false; }
        auto& v = refs[r].bins[bin];
        for (uint32_t c = 0; c < n_chunk; ++c) { v.emplace_back(le64(&d[p]), le64(&d[p + 8])); p += 16; }
      }
      if (!need(4)) { err = "truncated BAI"; return false; }
      const uint32_t n_intv = le32(&d[p]); p += 4;
      if (!need(8ull * n_intv)) { err = "truncated BAI"; return false; }
      refs[r].linear.resize(n_intv);
      for (uint32_t i = 0; i < n_intv; ++i) { refs[r].linear[i] = le64(&d[p]); p += 8; }
    }
    return true;
  }
  // chunks that may hold records overlapping [beg, end) of reference tid, sorted and merged (SAM spec 5.3: reg2bins)
  std::vector<std::pair<uint64_t, uint64_t>> query(int tid, int64_t beg, int64_t end) const {
    std::vector<std::pair<uint64_t, uint64_t>> out;
    if (tid < 0 || (size_t)tid >= refs.size() || end <= beg) return out;
    const BaiRef& R = refs[(size_t)tid];
    if (beg < 0) beg = 0;
    const int64_t e = std::min<int64_t>(end, 1ll << 29) - 1;
    if (beg > e) return out;
    const size_t li = (size_t)(beg >> 14);
    const uint64_t min_off = R.linear.empty() ? 0 : R.linear[std::min(li, R.linear.size() - 1)];
    auto add = [&](uint32_t bin) {
      auto it = R.bins.find(bin);
      if (it == R.bins.end()) return;
      for (auto& ch : it->second) if (ch.second > min_off) out.push_back(ch);
    };
    add(0);
    for (int64_t k = 1 + (beg >> 26); k <= 1 + (e >> 26); ++k) add((uint32_t)k);
    for (int64_t k = 9 + (beg >> 23); k <= 9 + (e >> 23); ++k) add((uint32_t)k);
    for (int64_t k = 73 + (beg >> 20); k <= 73 + (e >> 20); ++k) add((uint32_t)k);
    for (int64_t k = 585 + (beg >> 17); k <= 585 + (e >> 17); ++k) add((uint32_t)k);
    for (int64_t k = 4681 + (beg >> 14); k <= 4681 + (e >> 14); ++k) add((uint32_t)k);
    std::sort(out.begin(), out.end());
    std::vector<std::pair<uint64_t, uint64_t>> merged;
    for (auto& ch : out) {
      if (!merged.empty() && ch.first <= merged.back().second) merged.back().second = std::max(merged.back().second, ch.second);
      else merged.push_back(ch);
    }
    return merged;
  }
};

// ---------------------------------------------------------------------------------------------- FASTA + .fai
struct FaiEntry { int64_t len = 0, offset = 0, line_bases = 0, line_width = 0; };
struct Fasta {
  int fd = -1;
  std::map<std::string, FaiEntry> idx;
  ~Fasta() { if (fd >= 0) ::close(fd); }
  bool open(const std::string& path, std::string& err) {
    fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) { err = "cannot open " + path; return false; }
    std::ifstream f(path + ".fai");
    if (!f) { err = "cannot open " + path + ".fai"; return false; }
    std::string line;
    while (std::getline(f, line)) {
      std::istringstream ss(line);
      std::string name; FaiEntry e;
      if (ss >> name >> e.len >> e.offset >> e.line_bases >> e.line_width) idx[name] = e;
    }
    return true;
  }
  int64_t length(const std::string& contig) const { auto it = idx.find(contig); return it == idx.end() ? -1 : it->second.len; }
  // bases [beg, end) of a contig, upper-cased (locus.rs:168-190: fetch_seq_string + to_uppercase)
  bool fetch(const std::string& contig, int64_t beg, int64_t end, std::string& out, std::string& err) const {
    auto it = idx.find(contig);
    if (it == idx.end()) { err = "contig " + contig + " is not in the genome"; return false; }
    const FaiEntry& e = it->second;
    if (beg < 0 || end > e.len || beg > end || e.line_bases <= 0) { err = "Error fetching sequence for region " + contig + ":" + std::to_string(beg) + "-" + std::to_string(end); return false; }
    out.clear();
    if (beg == end) return true;
    const int64_t o0 = e.offset + beg / e.line_bases * e.line_width + beg % e.line_bases;
    const int64_t o1 = e.offset + (end - 1) / e.line_bases * e.line_width + (end - 1) % e.line_bases + 1;
    std::string rawb((size_t)(o1 - o0), '\0');
    if (::pread(fd, &rawb[0], rawb.size(), (off_t)o0) != (ssize_t)rawb.size()) { err = "short read from the genome"; return false; }
    out.reserve((size_t)(end - beg));
    for (char ch : rawb) if (ch != '\n' && ch != '\r') out.push_back((char)std::toupper((unsigned char)ch));
    if ((int64_t)out.size() != end - beg) { err = "genome index does not match the FASTA"; return false; }
    return true;
  }
};

// ---------------------------------------------------------------------------------------------- records
constexpr int OP_M = 0, OP_I = 1, OP_D = 2, OP_N = 3, OP_S = 4, OP_EQ = 7, OP_X = 8;
inline int64_t ref_len(uint32_t op) { const int c = (int)(op & 0xF); return (c == OP_M || c == OP_D || c == OP_N || c == OP_EQ || c == OP_X) ? (int64_t)(op >> 4) : 0; }
inline int64_t qry_len(uint32_t op) { const int c = (int)(op & 0xF); return (c == OP_M || c == OP_I || c == OP_S || c == OP_EQ || c == OP_X) ? (int64_t)(op >> 4) : 0; }

struct Read {  // HiFiRead (read.rs:10-36)
  std::string id;
  bool is_reverse = false, has_meth = false, has_cigar = false;
  std::string bases; std::vector<uint8_t> quals, meth;
  double rq = std::numeric_limits<double>::quiet_NaN();
  std::vector<int32_t> mismatch_offsets;
  int32_t start_offset = 0, end_offset = 0;
  int64_t ref_pos = 0; std::vector<uint32_t> cigar;  // BAM encoding: len << 4 | op
  int hp = -1; uint8_t mapq = 0;
};

struct RawRec { std::vector<uint8_t> d; int32_t ref_id, pos, l_seq; uint32_t l_rn, n_cig, flag; uint8_t mapq; size_t o_cig, o_seq, o_qual, o_aux; };

bool parse_rec(RawRec& r) {
  const uint8_t* d = r.d.data();
  if (r.d.size() < 32) return false;
  r.ref_id = (int32_t)le32(d); r.pos = (int32_t)le32(d + 4); r.l_rn = d[8]; r.mapq = d[9];
  r.n_cig = d[12] | (d[13] << 8); r.flag = d[14] | (d[15] << 8); r.l_seq = (int32_t)le32(d + 16);
  r.o_cig = 32 + r.l_rn; r.o_seq = r.o_cig + 4ull * r.n_cig; r.o_qual = r.o_seq + ((size_t)r.l_seq + 1) / 2; r.o_aux = r.o_qual + (size_t)r.l_seq;
  return r.o_aux <= r.d.size();
}
int64_t rec_ref_end(const RawRec& r) {
  int64_t e = r.pos;
  for (uint32_t i = 0; i < r.n_cig; ++i) e += ref_len(le32(r.d.data() + r.o_cig + 4 * i));
  return e;
}
// aux field of a record: pointer to its type byte, or null
const uint8_t* find_aux(const RawRec& r, const char* tag) {
  const uint8_t* p = r.d.data() + r.o_aux; const uint8_t* e = r.d.data() + r.d.size();
  auto size_of = [](uint8_t t) -> int { switch (t) { case 'A': case 'c': case 'C': return 1; case 's': case 'S': return 2; case 'i': case 'I': case 'f': return 4; default: return 0; } };
  while (p + 3 <= e) {
    const uint8_t ty = p[2];
    if (p[0] == (uint8_t)tag[0] && p[1] == (uint8_t)tag[1]) return p + 2;
    p += 3;
    if (const int s = size_of(ty)) p += s;
    else if (ty == 'Z' || ty == 'H') { while (p < e && *p) ++p; ++p; }
    else if (ty == 'B') { if (p + 5 > e) return nullptr; const int s2 = size_of(p[0]); const uint32_t n = le32(p + 1); p += 5 + (size_t)s2 * n; }
    else return nullptr;
  }
  return nullptr;
}
bool aux_int(const uint8_t* a, int64_t& v) {
  if (!a) return false;
  switch (a[0]) {
    case 'c': v = (int8_t)a[1]; return true;
    case 'C': v = a[1]; return true;
    case 's': v = (int16_t)(a[1] | (a[2] << 8)); return true;
    case 'S': v = a[1] | (a[2] << 8); return true;
    case 'i': v = (int32_t)le32(a + 1); return true;
    case 'I': v = le32(a + 1); return true;
    default: return false;
  }
}

// 5mC calls of a record as (position in the stored sequence, probability), in stored order: the "C+m" entries of the MM tag with the
// ML values that belong to them (SAM tags spec 1.7: skip counts over the canonical base of the ORIGINAL strand; a reverse-strand
// record stores the reverse complement, so its C's are G's counted from the end).  rust-htslib's basemods_iter reports them in the
// same order; get_meth (read.rs:55-96) keeps those that sit on a CpG.
bool basemods_5mc_tags(const char* mm_text, const uint8_t* mlv, uint32_t n_ml, const std::string& bases, bool reverse, std::vector<std::pair<uint32_t, uint8_t>>& out);
bool basemods_5mc(const RawRec& r, const std::string& bases, bool reverse, std::vector<std::pair<uint32_t, uint8_t>>& out) {
  out.clear();
  const uint8_t* mm = find_aux(r, "MM"); if (!mm) mm = find_aux(r, "Mm");
  const uint8_t* ml = find_aux(r, "ML"); if (!ml) ml = find_aux(r, "Ml");
  if (!mm || mm[0] != 'Z' || !ml || ml[0] != 'B' || (ml[1] != 'C' && ml[1] != 'c')) return false;
  return basemods_5mc_tags((const char*)mm + 1, ml + 6, le32(ml + 2), bases, reverse, out);
}
// (false: a tag htslib's bam_parse_basemod refuses -- get_meth then returns None, read.rs:189-195 test_basemods_error)
bool basemods_5mc_tags(const char* mm_text, const uint8_t* mlv, uint32_t n_ml, const std::string& bases, bool reverse, std::vector<std::pair<uint32_t, uint8_t>>& out) {
  out.clear();
  const char* s = mm_text;
  uint32_t ml_at = 0;
  const size_t n = bases.size();
  while (*s) {
    // one entry: base strand codes [.?] {,delta} ;
    const char base = *s; if (!base) break;
    const char strand = s[1];
    if (!std::strchr("ACGTUN", base) || (strand != '+' && strand != '-')) return false;  // "no": not a base-modification string (htslib refuses it)
    const char* q = s + 2;
    std::string codes;
    while (*q && *q != ',' && *q != ';' && *q != '.' && *q != '?') codes.push_back(*q++);
    if (codes.empty()) return false;
    // a numeric ChEBI code ("C+76792") is ONE modification, not one per digit
    if (codes.find_first_not_of("0123456789") == std::string::npos) codes = "#";
    if (*q == '.' || *q == '?') ++q;
    std::vector<uint32_t> deltas;
    while (*q == ',') { ++q; uint32_t v = 0; while (*q >= '0' && *q <= '9') v = v * 10 + (uint32_t)(*q++ - '0'); deltas.push_back(v); }
    if (*q == ';') ++q;
    s = q;
    const size_t n_codes = std::max<size_t>(1, codes.size());  // "C+mh": the ML values of a position are interleaved per code
    const size_t m_idx = codes.find('m');
    if (base == 'C' && strand == '+' && m_idx != std::string::npos) {
      const char want = reverse ? 'G' : 'C';
      size_t i = 0; bool ok = true;  // index in original-strand order
      std::vector<std::pair<uint32_t, uint8_t>> found;
      for (size_t k = 0; k < deltas.size() && ok; ++k) {
        uint32_t skip = deltas[k];
        for (;; ++i) {
          if (i >= n) { ok = false; break; }
          const size_t at = reverse ? n - 1 - i : i;
          if (bases[at] == want) { if (skip == 0) break; --skip; }
        }
        if (!ok) break;
        const size_t at = reverse ? n - 1 - i : i;
        const uint32_t mi = ml_at + (uint32_t)(k * n_codes + m_idx);
        if (mi < n_ml) found.emplace_back((uint32_t)at, mlv[mi]);
        ++i;
      }
      if (reverse) std::reverse(found.begin(), found.end());
      out.insert(out.end(), found.begin(), found.end());
    }
    ml_at += (uint32_t)(deltas.size() * n_codes);
  }
  return true;
}

// get_meth (read.rs:55-96): one value per CpG of the stored sequence (0 where no call sits on it); false = None (no call on any CpG)
bool meth_per_cpg(const std::string& bases, bool reverse, const std::vector<std::pair<uint32_t, uint8_t>>& mods, std::vector<uint8_t>& meth) {
  std::vector<size_t> cpg;
  for (size_t i = 0; i + 1 < bases.size(); ++i) if (bases[i] == 'C' && bases[i + 1] == 'G') cpg.push_back(i + (reverse ? 1 : 0));
  std::vector<uint8_t> ans(cpg.size(), 0);
  size_t ind = 0;
  for (auto& m : mods) {
    while (ind < cpg.size() && cpg[ind] < m.first) ++ind;
    if (ind < cpg.size() && m.first == cpg[ind]) { ans[ind] = m.second; ++ind; }
  }
  if (ind == 0) return false;
  if (reverse) std::reverse(ans.begin(), ans.end());
  meth.swap(ans);
  return true;
}
// extract_snps_offset (snp.rs:51-79): X runs that start outside [start, end] of the region, relative to the nearer end
void snps_offset(const uint32_t* cigar, size_t n_ops, int64_t pos, int64_t region_start, int64_t region_end, std::vector<int32_t>& out) {
  uint32_t start_ref = (uint32_t)pos;
  for (size_t k = 0; k < n_ops; ++k) {
    const uint32_t op = cigar[k];
    const int c = (int)(op & 0xF); const uint32_t n = op >> 4;
    const bool inside = (int64_t)start_ref >= region_start && (int64_t)start_ref <= region_end;
    if (c == OP_X && !inside) {
      const int32_t diff = (int64_t)start_ref < region_start ? (int32_t)start_ref - (int32_t)region_start : (int32_t)start_ref - (int32_t)region_end;
      for (uint32_t i = 0; i < n; ++i) out.push_back(diff + (int32_t)i);
      start_ref += n;
    } else if (c == OP_M || c == OP_X || c == OP_EQ || c == OP_D || c == OP_N) start_ref += n;
  }
}

// HiFiRead::from_hts_rec (read.rs:98-141)
void make_read(const RawRec& r, int64_t region_start, int64_t region_end, Read& out) {
  static const char code[] = "=ACMGRSVTWYHKDBN";
  const uint8_t* d = r.d.data();
  out.id.assign((const char*)d + 32, r.l_rn ? r.l_rn - 1 : 0);
  out.is_reverse = (r.flag & 0x10) != 0;
  out.bases.resize((size_t)r.l_seq);
  for (int32_t i = 0; i < r.l_seq; ++i) out.bases[(size_t)i] = code[(d[r.o_seq + (size_t)(i >> 1)] >> ((i & 1) ? 0 : 4)) & 0xF];
  out.quals.assign(d + r.o_qual, d + r.o_qual + r.l_seq);
  out.mapq = r.mapq;
  { const uint8_t* a = find_aux(r, "HP"); out.hp = a && a[0] == 'C' ? (int)a[1] : -1; }  // get_hp_tag (read.rs:167-172): Aux::U8 only
  out.rq = std::numeric_limits<double>::quiet_NaN();  // (a reservoir slot is overwritten in place: a read without the tag must not keep its predecessor's value -- found by tests/tools/ingest_fuzz.py, round 6)
  { const uint8_t* a = find_aux(r, "rq"); float f; if (a && a[0] == 'f') { std::memcpy(&f, a + 1, 4); out.rq = (double)f; } }
  {  // get_meth (read.rs:55-96)
    std::vector<std::pair<uint32_t, uint8_t>> mods;
    out.has_meth = false; out.meth.clear();
    if (basemods_5mc(r, out.bases, out.is_reverse, mods)) out.has_meth = meth_per_cpg(out.bases, out.is_reverse, mods, out.meth);
  }
  out.has_cigar = !(r.flag & 0x4);
  out.ref_pos = r.pos;
  out.cigar.resize(out.has_cigar ? r.n_cig : 0);
  for (size_t i = 0; i < out.cigar.size(); ++i) out.cigar[i] = le32(d + r.o_cig + 4 * i);
  out.start_offset = (int32_t)((int64_t)r.pos - region_start);
  out.end_offset = (int32_t)(rec_ref_end(r) - region_end);
  out.mismatch_offsets.clear();
  if (out.has_cigar) snps_offset(out.cigar.data(), out.cigar.size(), r.pos, region_start, region_end, out.mismatch_offsets);
}

// clip_cigar + HiFiRead::clip_to_region (clip_region.rs:19-184); false = no overlap (or no alignment)
bool clip_to_region(const Read& in, int64_t rs, int64_t re, Read& out) {
  if (!in.has_cigar) return false;
  int64_t read_end = in.ref_pos;
  for (uint32_t op : in.cigar) read_end += ref_len(op);
  if (read_end <= rs || re <= in.ref_pos) return false;
  int64_t ref_pos = in.ref_pos, query_pos = 0;
  size_t i = 0;
  std::vector<uint32_t> ops;
  while (i < in.cigar.size() && ref_pos + ref_len(in.cigar[i]) <= rs) { ref_pos += ref_len(in.cigar[i]); query_pos += qry_len(in.cigar[i]); ++i; }
  int64_t c_ref = ref_pos, c_qry = query_pos;
  if (ref_pos < rs) {  // the operation that straddles the start of the region is split (only reference-consuming ones can)
    const uint32_t op = in.cigar[i];
    const int64_t outside = rs - ref_pos;
    const int64_t keep = ref_pos + ref_len(op) <= re ? ref_len(op) - outside : re - rs;
    const uint32_t part = ((uint32_t)keep << 4) | (op & 0xF);
    ops.push_back(part);
    c_ref += outside;
    if (qry_len(part) != 0) c_qry += outside;
    ref_pos += ref_len(op); query_pos += qry_len(op); ++i;
  }
  while (i < in.cigar.size() && ref_pos + ref_len(in.cigar[i]) <= re) { ops.push_back(in.cigar[i]); ref_pos += ref_len(in.cigar[i]); query_pos += qry_len(in.cigar[i]); ++i; }
  if (i < in.cigar.size() && ref_pos < re) ops.push_back(((uint32_t)(re - ref_pos) << 4) | (in.cigar[i] & 0xF));
  out = Read();
  out.id = in.id; out.is_reverse = in.is_reverse; out.rq = in.rq; out.mismatch_offsets = in.mismatch_offsets;
  out.start_offset = in.start_offset; out.end_offset = in.end_offset; out.hp = in.hp; out.mapq = in.mapq;
  int64_t q = c_qry;
  for (uint32_t op : ops) {
    const int64_t n = qry_len(op);
    if (q + n > (int64_t)in.bases.size()) return false;  // (a CIGAR longer than its sequence: not a record htslib would hand out)
    out.bases.append(in.bases, (size_t)q, (size_t)n);
    out.quals.insert(out.quals.end(), in.quals.begin() + q, in.quals.begin() + q + n);
    q += n;
  }
  const int64_t q_end = q;
  if (in.has_meth) {  // the calls of the CpGs whose C lies inside the clipped part
    size_t mi = 0;
    out.has_meth = true;
    for (size_t idx = 0; idx + 1 < in.bases.size(); ++idx)
      if (in.bases[idx] == 'C' && in.bases[idx + 1] == 'G') {
        if (mi >= in.meth.size()) break;
        if ((int64_t)idx >= c_qry && (int64_t)idx < q_end) out.meth.push_back(in.meth[mi]);
        ++mi;
      }
  }
  out.has_cigar = true; out.ref_pos = c_ref; out.cigar.swap(ops);
  return true;
}

// ---------------------------------------------------------------------------------------------- StdRng::seed_from_u64(42).random_range
// The reservoir of extract_reads (tr.rs:311-335) draws from rand 0.9's StdRng = ChaCha12 seeded through rand_core's seed_from_u64
// (a PCG32 stream fills the 32-byte key).  rand is not vendored in the reference tree: the stream below restates its published
// algorithms (ChaCha block function, 64-word buffer read in order, Canon's widening-multiply range sampling for 32-bit and 64-bit
// ranges) -- parity UNPINNED (no fixture in the reference reaches this path).
struct StdRng {
  uint32_t key[8]; uint64_t counter = 0; uint32_t buf[64]; int at = 64;
  explicit StdRng(uint64_t state) {
    const uint64_t MUL = 6364136223846793005ull, INC = 11634580027462260723ull;
    for (int i = 0; i < 8; ++i) {
      state = state * MUL + INC;
      const uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27), rot = (uint32_t)(state >> 59);
      key[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
    }
  }
  static inline uint32_t rotl(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }
  void refill() {
    for (int b = 0; b < 4; ++b) {
      uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                        (uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
      uint32_t x[16];
      std::memcpy(x, s, sizeof x);
      auto qr = [&](int a, int bb, int c, int d) {
        x[a] += x[bb]; x[d] = rotl(x[d] ^ x[a], 16); x[c] += x[d]; x[bb] = rotl(x[bb] ^ x[c], 12);
        x[a] += x[bb]; x[d] = rotl(x[d] ^ x[a], 8); x[c] += x[d]; x[bb] = rotl(x[bb] ^ x[c], 7);
      };
      for (int r = 0; r < 6; ++r) { qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15); qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14); }
      for (int i = 0; i < 16; ++i) buf[16 * b + i] = x[i] + s[i];
      ++counter;
    }
    at = 0;
  }
  uint32_t next_u32() { if (at >= 64) refill(); return buf[at++]; }
  uint64_t next_u64() {
    if (at < 63) { const uint64_t lo = buf[at], hi = buf[at + 1]; at += 2; return (hi << 32) | lo; }
    if (at >= 64) { refill(); const uint64_t lo = buf[0], hi = buf[1]; at = 2; return (hi << 32) | lo; }
    const uint64_t lo = buf[63]; refill(); const uint64_t hi = buf[0]; at = 1; return (hi << 32) | lo;
  }
  uint64_t range(uint64_t n) {  // 0 .. n (exclusive), n >= 1
    if (n <= 0xFFFFFFFFull) {
      const uint32_t r = (uint32_t)n;
      uint64_t m = (uint64_t)next_u32() * r;
      uint32_t hi = (uint32_t)(m >> 32); const uint32_t lo = (uint32_t)m;
      if (lo > (uint32_t)(0u - r)) { const uint32_t hi2 = (uint32_t)(((uint64_t)next_u32() * r) >> 32); if ((uint64_t)lo + hi2 > 0xFFFFFFFFull) ++hi; }
      return hi;
    }
    const unsigned __int128 m = (unsigned __int128)next_u64() * n;
    uint64_t hi = (uint64_t)(m >> 64); const uint64_t lo = (uint64_t)m;
    if (lo > 0ull - n) { const uint64_t hi2 = (uint64_t)(((unsigned __int128)next_u64() * n) >> 64); if (lo + hi2 < lo) ++hi; }
    return hi;
  }
};

}  // namespace

// ============================================================================================== C ABI
struct trgt_ingest {
  std::string bam_path, fasta_path, err, header_text;
  std::vector<std::string> ref_names;
  std::vector<uint32_t> ref_len;
  std::map<std::string, int> ref_id;
  uint64_t first_record_voff = 0;
  Bai bai;
  Fasta fasta;
  // readers of the worker threads, kept between calls: an open file and the last inflated blocks (3 MB each) -- a call that continues
  // where the last one stopped finds the blocks of the chunk boundary inflated, and no call pays for fresh pages again
  std::mutex idle_mu;
  std::vector<std::unique_ptr<Bgzf>> idle_readers;
  // trgt_ingest_params.ingest_device (ABI 10): slots of device state (a stream, device buffers, pinned staging) -- a call takes a free one, so
  // calls from several host threads overlap their file reads, uploads, kernels and downloads -- and the pool the batches' slabs return to
  static constexpr int DEV_SLOTS = 6;
  std::mutex dev_mu; std::condition_variable dev_cv;
  struct DevSlot { trgt::ingd::Slot* s = nullptr; bool busy = false; };
  DevSlot dev_slots[DEV_SLOTS]; int dev_device = -1;
  std::shared_ptr<trgt::ingd::SlabPool> slab_pool = std::make_shared<trgt::ingd::SlabPool>();
  std::atomic<int64_t> dev_calls{0}, dev_fallbacks{0}, dev_last_reason{0}, dev_blocks{0}, dev_blocks_host{0};
  ~trgt_ingest() {
    for (auto& d : dev_slots) if (d.s) trgt::ingd::slot_destroy(d.s);
  }
};

// ---- trgt_ingest_params.ingest_device: the reads of a batch of loci through the kernels of ingest_dev.hip.  Host work: the .bai chunks of
// every locus, the compressed ranges they span read into pinned memory (a few threads), the BGZF headers walked (block table, footers'
// CRC-32 / ISIZE; the blocks of a range are laid end to end in the inflated buffer, so a record may run across blocks as in the file),
// every chunk's virtual offsets turned into positions in that buffer.
// rc TRGT_OK: res.fallback == 0 -> the arrays of `res` are the batch's; != 0 -> the host path runs (and reports what is wrong); < 0: error
struct DevLocusIn { int tid; int64_t start, end; };
static int device_reads(trgt_ingest* h, const trgt_ingest_params* p, const std::vector<DevLocusIn>& dl, trgt::ingd::RunOut& res, std::string& err, bool trace) {
  namespace ingd = trgt::ingd;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  auto bad = [&](const std::string& m) { err = m; return TRGT_ERR_INVALID; };
  // a free slot of device state
  int si = -1;
  {
    std::unique_lock<std::mutex> g(h->dev_mu);
    if (h->dev_device != p->ingest_device) {
      h->dev_cv.wait(g, [&] { for (auto& d : h->dev_slots) if (d.busy) return false; return true; });
      for (auto& d : h->dev_slots) if (d.s) { ingd::slot_destroy(d.s); d.s = nullptr; }
      h->dev_device = p->ingest_device;
    }
    h->dev_cv.wait(g, [&] { for (int i = 0; i < trgt_ingest::DEV_SLOTS; ++i) if (!h->dev_slots[i].busy) { si = i; return true; } return false; });
    if (!h->dev_slots[si].s && !(h->dev_slots[si].s = ingd::slot_create(p->ingest_device, err))) return TRGT_ERR_NO_DEVICE;
    h->dev_slots[si].busy = true;
  }
  struct Release { trgt_ingest* h; int si; ~Release() { { std::lock_guard<std::mutex> g(h->dev_mu); h->dev_slots[si].busy = false; } h->dev_cv.notify_all(); } } release{h, si};
  ingd::Slot* slot = h->dev_slots[si].s;
  const size_t nl = dl.size();
  std::vector<ingd::LocusDesc> ld(nl);
  std::vector<std::pair<uint64_t, uint64_t>> cv;  // the chunks' virtual offsets
  for (size_t li = 0; li < nl; ++li) {
    ingd::LocusDesc& d = ld[li];
    d.tid = dl[li].tid; d.pad = 0;
    d.beg = std::max<int64_t>(0, dl[li].start - p->flank_len); d.end = dl[li].end + p->flank_len;
    d.region_start = dl[li].start; d.region_end = dl[li].end;
    d.clip_start = dl[li].start - 2ll * p->flank_len; d.clip_end = dl[li].end + 2ll * p->flank_len;
    d.chunk_begin = (int32_t)cv.size();
    if (d.tid >= 0) for (auto& ch : h->bai.query(d.tid, d.beg, d.end)) cv.push_back(ch);
    d.chunk_end = (int32_t)cv.size();
  }
  std::vector<std::pair<uint64_t, uint64_t>> ranges, merged;
  for (auto& c : cv) ranges.emplace_back(c.first >> 16, c.second >> 16);
  std::sort(ranges.begin(), ranges.end());
  for (auto& r : ranges) {
    if (!merged.empty() && r.first <= merged.back().second + 0x10000) merged.back().second = std::max(merged.back().second, r.second);
    else merged.push_back(r);
  }
  const int fd = ::open(h->bam_path.c_str(), O_RDONLY);
  if (fd < 0) return bad("cannot open " + h->bam_path);
  struct Fd { int fd; ~Fd() { ::close(fd); } } fdg{fd};
  struct stat st;
  if (::fstat(fd, &st) != 0) return bad("cannot stat " + h->bam_path);
  const uint64_t fsize = (uint64_t)st.st_size;
  uint64_t src_total = 0;
  std::vector<uint64_t> src_at(merged.size());
  for (size_t i = 0; i < merged.size(); ++i) {
    const uint64_t c0 = merged[i].first, c1 = std::min<uint64_t>(fsize, merged[i].second + 0x10000 + 64);
    src_at[i] = src_total;
    if (c1 > c0) src_total += ((c1 - c0) + 63) & ~63ull;
  }
  uint8_t* const src = ingd::slot_src(slot, (size_t)src_total + 64, err);
  if (!src) return TRGT_ERR_NOMEM;
  {  // the compressed ranges, read by a few threads (page cache or disk), in pieces
    struct Piece { uint64_t off; uint8_t* dst; size_t n; };
    std::vector<Piece> pieces;
    for (size_t i = 0; i < merged.size(); ++i) {
      const uint64_t c0 = merged[i].first, c1 = std::min<uint64_t>(fsize, merged[i].second + 0x10000 + 64);
      for (uint64_t o = c0; o < c1; o += 4u << 20) pieces.push_back({o, src + src_at[i] + (o - c0), (size_t)std::min<uint64_t>(c1 - o, 4u << 20)});
    }
    std::atomic<size_t> next{0}; std::atomic<int> failed{0};
    auto rd = [&]() {
      for (;;) {
        const size_t i = next.fetch_add(1);
        if (i >= pieces.size()) break;
        for (size_t done = 0; done < pieces[i].n;) {
          const ssize_t g = ::pread(fd, pieces[i].dst + done, pieces[i].n - done, (off_t)(pieces[i].off + done));
          if (g <= 0) { failed = 1; return; }
          done += (size_t)g;
        }
      }
    };
    const int want = p->threads > 0 ? p->threads : 8;
    const int nt = (int)std::min<size_t>((size_t)std::max(1, std::min(want, 8)), pieces.size());
    if (nt <= 1) rd();
    else { std::vector<std::thread> th; for (int t = 0; t < nt; ++t) th.emplace_back(rd); for (auto& t : th) t.join(); }
    if (failed) return bad("trgt_ingest: reading the compressed blocks failed");
  }
  const double t_read = now();
  // ---- block table: payloads, where they inflate to, the footers
  std::vector<trgt::infl::BlockDesc> blocks; std::vector<uint32_t> crcs;
  struct Known { uint64_t coff, lin; uint32_t isize, range; };
  std::vector<Known> known;  // every block met (also empty ones), sorted by coff
  std::vector<uint64_t> lin_end(merged.size());
  uint64_t lin = 0;
  auto fall = [&](int why) { res = ingd::RunOut(); res.fallback = why; return TRGT_OK; };
  for (size_t i = 0; i < merged.size(); ++i) {
    const uint64_t c0 = merged[i].first, cend = std::min<uint64_t>(fsize, merged[i].second + 0x10000 + 64);
    uint64_t coff = c0;
    while (coff <= merged[i].second && coff < fsize) {
      if (coff + 18 > cend) return fall(ingd::FB_BLOCK);
      const uint8_t* hp = src + src_at[i] + (coff - c0);
      if (hp[0] != 31 || hp[1] != 139 || hp[2] != 8 || !(hp[3] & 4)) return fall(ingd::FB_BLOCK);  // (the host path says what is wrong)
      const uint32_t xlen = hp[10] | (hp[11] << 8);
      if (coff + 12 + xlen > cend) return fall(ingd::FB_BLOCK);
      uint32_t bsize = 0; bool found = false;
      for (uint32_t k = 0; k + 4 <= xlen;) {
        const uint32_t slen = hp[12 + k + 2] | (hp[12 + k + 3] << 8);
        if (hp[12 + k] == 'B' && hp[12 + k + 1] == 'C' && slen == 2 && k + 6 <= xlen) { bsize = hp[12 + k + 4] | (hp[12 + k + 5] << 8); found = true; break; }
        k += 4 + slen;
      }
      if (!found) return fall(ingd::FB_BLOCK);
      const uint32_t total = bsize + 1, hdr = 12 + xlen;
      if (total < hdr + 8 || coff + total > cend) return fall(ingd::FB_BLOCK);
      const uint32_t isize = le32(hp + total - 4);
      if (isize > 0x10000) return fall(ingd::FB_BLOCK);
      known.push_back(Known{coff, lin, isize, (uint32_t)i});
      if (isize) {
        blocks.push_back(trgt::infl::BlockDesc{src_at[i] + (coff - c0) + hdr, lin, total - hdr - 8, isize});
        crcs.push_back(le32(hp + total - 8));
        lin += isize;
      }
      coff += total;
    }
    lin_end[i] = lin;
    lin = ((lin + 63) & ~63ull) + 64;  // (a gap between the ranges: nothing of one is read as the other's)
  }
  const double t_walk = now();
  // ---- the chunks as positions in the inflated bytes
  std::vector<ingd::ChunkDesc> cd(cv.size());
  auto find = [&](uint64_t coff) -> const Known* {
    auto it = std::lower_bound(known.begin(), known.end(), coff, [](const Known& k, uint64_t c) { return k.coff < c; });
    return it != known.end() && it->coff == coff ? &*it : nullptr;
  };
  for (size_t c = 0; c < cv.size(); ++c) {
    const Known* a = find(cv[c].first >> 16);
    if (!a || (cv[c].first & 0xFFFF) > a->isize) return fall(ingd::FB_WALK);
    const Known* b = find(cv[c].second >> 16);
    cd[c].lin0 = a->lin + (cv[c].first & 0xFFFF);
    cd[c].lin_limit = lin_end[a->range];
    cd[c].lin1 = b && b->range == a->range ? std::min<uint64_t>(b->lin + std::min<uint64_t>(cv[c].second & 0xFFFF, b->isize), cd[c].lin_limit) : cd[c].lin_limit;
  }
  ingd::RunIn in;
  in.src_bytes = src_total; in.n_blocks = (int64_t)blocks.size(); in.blocks = blocks.data(); in.crc = crcs.data(); in.infl_bytes = lin;
  in.n_loci = (int64_t)nl; in.loci = ld.data(); in.n_chunks = (int64_t)cd.size(); in.chunks = cd.data();
  in.reservoir = (uint32_t)(3ll * p->max_depth); in.min_rq = p->min_read_qual; in.keep_bam4 = p->keep_bam4 != 0; in.waves_per_cu = p->inflate_waves_per_cu;
  const int rc = ingd::slot_run(slot, in, *h->slab_pool, res, err);
  if (rc) return rc;
  h->dev_blocks += (int64_t)blocks.size(); h->dev_blocks_host += (int64_t)res.blocks_host_inflated;
  if (trace) std::fprintf(stderr, "[ingest]   device: %zu ranges, %zu blocks (%.1f MB -> %.1f MB), file %.1f ms, headers %.1f ms, upload+inflate+crc %.1f ms (%llu blocks by zlib), walk %.1f ms, read sizes %.1f ms, slab + fill launch %.1f ms, fill + download %.1f ms, fallback %d\n",
                          merged.size(), blocks.size(), (double)src_total / 1e6, (double)lin / 1e6, t_read - t0, t_walk - t_read, res.ms_inflate, (unsigned long long)res.blocks_host_inflated, res.ms_walk, res.ms_reads, res.ms_upload, res.ms_download, res.fallback);
  return TRGT_OK;
}

struct BatchStore {  // owner of the arrays a trgt_ingest_batch points to
  std::string flank, tr, motifs, reads, quals, names, contigs, ids, strucs;
  std::vector<uint64_t> lf_off, rf_off, tr_off, lrb, read_off, name_off, moff, snp_off;
  std::vector<uint32_t> lf_len, rf_len, tr_len, motif_off, set_begin, read_len;
  std::vector<uint8_t> ploidy, genotyper, is_reverse, mapq, meth, has_meth;
  std::vector<int16_t> hp;
  std::vector<int32_t> start_offset, end_offset, snp, n_filtered;
  std::vector<int64_t> n_seen, region_start, region_end;
  std::vector<double> rq;
  std::vector<uint64_t> contig_off, id_off, struc_off, cig_off;
  std::vector<uint32_t> cig;
  std::vector<int64_t> cig_ref_pos;
  std::vector<uint8_t> bam4;
  std::vector<uint64_t> bam4_off;
  std::string skipped; std::vector<uint64_t> skipped_off;
  // ingest_device: the per-read arrays are pieces of one device slab + its pinned mirror (back to the reader's pool when the batch is freed)
  trgt::ingd::Slab slab; std::shared_ptr<trgt::ingd::SlabPool> slab_pool;
  ~BatchStore() { if (slab_pool) slab_pool->give(slab); }
  trgt_ingest_batch pub;
};

extern "C" {

const char* trgt_ingest_last_error(const trgt_ingest* h) { return h ? h->err.c_str() : "null handle"; }

// (for the writers, trgt_amd/csrc/writers.hip: the template header and the contigs of the BAM)
const char* trgt_ingest_header_text(const trgt_ingest* h) { return h ? h->header_text.c_str() : ""; }
int32_t trgt_ingest_n_contigs(const trgt_ingest* h) { return h ? (int32_t)h->ref_names.size() : 0; }
const char* trgt_ingest_contig_name(const trgt_ingest* h, int32_t i) { return h && i >= 0 && (size_t)i < h->ref_names.size() ? h->ref_names[(size_t)i].c_str() : ""; }
uint32_t trgt_ingest_contig_length(const trgt_ingest* h, int32_t i) { return h && i >= 0 && (size_t)i < h->ref_len.size() ? h->ref_len[(size_t)i] : 0; }

static int ingest_open_impl(const char* bam_path, const char* fasta_path, trgt_ingest** out) {
  if (!bam_path || !fasta_path || !out) return TRGT_ERR_INVALID;
  std::unique_ptr<trgt_ingest> h(new trgt_ingest());
  *out = nullptr;
  h->bam_path = bam_path; h->fasta_path = fasta_path;
  auto bad = [&](const std::string& m) { h->err = m; *out = h.release(); return TRGT_ERR_INVALID; };  // (the handle carries the message)
  Bgzf z;
  if (!z.open(bam_path) || !z.load(0)) return bad(z.err.empty() ? "cannot read the BAM" : z.err);
  uint8_t b[8];
  if (z.read(b, 8) != 1 || std::memcmp(b, "BAM\1", 4) != 0) return bad("not a BAM file");
  const uint32_t l_text = le32(b + 4);
  if (l_text > (1u << 30)) return bad("corrupt BAM header");
  std::vector<uint8_t> text(l_text);
  if (l_text && z.read(text.data(), l_text) != 1) return bad("truncated BAM header");
  h->header_text.assign(text.begin(), text.end());
  while (!h->header_text.empty() && h->header_text.back() == '\0') h->header_text.pop_back();
  if (z.read(b, 4) != 1) return bad("truncated BAM header");
  const uint32_t n_ref = le32(b);
  if (n_ref > (1u << 24)) return bad("corrupt BAM header");
  for (uint32_t r = 0; r < n_ref; ++r) {
    if (z.read(b, 4) != 1) return bad("truncated BAM header");
    const uint32_t l_name = le32(b);
    if (l_name > 65536) return bad("corrupt BAM header");
    std::string name(l_name, '\0');
    if (l_name && z.read(&name[0], l_name) != 1) return bad("truncated BAM header");
    if (z.read(b, 4) != 1) return bad("truncated BAM header");
    h->ref_len.push_back(le32(b));
    if (!name.empty() && name.back() == '\0') name.pop_back();
    h->ref_id[name] = (int)r;
    h->ref_names.push_back(name);
  }
  h->first_record_voff = z.tell();
  std::string e;
  if (!h->bai.load(std::string(bam_path) + ".bai", e)) return bad(e);
  if (!h->fasta.open(fasta_path, e)) return bad(e);
  *out = h.release();
  return TRGT_OK;
}

void trgt_ingest_close(trgt_ingest* h) { delete h; }

// ABI 10: what trgt_ingest_params.ingest_device did so far -- [0] calls that asked for the device, [1] of those, calls that went back to the
// host path, [2] the reason of the last one (trgt::ingd::FB_*: 1 block, 2 record walk, 4 MM / ML caps), [3] BGZF blocks through
// the device path, [4] of those, blocks the inflate kernel declined (inflated by zlib, uploaded)
void trgt_ingest_device_stats(const trgt_ingest* h, int64_t out[5]) {
  if (!h || !out) return;
  out[0] = h->dev_calls.load(); out[1] = h->dev_fallbacks.load(); out[2] = h->dev_last_reason.load(); out[3] = h->dev_blocks.load(); out[4] = h->dev_blocks_host.load();
}

void trgt_ingest_default_params(trgt_ingest_params* p) {
  if (!p) return;
  p->flank_len = 250; p->max_depth = 250; p->min_read_qual = 0.98; p->threads = 0; p->genotyper = 0; p->default_ploidy = 2; p->keep_bam4 = 0; p->ingest_device = -1; p->inflate_waves_per_cu = 0;
}

void trgt_ingest_free(trgt_ingest_batch* b) {
  if (!b) return;
  delete reinterpret_cast<BatchStore*>(b->owner);
}

// one line of the repeat catalog (BED: contig, start, end, ID=..;MOTIFS=..;STRUC=..), locus.rs:31-98
// Locus::new (locus.rs:31-60) in its order: the four fields and GenomicRegion::from_string (parse_bed_region), then -- by the caller --
// check_region_bounds, then decode_fields / get_field (parse_bed_info): a line with a bounds problem AND a bad info field reports the bounds
static bool parse_bed_region(const std::string& line, std::string& contig, int64_t& start, int64_t& end, std::string& info, std::string& err) {
  std::vector<std::string> f;
  { std::istringstream ss(line); std::string t; while (ss >> t) f.push_back(t); }
  if (f.size() != 4) { err = "Expected 4 fields in the format 'chrom start end info', found " + std::to_string(f.size()) + ": " + line; return false; }
  contig = f[0];
  // GenomicRegion::from_string (utils/region.rs:23-35): "contig:start-end" split at ':' and '-' must give three elements, the
  // coordinates parse as u32 (digits with an optional '+', no sign, no trailing text: "12abc" and "-3" are refused), start < end
  auto u32_of = [](const std::string& t, int64_t& v) {
    size_t i = !t.empty() && t[0] == '+' ? 1 : 0;
    if (i >= t.size()) return false;
    uint64_t x = 0;
    for (; i < t.size(); ++i) { if (t[i] < '0' || t[i] > '9') return false; x = x * 10 + (uint64_t)(t[i] - '0'); if (x > 0xFFFFFFFFull) return false; }
    v = (int64_t)x; return true;
  };
  const std::string enc = f[0] + ":" + f[1] + "-" + f[2];
  if (f[0].find_first_of(":-") != std::string::npos || f[1].find_first_of(":-") != std::string::npos || f[2].find_first_of(":-") != std::string::npos ||
      !u32_of(f[1], start) || !u32_of(f[2], end)) { err = "Invalid region encoding: " + enc; return false; }
  if (start >= end) { err = "Invalid region: start " + std::to_string(start) + " >= end " + std::to_string(end); return false; }
  info = f[3];
  return true;
}
static bool parse_bed_info(const std::string& info, std::string& id, std::vector<std::string>& motifs, std::string& struc, std::string& err) {
  std::map<std::string, std::string> fields;
  { std::istringstream ss(info); std::string kv;
    while (std::getline(ss, kv, ';')) {
      const size_t eq = kv.find('=');
      if (eq == std::string::npos || eq == 0 || eq + 1 >= kv.size()) { err = "Field must be in 'name=value' format: '" + kv + "'"; return false; }
      if (!fields.emplace(kv.substr(0, eq), kv.substr(eq + 1)).second) { err = "Duplicate field name: '" + kv.substr(0, eq) + "'"; return false; }
    } }
  for (const char* key : {"ID", "MOTIFS", "STRUC"}) if (!fields.count(key)) { err = std::string(key) + " field missing"; return false; }
  id = fields["ID"]; struc = fields["STRUC"];
  motifs.clear();
  { std::istringstream ss(fields["MOTIFS"]); std::string m; while (std::getline(ss, m, ',')) motifs.push_back(m); }
  return true;
}

static int ingest_batch_impl(trgt_ingest* h, const trgt_ingest_params* p, const char* bed_path, int64_t first_locus, int64_t max_loci,
                             trgt_ingest_batch** out) {
  if (!h || !p || !bed_path || !out) return TRGT_ERR_INVALID;
  *out = nullptr;
  auto bad = [&](const std::string& m) { h->err = m; return TRGT_ERR_INVALID; };
  if (p->flank_len <= 0 || p->max_depth <= 0) return bad("trgt_ingest: flank_len and max_depth must be positive");
  std::ifstream bed(bed_path);
  if (!bed) return bad(std::string("cannot open ") + bed_path);
  const bool trace = std::getenv("TRGT_INGEST_TRACE") != nullptr;  // phase times on stderr
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  struct L { std::string contig, id, struc; int64_t start, end; std::vector<std::string> motifs; std::string lf, tr, rf; std::vector<Read> reads; int32_t n_filt = 0; int64_t n_seen = 0; std::string err; };
  std::vector<L> loci;
  std::vector<std::string> skipped;  // one message per catalog line that gave no locus
  {
    // stream_loci_into_channel (locus.rs:93-137): a line that does not give a Locus is reported ("Error at BED line N: ...") and the
    // next line is read; blank lines are lines like any other ("Expected 4 fields ..., found 0").  first_locus / max_loci count
    // catalog LINES.
    std::string line; int64_t line_no = 0;
    while (std::getline(bed, line)) {
      ++line_no;
      if (line_no - 1 < first_locus) continue;
      if (max_loci >= 0 && line_no - 1 >= first_locus + max_loci) break;
      if (!line.empty() && line.back() == '\r') line.pop_back();
      L l; std::string e;
      auto skip = [&](const std::string& m) { skipped.push_back("Error at BED line " + std::to_string(line_no) + ": " + m); };
      std::string info;
      if (!parse_bed_region(line, l.contig, l.start, l.end, info, e)) { skip(e); continue; }
      // check_region_bounds (locus.rs:220-257)
      const int64_t chrom_len = h->fasta.length(l.contig);
      if (chrom_len < 0) { skip("FASTA reference does not contain chromosome '" + l.contig + "' in BED file"); continue; }
      if (l.start < (int64_t)p->flank_len + 1) { skip("Region start '" + std::to_string(l.start) + "' with flank length '" + std::to_string(p->flank_len) + "' underflows for chromosome '" + l.contig + "'."); continue; }
      if (l.end + p->flank_len > 0xFFFFFFFFll) { skip("Region end '" + std::to_string(l.end) + "' with flank length '" + std::to_string(p->flank_len) + "' overflows for chromosome '" + l.contig + "'."); continue; }
      if (l.end + p->flank_len > chrom_len) { skip("Region end '" + std::to_string(l.end + p->flank_len) + "' with flank length '" + std::to_string(p->flank_len) + "' exceeds chromosome '" + l.contig + "' bounds (0.." + std::to_string(chrom_len) + ")."); continue; }
      if (!parse_bed_info(info, l.id, l.motifs, l.struc, e)) { skip(e); continue; }
      // get_tr_and_flanks (locus.rs:168-190)
      if (!h->fasta.fetch(l.contig, l.start - p->flank_len, l.start, l.lf, e) || !h->fasta.fetch(l.contig, l.start, l.end, l.tr, e) ||
          !h->fasta.fetch(l.contig, l.end, l.end + p->flank_len, l.rf, e)) { skip(e); continue; }
      loci.push_back(std::move(l));
    }
  }
  const int64_t nl = (int64_t)loci.size();
  const double t1 = now();
  // ---- trgt_ingest_params.ingest_device: the reads of these loci through the kernels of ingest_dev.hip; what they do not take (res.fallback)
  // goes through the host path below, which yields the data or the error message
  trgt::ingd::RunOut dev;
  bool from_device = false;
  if (p->ingest_device >= 0 && nl > 0) {
    std::vector<DevLocusIn> dl((size_t)nl);
    for (int64_t li = 0; li < nl; ++li) { auto it = h->ref_id.find(loci[(size_t)li].contig); dl[(size_t)li] = DevLocusIn{it == h->ref_id.end() ? -1 : it->second, loci[(size_t)li].start, loci[(size_t)li].end}; }
    std::string derr;
    const int rc = device_reads(h, p, dl, dev, derr, trace);
    if (rc) { h->err = derr.empty() ? "trgt_ingest: device ingestion failed" : derr; return rc; }  // a device that cannot be used fails the call: no silent host-only run
    ++h->dev_calls;
    if (dev.fallback) { ++h->dev_fallbacks; h->dev_last_reason = dev.fallback; }
    else from_device = true;
  }
  // ---- reads: extract_reads + clip_reads per locus, loci spread over threads (one file handle each)
  int nthr = p->threads > 0 ? p->threads : (int)std::max(1u, std::min(32u, std::thread::hardware_concurrency()));  // (more than 32 workers lose: tools/ingest_scaling.py)
  nthr = (int)std::max<int64_t>(1, std::min<int64_t>(nthr, nl));
  std::atomic<int64_t> next{0};
  std::atomic<uint64_t> n_inflated{0}, n_cache_hits{0};
  // a worker takes a run of consecutive catalog lines (sorted catalogs: neighbours share BGZF blocks, see Bgzf), short enough that
  // every thread still gets several runs
  const int64_t run = std::max<int64_t>(1, std::min<int64_t>(8, nl / (4ll * nthr)));
  auto work_body = [&]() {
    std::unique_ptr<Bgzf> zp;
    { std::lock_guard<std::mutex> g(h->idle_mu); if (!h->idle_readers.empty()) { zp = std::move(h->idle_readers.back()); h->idle_readers.pop_back(); } }
    if (!zp) {
      zp.reset(new Bgzf(48));
      if (!zp->open(h->bam_path.c_str())) { for (auto& l : loci) if (l.err.empty()) { l.err = zp->err; break; } return; }
    }
    struct Back { trgt_ingest* h; std::unique_ptr<Bgzf>& z; ~Back() { if (z && z->err.empty()) { std::lock_guard<std::mutex> g(h->idle_mu); h->idle_readers.push_back(std::move(z)); } } } back{h, zp};
    Bgzf& z = *zp;
    z.block_coff = ~0ull; z.block_csize = 0; z.cur_data = nullptr; z.cur_size = 0; z.pos = 0;
    const uint64_t inflated0 = z.n_inflated, hits0 = z.n_hits;
    RawRec rec;
    for (;;) {
      const int64_t l0 = next.fetch_add(run);
      if (l0 >= nl) break;
     for (int64_t li = l0; li < std::min(nl, l0 + run); ++li) {
      L& l = loci[(size_t)li];
      auto it = h->ref_id.find(l.contig);
      if (it == h->ref_id.end()) continue;  // "Fetch error" is a warning in the reference: the locus gets no reads
      const int tid = it->second;
      const int64_t beg = std::max<int64_t>(0, l.start - p->flank_len), end = l.end + p->flank_len;
      const int64_t reservoir = 3ll * p->max_depth;
      std::unique_ptr<StdRng> rng;
      int64_t n_reads = 0;
      bool stop = false;
      for (auto& ch : h->bai.query(tid, beg, end)) {
        if (stop) break;
        if (!z.seek(ch.first)) { l.err = z.err; break; }
        while (z.tell() < ch.second) {
          uint8_t b4[4];
          const int g = z.read(b4, 4);
          if (g == 0) break;
          if (g < 0) { l.err = z.err; stop = true; break; }
          { const uint32_t bs = le32(b4); if (bs < 32 || bs > (1u << 29)) { l.err = "corrupt BAM record"; stop = true; break; } rec.d.resize(bs); }
          if (z.read(rec.d.data(), rec.d.size()) != 1 || !parse_rec(rec)) { l.err = z.err.empty() ? "corrupt BAM record" : z.err; stop = true; break; }
          if (rec.ref_id != tid || rec.pos >= end) { stop = true; break; }  // sorted: nothing further overlaps
          int64_t rend = rec_ref_end(rec);
          if (rend == rec.pos) rend = rec.pos + 1;
          if (rend <= beg) continue;
          if (rec.flag & (0x800 | 0x100)) continue;  // supplementary / secondary
          { const uint8_t* a = find_aux(rec, "rq"); float f = 1.0f; if (a && a[0] == 'f') std::memcpy(&f, a + 1, 4);
            if ((a && a[0] == 'f' ? (double)f : 1.0) < p->min_read_qual) { ++l.n_filt; continue; } }
          if (n_reads < reservoir) { l.reads.emplace_back(); make_read(rec, l.start, l.end, l.reads.back()); }
          else {  // the reservoir is full: every further read replaces a random one with probability reservoir / (n + 1)
            if (!rng) rng.reset(new StdRng(42));
            const uint64_t j = rng->range((uint64_t)n_reads);
            if ((int64_t)j < reservoir) make_read(rec, l.start, l.end, l.reads[(size_t)j]);
          }
          ++n_reads;
        }
      }
      l.n_seen = n_reads;
      // clip_reads (tr.rs:186-196): radius 2 * flank_len
      const int64_t rs = l.start - 2ll * p->flank_len, re = l.end + 2ll * p->flank_len;
      std::vector<Read> clipped;
      for (auto& r : l.reads) { Read c; if (clip_to_region(r, rs, re, c)) clipped.push_back(std::move(c)); }
      l.reads.swap(clipped);
     }
    }
    n_inflated += z.n_inflated - inflated0; n_cache_hits += z.n_hits - hits0;
  };
  std::atomic<int> worker_failed{0};
  auto work = [&]() {  // (an exception must not leave a thread, nor cross the C ABI)
    try { work_body(); } catch (const std::exception& e) { worker_failed = 1; for (auto& l : loci) if (l.err.empty()) { l.err = std::string("reading the BAM: ") + e.what(); break; } }
  };
  if (from_device) {}
  else if (nthr <= 1) work();
  else { std::vector<std::thread> th; for (int t = 0; t < nthr; ++t) th.emplace_back(work); for (auto& t : th) t.join(); }
  for (auto& l : loci) if (!l.err.empty()) return bad(l.id + ": " + l.err);
  const double t2 = now();
  // ---- the arrays of trgt_locus_batch_in (+ what the writers need per read)
  std::unique_ptr<BatchStore> S(new BatchStore());
  S->lrb.push_back(0); S->motif_off.push_back(0); S->set_begin.push_back(0);
  S->contig_off.push_back(0); S->id_off.push_back(0); S->struc_off.push_back(0);
  // per locus: the catalog fields, and where its reads go in the per-read arrays (sizes first, then the loci are copied in by the workers)
  struct At { uint64_t read = 0, bytes = 0, name = 0, snp = 0, meth = 0, cig = 0, bam4 = 0; };
  std::vector<At> at((size_t)nl + 1);
  for (int64_t li = 0; li < nl; ++li) {
    L& l = loci[(size_t)li];
    S->lf_off.push_back(S->flank.size()); S->lf_len.push_back((uint32_t)l.lf.size()); S->flank += l.lf;
    S->rf_off.push_back(S->flank.size()); S->rf_len.push_back((uint32_t)l.rf.size()); S->flank += l.rf;
    S->tr_off.push_back(S->tr.size()); S->tr_len.push_back((uint32_t)l.tr.size()); S->tr += l.tr;
    for (auto& m : l.motifs) { S->motifs += m; S->motif_off.push_back((uint32_t)S->motifs.size()); }
    S->set_begin.push_back((uint32_t)S->motif_off.size() - 1);
    S->ploidy.push_back((uint8_t)p->default_ploidy); S->genotyper.push_back((uint8_t)p->genotyper);
    S->contigs += l.contig; S->contig_off.push_back(S->contigs.size()); S->ids += l.id; S->id_off.push_back(S->ids.size());
    S->strucs += l.struc; S->struc_off.push_back(S->strucs.size());
    S->region_start.push_back(l.start); S->region_end.push_back(l.end);
    S->n_filtered.push_back(from_device ? dev.out.n_filt[li] : l.n_filt); S->n_seen.push_back(from_device ? dev.out.n_seen[li] : l.n_seen);
    At n = at[(size_t)li];
    for (auto& r : l.reads) {
      ++n.read; n.bytes += r.bases.size(); n.name += r.id.size(); n.snp += r.mismatch_offsets.size(); n.meth += r.has_meth ? r.meth.size() : 0;
      n.cig += r.cigar.size(); n.bam4 += (r.bases.size() + 1) / 2;
    }
    at[(size_t)li + 1] = n;
    S->lrb.push_back(from_device ? dev.out.lrb[li + 1] : n.read);
  }
  const At tot = at[(size_t)nl];
  const size_t nr = (size_t)tot.read;
  S->read_off.resize(nr); S->read_len.resize(nr); S->reads.resize((size_t)tot.bytes); S->quals.resize((size_t)tot.bytes);
  S->names.resize((size_t)tot.name); S->name_off.assign(nr + 1, 0); S->rq.resize(nr); S->is_reverse.resize(nr); S->mapq.resize(nr); S->hp.resize(nr);
  S->start_offset.resize(nr); S->end_offset.resize(nr); S->snp.resize((size_t)tot.snp); S->snp_off.assign(nr + 1, 0);
  S->meth.resize((size_t)tot.meth); S->moff.assign(nr + 1, 0); S->has_meth.resize(nr);
  S->cig.resize((size_t)tot.cig); S->cig_off.assign(nr + 1, 0); S->cig_ref_pos.resize(nr);
  if (p->keep_bam4) { S->bam4.assign((size_t)tot.bam4 + 1, 0); S->bam4_off.assign(nr + 1, 0); }  // the reads once more, two bases per byte (half the bytes to move to the GPU)
  std::atomic<int64_t> next_fill{0};
  std::atomic<int> fill_failed{0};
  auto fill = [&]() {
    for (;;) {
      const int64_t l0 = next_fill.fetch_add(16);
      if (l0 >= nl) break;
      for (int64_t li = l0; li < std::min(nl, l0 + 16); ++li) {
        At n = at[(size_t)li];
        const size_t r0 = (size_t)n.read;
        for (auto& r : loci[(size_t)li].reads) {
          const size_t k = (size_t)n.read;
          S->read_off[k] = n.bytes; S->read_len[k] = (uint32_t)r.bases.size();
          std::memcpy(&S->reads[0] + n.bytes, r.bases.data(), r.bases.size());
          std::memcpy(&S->quals[0] + n.bytes, r.quals.data(), std::min(r.quals.size(), r.bases.size()));
          std::memcpy(&S->names[0] + n.name, r.id.data(), r.id.size());
          S->rq[k] = r.rq; S->is_reverse[k] = r.is_reverse ? 1 : 0; S->mapq[k] = r.mapq; S->hp[k] = (int16_t)r.hp;
          S->start_offset[k] = r.start_offset; S->end_offset[k] = r.end_offset;
          std::copy(r.mismatch_offsets.begin(), r.mismatch_offsets.end(), S->snp.begin() + (ptrdiff_t)n.snp);
          S->has_meth[k] = r.has_meth ? 1 : 0;
          if (r.has_meth) std::copy(r.meth.begin(), r.meth.end(), S->meth.begin() + (ptrdiff_t)n.meth);
          std::copy(r.cigar.begin(), r.cigar.end(), S->cig.begin() + (ptrdiff_t)n.cig); S->cig_ref_pos[k] = r.ref_pos;
          ++n.read; n.bytes += r.bases.size(); n.name += r.id.size(); n.snp += r.mismatch_offsets.size(); n.meth += r.has_meth ? r.meth.size() : 0; n.cig += r.cigar.size();
          S->name_off[k + 1] = n.name; S->snp_off[k + 1] = n.snp; S->moff[k + 1] = n.meth; S->cig_off[k + 1] = n.cig;
        }
        if (p->keep_bam4 && n.read > r0) {
          const uint64_t base = at[(size_t)li].bam4;
          const int64_t got = trgt_reads_pack_bam4((const uint8_t*)S->reads.data(), (int64_t)(n.read - r0), S->read_off.data() + r0, S->read_len.data() + r0, S->bam4.data() + base, S->bam4_off.data() + r0);
          if (got != (int64_t)(at[(size_t)li + 1].bam4 - base)) fill_failed = 1;
          for (size_t k = r0; k < (size_t)n.read; ++k) S->bam4_off[k] += base;
        }
        std::vector<Read>().swap(loci[(size_t)li].reads);  // (freed by the worker, not by the caller's thread at the end)
      }
    }
  };
  {
    auto guarded = [&]() { try { fill(); } catch (const std::exception&) { fill_failed = 1; } };
    if (from_device) {}
    else if (nthr <= 1) guarded();
    else { std::vector<std::thread> th; for (int t = 0; t < nthr; ++t) th.emplace_back(guarded); for (auto& t : th) t.join(); }
  }
  if (fill_failed) return bad("trgt_ingest: assembling the batch failed");
  trgt_ingest_batch& B = S->pub;
  std::memset(&B, 0, sizeof B);
  B.read_blob_device = -1;
  auto u8 = [](const std::string& s) { return (const uint8_t*)s.data(); };
  if (S->flank.empty()) S->flank.push_back('\0');
  if (S->tr.empty()) S->tr.push_back('\0');
  if (S->reads.empty()) S->reads.push_back('\0');
  B.n_loci = nl; B.n_reads = (int64_t)S->read_off.size(); B.n_motifs = (int64_t)S->motif_off.size() - 1;
  B.flank_bytes = S->flank.size(); B.tr_bytes = S->tr.size(); B.motif_bytes = S->motifs.size(); B.read_bytes = S->reads.size();
  B.flank_blob = u8(S->flank); B.lf_off = S->lf_off.data(); B.lf_len = S->lf_len.data(); B.rf_off = S->rf_off.data(); B.rf_len = S->rf_len.data();
  B.tr_blob = u8(S->tr); B.tr_off = S->tr_off.data(); B.tr_len = S->tr_len.data();
  B.motif_blob = u8(S->motifs); B.motif_off = S->motif_off.data(); B.set_motif_begin = S->set_begin.data();
  B.ploidy = S->ploidy.data(); B.genotyper = S->genotyper.data(); B.locus_read_begin = S->lrb.data();
  B.read_blob = u8(S->reads); B.read_off = S->read_off.data(); B.read_len = S->read_len.data(); B.read_qual = S->rq.data();
  B.qual_blob = u8(S->quals); B.name_blob = S->names.data(); B.name_off = S->name_off.data();
  B.is_reverse = S->is_reverse.data(); B.mapq = S->mapq.data(); B.hp_tag = S->hp.data(); B.has_meth = S->has_meth.data();
  B.start_offset = S->start_offset.data(); B.end_offset = S->end_offset.data();
  B.mismatch_offsets = S->snp.data(); B.mismatch_off = S->snp_off.data();
  B.meth = S->meth.data(); B.meth_off = S->moff.data();
  B.n_quality_filtered = S->n_filtered.data(); B.n_reads_seen = S->n_seen.data();
  B.contig_blob = S->contigs.data(); B.contig_off = S->contig_off.data(); B.id_blob = S->ids.data(); B.id_off = S->id_off.data();
  B.struc_blob = S->strucs.data(); B.struc_off = S->struc_off.data(); B.region_start = S->region_start.data(); B.region_end = S->region_end.data();
  B.cigar = S->cig.data(); B.cigar_off = S->cig_off.data(); B.cigar_ref_pos = S->cig_ref_pos.data();
  S->skipped_off.push_back(0);
  for (auto& m : skipped) { S->skipped += m; S->skipped_off.push_back(S->skipped.size()); }
  B.n_skipped = (int64_t)skipped.size(); B.skipped_blob = S->skipped.data(); B.skipped_off = S->skipped_off.data();
  if (p->keep_bam4) { B.read_bam4 = S->bam4.data(); B.read_bam4_off = S->bam4_off.data(); B.read_bam4_bytes = tot.bam4; }
  if (from_device) {  // the per-read arrays are pieces of the slab the kernels filled (pinned mirror); the ASCII reads stay in HBM as well
    const trgt::ingd::HostOut& D = dev.out;
    S->slab = dev.slab; dev.slab = trgt::ingd::Slab(); S->slab_pool = h->slab_pool;
    B.n_reads = D.n_reads; B.read_bytes = std::max<uint64_t>(1, D.read_bytes);
    B.read_blob = D.reads; B.read_off = D.read_off; B.read_len = D.read_len; B.read_qual = D.rq;
    B.qual_blob = D.quals; B.name_blob = D.names; B.name_off = D.name_off;
    B.is_reverse = D.is_reverse; B.mapq = D.mapq; B.hp_tag = D.hp; B.has_meth = D.has_meth;
    B.start_offset = D.start_offset; B.end_offset = D.end_offset;
    B.mismatch_offsets = D.snp; B.mismatch_off = D.snp_off; B.meth = D.meth; B.meth_off = D.meth_off;
    B.cigar = D.cig; B.cigar_off = D.cig_off; B.cigar_ref_pos = D.cig_ref_pos;
    if (p->keep_bam4) { B.read_bam4 = D.bam4; B.read_bam4_off = D.bam4_off; B.read_bam4_bytes = D.bam4_bytes; }
    B.read_blob_dev = D.dev_reads; B.read_blob_device = p->ingest_device;
  }
  if (trace) std::fprintf(stderr, "[ingest] %lld loci, %d threads (runs of %lld): catalog+genome %.1f ms, reads %.1f ms (%s; %llu blocks inflated by the workers, %llu found in their caches), arrays %.1f ms\n", (long long)nl, nthr, (long long)run, t1 - t0, t2 - t1, from_device ? "on the device" : "host workers", (unsigned long long)n_inflated.load(), (unsigned long long)n_cache_hits.load(), now() - t2);
  B.owner = S.release();
  *out = &reinterpret_cast<BatchStore*>(B.owner)->pub;
  return TRGT_OK;
}

// exceptions (std::bad_alloc on a corrupt size field, ...) never cross the C ABI
int trgt_ingest_open(const char* bam_path, const char* fasta_path, trgt_ingest** out) {
  try { return ingest_open_impl(bam_path, fasta_path, out); }
  catch (const std::exception& e) {
    if (out) { trgt_ingest* h = new (std::nothrow) trgt_ingest(); if (h) h->err = std::string("trgt_ingest_open: ") + e.what(); *out = h; }
    return TRGT_ERR_NOMEM;
  }
}
int trgt_ingest_batch_from_catalog(trgt_ingest* h, const trgt_ingest_params* p, const char* bed_path, int64_t first_locus, int64_t max_loci,
                                   trgt_ingest_batch** out) {
  try { return ingest_batch_impl(h, p, bed_path, first_locus, max_loci, out); }
  catch (const std::exception& e) { if (h) h->err = std::string("trgt_ingest_batch_from_catalog: ") + e.what(); return TRGT_ERR_NOMEM; }
}

// raw DEFLATE of one block by the library's own decoder (mode 0; 1 = inflated, 0 = declined: ingestion then runs zlib) or by zlib (mode 1)
int32_t trgt_inflate_raw(const uint8_t* in, int64_t n_in, uint8_t* out, int64_t n_out, int32_t mode) {
  if (n_in < 0 || n_out < 0 || (n_in > 0 && !in) || (n_out > 0 && !out)) return TRGT_ERR_INVALID;
  try {
    if (mode == 0) { std::unique_ptr<trgt::inflate_fast::Tables> t(new trgt::inflate_fast::Tables()); return trgt::inflate_fast::inflate_block(in, (size_t)n_in, out, (size_t)n_out, *t) ? 1 : 0; }
    z_stream zs; std::memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return TRGT_ERR_NOMEM;
    zs.next_in = const_cast<uint8_t*>(in); zs.avail_in = (uInt)n_in; zs.next_out = out; zs.avail_out = (uInt)n_out;
    const int rc = inflate(&zs, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && zs.avail_out == 0;
    inflateEnd(&zs);
    return ok ? 1 : 0;
  } catch (const std::exception&) { return TRGT_ERR_NOMEM; }
}

// ---- the per-read helpers on their own (include/trgt_hip.h: "per-read helpers"): thin wrappers over the functions above
int64_t trgt_cigar_ref_len(uint32_t op) { return ref_len(op); }
int64_t trgt_cigar_query_len(uint32_t op) { return qry_len(op); }
int64_t trgt_cigar_total_query_len(const uint32_t* cigar, int64_t n_ops) {
  if (n_ops < 0 || (n_ops > 0 && !cigar)) return TRGT_ERR_INVALID;
  int64_t t = 0; for (int64_t i = 0; i < n_ops; ++i) t += qry_len(cigar[i]); return t;
}
int64_t trgt_read_mismatch_offsets(const uint32_t* cigar, int64_t n_ops, int64_t ref_pos, int64_t region_start, int64_t region_end, int32_t* out, int64_t cap) {
  if (n_ops < 0 || (n_ops > 0 && !cigar) || cap < 0 || (cap > 0 && !out)) return TRGT_ERR_INVALID;
  try {
    std::vector<int32_t> v;
    snps_offset(cigar, (size_t)n_ops, ref_pos, region_start, region_end, v);
    if ((int64_t)v.size() > cap) return TRGT_ERR_INVALID;
    std::copy(v.begin(), v.end(), out);
    return (int64_t)v.size();
  } catch (const std::exception&) { return TRGT_ERR_NOMEM; }
}
int64_t trgt_read_meth(const uint8_t* bases, int64_t n_bases, const char* mm, const uint8_t* ml, int64_t n_ml, int32_t is_reverse, uint8_t* out, int64_t cap) {
  if (n_bases < 0 || (n_bases > 0 && !bases) || !mm || n_ml < 0 || (n_ml > 0 && !ml) || cap < 0 || (cap > 0 && !out)) return TRGT_ERR_INVALID;
  try {
    const std::string b((const char*)bases, (size_t)n_bases);
    std::vector<std::pair<uint32_t, uint8_t>> mods; std::vector<uint8_t> meth;
    if (!basemods_5mc_tags(mm, ml, (uint32_t)n_ml, b, is_reverse != 0, mods) || !meth_per_cpg(b, is_reverse != 0, mods, meth)) return -1;
    if ((int64_t)meth.size() > cap) return TRGT_ERR_INVALID;
    std::copy(meth.begin(), meth.end(), out);
    return (int64_t)meth.size();
  } catch (const std::exception&) { return TRGT_ERR_NOMEM; }
}
int64_t trgt_read_clip_to_region(const uint8_t* bases, const uint8_t* quals, int64_t n_bases, const uint8_t* meth, int64_t n_meth, const uint32_t* cigar,
                                 int64_t n_ops, int64_t ref_pos, int64_t region_start, int64_t region_end, uint8_t* out_bases, uint8_t* out_quals,
                                 uint8_t* out_meth, int64_t* out_meth_n, uint32_t* out_cigar, int64_t* out_n_ops, int64_t* out_ref_pos) {
  if (n_bases < 0 || (n_bases > 0 && (!bases || !quals || !out_bases || !out_quals)) || n_ops < 0 || (n_ops > 0 && !cigar) || !out_cigar || !out_n_ops ||
      !out_ref_pos || !out_meth_n || (n_meth > 0 && (!meth || !out_meth)))
    return TRGT_ERR_INVALID;
  try {
    Read in, c;
    in.bases.assign((const char*)bases, (size_t)n_bases); in.quals.assign(quals, quals + n_bases);
    in.has_meth = n_meth >= 0; if (n_meth > 0) in.meth.assign(meth, meth + n_meth);
    in.has_cigar = true; in.ref_pos = ref_pos; in.cigar.assign(cigar, cigar + n_ops);
    if (!clip_to_region(in, region_start, region_end, c)) return -1;
    std::copy(c.bases.begin(), c.bases.end(), out_bases); std::copy(c.quals.begin(), c.quals.end(), out_quals);
    *out_meth_n = c.has_meth ? (int64_t)c.meth.size() : -1;
    if (c.has_meth) std::copy(c.meth.begin(), c.meth.end(), out_meth);
    std::copy(c.cigar.begin(), c.cigar.end(), out_cigar); *out_n_ops = (int64_t)c.cigar.size(); *out_ref_pos = c.ref_pos;
    return (int64_t)c.bases.size();
  } catch (const std::exception&) { return TRGT_ERR_NOMEM; }
}

}  // extern "C"
