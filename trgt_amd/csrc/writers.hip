// trgt_amd/csrc/writers.hip -- the step behind the GPU path, host C++ (no device code):
//   VcfWriter::new / write / set_gt / encode_*            src/trgt/writers/write_vcf.rs:19-397
//   BamWriter::create_header / write                      src/trgt/writers/write_bam.rs:33-144
//   HiFiRead::clip_bases                                  src/trgt/reads/clip_bases.rs:9-120
//   get_meth / assign_read / get_tr_meth                  src/trgt/workflows/tr.rs:196-262, 363-398 (the AM field)
// Input: a batch of the native ingestion (trgt_ingest_batch: catalog fields, clipped reads with their HiFiRead fields) and the result
// arrays trgt_locus_batch filled for it.  The reference writes through htslib; here VCF lines are formatted directly (BGZF-compressed when
// the path ends in .gz -- the reference's bcf::Writer compresses) and BAM records are encoded and BGZF-compressed with zlib.
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "crc32_fast.hpp"
#include "../../include/trgt_hip.h"

extern "C" {
const char* trgt_ingest_header_text(const trgt_ingest* h);
int32_t trgt_ingest_n_contigs(const trgt_ingest* h);
const char* trgt_ingest_contig_name(const trgt_ingest* h, int32_t i);
uint32_t trgt_ingest_contig_length(const trgt_ingest* h, int32_t i);
}

namespace {

struct BgzfOut {  // a file, plain or as a series of BGZF blocks (cut every 0xFF00 bytes of payload, whatever the number of threads)
  FILE* f = nullptr; bool bgzf = false; std::vector<uint8_t> buf; int threads = 1; int level = 6;
  bool open(const char* path, bool compress) { f = std::fopen(path, "wb"); bgzf = compress; return f != nullptr; }
  static bool deflate_block(const uint8_t* d, size_t n, std::vector<uint8_t>& out, int level = 6) {
    out.resize(0x10000 + 64);
    z_stream zs; std::memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    zs.next_in = const_cast<uint8_t*>(d); zs.avail_in = (uInt)n; zs.next_out = out.data() + 18; zs.avail_out = (uInt)out.size() - 18 - 8;
    const int rc = deflate(&zs, Z_FINISH);
    deflateEnd(&zs);
    if (rc != Z_STREAM_END) return false;
    const size_t clen = zs.total_out, total = 18 + clen + 8;
    static const uint8_t head[16] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0};
    std::memcpy(out.data(), head, 16);
    out[16] = (uint8_t)((total - 1) & 0xFF); out[17] = (uint8_t)((total - 1) >> 8);
    const uint32_t crc = n ? trgt::crc32_fast(d, n) : 0u;
    for (int i = 0; i < 4; ++i) { out[18 + clen + i] = (uint8_t)(crc >> (8 * i)); out[22 + clen + i] = (uint8_t)((uint32_t)n >> (8 * i)); }
    out.resize(total);
    return true;
  }
  bool block(const uint8_t* d, size_t n) { std::vector<uint8_t> out; return deflate_block(d, n, out, level) && std::fwrite(out.data(), 1, out.size(), f) == out.size(); }
  // a BGZF block around a payload that is deflated already (trgt_deflate_blocks): header, payload, CRC-32 and size of the data
  static void frame_block(const uint8_t* d, size_t n, const uint8_t* payload, size_t clen, std::vector<uint8_t>& out) {
    const size_t total = 18 + clen + 8;
    out.resize(total);
    static const uint8_t head[16] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0};
    std::memcpy(out.data(), head, 16);
    out[16] = (uint8_t)((total - 1) & 0xFF); out[17] = (uint8_t)((total - 1) >> 8);
    std::memcpy(out.data() + 18, payload, clen);
    const uint32_t crc = n ? trgt::crc32_fast(d, n) : 0u;
    for (int i = 0; i < 4; ++i) { out[18 + clen + i] = (uint8_t)(crc >> (8 * i)); out[22 + clen + i] = (uint8_t)((uint32_t)n >> (8 * i)); }
  }
  trgt_hip_ctx* dev = nullptr;   // trgt_writer_params.deflate_device: the full blocks of a flush are deflated on this context's GPU
  std::string dev_err;           // a device deflate that FAILED fails the write (no silent host-only run, like ingest_device); this says why
  int64_t n_dev = 0, n_declined = 0, n_host = 0;  // blocks the device deflated / declined (zlib took them) / zlib deflated because the flush was too small or no device was named
  std::vector<uint8_t> dev_out; std::vector<uint64_t> dev_soff, dev_doff; std::vector<uint32_t> dev_slen, dev_cap, dev_len;
  // the full blocks of buf: deflated by `threads` workers (a block is independent of its neighbours) -- or on the GPU in one go, the
  // workers then only frame them (CRC-32) and deflate what the device declined -- written in order
  bool flush_full_blocks() {
    const size_t nb = buf.size() / 0xFF00;
    if (!nb) return true;
    std::vector<std::vector<uint8_t>> outs(nb);
    std::atomic<size_t> next{0}; std::atomic<int> failed{0};
    constexpr size_t DEV_SLOT = 0x10000, DEV_CAP = 0xFF00;  // room per block on the device side: payload + 26 must fit a BGZF block; 256 bytes of slack between the regions
    bool on_dev = false;
    if (dev && nb >= 16) {  // (a handful of blocks is not worth the round trip)
      dev_out.resize(nb * DEV_SLOT + 64); dev_soff.resize(nb); dev_doff.resize(nb); dev_slen.assign(nb, 0xFF00u); dev_cap.assign(nb, (uint32_t)DEV_CAP); dev_len.assign(nb, 0u);
      for (size_t k = 0; k < nb; ++k) { dev_soff[k] = k * 0xFF00; dev_doff[k] = k * DEV_SLOT; }
      const int rc = trgt_deflate_blocks(dev, (int64_t)nb, buf.data(), dev_soff.data(), dev_slen.data(), dev_out.data(), dev_doff.data(), dev_cap.data(), dev_len.data());
      if (rc != TRGT_OK) { dev_err = std::string("deflate_device: ") + trgt_hip_last_error(dev); return false; }
      on_dev = true;
      for (size_t k = 0; k < nb; ++k) { if (dev_len[k] > 0 && dev_len[k] + 26u <= 0x10000u) ++n_dev; else ++n_declined; }
    } else n_host += (int64_t)nb;
    auto work = [&]() {
      try {
        for (;;) {
          const size_t k = next.fetch_add(1);
          if (k >= nb) break;
          if (on_dev && dev_len[k] > 0 && dev_len[k] + 26u <= 0x10000u) frame_block(buf.data() + k * 0xFF00, 0xFF00, dev_out.data() + dev_doff[k], dev_len[k], outs[k]);
          else if (!deflate_block(buf.data() + k * 0xFF00, 0xFF00, outs[k], level)) failed = 1;
        }
      } catch (const std::exception&) { failed = 1; }
    };
    const int nt = (int)std::min<size_t>((size_t)std::max(1, threads), nb);
    if (nt <= 1) work();
    else { std::vector<std::thread> th; for (int t = 0; t < nt; ++t) th.emplace_back(work); for (auto& t : th) t.join(); }
    if (failed) return false;
    for (auto& o : outs) if (std::fwrite(o.data(), 1, o.size(), f) != o.size()) return false;
    buf.erase(buf.begin(), buf.begin() + (ptrdiff_t)(nb * 0xFF00));
    return true;
  }
  // append() + one flush() per batch: the flush then sees all the blocks of the batch at once (enough for the workers, or for the device)
  bool append(const void* d, size_t n) {
    if (!bgzf) return std::fwrite(d, 1, n, f) == n;
    const uint8_t* p = (const uint8_t*)d;
    buf.insert(buf.end(), p, p + n);
    return true;
  }
  bool flush() { return !bgzf || flush_full_blocks(); }
  bool write(const void* d, size_t n) { return append(d, n) && flush(); }
  bool close() {
    bool ok = true;
    if (f) {
      if (bgzf) {  // what is left: full blocks first (an append without its flush, after an error), then the short one, + the empty EOF block
        ok = flush_full_blocks();
        if (!buf.empty()) ok = block(buf.data(), buf.size()) && ok;
        buf.clear(); ok = block(nullptr, 0) && ok;
      }
      ok = std::fclose(f) == 0 && ok; f = nullptr;
    }
    return ok;
  }
};

inline void put32(std::vector<uint8_t>& v, uint32_t x) { for (int i = 0; i < 4; ++i) v.push_back((uint8_t)(x >> (8 * i))); }
inline void put16(std::vector<uint8_t>& v, uint32_t x) { v.push_back((uint8_t)x); v.push_back((uint8_t)(x >> 8)); }
inline bool ends_with(const std::string& s, const char* suf) { const size_t n = std::strlen(suf); return s.size() >= n && s.compare(s.size() - n, n, suf) == 0; }
inline uint32_t qlen_of(uint32_t op) { const uint32_t c = op & 0xF; return (c == 0 || c == 1 || c == 4 || c == 7 || c == 8) ? op >> 4 : 0; }
inline uint32_t rlen_of(uint32_t op) { const uint32_t c = op & 0xF; return (c == 0 || c == 2 || c == 3 || c == 7 || c == 8) ? op >> 4 : 0; }
// clip_cigar of HiFiRead::clip_bases (clip_bases.rs:59-118): the operations left after dropping left / right query bases; ref_pos moves
// past the reference bases the dropped prefix consumed.  false: the CIGAR covers fewer query bases than are clipped.
bool clip_bases_cigar(const uint32_t* cg, size_t nc, size_t left, size_t right, std::vector<uint32_t>& ops, int64_t& ref_pos) {
  size_t qsum = 0; for (size_t i = 0; i < nc; ++i) qsum += qlen_of(cg[i]);
  if (qsum < left + right) return false;
  size_t keep = qsum - left - right, left_len = left, i = 0;
  uint32_t cur = nc ? cg[0] : 0; bool have = nc > 0;
  while (left_len != 0 && have) {
    const size_t q = qlen_of(cur);
    if (q > left_len) { const uint32_t rest = (uint32_t)(q - left_len); if (rlen_of(cur)) ref_pos += (int64_t)left_len; cur = (rest << 4) | (cur & 0xF); left_len = 0; }
    else { left_len -= q; ref_pos += rlen_of(cur); ++i; have = i < nc; if (have) cur = cg[i]; }
  }
  while (have && keep != 0) {
    const size_t q = qlen_of(cur);
    if (q > keep) { ops.push_back(((uint32_t)keep << 4) | (cur & 0xF)); keep = 0; }
    else { keep -= q; ops.push_back(cur); ++i; have = i < nc; if (have) cur = cg[i]; }
  }
  return true;
}
// the per-CpG methylation values of the CpGs whose C stays inside [left, n - right) (clip_bases.rs:34-52)
void clip_bases_meth(const uint8_t* all, size_t n, const uint8_t* me, size_t nme, size_t left, size_t right, std::vector<uint8_t>& meth) {
  size_t ci = 0;
  for (size_t idx = 0; idx + 1 < n; ++idx) if (all[idx] == 'C' && all[idx + 1] == 'G') { if (ci < nme && left <= idx && idx < n - right) meth.push_back(me[ci]); ++ci; }
}

inline int reg2bin(int64_t beg, int64_t end) {
  --end;
  if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
  if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
  if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
  if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
  if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
  return 0;
}

}  // namespace

struct trgt_writer {
  std::string err, sample;
  BgzfOut vcf, bam;
  bool has_bam = false, keep_unmapped = false;
  int32_t flank_len = 50;
  int threads = 1;
  std::vector<std::string> contigs;
  trgt_hip_ctx* dev = nullptr;  // owned: the context of trgt_writer_params.deflate_device
  // write_behind (ABI 11): the formatted pieces of a batch are appended, deflated and written by a thread of the writer's own while the
  // caller formats the next batch; one batch in flight; what that thread reports is returned by the next write or by the close
  bool write_behind = false;
  std::thread bg; int bg_rc = TRGT_OK; std::string bg_err;
  int bg_wait() { if (bg.joinable()) bg.join(); if (bg_rc != TRGT_OK && !bg_err.empty()) { err = bg_err; bg_err.clear(); } const int rc = bg_rc; bg_rc = TRGT_OK; return rc; }
  ~trgt_writer() { if (bg.joinable()) bg.join(); if (dev) trgt_hip_destroy(dev); }
};

extern "C" {

const char* trgt_writer_last_error(const trgt_writer* w) { return w ? w->err.c_str() : "null handle"; }
// ABI 10: BGZF blocks of the spanning BAM so far -- out[0] deflated on the device, [1] declined by it (zlib took them), [2] deflated by zlib
// because no device was named or the flush held fewer than 16 full blocks
void trgt_writer_device_stats(const trgt_writer* w, int64_t out[3]) { if (w && out) { out[0] = w->bam.n_dev; out[1] = w->bam.n_declined; out[2] = w->bam.n_host; } }

void trgt_writer_default_params(trgt_writer_params* p) {
  if (!p) return;
  p->output_flank_len = 50; p->sample_name = "sample"; p->program = "trgt"; p->version = "3.0.0"; p->command_line = ""; p->keep_unmapped_flag = 1; p->threads = 0; p->bam_compress_level = 6; p->deflate_device = -1; p->write_behind = 0;
}

static int writer_open_impl(const trgt_ingest* src, const trgt_writer_params* p, const char* vcf_path, const char* bam_path, trgt_writer** out) {
  if (!src || !p || !vcf_path || !out) return TRGT_ERR_INVALID;
  std::unique_ptr<trgt_writer> w(new trgt_writer());
  *out = nullptr;
  auto bad = [&](const std::string& m) { w->err = m; *out = w.release(); return TRGT_ERR_INVALID; };
  w->flank_len = p->output_flank_len; w->keep_unmapped = p->keep_unmapped_flag != 0;
  w->threads = p->threads > 0 ? p->threads : (int)std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
  w->write_behind = p->write_behind != 0;
  w->vcf.threads = w->bam.threads = w->threads;
  w->bam.level = std::min(9, std::max(0, p->bam_compress_level));
  if (p->deflate_device >= 0 && bam_path) {  // the spanning BAM's blocks on the GPU (the VCF stays with zlib: it is small)
    if (trgt_hip_create(p->deflate_device, &w->dev) != TRGT_OK) { const std::string m = w->dev ? trgt_hip_last_error(w->dev) : "no such device"; if (w->dev) { trgt_hip_destroy(w->dev); w->dev = nullptr; } return bad("deflate_device: " + m); }
    w->bam.dev = w->dev;
  }
  const std::string prog = p->program ? p->program : "trgt", ver = p->version ? p->version : "", cl = p->command_line ? p->command_line : "";
  w->sample = p->sample_name ? p->sample_name : "sample";
  for (int32_t i = 0; i < trgt_ingest_n_contigs(src); ++i) w->contigs.push_back(trgt_ingest_contig_name(src, i));
  if (!w->vcf.open(vcf_path, ends_with(vcf_path, ".gz"))) return bad(std::string("Invalid VCF output path: ") + vcf_path);
  {  // VcfWriter::new (write_vcf.rs:49-92); the first two lines are what htslib puts into every new header
    std::string h = "##fileformat=VCFv4.2\n##FILTER=<ID=PASS,Description=\"All filters passed\">\n";
    h += "##INFO=<ID=TRID,Number=1,Type=String,Description=\"Tandem repeat ID\">\n"
         "##INFO=<ID=END,Number=1,Type=Integer,Description=\"End position of the variant described in this record\">\n"
         "##INFO=<ID=MOTIFS,Number=.,Type=String,Description=\"Motifs that the tandem repeat is composed of\">\n"
         "##INFO=<ID=STRUC,Number=1,Type=String,Description=\"Structure of the region\">\n"
         "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n"
         "##FORMAT=<ID=AL,Number=.,Type=Integer,Description=\"Length of each allele\">\n"
         "##FORMAT=<ID=ALLR,Number=.,Type=String,Description=\"Length range per allele\">\n"
         "##FORMAT=<ID=SD,Number=.,Type=Integer,Description=\"Number of spanning reads supporting per allele\">\n"
         "##FORMAT=<ID=MC,Number=.,Type=String,Description=\"Motif counts per allele\">\n"
         "##FORMAT=<ID=MS,Number=.,Type=String,Description=\"Motif spans per allele\">\n"
         "##FORMAT=<ID=AP,Number=.,Type=Float,Description=\"Allele purity per allele\">\n"
         "##FORMAT=<ID=AM,Number=.,Type=Float,Description=\"Mean methylation level per allele\">\n";
    for (int32_t i = 0; i < trgt_ingest_n_contigs(src); ++i)
      h += "##contig=<ID=" + w->contigs[(size_t)i] + ",length=" + std::to_string(trgt_ingest_contig_length(src, i)) + ">\n";
    h += "##" + prog + "Version=" + ver + "\n##" + prog + "Command=" + cl + "\n";
    h += "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + w->sample + "\n";
    if (!w->vcf.write(h.data(), h.size())) return bad("cannot write the VCF header");
  }
  if (bam_path) {
    if (!w->bam.open(bam_path, true)) return bad(std::string("cannot open ") + bam_path);
    w->has_bam = true;
    std::string text = trgt_ingest_header_text(src);  // BamWriter::create_header (write_bam.rs:52-65): the template + a @PG record
    if (!text.empty() && text.back() != '\n') text += "\n";
    text += "@PG\tID:" + prog + "\tPN:" + prog + "\tCL:" + cl + "\tVN:" + ver + "\n";
    std::vector<uint8_t> h = {'B', 'A', 'M', 1};
    put32(h, (uint32_t)text.size()); h.insert(h.end(), text.begin(), text.end());
    put32(h, (uint32_t)w->contigs.size());
    for (size_t i = 0; i < w->contigs.size(); ++i) {
      put32(h, (uint32_t)w->contigs[i].size() + 1); h.insert(h.end(), w->contigs[i].begin(), w->contigs[i].end()); h.push_back(0);
      put32(h, trgt_ingest_contig_length(src, (int32_t)i));
    }
    if (!w->bam.write(h.data(), h.size())) return bad("cannot write the BAM header");
  }
  *out = w.release();
  return TRGT_OK;
}

static int writer_write_impl(trgt_writer* w, const trgt_ingest_batch* b, const trgt_locus_batch_out* o) {
  if (!w || !b || !o) return TRGT_ERR_INVALID;
  auto bad = [&](const std::string& m) { w->err = m; return TRGT_ERR_INVALID; };
  if (!o->n_alleles || !o->allele_blob || !o->allele_off || !o->allele_len || !o->ci || !o->num_spanning || !o->classification || !o->read_rank ||
      !o->span_start || !o->span_end || !o->spans3 || !o->span_off || !o->n_spans || !o->motif_counts || !o->count_off || !o->purity)
    return bad("trgt_writer_write: incomplete result arrays");
  // One locus: its VCF line and its spanning-BAM records, appended to the caller's buffers (loci are independent: a batch is formatted
  // by w->threads workers over contiguous ranges of loci, and the pieces are written in locus order).
  auto format_locus = [&](int64_t l, std::string& lines, std::vector<uint8_t>& recs, std::vector<uint8_t>& rec, std::string& err) -> bool {
    auto bad = [&](const std::string& m) { err = m; return false; };
    std::string line;
    char num[64];
    const std::string contig(b->contig_blob + b->contig_off[l], b->contig_off[l + 1] - b->contig_off[l]);
    const std::string id(b->id_blob + b->id_off[l], b->id_off[l + 1] - b->id_off[l]);
    const std::string struc(b->struc_blob + b->struc_off[l], b->struc_off[l + 1] - b->struc_off[l]);
    const std::string tr((const char*)b->tr_blob + b->tr_off[l], b->tr_len[l]);
    if (b->lf_len[l] == 0) return bad("Empty flanks are not allowed");
    const char pad = (char)b->flank_blob[b->lf_off[l] + b->lf_len[l] - 1];
    const int n_al = o->n_alleles[l];
    const uint32_t m0 = b->set_motif_begin[l], m1 = b->set_motif_begin[l + 1];
    // the kept spanning reads in LocusResult.reads order
    const uint64_t r0 = b->locus_read_begin[l], r1 = b->locus_read_begin[l + 1];
    std::vector<uint64_t> kept;
    for (uint64_t r = r0; r < r1; ++r) if (o->read_rank[r] >= 0) { if ((size_t)o->read_rank[r] >= kept.size()) kept.resize((size_t)o->read_rank[r] + 1, r0); kept[(size_t)o->read_rank[r]] = r; }
    // ---- VCF record (write_vcf.rs:95-260)
    line.clear();
    line += contig; line += '\t'; line += std::to_string(b->region_start[l] > 0 ? b->region_start[l] : 1); line += "\t.\t";  // POS = saturating_sub(start, 1) + 1
    std::string info = "TRID=" + id + ";END=" + std::to_string(b->region_end[l]) + ";MOTIFS=";
    for (uint32_t m = m0; m < m1; ++m) { if (m > m0) info += ","; info.append((const char*)b->motif_blob + b->motif_off[m], b->motif_off[m + 1] - b->motif_off[m]); }
    info += ";STRUC=" + struc;
    if (n_al == 0) {  // add_missing_allele_info
      line += pad; line += tr; line += "\t.\t.\t.\t"; line += info; line += "\tGT:AL:ALLR:SD:MC:MS:AP:AM\t.:.:.:.:.:.:.:.\n";
    } else {
      std::string al[2];
      for (int a = 0; a < n_al; ++a) al[a].assign((const char*)o->allele_blob + o->allele_off[2 * l + a], o->allele_len[2 * l + a]);
      std::vector<const std::string*> seqs{&tr};  // set_gt (:219-260)
      std::string gt;
      for (int a = 0; a < n_al; ++a) {
        int idx;
        if (al[a] == tr) idx = 0;
        else if (seqs.size() == 1) { idx = 1; seqs.push_back(&al[a]); }
        else if (al[0] == al[1]) idx = 1;
        else { idx = 2; seqs.push_back(&al[a]); }
        if (a) gt += "/";
        gt += std::to_string(idx);
      }
      line += pad; line += *seqs[0]; line += '\t';
      if (seqs.size() == 1) line += ".";
      for (size_t s = 1; s < seqs.size(); ++s) { if (s > 1) line += ","; line += pad; line += *seqs[s]; }
      line += "\t.\t.\t"; line += info; line += "\tGT:AL:ALLR:SD:MC:MS:AP:AM\t"; line += gt;
      std::string f_al, f_allr, f_sd, f_mc, f_ms, f_ap, f_am;
      // get_meth (tr.rs:196-229) in the genotype's own order (before "reference allele first")
      const bool flipped = o->flipped && o->flipped[l] && n_al == 2;
      double meth_sum[2] = {0, 0}; size_t meth_n[2] = {0, 0};
      if (b->has_meth && b->meth && b->meth_off) {
        auto pre = [&](int a) { return flipped ? 1 - a : a; };  // allele of the original order -> output slot
        const size_t sz[2] = {(size_t)(o->gt_size ? o->gt_size[2 * l + pre(0)] : (int32_t)o->allele_len[2 * l + pre(0)]),
                              n_al == 2 ? (size_t)(o->gt_size ? o->gt_size[2 * l + pre(1)] : (int32_t)o->allele_len[2 * l + pre(1)]) : 0};
        for (uint64_t r : kept) {
          if (!b->has_meth[r] || b->meth_off[r + 1] == b->meth_off[r]) continue;
          const size_t s0 = (size_t)o->span_start[r], s1 = (size_t)o->span_end[r];
          const uint8_t* bases = b->read_blob + b->read_off[r]; const size_t n = b->read_len[r];
          const uint8_t* me = b->meth + b->meth_off[r]; const size_t nme = (size_t)(b->meth_off[r + 1] - b->meth_off[r]);
          double total = 0.0; size_t cpg = 0, ci = 0; bool malformed = false;
          for (size_t pos = 0; pos + 1 < n; ++pos)
            if (bases[pos] == 'C' && bases[pos + 1] == 'G') {
              if (s0 <= pos && pos < s1) { if (ci >= nme) { malformed = true; break; } ++cpg; total += (double)me[ci] / 255.0; }
              ++ci;
            }
          if (malformed) return bad("Read " + std::string(b->name_blob + b->name_off[r], b->name_off[r + 1] - b->name_off[r]) + " has malformed methylation profile");
          if (!cpg) continue;
          const double level = total / (double)cpg;
          const size_t len = s1 - s0;
          if (n_al == 1) { meth_sum[0] += level; ++meth_n[0]; continue; }  // assign_read (:238-262)
          const auto adiff = [](size_t x, size_t y) { return x > y ? x - y : y - x; };
          const bool sp1 = (size_t)o->ci[4 * l + 2 * pre(0)] <= len && len <= (size_t)o->ci[4 * l + 2 * pre(0) + 1];
          const bool sp2 = (size_t)o->ci[4 * l + 2 * pre(1)] <= len && len <= (size_t)o->ci[4 * l + 2 * pre(1) + 1];
          const size_t d1 = adiff(len, sz[0]), d2 = adiff(len, sz[1]);
          if (d1 < d2 && sp1) { meth_sum[0] += level; ++meth_n[0]; }
          else if (d2 < d1 && sp2) { meth_sum[1] += level; ++meth_n[1]; }
          else if (sz[0] == sz[1] && sp1) { meth_sum[0] += level; ++meth_n[0]; meth_sum[1] += level; ++meth_n[1]; }
        }
      }
      for (int a = 0; a < n_al; ++a) {
        const char* sep = a ? "," : "";
        f_al += sep; f_al += std::to_string(o->allele_len[2 * l + a]);
        f_allr += sep; f_allr += std::to_string(o->ci[4 * l + 2 * a]) + "-" + std::to_string(o->ci[4 * l + 2 * a + 1]);
        f_sd += sep; f_sd += std::to_string(o->num_spanning[2 * l + a]);
        f_mc += sep;
        for (uint32_t m = 0; m < m1 - m0; ++m) { if (m) f_mc += "_"; f_mc += std::to_string(o->motif_counts[o->count_off[2 * l + a] + m]); }
        f_ms += sep;
        const uint32_t ns = o->n_spans[2 * l + a];
        if (ns == 0) f_ms += ".";
        for (uint32_t k = 0; k < ns; ++k) {
          const int32_t* sp = o->spans3 + 3 * (o->span_off[2 * l + a] + k);
          if (k) f_ms += "_";
          f_ms += std::to_string(sp[0]) + "(" + std::to_string(sp[1]) + "-" + std::to_string(sp[2]) + ")";
        }
        f_ap += sep;
        if (std::isnan(o->purity[2 * l + a])) f_ap += "."; else { std::snprintf(num, sizeof num, "%.6f", o->purity[2 * l + a]); f_ap += num; }
        f_am += sep;
        const int src = flipped ? 1 - a : a;  // the allele's slot in the original order
        if (meth_n[src]) { std::snprintf(num, sizeof num, "%.2f", meth_sum[src] / (double)meth_n[src]); f_am += num; } else f_am += ".";
      }
      line += ":" + f_al + ":" + f_allr + ":" + f_sd + ":" + f_mc + ":" + f_ms + ":" + f_ap + ":" + f_am + "\n";
    }
    lines += line;
    // ---- spanning reads (write_bam.rs:72-144)
    if (!w->has_bam) return true;
    int tid = -1;
    for (size_t i = 0; i < w->contigs.size(); ++i) if (w->contigs[i] == contig) { tid = (int)i; break; }
    if (tid < 0) return bad("contig " + contig + " is not in the BAM header");
    const size_t F = (size_t)w->flank_len;
    for (uint64_t r : kept) {
      const size_t s0 = (size_t)o->span_start[r], s1 = (size_t)o->span_end[r], n = b->read_len[r];
      if (s0 < F || n < s1 + F) continue;  // "unexpectedly short flanks"
      const size_t left = s0 - F, right = n - s1 - F;
      if (left + right >= n) continue;     // clip_bases: None
      const uint8_t* bases = b->read_blob + b->read_off[r] + left; const uint8_t* quals = b->qual_blob + b->read_off[r] + left;
      const size_t len = n - left - right;
      std::vector<uint32_t> ops; int64_t ref_pos = b->cigar_ref_pos[r];
      if (!clip_bases_cigar(b->cigar + b->cigar_off[r], (size_t)(b->cigar_off[r + 1] - b->cigar_off[r]), left, right, ops, ref_pos)) return bad("CIGAR shorter than the read");
      if (ops.size() > 65535) return bad("more than 65535 CIGAR operations in a clipped read (the BAM record would need a CG tag)");
      std::vector<uint8_t> meth; bool has_meth = false;
      if (b->has_meth && b->has_meth[r]) {
        has_meth = true;
        clip_bases_meth(b->read_blob + b->read_off[r], n, b->meth + b->meth_off[r], (size_t)(b->meth_off[r + 1] - b->meth_off[r]), left, right, meth);
      }
      const std::string name(b->name_blob + b->name_off[r], b->name_off[r + 1] - b->name_off[r]);
      int64_t ref_end = ref_pos; for (uint32_t op : ops) ref_end += rlen_of(op);
      rec.clear();
      put32(rec, 0);  // block_size, patched below
      put32(rec, (uint32_t)tid); put32(rec, (uint32_t)ref_pos);
      rec.push_back((uint8_t)(name.size() + 1)); rec.push_back(b->mapq[r]);
      put16(rec, (uint32_t)reg2bin(ref_pos, ref_end > ref_pos ? ref_end : ref_pos + 1)); put16(rec, (uint32_t)ops.size());
      put16(rec, (b->is_reverse[r] ? 0x10u : 0u) | (w->keep_unmapped ? 0x4u : 0u)); put32(rec, (uint32_t)len);
      put32(rec, 0xFFFFFFFFu); put32(rec, 0xFFFFFFFFu); put32(rec, 0);  // mate: none
      rec.insert(rec.end(), name.begin(), name.end()); rec.push_back(0);
      for (uint32_t op : ops) put32(rec, op);
      { auto nib = [](uint8_t c) -> uint8_t { switch (c) { case '=': return 0; case 'A': return 1; case 'C': return 2; case 'M': return 3; case 'G': return 4; case 'R': return 5; case 'S': return 6; case 'V': return 7;
                                                           case 'T': return 8; case 'W': return 9; case 'Y': return 10; case 'H': return 11; case 'K': return 12; case 'D': return 13; case 'B': return 14; default: return 15; } };
        for (size_t i = 0; i < len; i += 2) rec.push_back((uint8_t)((nib(bases[i]) << 4) | (i + 1 < len ? nib(bases[i + 1]) : 0))); }
      rec.insert(rec.end(), quals, quals + len);
      auto tag = [&](const char* t, char ty) { rec.push_back((uint8_t)t[0]); rec.push_back((uint8_t)t[1]); rec.push_back((uint8_t)ty); };
      tag("TR", 'Z'); rec.insert(rec.end(), id.begin(), id.end()); rec.push_back(0);
      { tag("rq", 'f'); const float f = std::isnan(b->read_qual[r]) ? -1.0f : (float)b->read_qual[r]; uint32_t u; std::memcpy(&u, &f, 4); put32(rec, u); }
      if (has_meth) { tag("MC", 'B'); rec.push_back('C'); put32(rec, (uint32_t)meth.size()); rec.insert(rec.end(), meth.begin(), meth.end()); }
      { tag("MO", 'B'); rec.push_back('i'); const uint64_t a0 = b->mismatch_off[r], a1 = b->mismatch_off[r + 1]; put32(rec, (uint32_t)(a1 - a0)); for (uint64_t k = a0; k < a1; ++k) put32(rec, (uint32_t)b->mismatch_offsets[k]); }
      if (b->hp_tag[r] >= 0) { tag("HP", 'C'); rec.push_back((uint8_t)b->hp_tag[r]); }
      tag("SO", 'i'); put32(rec, (uint32_t)b->start_offset[r]);
      tag("EO", 'i'); put32(rec, (uint32_t)b->end_offset[r]);
      tag("AL", 'i'); put32(rec, (uint32_t)o->classification[r]);
      tag("FL", 'B'); rec.push_back('I'); put32(rec, 2); put32(rec, (uint32_t)F); put32(rec, (uint32_t)F);
      const uint32_t bs = (uint32_t)rec.size() - 4;
      for (int i = 0; i < 4; ++i) rec[(size_t)i] = (uint8_t)(bs >> (8 * i));
      recs.insert(recs.end(), rec.begin(), rec.end());
    }
    return true;
  };
  const int64_t nl = b->n_loci;
  static const bool trace = std::getenv("TRGT_WRITER_TRACE") != nullptr;  // phase times on stderr
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(w->threads, nl / 16));
  std::vector<std::string> lines((size_t)nt), errs((size_t)nt);
  std::vector<std::vector<uint8_t>> recs((size_t)nt);
  auto work = [&](int t) {
    try {
      std::vector<uint8_t> rec;
      for (int64_t l = nl * t / nt; l < nl * (t + 1) / nt; ++l) if (!format_locus(l, lines[(size_t)t], recs[(size_t)t], rec, errs[(size_t)t])) return;
    } catch (const std::exception& e) { errs[(size_t)t] = std::string("trgt_writer_write: ") + e.what(); }
  };
  if (nt <= 1) work(0);
  else { std::vector<std::thread> th; for (int t = 0; t < nt; ++t) th.emplace_back(work, t); for (auto& t : th) t.join(); }
  const double t1 = now();
  size_t vcf_bytes = 0, bam_bytes = 0;
  for (int t = 0; t < nt; ++t) { vcf_bytes += lines[(size_t)t].size(); bam_bytes += recs[(size_t)t].size(); }
  struct Tr { bool on; double t0, t1; int64_t nl; int nt; size_t v, b; decltype(now)& now; ~Tr() { if (on) std::fprintf(stderr, "[writer] %lld loci, %d threads: formatting %.1f ms (%zu B of VCF, %zu B of BAM records), deflate + write %.1f ms\n", (long long)nl, nt, t1 - t0, v, b, now() - t1); } } tr{trace, t0, t1, nl, nt, vcf_bytes, bam_bytes, now};
  // what precedes the first failing locus is written, as a serial writer would have; msg: what went wrong (empty: nothing)
  auto flush_pieces = [](trgt_writer* ww, std::vector<std::string>& ln, std::vector<std::vector<uint8_t>>& rc, std::vector<std::string>& er, std::string& msg) -> int {
    for (size_t t = 0; t < ln.size(); ++t) {
      if (!ln[t].empty() && !ww->vcf.append(ln[t].data(), ln[t].size())) { msg = "cannot write the VCF"; return TRGT_ERR_INVALID; }
      if (!rc[t].empty() && !ww->bam.append(rc[t].data(), rc[t].size())) { msg = "cannot write the BAM"; return TRGT_ERR_INVALID; }
      if (!er[t].empty()) { ww->vcf.flush(); ww->bam.flush(); msg = er[t]; return TRGT_ERR_INVALID; }
    }
    if (!ww->vcf.flush()) { msg = "cannot write the VCF"; return TRGT_ERR_INVALID; }
    if (!ww->bam.flush()) { msg = ww->bam.dev_err.empty() ? "cannot write the BAM" : ww->bam.dev_err; return TRGT_ERR_INVALID; }
    return TRGT_OK;
  };
  if (w->write_behind) {
    // the batch before this one must be on its way out (one in flight): its verdict is this call's when it failed
    if (const int prc = w->bg_wait()) return prc;
    struct Pieces { std::vector<std::string> lines, errs; std::vector<std::vector<uint8_t>> recs; };
    auto pcs = std::make_shared<Pieces>();
    pcs->lines = std::move(lines); pcs->errs = std::move(errs); pcs->recs = std::move(recs);
    tr.on = false;  // (the phase line would time the hand-over, not the flush)
    w->bg = std::thread([w, pcs, flush_pieces]() {
      try { std::string msg; w->bg_rc = flush_pieces(w, pcs->lines, pcs->recs, pcs->errs, msg); if (w->bg_rc != TRGT_OK) w->bg_err = msg; }
      catch (const std::exception& e) { w->bg_rc = TRGT_ERR_NOMEM; w->bg_err = std::string("trgt_writer_write: ") + e.what(); }
    });
    return TRGT_OK;
  }
  std::string msg;
  if (const int frc = flush_pieces(w, lines, recs, errs, msg)) return bad(msg);
  return TRGT_OK;
}

// HiFiRead::clip_bases on its own (include/trgt_hip.h: "per-read helpers")
int64_t trgt_read_clip_bases(const uint8_t* bases, const uint8_t* quals, int64_t n_bases, const uint8_t* meth, int64_t n_meth, const uint32_t* cigar,
                             int64_t n_ops, int64_t ref_pos, int64_t left_len, int64_t right_len, uint8_t* out_bases, uint8_t* out_quals,
                             uint8_t* out_meth, int64_t* out_meth_n, uint32_t* out_cigar, int64_t* out_n_ops, int64_t* out_ref_pos) {
  if (n_bases < 0 || (n_bases > 0 && (!bases || !quals || !out_bases || !out_quals)) || n_ops < 0 || (n_ops > 0 && !cigar) || !out_cigar || !out_n_ops ||
      !out_ref_pos || !out_meth_n || (n_meth > 0 && (!meth || !out_meth)) || left_len < 0 || right_len < 0)
    return TRGT_ERR_INVALID;
  try {
    if (left_len + right_len >= n_bases) return -1;  // clip_bases.rs:10-12
    const size_t left = (size_t)left_len, right = (size_t)right_len, n = (size_t)n_bases, len = n - left - right;
    std::vector<uint32_t> ops; int64_t rp = ref_pos;
    if (!clip_bases_cigar(cigar, (size_t)n_ops, left, right, ops, rp)) return TRGT_ERR_INVALID;
    std::memcpy(out_bases, bases + left, len); std::memcpy(out_quals, quals + left, len);
    *out_meth_n = -1;
    if (n_meth >= 0) { std::vector<uint8_t> m; clip_bases_meth(bases, n, meth, (size_t)n_meth, left, right, m); std::copy(m.begin(), m.end(), out_meth); *out_meth_n = (int64_t)m.size(); }
    std::copy(ops.begin(), ops.end(), out_cigar); *out_n_ops = (int64_t)ops.size(); *out_ref_pos = rp;
    return (int64_t)len;
  } catch (const std::exception&) { return TRGT_ERR_NOMEM; }
}

int trgt_writer_open(const trgt_ingest* src, const trgt_writer_params* p, const char* vcf_path, const char* bam_path, trgt_writer** out) {
  try { return writer_open_impl(src, p, vcf_path, bam_path, out); } catch (const std::exception&) { return TRGT_ERR_NOMEM; }  // (exceptions never cross the C ABI)
}
int trgt_writer_write(trgt_writer* w, const trgt_ingest_batch* b, const trgt_locus_batch_out* o) {
  try { return writer_write_impl(w, b, o); } catch (const std::exception& e) { if (w) w->err = std::string("trgt_writer_write: ") + e.what(); return TRGT_ERR_NOMEM; }
}

int trgt_writer_close(trgt_writer* w) {
  if (!w) return TRGT_OK;
  const int brc = w->bg_wait();  // (write_behind: the last batch's pieces)
  const bool ok1 = w->vcf.close(), ok2 = !w->has_bam || w->bam.close();
  delete w;
  return brc != TRGT_OK ? brc : ok1 && ok2 ? TRGT_OK : TRGT_ERR_INVALID;
}

}  // extern "C"
