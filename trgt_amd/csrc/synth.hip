// trgt_amd/csrc/synth.hip -- deterministic synthetic locus batches (host code only).
//
// Implements the workloads of SURVEY.md Appendix E: BASELINE.json configs[1] ("10k synthetic single-motif STR loci,
// motif 3-6 bp, allele <= 200 bp, 30x HiFi", config 2), the genome-wide catalog mix of configs[3] (config 4) and the
// compound / N-motif loci of configs[4] (config 5): per-locus splitmix64 stream,
// draw order motif -> copy numbers -> flanks -> reads, HiFi-like error channel, +-1 unit stutter and
// 10 % truncated reads (which must end up with span = None).  Reads are "already clipped" to
// 2*flank_len of context (what clip_reads leaves, src/trgt/workflows/tr.rs:33-34).
#include <algorithm>
#include <cmath>
#include <string>
#include <thread>

#include "common.hpp"

namespace {

struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed) {}
  inline uint64_t next() {  // splitmix64
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  inline uint32_t below(uint32_t n) { return (uint32_t)(next() % n); }
  inline int range(int lo, int hi) { return lo + (int)below((uint32_t)(hi - lo + 1)); }  // inclusive
  inline double real() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  inline char base() { return "ACGT"[next() & 3]; }
};

struct LocusData {
  std::string motif, left, right, tr;  // contexts of context_len bases, reference allele
  std::vector<std::string> motifs;     // the locus' motif set (cfg2: the one motif)
  uint8_t genotyper = 0;               // 0 size, 1 cluster
  uint32_t allele_len[2];
  std::vector<std::string> reads;
  std::vector<uint8_t> hap, trunc;
};

bool is_power_of_shorter_unit(const std::string& m) {
  const size_t n = m.size();
  for (size_t u = 1; u < n; ++u) {
    if (n % u) continue;
    bool rep = true;
    for (size_t i = u; i < n && rep; ++i) rep = m[i] == m[i - u];
    if (rep) return true;
  }
  return false;
}

std::string channel(Rng& g, const std::string& in, const trgt_synth_params& p) {
  std::string out;
  out.reserve(in.size() + 8);
  for (char b : in) {
    const double r = g.real();
    if (r < p.del_rate) {
      // deleted
    } else if (r < p.del_rate + p.sub_rate) {
      char nb = b;
      while (nb == b) nb = g.base();
      out.push_back(nb);
    } else {
      out.push_back(b);
    }
    if (g.real() < p.ins_rate) out.push_back(g.base());
  }
  return out;
}

// cfg5: 2-10 motifs of length 2-12, at least one containing N, alleles <= max_allele_bp (300) built as
// consecutive runs of every motif (N positions filled uniformly per copy), Genotyper::Cluster.  Same draw order as cfg2:
// motifs, copy numbers, flanks, reads.
struct CompoundShape { int nm_lo, nm_hi, len_lo, len_hi; bool with_n; uint8_t genotyper; int max_bp; };
void gen_locus_compound(const trgt_synth_params& p, int64_t idx, LocusData& L, Rng& g, const CompoundShape& sh) {
  const int nm = g.range(sh.nm_lo, sh.nm_hi);
  L.motifs.resize((size_t)nm);
  for (auto& m : L.motifs) {
    const int n = g.range(sh.len_lo, sh.len_hi);
    do { m.assign((size_t)n, 'A'); for (int i = 0; i < n; ++i) m[(size_t)i] = g.base(); } while (n > 1 && is_power_of_shorter_unit(m));
  }
  if (sh.with_n) { std::string& m = L.motifs[g.below((uint32_t)nm)]; m[g.below((uint32_t)m.size())] = 'N'; }
  L.genotyper = sh.genotyper;
  std::vector<int> copies[2];
  for (int m = 0; m < nm; ++m) {
    const int cap = std::max(1, (sh.max_bp / nm) / (int)L.motifs[(size_t)m].size());
    const int c1 = g.range(1, cap);
    const double r = g.real();
    int delta = 0;
    if (r >= 0.30) { const int mag = g.range(1, 5); delta = (g.next() & 1) ? mag : -mag; }
    copies[0].push_back(c1); copies[1].push_back(std::min(cap, std::max(1, c1 + delta)));
  }
  struct Unit { size_t start, len; };
  std::string allele[2]; std::vector<Unit> units[2];
  for (int a = 0; a < 2; ++a) {
    for (int m = 0; m < nm; ++m)
      for (int k = 0; k < copies[a][(size_t)m]; ++k) {
        units[a].push_back({allele[a].size(), L.motifs[(size_t)m].size()});
        for (char ch : L.motifs[(size_t)m]) allele[a].push_back(ch == 'N' ? g.base() : ch);
      }
    L.allele_len[a] = (uint32_t)allele[a].size();
  }
  L.tr = allele[0];
  L.left.resize((size_t)p.context_len); L.right.resize((size_t)p.context_len);
  for (auto& ch : L.left) ch = g.base();
  for (auto& ch : L.right) ch = g.base();
  const int R = p.reads_per_locus;
  std::vector<uint8_t> hap((size_t)R);
  while (true) {
    int cnt1 = 0;
    for (int i = 0; i < R; ++i) { hap[(size_t)i] = (uint8_t)(g.next() & 1); cnt1 += hap[(size_t)i]; }
    const int need = std::min(5, R / 2);
    if (cnt1 >= need && R - cnt1 >= need) break;
  }
  L.reads.resize((size_t)R); L.hap = hap; L.trunc.assign((size_t)R, 0);
  for (int i = 0; i < R; ++i) {
    const int h = hap[(size_t)i];
    std::string rep = allele[h];
    if (g.real() < p.stutter_rate) {  // one unit copied or dropped
      const Unit u = units[h][g.below((uint32_t)units[h].size())];
      if (g.next() & 1) rep.insert(u.start + u.len, allele[h].substr(u.start, u.len));
      else if (units[h].size() > 1) rep.erase(u.start, u.len);
    }
    std::string rd = channel(g, L.left + rep + L.right, p);
    if (g.real() < p.truncate_rate && rd.size() > 2) {
      const size_t cut = 1 + g.below((uint32_t)(rd.size() - 1));
      rd = cut >= rd.size() - cut ? rd.substr(0, cut) : rd.substr(cut);
      L.trunc[(size_t)i] = 1;
    }
    L.reads[(size_t)i].swap(rd);
  }
}

void gen_locus(const trgt_synth_params& p, int64_t idx, LocusData& L) {
  Rng g(p.seed ^ ((uint64_t)(idx + 1) * 0x9E3779B97F4A7C15ull));
  if (p.config == 5) { gen_locus_compound(p, idx, L, g, CompoundShape{2, 10, 2, 12, true, 1, p.max_allele_bp}); return; }
  if (p.config == 4) {
    // genome-wide catalog stand-in (SURVEY.md Appendix E, cfg4 note): 70 % single STR loci as in cfg2, 20 % loci with 2-5 motifs
    // of 2-12 bp, 10 % VNTR loci with one motif of 7-60 bp and alleles up to 600 bp.  Size genotyper throughout.
    const double r = g.real();
    if (r >= 0.90) { gen_locus_compound(p, idx, L, g, CompoundShape{1, 1, 7, 60, false, 0, 600}); return; }
    if (r >= 0.70) { gen_locus_compound(p, idx, L, g, CompoundShape{2, 5, 2, 12, false, 0, std::max(p.max_allele_bp, 240)}); return; }
  }
  // motif
  const int n = g.range(3, 6);
  do {
    L.motif.assign((size_t)n, 'A');
    for (int i = 0; i < n; ++i) L.motif[i] = g.base();
  } while (is_power_of_shorter_unit(L.motif));
  // copy numbers
  const int max_copies = std::max(5, p.max_allele_bp / n);
  int c1 = g.range(5, max_copies), c2;
  {
    const double r = g.real();
    int delta = 0;
    if (r >= 0.30) {
      const int mag = r < 0.80 ? g.range(1, 5) : g.range(6, 20);
      delta = (g.next() & 1) ? mag : -mag;
    }
    c2 = std::min(max_copies, std::max(3, c1 + delta));
  }
  std::string allele[2];
  const int copies[2] = {c1, c2};
  for (int a = 0; a < 2; ++a) {
    for (int i = 0; i < copies[a]; ++i) allele[a] += L.motif;
    L.allele_len[a] = (uint32_t)allele[a].size();
  }
  L.tr = allele[0];  // the reference allele of the synthetic locus
  L.motifs.assign(1, L.motif);
  // flanks / context
  L.left.resize((size_t)p.context_len);
  L.right.resize((size_t)p.context_len);
  for (auto& ch : L.left) ch = g.base();
  for (auto& ch : L.right) ch = g.base();
  // reads
  const int R = p.reads_per_locus;
  std::vector<uint8_t> hap((size_t)R);
  while (true) {
    int cnt1 = 0;
    for (int i = 0; i < R; ++i) { hap[i] = (uint8_t)(g.next() & 1); cnt1 += hap[i]; }
    const int need = std::min(5, R / 2);
    if (cnt1 >= need && R - cnt1 >= need) break;
  }
  L.reads.resize((size_t)R); L.hap = hap; L.trunc.assign((size_t)R, 0);
  for (int i = 0; i < R; ++i) {
    int cp = copies[hap[i]];
    if (g.real() < p.stutter_rate) cp = std::max(1, cp + ((g.next() & 1) ? 1 : -1));
    std::string hapseq = L.left;
    for (int k = 0; k < cp; ++k) hapseq += L.motif;
    hapseq += L.right;
    std::string rd = channel(g, hapseq, p);
    if (g.real() < p.truncate_rate && rd.size() > 2) {
      const size_t cut = 1 + g.below((uint32_t)(rd.size() - 1));
      rd = cut >= rd.size() - cut ? rd.substr(0, cut) : rd.substr(cut);  // keep the longer side
      L.trunc[i] = 1;
    }
    L.reads[i].swap(rd);
  }
}

template <typename T>
T* dup(const std::vector<T>& v) {
  T* p = (T*)std::malloc(std::max<size_t>(1, v.size()) * sizeof(T));
  if (!v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(T));
  return p;
}

}  // namespace

extern "C" {

void trgt_synth_default_params(trgt_synth_params* p, int config) {
  std::memset(p, 0, sizeof *p);
  p->seed = 20250509ull; p->config = config; p->reads_per_locus = 30; p->context_len = 500; p->flank_len = 250;
  p->max_allele_bp = config == 5 ? 300 : 200;
  p->sub_rate = 5e-4; p->del_rate = 2.5e-4; p->ins_rate = 2.5e-4; p->stutter_rate = 0.05; p->truncate_rate = 0.10;
}

int trgt_synth_generate(const trgt_synth_params* p, int64_t first_locus, int64_t n_loci, int threads, trgt_synth_batch** out) {
  if (!p || !out || n_loci < 0 || p->context_len < p->flank_len || p->reads_per_locus < 1) return TRGT_ERR_INVALID;
  if (p->config != 2 && p->config != 4 && p->config != 5) return TRGT_ERR_UNSUPPORTED;  // cfg2 / cfg4 STR loci, cfg5 compound loci
  std::vector<LocusData> loci((size_t)n_loci);
  if (threads < 1) threads = (int)std::max(1u, std::thread::hardware_concurrency());
  threads = (int)std::min<int64_t>(threads, std::max<int64_t>(1, n_loci));
  {
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t)
      th.emplace_back([&, t]() { for (int64_t l = t; l < n_loci; l += threads) gen_locus(*p, first_locus + l, loci[(size_t)l]); });
    for (auto& x : th) x.join();
  }
  trgt_synth_batch* b = (trgt_synth_batch*)std::calloc(1, sizeof(trgt_synth_batch));
  const int F = p->flank_len;
  std::vector<uint64_t> lf_off, rf_off, tr_off, lrb{0}, read_off;
  std::vector<uint32_t> lf_len, rf_len, tr_len, motif_off{0}, set_begin{0}, read_len, true_len;
  std::vector<uint8_t> ploidy, hap, trunc, genotyper;
  std::string flank, tr, motifs;
  uint64_t read_bytes = 0, n_reads = 0;
  for (auto& L : loci) { for (auto& r : L.reads) read_bytes += r.size(); n_reads += L.reads.size(); }
  uint8_t* read_blob = (uint8_t*)std::malloc(std::max<uint64_t>(1, read_bytes));
  uint64_t rpos = 0;
  for (auto& L : loci) {
    lf_off.push_back(flank.size()); lf_len.push_back((uint32_t)F); flank += L.left.substr(L.left.size() - (size_t)F);
    rf_off.push_back(flank.size()); rf_len.push_back((uint32_t)F); flank += L.right.substr(0, (size_t)F);
    tr_off.push_back(tr.size()); tr_len.push_back((uint32_t)L.tr.size()); tr += L.tr;
    for (auto& m : L.motifs) { motifs += m; motif_off.push_back((uint32_t)motifs.size()); }
    set_begin.push_back((uint32_t)motif_off.size() - 1);
    ploidy.push_back(2); genotyper.push_back(L.genotyper);
    true_len.push_back(L.allele_len[0]); true_len.push_back(L.allele_len[1]);
    for (size_t i = 0; i < L.reads.size(); ++i) {
      read_off.push_back(rpos); read_len.push_back((uint32_t)L.reads[i].size());
      std::memcpy(read_blob + rpos, L.reads[i].data(), L.reads[i].size());
      rpos += L.reads[i].size();
      hap.push_back(L.hap[i]); trunc.push_back(L.trunc[i]);
    }
    lrb.push_back(read_off.size());
    std::vector<std::string>().swap(L.reads);
  }
  b->n_loci = n_loci; b->n_reads = (int64_t)n_reads; b->n_motifs = (int64_t)motif_off.size() - 1;
  b->flank_bytes = flank.size(); b->tr_bytes = tr.size(); b->motif_bytes = motifs.size(); b->read_bytes = read_bytes;
  b->flank_blob = (uint8_t*)std::malloc(std::max<size_t>(1, flank.size())); std::memcpy(b->flank_blob, flank.data(), flank.size());
  b->tr_blob = (uint8_t*)std::malloc(std::max<size_t>(1, tr.size())); std::memcpy(b->tr_blob, tr.data(), tr.size());
  b->motif_blob = (uint8_t*)std::malloc(std::max<size_t>(1, motifs.size())); std::memcpy(b->motif_blob, motifs.data(), motifs.size());
  b->lf_off = dup(lf_off); b->lf_len = dup(lf_len); b->rf_off = dup(rf_off); b->rf_len = dup(rf_len);
  b->tr_off = dup(tr_off); b->tr_len = dup(tr_len); b->motif_off = dup(motif_off); b->set_motif_begin = dup(set_begin);
  b->ploidy = dup(ploidy); b->locus_read_begin = dup(lrb); b->read_blob = read_blob; b->read_off = dup(read_off);
  b->read_len = dup(read_len); b->true_allele_len = dup(true_len); b->read_hap = dup(hap); b->read_truncated = dup(trunc);
  b->genotyper = dup(genotyper);
  *out = b;
  return TRGT_OK;
}

void trgt_synth_free(trgt_synth_batch* b) {
  if (!b) return;
  void* ptrs[] = {b->flank_blob, b->lf_off, b->lf_len, b->rf_off, b->rf_len, b->tr_blob, b->tr_off, b->tr_len, b->motif_blob,
                  b->motif_off, b->set_motif_begin, b->ploidy, b->locus_read_begin, b->read_blob, b->read_off, b->read_len,
                  b->true_allele_len, b->read_hap, b->read_truncated, b->genotyper};
  for (void* q : ptrs) std::free(q);
  std::free(b);
}

}  // extern "C"
