// oracle/locus.cpp -- CPU ORACLE (test infrastructure, never shipped / never on
// the product path).  Restatement of the callers that sit either side of the
// WFA/HMM hot path in PacificBiosciences/trgt v3.0.0, restricted to what
// synthetic (already clipped, no HP tag, no SNV, no methylation) reads reach:
//   find_spans / find_tr_spans      src/trgt/genotype/span_locater.rs:7-68
//   get_spanning_reads              src/trgt/workflows/tr.rs:111-184
//   utils::align                    src/utils/align.rs:14-28
//   repair_consensus & friends      src/trgt/genotype/consensus.rs:5-154
//   genotype_size::genotype         src/trgt/genotype/genotype_size.rs:6-125
//   diploid / haploid genotype      diploid.rs:5-103, haploid.rs:3-30
//   label_with_hmm, allele assembly src/trgt/workflows/tr.rs:77-101,454-492
//   filter_impure_trs               src/trgt/workflows/tr.rs:400-452
//   genotype_cluster::genotype      src/trgt/genotype/genotype_cluster.rs:12-286
//   Ward linkage                    kodama 0.3.0 (crates.io, Cargo.lock:839-842) -- UN-VENDORED: restated from the
//                                   published NN-chain algorithm (Muellner 2011, fastcluster; kodama is its Rust port).
//                                   PARITY UNPINNED: the reference holds no test of cluster(); the dendrogram is checked
//                                   against scipy.cluster.hierarchy.linkage(method="ward") in tests/test_oracle_cluster.py,
//                                   the in-place mutation of the distance matrix (read back by central_read,
//                                   genotype_cluster.rs:27-29,74,82-83) follows the NN-chain update order.
//   MC / MS / AP / AL / ALLR / SD   src/trgt/writers/write_vcf.rs:286-377
//   genotype_flank::genotype        src/trgt/genotype/genotype_flank.rs:9-290 (tr.rs:69-75), pinned by its two inline tests
//                                   (tests/golden/caller_kats.json F1 / F2).  Runs when the caller passes the per-read fields it
//                                   reads (orc_locus_analyze_meta); for reads without HP tags and mismatch offsets it returns None
//                                   (get_trs_with_hp needs 70 % tagged reads, a single candidate genotype is "homozygous").
#include "oracle_internal.h"

#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <map>
#include <thread>

namespace orc {

static orc_wfa_params flank_params(int mism, int gapo, int gape) {  // genotype.rs:66-80
  orc_wfa_params p;
  orc_wfa_default_params(&p);
  p.metric = 3; p.mismatch = mism; p.gap_open1 = gapo; p.gap_ext1 = gape;
  p.span = 1; p.pattern_begin_free = 0; p.pattern_end_free = 0; p.text_begin_free = -1; p.text_end_free = -1;
  p.scope = 1; p.memory_mode = 0; p.heuristic = 0;
  return p;
}
static orc_wfa_params consensus_params() {  // genotype.rs:82-86: BiWFA affine(2,5,1), default heuristic
  orc_wfa_params p;
  orc_wfa_default_params(&p);
  p.metric = 3; p.mismatch = 2; p.gap_open1 = 5; p.gap_ext1 = 1;
  p.span = 0; p.scope = 1; p.memory_mode = 3;
  return p;
}

// span_locater.rs:9-27 for one read
SpanOpt find_span(const uint8_t* piece, int plen, const uint8_t* s, int slen, const orc_wfa_params& fp, double threshold,
                  bool* used_wfa, int64_t* cells) {
  SpanOpt r;
  if (used_wfa) *used_wfa = false;
  for (int st = 0; st + plen <= slen; ++st)
    if (std::memcmp(s + st, piece, (size_t)plen) == 0) { r.start = st; r.end = st + plen; return r; }
  if (used_wfa) *used_wfa = true;
  WfaResult a = wfa_align(fp, piece, plen, s, slen);
  if (cells) *cells += a.cells;
  const int nm = cigar_count_matches(a.ops);
  if ((double)(size_t)nm >= threshold) {
    uint32_t s4[4];
    alignment_span(fp, a.ops, plen, slen, s4);
    r.start = (int)s4[2]; r.end = (int)s4[3];
  }
  return r;
}

// ---- diploid.rs / haploid.rs ---------------------------------------------
struct TrSize { int size, ci_lo, ci_hi; };

static double dip_penalty(int sa, int la, const std::vector<int>& sizes, const std::vector<int>& counts) {
  double penalty = 0.0;
  const double max_frac = (std::abs(sa - la) <= 100) ? 0.25 : 0.05;
  for (size_t i = 0; i < sizes.size(); ++i) {
    const int st = sizes[i] != sa ? 10 + 2 * std::abs(sa - sizes[i]) : 0;
    const int lt = sizes[i] != la ? 10 + 2 * std::abs(la - sizes[i]) : 0;
    const double term = (double)std::min(st, lt) + max_frac * (double)std::max(st, lt);
    penalty += term * (double)counts[i];
  }
  return penalty;
}

static std::vector<TrSize> diploid_genotype(const std::vector<int>& sizes, const std::vector<int>& counts) {
  struct Cand { int a, b; double pen; };
  std::vector<Cand> c;
  for (size_t si = 0; si < sizes.size(); ++si)
    for (size_t li = si; li < sizes.size(); ++li) c.push_back({sizes[si], sizes[li], dip_penalty(sizes[si], sizes[li], sizes, counts)});
  std::stable_sort(c.begin(), c.end(), [](const Cand& x, const Cand& y) { return x.pen < y.pen; });
  int short_size = std::min(c[0].a, c[0].b), long_size = std::max(c[0].a, c[0].b);
  if (short_size != long_size && sizes.size() >= 2) {
    int coverage = 0;
    for (int x : counts) coverage += x;
    std::vector<size_t> idx(sizes.size());
    for (size_t i = 0; i < idx.size(); ++i) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return counts[a] > counts[b]; });
    const double top_frac = (double)counts[idx[0]] / (double)coverage;
    const int max_len = *std::max_element(sizes.begin(), sizes.end()), min_len = *std::min_element(sizes.begin(), sizes.end());
    if (top_frac > 0.60 && max_len - min_len <= 6) { short_size = long_size = sizes[idx[0]]; }
  }
  TrSize s{short_size, short_size, short_size}, l{long_size, long_size, long_size};
  for (int size : sizes) {  // get_ci
    if (std::abs(size - short_size) <= std::abs(size - long_size)) { s.ci_lo = std::min(s.ci_lo, size); s.ci_hi = std::max(s.ci_hi, size); }
    else { l.ci_lo = std::min(l.ci_lo, size); l.ci_hi = std::max(l.ci_hi, size); }
  }
  return {s, l};
}

static std::vector<TrSize> haploid_genotype(const std::vector<int>& sizes, const std::vector<int>& counts) {
  int best = -1; double best_pen = 0;
  for (size_t a = 0; a < sizes.size(); ++a) {
    double pen = 0.0;
    for (size_t i = 0; i < sizes.size(); ++i) {
      const double term = sizes[i] != sizes[a] ? 10.0 + 2.0 * (double)std::abs(sizes[a] - sizes[i]) : 0.0;
      pen += term * (double)counts[i];
    }
    if (best < 0 || pen < best_pen) { best = (int)a; best_pen = pen; }  // stable sort + first
  }
  return {TrSize{sizes[best], *std::min_element(sizes.begin(), sizes.end()), *std::max_element(sizes.begin(), sizes.end())}};
}

// ---- utils::align + consensus.rs ------------------------------------------
typedef std::vector<std::pair<int, char>> Cigar;

static std::vector<Cigar> align_all(const std::string& backbone, const std::vector<std::string>& seqs, int64_t* cells, int64_t* n_aln) {
  const orc_wfa_params cp = consensus_params();
  std::vector<Cigar> out;
  for (auto& s : seqs) {
    WfaResult r = wfa_align(cp, (const uint8_t*)backbone.data(), (int)backbone.size(), (const uint8_t*)s.data(), (int)s.size());
    if (cells) *cells += r.cells;
    if (n_aln) *n_aln += 1;
    Cigar c;
    for (uint32_t e : cigar_rle(r.ops, true)) {  // get_sam_cigar(true) + decode_sam_cigar
      static const char* dec = "MIDNSHP=X";
      const uint32_t code = e & 0xF;
      c.push_back({(int)(e >> 4), code <= 8 ? dec[code] : '?'});
    }
    out.push_back(c);
  }
  return out;
}

static std::string repair_consensus(const std::string& reference, const std::vector<std::string>& seqs, const std::vector<Cigar>& aligns) {
  std::vector<std::array<int, 5>> ref_counts(reference.size(), std::array<int, 5>{0, 0, 0, 0, 0});
  std::vector<std::vector<std::string>> ref_inserts(reference.size() + 1);
  auto base_idx = [](char b) { switch (b) { case 'A': return 0; case 'T': return 1; case 'C': return 2; case 'G': return 3; default: assert(!"unexpected base"); return 0; } };
  for (size_t si = 0; si < aligns.size(); ++si) {
    const std::string& seq = seqs[si];
    size_t x = 0, y = 0;
    for (auto& op : aligns[si]) {
      const size_t n = (size_t)op.first;
      switch (op.second) {
        case '=': case 'M': case 'X':
          for (size_t i = 0; i < n; ++i) ref_counts[y + i][base_idx(seq[x + i])] += 1;
          x += n; y += n; break;
        case 'D':
          for (size_t i = 0; i < n; ++i) ref_counts[y + i][4] += 1;
          y += n; break;
        case 'I': ref_inserts[y].push_back(seq.substr(x, n)); x += n; break;
        default: assert(!"unexpected cigar op");
      }
    }
  }
  std::string consensus;
  for (size_t pos = 0; pos < reference.size(); ++pos) {
    int best = 0;  // max_by_key returns the LAST maximum
    for (int i = 1; i < 5; ++i)
      if (ref_counts[pos][i] >= ref_counts[pos][best]) best = i;
    if (ref_inserts[pos].size() > seqs.size() / 2) {  // get_ins_consensus
      auto& ins = ref_inserts[pos];
      std::sort(ins.begin(), ins.end());
      const size_t without = seqs.size() - ins.size();
      std::string top; size_t top_count = 0;
      for (size_t i = 0; i < ins.size();) {
        size_t j = i;
        while (j < ins.size() && ins[j] == ins[i]) ++j;
        if (j - i > top_count) { top_count = j - i; top = ins[i]; }  // stable desc sort, first
        i = j;
      }
      if (top_count > without) consensus += top;
    }
    if (best != 4) consensus.push_back("ATCG"[best]);
  }
  return consensus;
}

struct SizeGt {
  std::vector<TrSize> gt; std::vector<std::string> alleles; std::vector<int> classification;
};

// genotype_size::genotype
static SizeGt genotype_size(int ploidy, const std::vector<std::string>& seqs, int64_t* cells, int64_t* n_aln) {
  SizeGt out;
  std::vector<int> lens;
  for (auto& s : seqs) lens.push_back((int)s.size());
  std::sort(lens.begin(), lens.end());
  std::vector<int> ulen, ucnt;
  for (size_t i = 0; i < lens.size();) { size_t j = i; while (j < lens.size() && lens[j] == lens[i]) ++j; ulen.push_back(lens[i]); ucnt.push_back((int)(j - i)); i = j; }
  out.gt = ploidy == 1 ? haploid_genotype(ulen, ucnt) : diploid_genotype(ulen, ucnt);
  std::vector<int> allele_lens;
  for (auto& a : out.gt) allele_lens.push_back(a.size);
  std::vector<std::string> sorted = seqs;  // get_seq_hist
  std::sort(sorted.begin(), sorted.end());
  std::vector<std::string> useq; std::vector<int> cnt;
  for (size_t i = 0; i < sorted.size();) { size_t j = i; while (j < sorted.size() && sorted[j] == sorted[i]) ++j; useq.push_back(sorted[i]); cnt.push_back((int)(j - i)); i = j; }
  auto closest = [&](int allele) { int c = -1; for (auto& s : useq) { const int l = (int)s.size(); if (c < 0) { c = l; continue; } if (std::abs(c - allele) > std::abs(l - allele)) c = l; } return c; };
  auto most_freq = [&](int length) { int best = -1; for (size_t i = 0; i < useq.size(); ++i) if ((int)useq[i].size() == length && (best < 0 || cnt[i] >= cnt[best])) best = (int)i; return useq[best]; };
  std::vector<std::string> alleles;  // consensus::get_consensus
  alleles.push_back(most_freq(closest(allele_lens[0])));
  if (allele_lens.size() != 1 && allele_lens[0] != allele_lens[1]) alleles.push_back(most_freq(closest(allele_lens[1])));
  // split()
  std::vector<std::pair<std::vector<std::string>, std::vector<int>>> by_allele;
  if (allele_lens.size() == 1) by_allele.push_back({useq, cnt});
  else {
    const int a1 = allele_lens[0], a2 = allele_lens[1];
    std::pair<std::vector<std::string>, std::vector<int>> g1, g2;
    for (size_t i = 0; i < useq.size(); ++i) {
      const int l = (int)useq[i].size();
      if (std::abs(l - a1) <= std::abs(l - a2)) { g1.first.push_back(useq[i]); g1.second.push_back(cnt[i]); }
      if (std::abs(l - a2) < std::abs(l - a1)) { g2.first.push_back(useq[i]); g2.second.push_back(cnt[i]); }
    }
    by_allele.push_back(g1); by_allele.push_back(g2);
  }
  for (size_t ai = 0; ai < alleles.size(); ++ai) {
    auto& g = by_allele[ai];
    int coverage = 0, ref_count = 0;
    for (int c : g.second) coverage += c;
    for (size_t i = 0; i < g.first.size(); ++i) if (g.first[i] == alleles[ai]) { ref_count = g.second[i]; break; }
    if (!(2 * ref_count >= coverage)) {
      auto aligns = align_all(alleles[ai], g.first, cells, n_aln);
      alleles[ai] = repair_consensus(alleles[ai], g.first, aligns);
    }
  }
  if (ploidy == 2 && alleles.size() == 1) alleles.push_back(alleles[0]);
  out.classification.assign(seqs.size(), 0);
  int tie = 1;
  for (size_t i = 0; i < seqs.size(); ++i) {
    if (alleles.size() == 2) {
      const int d1 = std::abs((int)seqs[i].size() - (int)alleles[0].size()), d2 = std::abs((int)seqs[i].size() - (int)alleles[1].size());
      if (d1 < d2) out.classification[i] = 0;
      else if (d1 > d2) out.classification[i] = 1;
      else { tie = (tie + 1) % 2; out.classification[i] = tie; }
    }
  }
  out.alleles = alleles;
  return out;
}


// ---- genotype_flank::genotype (src/trgt/genotype/genotype_flank.rs:9-290): re-genotyping of a locus with two alleles of similar
// length from the haplotype tags of the reads or, failing that, from heterozygous SNVs in the flanks.
struct FlankRead { int hp; int start_off, end_off; const int32_t* mm; size_t n_mm; };  // hp < 0: no HP tag
struct FlankGt { std::vector<TrSize> gt; std::vector<std::string> alleles; std::vector<int> assignment; };
typedef std::vector<int8_t> Profile;  // -1 None, 0 Some(false), 1 Some(true): Option<bool>'s derived order

static double ln_sum_exp(double a, double b) { const double m = std::max(a, b); return m + std::log(std::exp(a - m) + std::exp(b - m)); }

static bool flank_groups_hp(const std::vector<FlankRead>& reads, std::vector<int> groups[2], std::vector<int>& assignment) {  // :43-76
  int tie = 1; size_t unassigned = 0;
  for (size_t i = 0; i < reads.size(); ++i) {
    if (reads[i].hp == 1) { assignment.push_back(0); groups[0].push_back((int)i); }
    else if (reads[i].hp == 2) { assignment.push_back(1); groups[1].push_back((int)i); }
    else { tie = (tie + 1) % 2; assignment.push_back(tie); groups[tie].push_back((int)i); ++unassigned; }
  }
  const double prop = (double)(reads.size() - unassigned) / (double)reads.size();
  return !groups[0].empty() && !groups[1].empty() && prop >= 0.7;
}

static bool flank_groups_snv(const std::vector<FlankRead>& reads, std::vector<int> groups[2], std::vector<int>& assignment) {  // :78-138
  const size_t n = reads.size();
  if (n == 0) return false;
  // get_analysis_region (:206-226)
  const size_t skip = (size_t)std::round((double)n * (1.0 - 0.85));
  std::vector<int> so, eo;
  for (auto& r : reads) { so.push_back(r.start_off); eo.push_back(r.end_off); }
  std::sort(so.begin(), so.end()); std::sort(eo.begin(), eo.end());
  if (skip >= n) return false;  // (nth on too short an iterator: unwrap of None -- cannot happen, round(0.15 n) < n)
  const int reg0 = so[n - 1 - skip], reg1 = eo[skip];
  // call_snvs (:271-286)
  std::map<int, size_t> counts;
  for (auto& r : reads) for (size_t i = 0; i < r.n_mm; ++i) if (reg0 <= r.mm[i] && r.mm[i] <= reg1) counts[r.mm[i]] += 1;
  std::vector<int> snvs;
  for (auto& kv : counts) if ((double)kv.second / (double)n >= 0.20) snvs.push_back(kv.first);
  // get_profiles (:250-269)
  std::vector<Profile> prof(n);
  for (size_t i = 0; i < n; ++i)
    for (int snv : snvs) {
      if (snv < reads[i].start_off || snv > reads[i].end_off) prof[i].push_back(-1);
      else prof[i].push_back(std::binary_search(reads[i].mm, reads[i].mm + reads[i].n_mm, snv) ? 1 : 0);
    }
  // get_candidate_gts (:228-248)
  std::vector<Profile> haps;
  for (auto& p : prof) if (std::all_of(p.begin(), p.end(), [](int8_t v) { return v >= 0; })) haps.push_back(p);
  std::sort(haps.begin(), haps.end());
  if ((double)haps.size() / (double)n < 0.40) return false;
  haps.erase(std::unique(haps.begin(), haps.end()), haps.end());
  std::vector<std::pair<Profile, Profile>> cands;
  for (size_t i = 0; i < haps.size(); ++i) for (size_t j = i; j < haps.size(); ++j) cands.push_back({haps[i], haps[j]});
  if (cands.size() <= 1) return false;
  auto eval = [&](const Profile& p, const Profile& h) { double s = 0.0; for (size_t k = 0; k < p.size() && k < h.size(); ++k) if (p[k] >= 0) s += p[k] == h[k] ? std::log(0.9) : std::log(1.0 - 0.9); return s; };
  size_t top = 0; double top_ll = 0.0;
  for (size_t c = 0; c < cands.size(); ++c) {
    double ll = 0.0;
    for (auto& p : prof) ll += ln_sum_exp(eval(p, cands[c].first), eval(p, cands[c].second)) - std::log(2.0);
    if (c == 0 || ll >= top_ll) { top = c; top_ll = ll; }  // Iterator::max_by: the last maximum
  }
  if (cands[top].first == cands[top].second) return false;
  auto dist = [](const Profile& p, const Profile& h) { size_t d = 0; for (size_t k = 0; k < p.size() && k < h.size(); ++k) d += p[k] >= 0 && p[k] == h[k]; return d; };
  int tie = 1;
  for (size_t i = 0; i < n; ++i) {
    const size_t d1 = dist(prof[i], cands[top].first), d2 = dist(prof[i], cands[top].second);
    if (d1 < d2) { assignment.push_back(0); groups[0].push_back((int)i); }
    else if (d1 > d2) { assignment.push_back(1); groups[1].push_back((int)i); }
    else { tie = (tie + 1) % 2; assignment.push_back(tie); groups[0].push_back((int)i); groups[1].push_back((int)i); }
  }
  return true;
}

static bool genotype_flank(const std::vector<FlankRead>& reads, const std::vector<std::string>& trs, FlankGt& out, int64_t* cells, int64_t* n_aln) {
  std::vector<int> groups[2]; std::vector<int> assignment;
  if (reads.empty()) return false;
  if (!flank_groups_hp(reads, groups, assignment)) {
    groups[0].clear(); groups[1].clear(); assignment.clear();
    if (trs.empty() || !flank_groups_snv(reads, groups, assignment)) return false;
  }
  out = FlankGt();
  for (int g = 0; g < 2; ++g) {
    std::vector<std::string> seqs;
    for (int i : groups[g]) seqs.push_back(trs[(size_t)i]);
    if (seqs.empty()) return false;  // median of nothing: simple_consensus returns None
    // simple_consensus (:147-170): utils::median is an f32 -- (a + b) as f32 / 2.0 for an even count -- truncated to usize
    std::vector<int> lens;
    for (auto& q : seqs) lens.push_back((int)q.size());
    std::sort(lens.begin(), lens.end());
    const float med = lens.size() % 2 ? (float)lens[lens.size() / 2] : (float)(lens[lens.size() / 2 - 1] + lens[lens.size() / 2]) / 2.0f;
    const size_t median_len = (size_t)med;
    std::map<std::string, size_t> cnt;
    for (auto& q : seqs) cnt[q] += 1;
    size_t top = 0;
    for (auto& kv : cnt) top = std::max(top, kv.second);
    const std::string* best = nullptr; size_t best_delta = 0;
    for (auto& kv : cnt) {
      if (kv.second != top) continue;
      const size_t d = kv.first.size() > median_len ? kv.first.size() - median_len : median_len - kv.first.size();
      if (!best || d < best_delta) { best = &kv.first; best_delta = d; }  // min_by_key: the first minimum
    }
    const double freq = (double)top / (double)seqs.size();
    std::string allele = *best;
    if (freq < 0.5) { auto al = align_all(*best, seqs, cells, n_aln); allele = repair_consensus(*best, seqs, al); }
    const int lo = lens.front(), hi = lens.back();
    out.gt.push_back(TrSize{(int)allele.size(), lo, hi});
    out.alleles.push_back(allele);
  }
  out.assignment = assignment;
  if (out.alleles[0].size() > out.alleles[1].size()) {  // smaller allele first
    std::swap(out.gt[0], out.gt[1]); std::swap(out.alleles[0], out.alleles[1]);
    for (int& a : out.assignment) a = (a + 1) % 2;
  }
  return true;
}

// ---- kodama 0.3.0 linkage(dists, n, Method::Ward), restated --------------------------------------------------------
// Ward in kodama: square the condensed matrix in place, NN-chain with Lance-Williams updates written into the
// surviving (larger-index) cluster's row/column, stable sort of the steps by dissimilarity, SciPy-style relabelling
// through a union-find, then sqrt of the step dissimilarities.  The matrix stays mutated (squared + updated).
struct WardStep { int c1, c2; double diss; int size; };

static inline size_t cidx(int n, int i, int j) {  // i < j
  return (size_t)n * (size_t)i - (size_t)i * ((size_t)i + 1) / 2 + (size_t)(j - i - 1);
}

std::vector<WardStep> ward_linkage(std::vector<double>& dis, int n) {
  std::vector<WardStep> steps;
  for (double& d : dis) d = d * d;
  if (n < 2) return steps;
  std::vector<char> active((size_t)n, 1);
  std::vector<int> sizes((size_t)n, 1), chain;
  auto D = [&](int i, int j) -> double& { return dis[cidx(n, i, j)]; };
  for (int it = 0; it < n - 1; ++it) {
    int a, b; double mn;
    if (chain.size() < 4) {
      a = 0; while (!active[(size_t)a]) ++a;
      chain.clear(); chain.push_back(a);
      b = a + 1; while (!active[(size_t)b]) ++b;
      mn = D(a, b);
      for (int i = b + 1; i < n; ++i) if (active[(size_t)i] && D(a, i) < mn) { mn = D(a, i); b = i; }
    } else {
      chain.pop_back(); chain.pop_back();
      b = chain.back(); chain.pop_back();
      a = chain.back();
      mn = a < b ? D(a, b) : D(b, a);
    }
    while (true) {
      chain.push_back(b);
      for (int x = 0; x < b; ++x) if (active[(size_t)x] && D(x, b) < mn) { mn = D(x, b); a = x; }
      for (int x = b + 1; x < n; ++x) if (active[(size_t)x] && D(b, x) < mn) { mn = D(b, x); a = x; }
      b = a;
      a = chain.back();
      if (b == chain[chain.size() - 2]) break;
    }
    if (a > b) std::swap(a, b);
    const double sa = (double)sizes[(size_t)a], sb = (double)sizes[(size_t)b];
    auto upd = [&](double da, double& db, int x) {
      const double sx = (double)sizes[(size_t)x];
      const double num = ((sx + sa) * da) + ((sx + sb) * db) - (sx * mn);
      db = num / (sa + sb + sx);
    };
    for (int x = 0; x < a; ++x) if (active[(size_t)x]) upd(D(x, a), D(x, b), x);
    for (int x = a + 1; x < b; ++x) if (active[(size_t)x]) upd(D(a, x), D(x, b), x);
    for (int x = b + 1; x < n; ++x) if (active[(size_t)x]) upd(D(a, x), D(b, x), x);
    sizes[(size_t)b] += sizes[(size_t)a];
    active[(size_t)a] = 0;
    steps.push_back({a, b, mn, sizes[(size_t)b]});
  }
  std::stable_sort(steps.begin(), steps.end(), [](const WardStep& x, const WardStep& y) { return x.diss < y.diss; });
  std::vector<int> parent((size_t)(2 * n - 1), -1);
  auto find = [&](int x) { int r = x; while (parent[(size_t)r] >= 0) r = parent[(size_t)r]; while (parent[(size_t)x] >= 0) { const int nx = parent[(size_t)x]; parent[(size_t)x] = r; x = nx; } return r; };
  auto csize = [&](int label) { return label < n ? 1 : steps[(size_t)(label - n)].size; };
  for (size_t i = 0; i < steps.size(); ++i) {
    int ra = find(steps[i].c1), rb = find(steps[i].c2);
    if (ra > rb) std::swap(ra, rb);
    steps[i] = {ra, rb, steps[i].diss, csize(ra) + csize(rb)};
    parent[(size_t)ra] = parent[(size_t)rb] = n + (int)i;
  }
  for (auto& s : steps) s.diss = std::sqrt(s.diss);
  return steps;
}

// genotype_cluster.rs:154-227
static std::vector<std::vector<int>> cluster_groups(int n, std::vector<double>& dists) {
  if (n == 2) return {{0}, {1}};
  std::vector<WardStep> steps = ward_linkage(dists, n);
  auto csize = [&](int label) { return label < n ? 1 : steps[(size_t)(label - n)].size; };
  double cutoff = 0.0;
  const int min_cluster = std::max(2, (int)std::round(0.01 * (double)n));
  for (size_t i = steps.size(); i-- > 0;) {
    if (std::min(csize(steps[i].c1), csize(steps[i].c2)) >= min_cluster) { cutoff = steps[i].diss - 0.0001; break; }
  }
  std::vector<std::vector<int>> out;
  if (cutoff == 0.0) {
    out.resize(2);
    for (int i = 0; i < n; ++i) out[(size_t)(i & 1)].push_back(i);
    return out;
  }
  int num_groups = 0;
  std::vector<int> member((size_t)(2 * n - 1), -1);
  for (size_t i = steps.size(); i-- > 0;) {
    const int c = (int)i + n;
    if (steps[i].diss <= cutoff) {
      if (member[(size_t)c] < 0) member[(size_t)c] = num_groups++;
      member[(size_t)steps[i].c1] = member[(size_t)c];
      member[(size_t)steps[i].c2] = member[(size_t)c];
    }
  }
  std::vector<int> g((size_t)n);
  for (int i = 0; i < n; ++i) g[(size_t)i] = member[(size_t)i] >= 0 ? member[(size_t)i] : num_groups++;
  out.resize((size_t)num_groups);
  for (int i = 0; i < n; ++i) out[(size_t)g[(size_t)i]].push_back(i);
  return out;
}

static orc_wfa_params ed_params() {  // genotype.rs:88-92: Score scope, BiWFA, edit, default heuristic
  orc_wfa_params p;
  orc_wfa_default_params(&p);
  p.metric = 1; p.span = 0; p.scope = 0; p.memory_mode = 3;
  return p;
}

static double get_dist(const std::string& a, const std::string& b, int64_t* cells, int64_t* n_ed) {  // genotype_cluster.rs:238-248
  const int diff = std::abs((int)a.size() - (int)b.size());
  int dist;
  if (a.size() * b.size() > 10000) dist = diff;
  else {
    static const orc_wfa_params ep = ed_params();
    WfaResult r = wfa_align(ep, (const uint8_t*)a.data(), (int)a.size(), (const uint8_t*)b.data(), (int)b.size());
    if (cells) *cells += r.cells;
    if (n_ed) *n_ed += 1;
    dist = r.score;
  }
  return std::sqrt((double)dist);
}

static int central_read(int n, const std::vector<int>& group, const std::vector<double>& dists) {  // :12-39
  const size_t gs = group.size();
  if (gs <= 2) return group[0];
  std::vector<double> sums(gs, 0.0);
  for (size_t i = 0; i + 1 < gs; ++i)
    for (size_t j = i + 1; j < gs; ++j) {
      const size_t i1 = (size_t)group[i], i2 = (size_t)group[j];
      const size_t mi = (size_t)n * i1 - i1 * (i1 + 3) / 2 + i2 - 1;
      sums[i] += dists[mi]; sums[j] += dists[mi];
    }
  size_t best = 0;
  for (size_t i = 1; i < gs; ++i) if (sums[i] < sums[best]) best = i;  // min_by: first minimum
  return group[best];
}

struct ClusterGt { std::vector<TrSize> gt; std::vector<std::string> alleles; std::vector<int> classification; };

static ClusterGt genotype_cluster(int ploidy, const std::vector<std::string>& trs, int64_t* cells, int64_t* n_aln, int64_t* n_ed) {
  const int n = (int)trs.size();
  std::vector<double> dists;
  for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) dists.push_back(get_dist(trs[(size_t)i], trs[(size_t)j], cells, n_ed));
  auto make_consensus = [&](const std::vector<int>& group, std::string& allele, TrSize& size) {
    std::vector<std::string> seqs;
    for (int i : group) seqs.push_back(trs[(size_t)i]);
    const std::string& backbone = trs[(size_t)central_read(n, group, dists)];
    auto aligns = align_all(backbone, seqs, cells, n_aln);
    allele = repair_consensus(backbone, seqs, aligns);
    int lo = (int)seqs[0].size(), hi = lo;
    for (auto& s : seqs) { lo = std::min(lo, (int)s.size()); hi = std::max(hi, (int)s.size()); }
    size = TrSize{(int)allele.size(), lo, hi};
  };
  ClusterGt out;
  if (ploidy == 1 || n == 1) {
    std::vector<int> group;
    for (int i = 0; i < n; ++i) group.push_back(i);
    std::string allele; TrSize size;
    make_consensus(group, allele, size);
    out.classification.assign((size_t)n, 0);
    if (ploidy == 1) { out.gt = {size}; out.alleles = {allele}; }
    else { out.gt = {size, size}; out.alleles = {allele, allele}; }
    return out;
  }
  std::vector<std::vector<int>> groups = cluster_groups(n, dists);
  std::stable_sort(groups.begin(), groups.end(), [](const std::vector<int>& a, const std::vector<int>& b) { return a.size() < b.size(); });
  std::vector<int> group1 = groups.back(); groups.pop_back();
  std::vector<int> group2 = groups.back(); groups.pop_back();
  std::string allele1, allele2; TrSize size1, size2;
  make_consensus(group1, allele1, size1);
  make_consensus(group2, allele2, size2);
  auto outlier = [](size_t len1, size_t len2, size_t cov1, size_t cov2) {
    const size_t d = len1 > len2 ? len1 - len2 : len2 - len1;
    return d < 100 && std::min(cov1, cov2) * 4 < std::max(cov1, cov2);
  };
  if (outlier(allele1.size(), allele2.size(), group1.size(), group2.size())) {
    group1.clear(); group2.clear();
    for (int i = 0; i < n; ++i) (i % 2 == 0 ? group1 : group2).push_back(i);
    make_consensus(group1, allele1, size1);
    make_consensus(group2, allele2, size2);
    out.classification.resize((size_t)n);
    for (int i = 0; i < n; ++i) out.classification[(size_t)i] = i % 2;
    if (allele1.size() > allele2.size()) {
      for (int& c : out.classification) c = 1 - c;
      out.gt = {size2, size1}; out.alleles = {allele2, allele1};
    } else { out.gt = {size1, size2}; out.alleles = {allele1, allele2}; }
    return out;
  }
  out.classification.assign((size_t)n, 2);
  for (int i : group1) out.classification[(size_t)i] = 0;
  for (int i : group2) out.classification[(size_t)i] = 1;
  for (int i = 0; i < n; ++i) {
    int tie = 1;  // re-initialised per read (genotype_cluster.rs:125): an exact tie always lands on allele 0
    if (out.classification[(size_t)i] == 2) {
      const double d1 = get_dist(trs[(size_t)i], allele1, cells, n_ed), d2 = get_dist(trs[(size_t)i], allele2, cells, n_ed);
      if (d1 < d2) out.classification[(size_t)i] = 0;
      else if (d2 < d1) out.classification[(size_t)i] = 1;
      else { tie = (tie + 1) % 2; out.classification[(size_t)i] = tie; }
    }
  }
  if (allele1.size() > allele2.size()) {
    for (int& c : out.classification) c = 1 - c;
    out.gt = {size2, size1}; out.alleles = {allele2, allele1};
  } else { out.gt = {size1, size2}; out.alleles = {allele1, allele2}; }
  return out;
}

static inline int64_t total_cmp_key(double d) {  // f64::total_cmp
  int64_t b; std::memcpy(&b, &d, 8);
  b ^= (int64_t)((uint64_t)(b >> 63) >> 1);
  return b;
}

}  // namespace orc

using namespace orc;
extern "C" {

int orc_find_spans(const uint8_t* piece, int piece_len, int64_t n_reads, const uint8_t* read_blob, const uint64_t* read_off,
                   const uint32_t* read_len, int mism, int gapo, int gape, double threshold, int32_t* start, int32_t* end,
                   int32_t* used_wfa, int64_t* cells) {
  const orc_wfa_params fp = flank_params(mism, gapo, gape);
  int64_t c = 0;
  for (int64_t i = 0; i < n_reads; ++i) {
    bool used = false;
    SpanOpt s = find_span(piece, piece_len, read_blob + read_off[i], (int)read_len[i], fp, threshold, &used, &c);
    start[i] = s.start; end[i] = s.end;
    if (used_wfa) used_wfa[i] = used;
  }
  if (cells) *cells = c;
  return 0;
}

// kodama-style Ward linkage on a condensed matrix (mutated in place); steps4 = n-1 x (cluster1, cluster2, size), diss = n-1 doubles
// haploid::genotype / diploid::genotype on a length histogram (haploid.rs:3-15, diploid.rs:5-49); gt3 = (size, ci_lo, ci_hi) per allele
int orc_genotype_sizes(int ploidy, const int32_t* sizes, const int32_t* counts, int n, int32_t* gt3) {
  std::vector<int> s(sizes, sizes + n), c(counts, counts + n);
  const std::vector<TrSize> gt = ploidy == 1 ? haploid_genotype(s, c) : diploid_genotype(s, c);
  for (size_t a = 0; a < gt.size(); ++a) { gt3[3 * a] = gt[a].size; gt3[3 * a + 1] = gt[a].ci_lo; gt3[3 * a + 2] = gt[a].ci_hi; }
  return (int)gt.size();
}

int orc_ward_linkage(double* dists, int n, int32_t* steps3, double* diss) {
  std::vector<double> d(dists, dists + (size_t)n * (size_t)(n - 1) / 2);
  std::vector<WardStep> st = ward_linkage(d, n);
  std::copy(d.begin(), d.end(), dists);
  for (size_t i = 0; i < st.size(); ++i) { steps3[3 * i] = st[i].c1; steps3[3 * i + 1] = st[i].c2; steps3[3 * i + 2] = st[i].size; diss[i] = st[i].diss; }
  return (int)st.size();
}

// genotype_cluster::cluster (genotype_cluster.rs:154-227): group id per sequence, groups numbered in the reference's order
int orc_cluster_groups(double* dists, int n, int32_t* group_of) {
  std::vector<double> d(dists, dists + (size_t)n * (size_t)(n - 1) / 2);
  auto groups = cluster_groups(n, d);
  std::copy(d.begin(), d.end(), dists);
  for (size_t g = 0; g < groups.size(); ++g) for (int i : groups[g]) group_of[i] = (int32_t)g;
  return (int)groups.size();
}

int orc_genotype_flank(int n, const uint8_t* tr_blob, const uint64_t* tr_off, const uint32_t* tr_len, const orc_read_meta* meta,
                       int32_t* sizes, int32_t* ci, char* allele0, char* allele1, int allele_cap, int32_t* assignment) {
  std::vector<FlankRead> fr; std::vector<std::string> trs;
  for (int i = 0; i < n; ++i) {
    FlankRead r;
    r.hp = meta && meta->hp_tag ? (int)meta->hp_tag[i] : -1;
    r.start_off = meta && meta->start_offset ? meta->start_offset[i] : 0; r.end_off = meta && meta->end_offset ? meta->end_offset[i] : 0;
    r.mm = meta && meta->mismatch_offsets && meta->mismatch_off ? meta->mismatch_offsets + meta->mismatch_off[i] : nullptr;
    r.n_mm = r.mm ? (size_t)(meta->mismatch_off[i + 1] - meta->mismatch_off[i]) : 0;
    fr.push_back(r);
    trs.emplace_back((const char*)tr_blob + tr_off[i], (size_t)tr_len[i]);
  }
  FlankGt fg; int64_t cells = 0, n_aln = 0;
  if (!genotype_flank(fr, trs, fg, &cells, &n_aln)) return 0;
  for (int a = 0; a < 2; ++a) {
    sizes[a] = fg.gt[a].size; ci[2 * a] = fg.gt[a].ci_lo; ci[2 * a + 1] = fg.gt[a].ci_hi;
    if ((int)fg.alleles[a].size() + 1 > allele_cap) return -1;
    std::memcpy(a == 0 ? allele0 : allele1, fg.alleles[a].c_str(), fg.alleles[a].size() + 1);
  }
  for (int i = 0; i < n; ++i) assignment[i] = fg.assignment[(size_t)i];
  return 1;
}

int orc_locus_analyze(const orc_locus_params* p, const uint8_t* left_flank, int lf_len, const uint8_t* right_flank, int rf_len,
                      const uint8_t* ref_tr, int ref_tr_len,
                      const uint8_t* motif_blob, const uint32_t* motif_off, int n_motifs, int64_t n_reads,
                      const uint8_t* read_blob, const uint64_t* read_off, const uint32_t* read_len, int32_t* span_start,
                      int32_t* span_end, int32_t* n_alleles, char* allele0, char* allele1, int allele_cap, int32_t* gt_size,
                      int32_t* gt_ci, int32_t* n_spanning, int32_t* kept_read, int32_t* classification, int32_t* num_spanning_by_hap,
                      char* mc, char* ms, char* ap, int str_cap, int64_t* stats, const double* read_qual) {
  return orc_locus_analyze_meta(p, left_flank, lf_len, right_flank, rf_len, ref_tr, ref_tr_len, motif_blob, motif_off, n_motifs, n_reads, read_blob, read_off,
                                read_len, span_start, span_end, n_alleles, allele0, allele1, allele_cap, gt_size, gt_ci, n_spanning, kept_read, classification,
                                num_spanning_by_hap, mc, ms, ap, str_cap, stats, read_qual, nullptr);
}

int orc_locus_analyze_meta(const orc_locus_params* p, const uint8_t* left_flank, int lf_len, const uint8_t* right_flank, int rf_len,
                           const uint8_t* ref_tr, int ref_tr_len,
                           const uint8_t* motif_blob, const uint32_t* motif_off, int n_motifs, int64_t n_reads,
                           const uint8_t* read_blob, const uint64_t* read_off, const uint32_t* read_len, int32_t* span_start,
                           int32_t* span_end, int32_t* n_alleles, char* allele0, char* allele1, int allele_cap, int32_t* gt_size,
                           int32_t* gt_ci, int32_t* n_spanning, int32_t* kept_read, int32_t* classification, int32_t* num_spanning_by_hap,
                           char* mc, char* ms, char* ap, int str_cap, int64_t* stats, const double* read_qual, const orc_read_meta* meta) {
  const int F = p->flank_len;
  int64_t wfa_cells = 0, vit_cells = 0, n_flank_wfa = 0, n_cons = 0, bytes_io = 0, n_ed = 0, n_purity = 0;
  *n_alleles = 0; *n_spanning = 0;
  if (allele0) allele0[0] = 0;
  if (allele1) allele1[0] = 0;
  if (mc) mc[0] = 0;
  if (ms) ms[0] = 0;
  if (ap) ap[0] = 0;
  // find_tr_spans
  const uint8_t* lf_piece = left_flank + (lf_len - F);
  const uint8_t* rf_piece = right_flank;
  const double threshold = (double)(size_t)F * p->min_flank_id_frac;
  const orc_wfa_params fp = flank_params(p->mism, p->gapo, p->gape);
  struct RS { int64_t read; int s, e; };
  std::vector<RS> rs;
  for (int64_t i = 0; i < n_reads; ++i) {
    const uint8_t* s = read_blob + read_off[i];
    const int slen = (int)read_len[i];
    bytes_io += slen;
    bool u1 = false, u2 = false;
    SpanOpt l = find_span(lf_piece, F, s, slen, fp, threshold, &u1, &wfa_cells);
    SpanOpt r = find_span(rf_piece, F, s, slen, fp, threshold, &u2, &wfa_cells);
    n_flank_wfa += (int)u1 + (int)u2;
    span_start[i] = span_end[i] = -1;
    if (l.some() && r.some() && l.end <= r.start) { span_start[i] = l.end; span_end[i] = r.start; }
    if (span_start[i] >= 0) rs.push_back({i, span_start[i], span_end[i]});
  }
  auto finish = [&]() {
    if (stats) { stats[0] = wfa_cells; stats[1] = vit_cells; stats[2] = n_flank_wfa; stats[3] = n_cons; stats[4] = bytes_io; stats[5] = n_ed; stats[6] = n_purity; }
    return 0;
  };
  if (rs.empty()) return finish();
  // get_spanning_reads: flank check, stable sort by span length, uniform downsample
  std::vector<RS> kept;
  for (auto& x : rs)
    if (x.s >= F && (int)read_len[x.read] - x.e >= F) kept.push_back(x);
  if (kept.empty()) return finish();
  std::stable_sort(kept.begin(), kept.end(), [](const RS& a, const RS& b) { return (a.e - a.s) < (b.e - b.s); });
  if ((int)kept.size() > p->max_depth) {  // uniform_downsample (tr.rs:172-184)
    const double num = (double)kept.size();
    double fast = 0.0;
    const double step = num / (double)p->max_depth;
    for (int i = 0; i < p->max_depth; ++i) {
      const size_t ind = (size_t)std::floor(fast);
      if (ind != (size_t)i) std::swap(kept[i], kept[ind]);
      fast += step;
    }
    kept.resize((size_t)p->max_depth);
  }
  auto motifs = motifs_from_blob(motif_blob, motif_off, n_motifs);
  for (auto& m : motifs) { m = replace_invalid_bases(m, "ATCGN"); bytes_io += (int64_t)m.size(); }
  Hmm hmm = build_hmm(motifs);
  if (p->min_read_qual < 0.9) {  // filter_impure_trs (tr.rs:37-50, 400-452)
    const size_t max_filter = std::max<size_t>(1, (size_t)std::round(0.1 * (double)kept.size()));
    std::vector<std::pair<double, RS>> pr;
    for (auto& x : kept) {
      const double rq = read_qual ? read_qual[x.read] : std::nan("");
      double purity = 1.0;
      if (!(rq >= 0.9)) {  // None (NaN) or below the cut-off
        const std::string seq = replace_invalid_bases(std::string((const char*)read_blob + read_off[x.read] + x.s, (size_t)(x.e - x.s)), "ATCG");
        if (seq.empty()) purity = std::nan("");
        else { std::vector<int> labels = hmm_label(hmm, seq, &vit_cells); purity = hmm_purity(hmm, motifs, labels, seq, nullptr, nullptr); }
        ++n_purity;
      }
      pr.push_back({purity, x});
    }
    std::stable_sort(pr.begin(), pr.end(), [](const std::pair<double, RS>& a, const std::pair<double, RS>& b) { return total_cmp_key(a.first) < total_cmp_key(b.first); });
    kept.clear();
    size_t num_filtered = 0;
    for (auto& q : pr) {
      if (q.first >= 0.9 || num_filtered >= max_filter) kept.push_back(q.second);
      else ++num_filtered;
    }
    if (kept.empty()) return finish();
  }
  std::vector<std::string> trs;
  for (auto& x : kept) trs.emplace_back((const char*)read_blob + read_off[x.read] + x.s, (size_t)(x.e - x.s));
  SizeGt g;
  if (p->genotyper == 1) {
    ClusterGt cg = genotype_cluster(p->ploidy, trs, &wfa_cells, &n_cons, &n_ed);
    g.gt = cg.gt; g.alleles = cg.alleles; g.classification = cg.classification;
  } else g = genotype_size(p->ploidy, trs, &wfa_cells, &n_cons);
  // flank re-genotyping, only if the alleles have similar length (tr.rs:69-75)
  if (meta && g.gt.size() == 2 && std::abs(g.gt[0].size - g.gt[1].size) <= 10) {
    std::vector<FlankRead> fr;
    for (auto& x : kept) {
      FlankRead r;
      r.hp = meta->hp_tag ? (int)meta->hp_tag[x.read] : -1;
      r.start_off = meta->start_offset ? meta->start_offset[x.read] : 0; r.end_off = meta->end_offset ? meta->end_offset[x.read] : 0;
      r.mm = meta->mismatch_offsets && meta->mismatch_off ? meta->mismatch_offsets + meta->mismatch_off[x.read] : nullptr;
      r.n_mm = r.mm ? (size_t)(meta->mismatch_off[x.read + 1] - meta->mismatch_off[x.read]) : 0;
      fr.push_back(r);
    }
    FlankGt fg;
    if (genotype_flank(fr, trs, fg, &wfa_cells, &n_cons)) { g.gt = fg.gt; g.alleles = fg.alleles; g.classification = fg.assignment; }
  }
  // label_with_hmm
  std::vector<Annotation> ann;
  for (auto& a : g.alleles) { ann.push_back(annotate_allele(hmm, motifs, a, nullptr, &vit_cells)); bytes_io += (int64_t)a.size(); }
  int by_hap[2] = {0, 0};
  for (int c : g.classification) by_hap[c] += 1;
  std::vector<int> order;
  for (size_t i = 0; i < g.gt.size(); ++i) order.push_back((int)i);
  const std::string tr((const char*)ref_tr, (size_t)ref_tr_len);
  if (g.gt.size() != 1 && g.alleles[0] != tr && g.alleles[1] == tr) {  // put reference allele first (tr.rs:95-101)
    std::swap(order[0], order[1]);
    for (int& c : g.classification) c = 1 - c;
  }
  *n_alleles = (int)g.gt.size();
  *n_spanning = (int)kept.size();
  for (size_t i = 0; i < kept.size(); ++i) { kept_read[i] = (int32_t)kept[i].read; classification[i] = g.classification[i]; }
  std::string smc, sms, sap;
  for (size_t oi = 0; oi < order.size(); ++oi) {
    const int a = order[oi];
    char* dst = oi == 0 ? allele0 : allele1;
    if ((int)g.alleles[a].size() + 1 > allele_cap) return -1;
    std::memcpy(dst, g.alleles[a].c_str(), g.alleles[a].size() + 1);
    gt_size[oi] = g.gt[a].size; gt_ci[2 * oi] = g.gt[a].ci_lo; gt_ci[2 * oi + 1] = g.gt[a].ci_hi;
    num_spanning_by_hap[oi] = by_hap[a];
    if (oi) { smc += ","; sms += ","; sap += ","; }
    for (size_t m = 0; m < ann[a].motif_counts.size(); ++m) { if (m) smc += "_"; smc += std::to_string(ann[a].motif_counts[m]); }
    if (ann[a].labels.empty()) sms += ".";
    else for (size_t l = 0; l < ann[a].labels.size(); ++l) {
      if (l) sms += "_";
      auto& sp = ann[a].labels[l];
      sms += std::to_string(sp.motif_index) + "(" + std::to_string(sp.start) + "-" + std::to_string(sp.end) + ")";
    }
    if (std::isnan(ann[a].purity)) sap += ".";
    else { char b[64]; std::snprintf(b, sizeof b, "%.6f", ann[a].purity); sap += b; }
  }
  if ((int)smc.size() + 1 > str_cap || (int)sms.size() + 1 > str_cap || (int)sap.size() + 1 > str_cap) return -1;
  std::memcpy(mc, smc.c_str(), smc.size() + 1);
  std::memcpy(ms, sms.c_str(), sms.size() + 1);
  std::memcpy(ap, sap.c_str(), sap.size() + 1);
  return finish();
}


// analyze_tr for loci [first, first + n) of a batch in the trgt_locus_batch_in layout, on n_threads std::threads pulling chunks of 8 loci
// from a shared counter (the reference runs one rayon task per locus, genotype.rs:179-187).  Only counts what it did: returns the number of loci
// analysed, *alleles_out receives the number of alleles called (a checksum that keeps the work alive).  cpu_baseline helper.
// One text record per locus (NUL-terminated, `rec_stride` bytes apart) for whole-catalog parity sweeps (tests/tools/parity_sweep.py):
//   S:<span_start>,<span_end>;...|A:<allele>,<allele>|K:<kept read indices>|C:<classification>|ALLR:..|SD:..|MC:..|MS:..|AP:..
// `genotyper` (one byte per locus, may be NULL) selects size (0) / cluster (1) like Locus::genotyper.
static int64_t analyze_many_impl(const orc_locus_params* p, int64_t first, int64_t n, const uint8_t* flank_blob, const uint64_t* lf_off,
                               const uint32_t* lf_len, const uint64_t* rf_off, const uint32_t* rf_len, const uint8_t* tr_blob,
                               const uint64_t* tr_off, const uint32_t* tr_len, const uint8_t* motif_blob, const uint32_t* motif_off,
                               const uint32_t* set_motif_begin, const uint64_t* locus_read_begin, const uint8_t* read_blob,
                               const uint64_t* read_off, const uint32_t* read_len, int n_threads, int64_t* alleles_out,
                               const uint8_t* genotyper, const uint8_t* ploidy, char* rec_blob, uint64_t rec_stride, const double* read_qual) {
  if (n_threads < 1) n_threads = 1;
  std::vector<int64_t> alleles((size_t)n_threads * 8, 0), done((size_t)n_threads * 8, 0);
  std::atomic<int64_t> next{0};
  auto work = [&](int t) {
    std::vector<int32_t> ss, se, kept, cls;
    std::vector<char> a0, a1, mc(65536), ms(65536), ap(65536);
    for (;;) {
      const int64_t c0 = next.fetch_add(8);  // dynamic chunks of 8 loci
      if (c0 >= n) break;
    for (int64_t l = first + c0; l < first + std::min(n, c0 + 8); ++l) {
      const uint64_t r0 = locus_read_begin[l], r1 = locus_read_begin[l + 1];
      const int64_t nr = (int64_t)(r1 - r0);
      uint32_t cap = 8;
      for (uint64_t r = r0; r < r1; ++r) cap = std::max(cap, read_len[r] + 8);
      ss.assign((size_t)nr + 1, 0); se.assign((size_t)nr + 1, 0); kept.assign((size_t)nr + 1, 0); cls.assign((size_t)nr + 1, 0);
      a0.assign(cap, 0); a1.assign(cap, 0);
      const uint32_t m0 = set_motif_begin[l], m1 = set_motif_begin[l + 1];
      std::vector<uint32_t> mo(m1 - m0 + 1);
      for (uint32_t m = m0; m <= m1; ++m) mo[m - m0] = motif_off[m] - motif_off[m0];
      int32_t n_alleles = 0, n_sp = 0, gt_size[2], gt_ci[4], by_hap[2];
      int64_t stats[8];
      orc_locus_params pl = *p;
      if (genotyper) pl.genotyper = genotyper[l];
      if (ploidy) pl.ploidy = ploidy[l];
      orc_locus_analyze(&pl, flank_blob + lf_off[l], (int)lf_len[l], flank_blob + rf_off[l], (int)rf_len[l], tr_blob + tr_off[l], (int)tr_len[l],
                        motif_blob + motif_off[m0], mo.data(), (int)(m1 - m0), nr, read_blob, read_off + r0, read_len + r0, ss.data(), se.data(),
                        &n_alleles, a0.data(), a1.data(), (int)cap, gt_size, gt_ci, &n_sp, kept.data(), cls.data(), by_hap, mc.data(), ms.data(),
                        ap.data(), 65536, stats, read_qual ? read_qual + r0 : nullptr);
      alleles[(size_t)t * 8] += n_alleles; done[(size_t)t * 8] += 1;
      if (rec_blob) {
        std::string r = "S:";
        for (int64_t i = 0; i < nr; ++i) { r += std::to_string(ss[(size_t)i]); r += ','; r += std::to_string(se[(size_t)i]); r += ';'; }
        r += "|A:";
        if (n_alleles > 0) r += a0.data();
        if (n_alleles > 1) { r += ','; r += a1.data(); }
        r += "|K:";
        for (int32_t i = 0; i < n_sp; ++i) { r += std::to_string(kept[(size_t)i]); r += ','; }
        r += "|C:";
        for (int32_t i = 0; i < n_sp; ++i) { r += std::to_string(cls[(size_t)i]); r += ','; }
        r += "|ALLR:";
        for (int32_t a = 0; a < n_alleles; ++a) { r += std::to_string(gt_ci[2 * a]); r += '-'; r += std::to_string(gt_ci[2 * a + 1]); r += ','; }
        r += "|SD:";
        for (int32_t a = 0; a < n_alleles; ++a) { r += std::to_string(by_hap[a]); r += ','; }
        if (n_alleles > 0) { r += "|MC:"; r += mc.data(); r += "|MS:"; r += ms.data(); r += "|AP:"; r += ap.data(); }
        char* dst = rec_blob + (uint64_t)(l - first) * rec_stride;
        if (r.size() + 1 > rec_stride) r = "OVERFLOW";
        std::memcpy(dst, r.c_str(), r.size() + 1);
      }
    }
    }
  };
  std::vector<std::thread> th;
  for (int t = 0; t < n_threads; ++t) th.emplace_back(work, t);
  for (auto& x : th) x.join();
  int64_t total = 0, al = 0;
  for (int t = 0; t < n_threads; ++t) { total += done[(size_t)t * 8]; al += alleles[(size_t)t * 8]; }
  if (alleles_out) *alleles_out = al;
  return total;
}

int64_t orc_locus_analyze_many(const orc_locus_params* p, int64_t first, int64_t n, const uint8_t* flank_blob, const uint64_t* lf_off,
                               const uint32_t* lf_len, const uint64_t* rf_off, const uint32_t* rf_len, const uint8_t* tr_blob,
                               const uint64_t* tr_off, const uint32_t* tr_len, const uint8_t* motif_blob, const uint32_t* motif_off,
                               const uint32_t* set_motif_begin, const uint64_t* locus_read_begin, const uint8_t* read_blob,
                               const uint64_t* read_off, const uint32_t* read_len, int n_threads, int64_t* alleles_out) {
  return analyze_many_impl(p, first, n, flank_blob, lf_off, lf_len, rf_off, rf_len, tr_blob, tr_off, tr_len, motif_blob, motif_off, set_motif_begin,
                           locus_read_begin, read_blob, read_off, read_len, n_threads, alleles_out, nullptr, nullptr, nullptr, 0, nullptr);
}

int64_t orc_locus_analyze_records(const orc_locus_params* p, int64_t first, int64_t n, const uint8_t* flank_blob, const uint64_t* lf_off,
                                  const uint32_t* lf_len, const uint64_t* rf_off, const uint32_t* rf_len, const uint8_t* tr_blob,
                                  const uint64_t* tr_off, const uint32_t* tr_len, const uint8_t* motif_blob, const uint32_t* motif_off,
                                  const uint32_t* set_motif_begin, const uint64_t* locus_read_begin, const uint8_t* read_blob,
                                  const uint64_t* read_off, const uint32_t* read_len, int n_threads, const uint8_t* genotyper,
                                  const uint8_t* ploidy, char* rec_blob, uint64_t rec_stride, const double* read_qual) {
  return analyze_many_impl(p, first, n, flank_blob, lf_off, lf_len, rf_off, rf_len, tr_blob, tr_off, tr_len, motif_blob, motif_off, set_motif_begin,
                           locus_read_begin, read_blob, read_off, read_len, n_threads, nullptr, genotyper, ploidy, rec_blob, rec_stride, read_qual);
}

}  // extern "C"
