// trgt_amd/csrc/locus.hip -- trgt_locus_batch: the per-locus genotyper contract for a whole batch.
//
// Replaces analyze_tr (PacificBiosciences/trgt v3.0.0 src/trgt/workflows/tr.rs:24-109) for reads that
// are already clipped (tr.rs:33-34), Genotyper::Size, no HP tags / SNV offsets / methylation:
//   stage A (GPU)   find_tr_spans                          span_locater.rs:32-68   -> spans.hip
//   host glue       get_spanning_reads                     tr.rs:111-184
//                   genotype_size::genotype up to the      genotype_size.rs:6-64, diploid.rs:5-103,
//                   "needs consensus repair" decision      haploid.rs:3-30, consensus.rs:113-154
//   stage B (GPU)   utils::align (BiWFA affine 2,5,1)      utils/align.rs:14-28    -> wfa.hip
//   host glue       repair_consensus, classification,      consensus.rs:5-111, genotype_size.rs:42-61,
//                   reference allele first                 tr.rs:95-101
//   stage C (GPU)   label_with_hmm                         tr.rs:454-492           -> hmm.hip
// genotype_flank::genotype (tr.rs:70-75) returns None for such reads and is not on this path.
// The host glue is integer / byte work of a few microseconds per locus, spread over a persistent pool of host threads;
// moving it onto the device is SURVEY.md 8(f) row 1.
//
// Flow of one call: stage A of the whole batch is enqueued (nothing in it waits for the host), then -- when the reads live in
// HBM -- the device genotyper (locus_gt.hpp: spanning reads, length genotyper, consensus pick, classification) and the copies of
// everything the host needs into pinned memory.  While the GPU works the host builds the HMM tables and initialises outputs.
// Loci the device genotyper hands back (an allele without majority support needs stage B; oversized loci) go through the host
// path below; the HMM batch of all other loci is already running by then.
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <functional>
#include <thread>

#include "hmm_host.hpp"
#include "host_pool.hpp"
#include "locus_gt.hpp"
#include "locus_cluster_dev.hpp"
#include "wfa_host.hpp"

namespace trgt {

int find_spans_device(trgt_hip_ctx* c, const trgt_span_params& p, int64_t n_loci, int64_t n_reads, const uint8_t* d_flank,
                      const uint64_t* d_piece_off, const uint8_t* d_reads, const uint64_t* d_read_off, const uint32_t* d_read_len,
                      const uint32_t* d_read_locus, uint32_t max_read_len, int32_t* d_span_start, int32_t* d_span_end,
                      uint8_t* d_lf_hit, uint8_t* d_rf_hit, const uint32_t* d_heavy_len, uint32_t heavy_tlen_max);

namespace {

struct Seg { const uint8_t* p; uint32_t n; };
inline int cmp_seg(const Seg& a, const Seg& b) {
  const int c = std::memcmp(a.p, b.p, std::min(a.n, b.n));
  return c ? c : (a.n < b.n ? -1 : (a.n > b.n ? 1 : 0));
}
inline bool eq_seg(const Seg& a, const Seg& b) { return a.n == b.n && std::memcmp(a.p, b.p, a.n) == 0; }
inline uint32_t adiff(uint32_t a, uint32_t b) { return a > b ? a - b : b - a; }

struct Repair { int64_t locus; int allele; std::vector<Seg> members; std::string result; };  // rare: no majority sequence

struct LocusWork {                  // plain data: no per-locus heap traffic on the common path
  uint64_t seg_begin = 0, seg_end = 0;  // kept spanning reads = flat segment range, LocusResult.reads order
  int n_gt = 0, n_pick = 0;
  uint32_t size[2] = {0, 0}; uint32_t ci[4] = {0, 0, 0, 0};
  Seg pick[2] = {{nullptr, 0}, {nullptr, 0}};   // consensus::get_consensus picks (point into the segment bytes)
  int repair[2] = {-1, -1};                      // index into the per-thread repair list, -1 = keep the pick
  int repair_thread = 0;
  int cluster = -1;                              // Genotyper::Cluster: index into the call's ClusterLocus list
};

struct Scratch {                    // per host thread, reused across loci
  std::vector<uint32_t> lens, ulen, ucnt, ucount;
  std::vector<Seg> trs, sorted, uniq;
  std::vector<Repair> repairs;
};

// diploid::genotype (diploid.rs:5-103)
void genotype_diploid(const std::vector<uint32_t>& sizes, const std::vector<uint32_t>& counts, LocusWork& w) {
  double best_pen = 0; bool have = false; uint32_t bs = 0, bl = 0;
  for (size_t si = 0; si < sizes.size(); ++si)
    for (size_t li = si; li < sizes.size(); ++li) {
      const uint32_t sa = sizes[si], la = sizes[li];
      const double max_frac = adiff(sa, la) <= 100 ? 0.25 : 0.05;
      double pen = 0.0;
      for (size_t i = 0; i < sizes.size(); ++i) {
        const uint32_t st = sizes[i] != sa ? 10 + 2 * adiff(sa, sizes[i]) : 0, lt = sizes[i] != la ? 10 + 2 * adiff(la, sizes[i]) : 0;
        const double term = (double)std::min(st, lt) + max_frac * (double)std::max(st, lt);
        pen += term * (double)counts[i];
      }
      if (!have || pen < best_pen) { have = true; best_pen = pen; bs = sa; bl = la; }  // stable sort, first minimum
    }
  uint32_t short_size = std::min(bs, bl), long_size = std::max(bs, bl);
  if (short_size != long_size && sizes.size() >= 2) {
    uint64_t coverage = 0;
    size_t top = 0;
    for (size_t i = 0; i < counts.size(); ++i) { coverage += counts[i]; if (counts[i] > counts[top]) top = i; }  // stable desc sort, first
    const double top_frac = (double)counts[top] / (double)coverage;
    const uint32_t range = *std::max_element(sizes.begin(), sizes.end()) - *std::min_element(sizes.begin(), sizes.end());
    if (top_frac > 0.60 && range <= 6) short_size = long_size = sizes[top];
  }
  w.n_gt = 2; w.size[0] = short_size; w.size[1] = long_size;
  w.ci[0] = w.ci[1] = short_size; w.ci[2] = w.ci[3] = long_size;
  for (uint32_t s : sizes) {
    if (adiff(s, short_size) <= adiff(s, long_size)) { w.ci[0] = std::min(w.ci[0], s); w.ci[1] = std::max(w.ci[1], s); }
    else { w.ci[2] = std::min(w.ci[2], s); w.ci[3] = std::max(w.ci[3], s); }
  }
}

// haploid::genotype (haploid.rs:3-30)
void genotype_haploid(const std::vector<uint32_t>& sizes, const std::vector<uint32_t>& counts, LocusWork& w) {
  size_t best = 0; double best_pen = 0;
  for (size_t a = 0; a < sizes.size(); ++a) {
    double pen = 0.0;
    for (size_t i = 0; i < sizes.size(); ++i) {
      const double term = sizes[i] != sizes[a] ? 10.0 + 2.0 * (double)adiff(sizes[a], sizes[i]) : 0.0;
      pen += term * (double)counts[i];
    }
    if (a == 0 || pen < best_pen) { best = a; best_pen = pen; }
  }
  w.n_gt = 1; w.size[0] = sizes[best];
  w.ci[0] = *std::min_element(sizes.begin(), sizes.end()); w.ci[1] = *std::max_element(sizes.begin(), sizes.end());
}

// genotype_size::genotype up to the point where consensus alignments are needed (genotype_size.rs:6-37)
void genotype_size_front(int ploidy, int64_t locus, int thread, LocusWork& w, Scratch& sc) {
  auto& lens = sc.lens; auto& ulen = sc.ulen; auto& ucnt = sc.ucnt;
  lens.clear(); ulen.clear(); ucnt.clear();
  for (auto& s : sc.trs) lens.push_back(s.n);
  std::sort(lens.begin(), lens.end());
  for (size_t i = 0; i < lens.size();) { size_t j = i; while (j < lens.size() && lens[j] == lens[i]) ++j; ulen.push_back(lens[i]); ucnt.push_back((uint32_t)(j - i)); i = j; }
  if (ploidy == 1) genotype_haploid(ulen, ucnt, w); else genotype_diploid(ulen, ucnt, w);
  // get_seq_hist: unique sequences in byte-lexicographic order
  auto& sorted = sc.sorted; auto& uniq = sc.uniq; auto& ucount = sc.ucount;
  sorted = sc.trs; uniq.clear(); ucount.clear();
  std::sort(sorted.begin(), sorted.end(), [](const Seg& a, const Seg& b) { return cmp_seg(a, b) < 0; });
  for (size_t i = 0; i < sorted.size();) {
    size_t j = i;
    while (j < sorted.size() && eq_seg(sorted[j], sorted[i])) ++j;
    uniq.push_back(sorted[i]); ucount.push_back((uint32_t)(j - i));
    i = j;
  }
  auto closest = [&](uint32_t target) { uint32_t c = uniq[0].n; for (auto& s : uniq) if (adiff(c, target) > adiff(s.n, target)) c = s.n; return c; };
  auto most_frequent = [&](uint32_t len) { int best = -1; for (size_t i = 0; i < uniq.size(); ++i) if (uniq[i].n == len && (best < 0 || ucount[i] >= ucount[best])) best = (int)i; return best; };
  int pick[2] = {most_frequent(closest(w.size[0])), -1};
  w.n_pick = 1;
  if (w.n_gt != 1 && w.size[0] != w.size[1]) { pick[1] = most_frequent(closest(w.size[1])); w.n_pick = 2; }
  w.repair_thread = thread;
  for (int a = 0; a < w.n_pick; ++a) {
    w.pick[a] = uniq[pick[a]];
    // split(): members of this allele's group
    uint64_t coverage = 0, ref_count = 0;
    for (size_t i = 0; i < uniq.size(); ++i) {
      bool in;
      if (w.n_gt == 1) in = true;
      else {
        const uint32_t d1 = adiff(uniq[i].n, w.size[0]), d2 = adiff(uniq[i].n, w.size[1]);
        in = a == 0 ? d1 <= d2 : d2 < d1;
      }
      if (!in) continue;
      coverage += ucount[i];
      if ((int)i == pick[a]) ref_count = ucount[i];
    }
    if (!(2 * ref_count >= coverage)) {
      Repair r; r.locus = locus; r.allele = a;
      for (size_t i = 0; i < uniq.size(); ++i) {
        bool in;
        if (w.n_gt == 1) in = true;
        else { const uint32_t d1 = adiff(uniq[i].n, w.size[0]), d2 = adiff(uniq[i].n, w.size[1]); in = a == 0 ? d1 <= d2 : d2 < d1; }
        if (in) r.members.push_back(uniq[i]);
      }
      w.repair[a] = (int)sc.repairs.size();
      sc.repairs.push_back(std::move(r));
    }
  }
}

// ---- genotype_flank::genotype (src/trgt/genotype/genotype_flank.rs:9-290; tr.rs:69-75): a locus with two alleles at most 10 bases apart
// is genotyped again from the reads' haplotype tags or, without them, from heterozygous SNVs of the flanks.  Host work on the per-read
// fields of trgt_locus_batch_in (hp_tag, start_offset, end_offset, mismatch offsets); a locus whose reads carry neither tags nor
// mismatches leaves here in a few comparisons.  Reads are the kept spanning reads in LocusResult.reads order, given by their index
// into the batch's read arrays.
struct FlankMeta { const int16_t* hp; const int32_t* so; const int32_t* eo; const int32_t* mm; const uint64_t* mm_off; };
struct FlankSplit { std::vector<uint32_t> group[2]; std::vector<int8_t> assignment; };  // positions in the read list

bool flank_split(const FlankMeta& M, const uint32_t* reads, size_t n, FlankSplit& out) {
  out.group[0].clear(); out.group[1].clear(); out.assignment.clear();
  if (n == 0) return false;
  // get_trs_with_hp (:43-76): tagged reads go to their haplotype, the others alternate; needs 70 % tagged and both haplotypes
  if (M.hp) {
    size_t untagged = 0; int tie = 1;
    for (size_t i = 0; i < n; ++i) {
      const int h = M.hp[reads[i]];
      int a;
      if (h == 1) a = 0; else if (h == 2) a = 1; else { tie = (tie + 1) % 2; a = tie; ++untagged; }
      out.assignment.push_back((int8_t)a); out.group[a].push_back((uint32_t)i);
    }
    if (!out.group[0].empty() && !out.group[1].empty() && (double)(n - untagged) / (double)n >= 0.7) return true;
    out.group[0].clear(); out.group[1].clear(); out.assignment.clear();
  }
  // get_trs_with_clustering (:78-138)
  auto mm_begin = [&](size_t i) { return M.mm && M.mm_off ? M.mm + M.mm_off[reads[i]] : nullptr; };
  auto mm_count = [&](size_t i) { return M.mm && M.mm_off ? (size_t)(M.mm_off[reads[i] + 1] - M.mm_off[reads[i]]) : (size_t)0; };
  auto s_off = [&](size_t i) { return M.so ? M.so[reads[i]] : 0; };
  auto e_off = [&](size_t i) { return M.eo ? M.eo[reads[i]] : 0; };
  size_t any_mm = 0;
  for (size_t i = 0; i < n; ++i) any_mm += mm_count(i);
  if (any_mm == 0) return false;  // no SNV can be called: every profile is empty, one candidate genotype, "homozygous"
  const size_t skip = (size_t)std::round((double)n * (1.0 - 0.85));  // get_analysis_region (:206-226)
  if (skip >= n) return false;
  std::vector<int32_t> so(n), eo(n);
  for (size_t i = 0; i < n; ++i) { so[i] = s_off(i); eo[i] = e_off(i); }
  std::sort(so.begin(), so.end()); std::sort(eo.begin(), eo.end());
  const int32_t reg0 = so[n - 1 - skip], reg1 = eo[skip];
  std::vector<int32_t> seen;  // call_snvs (:271-286): offsets inside the region carried by at least 20 % of the reads
  for (size_t i = 0; i < n; ++i) { const int32_t* m = mm_begin(i); for (size_t k = 0; k < mm_count(i); ++k) if (reg0 <= m[k] && m[k] <= reg1) seen.push_back(m[k]); }
  std::sort(seen.begin(), seen.end());
  std::vector<int32_t> snvs;
  for (size_t a = 0; a < seen.size();) { size_t b = a; while (b < seen.size() && seen[b] == seen[a]) ++b; if ((double)(b - a) / (double)n >= 0.20) snvs.push_back(seen[a]); a = b; }
  const size_t ns = snvs.size();
  // profiles (:250-269), one row of ns cells per read: 0 None, 1 Some(false), 2 Some(true) (the derived order of Option<bool>)
  std::vector<uint8_t> prof(n * ns);
  for (size_t i = 0; i < n; ++i) {
    const int32_t* m = mm_begin(i); const size_t nm = mm_count(i);
    for (size_t k = 0; k < ns; ++k) prof[i * ns + k] = (snvs[k] < s_off(i) || snvs[k] > e_off(i)) ? 0 : (std::binary_search(m, m + nm, snvs[k]) ? 2 : 1);
  }
  auto row = [&](size_t i) { return prof.data() + i * ns; };
  // candidate genotypes (:228-248): pairs (i <= j) of the distinct fully observed profiles, needs 40 % of the reads fully observed
  std::vector<size_t> full;
  for (size_t i = 0; i < n; ++i) { bool all = true; for (size_t k = 0; k < ns; ++k) all = all && row(i)[k] != 0; if (all) full.push_back(i); }
  if ((double)full.size() / (double)n < 0.40) return false;
  std::sort(full.begin(), full.end(), [&](size_t a, size_t b) { return std::lexicographical_compare(row(a), row(a) + ns, row(b), row(b) + ns); });
  full.erase(std::unique(full.begin(), full.end(), [&](size_t a, size_t b) { return std::equal(row(a), row(a) + ns, row(b)); }), full.end());
  const size_t nh = full.size();
  if (nh * (nh + 1) / 2 <= 1) return false;
  const double ln_match = std::log(0.9), ln_mis = std::log(1.0 - 0.9), ln2 = std::log(2.0);
  auto eval = [&](size_t i, size_t h) { double t = 0.0; for (size_t k = 0; k < ns; ++k) if (row(i)[k]) t += row(i)[k] == row(h)[k] ? ln_match : ln_mis; return t; };
  size_t top1 = 0, top2 = 0; double top_ll = 0.0; bool first = true;
  for (size_t a = 0; a < nh; ++a)
    for (size_t b = a; b < nh; ++b) {
      double ll = 0.0;
      for (size_t i = 0; i < n; ++i) {
        const double t1 = eval(i, full[a]), t2 = eval(i, full[b]), mx = std::max(t1, t2);
        ll += (mx + std::log(std::exp(t1 - mx) + std::exp(t2 - mx))) - ln2;
      }
      if (first || ll >= top_ll) { first = false; top1 = full[a]; top2 = full[b]; top_ll = ll; }  // max_by: the last maximum
    }
  if (std::equal(row(top1), row(top1) + ns, row(top2))) return false;
  auto agree = [&](size_t i, size_t h) { size_t d = 0; for (size_t k = 0; k < ns; ++k) d += row(i)[k] && row(i)[k] == row(h)[k]; return d; };
  int tie = 1;
  for (size_t i = 0; i < n; ++i) {
    const size_t d1 = agree(i, top1), d2 = agree(i, top2);
    if (d1 < d2) { out.assignment.push_back(0); out.group[0].push_back((uint32_t)i); }
    else if (d1 > d2) { out.assignment.push_back(1); out.group[1].push_back((uint32_t)i); }
    else { tie = (tie + 1) % 2; out.assignment.push_back((int8_t)tie); out.group[0].push_back((uint32_t)i); out.group[1].push_back((uint32_t)i); }
  }
  return true;
}

// utils::math::median (src/utils/math.rs:73-98): the middle value, or -- even sizes -- (a + b) as i32, then f32 / 2.0 (quickselect in the
// reference; any selection gives the same values).  data is reordered; empty input has no median (the callers never pass one).
float median_f32(std::vector<int32_t>& lens) {
  std::sort(lens.begin(), lens.end());
  return lens.size() % 2 ? (float)lens[lens.size() / 2] : (float)(lens[lens.size() / 2 - 1] + lens[lens.size() / 2]) / 2.0f;
}

// simple_consensus (:147-170) over the segments of a group: the most frequent sequence (among equals: length closest to the f32 median of
// the lengths, truncated; among those the smallest sequence) and its relative frequency; false for an empty group
bool flank_simple_consensus(const std::vector<Seg>& seqs, Seg& best, double& freq) {
  if (seqs.empty()) return false;
  std::vector<int32_t> lens;
  for (auto& q : seqs) lens.push_back((int32_t)q.n);
  const float med = median_f32(lens);
  const size_t median_len = (size_t)med;
  std::vector<Seg> sorted = seqs;
  std::sort(sorted.begin(), sorted.end(), [](const Seg& a, const Seg& b) { return cmp_seg(a, b) < 0; });
  size_t top = 0;
  for (size_t i = 0; i < sorted.size();) { size_t j = i; while (j < sorted.size() && eq_seg(sorted[j], sorted[i])) ++j; top = std::max(top, j - i); i = j; }
  bool have = false; size_t best_delta = 0;
  for (size_t i = 0; i < sorted.size();) {
    size_t j = i; while (j < sorted.size() && eq_seg(sorted[j], sorted[i])) ++j;
    if (j - i == top) {
      const size_t d = sorted[i].n > median_len ? sorted[i].n - median_len : median_len - sorted[i].n;
      if (!have || d < best_delta) { have = true; best = sorted[i]; best_delta = d; }  // min_by_key: the first minimum, in sequence order
    }
    i = j;
  }
  freq = (double)top / (double)seqs.size();
  return true;
}

// Developer switches of tools/unpinned_sensitivity.py: the two readings of WFA2-lib's BiWFA that no reference test pins, applied
// to the consensus alignments and edit distances of the locus path
inline void sens_apply(const trgt_hip_ctx* c, trgt_wfa_params& wp) {
  if (c->knobs.sens_bialign_min_len >= 0) wp.bialign_min_length = c->knobs.sens_bialign_min_len;
  if (c->knobs.sens_cons_unidir && wp.scope == 1) wp.memory_mode = 0;
}

#include "consensus_vote.hpp"
static_assert(sizeof(vote::Group) == sizeof(gt::RGroup) && offsetof(vote::Group, bb_off) == offsetof(gt::RGroup, bb_off) &&
              offsetof(vote::Group, out_off) == offsetof(gt::RGroup, out_off) && offsetof(vote::Group, scratch_off) == offsetof(gt::RGroup, scratch_off) &&
              offsetof(vote::Group, out_cap) == offsetof(gt::RGroup, out_cap), "gt::RGroup mirrors vote::Group");

// make_consensus / repair_consensus (consensus.rs:5-111) for a batch of groups: the members of group g are jobs [first[g], first[g + 1])
// of ONE alignment batch (BiWFA, gap-affine 2,5,1, default heuristic: THREAD_WFA_CONSENSUS, genotype.rs:82-86), each against the
// group's backbone (the pattern of its jobs).  The run-length CIGARs stay in HBM; the column voting runs there too
// (consensus_vote_kernel) and only the consensus sequences come back.  results[g] = repaired sequence of group g.
int consensus_repair_batch(trgt_hip_ctx* c, int64_t n_jobs, const uint8_t* seqs, const uint64_t* po, const uint32_t* pl, const uint64_t* to,
                           const uint32_t* tl, const std::vector<size_t>& first, std::vector<std::string>& results,
                           const std::function<int()>* while_running = nullptr) {
  const size_t n_groups = first.empty() ? 0 : first.size() - 1;
  results.assign(n_groups, std::string());
  if (n_jobs == 0 || n_groups == 0) { if (while_running && *while_running) return (*while_running)(); return TRGT_OK; }
  trgt_wfa_params wp;
  trgt_wfa_default_params(&wp);
  wp.metric = 3; wp.mismatch = 2; wp.gap_open1 = 5; wp.gap_ext1 = 1; wp.span = 0; wp.scope = 1; wp.memory_mode = 3; sens_apply(c, wp);
  const bool tl_on = c->knobs.timeline;
  const int64_t tl0 = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
#define RTL(name) do { if (tl_on) fprintf(stderr, "[tl]     repair %-20s +%7.2f ms\n", name, (double)(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() - tl0) / 1e6); } while (0)
  WfaOnDevice dev;
  int rc = wfa_batch_impl(c, &wp, n_jobs, seqs, po, pl, to, tl, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                          nullptr, nullptr, while_running, &dev);
  if (rc) return rc;
  RTL("alignments done");
  std::vector<vote::Group> groups(n_groups);
  uint64_t out_total = 0, scratch_words = 0;
  for (size_t g = 0; g < n_groups; ++g) {
    vote::Group& G = groups[g];
    const size_t j0 = first[g], j1 = first[g + 1];
    G.job_first = (uint32_t)j0; G.n_members = (uint32_t)(j1 - j0);
    // (the votes are 16-bit counters packed in pairs: a larger group would carry from one counter into its neighbour)
    if (j1 - j0 > 65535) return fail(c, TRGT_ERR_UNSUPPORTED, "consensus: group %zu has %zu members (at most 65535: 3 * max_depth reads per locus)", g, j1 - j0);
    G.bb_len = j1 > j0 ? pl[j0] : 0; G.bb_off = j1 > j0 ? po[j0] : 0;
    uint64_t member_bytes = 0;
    for (size_t j = j0; j < j1; ++j) member_bytes += tl[j];
    // at most one base per backbone position plus the insertions taken, each of which is a piece of some member
    G.out_cap = (uint32_t)std::min<uint64_t>((uint64_t)G.bb_len + member_bytes + 16, 0xFFFFFFF0ull);
    G.out_off = out_total; out_total += ((uint64_t)G.out_cap + 15) & ~15ull;
    G.scratch_off = scratch_words;
    scratch_words += (G.bb_len + 1 <= (uint32_t)vote::VOTE_LDS_POS + 1 ? 0 : 3 * ((uint64_t)G.bb_len + 1)) + 3 * (uint64_t)G.n_members;
  }
  RTL("groups built");
  void *d_groups = nullptr, *d_scratch = nullptr, *d_out = nullptr, *d_len = nullptr;
  if ((rc = dev_get(c, S_VOTE_GROUPS, n_groups * sizeof(vote::Group), &d_groups)) || (rc = dev_get(c, S_VOTE_SCRATCH, (size_t)scratch_words * 4 + 16, &d_scratch)) ||
      (rc = dev_get(c, S_VOTE_OUT, (size_t)out_total + 16, &d_out)) || (rc = dev_get(c, S_VOTE_LEN, n_groups * 4, &d_len)))
    return rc;
  if ((rc = h2d_small(c, d_groups, groups.data(), n_groups * sizeof(vote::Group), c->stream, S_VOTE_GROUPS))) return rc;
  vote::VoteArgs va{(const vote::Group*)d_groups, (uint32_t)n_groups, nullptr, dev.seqs, dev.jobs, dev.cigar, dev.cigar_len, (uint32_t*)d_scratch, (uint8_t*)d_out, (uint32_t*)d_len};
  hipLaunchKernelGGL(vote::consensus_vote_kernel, dim3((unsigned)n_groups), dim3(vote::VOTE_THREADS), 0, c->stream, va);
  TRGT_HIP_TRY(c, hipGetLastError());
  RTL("vote launched");
  std::vector<uint32_t> lens(n_groups);
  std::vector<uint8_t> bytes((size_t)out_total);
  { const int d2h_rc = trgt::d2h(c, lens.data(), d_len, n_groups * 4, c->stream); if (d2h_rc) return d2h_rc; }
  { const int d2h_rc = trgt::d2h(c, bytes.data(), d_out, (size_t)out_total, c->stream); if (d2h_rc) return d2h_rc; }
  RTL("d2h enqueued");
  TRGT_HIP_TRY(c, trgt::stream_wait(c, c->stream));
  if (tl_on) fprintf(stderr, "[tl]     repair outputs: %zu groups, %.2f MB of output slots, %.2f MB of scratch\n", n_groups, (double)out_total / 1e6, (double)scratch_words * 4 / 1e6);
  RTL("synced");
  for (size_t g = 0; g < n_groups; ++g) {
    if (lens[g] == 0xFFFFFFFFu) return fail(c, TRGT_ERR_UNSUPPORTED, "consensus: repaired sequence of group %zu longer than %u bases", g, groups[g].out_cap);
    results[g].assign((const char*)bytes.data() + groups[g].out_off, lens[g]);
  }
  return TRGT_OK;
}

#include "locus_cluster.hpp"

// developer aid (TRGT_REPAIR_CHECK=1): the job list the genotyper wrote for the device-side repair, checked before the alignment kernel reads it
__global__ void repair_check_kernel(const gt::RepairBufs rp, uint64_t read_bytes) {
  const uint32_t nj = rp.counts[gt::RC_JOBS], ng = rp.counts[gt::RC_GROUPS];
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < nj; j += gridDim.x * blockDim.x) {
    const JobDev jd = rp.jobs[j];
    const bool bad = jd.pat_len > rp.max_seg || jd.txt_len > rp.max_seg || jd.pat_off + jd.pat_len > read_bytes || jd.txt_off + jd.txt_len > read_bytes ||
                     jd.cigar_off + jd.pat_len + jd.txt_len + 1 > rp.cap_cigar || jd.out_index != j;
    if (bad) { atomicAdd(rp.counts + 10, 1u); rp.counts[11] = j; rp.counts[12] = jd.pat_len; rp.counts[13] = jd.txt_len; rp.counts[14] = jd.out_index; rp.counts[15] = (uint32_t)jd.cigar_off; }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && (nj > rp.cap_jobs || ng > rp.cap_groups)) atomicAdd(rp.counts + 10, 1000000u);
}

static uint32_t* g_trace_dump = nullptr;
static uint32_t* repair_trace_buf() {
  if (!TRGT_DEV_ENV("TRGT_REPAIR_TRACE")) return nullptr;
  if (!g_trace_dump) (void)hipHostMalloc((void**)&g_trace_dump, 4096, hipHostMallocDefault);
  return g_trace_dump;
}
// developer aid (TRGT_REPAIR_TRACE=1): the repair counters and the first jobs as the GPU sees them right behind the genotyper, written to
// pinned host memory (readable after a later device fault)
__global__ void repair_trace_kernel(const gt::RepairBufs rp, uint32_t* host) {
  if (threadIdx.x < gt::RC_WORDS) host[threadIdx.x] = __hip_atomic_load(rp.counts + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x < 8) { const JobDev jd = rp.jobs[threadIdx.x]; uint32_t* h = host + 16 + 8 * threadIdx.x;
    h[0] = (uint32_t)jd.pat_off; h[1] = jd.pat_len; h[2] = (uint32_t)jd.txt_off; h[3] = jd.txt_len; h[4] = (uint32_t)jd.cigar_off; h[5] = jd.out_index; h[6] = (uint32_t)(jd.pat_off >> 32); h[7] = (uint32_t)(jd.txt_off >> 32); }
  __threadfence_system();
}
// counters that kernels bump with atomics are zeroed the same way (device-scope atomic stores), not with hipMemsetAsync: see DESIGN.md
__global__ void zero_words_kernel(uint32_t* p, uint32_t n) { for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) atomicExch(p + i, 0u); }

struct GatherArgs { const uint8_t* reads; const uint64_t* src_off; const uint64_t* dst_off; const uint32_t* len; uint64_t n; uint8_t* out; };
__global__ void gather_segments_kernel(const GatherArgs a) {  // one wavefront per segment
  const uint64_t s = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= a.n) return;
  const uint8_t* __restrict__ src = a.reads + a.src_off[s];
  uint8_t* __restrict__ dst = a.out + a.dst_off[s];
  for (uint32_t i = threadIdx.x & 63; i < a.len[s]; i += 64) dst[i] = src[i];
}

inline int64_t now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

HostPool* host_pool(trgt_hip_ctx* c, int threads) {
  if (c->host_pool && c->host_pool_threads != threads) { delete static_cast<HostPool*>(c->host_pool); c->host_pool = nullptr; }
  if (!c->host_pool) { c->host_pool = new HostPool(threads); c->host_pool_threads = threads; }
  return static_cast<HostPool*>(c->host_pool);
}

// Exclusive prefix over the allele lengths (one workgroup) and the packing of the alleles into one dense buffer for the D2H copy
// (... and, riding along: the count blocks of the device-side repair and of the cluster genotyper copied into the result slab -- null
//  source: zeros -- which were a D2D copy or a fill each)
struct CountCopy { const uint32_t* src1; uint32_t* dst1; uint32_t n1; const uint32_t* src2; uint32_t* dst2; uint32_t n2; };
__global__ void __launch_bounds__(1024) allele_prefix_kernel(const uint32_t* __restrict__ len, uint64_t* __restrict__ off, int64_t n, const CountCopy cc) {
  __shared__ uint64_t part[1024];
  const int t = threadIdx.x;
  if ((uint32_t)t < cc.n1) cc.dst1[t] = cc.src1 ? __hip_atomic_load(cc.src1 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  if ((uint32_t)t < cc.n2) cc.dst2[t] = cc.src2 ? __hip_atomic_load(cc.src2 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  const int64_t per = (n + 1023) / 1024, b = t * per, e = b + per < n ? b + per : n;
  uint64_t sum = 0;
  for (int64_t i = b; i < e; ++i) sum += len[i];
  part[t] = sum;
  __syncthreads();
  if (t == 0) { uint64_t run = 0; for (int i = 0; i < 1024; ++i) { const uint64_t v = part[i]; part[i] = run; run += v; } off[n] = run; }
  __syncthreads();
  uint64_t run = part[t];
  for (int64_t i = b; i < e; ++i) { off[i] = run; run += len[i]; }
}
__global__ void allele_pack_kernel(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ src_off, const uint32_t* __restrict__ len,
                                   const uint64_t* __restrict__ dst_off, uint8_t* __restrict__ packed, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const uint8_t* s = blob + src_off[i];
  uint8_t* d = packed + dst_off[i];
  for (uint32_t b = threadIdx.x & 63; b < len[i]; b += 64) d[b] = s[b];
}

}  // namespace
}  // namespace trgt

using namespace trgt;

extern "C" int32_t trgt_median_i32(const int32_t* data, int64_t n, float* out) {  // include/trgt_hip.h: "per-read helpers"
  if (n < 0 || (n > 0 && !data) || !out) return TRGT_ERR_INVALID;
  if (n == 0) return 0;
  try { std::vector<int32_t> v(data, data + n); *out = median_f32(v); return 1; } catch (const std::exception&) { return TRGT_ERR_NOMEM; }
}

extern "C" void trgt_locus_default_params(trgt_locus_params* p) {  // cli.rs:271-344
  if (!p) return;
  std::memset(p, 0, sizeof *p);
  p->flank_len = 250; p->min_flank_id_frac = 0.7; p->max_depth = 250; p->mism = 2; p->gapo = 5; p->gape = 1; p->host_threads = 0;
  p->min_read_qual = 0.98;
}

// Copies of submitted batches that have not been issued yet.  Measured on MI355X / ROCm 7.2: host-to-device copies of all streams
// drain through one queue in issue order, so a 360 MB upload issued BEFORE a call's own small uploads (offset tables, job lists)
// holds each of them up until it is through, and nothing overlaps (copy || call = copy + call).  Issued right BEHIND the call's
// tables it runs next to stage A, which uploads nothing, and is through before the later stages upload their job lists.
// All contexts of a process send their bulk uploads to a device through ONE stream (TRGT_SHARED_UPLOAD_STREAM=0: each through its own copy
// stream).  Every stream with copies in flight may get a copy engine of its own; four engines pulling at once keep the link's read queue
// four times as deep, and everything latency-bound that reads host memory -- queue packets, kernel arguments, the small tables kernels
// fetch from pinned memory -- waits behind it (measured: see DESIGN.md).
static std::mutex g_upload_mutex[16];
static hipStream_t g_upload_stream[16];
static hipStream_t bulk_upload_stream(trgt_hip_ctx* c) {
  static const bool shared = [] { const char* e = TRGT_DEV_ENV("TRGT_SHARED_UPLOAD_STREAM"); return !(e && *e == '0'); }();
  if (!shared) return c->stream_copy;
  const size_t d = (size_t)c->device % 16;
  if (!g_upload_stream[d] && hipStreamCreateWithFlags(&g_upload_stream[d], hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return c->stream_copy; }
  return g_upload_stream[d];
}
static int issue_pending_uploads(trgt_hip_ctx* c) {
  bool any = false;
  for (auto& st : c->staged) any = any || (st.in_use && st.copy_pending);
  if (!any) return TRGT_OK;
  std::lock_guard<std::mutex> upload_lock(g_upload_mutex[(size_t)c->device % 16]);  // (one batch's pieces stay together in the queue)
  const hipStream_t up = bulk_upload_stream(c);
  for (auto& st : c->staged) {
    if (!st.in_use || !st.copy_pending) continue;
    st.copy_pending = false;
    if (st.d_reads) {
      // in pieces: one copy command of hundreds of MB holds the engine until it is through, and the small copies of the other contexts
      // of a pool queue behind it (TRGT_UPLOAD_CHUNK_MB, default 32; 0 = one command)
      static const size_t piece = [] { const char* e = TRGT_DEV_ENV("TRGT_UPLOAD_CHUNK_MB"); const long v = e && *e ? atol(e) : 32; return v > 0 ? (size_t)v << 20 : (size_t)0; }();
      const size_t total = (size_t)st.read_bytes;
      // TRGT_UPLOAD_KERNEL=<workgroups>: the read bytes are pulled by a copy KERNEL of that many workgroups instead of the copy engine
      // (pinned sources only).  The engine keeps the link's read queue full, and every dispatch of every context -- its queue packet and
      // its arguments are fetched from host memory -- waits behind that queue; a kernel with a bounded number of loads in flight leaves
      // the queue short.  0 = the copy engine.
      static const int kernel_wgs = [] { const char* e = TRGT_DEV_ENV("TRGT_UPLOAD_KERNEL"); return e && *e ? atoi(e) : 0; }();
      if (kernel_wgs > 0 && is_pinned_host_ptr(st.in->read_blob)) {
        hipLaunchKernelGGL(h2d_copy_kernel, dim3((unsigned)kernel_wgs), dim3(256), 0, up, const_cast<uint8_t*>(st.d_reads), st.in->read_blob, total);
        TRGT_HIP_TRY(c, hipGetLastError());
      } else
      for (size_t o = 0; o < total; o += piece ? piece : total) {
        const size_t n = piece ? std::min(piece, total - o) : total;
        TRGT_HIP_TRY(c, hipMemcpyAsync(const_cast<uint8_t*>(st.d_reads) + o, st.in->read_blob + o, n, hipMemcpyHostToDevice, up));
      }
    }
    if (st.d_flank) TRGT_HIP_TRY(c, hipMemcpyAsync(const_cast<uint8_t*>(st.d_flank), st.in->flank_blob, (size_t)st.flank_bytes, hipMemcpyHostToDevice, up));
    TRGT_HIP_TRY(c, hipEventRecord(st.ready, up));
  }
  return TRGT_OK;
}

static std::mutex g_stage_a_mutex[16];

// TRGT_READS_BAM4 -> ASCII: byte i of the packed blob becomes bytes 2i, 2i+1 of the expanded one (so a read that starts at packed byte
// o starts at expanded byte 2o).  A thread turns 16 packed bytes into 32: one 16-byte load, two 16-byte stores.
__device__ __forceinline__ uint32_t bam4_pair_to_ascii(uint32_t b) {  // one packed byte -> two ASCII bytes (first base in the low byte)
  const uint64_t lo = 0x565352474d43413dull /* "=ACMGRSV" */, hi = 0x4e42444b48595754ull /* "TWYHKDBN" */;
  const uint32_t a = b >> 4, z = b & 15u;
  const uint32_t ca = (uint32_t)(((a & 8u) ? hi : lo) >> (8 * (a & 7u))) & 0xFFu, cz = (uint32_t)(((z & 8u) ? hi : lo) >> (8 * (z & 7u))) & 0xFFu;
  return ca | (cz << 8);
}
__global__ void __launch_bounds__(256) expand_bam4_kernel(const uint8_t* __restrict__ packed, uint8_t* __restrict__ out, uint64_t n_bytes) {
  const uint64_t n16 = n_bytes / 16;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint4 v = ((const uint4*)packed)[i];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t o[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      o[2 * k] = bam4_pair_to_ascii(w[k] & 0xFFu) | (bam4_pair_to_ascii((w[k] >> 8) & 0xFFu) << 16);
      o[2 * k + 1] = bam4_pair_to_ascii((w[k] >> 16) & 0xFFu) | (bam4_pair_to_ascii(w[k] >> 24) << 16);
    }
    ((uint4*)out)[2 * i] = make_uint4(o[0], o[1], o[2], o[3]);
    ((uint4*)out)[2 * i + 1] = make_uint4(o[4], o[5], o[6], o[7]);
  }
  if (blockIdx.x == 0) {  // the tail of fewer than 16 bytes
    const uint64_t i = n16 * 16 + threadIdx.x;
    if (i < n_bytes) { const uint32_t t = bam4_pair_to_ascii(packed[i]); out[2 * i] = (uint8_t)t; out[2 * i + 1] = (uint8_t)(t >> 8); }
  }
}

// staged_reads / staged_flank: device copies of in->read_blob / in->flank_blob made ahead of time by trgt_locus_batch_submit (the
// caller's pointers stay what the host glue reads); `ready`: the event behind those copies.
static int locus_batch_run(trgt_hip_ctx* c, const trgt_locus_params* p, const trgt_locus_batch_in* in, trgt_locus_batch_out* out,
                           const uint8_t* staged_reads, const uint8_t* staged_flank, hipEvent_t ready) {
  if (!c) return TRGT_ERR_INVALID;
  if (!p || !in || !out) return fail(c, TRGT_ERR_INVALID, "trgt_locus_batch: null argument");
  const int64_t nl = in->n_loci;
  if (nl < 0) return fail(c, TRGT_ERR_INVALID, "trgt_locus_batch: negative n_loci");
  if (nl == 0) return TRGT_OK;
  if (!in->flank_blob || !in->lf_off || !in->lf_len || !in->rf_off || !in->rf_len || !in->tr_blob || !in->tr_off || !in->tr_len ||
      !in->motif_blob || !in->motif_off || !in->set_motif_begin || !in->ploidy || !in->locus_read_begin || !in->read_blob ||
      !in->read_off || !in->read_len || !out->span_start || !out->span_end || !out->n_alleles || !out->allele_blob ||
      !out->allele_off || !out->allele_cap || !out->allele_len || !out->ci || !out->num_spanning || !out->classification ||
      !out->read_rank || !out->spans3 || !out->span_off || !out->n_spans || !out->motif_counts || !out->count_off || !out->purity)
    return fail(c, TRGT_ERR_INVALID, "trgt_locus_batch: null field");
  TRGT_HIP_TRY(c, hipSetDevice(c->device));
  struct ZeroGuard { trgt_hip_ctx* c; ~ZeroGuard() { trgt::zero_end(c); } } zero_guard{c};
  { const int zrc = trgt::zero_begin(c); if (zrc) return zrc; }  // (on the call's stream, in front of everything: every other stream forks off behind it)
  const int F = p->flank_len;
  if (F <= 0) return fail(c, TRGT_ERR_INVALID, "trgt_locus_batch: flank_len must be positive");
  const int64_t nr = (int64_t)in->locus_read_begin[nl];
  if (2 * nr > 0xFFFFFFF0ll) return fail(c, TRGT_ERR_UNSUPPORTED, "trgt_locus_batch: too many reads in one call");
  // reads handed over as BAM 4-bit codes: expanded in HBM, and from here on the batch is one with its reads on the device (the host
  // glue of the later stages fetches the few read segments it needs from there, as it does for a caller's device blob)
  trgt_locus_batch_in in_expanded;
  std::vector<uint64_t> expanded_off;
  if (in->read_encoding != TRGT_READS_ASCII && in->read_encoding != TRGT_READS_BAM4) return fail(c, TRGT_ERR_INVALID, "trgt_locus_batch: read_encoding %d", in->read_encoding);
  if (in->read_encoding == TRGT_READS_BAM4 && nr > 0) {
    uint64_t packed_total = 0;
    expanded_off.resize((size_t)nr);
    for (int64_t r = 0; r < nr; ++r) {
      packed_total = std::max<uint64_t>(packed_total, in->read_off[r] + ((uint64_t)in->read_len[r] + 1) / 2);
      expanded_off[(size_t)r] = 2 * in->read_off[r];
    }
    const uint8_t* d_packed = nullptr;
    void* d_exp = nullptr;
    int erc;
    if (ready) TRGT_HIP_TRY(c, hipStreamWaitEvent(c->stream, ready, 0));
    if (staged_reads) d_packed = staged_reads;
    else if ((erc = dev_in(c, S_READS_PACKED, in->read_blob, (size_t)packed_total, &d_packed))) return erc;
    if ((erc = dev_get(c, S_READS_EXPANDED, 2 * (size_t)packed_total + 64, &d_exp))) return erc;
    const unsigned blocks = (unsigned)std::min<uint64_t>((uint64_t)c->num_cus * 16, packed_total / (16 * 256) + 1);
    hipLaunchKernelGGL(expand_bam4_kernel, dim3(blocks), dim3(256), 0, c->stream, d_packed, (uint8_t*)d_exp, packed_total);
    TRGT_HIP_TRY(c, hipGetLastError());
    in_expanded = *in;
    in_expanded.read_blob = (const uint8_t*)d_exp; in_expanded.read_off = expanded_off.data(); in_expanded.read_encoding = TRGT_READS_ASCII;
    in = &in_expanded;
    staged_reads = nullptr;
  }
  int threads = p->host_threads > 0 ? p->host_threads : (int)std::max(1u, std::thread::hardware_concurrency());
  // a few microseconds of work per locus: more threads only add wake-up cost.  The device genotyper leaves the
  // host little to do (4 threads measure the same as 32, and large pools produce the occasional late wake-up)
  threads = std::min(threads, !c->knobs.host_genotyper && p->min_read_qual >= 0.9 ? 8 : 32);
  HostPool* pool = host_pool(c, threads);
  const int64_t t0 = now_ns();
  c->tl_t0 = t0;
  const bool tl_on = c->knobs.timeline;
#define TL(name) do { if (tl_on) fprintf(stderr, "[tl] %-28s %7.2f ms  ctx=%p\n", name, (double)(now_ns() - t0) / 1e6, (void*)c); } while (0)
  int64_t tA = 0, tB = 0, tC = 0, tHost = 0;
  int64_t stat_flank_jobs = 0, stat_flank_heavy = 0, stat_cons_jobs = 0, stat_spanning = 0, stat_hmm_jobs = 0, stat_ed_jobs = 0;
  auto init_outputs = [&]() {
    for (int64_t l = 0; l < nl; ++l) {
      out->n_alleles[l] = 0; out->allele_len[2 * l] = out->allele_len[2 * l + 1] = 0; out->num_spanning[2 * l] = out->num_spanning[2 * l + 1] = 0;
      if (out->gt_size) out->gt_size[2 * l] = out->gt_size[2 * l + 1] = 0;
      if (out->flipped) out->flipped[l] = 0;
    }
    for (int64_t r = 0; r < nr; ++r) { out->classification[r] = -1; out->read_rank[r] = -1; out->span_start[r] = out->span_end[r] = -1; }
    for (int64_t s = 0; s < 2 * nl; ++s) { out->n_spans[s] = 0; out->purity[s] = std::nan(""); }
  };
  if (nr == 0) { init_outputs(); return TRGT_OK; }
  if (!c->stream2) TRGT_HIP_TRY(c, trgt::make_stream(c, &c->stream2));
  // The motif-HMM tables depend only on the catalog: build them on a host thread while the GPU locates flanks.
  // They are uploaded from the same thread (second stream), so stage C finds them in HBM.
  // (the thread touches nothing of the context: its result and error live in `models`; the upload happens on this thread, below,
  //  once stage A is on the GPU)
  HmmModels models;  // the motif-HMM tables depend only on the catalog: built on the device, behind stage A's launches (below)
  const hipStream_t upload_stream = c->stream2;  // (the "second stream" of this call, whichever of the two is current later on)
  HmmPending *hmm_pending = nullptr, *hmm_pending2 = nullptr, *hmm_pendingB = nullptr;
  struct PendGuard { HmmPending*& p; ~PendGuard() { if (p) hmm_pending_free(p); } } pend_guard{hmm_pending}, pend_guard2{hmm_pending2}, pend_guardB{hmm_pendingB};
  std::vector<uint32_t> js2, sl2, ns2; std::vector<uint64_t> so2, spo2, co2; std::vector<double> pu2; std::vector<int64_t> slot2;  // stage C, host-path loci

  // ---------------- stage A: flank location on the GPU (span_locater.rs:32-68), enqueued without host waits
  std::vector<uint64_t> piece_off(2 * (size_t)nl);
  std::vector<uint32_t> read_locus((size_t)nr), heavy_len((size_t)nl);
  uint64_t flank_total = 0, read_total = 0, tr_total = 0, allele_total = 0;
  uint32_t max_read_len = 0, heavy_tlen_max = 0;
  uint64_t max_locus_reads = 0;
  {
    struct alignas(64) Acc { uint64_t flank = 0, read = 0, tr = 0, allele = 0, reads = 0; uint32_t max_len = 0, heavy = 0; };  // one cache line per worker
    std::vector<Acc> acc((size_t)pool->size());
    std::atomic<int64_t> short_flank{-1};
    pool->parallel_for(nl, 256, [&](int64_t l, int t) {
      if ((int64_t)in->lf_len[l] < F || (int64_t)in->rf_len[l] < F) { short_flank = l; return; }
      piece_off[2 * l] = in->lf_off[l] + in->lf_len[l] - (uint64_t)F;  // lf[lf.len()-F..]  (span_locater.rs:38)
      piece_off[2 * l + 1] = in->rf_off[l];                            // rf[..F]           (:39)
      uint64_t rt = 0; uint32_t ml = 0;
      for (uint64_t r = in->locus_read_begin[l]; r < in->locus_read_begin[l + 1]; ++r) {
        read_locus[r] = (uint32_t)l;
        rt = std::max<uint64_t>(rt, in->read_off[r] + in->read_len[r]);
        ml = std::max(ml, in->read_len[r]);
      }
      heavy_len[(size_t)l] = heavy_read_len(ml, F);
      Acc& a = acc[(size_t)t];
      a.flank = std::max<uint64_t>(a.flank, std::max(in->lf_off[l] + in->lf_len[l], in->rf_off[l] + in->rf_len[l]));
      a.read = std::max(a.read, rt); a.max_len = std::max(a.max_len, ml); a.heavy = std::max(a.heavy, heavy_len[(size_t)l]);
      a.reads = std::max<uint64_t>(a.reads, in->locus_read_begin[l + 1] - in->locus_read_begin[l]);
      a.tr = std::max<uint64_t>(a.tr, in->tr_off[l] + in->tr_len[l]);
      a.allele = std::max<uint64_t>(a.allele, std::max(out->allele_off[2 * l], out->allele_off[2 * l + 1]) + out->allele_cap[l]);
    });
    if (short_flank >= 0) return fail(c, TRGT_ERR_INVALID, "trgt_locus_batch: locus %lld flank shorter than flank_len", (long long)short_flank.load());
    for (const Acc& a : acc) {
      flank_total = std::max(flank_total, a.flank); read_total = std::max(read_total, a.read); max_read_len = std::max(max_read_len, a.max_len);
      tr_total = std::max(tr_total, a.tr); allele_total = std::max(allele_total, a.allele); heavy_tlen_max = std::max(heavy_tlen_max, a.heavy);
      max_locus_reads = std::max(max_locus_reads, a.reads);
    }
  }
  c->dbg_ns[0] = now_ns() - t0;  // set-up: thread pool, model thread, piece / read-locus tables
  TL("set-up");
  const bool reads_on_device = is_device_ptr(in->read_blob);
  // filter_impure_trs (tr.rs:37-50) sits between get_spanning_reads and the genotyper: with it on, every locus takes the host path
  const bool impure_filter = p->min_read_qual < 0.9;
  // (host reads are uploaded for the flank scan anyway: the device genotyper then works on that copy just as well)
  const bool dev_gt = !c->knobs.host_genotyper && !impure_filter;
  auto is_cluster = [&](int64_t l) { return in->genotyper && in->genotyper[l] == 1; };
  // genotype_flank (tr.rs:69-75) can only change a genotype when reads carry haplotype tags or mismatch offsets
  const bool flank_on = in->hp_tag != nullptr || (in->mismatch_offsets != nullptr && in->mismatch_off != nullptr);
  const FlankMeta flank_meta{in->hp_tag, in->start_offset, in->end_offset, in->mismatch_offsets, in->mismatch_off};
  int rc;
  const uint8_t *d_flank = nullptr, *d_reads = nullptr;
  const uint64_t *d_piece = nullptr, *d_roff = nullptr;
  const uint32_t *d_rlen = nullptr, *d_rloc = nullptr, *d_heavy = nullptr;
  void *d_ss = nullptr, *d_se = nullptr, *d_hl = nullptr, *d_hr = nullptr;
  void *h_ss = nullptr, *h_se = nullptr, *h_hl = nullptr, *h_hr = nullptr, *h_cells = nullptr;
  if (ready) TRGT_HIP_TRY(c, hipStreamWaitEvent(c->stream, ready, 0));
  if (staged_flank) d_flank = staged_flank;
  else if ((rc = dev_in(c, S_FS_FLANK, in->flank_blob, (size_t)flank_total, &d_flank))) return rc;
  if (staged_reads) d_reads = staged_reads;
  else if ((rc = dev_in(c, S_FS_READS, in->read_blob, (size_t)read_total, &d_reads))) return rc;
  // the offset tables of stage A and of the device genotyper go up as ONE dispatch (a dozen separate ones were 0.35 ms in front of the scan)
  UploadBatch ub(c, c->stream);
  if ((rc = dev_in(c, S_FS_JOBS, piece_off.data(), piece_off.size(), &d_piece, &ub)) ||
      (rc = dev_in(c, S_FS_LIST, in->read_off, (size_t)nr, &d_roff, &ub)) ||
      (rc = dev_in(c, S_FS_OUT0, in->read_len, (size_t)nr, &d_rlen, &ub)) ||
      (rc = dev_in(c, S_FS_OUT1, read_locus.data(), (size_t)nr, &d_rloc, &ub)) ||
      (rc = dev_in(c, S_FS_HEAVY, heavy_len.data(), (size_t)nl, &d_heavy, &ub)) ||
      (rc = pin_get(c, P_CELLS, 64, &h_cells)))
    return rc;
  // Everything stage A hands back to the host lives in ONE device slab mirrored by ONE pinned slab, so that it comes back in a single
  // copy: fourteen separate hipMemcpyAsync D2H cost the host 0.3 ms of completion handling right after the event wait.
  struct Slab {
    size_t total = 0;
    size_t add(size_t bytes) { const size_t o = total; total += (bytes + 255) & ~(size_t)255; return o; }
  } slab;
  const size_t o_ss = slab.add((size_t)nr * 4), o_se = slab.add((size_t)nr * 4), o_hl = slab.add((size_t)nr), o_hr = slab.add((size_t)nr);
  size_t o_need = 0, o_nal = 0, o_alen = 0, o_ci = 0, o_nsp = 0, o_cls = 0, o_rank = 0, o_nspan = 0, o_toff = 0, o_flip = 0, o_gsz = 0, o_rpc = 0, o_skipb = 0, o_clc = 0;
  if (dev_gt) {
    o_need = slab.add((size_t)nl); o_nal = slab.add((size_t)nl * 4); o_alen = slab.add(2 * (size_t)nl * 4); o_ci = slab.add(4 * (size_t)nl * 4);
    o_nsp = slab.add(2 * (size_t)nl * 4); o_cls = slab.add((size_t)nr * 4); o_rank = slab.add((size_t)nr * 4); o_nspan = slab.add((size_t)nl * 4);
    o_toff = slab.add((2 * (size_t)nl + 1) * 8); o_flip = slab.add((size_t)nl);
    o_gsz = slab.add(2 * (size_t)nl * 4); o_rpc = slab.add(gt::RC_WORDS * 4); o_skipb = slab.add((size_t)nl); o_clc = slab.add(cl::CC_WORDS * 4);
  }
  void *d_slab = nullptr, *h_slab = nullptr;
  if ((rc = dev_get(c, S_LOCUS_4, slab.total, &d_slab)) || (rc = pin_get(c, P_SPAN_S, slab.total, &h_slab))) return rc;
  auto dsl = [&](size_t o) { return (void*)((uint8_t*)d_slab + o); };
  auto hsl = [&](size_t o) { return (void*)((uint8_t*)h_slab + o); };
  d_ss = dsl(o_ss); d_se = dsl(o_se); d_hl = dsl(o_hl); d_hr = dsl(o_hr);
  h_ss = hsl(o_ss); h_se = hsl(o_se); h_hl = hsl(o_hl); h_hr = hsl(o_hr);
  // device genotyper: inputs it needs beyond stage A's, and its outputs (device + pinned mirrors)
  struct GtDev {
    const uint64_t* lrb = nullptr; const uint8_t* ploidy = nullptr; const uint8_t* tr = nullptr; const uint64_t* tr_off = nullptr;
    const uint32_t* tr_len = nullptr; const uint64_t* al_off = nullptr; const uint32_t* al_cap = nullptr; const uint8_t* geno = nullptr;
    void *need = nullptr, *nal = nullptr, *blob = nullptr, *alen = nullptr, *ci = nullptr, *nsp = nullptr, *cls = nullptr, *rank = nullptr, *nspan = nullptr,
         *toff = nullptr, *packed = nullptr;
  } g;
  std::vector<uint32_t> cl_list; std::vector<uint64_t> cl_moff;
  uint64_t cl_pairs = 0, cl_reads = 0; uint32_t cl_max_nr = 0;
  const uint32_t* d_cl_list = nullptr; const uint64_t* d_cl_moff = nullptr;
  struct GtHost { void *need = nullptr, *nal = nullptr, *alen = nullptr, *ci = nullptr, *nsp = nullptr, *cls = nullptr, *rank = nullptr, *nspan = nullptr, *toff = nullptr, *packed = nullptr; } gh;
  if (dev_gt) {
    if ((rc = dev_in(c, S_GT_LRB, in->locus_read_begin, (size_t)nl + 1, &g.lrb, &ub)) || (rc = dev_in(c, S_GT_PLOIDY, in->ploidy, (size_t)nl, &g.ploidy, &ub)) ||
        (rc = dev_in(c, S_GT_TR, in->tr_blob, (size_t)tr_total, &g.tr, &ub)) || (rc = dev_in(c, S_GT_TROFF, in->tr_off, (size_t)nl, &g.tr_off, &ub)) ||
        (rc = dev_in(c, S_GT_TRLEN, in->tr_len, (size_t)nl, &g.tr_len, &ub)) || (rc = dev_in(c, S_GT_ALOFF, out->allele_off, 2 * (size_t)nl, &g.al_off, &ub)) ||
        (rc = dev_in(c, S_GT_ALCAP, out->allele_cap, (size_t)nl, &g.al_cap, &ub)) || (in->genotyper && (rc = dev_in(c, S_GT_GENO, in->genotyper, (size_t)nl, &g.geno, &ub))) ||
        (rc = dev_get(c, S_GT_BLOB, (size_t)allele_total + 16, &g.blob)) || (rc = dev_get(c, S_GT_PACKED, (size_t)allele_total + 16, &g.packed)))
      return rc;
    // Genotyper::Cluster loci stay on the device too (locus_cluster_dev.hpp) unless TRGT_HOST_CLUSTER / TRGT_SPLIT_HMM say otherwise: the
    // list of them and the first pair slot of each (condensed distance matrix, sized for all reads of the locus)
    if (in->genotyper && !c->knobs.host_cluster && !c->knobs.split_hmm) {
      for (int64_t l = 0; l < nl; ++l) {
        if (in->genotyper[l] != 1 || in->ploidy[l] == 0) continue;
        const uint64_t n = in->locus_read_begin[l + 1] - in->locus_read_begin[l];
        if (n == 0 || n > (uint64_t)gt::GT_MAX_READS) continue;
        cl_list.push_back((uint32_t)l); cl_moff.push_back(cl_pairs);
        cl_pairs += n * (n - 1) / 2; cl_reads += n; cl_max_nr = std::max(cl_max_nr, (uint32_t)n);
      }
      if (cl_pairs > 0x7FFFFFF0ull) { cl_list.clear(); cl_moff.clear(); }  // (pair slots are 32-bit output indices: the host path takes such a batch)
      if (!cl_list.empty() && ((rc = dev_in(c, S_CL_LIST, cl_list.data(), cl_list.size(), &d_cl_list, &ub)) || (rc = dev_in(c, S_CL_MOFF, cl_moff.data(), cl_moff.size(), &d_cl_moff, &ub))))
        return rc;
    }
    g.need = dsl(o_need); g.nal = dsl(o_nal); g.alen = dsl(o_alen); g.ci = dsl(o_ci); g.nsp = dsl(o_nsp); g.cls = dsl(o_cls); g.rank = dsl(o_rank);
    g.nspan = dsl(o_nspan); g.toff = dsl(o_toff);
    gh.need = hsl(o_need); gh.nal = hsl(o_nal); gh.alen = hsl(o_alen); gh.ci = hsl(o_ci); gh.nsp = hsl(o_nsp); gh.cls = hsl(o_cls); gh.rank = hsl(o_rank);
    gh.nspan = hsl(o_nspan); gh.toff = hsl(o_toff);
  }
  if (!ub.staging.empty()) {  // the host copies into pinned staging, spread over the pool's threads
    auto& st = ub.staging;
    constexpr size_t PIECE = 256 << 10;
    std::vector<std::array<size_t, 3>> parts;  // staging index, offset, bytes
    for (size_t i = 0; i < st.size(); ++i) for (size_t o = 0; o < st[i].bytes; o += PIECE) parts.push_back({i, o, std::min(PIECE, st[i].bytes - o)});
    pool->parallel_for((int64_t)parts.size(), 1, [&](int64_t k, int) { const auto& q = parts[(size_t)k]; std::memcpy((uint8_t*)st[q[0]].pinned + q[1], (const uint8_t*)st[q[0]].src + q[1], q[2]); });
    st.clear();
  }
  if ((rc = ub.flush())) return rc;
  c->dbg_ns[1] = now_ns() - t0;  // + uploads of the offset tables, buffer (re)allocation
  TL("tables uploaded, buffers ready");
  trgt_span_params sp; sp.flank_len = F; sp.min_flank_id_frac = p->min_flank_id_frac; sp.mism = p->mism; sp.gapo = p->gapo; sp.gape = p->gape;
  hipEvent_t evA = nullptr;
  struct EvGuard { hipEvent_t& e; ~EvGuard() { if (e) (void)hipEventDestroy(e); } } ev_guard{evA};
  TRGT_HIP_TRY(c, hipEventCreateWithFlags(&evA, hipEventDisableTiming));
  ((uint64_t*)h_cells)[0] = ((uint64_t*)h_cells)[1] = ((uint64_t*)h_cells)[2] = 0;
  // Several contexts on one GPU (one host thread each, trgt_amd/driver.py) form a pipeline: while one call is in its tail (results
  // back, the loci of the host path, HMM) the next ones' flank location has the GPU.  TRGT_STAGE_LOCK=1 lets only one call per device
  // be in stage A at a time -- that was worth 2x while the tails were host-bound (first half of round 2); since the stage-C job list
  // and the model tables are made on the device, letting the stages overlap freely is 14 % faster on the 10k-locus batch (1.50 -> 1.71 M
  // loci/s with four contexts) and neutral on the others.
  std::unique_lock<std::mutex> stage_a_token(g_stage_a_mutex[(size_t)c->device % 16], std::defer_lock);
  if (c->knobs.stage_lock) stage_a_token.lock();
  if ((rc = find_spans_device(c, sp, nl, nr, d_flank, d_piece, d_reads, d_roff, d_rlen, d_rloc, max_read_len, (int32_t*)d_ss, (int32_t*)d_se,
                              (uint8_t*)d_hl, (uint8_t*)d_hr, d_heavy, heavy_tlen_max > 0 ? heavy_tlen_max - 1 : 0)))
    return rc;
  if (c->last_wfa_cells_dev) { const int d2h_rc = trgt::d2h(c, h_cells, c->last_wfa_cells_dev, 24, c->stream); if (d2h_rc) return d2h_rc; }
  ((uint64_t*)h_cells)[4] = ((uint64_t*)h_cells)[5] = 0;  // pre-filter: offsets computed, alignments kept
  if (c->last_filter_cells_dev) { const int d2h_rc = trgt::d2h(c, (uint64_t*)h_cells + 4, c->last_filter_cells_dev, 16, c->stream); if (d2h_rc) return d2h_rc; }
  // the motif-HMM tables: built on the device (one small upload and one kernel on the copy stream, which has nothing in front of it:
  // the second stream may be busy with the heavy flank alignments for milliseconds, and the copy engine serves the streams' copies in
  // the order they were issued)
  if (!c->stream_copy) TRGT_HIP_TRY(c, trgt::make_stream(c, &c->stream_copy));
  if (!c->ev_upload) TRGT_HIP_TRY(c, hipEventCreateWithFlags(&c->ev_upload, hipEventDisableTiming));
  tl_mark(c, "find_spans enqueued");
  if ((rc = hmm_models_on_device(c, (int32_t)nl, in->motif_blob, in->motif_off, in->set_motif_begin, models, c->stream_copy, c->ev_upload)))
    return models.err.empty() ? rc : fail(c, rc, "%s", models.err.c_str());
  TRGT_HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_upload, 0));
  TRGT_HIP_TRY(c, hipStreamWaitEvent(c->stream2, c->ev_upload, 0));
  TL("models uploaded");
  // ---------------- stage C of the device-genotyped loci is enqueued behind the genotyper and before the host has seen its results: the
  // job list is resolved on the device (hmm_enqueue_slots), so the HMM kernels start the moment the genotyper ends instead of after a
  // host round trip (event wait, job lists, uploads: 0.8 ms of a 10k-locus call, 2.6 ms with mixed model sizes).  Two batches: the loci
  // the genotyper settles (on the call's stream, next to the consensus repair of the others on the second stream), then the repaired ones.
  const bool use_slots = dev_gt && models.rc == 0 && models.d_sets && !is_device_ptr(out->spans3) && !c->knobs.host_hmm_lists;
  std::vector<uint8_t> slot_skip;
  HmmSlots hs;
  std::vector<uint32_t> nsB; std::vector<double> puB;  // n_spans / purity of the second batch (merged into the outputs slot by slot)
  if (use_slots) {
    hs.n_loci = nl; hs.cap = out->allele_cap; hs.seq_off = out->allele_off; hs.seq_blob_dev = (const uint8_t*)g.blob;
    hs.d_n_alleles = (const int32_t*)g.nal; hs.d_allele_len = (const uint32_t*)g.alen;
    if (in->genotyper && cl_list.empty()) { slot_skip.assign(in->genotyper, in->genotyper + nl); for (auto& v : slot_skip) v = v == 1; hs.host_skip = slot_skip.data(); }
  }
  if (dev_gt) {
    gt::GtArgs ga;
    ga.reads = d_reads; ga.read_off = d_roff; ga.read_len = d_rlen; ga.locus_read_begin = g.lrb;
    ga.span_start = (const int32_t*)d_ss; ga.span_end = (const int32_t*)d_se;
    ga.ploidy = g.ploidy; ga.tr_blob = g.tr; ga.tr_off = g.tr_off; ga.tr_len = g.tr_len; ga.allele_off = g.al_off; ga.allele_cap = g.al_cap;
    ga.n_loci = nl; ga.flank_len = F; ga.max_depth = p->max_depth;
    ga.need_host = (uint8_t*)g.need; ga.n_alleles = (int32_t*)g.nal; ga.allele_blob = (uint8_t*)g.blob; ga.allele_len = (uint32_t*)g.alen;
    ga.ci = (int32_t*)g.ci; ga.num_spanning = (int32_t*)g.nsp; ga.classification = (int32_t*)g.cls; ga.read_rank = (int32_t*)g.rank;
    ga.n_spanning_reads = (uint32_t*)g.nspan; ga.flipped = (uint8_t*)dsl(o_flip);
    ga.gt_size = (int32_t*)dsl(o_gsz); ga.skip_b = (uint8_t*)dsl(o_skipb); ga.genotyper = g.geno;
    // ---- stage B on the device for the loci whose pick lacks majority support (locus_gt.hpp): job list, vote groups and the
    //      record of the decisions are written by the genotyper; the alignment kernel, the column voting and the finishing kernel
    //      follow on the same stream, and the host first hears of these loci when they are done.  TRGT_HOST_REPAIR=1: the host path.
    gt::RepairBufs& rp = ga.rp;
    std::memset(&rp, 0, sizeof rp);
    void *d_vout = nullptr, *d_vlen = nullptr, *d_vscr = nullptr, *d_rcig = nullptr, *d_rclen = nullptr;
    const bool dev_repair = !c->knobs.host_repair;
    const uint32_t* cl_counts_dev = nullptr;  // count block of the device-side cluster genotyper, when it runs
    if (dev_repair) {
      void* const z_rpc = zero_take(c, 256);  // (cleared with the call's zero arena; else by the kernel below)
      if (z_rpc) rp.counts = (uint32_t*)z_rpc;
      else { void* d_cnt = nullptr; if ((rc = dev_get(c, S_RP_COUNTS, 256, &d_cnt))) return rc; rp.counts = (uint32_t*)d_cnt; }
      rp.cap_groups = (uint32_t)std::min<int64_t>(2 * nl, 0x7FFFFFFF); rp.cap_jobs = (uint32_t)nr;
      // (segments beyond what the register-resident kernels take go to the generic engine, whose workgroups are bounded by ws_budget: a batch of
      //  10-kb alleles -- cfg3 -- is repaired on the device too instead of going back to the host: one-context call 22.2 -> 14.4 ms)
      const uint32_t seg_auto = std::min<uint32_t>(16384u, std::max<uint32_t>(1024u, max_read_len > 2u * (uint32_t)F ? max_read_len - 2u * (uint32_t)F : 0u));
      rp.max_seg = c->knobs.repair_max_seg > 0 ? (uint32_t)std::max(16, c->knobs.repair_max_seg) : seg_auto; rp.vote_lds_pos = (uint32_t)vote::VOTE_LDS_POS;
      // room for the typical batch (a few per cent of the loci, segments of a few hundred bases); a locus that finds none takes the host path
      rp.cap_cigar = std::min<uint64_t>((uint64_t)nr * (2ull * rp.max_seg + 1), 32ull << 20);       // words
      rp.cap_out = std::min<uint64_t>((uint64_t)nr * (rp.max_seg + 16ull) + 64, 256ull << 20);      // bytes
      rp.cap_scratch = std::min<uint64_t>(3ull * (uint64_t)nr + 3ull * (uint64_t)rp.cap_groups * (rp.max_seg + 1ull) + 64, 32ull << 20);  // words
      void *d_g = nullptr, *d_j = nullptr, *d_l = nullptr, *d_p = nullptr;
      if ((rc = dev_get(c, S_RP_GROUPS, (size_t)rp.cap_groups * sizeof(gt::RGroup), &d_g)) || (rc = dev_get(c, S_RP_JOBS, (size_t)rp.cap_jobs * sizeof(JobDev), &d_j)) ||
          (rc = dev_get(c, S_RP_LOCI, (size_t)nl * 4, &d_l)) || (rc = dev_get(c, S_RP_PEND, (size_t)nl * sizeof(gt::RepairPend), &d_p)) ||
          (rc = dev_get(c, S_RP_CIGAR, (size_t)rp.cap_cigar * 4, &d_rcig)) || (rc = dev_get(c, S_RP_CLEN, (size_t)rp.cap_jobs * 4, &d_rclen)) ||
          (rc = dev_get(c, S_RP_VOUT, (size_t)rp.cap_out + 16, &d_vout)) || (rc = dev_get(c, S_RP_VLEN, (size_t)rp.cap_groups * 4, &d_vlen)) ||
          (rc = dev_get(c, S_RP_VSCR, (size_t)rp.cap_scratch * 4 + 16, &d_vscr)))
        return rc;
      rp.groups = (gt::RGroup*)d_g; rp.jobs = (JobDev*)d_j; rp.loci = (uint32_t*)d_l; rp.pend = (gt::RepairPend*)d_p;
      if (!z_rpc) hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(64), 0, c->stream, rp.counts, (uint32_t)gt::RC_WORDS);
    }
    const bool small_gt = max_locus_reads <= 64;
    if (small_gt) hipLaunchKernelGGL((gt::locus_genotype_kernel<64, 8 * 1024>), dim3((unsigned)nl), dim3(64), 0, c->stream, ga);
    else hipLaunchKernelGGL((gt::locus_genotype_kernel<gt::GT_MAX_READS, gt::GT_SEG_LDS>), dim3((unsigned)nl), dim3(64), 0, c->stream, ga);
    TRGT_HIP_TRY(c, hipGetLastError());
    tl_mark(c, "genotyper launched");
    // Two ways to order the HMM of the settled loci and the repair of the others: one HMM batch behind the repair (default), or -- split_hmm,
    // TRGT_SPLIT_HMM=1 -- the HMM of the settled loci on a stream of its own next to the repair and a second batch for the repaired loci.
    // Measured: the split costs more in extra launches and streams than the overlap gives back, on every config.
    const bool split = use_slots && dev_repair && c->knobs.split_hmm;
    ga.finish_clears_need = split ? 0 : 1;  // (read by repair_finish_kernel only)
    if (use_slots && split) {
      if (!c->ev_gt) TRGT_HIP_TRY(c, hipEventCreateWithFlags(&c->ev_gt, hipEventDisableTiming));
      if (!c->ev_rp) TRGT_HIP_TRY(c, hipEventCreateWithFlags(&c->ev_rp, hipEventDisableTiming));
      if (!c->stream_hmm) TRGT_HIP_TRY(c, trgt::make_side_stream(c, &c->stream_hmm));
      TRGT_HIP_TRY(c, hipEventRecord(c->ev_gt, c->stream));
    }
    if (use_slots && split) {
      // the HMM batch of the loci the genotyper settled (need_host == 0), on a stream of its own: the call's stream goes on to the results
      // the host waits for (they must not queue behind the HMM kernels), the second stream to the device-side repair
      const hipStream_t main_stream = c->stream;
      TRGT_HIP_TRY(c, hipStreamWaitEvent(c->stream_hmm, c->ev_gt, 0)); TRGT_HIP_TRY(c, hipStreamWaitEvent(c->stream_hmm, c->ev_upload, 0)); c->stream = c->stream_hmm;
      hs.d_skip = (const uint8_t*)g.need;
      const int64_t tc0 = now_ns();
      rc = hmm_enqueue_slots(c, &models, hs, out->spans3, out->span_off, out->n_spans, out->motif_counts, out->count_off, out->purity, &hmm_pending);
      c->stream = main_stream;
      if (rc) return rc;
      tC += now_ns() - tc0;
      TL("hmm1 enqueued (device-resolved job list)");
    }
    if (dev_repair) {
      // on the second stream: alignments, voting, finish, and the HMM batch of the repaired loci right behind
      if (split) {
        TRGT_HIP_TRY(c, hipStreamWaitEvent(c->stream2, c->ev_gt, 0));
        std::swap(c->stream, c->stream2);
      }
      struct SwapBackRp { trgt_hip_ctx* c; bool on; ~SwapBackRp() { if (on) std::swap(c->stream, c->stream2); } } swap_back_rp{c, split};
      trgt_wfa_params wp;
      trgt_wfa_default_params(&wp);  // THREAD_WFA_CONSENSUS (genotype.rs:82-86): BiWFA, gap-affine 2,5,1, default heuristic
      wp.metric = 3; wp.mismatch = 2; wp.gap_open1 = 5; wp.gap_ext1 = 1; wp.span = 0; wp.scope = 1; wp.memory_mode = 3; sens_apply(c, wp);
      WfaLaunch LR;
      // (n_jobs_host bounds the workgroups and their workspaces, not the jobs: the count is read on the device)
      LR.jobs_dev = rp.jobs; LR.n_jobs_host = (int64_t)std::min<uint64_t>(rp.cap_jobs, (uint64_t)std::max(64, c->knobs.repair_blocks)); LR.n_jobs_dev = rp.counts + gt::RC_JOBS; LR.jobs_bound = rp.cap_jobs;
      LR.pat_base = d_reads; LR.txt_base = d_reads;
      LR.max_plen = rp.max_seg; LR.max_tlen = rp.max_seg; LR.max_sum = 2 * (int64_t)rp.max_seg;
      LR.cigar = (uint32_t*)d_rcig; LR.cigar_len = (uint32_t*)d_rclen; LR.buffer_set = 2; LR.ws_budget = 512ull << 20; LR.refused = rp.counts + gt::RC_REFUSED;
      auto dbg_sync = [&](const char* what) -> int {  // TRGT_WFA_DEBUG: which kernel of the chain a device fault belongs to
        if (!c->knobs.debug) return TRGT_OK;
        const hipError_t e = trgt::stream_wait(c, c->stream);
        uint32_t h[gt::RC_WORDS]; std::memset(h, 0, sizeof h);
        if (e == hipSuccess) (void)hipMemcpy(h, rp.counts, sizeof h, hipMemcpyDeviceToHost);
        fprintf(stderr, "[repair] %s: %s (groups %u jobs %u loci %u failed %u cigar words %llu)\n", what, hipGetErrorString(e), h[gt::RC_GROUPS], h[gt::RC_JOBS], h[gt::RC_LOCI], h[gt::RC_FAILED],
                (unsigned long long)h[gt::RC_CIGAR] | ((unsigned long long)h[gt::RC_CIGAR + 1] << 32));
        return e == hipSuccess ? TRGT_OK : fail(c, TRGT_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
      };
      if ((rc = dbg_sync("genotyper"))) return rc;
      if (uint32_t* trace_host = repair_trace_buf()) { std::memset(trace_host, 0, 4096); hipLaunchKernelGGL(repair_trace_kernel, dim3(1), dim3(64), 0, c->stream, rp, trace_host); }
      static const bool repair_check = TRGT_DEV_ENV("TRGT_REPAIR_CHECK") != nullptr;
      if (repair_check) {
        hipLaunchKernelGGL(repair_check_kernel, dim3(64), dim3(256), 0, c->stream, rp, (uint64_t)read_total);
        uint32_t h[gt::RC_WORDS];
        (void)trgt::stream_wait(c, c->stream);
        (void)hipMemcpy(h, rp.counts, sizeof h, hipMemcpyDeviceToHost);
        fprintf(stderr, "[repair check] jobs %u groups %u loci %u bad %u (last bad job %u: plen %u tlen %u out_index %u cigar_off %u) caps: jobs %u cigar %llu max_seg %u read bytes %llu\n", h[1], h[0], h[2], h[10], h[11], h[12], h[13],
                h[14], h[15], rp.cap_jobs, (unsigned long long)rp.cap_cigar, rp.max_seg, (unsigned long long)read_total);
      }
      if ((rc = wfa_launch(c, wp, LR))) return rc;
      if ((rc = dbg_sync("consensus alignments"))) return rc;
      vote::VoteArgs va{(const vote::Group*)rp.groups, 0u, rp.counts + gt::RC_GROUPS, d_reads, rp.jobs, (const uint32_t*)d_rcig, (const uint32_t*)d_rclen,
                        (uint32_t*)d_vscr, (uint8_t*)d_vout, (uint32_t*)d_vlen};
      hipLaunchKernelGGL(vote::consensus_vote_kernel, dim3((unsigned)rp.cap_groups), dim3(vote::VOTE_THREADS), 0, c->stream, va);
      const gt::FinishArgs fa{(const uint8_t*)d_vout, (const uint32_t*)d_vlen};
      const dim3 fgrid((unsigned)nl);
      if (small_gt) hipLaunchKernelGGL((gt::repair_finish_kernel<64>), fgrid, dim3(64), 0, c->stream, ga, fa);
      else hipLaunchKernelGGL((gt::repair_finish_kernel<gt::GT_MAX_READS>), fgrid, dim3(64), 0, c->stream, ga, fa);
      TRGT_HIP_TRY(c, hipGetLastError());
      if ((rc = dbg_sync("vote + finish"))) return rc;
      tl_mark(c, "repair chain enqueued");
      if (split) {
        TRGT_HIP_TRY(c, hipEventRecord(c->ev_rp, c->stream));  // (the results the host waits for do not wait for the second HMM batch)
        hs.d_skip = (const uint8_t*)dsl(o_skipb);
        nsB.assign(2 * (size_t)nl, 0); puB.assign(2 * (size_t)nl, 0.0);
        const int64_t tc0 = now_ns();
        if ((rc = hmm_enqueue_slots(c, &models, hs, out->spans3, out->span_off, nsB.data(), out->motif_counts, out->count_off, puB.data(), &hmm_pendingB, 1))) return rc;
        tC += now_ns() - tc0;
        std::swap(c->stream, c->stream2); swap_back_rp.on = false;
        TRGT_HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_rp, 0));
      }
    }
    if (!cl_list.empty()) {
      // ---- Genotyper::Cluster on the device (locus_cluster_dev.hpp): pair list -> edit distances -> linkage / groups / backbones ->
      //      consensus round -> redo or dropped reads -> second round + edit distances -> genotype.  Four alignment launches and two
      //      votes, most of them over empty lists after the first round; nothing waits for the host.
      const uint32_t n_cl = (uint32_t)cl_list.size();
      const uint32_t max_seg = max_read_len > 2u * (uint32_t)F ? max_read_len - 2u * (uint32_t)F : 1u;
      cl::ClArgs ca;
      std::memset(&ca, 0, sizeof ca);
      ca.g = ga; ca.list = d_cl_list; ca.n_list = n_cl; ca.mat_off = d_cl_moff;
      ca.flags = (c->knobs.sens_ward_ties ? 1u : 0u) | (c->knobs.sens_lw_order ? 2u : 0u);
      ca.cap_j = (uint32_t)cl_reads; ca.cap_g = 2 * n_cl; ca.vote_lds_pos = (uint32_t)vote::VOTE_LDS_POS;
      // arenas for two consensus rounds at worst-case slots per alignment, bounded: a locus that finds no room takes the host path
      ca.cap_cigar = std::min<uint64_t>(2ull * cl_reads * (2ull * max_seg + 1), 96ull << 20);   // words
      ca.cap_out = std::min<uint64_t>(2ull * (cl_reads + 2ull * n_cl) * ((uint64_t)max_seg + 16) + 64, 256ull << 20);  // bytes
      ca.cap_scratch = std::min<uint64_t>(6ull * cl_reads + 12ull * n_cl * ((uint64_t)max_seg + 1) + 64, 32ull << 20);  // words
      if (c->knobs.cluster_arena_kb > 0) {
        const uint64_t kb = (uint64_t)c->knobs.cluster_arena_kb;
        ca.cap_cigar = std::min<uint64_t>(ca.cap_cigar, kb * 256); ca.cap_out = std::min<uint64_t>(ca.cap_out, kb * 1024); ca.cap_scratch = std::min<uint64_t>(ca.cap_scratch, kb * 256);
      }
      const bool big = cl_max_nr > 64;
      void *d_cnt = nullptr, *d_rec = nullptr, *d_cls = nullptr, *d_es = nullptr, *d_gm = nullptr, *d_edj = nullptr, *d_j = nullptr, *d_g = nullptr, *d_ed2 = nullptr, *d_es2 = nullptr,
           *d_cig = nullptr, *d_clen = nullptr, *d_vout = nullptr, *d_vlen = nullptr, *d_vscr = nullptr;
      if ((rc = dev_get(c, S_CL_COUNTS, 256, &d_cnt)) || (rc = dev_get(c, S_CL_REC, (size_t)n_cl * sizeof(cl::ClRec), &d_rec)) || (rc = dev_get(c, S_CL_CLS, (size_t)nr + 16, &d_cls)) ||
          (rc = dev_get(c, S_CL_ESCORE, (size_t)cl_pairs * 4 + 16, &d_es)) || (big && (rc = dev_get(c, S_CL_GMAT, (size_t)cl_pairs * 8 + 16, &d_gm))) ||
          (rc = dev_get(c, S_CL_EDJOBS, (size_t)cl_pairs * sizeof(JobDev) + 16, &d_edj)) || (rc = dev_get(c, S_CL_JOBS, 2 * (size_t)ca.cap_j * sizeof(JobDev), &d_j)) ||
          (rc = dev_get(c, S_CL_GROUPS, 2 * (size_t)ca.cap_g * sizeof(gt::RGroup), &d_g)) || (rc = dev_get(c, S_CL_ED2JOBS, 2 * (size_t)cl_reads * sizeof(JobDev), &d_ed2)) ||
          (rc = dev_get(c, S_CL_ESCORE2, 2 * (size_t)nr * 4 + 16, &d_es2)) || (rc = dev_get(c, S_CL_CIGAR, (size_t)ca.cap_cigar * 4, &d_cig)) ||
          (rc = dev_get(c, S_CL_CLEN, 2 * (size_t)ca.cap_j * 4, &d_clen)) || (rc = dev_get(c, S_CL_VOUT, (size_t)ca.cap_out + 16, &d_vout)) ||
          (rc = dev_get(c, S_CL_VLEN, 2 * (size_t)ca.cap_g * 4, &d_vlen)) || (rc = dev_get(c, S_CL_VSCR, (size_t)ca.cap_scratch * 4 + 16, &d_vscr)))
        return rc;
      void* const z_clc = zero_take(c, 256);  // (cleared with the call's zero arena; else by the kernel below)
      ca.counts = z_clc ? (uint32_t*)z_clc : (uint32_t*)d_cnt; ca.rec = (cl::ClRec*)d_rec; ca.cls = (int8_t*)d_cls; ca.escore = (int32_t*)d_es; ca.gmat = (double*)d_gm;
      ca.ed_jobs = (JobDev*)d_edj; ca.jobs = (JobDev*)d_j; ca.groups = (gt::RGroup*)d_g; ca.ed2_jobs = (JobDev*)d_ed2; ca.escore2 = (int32_t*)d_es2;
      ca.vote_out = (const uint8_t*)d_vout; ca.vote_len = (const uint32_t*)d_vlen;
      if (!z_clc) hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(64), 0, c->stream, ca.counts, (uint32_t)cl::CC_WORDS);
      const dim3 cgrid(n_cl);
      if (big) hipLaunchKernelGGL((cl::cluster_front_kernel<gt::GT_MAX_READS>), cgrid, dim3(64), 0, c->stream, ca);
      else hipLaunchKernelGGL((cl::cluster_front_kernel<64>), cgrid, dim3(64), 0, c->stream, ca);
      TRGT_HIP_TRY(c, hipGetLastError());
      const int64_t wg_bound = (int64_t)std::max(64, 2 * c->knobs.repair_blocks);
      const int64_t ed_len = (int64_t)std::min<uint64_t>(max_seg, cl::CL_MAX_OPS);
      trgt_wfa_params wed;
      trgt_wfa_default_params(&wed);  // THREAD_WFA_ED (genotype.rs:88-92): edit distance, score only, BiWFA, default heuristic
      wed.metric = 1; wed.span = 0; wed.scope = 0; wed.memory_mode = 3; sens_apply(c, wed);
      trgt_wfa_params wco;
      trgt_wfa_default_params(&wco);  // THREAD_WFA_CONSENSUS (genotype.rs:82-86)
      wco.metric = 3; wco.mismatch = 2; wco.gap_open1 = 5; wco.gap_ext1 = 1; wco.span = 0; wco.scope = 1; wco.memory_mode = 3; sens_apply(c, wco);
      auto ed_launch = [&](const JobDev* jobs, uint64_t bound, const uint32_t* count, const uint8_t* txt_base, int32_t* score) -> int {
        WfaLaunch LE;
        LE.jobs_dev = jobs; LE.n_jobs_host = std::min<int64_t>((int64_t)std::max<uint64_t>(bound, 1), wg_bound); LE.n_jobs_dev = count; LE.jobs_bound = (int64_t)std::max<uint64_t>(bound, 1);
        // (a pair is aligned when li * lj <= MAX_OPS: an EMPTY segment pairs with one of any length -- gt_front keeps zero-length repeat
        //  segments -- so either side reaches max_seg; with both non-empty the sum stays within MAX_OPS + 1.  ADVICE r4: sized by ed_len alone
        //  the generic kernel's rings were too short for (0, > 10 kb) pairs)
        LE.pat_base = d_reads; LE.txt_base = txt_base; LE.max_plen = (int64_t)max_seg; LE.max_tlen = (int64_t)max_seg;
        LE.max_sum = std::max<int64_t>((int64_t)max_seg, std::min<int64_t>(2 * ed_len, (int64_t)cl::CL_MAX_OPS + 1));
        LE.score = score; LE.buffer_set = 2; LE.ws_budget = 512ull << 20; LE.refused = ca.counts + cl::CC_REFUSED;
        return wfa_launch(c, wed, LE);
      };
      auto cons_launch = [&](uint32_t first_job, const uint32_t* count, uint32_t first_group, const uint32_t* group_count) -> int {
        WfaLaunch LC;
        LC.jobs_dev = ca.jobs + first_job; LC.n_jobs_host = std::min<int64_t>((int64_t)ca.cap_j, wg_bound); LC.n_jobs_dev = count; LC.jobs_bound = (int64_t)ca.cap_j;
        LC.pat_base = d_reads; LC.txt_base = d_reads; LC.max_plen = max_seg; LC.max_tlen = max_seg; LC.max_sum = 2 * (int64_t)max_seg;
        LC.cigar = (uint32_t*)d_cig; LC.cigar_len = (uint32_t*)d_clen; LC.buffer_set = 2; LC.ws_budget = 512ull << 20; LC.refused = ca.counts + cl::CC_REFUSED;
        if (int r = wfa_launch(c, wco, LC)) return r;
        vote::VoteArgs va{(const vote::Group*)(ca.groups + first_group), 0u, group_count, d_reads, ca.jobs, (const uint32_t*)d_cig, (const uint32_t*)d_clen,
                          (uint32_t*)d_vscr, (uint8_t*)d_vout, (uint32_t*)d_vlen + first_group};
        hipLaunchKernelGGL(vote::consensus_vote_kernel, dim3((unsigned)ca.cap_g), dim3(vote::VOTE_THREADS), 0, c->stream, va);
        return TRGT_OK;
      };
      if ((rc = ed_launch(ca.ed_jobs, cl_pairs, ca.counts + cl::CC_ED, d_reads, ca.escore))) return rc;
      if (big) hipLaunchKernelGGL((cl::cluster_ward_kernel<gt::GT_MAX_READS, false>), cgrid, dim3(64), 0, c->stream, ca);
      else hipLaunchKernelGGL((cl::cluster_ward_kernel<64, true>), cgrid, dim3(64), 0, c->stream, ca);
      TRGT_HIP_TRY(c, hipGetLastError());
      if ((rc = cons_launch(0, ca.counts + cl::CC_J1, 0, ca.counts + cl::CC_G1))) return rc;
      if (big) hipLaunchKernelGGL((cl::cluster_round2_kernel<gt::GT_MAX_READS>), cgrid, dim3(64), 0, c->stream, ca);
      else hipLaunchKernelGGL((cl::cluster_round2_kernel<64>), cgrid, dim3(64), 0, c->stream, ca);
      TRGT_HIP_TRY(c, hipGetLastError());
      if ((rc = cons_launch(ca.cap_j, ca.counts + cl::CC_J2, ca.cap_g, ca.counts + cl::CC_G2))) return rc;
      if ((rc = ed_launch(ca.ed2_jobs, 2 * cl_reads, ca.counts + cl::CC_ED2, (const uint8_t*)d_vout, ca.escore2))) return rc;
      if (big) hipLaunchKernelGGL((cl::cluster_finish_kernel<gt::GT_MAX_READS>), cgrid, dim3(64), 0, c->stream, ca);
      else hipLaunchKernelGGL((cl::cluster_finish_kernel<64>), cgrid, dim3(64), 0, c->stream, ca);
      TRGT_HIP_TRY(c, hipGetLastError());
      tl_mark(c, "cluster chain enqueued");
      cl_counts_dev = ca.counts;
    }
    // (the count blocks come back with the slab)
    const CountCopy cc{dev_repair ? rp.counts : nullptr, (uint32_t*)dsl(o_rpc), (uint32_t)gt::RC_WORDS, cl_counts_dev, (uint32_t*)dsl(o_clc), (uint32_t)cl::CC_WORDS};
    hipLaunchKernelGGL(allele_prefix_kernel, dim3(1), dim3(1024), 0, c->stream, (const uint32_t*)g.alen, (uint64_t*)g.toff, (int64_t)(2 * nl), cc);
    hipLaunchKernelGGL(allele_pack_kernel, dim3((unsigned)((2 * nl + 3) / 4)), dim3(256), 0, c->stream, (const uint8_t*)g.blob, g.al_off,
                       (const uint32_t*)g.alen, (const uint64_t*)g.toff, (uint8_t*)g.packed, (int64_t)(2 * nl));
    TRGT_HIP_TRY(c, hipGetLastError());
  }
  ((uint64_t*)h_cells)[6] = 0;  // wavefront offsets of the device-side consensus / edit-distance launches (timing runs only)
  if (c->timing && c->wfa_cells_cur[2]) { const int d2h_rc = trgt::d2h(c, (uint64_t*)h_cells + 6, c->wfa_cells_cur[2], 8, c->stream); if (d2h_rc) return d2h_rc; }
  { const int d2h_rc = trgt::d2h(c, h_slab, d_slab, slab.total, c->stream); if (d2h_rc) return d2h_rc; }
  TRGT_HIP_TRY(c, hipEventRecord(evA, c->stream));
  c->dbg_ns[2] = now_ns() - t0;  // + stage A enqueued
  TL("stage A enqueued");
  init_outputs();  // host-only work: done while the GPU is already busy
  const int64_t tw_a = now_ns();  // from here on the host waits for stage A (the table upload below sits behind it in the copy queue)
  if (use_slots && !hmm_pending && !hmm_pendingB) {  // one HMM batch behind the genotyper and the repair (need_host is final by then)
    hs.d_skip = (const uint8_t*)g.need;
    const int64_t tc0 = now_ns();
    if ((rc = hmm_enqueue_slots(c, &models, hs, out->spans3, out->span_off, out->n_spans, out->motif_counts, out->count_off, out->purity, &hmm_pending))) return rc;
    tC += now_ns() - tc0;
    TL("hmm1 enqueued (device-resolved job list)");
  }
  if ((rc = issue_pending_uploads(c))) return rc;  // the next batch's bytes: behind this call's tables and models, next to stage A

  // ---------------- wait for the GPU, publish spans
  {
    { const hipError_t ea = trgt::event_wait(evA);
      if (g_trace_dump && TRGT_DEV_ENV("TRGT_REPAIR_TRACE")) { const uint32_t* h = g_trace_dump; fprintf(stderr, "[repair trace] evA %s counts:", hipGetErrorString(ea)); for (int i = 0; i < 16; ++i) fprintf(stderr, " %u", h[i]);
        fprintf(stderr, "\n"); }
      if (ea != hipSuccess) return trgt::fail(c, TRGT_ERR_HIP, "trgt::event_wait(evA) failed: %s", hipGetErrorString(ea)); }
    if (stage_a_token.owns_lock()) stage_a_token.unlock();
    tA = now_ns() - tw_a;
  TL("evA");
  }
  int64_t th_begin = now_ns();
  // ---------------- loci for the host path: all of them without the device genotyper, else the ones it handed back
  std::vector<int64_t> R;
  std::vector<uint8_t> need_a;  // need_host as the first HMM batch saw it (2 = waiting for the device-side repair: not its job)
  if (dev_gt) {
    uint8_t* need = (uint8_t*)gh.need;
    const uint8_t* skip_b = (const uint8_t*)hsl(o_skipb);
    if (hmm_pendingB) need_a.assign(need, need + nl);
    for (int64_t l = 0; l < nl; ++l) if (need[l] == 2) need[l] = skip_b[l] ? 1 : 0;  // repaired on the device, or back to the host path after all
    if (in->genotyper && cl_list.empty()) for (int64_t l = 0; l < nl; ++l) if (in->genotyper[l] == 1) need[l] = 1;  // Genotyper::Cluster: host-driven rounds (else: genotyped by the device chain, need_host = 0)
    if (flank_on) {
      // device-genotyped loci whose two alleles are at most 10 bases apart and whose reads DO split by haplotype tag or flank SNVs
      // take the host path, where the genotype is replaced (genotype_flank below); the split only needs the per-read fields
      const int32_t* nal = (const int32_t*)gh.nal; const uint32_t* alen = (const uint32_t*)gh.alen; const int32_t* rank = (const int32_t*)gh.rank;
      pool->parallel_for(nl, 64, [&](int64_t l, int) {
        if (need[l] || nal[l] != 2 || adiff(alen[2 * l], alen[2 * l + 1]) > 10) return;
        const uint64_t r0 = in->locus_read_begin[l], r1 = in->locus_read_begin[l + 1];
        std::vector<uint32_t> order;
        for (uint64_t r = r0; r < r1; ++r) if (rank[r] >= 0) { if ((size_t)rank[r] >= order.size()) order.resize((size_t)rank[r] + 1, 0); order[(size_t)rank[r]] = (uint32_t)r; }
        FlankSplit sp;
        if (flank_split(flank_meta, order.data(), order.size(), sp)) need[l] = 1;
      });
    }
    for (int64_t l = 0; l < nl; ++l) if (need[l]) R.push_back(l);
    // an alignment job of a device-built list that the generic kernel refused (longer than the workspace planned from the batch's maxima)
    // left INT32_MIN / an empty CIGAR behind: nothing downstream may use that -- the call fails (never seen in the sweeps)
    if (const uint32_t refused = ((const uint32_t*)hsl(o_rpc))[gt::RC_REFUSED] + ((const uint32_t*)hsl(o_clc))[cl::CC_REFUSED])
      return fail(c, TRGT_ERR_UNSUPPORTED, "trgt_locus_batch: %u alignment job(s) of the device-side genotyper chains exceed the planned workspace", refused);
    stat_cons_jobs += (int64_t)((const uint32_t*)hsl(o_rpc))[gt::RC_JOBS];  // consensus alignments of the device-side repair
    { const uint32_t* cc = (const uint32_t*)hsl(o_clc);  // ... and of the device-side cluster genotyper, with its edit distances
      stat_cons_jobs += (int64_t)cc[cl::CC_J1] + (int64_t)cc[cl::CC_J2]; stat_ed_jobs += (int64_t)cc[cl::CC_ED] + (int64_t)cc[cl::CC_ED2]; }
  }
  else { R.resize((size_t)nl); for (int64_t l = 0; l < nl; ++l) R[(size_t)l] = l; }
  const int64_t nR = (int64_t)R.size();
  const int32_t* const sp_s = (const int32_t*)h_ss; const int32_t* const sp_e = (const int32_t*)h_se;  // spans (pinned copies)
  // host path, first part: spanning reads of the loci in R and the gather of their repeat segments (second stream), enqueued
  // before anything else so that it does not have to share the GPU with the HMM batch below
  TL("R listed");
  // job lists of stage C for the device-genotyped loci: host-only work, done first because HIP calls made in the ~0.3 ms after the
  // event wait returns block until the runtime has retired stage A's commands
  std::vector<uint32_t> job_set, seq_len; std::vector<uint64_t> seq_off, span_off, count_off; std::vector<int64_t> slot;
  std::vector<uint32_t> nsp; std::vector<double> pur;
  if (dev_gt && use_slots) {
    for (int64_t l = 0; l < nl; ++l) if (!((const uint8_t*)gh.need)[l]) stat_spanning += ((const uint32_t*)gh.nspan)[l];
    stat_hmm_jobs += hmm_slots_resolved(c, hmm_pending, &models, need_a.empty() ? (const uint8_t*)gh.need : need_a.data(), (const int32_t*)gh.nal, (const uint32_t*)gh.alen);
    if (hmm_pendingB) stat_hmm_jobs += hmm_slots_resolved(c, hmm_pendingB, &models, (const uint8_t*)hsl(o_skipb), (const int32_t*)gh.nal, (const uint32_t*)gh.alen);
  } else if (dev_gt) {
    const uint8_t* need = (const uint8_t*)gh.need;
    const int32_t* nal = (const int32_t*)gh.nal; const uint32_t* alen = (const uint32_t*)gh.alen;
    job_set.reserve(2 * (size_t)nl); seq_off.reserve(2 * (size_t)nl); seq_len.reserve(2 * (size_t)nl); span_off.reserve(2 * (size_t)nl);
    count_off.reserve(2 * (size_t)nl); slot.reserve(2 * (size_t)nl);
    for (int64_t l = 0; l < nl; ++l) {
      if (need[l]) continue;
      stat_spanning += ((const uint32_t*)gh.nspan)[l];
      for (int a = 0; a < nal[l]; ++a) {
        job_set.push_back((uint32_t)l); seq_off.push_back(out->allele_off[2 * l + a]); seq_len.push_back(alen[2 * l + a]);
        span_off.push_back(out->span_off[2 * l + a]); count_off.push_back(out->count_off[2 * l + a]); slot.push_back(2 * l + a);
      }
    }
    TL("hmm1 job lists");
  }
  std::vector<LocusWork> work((size_t)nR);
  struct K { uint32_t read, s, e; };
  std::vector<uint64_t> sel_begin((size_t)nR + 1, 0);
  for (int64_t li = 0; li < nR; ++li) sel_begin[(size_t)li + 1] = sel_begin[(size_t)li] + (in->locus_read_begin[R[(size_t)li] + 1] - in->locus_read_begin[R[(size_t)li]]);
  std::vector<K> sel((size_t)sel_begin[(size_t)nR]);
  std::vector<uint32_t> n_sel((size_t)nR, 0);
  std::vector<Scratch> scratch((size_t)pool->size());
  std::vector<uint64_t> seg_src, seg_dst; std::vector<uint32_t> seg_len, seg_read; std::vector<const uint8_t*> seg_ptr;
  uint64_t n_seg = 0, seg_bytes = 0;
  void* h_seg = nullptr;
  if (nR > 0) {
    int64_t th0 = now_ns();
    // pass 1 (parallel over loci): filter (tr.rs:139-145), stable sort by span length (:157), uniform downsample (:172-184)
    pool->parallel_for(nR, 32, [&](int64_t li, int) {
      const int64_t l = R[(size_t)li];
      if (in->ploidy[l] == 0) return;  // Ploidy::Zero -> LocusResult::empty (tr.rs:29-31)
      K* ks = sel.data() + sel_begin[(size_t)li];
      uint32_t n = 0;
      for (uint64_t r = in->locus_read_begin[l]; r < in->locus_read_begin[l + 1]; ++r) {
        const int32_t s = sp_s[r], e = sp_e[r];
        if (s < 0) continue;
        if (s >= F && (int64_t)in->read_len[r] - e >= F) {
          const K kk{(uint32_t)r, (uint32_t)s, (uint32_t)e};
          uint32_t i = n++;  // stable insertion sort by span length
          while (i > 0 && (ks[i - 1].e - ks[i - 1].s) > (kk.e - kk.s)) { ks[i] = ks[i - 1]; --i; }
          ks[i] = kk;
        }
      }
      if ((int64_t)n > p->max_depth) {
        const double step = (double)n / (double)p->max_depth;
        double fast = 0.0;
        for (int i = 0; i < p->max_depth; ++i) { const size_t ind = (size_t)std::floor(fast); if (ind != (size_t)i) std::swap(ks[i], ks[ind]); fast += step; }
        n = (uint32_t)p->max_depth;
      }
      n_sel[(size_t)li] = n;
    });
    // filter_impure_trs (tr.rs:400-452): purity of the repeat segment of every read without a quality >= 0.9 (one HMM batch),
    // stable sort by purity (f64::total_cmp), drop at most max(1, round(0.1 n)) reads below 0.9
    if (impure_filter) {
      std::vector<uint32_t> pj_set, pj_len, pj_nsp; std::vector<uint64_t> pj_off, pj_cnt_off; std::vector<double> pj_pur;
      std::vector<uint64_t> pj_begin((size_t)nR + 1, 0);
      uint64_t cnt_total = 0;
      for (int64_t li = 0; li < nR; ++li) {
        const int64_t l = R[(size_t)li];
        const K* ks = sel.data() + sel_begin[(size_t)li];
        const uint32_t nm = in->set_motif_begin[l + 1] - in->set_motif_begin[l];
        for (uint32_t i = 0; i < n_sel[(size_t)li]; ++i) {
          const double rq = in->read_qual ? in->read_qual[ks[i].read] : std::nan("");
          if (rq >= 0.9) continue;  // Some(rq) with rq >= cutoff keeps purity 1.0; None (NaN) and low qualities are scored
          pj_set.push_back((uint32_t)l); pj_off.push_back(in->read_off[ks[i].read] + ks[i].s); pj_len.push_back(ks[i].e - ks[i].s);
          pj_cnt_off.push_back(cnt_total); cnt_total += nm;
        }
        pj_begin[(size_t)li + 1] = pj_set.size();
      }
      if (!pj_set.empty()) {
        pj_nsp.resize(pj_set.size()); pj_pur.resize(pj_set.size());
        std::vector<uint32_t> pj_counts((size_t)cnt_total + 1);
        rc = hmm_batch_impl(c, &models, (int32_t)nl, in->motif_blob, in->motif_off, in->set_motif_begin, (int64_t)pj_set.size(), pj_set.data(),
                            reads_on_device ? d_reads : in->read_blob, pj_off.data(), pj_len.data(), nullptr, nullptr, nullptr, nullptr, nullptr,
                            pj_nsp.data(), pj_counts.data(), pj_cnt_off.data(), pj_pur.data(), nullptr, nullptr);
        if (rc) return rc;
        stat_hmm_jobs += (int64_t)pj_set.size();
      }
      auto total_key = [](double d) { int64_t b; std::memcpy(&b, &d, 8); b ^= (int64_t)((uint64_t)(b >> 63) >> 1); return b; };
      pool->parallel_for(nR, 32, [&](int64_t li, int) {
        K* ks = sel.data() + sel_begin[(size_t)li];
        const uint32_t n = n_sel[(size_t)li];
        if (n == 0) return;
        std::vector<std::pair<double, K>> pr((size_t)n);
        uint64_t j = pj_begin[(size_t)li];
        for (uint32_t i = 0; i < n; ++i) {
          const double rq = in->read_qual ? in->read_qual[ks[i].read] : std::nan("");
          pr[i] = {rq >= 0.9 ? 1.0 : pj_pur[(size_t)j++], ks[i]};
        }
        std::stable_sort(pr.begin(), pr.end(), [&](const std::pair<double, K>& a, const std::pair<double, K>& b) { return total_key(a.first) < total_key(b.first); });
        const size_t max_filter = std::max<size_t>(1, (size_t)std::round(0.1 * (double)n));
        size_t filtered = 0; uint32_t m = 0;
        for (auto& q : pr) { if (q.first >= 0.9 || filtered >= max_filter) ks[m++] = q.second; else ++filtered; }
        n_sel[(size_t)li] = m;
      });
    }
    TL("R pass 1");
    // pass 2: flat segment arrays (LocusResult.reads order within each locus)
    for (int64_t li = 0; li < nR; ++li) { work[(size_t)li].seg_begin = n_seg; n_seg += n_sel[(size_t)li]; work[(size_t)li].seg_end = n_seg; }
    seg_src.resize((size_t)n_seg); seg_dst.resize((size_t)n_seg); seg_len.resize((size_t)n_seg); seg_read.resize((size_t)n_seg); seg_ptr.resize((size_t)n_seg);
    pool->parallel_for(nR, 64, [&](int64_t li, int) {
      const K* ks = sel.data() + sel_begin[(size_t)li];
      uint64_t s = work[(size_t)li].seg_begin;
      for (uint32_t i = 0; i < n_sel[(size_t)li]; ++i, ++s) {
        seg_read[s] = ks[i].read; seg_src[s] = in->read_off[ks[i].read] + ks[i].s; seg_len[s] = ks[i].e - ks[i].s;
      }
    });
    for (uint64_t s = 0; s < n_seg; ++s) { seg_dst[s] = seg_bytes; seg_bytes += seg_len[s]; }
    stat_spanning += (int64_t)n_seg;
    c->dbg_ns[4] = now_ns() - th0;
    TL("R pass 2");
    if (n_seg > 0 && reads_on_device) {
      // source offsets, destination offsets and lengths: the gather kernel reads them straight from pinned host memory (20 B per
      // segment over PCIe).  A hipMemcpyAsync of them took 0.37 ms on the host however small it was.
      void *d_out, *h_meta;
      const size_t meta_bytes = (size_t)n_seg * 20;
      if ((rc = dev_get(c, S_LOCUS_3, (size_t)seg_bytes + 1, &d_out)) ||
          (rc = pin_get(c, P_SEG0, (size_t)seg_bytes + 1, &h_seg)) || (rc = pin_get(c, P_SEG_META, meta_bytes, &h_meta)))
        return rc;
      std::memcpy(h_meta, seg_src.data(), (size_t)n_seg * 8);
      std::memcpy((uint8_t*)h_meta + (size_t)n_seg * 8, seg_dst.data(), (size_t)n_seg * 8);
      std::memcpy((uint8_t*)h_meta + (size_t)n_seg * 16, seg_len.data(), (size_t)n_seg * 4);
      TL("R meta staged");
      void* const d_src = h_meta; void* const d_dst = (uint8_t*)h_meta + (size_t)n_seg * 8; void* const d_len = (uint8_t*)h_meta + (size_t)n_seg * 16;
      GatherArgs ga{d_reads, (const uint64_t*)d_src, (const uint64_t*)d_dst, (const uint32_t*)d_len, (uint64_t)n_seg, (uint8_t*)d_out};
      hipLaunchKernelGGL(gather_segments_kernel, dim3((unsigned)((n_seg + 3) / 4)), dim3(256), 0, c->stream2, ga);
      TRGT_HIP_TRY(c, hipGetLastError());
      TL("R gather launched");
      { const int d2h_rc = trgt::d2h(c, h_seg, d_out, (size_t)seg_bytes, c->stream2); if (d2h_rc) return d2h_rc; }
    }
  }
  TL("R selected, gather enqueued");
  // ---------------- the alleles of the device-genotyped loci come back packed: a second, exact-size copy (second stream)
  if (dev_gt) {
    const uint64_t packed_total = ((const uint64_t*)gh.toff)[2 * nl];
    if ((rc = pin_get(c, P_GT_PACKED, (size_t)packed_total + 16, &gh.packed))) return rc;
    if (packed_total) { const int d2h_rc = trgt::d2h(c, gh.packed, g.packed, (size_t)packed_total, c->stream2); if (d2h_rc) return d2h_rc; }
  }
  // ---------------- stage C for the device-genotyped loci (on alleles that already sit in HBM; collected at the end) and the
  // publishing of spans and device-genotyper results: host work of ~1.2 ms that needs the GPU only to start the HMM batch.  It runs
  // while the consensus alignments of the host-path loci are on the GPU (wfa_batch_impl calls it between launch and wait), or
  // right here when there are none.
  // the two device-resolved HMM batches: the first one's counts / purities arrive as whole arrays ("no allele" in every slot that was not
  // its job), so it is collected first and the second one's are merged in slot by slot
  auto collect_ab = [&]() -> int {
    if (!hmm_pendingB) return TRGT_OK;  // (a single batch is collected where it always was, below)
    const int64_t tc0 = now_ns();
    if (hmm_pending) {
      HmmPending* pend = hmm_pending; hmm_pending = nullptr;
      if (int r = hmm_collect(c, pend)) return r;
      const uint8_t* need = (const uint8_t*)gh.need;
      for (int64_t l = 0; l < nl; ++l) if (need[l]) { out->n_spans[2 * l] = out->n_spans[2 * l + 1] = 0; out->purity[2 * l] = out->purity[2 * l + 1] = std::nan(""); }
      TL("hmm1 collected");
    }
    {
      HmmPending* pend = hmm_pendingB; hmm_pendingB = nullptr;
      if (int r = hmm_collect(c, pend)) return r;
      const uint8_t* skip_b = (const uint8_t*)hsl(o_skipb); const int32_t* nal = (const int32_t*)gh.nal;
      for (int64_t l = 0; l < nl; ++l)
        if (!skip_b[l]) for (int a = 0; a < nal[l]; ++a) { out->n_spans[2 * l + a] = nsB[(size_t)(2 * l + a)]; out->purity[2 * l + a] = puB[(size_t)(2 * l + a)]; }
      TL("hmm1b collected");
    }
    tC += now_ns() - tc0;
    return TRGT_OK;
  };
  bool published = false;
  auto hmm1_enqueue = [&]() -> int {
  if (dev_gt && !use_slots) {
    if (!job_set.empty()) {
      nsp.resize(job_set.size()); pur.resize(job_set.size());
      const int64_t tc0 = now_ns();
      rc = hmm_enqueue(c, &models, (int32_t)nl, in->motif_blob, in->motif_off, in->set_motif_begin, (int64_t)job_set.size(), job_set.data(),
                       (const uint8_t*)g.blob, seq_off.data(), seq_len.data(), nullptr, nullptr, nullptr, out->spans3, span_off.data(), nsp.data(),
                       out->motif_counts, count_off.data(), pur.data(), nullptr, nullptr, &hmm_pending);
      if (rc) return rc;
      tC += now_ns() - tc0;
      TL("hmm1 enqueued");
      stat_hmm_jobs += (int64_t)job_set.size();
    }
  }
  return TRGT_OK;
  };
  // (enqueued from the callback below when there are consensus alignments: its ~0.7 ms of host work then runs next to that kernel
  //  instead of in front of the host path that leads to it)
  bool hmm1_done = false;
  auto hmm1_once = [&]() -> int {  // on the first stream, whichever stream is current
    if (hmm1_done) return TRGT_OK;
    hmm1_done = true;
    const bool swapped = c->stream == upload_stream;
    if (swapped) std::swap(c->stream, c->stream2);
    const int r = hmm1_enqueue();
    if (swapped) std::swap(c->stream, c->stream2);
    return r;
  };
  if (nR == 0 && (rc = hmm1_once())) return rc;
  if (dev_gt || (n_seg > 0 && reads_on_device)) TRGT_HIP_TRY(c, trgt::stream_wait(c, c->stream2));  // gathered segments, packed alleles
  tHost += now_ns() - th_begin;
  TL("stream2 synced");
  // ---------------- publish spans and the device genotyper's results: host-only work, done while the consensus alignments of the
  // host-path loci are on the GPU (wfa_batch_impl calls it between launch and wait), or at the end when there are none
  auto publish = [&]() -> int {
  published = true;
  TL("publish begins");
  pool->parallel_for(8, 1, [&](int64_t part8, int) {  // ~6 MB of result arrays: spread the copies over a few threads
    auto piece = [&](void* dst, const void* src, size_t bytes, int64_t k, int64_t n) {
      const size_t b = bytes * (size_t)k / (size_t)n, e = bytes * (size_t)(k + 1) / (size_t)n;
      std::memcpy((uint8_t*)dst + b, (const uint8_t*)src + b, e - b);
    };
    piece(out->span_start, h_ss, (size_t)nr * 4, part8, 8);
    piece(out->span_end, h_se, (size_t)nr * 4, part8, 8);
    if (dev_gt) {
      piece(out->classification, gh.cls, (size_t)nr * 4, part8, 8); piece(out->read_rank, gh.rank, (size_t)nr * 4, part8, 8);
      piece(out->n_alleles, gh.nal, (size_t)nl * 4, part8, 8); piece(out->allele_len, gh.alen, 2 * (size_t)nl * 4, part8, 8);
      piece(out->ci, gh.ci, 4 * (size_t)nl * 4, part8, 8); piece(out->num_spanning, gh.nsp, 2 * (size_t)nl * 4, part8, 8);
    }
  });
  TL("publish: arrays copied");
  {
    std::vector<int64_t> part((size_t)pool->size() * 8, 0);
    pool->parallel_for(nr, 8192, [&](int64_t r, int t) {
      const int64_t n = (((const uint8_t*)h_hl)[r] != 1) + (((const uint8_t*)h_hr)[r] != 1);
      part[(size_t)t * 8] += n;
      if (in->read_len[r] < heavy_len[read_locus[(size_t)r]]) part[(size_t)t * 8 + 1] += n;  // alignments of the first (dominant) launch
    });
    for (int t = 0; t < pool->size(); ++t) { stat_flank_jobs += part[(size_t)t * 8]; stat_flank_heavy += part[(size_t)t * 8 + 1]; }
  }
  if (c->timing) {  // [0] all flank alignments, [1] those of the first (dominant) launch
    c->k_cells[TRGT_K_WFA_FLANK] += (int64_t)((uint64_t*)h_cells)[1];
    c->k_cells[TRGT_K_WFA_FLANK_REST] += (int64_t)(((uint64_t*)h_cells)[0] - ((uint64_t*)h_cells)[1]);
    c->k_cells[TRGT_K_WFA_FILTER] += (int64_t)((uint64_t*)h_cells)[4];
    c->k_cells[TRGT_K_WFA] += (int64_t)((uint64_t*)h_cells)[6];
  }
  if (dev_gt) {
    const uint8_t* need = (const uint8_t*)gh.need;
    const uint64_t* toff = (const uint64_t*)gh.toff; const uint8_t* packed = (const uint8_t*)gh.packed;
    const uint8_t* dev_flip = (const uint8_t*)hsl(o_flip);
    const int32_t* dev_gsz = (const int32_t*)hsl(o_gsz);
    pool->parallel_for(nl, 256, [&](int64_t l, int) {
      if (out->flipped) out->flipped[l] = need[l] ? 0 : dev_flip[l];
      // (alleles the device genotyper settles have majority support: their length is the genotype's size)
      if (out->gt_size) { out->gt_size[2 * l] = need[l] ? 0 : dev_gsz[2 * l]; out->gt_size[2 * l + 1] = need[l] ? 0 : dev_gsz[2 * l + 1]; }
      if (need[l]) { out->n_alleles[l] = 0; out->allele_len[2 * l] = out->allele_len[2 * l + 1] = 0; return; }
      for (int a = 0; a < out->n_alleles[l]; ++a)
        std::memcpy(out->allele_blob + out->allele_off[2 * l + a], packed + toff[2 * l + a], out->allele_len[2 * l + a]);
    });
  }
  TL("published");
  return TRGT_OK;
  };

  // ---------------- host path for the loci in R, second part (second stream for its GPU work: consensus alignments)
  if (nR > 0) {
    int64_t th0 = now_ns();
    if (n_seg > 0 && reads_on_device) { for (uint64_t s = 0; s < n_seg; ++s) seg_ptr[s] = (const uint8_t*)h_seg + seg_dst[s]; }
    else { for (uint64_t s = 0; s < n_seg; ++s) seg_ptr[s] = in->read_blob + seg_src[s]; }
    // front half of the length genotyper, threaded over loci (per-thread scratch, no per-locus allocation)
    int64_t tf0 = now_ns();
    std::vector<ClusterLocus> cl_loci;
    std::vector<std::vector<Seg>> cl_trs;
    for (int64_t li = 0; li < nR; ++li) {
      LocusWork& w = work[(size_t)li];
      if (w.seg_begin == w.seg_end || !is_cluster(R[(size_t)li])) continue;
      w.cluster = (int)cl_loci.size();
      ClusterLocus L; L.li = li; L.ploidy = in->ploidy[R[(size_t)li]] == 1 ? 1 : 2; L.n = (int)(w.seg_end - w.seg_begin);
      cl_loci.push_back(std::move(L));
      cl_trs.emplace_back();
      for (uint64_t s = w.seg_begin; s < w.seg_end; ++s) cl_trs.back().push_back(Seg{seg_ptr[s], seg_len[s]});
    }
    for (size_t k = 0; k < cl_loci.size(); ++k) cl_loci[k].trs = cl_trs[k].data();
    pool->parallel_for(nR, 16, [&](int64_t li, int t) {
      LocusWork& w = work[(size_t)li];
      if (w.seg_begin == w.seg_end || w.cluster >= 0) return;
      Scratch& sc = scratch[(size_t)t];
      sc.trs.clear();
      for (uint64_t s = w.seg_begin; s < w.seg_end; ++s) sc.trs.push_back(Seg{seg_ptr[s], seg_len[s]});
      genotype_size_front(in->ploidy[R[(size_t)li]] == 1 ? 1 : 2, li, t, w, sc);
    });
    c->dbg_ns[6] = now_ns() - tf0;
  TL("front");
    tHost += now_ns() - th0;
    // ---- stage B: consensus alignments (BiWFA, affine 2,5,1, default heuristic) for the loci that need them
    std::swap(c->stream, c->stream2);
    struct SwapBack { trgt_hip_ctx* c; ~SwapBack() { std::swap(c->stream, c->stream2); } } swap_back{c};
    int64_t tb0 = now_ns();
    struct JobRef { Repair* rep; int member; };
    std::vector<JobRef> jrefs;
    std::vector<uint8_t> cblob; std::vector<uint64_t> poff, toff; std::vector<uint32_t> plen, tlen;
    for (auto& sc : scratch)
      for (auto& rep : sc.repairs) {
        const Seg bb = work[(size_t)rep.locus].pick[rep.allele];
        const uint64_t bo = cblob.size();
        cblob.insert(cblob.end(), bb.p, bb.p + bb.n);
        for (size_t m = 0; m < rep.members.size(); ++m) {
          const Seg& sg = rep.members[m];
          poff.push_back(bo); plen.push_back(bb.n);
          toff.push_back(cblob.size()); tlen.push_back(sg.n);
          cblob.insert(cblob.end(), sg.p, sg.p + sg.n);
          jrefs.push_back({&rep, (int)m});
        }
      }
    std::vector<size_t> rep_first{0};  // jobs of one repair are contiguous
    std::vector<Repair*> rep_of;
    for (size_t j = 0; j < jrefs.size(); ++j)
      if (jrefs[j].member == 0) { if (j) rep_first.push_back(j); rep_of.push_back(jrefs[j].rep); }
    if (!jrefs.empty()) rep_first.push_back(jrefs.size());
    if (jrefs.empty() && ((rc = hmm1_once()) || (!published && (rc = publish())))) return rc;
    if (!jrefs.empty()) {
      const std::function<int()> overlap = [&]() -> int {  // next to the alignment kernel: start the HMM batch, publish results
        const int r = hmm1_once();
        return r ? r : publish();
      };
      std::vector<std::string> repaired;
      rc = consensus_repair_batch(c, (int64_t)jrefs.size(), cblob.data(), poff.data(), plen.data(), toff.data(), tlen.data(), rep_first, repaired, &overlap);
      if (rc) return rc;
      for (size_t g = 0; g < rep_of.size(); ++g) rep_of[g]->result.swap(repaired[g]);
      stat_cons_jobs += (int64_t)jrefs.size();
    }
    // ---- Genotyper::Cluster loci: distance matrix, Ward linkage, consensus rounds, outlier assignment (locus_cluster.hpp)
    if (!cl_loci.empty()) {
      ClusterBatch cb(c, pool, cl_loci);
      if ((rc = cb.run())) return rc;
      stat_cons_jobs += cb.n_cons;
      stat_ed_jobs += cb.n_ed;
    }
    tB = now_ns() - tb0;
  TL("stageB");
    // ---- host: repair_consensus, classification, reference allele first, output assembly
    th0 = now_ns();
    // ---- genotype_flank::genotype (genotype_flank.rs:9-42) for the loci whose two alleles are at most 10 bases apart (tr.rs:69-75)
    struct FlankRes { bool on = false; std::string repaired[2]; Seg al[2]; uint32_t ci[4] = {0, 0, 0, 0}; std::vector<int8_t> assignment; int rep_group[2] = {-1, -1}; };
    std::vector<FlankRes> fres((size_t)(flank_on ? nR : 0));
    if (flank_on) {
      struct FlankRepair { int64_t li; int a; Seg backbone; std::vector<Seg> members; };
      std::vector<std::vector<FlankRepair>> freps((size_t)pool->size());
      pool->parallel_for(nR, 16, [&](int64_t li, int t) {
        LocusWork& w = work[(size_t)li];
        if (w.seg_begin == w.seg_end) return;
        int n_gt; uint32_t s0, s1;  // Gt sizes: the length genotyper's sizes, the cluster genotyper's allele lengths
        if (w.cluster >= 0) { const ClusterLocus& L = cl_loci[(size_t)w.cluster]; n_gt = L.n_gt; s0 = (uint32_t)L.allele[0].size(); s1 = (uint32_t)L.allele[1].size(); }
        else { n_gt = w.n_gt; s0 = w.size[0]; s1 = w.size[1]; }
        if (n_gt != 2 || adiff(s0, s1) > 10) return;
        std::vector<uint32_t> rd;
        for (uint64_t s = w.seg_begin; s < w.seg_end; ++s) rd.push_back(seg_read[s]);
        FlankSplit sp;
        if (!flank_split(flank_meta, rd.data(), rd.size(), sp)) return;
        FlankRes fr;
        for (int g = 0; g < 2; ++g) {
          std::vector<Seg> seqs;
          uint32_t lo = 0xFFFFFFFFu, hi = 0;
          for (uint32_t i : sp.group[g]) { const uint64_t s = w.seg_begin + i; seqs.push_back(Seg{seg_ptr[s], seg_len[s]}); lo = std::min(lo, seg_len[s]); hi = std::max(hi, seg_len[s]); }
          Seg best; double freq;
          if (!flank_simple_consensus(seqs, best, freq)) return;  // an empty group: None
          fr.al[g] = best; fr.ci[2 * g] = lo; fr.ci[2 * g + 1] = hi;
          if (freq < 0.5) { fr.rep_group[g] = (int)freps[(size_t)t].size(); freps[(size_t)t].push_back(FlankRepair{li, g, best, std::move(seqs)}); }
        }
        fr.on = true; fr.assignment = std::move(sp.assignment);
        fres[(size_t)li] = std::move(fr);
      });
      // groups without a majority sequence: align(backbone, members) + repair_consensus, one more batch on the device
      std::vector<uint8_t> fb; std::vector<uint64_t> fpo, fto; std::vector<uint32_t> fpl, ftl; std::vector<size_t> ffirst{0};
      std::vector<std::pair<int64_t, int>> fwho;
      for (auto& v : freps)
        for (auto& q : v) {
          const uint64_t bo = fb.size();
          fb.insert(fb.end(), q.backbone.p, q.backbone.p + q.backbone.n);
          for (auto& m : q.members) { fpo.push_back(bo); fpl.push_back(q.backbone.n); fto.push_back(fb.size()); ftl.push_back(m.n); fb.insert(fb.end(), m.p, m.p + m.n); }
          ffirst.push_back(fpo.size()); fwho.push_back({q.li, q.a});
        }
      if (!fwho.empty()) {
        std::vector<std::string> repaired;
        if ((rc = consensus_repair_batch(c, (int64_t)fpo.size(), fb.data(), fpo.data(), fpl.data(), fto.data(), ftl.data(), ffirst, repaired))) return rc;
        stat_cons_jobs += (int64_t)fpo.size();
        for (size_t g = 0; g < fwho.size(); ++g) {
          FlankRes& fr = fres[(size_t)fwho[g].first];
          fr.repaired[fwho[g].second].swap(repaired[g]);
        }
      }
      for (auto& fr : fres) {
        if (!fr.on) continue;
        for (int g = 0; g < 2; ++g) if (fr.rep_group[g] >= 0) fr.al[g] = Seg{(const uint8_t*)fr.repaired[g].data(), (uint32_t)fr.repaired[g].size()};
        if (fr.al[0].n > fr.al[1].n) {  // smaller allele first (:33-38)
          std::swap(fr.al[0], fr.al[1]); std::swap(fr.ci[0], fr.ci[2]); std::swap(fr.ci[1], fr.ci[3]);
          for (auto& a : fr.assignment) a = (int8_t)((a + 1) % 2);
        }
      }
    }
    std::vector<int8_t> seg_cls((size_t)n_seg, 0);
    std::atomic<int> bad{0};
    pool->parallel_for(nR, 64, [&](int64_t li, int) {
      const int64_t l = R[(size_t)li];
      LocusWork& w = work[(size_t)li];
      if (w.seg_begin == w.seg_end) return;
      const int ploidy = in->ploidy[l] == 1 ? 1 : 2;
      Seg al[2];
      int by_hap[2] = {0, 0};
      if (w.cluster >= 0) {  // genotype_cluster::genotype results
        const ClusterLocus& L = cl_loci[(size_t)w.cluster];
        w.n_gt = L.n_gt;
        for (int a = 0; a < L.n_gt; ++a) { al[a] = Seg{(const uint8_t*)L.allele[a].data(), (uint32_t)L.allele[a].size()}; w.ci[2 * a] = L.ci[2 * a]; w.ci[2 * a + 1] = L.ci[2 * a + 1]; }
        for (uint64_t s = w.seg_begin; s < w.seg_end; ++s) { const int cc = L.cls[(size_t)(s - w.seg_begin)]; seg_cls[s] = (int8_t)cc; by_hap[cc] += 1; }
      } else {
        for (int a = 0; a < w.n_pick; ++a) {
          if (w.repair[a] >= 0) { const std::string& r = scratch[(size_t)w.repair_thread].repairs[(size_t)w.repair[a]].result; al[a] = Seg{(const uint8_t*)r.data(), (uint32_t)r.size()}; }
          else al[a] = w.pick[a];
        }
        int n_al = w.n_pick;
        if (ploidy == 2 && n_al == 1) { al[1] = al[0]; n_al = 2; }
        int tie = 1;
        for (uint64_t s = w.seg_begin; s < w.seg_end; ++s) {
          int cc = 0;
          if (n_al == 2) {
            const uint32_t d1 = adiff(seg_len[s], al[0].n), d2 = adiff(seg_len[s], al[1].n);
            if (d1 < d2) cc = 0; else if (d1 > d2) cc = 1; else { tie = (tie + 1) % 2; cc = tie; }
          }
          seg_cls[s] = (int8_t)cc; by_hap[cc] += 1;
        }
      }
      // TrSize::size of the genotype: the length genotyper's sizes; the cluster genotyper's and the flank genotype's are allele lengths
      uint32_t gsz[2] = {w.cluster >= 0 ? al[0].n : w.size[0], w.cluster >= 0 ? (w.n_gt > 1 ? al[1].n : 0u) : w.size[1]};
      if (flank_on && fres[(size_t)li].on) {  // the flank genotype replaces alleles, intervals and the read assignment
        const FlankRes& fr = fres[(size_t)li];
        w.n_gt = 2; al[0] = fr.al[0]; al[1] = fr.al[1]; gsz[0] = al[0].n; gsz[1] = al[1].n;
        for (int k = 0; k < 4; ++k) w.ci[k] = fr.ci[k];
        by_hap[0] = by_hap[1] = 0;
        for (uint64_t s = w.seg_begin; s < w.seg_end; ++s) { const int cc = fr.assignment[(size_t)(s - w.seg_begin)]; seg_cls[s] = (int8_t)cc; by_hap[cc] += 1; }
      }
      int order[2] = {0, 1};
      const Seg ref{in->tr_blob + in->tr_off[l], in->tr_len[l]};
      bool flip = false;
      if (w.n_gt != 1 && !eq_seg(al[0], ref) && eq_seg(al[1], ref)) { order[0] = 1; order[1] = 0; flip = true; }  // tr.rs:95-101
      out->n_alleles[l] = w.n_gt;
      for (int oi = 0; oi < w.n_gt; ++oi) {
        const int a = order[oi];
        if (al[a].n > out->allele_cap[l]) { bad = 1; return; }
        std::memcpy(out->allele_blob + out->allele_off[2 * l + oi], al[a].p, al[a].n);
        out->allele_len[2 * l + oi] = al[a].n;
        out->ci[4 * l + 2 * oi] = (int32_t)w.ci[2 * a]; out->ci[4 * l + 2 * oi + 1] = (int32_t)w.ci[2 * a + 1];
        out->num_spanning[2 * l + oi] = by_hap[a];
        if (out->gt_size) out->gt_size[2 * l + oi] = (int32_t)gsz[a];
      }
      if (out->flipped) out->flipped[l] = flip ? 1 : 0;
      for (uint64_t r = in->locus_read_begin[l]; r < in->locus_read_begin[l + 1]; ++r) { out->classification[r] = -1; out->read_rank[r] = -1; }
      for (uint64_t s = w.seg_begin; s < w.seg_end; ++s) {
        out->classification[seg_read[s]] = flip ? 1 - seg_cls[s] : seg_cls[s];
        out->read_rank[seg_read[s]] = (int32_t)(s - w.seg_begin);
      }
    });
    if (bad) return fail(c, TRGT_ERR_INVALID, "trgt_locus_batch: allele_cap too small");
    c->dbg_ns[7] = now_ns() - th0;
  TL("back");
    tHost += now_ns() - th0;
    // ---- stage C for the host-path loci (label_with_hmm for every allele): enqueued here, on the second stream and the second set
    // of HMM buffers, next to the batch of the device-genotyped loci that is still running
    const int64_t tc0 = now_ns();
    for (int64_t li = 0; li < nR; ++li) {
      const int64_t l = R[(size_t)li];
      for (int a = 0; a < out->n_alleles[l]; ++a) {
        js2.push_back((uint32_t)l); so2.push_back(out->allele_off[2 * l + a]); sl2.push_back(out->allele_len[2 * l + a]);
        spo2.push_back(out->span_off[2 * l + a]); co2.push_back(out->count_off[2 * l + a]); slot2.push_back(2 * l + a);
      }
    }
    if (!js2.empty()) {
      if ((rc = collect_ab())) return rc;  // (the second device-resolved batch holds buffer set 1)
      ns2.resize(js2.size()); pu2.resize(js2.size());
      // next to the batch of the device-genotyped loci, if there is one: the call's second stream and the second set of side streams
      // (the two batches are independent; queued on the same streams the second one waited for the first -- config 3: 65 -> 5x ms per call)
      struct StreamSwap { trgt_hip_ctx* c; hipStream_t saved; ~StreamSwap() { c->stream = saved; } } hmm2_stream{c, c->stream};
      if (hmm_pending) c->stream = c->stream2;
      rc = hmm_enqueue(c, &models, (int32_t)nl, in->motif_blob, in->motif_off, in->set_motif_begin, (int64_t)js2.size(), js2.data(),
                       out->allele_blob, so2.data(), sl2.data(), nullptr, nullptr, nullptr, out->spans3, spo2.data(), ns2.data(),
                       out->motif_counts, co2.data(), pu2.data(), nullptr, nullptr, &hmm_pending2, hmm_pending ? 1 : 0);
      if (rc) return rc;
      stat_hmm_jobs += (int64_t)js2.size();
    }
    tC += now_ns() - tc0;
  }
  if ((rc = hmm1_once())) return rc;
  if (!published && (rc = publish())) return rc;
  // ---------------- stage C results of the device-genotyped loci
  TL("hmm2 enqueued");
  if ((rc = collect_ab())) return rc;
  if (hmm_pending) {
    const int64_t tc0 = now_ns();
    HmmPending* pend = hmm_pending; hmm_pending = nullptr;
    rc = hmm_collect(c, pend);
    if (rc) return rc;
    TL("hmm1 collected");
    if (use_slots) {  // (the whole arrays came back: the slots of the loci that took the host path hold "no allele" until stage C of those)
      const uint8_t* need = (const uint8_t*)gh.need;
      for (int64_t l = 0; l < nl; ++l) if (need[l]) { out->n_spans[2 * l] = out->n_spans[2 * l + 1] = 0; out->purity[2 * l] = out->purity[2 * l + 1] = std::nan(""); }
    }
    for (size_t j = 0; j < slot.size(); ++j) { out->n_spans[slot[j]] = nsp[j]; out->purity[slot[j]] = pur[j]; }
    tC += now_ns() - tc0;
  }
  if (hmm_pending2) {
    const int64_t tc0 = now_ns();
    HmmPending* pend = hmm_pending2; hmm_pending2 = nullptr;
    rc = hmm_collect(c, pend);
    if (rc) return rc;
    for (size_t j = 0; j < slot2.size(); ++j) { out->n_spans[slot2[j]] = ns2[j]; out->purity[slot2[j]] = pu2[j]; }
    tC += now_ns() - tc0;
  }
  TL("all collected");
  if (c->timing) { resolve_timing(c); TL("timing events resolved"); }  // (all of the call's events have fired: reading them now keeps the list short)
  if (out->stats) {
    int64_t* s = out->stats;
    s[0] = stat_flank_jobs; s[1] = stat_cons_jobs; s[2] = stat_spanning; s[3] = stat_hmm_jobs;
    s[4] = tA; s[5] = tB; s[6] = tC; s[7] = tHost; s[8] = now_ns() - t0;
    for (int i = 0; i < 5; ++i) s[9 + i] = c->dbg_ns[i + (i >= 3 ? 1 : 0)];
    s[14] = stat_flank_heavy; s[15] = stat_ed_jobs;
    s[16] = (int64_t)((uint64_t*)h_cells)[5]; s[17] = (int64_t)((uint64_t*)h_cells)[4];  // pre-filter: alignments kept, offsets computed
    for (int i = 18; i < 24; ++i) s[i] = 0;
    s[21] = (int64_t)((uint64_t*)h_cells)[2];  // light fallback alignments settled by the substitution shortcut of the window search
    if (dev_gt) { const uint32_t* rc_ = (const uint32_t*)hsl(o_rpc); s[18] = rc_[gt::RC_LOCI]; s[19] = rc_[gt::RC_FAILED]; s[20] = rc_[gt::RC_JOBS];  // device-side consensus repair: loci, loci without room, alignments
      const uint32_t* cc = (const uint32_t*)hsl(o_clc); s[22] = cc[cl::CC_DONE]; s[23] = cc[cl::CC_FAILED]; }  // device-side cluster genotyper: loci genotyped, loci handed to the host path
  }
  return TRGT_OK;
}

// ---- C ABI: exceptions never cross it (std::bad_alloc from the host-side vectors becomes TRGT_ERR_NOMEM) ----
#define TRGT_ABI_GUARD(ctx, call)                                                                  \
  try { return (call); }                                                                           \
  catch (const std::bad_alloc&) { return trgt::fail((ctx), TRGT_ERR_NOMEM, "out of host memory"); } \
  catch (const std::exception& e) { return trgt::fail((ctx), TRGT_ERR_INVALID, "unexpected exception: %s", e.what()); }

extern "C" int trgt_locus_batch(trgt_hip_ctx* c, const trgt_locus_params* p, const trgt_locus_batch_in* in, trgt_locus_batch_out* out) {
  TRGT_ABI_GUARD(c, locus_batch_run(c, p, in, out, nullptr, nullptr, nullptr));
}

extern "C" int64_t trgt_reads_pack_bam4(const uint8_t* ascii, int64_t n_reads, const uint64_t* read_off, const uint32_t* read_len,
                                        uint8_t* packed, uint64_t* packed_off) {
  if (n_reads < 0 || (n_reads > 0 && (!ascii || !read_off || !read_len || !packed || !packed_off))) return TRGT_ERR_INVALID;
  uint8_t code[256];
  std::memset(code, 15, sizeof code);  // N
  const char* letters = "=ACMGRSVTWYHKDBN";
  for (int k = 0; k < 16; ++k) { code[(uint8_t)letters[k]] = (uint8_t)k; code[(uint8_t)std::tolower(letters[k])] = (uint8_t)k; }
  uint64_t o = 0;
  for (int64_t r = 0; r < n_reads; ++r) {
    const uint8_t* s = ascii + read_off[r];
    const uint32_t n = read_len[r];
    packed_off[r] = o;
    uint8_t* d = packed + o;
    for (uint32_t i = 0; i + 1 < n; i += 2) d[i / 2] = (uint8_t)((code[s[i]] << 4) | code[s[i + 1]]);
    if (n & 1) d[n / 2] = (uint8_t)(code[s[n - 1]] << 4);
    o += ((uint64_t)n + 1) / 2;
  }
  return (int64_t)o;
}

static int locus_submit(trgt_hip_ctx* c, const trgt_locus_params* p, const trgt_locus_batch_in* in, trgt_locus_batch_out* out, int64_t* ticket) {
  if (!c) return TRGT_ERR_INVALID;
  if (!p || !in || !out || !ticket) return fail(c, TRGT_ERR_INVALID, "trgt_locus_batch_submit: null argument");
  if (in->n_loci < 0 || (in->n_loci > 0 && (!in->flank_blob || !in->read_blob || !in->lf_off || !in->lf_len || !in->rf_off || !in->rf_len ||
                                            !in->locus_read_begin || !in->read_off || !in->read_len)))
    return fail(c, TRGT_ERR_INVALID, "trgt_locus_batch_submit: null field");
  int slot = -1;
  for (int i = 0; i < 2; ++i) if (!c->staged[i].in_use) { slot = i; break; }
  if (slot < 0) return fail(c, TRGT_ERR_INVALID, "trgt_locus_batch_submit: two batches are outstanding already (wait for the older one first)");
  TRGT_HIP_TRY(c, hipSetDevice(c->device));
  trgt_hip_ctx::Staged& st = c->staged[slot];
  st.params = *p; st.in = in; st.out = out; st.d_reads = nullptr; st.d_flank = nullptr; st.copy_pending = false;
  const int64_t nl = in->n_loci;
  const int64_t nr = nl > 0 ? (int64_t)in->locus_read_begin[nl] : 0;
  if (nl > 0 && nr > 0) {
    if (!c->stream_copy) TRGT_HIP_TRY(c, trgt::make_stream(c, &c->stream_copy));
    if (!st.ready) TRGT_HIP_TRY(c, hipEventCreateWithFlags(&st.ready, hipEventDisableTiming));
    uint64_t flank_total = 0, read_total = 0;
    for (int64_t l = 0; l < nl; ++l) flank_total = std::max<uint64_t>(flank_total, std::max(in->lf_off[l] + in->lf_len[l], in->rf_off[l] + in->rf_len[l]));
    const bool bam4 = in->read_encoding == TRGT_READS_BAM4;  // two bases per byte
    for (int64_t r = 0; r < nr; ++r) read_total = std::max<uint64_t>(read_total, in->read_off[r] + (bam4 ? ((uint64_t)in->read_len[r] + 1) / 2 : (uint64_t)in->read_len[r]));
    int rc;
    st.read_bytes = read_total; st.flank_bytes = flank_total;
    if (!is_device_ptr(in->read_blob)) {
      void* d = nullptr;
      if ((rc = dev_get(c, slot ? S_PF_READS1 : S_PF_READS0, (size_t)read_total, &d))) return rc;
      st.d_reads = (const uint8_t*)d;
    }
    if (!is_device_ptr(in->flank_blob)) {
      void* d = nullptr;
      if ((rc = dev_get(c, slot ? S_PF_FLANK1 : S_PF_FLANK0, (size_t)flank_total, &d))) return rc;
      st.d_flank = (const uint8_t*)d;
    }
    st.copy_pending = st.d_reads || st.d_flank;
    st.in_use = true;
    // with another batch outstanding the copy waits for that batch's call (it is issued behind its tables); else it starts now
    if (!c->staged[slot ^ 1].in_use && (rc = issue_pending_uploads(c))) { st.in_use = false; return rc; }
  }
  st.in_use = true; st.ticket = c->next_ticket++;
  *ticket = st.ticket;
  return TRGT_OK;
}

static int locus_wait(trgt_hip_ctx* c, int64_t ticket) {
  if (!c) return TRGT_ERR_INVALID;
  int slot = -1;
  for (int i = 0; i < 2; ++i) if (c->staged[i].in_use && c->staged[i].ticket == ticket) slot = i;
  if (slot < 0) return fail(c, TRGT_ERR_INVALID, "trgt_locus_batch_wait: unknown ticket %lld", (long long)ticket);
  for (int i = 0; i < 2; ++i)
    if (c->staged[i].in_use && c->staged[i].ticket < ticket) return fail(c, TRGT_ERR_INVALID, "trgt_locus_batch_wait: ticket %lld was submitted earlier and must be waited for first", (long long)c->staged[i].ticket);
  trgt_hip_ctx::Staged& st = c->staged[slot];
  const bool staged_any = st.d_reads || st.d_flank;
  if (st.copy_pending) { const int prc = issue_pending_uploads(c); if (prc) return prc; }  // (its own copy first, should it still be pending)
  const int64_t tw0 = now_ns();
  const int rc = locus_batch_run(c, &st.params, st.in, st.out, st.d_reads, st.d_flank, staged_any ? st.ready : nullptr);
  if (c->knobs.timeline) fprintf(stderr, "[tl] trgt_locus_batch_wait: %.2f ms in all\n", (double)(now_ns() - tw0) / 1e6);
  st.in_use = false;
  return rc;
}

extern "C" int trgt_locus_batch_submit(trgt_hip_ctx* c, const trgt_locus_params* p, const trgt_locus_batch_in* in, trgt_locus_batch_out* out, int64_t* ticket) {
  TRGT_ABI_GUARD(c, locus_submit(c, p, in, out, ticket));
}
extern "C" int trgt_locus_batch_wait(trgt_hip_ctx* c, int64_t ticket) { TRGT_ABI_GUARD(c, locus_wait(c, ticket)); }


// ---- several contexts, one queue of batches ------------------------------------------------------------------------------
struct trgt_hip_pool {
  std::vector<trgt_hip_ctx*> ctx;
  std::string err;
};

extern "C" int trgt_hip_pool_create(const int32_t* devices, int32_t n_contexts, trgt_hip_pool** out) {
  if (!devices || n_contexts < 1 || !out) return TRGT_ERR_INVALID;
  *out = nullptr;
  try {
    std::unique_ptr<trgt_hip_pool> P(new trgt_hip_pool());
    const char* pe = TRGT_DEV_ENV("TRGT_POOL_PRIORITIES");
    const bool priorities = !(pe && *pe == '0');
    for (int32_t i = 0; i < n_contexts; ++i) {
      trgt_hip_ctx* c = nullptr;
      if (priorities && n_contexts > 1) {  // contexts of a pool: different stream priorities, see make_stream (ctx.hip)
        int lo = 0, hi = 0;  // (lo = least urgent, numerically larger)
        if (hipSetDevice(devices[i]) == hipSuccess && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo > hi) trgt::ctx_next_stream_priority(hi + (int)(i % (lo - hi + 1)));
        else (void)hipGetLastError();
      }
      trgt::ctx_next_in_pool(n_contexts > 1);
      const int rc = trgt_hip_create(devices[i], &c);
      trgt::ctx_next_stream_priority(0); trgt::ctx_next_in_pool(false);
      if (rc) { for (auto* q : P->ctx) trgt_hip_destroy(q); return rc; }
      P->ctx.push_back(c);
    }
    // contexts that share a device share its memory: the workspace limit of each (32 GB by default, what the alignment and HMM launches
    // size their arenas by) is lowered to its share of 70 % of what is free now -- ten contexts on one GPU then plan 19 GB each instead
    // of failing in the middle of a batch.  (trgt_hip_set_workspace_limit on trgt_hip_pool_context(i) overrides it.)
    for (size_t i = 0; i < P->ctx.size(); ++i) {
      int64_t same = 0;
      for (auto* q : P->ctx) same += q->device == P->ctx[i]->device;
      size_t free_b = 0, total_b = 0;
      if (hipSetDevice(P->ctx[i]->device) == hipSuccess && hipMemGetInfo(&free_b, &total_b) == hipSuccess && same > 0) {
        const uint64_t share = (uint64_t)((double)free_b * 0.7 / (double)same);
        P->ctx[i]->ws_limit = std::max<uint64_t>(4ull << 30, std::min<uint64_t>(P->ctx[i]->ws_limit, share));
      } else (void)hipGetLastError();
    }
    *out = P.release();
    return TRGT_OK;
  } catch (const std::exception&) { return TRGT_ERR_NOMEM; }
}
extern "C" void trgt_hip_pool_destroy(trgt_hip_pool* P) { if (!P) return; for (auto* c : P->ctx) trgt_hip_destroy(c); delete P; }
extern "C" int32_t trgt_hip_pool_size(const trgt_hip_pool* P) { return P ? (int32_t)P->ctx.size() : 0; }
extern "C" trgt_hip_ctx* trgt_hip_pool_context(trgt_hip_pool* P, int32_t i) { return P && i >= 0 && (size_t)i < P->ctx.size() ? P->ctx[(size_t)i] : nullptr; }
extern "C" const char* trgt_hip_pool_last_error(const trgt_hip_pool* P) { return P ? P->err.c_str() : "null pool"; }

extern "C" int trgt_locus_batch_many(trgt_hip_pool* P, const trgt_locus_params* p, int64_t n_batches, const trgt_locus_batch_in* const* in,
                                     trgt_locus_batch_out* const* out, int32_t out_per_context, int32_t* ran_on) {
  if (!P || !p || n_batches < 0 || (n_batches > 0 && (!in || !out))) return TRGT_ERR_INVALID;
  try {
    std::atomic<int64_t> next{0};
    std::atomic<int> first_rc{0};
    std::mutex err_mutex;
    const bool trace = TRGT_DEV_ENV("TRGT_POOL_TRACE") != nullptr;  // (one line per batch on stderr: context, batch, start and end in ms)
    const int64_t t_many0 = now_ns();
    auto worker = [&](size_t w) {
      trgt_hip_ctx* c = P->ctx[w];
      auto failed = [&](int rc, int64_t i) {
        std::lock_guard<std::mutex> g(err_mutex);
        if (!first_rc.load()) { first_rc = rc; P->err = "batch " + std::to_string((long long)i) + " on context " + std::to_string(w) + ": " + trgt_hip_last_error(c); }
      };
      // one batch ahead: the next batch is submitted (its host-resident reads start crossing the link on the copy stream) before
      // the current one is analysed -- trgt_locus_batch_submit / _wait per context; for blobs already in HBM this is the blocking call
      auto take = [&](int64_t* i, int64_t* ticket) -> bool {  // false: nothing left (or an error somewhere)
        if (first_rc.load()) return false;
        *i = next.fetch_add(1);
        if (*i >= n_batches) return false;
        const int rc = trgt_locus_batch_submit(c, p, in[*i], out_per_context ? out[w] : out[*i], ticket);
        if (rc) { failed(rc, *i); return false; }
        return true;
      };
      int64_t cur = -1, cur_ticket = 0;
      if (!take(&cur, &cur_ticket)) return;
      for (;;) {
        int64_t nxt = -1, nxt_ticket = 0;
        const bool more = take(&nxt, &nxt_ticket);
        const int64_t tb0 = trace ? now_ns() : 0;
        const int rc = trgt_locus_batch_wait(c, cur_ticket);
        if (trace) fprintf(stderr, "[pool] %zu %lld %.3f %.3f\n", w, (long long)cur, (double)(tb0 - t_many0) / 1e6, (double)(now_ns() - t_many0) / 1e6);
        if (ran_on) ran_on[cur] = (int32_t)w;
        if (rc) failed(rc, cur);
        if (!more) return;  // (a submit that failed left nothing outstanding)
        if (rc) { (void)trgt_locus_batch_wait(c, nxt_ticket); return; }  // (drain the submitted batch: the caller's buffers must be free when we return)
        cur = nxt; cur_ticket = nxt_ticket;
      }
    };
    std::vector<std::thread> th;
    struct JoinAll { std::vector<std::thread>& t; ~JoinAll() { for (auto& x : t) if (x.joinable()) x.join(); } } join_all{th};
    for (size_t w = 1; w < P->ctx.size(); ++w) th.emplace_back(worker, w);
    worker(0);
    for (auto& t : th) t.join();
    return first_rc.load();
  } catch (const std::exception& e) { P->err = std::string("trgt_locus_batch_many: ") + e.what(); return TRGT_ERR_NOMEM; }
}
