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
// The host glue is integer / byte work of a few microseconds per locus, spread over host threads;
// moving it onto the device is SURVEY.md 8(f) row 1.
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <thread>

#include "hmm_host.hpp"
#include "wfa_host.hpp"

namespace trgt {

int find_spans_device(trgt_hip_ctx* c, const trgt_span_params& p, int64_t n_loci, int64_t n_reads, const uint8_t* d_flank,
                      const uint64_t* d_piece_off, const uint8_t* d_reads, const uint64_t* d_read_off, const uint32_t* d_read_len,
                      const uint32_t* d_read_locus, uint32_t max_read_len, int32_t* d_span_start, int32_t* d_span_end,
                      uint8_t* d_lf_hit, uint8_t* d_rf_hit);

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

// repair_consensus (consensus.rs:5-111); cigars[m] = run-length CIGAR (len<<4|code) of member m vs the backbone
std::string repair_consensus(const std::string& backbone, const std::vector<Seg>& seqs, const std::vector<std::vector<uint32_t>>& cigars) {
  const size_t L = backbone.size(), n = seqs.size();
  std::vector<std::array<int, 5>> votes(L, std::array<int, 5>{0, 0, 0, 0, 0});
  std::vector<std::vector<std::string>> inserts(L + 1);
  for (size_t m = 0; m < n; ++m) {
    size_t x = 0, y = 0;
    for (uint32_t e : cigars[m]) {
      const size_t len = e >> 4; const uint32_t code = e & 0xF;
      if (code == 7 || code == 8 || code == 0) {
        for (size_t i = 0; i < len; ++i) {
          const uint8_t b = seqs[m].p[x + i];
          const int bi = b == 'A' ? 0 : b == 'T' ? 1 : b == 'C' ? 2 : 3;
          votes[y + i][bi] += 1;
        }
        x += len; y += len;
      } else if (code == 2) { for (size_t i = 0; i < len; ++i) votes[y + i][4] += 1; y += len; }
      else if (code == 1) { inserts[y].emplace_back((const char*)seqs[m].p + x, len); x += len; }
    }
  }
  std::string out;
  for (size_t pos = 0; pos < L; ++pos) {
    int best = 0;
    for (int i = 1; i < 5; ++i) if (votes[pos][i] >= votes[pos][best]) best = i;  // max_by_key: last maximum
    if (inserts[pos].size() > n / 2) {
      auto& ins = inserts[pos];
      std::sort(ins.begin(), ins.end());
      const size_t without = n - ins.size();
      size_t top_count = 0; const std::string* top = nullptr;
      for (size_t i = 0; i < ins.size();) { size_t j = i; while (j < ins.size() && ins[j] == ins[i]) ++j; if (j - i > top_count) { top_count = j - i; top = &ins[i]; } i = j; }
      if (top_count > without) out += *top;
    }
    if (best != 4) out.push_back("ATCG"[best]);
  }
  return out;
}

template <typename F>
void parallel_for(int64_t n, int threads, F f) {  // f(index, thread)
  if (threads <= 1 || n < 64) { for (int64_t i = 0; i < n; ++i) f(i, 0); return; }
  std::vector<std::thread> th;
  const int64_t chunk = (n + threads - 1) / threads;
  for (int t = 0; t < threads; ++t) {
    const int64_t b = t * chunk, e = std::min<int64_t>(n, b + chunk);
    if (b >= e) break;
    th.emplace_back([=]() { for (int64_t i = b; i < e; ++i) f(i, t); });
  }
  for (auto& x : th) x.join();
}

struct GatherArgs { const uint8_t* reads; const uint64_t* src_off; const uint64_t* dst_off; const uint32_t* len; uint64_t n; uint8_t* out; };
__global__ void gather_segments_kernel(const GatherArgs a) {  // one wavefront per segment
  const uint64_t s = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= a.n) return;
  const uint8_t* __restrict__ src = a.reads + a.src_off[s];
  uint8_t* __restrict__ dst = a.out + a.dst_off[s];
  for (uint32_t i = threadIdx.x & 63; i < a.len[s]; i += 64) dst[i] = src[i];
}

inline int64_t now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace
}  // namespace trgt

using namespace trgt;

static int locus_batch_impl(trgt_hip_ctx* c, const trgt_locus_params* p, const trgt_locus_batch_in* in, trgt_locus_batch_out* out) {
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
  const int F = p->flank_len;
  const int64_t nr = (int64_t)in->locus_read_begin[nl];
  int threads = p->host_threads > 0 ? p->host_threads : (int)std::max(1u, std::thread::hardware_concurrency());
  threads = std::min(threads, 32);  // a few microseconds of work per locus: more threads only add spawn cost
  int64_t t0 = now_ns(), tA = 0, tB = 0, tC = 0, tHost = 0;
  int64_t stat_flank_jobs = 0, stat_cons_jobs = 0, stat_spanning = 0;
  for (int64_t l = 0; l < nl; ++l) { out->n_alleles[l] = 0; out->allele_len[2 * l] = out->allele_len[2 * l + 1] = 0; out->num_spanning[2 * l] = out->num_spanning[2 * l + 1] = 0; }
  for (int64_t r = 0; r < nr; ++r) { out->classification[r] = -1; out->read_rank[r] = -1; out->span_start[r] = out->span_end[r] = -1; }
  if (nr == 0) return TRGT_OK;
  // The motif-HMM tables depend only on the catalog: build them on a host thread while the GPU locates flanks.
  HmmModels models;
  std::thread model_thread([&]() { hmm_build_models((int32_t)nl, in->motif_blob, in->motif_off, in->set_motif_begin, models); });
  struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } model_joiner{model_thread};
  // ---------------- stage A: flank location on the GPU
  trgt_span_params sp; sp.flank_len = F; sp.min_flank_id_frac = p->min_flank_id_frac; sp.mism = p->mism; sp.gapo = p->gapo; sp.gape = p->gape;
  std::vector<uint8_t> lf_hit((size_t)nr), rf_hit((size_t)nr);
  int rc;
  {
    std::unique_lock<std::mutex> lk;
    if (c->stage_a_mutex && getenv("TRGT_SERIAL_A")) lk = std::unique_lock<std::mutex>(*c->stage_a_mutex);
    rc = trgt_find_spans_batch(c, &sp, nl, in->flank_blob, in->lf_off, in->lf_len, in->rf_off, in->rf_len, in->locus_read_begin,
                               in->read_blob, in->read_off, in->read_len, out->span_start, out->span_end, lf_hit.data(), rf_hit.data());
  }
  if (rc) return rc;
  for (int64_t r = 0; r < nr; ++r) stat_flank_jobs += (lf_hit[r] != 1) + (rf_hit[r] != 1);
  tA = now_ns() - t0;
  // ---------------- host: spanning reads (tr.rs:111-184); repeat segments gathered from HBM when the reads live there
  int64_t th0 = now_ns();
  std::vector<LocusWork> work((size_t)nl);
  const bool reads_on_device = is_device_ptr(in->read_blob);
  // pass 1 (parallel over loci): filter (tr.rs:139-145), stable sort by span length (:157), uniform downsample (:172-184);
  // each locus writes its selection into its own read range of `sel`
  struct K { uint32_t read, s, e; };
  std::vector<K> sel((size_t)nr);
  std::vector<uint32_t> n_sel((size_t)nl, 0);
  parallel_for(nl, threads, [&](int64_t l, int) {
    if (in->ploidy[l] == 0) return;  // Ploidy::Zero -> LocusResult::empty (tr.rs:29-31)
    K* ks = sel.data() + in->locus_read_begin[l];
    uint32_t n = 0;
    for (uint64_t r = in->locus_read_begin[l]; r < in->locus_read_begin[l + 1]; ++r) {
      const int32_t s = out->span_start[r], e = out->span_end[r];
      if (s < 0) continue;
      if (s >= F && (int64_t)in->read_len[r] - e >= F) {
        const K k{(uint32_t)r, (uint32_t)s, (uint32_t)e};
        uint32_t i = n++;  // stable insertion sort by span length
        while (i > 0 && (ks[i - 1].e - ks[i - 1].s) > (k.e - k.s)) { ks[i] = ks[i - 1]; --i; }
        ks[i] = k;
      }
    }
    if ((int64_t)n > p->max_depth) {
      const double step = (double)n / (double)p->max_depth;
      double fast = 0.0;
      for (int i = 0; i < p->max_depth; ++i) { const size_t ind = (size_t)std::floor(fast); if (ind != (size_t)i) std::swap(ks[i], ks[ind]); fast += step; }
      n = (uint32_t)p->max_depth;
    }
    n_sel[(size_t)l] = n;
  });
  // pass 2: flat segment arrays (LocusResult.reads order within each locus)
  uint64_t n_seg = 0;
  for (int64_t l = 0; l < nl; ++l) { work[(size_t)l].seg_begin = n_seg; n_seg += n_sel[(size_t)l]; work[(size_t)l].seg_end = n_seg; }
  std::vector<uint64_t> seg_src((size_t)n_seg), seg_dst((size_t)n_seg); std::vector<uint32_t> seg_len((size_t)n_seg), seg_read((size_t)n_seg);
  parallel_for(nl, threads, [&](int64_t l, int) {
    const K* ks = sel.data() + in->locus_read_begin[l];
    uint64_t s = work[(size_t)l].seg_begin;
    for (uint32_t i = 0; i < n_sel[(size_t)l]; ++i, ++s) {
      seg_read[s] = ks[i].read; seg_src[s] = in->read_off[ks[i].read] + ks[i].s; seg_len[s] = ks[i].e - ks[i].s;
    }
  });
  { uint64_t dst = 0; for (uint64_t s = 0; s < n_seg; ++s) { seg_dst[s] = dst; dst += seg_len[s]; } }
  stat_spanning = (int64_t)seg_src.size();
  c->dbg_ns[4] = now_ns() - th0;
  std::vector<uint8_t> seg_bytes;
  const uint8_t* seg_base = nullptr;
  if (!seg_src.empty() && reads_on_device) {
    const uint64_t total = seg_dst.back() + seg_len.back();
    seg_bytes.resize((size_t)total + 1);
    void *d_src, *d_dst, *d_len, *d_out;
    if ((rc = dev_get(c, S_LOCUS_0, seg_src.size() * 8, &d_src)) || (rc = dev_get(c, S_LOCUS_1, seg_dst.size() * 8, &d_dst)) ||
        (rc = dev_get(c, S_LOCUS_2, seg_len.size() * 4, &d_len)) || (rc = dev_get(c, S_LOCUS_3, (size_t)total + 1, &d_out)))
      return rc;
    TRGT_HIP_TRY(c, hipMemcpyAsync(d_src, seg_src.data(), seg_src.size() * 8, hipMemcpyHostToDevice, c->stream));
    TRGT_HIP_TRY(c, hipMemcpyAsync(d_dst, seg_dst.data(), seg_dst.size() * 8, hipMemcpyHostToDevice, c->stream));
    TRGT_HIP_TRY(c, hipMemcpyAsync(d_len, seg_len.data(), seg_len.size() * 4, hipMemcpyHostToDevice, c->stream));
    GatherArgs ga{in->read_blob, (const uint64_t*)d_src, (const uint64_t*)d_dst, (const uint32_t*)d_len, (uint64_t)seg_src.size(), (uint8_t*)d_out};
    hipLaunchKernelGGL(gather_segments_kernel, dim3((unsigned)((seg_src.size() + 3) / 4)), dim3(256), 0, c->stream, ga);
    TRGT_HIP_TRY(c, hipGetLastError());
    TRGT_HIP_TRY(c, hipMemcpyAsync(seg_bytes.data(), d_out, (size_t)total, hipMemcpyDeviceToHost, c->stream));
    TRGT_HIP_TRY(c, hipStreamSynchronize(c->stream));
    seg_base = seg_bytes.data();
  }
  c->dbg_ns[5] = now_ns() - th0;
  auto seg_of = [&](uint64_t s) { return reads_on_device ? Seg{seg_base + seg_dst[s], seg_len[s]} : Seg{in->read_blob + seg_src[s], seg_len[s]}; };
  // ---------------- host: length genotyping front half, threaded over loci (per-thread scratch, no per-locus allocation)
  std::vector<Scratch> scratch((size_t)std::max(1, threads));
  parallel_for(nl, threads, [&](int64_t l, int t) {
    LocusWork& w = work[(size_t)l];
    if (w.seg_begin == w.seg_end) return;
    Scratch& sc = scratch[(size_t)t];
    sc.trs.clear();
    for (uint64_t s = w.seg_begin; s < w.seg_end; ++s) sc.trs.push_back(seg_of(s));
    genotype_size_front(in->ploidy[l] == 1 ? 1 : 2, l, t, w, sc);
  });
  c->dbg_ns[6] = now_ns() - th0;
  tHost += now_ns() - th0;
  // ---------------- stage B: consensus alignments (BiWFA, affine 2,5,1, default heuristic) for the loci that need them
  int64_t tb0 = now_ns();
  struct JobRef { Repair* rep; int member; };
  std::vector<JobRef> jrefs;
  std::vector<uint8_t> cblob; std::vector<uint64_t> poff, toff, coff; std::vector<uint32_t> plen, tlen;
  for (auto& sc : scratch)
    for (auto& rep : sc.repairs) {
      const Seg bb = work[(size_t)rep.locus].pick[rep.allele];
      const uint64_t bo = cblob.size();
      cblob.insert(cblob.end(), bb.p, bb.p + bb.n);
      for (size_t m = 0; m < rep.members.size(); ++m) {
        const Seg& s = rep.members[m];
        coff.push_back(coff.empty() ? 0 : coff.back() + plen.back() + tlen.back() + 1);
        poff.push_back(bo); plen.push_back(bb.n);
        toff.push_back(cblob.size()); tlen.push_back(s.n);
        cblob.insert(cblob.end(), s.p, s.p + s.n);
        jrefs.push_back({&rep, (int)m});
      }
    }
  std::vector<uint32_t> cigars, clen(jrefs.size());
  if (!jrefs.empty()) {
    trgt_wfa_params wp;
    trgt_wfa_default_params(&wp);  // THREAD_WFA_CONSENSUS (genotype.rs:82-86)
    wp.metric = 3; wp.mismatch = 2; wp.gap_open1 = 5; wp.gap_ext1 = 1; wp.span = 0; wp.scope = 1; wp.memory_mode = 3;
    cigars.resize((size_t)(coff.back() + plen.back() + tlen.back() + 1));
    rc = trgt_wfa_batch(c, &wp, (int64_t)jrefs.size(), cblob.data(), poff.data(), plen.data(), toff.data(), tlen.data(), nullptr, nullptr,
                        nullptr, nullptr, cigars.data(), coff.data(), clen.data(), nullptr, nullptr, nullptr);
    if (rc) return rc;
    stat_cons_jobs = (int64_t)jrefs.size();
  }
  tB = now_ns() - tb0;
  // ---------------- host: repair_consensus, classification, reference allele first, output assembly
  th0 = now_ns();
  {
    size_t j = 0;
    while (j < jrefs.size()) {  // jobs of one repair are contiguous
      Repair* rep = jrefs[j].rep;
      std::vector<std::vector<uint32_t>> cg;
      for (size_t m = 0; m < rep->members.size(); ++m, ++j)
        cg.emplace_back(cigars.begin() + coff[j], cigars.begin() + coff[j] + clen[j]);  // failed alignment -> empty CIGAR
      const Seg bb = work[(size_t)rep->locus].pick[rep->allele];
      rep->result = repair_consensus(std::string((const char*)bb.p, bb.n), rep->members, cg);
    }
  }
  std::vector<int8_t> seg_cls(seg_src.size(), 0);
  int bad = 0;
  parallel_for(nl, threads, [&](int64_t l, int) {
    LocusWork& w = work[(size_t)l];
    if (w.seg_begin == w.seg_end) return;
    const int ploidy = in->ploidy[l] == 1 ? 1 : 2;
    Seg al[2];
    for (int a = 0; a < w.n_pick; ++a) {
      if (w.repair[a] >= 0) { const std::string& r = scratch[(size_t)w.repair_thread].repairs[(size_t)w.repair[a]].result; al[a] = Seg{(const uint8_t*)r.data(), (uint32_t)r.size()}; }
      else al[a] = w.pick[a];
    }
    int n_al = w.n_pick;
    if (ploidy == 2 && n_al == 1) { al[1] = al[0]; n_al = 2; }
    int by_hap[2] = {0, 0};
    int tie = 1;
    for (uint64_t s = w.seg_begin; s < w.seg_end; ++s) {
      int cc = 0;
      if (n_al == 2) {
        const uint32_t d1 = adiff(seg_len[s], al[0].n), d2 = adiff(seg_len[s], al[1].n);
        if (d1 < d2) cc = 0; else if (d1 > d2) cc = 1; else { tie = (tie + 1) % 2; cc = tie; }
      }
      seg_cls[s] = (int8_t)cc; by_hap[cc] += 1;
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
    }
    for (uint64_t s = w.seg_begin; s < w.seg_end; ++s) {
      out->classification[seg_read[s]] = flip ? 1 - seg_cls[s] : seg_cls[s];
      out->read_rank[seg_read[s]] = (int32_t)(s - w.seg_begin);
    }
  });
  if (bad) return fail(c, TRGT_ERR_INVALID, "trgt_locus_batch: allele_cap too small");
  c->dbg_ns[7] = now_ns() - th0;
  tHost += now_ns() - th0;
  // ---------------- stage C: label_with_hmm for every allele
  int64_t tc0 = now_ns();
  std::vector<uint32_t> job_set, seq_len, nsp; std::vector<uint64_t> seq_off, span_off, count_off; std::vector<double> pur;
  std::vector<int64_t> slot;
  for (int64_t l = 0; l < nl; ++l)
    for (int a = 0; a < out->n_alleles[l]; ++a) {
      job_set.push_back((uint32_t)l); seq_off.push_back(out->allele_off[2 * l + a]); seq_len.push_back(out->allele_len[2 * l + a]);
      span_off.push_back(out->span_off[2 * l + a]); count_off.push_back(out->count_off[2 * l + a]); slot.push_back(2 * l + a);
    }
  for (int64_t s = 0; s < 2 * nl; ++s) { out->n_spans[s] = 0; out->purity[s] = std::nan(""); }
  if (!job_set.empty()) {
    nsp.resize(job_set.size()); pur.resize(job_set.size());
    if (model_thread.joinable()) model_thread.join();
    rc = hmm_batch_impl(c, &models, (int32_t)nl, in->motif_blob, in->motif_off, in->set_motif_begin, (int64_t)job_set.size(), job_set.data(),
                        out->allele_blob, seq_off.data(), seq_len.data(), nullptr, nullptr, nullptr, out->spans3, span_off.data(),
                        nsp.data(), out->motif_counts, count_off.data(), pur.data(), nullptr, nullptr);
    if (rc) return rc;
    for (size_t j = 0; j < slot.size(); ++j) { out->n_spans[slot[j]] = nsp[j]; out->purity[slot[j]] = pur[j]; }
  }
  tC = now_ns() - tc0;
  if (out->stats) {
    int64_t* s = out->stats;
    s[0] = stat_flank_jobs; s[1] = stat_cons_jobs; s[2] = stat_spanning; s[3] = (int64_t)job_set.size();
    s[4] = tA; s[5] = tB; s[6] = tC; s[7] = tHost; s[8] = now_ns() - t0;
    for (int i = 0; i < 7; ++i) s[9 + i] = c->dbg_ns[i + (i >= 3 ? 1 : 0)];
  }
  return TRGT_OK;
}

// Public entry point.  Large batches are cut into chunks that two host threads ("lanes", each with its own stream and
// device buffers) work through alternately, so the host glue / transfers of one chunk overlap the GPU stages of the
// other.  Chunks touch disjoint locus / read ranges of the caller's buffers, so no synchronisation is needed.
extern "C" int trgt_locus_batch(trgt_hip_ctx* c, const trgt_locus_params* p, const trgt_locus_batch_in* in, trgt_locus_batch_out* out) {
  if (!c) return TRGT_ERR_INVALID;
  if (!p || !in || !out) return fail(c, TRGT_ERR_INVALID, "trgt_locus_batch: null argument");
  const int64_t nl = in->n_loci;
  const int64_t CHUNK = 2500;
  const char* env = getenv("TRGT_LOCUS_LANES");
  const int lanes = env ? atoi(env) : 1;  // measured: a persistent WFA kernel leaves no room for the other lane's kernels, so 2 lanes do not pay
  if (nl < 2 * CHUNK || lanes < 2 || !in->locus_read_begin) return locus_batch_impl(c, p, in, out);
  if (!c->aux) {
    int rc = trgt_hip_create(c->device, &c->aux);
    if (rc) return fail(c, rc, "trgt_locus_batch: cannot create the second lane: %s", trgt_hip_last_error(nullptr));
  }
  c->aux->timing = c->timing; c->aux->ws_limit = c->ws_limit / 2;
  const uint64_t saved_limit = c->ws_limit;
  c->ws_limit = saved_limit / 2;
  const int64_t n_chunks = (nl + CHUNK - 1) / CHUNK;
  trgt_hip_ctx* lane_ctx[2] = {c, c->aux};
  int lane_rc[2] = {0, 0};
  int64_t lane_stats[2][16];
  std::memset(lane_stats, 0, sizeof lane_stats);
  trgt_locus_params lp = *p;
  int threads = p->host_threads > 0 ? p->host_threads : (int)std::max(1u, std::thread::hardware_concurrency());
  lp.host_threads = std::max(1, std::min(threads, 32) / 2);
  const int64_t t0 = now_ns();
  std::mutex stage_a;
  c->stage_a_mutex = &stage_a; c->aux->stage_a_mutex = &stage_a;
  auto worker = [&](int lane) {
    trgt_hip_ctx* cc = lane_ctx[lane];
    for (int64_t ch = lane; ch < n_chunks && lane_rc[lane] == 0; ch += 2) {
      const int64_t l0 = ch * CHUNK, l1 = std::min(nl, l0 + CHUNK), n = l1 - l0;
      const uint64_t r0 = in->locus_read_begin[l0];
      std::vector<uint64_t> lrb((size_t)n + 1);
      for (int64_t i = 0; i <= n; ++i) lrb[(size_t)i] = in->locus_read_begin[l0 + i] - r0;
      trgt_locus_batch_in si = *in;
      si.n_loci = n;
      si.lf_off += l0; si.lf_len += l0; si.rf_off += l0; si.rf_len += l0; si.tr_off += l0; si.tr_len += l0;
      si.set_motif_begin += l0; si.ploidy += l0; si.locus_read_begin = lrb.data(); si.read_off += r0; si.read_len += r0;
      trgt_locus_batch_out so = *out;
      int64_t st[16];
      so.span_start += r0; so.span_end += r0; so.classification += r0; so.read_rank += r0;
      so.n_alleles += l0; so.allele_off += 2 * l0; so.allele_cap += l0; so.allele_len += 2 * l0; so.ci += 4 * l0; so.num_spanning += 2 * l0;
      so.span_off += 2 * l0; so.n_spans += 2 * l0; so.count_off += 2 * l0; so.purity += 2 * l0;
      so.stats = st;
      const int rc = locus_batch_impl(cc, &lp, &si, &so);
      if (rc) { lane_rc[lane] = rc; break; }
      for (int i = 0; i < 8; ++i) lane_stats[lane][i] += st[i];
    }
  };
  std::thread t1(worker, 1);
  worker(0);
  t1.join();
  c->stage_a_mutex = nullptr; c->aux->stage_a_mutex = nullptr;
  c->ws_limit = saved_limit;
  if (lane_rc[1]) c->err = c->aux->err;
  if (lane_rc[0] || lane_rc[1]) return lane_rc[0] ? lane_rc[0] : lane_rc[1];
  // fold the second lane's kernel timers into the caller's ctx
  resolve_timing(c->aux);
  for (int k = 0; k < TRGT_K_COUNT; ++k) {
    c->k_ms[k] += c->aux->k_ms[k]; c->k_launches[k] += c->aux->k_launches[k]; c->k_cells[k] += c->aux->k_cells[k];
    c->aux->k_ms[k] = 0; c->aux->k_launches[k] = 0; c->aux->k_cells[k] = 0;
  }
  if (out->stats) {
    for (int i = 0; i < 8; ++i) out->stats[i] = lane_stats[0][i] + lane_stats[1][i];
    out->stats[8] = now_ns() - t0;
    for (int i = 9; i < 16; ++i) out->stats[i] = 0;
  }
  return TRGT_OK;
}
