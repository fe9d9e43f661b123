// trgt_amd/csrc/locus_cluster.hpp -- Genotyper::Cluster for a batch of loci.  Included by locus.hip
// only (inside its anonymous namespace, after Seg / repair_consensus).
//
// Replaces genotype_cluster::genotype (PacificBiosciences/trgt v3.0.0 src/trgt/genotype/genotype_cluster.rs:58-152):
//   get_dist_matrix (:250-286)   every read pair of a locus -> edit distance, score-only BiWFA with the default heuristic
//                                (THREAD_WFA_ED, src/commands/genotype.rs:88-92) unless |a|*|b| > MAX_OPS (:236-248)
//                                -> ONE trgt_wfa_batch over all pairs of all cluster loci of the call
//   cluster (:154-227)           kodama::linkage(.., Method::Ward) + the cut-off search + group membership -> host, below
//   make_consensus (:41-56)      central_read (:12-39) on the matrix *as the linkage left it*, utils::align of every member
//                                against it (BiWFA affine 2,5,1) -> ONE trgt_wfa_batch per round, repair_consensus on the host
//   outlier reads (:117-142)     edit distance to both consensus alleles -> ONE more trgt_wfa_batch
// The GPU does all alignments (n(n-1)/2 + n + a few per locus, 465+ for 30 reads); the host keeps the O(n^2) bookkeeping.
//
// Ward linkage: kodama 0.3.0 is an un-vendored crates.io dependency of the reference (Cargo.lock:839-842); this is an
// implementation of the algorithm it ports (Muellner's NN-chain with in-place Lance-Williams updates, stable sort of the
// merges, SciPy labels).  Parity for that piece is unpinned by any reference test; tests compare against the oracle.
#pragma once

struct MergeStep { int a, b, size; double diss; };

// Condensed upper-triangular matrix of n observations (row-major, i < j).
struct Condensed {
  double* d; size_t n;
  inline double& at(size_t i, size_t j) const { return d[n * i - i * (i + 1) / 2 + (j - i - 1)]; }
  inline double& sym(size_t i, size_t j) const { return i < j ? at(i, j) : at(j, i); }
};

// linkage(dists, n, Method::Ward): dists is squared and updated in place; steps come back sorted and relabelled.
inline void ward_nnchain(double* dists, int n, std::vector<MergeStep>& steps) {
  steps.clear();
  const size_t m = (size_t)n * (size_t)(n - 1) / 2;
  for (size_t i = 0; i < m; ++i) dists[i] = dists[i] * dists[i];
  if (n < 2) return;
  Condensed D{dists, (size_t)n};
  // active observations as a doubly linked list in index order (index n = end sentinel)
  std::vector<int> succ((size_t)n + 1), pred((size_t)n + 1), members((size_t)n, 1), chain;
  for (int i = 0; i <= n; ++i) { succ[(size_t)i] = i + 1; pred[(size_t)i] = i - 1; }
  int head = 0;
  auto unlink = [&](int i) {
    if (i == head) head = succ[(size_t)i]; else succ[(size_t)pred[(size_t)i]] = succ[(size_t)i];
    pred[(size_t)succ[(size_t)i]] = pred[(size_t)i];
  };
  chain.reserve((size_t)n);
  for (int merge = 0; merge + 1 < n; ++merge) {
    int tip, nearest; double best;
    if (chain.size() <= 3) {
      tip = head; chain.assign(1, tip);
      nearest = succ[(size_t)tip]; best = D.at((size_t)tip, (size_t)nearest);
      for (int i = succ[(size_t)nearest]; i < n; i = succ[(size_t)i]) { const double v = D.at((size_t)tip, (size_t)i); if (v < best) { best = v; nearest = i; } }
    } else {
      nearest = chain[chain.size() - 3];  // drop the merged pair and the element before it; that element is looked at again
      chain.resize(chain.size() - 3);
      tip = chain.back();
      best = D.sym((size_t)tip, (size_t)nearest);
    }
    // grow the chain until two clusters are each other's nearest neighbour; ties keep the previous chain element
    for (;;) {
      const int cur = nearest;
      chain.push_back(cur);
      int nn = tip;
      for (int i = head; i < cur; i = succ[(size_t)i]) { const double v = D.at((size_t)i, (size_t)cur); if (v < best) { best = v; nn = i; } }
      for (int i = succ[(size_t)cur]; i < n; i = succ[(size_t)i]) { const double v = D.at((size_t)cur, (size_t)i); if (v < best) { best = v; nn = i; } }
      tip = cur; nearest = nn;
      if (nearest == chain[chain.size() - 2]) break;
    }
    int lo = tip < nearest ? tip : nearest, hi = tip < nearest ? nearest : tip;
    const double s_lo = (double)members[(size_t)lo], s_hi = (double)members[(size_t)hi];
    auto lance_williams = [&](double d_lo, double& d_hi, int x) {
      const double sx = (double)members[(size_t)x];
      d_hi = (((sx + s_lo) * d_lo) + ((sx + s_hi) * d_hi) - (sx * best)) / (s_lo + s_hi + sx);
    };
    int x = head;
    for (; x < lo; x = succ[(size_t)x]) lance_williams(D.at((size_t)x, (size_t)lo), D.at((size_t)x, (size_t)hi), x);
    for (x = succ[(size_t)lo]; x < hi; x = succ[(size_t)x]) lance_williams(D.at((size_t)lo, (size_t)x), D.at((size_t)x, (size_t)hi), x);
    for (x = succ[(size_t)hi]; x < n; x = succ[(size_t)x]) lance_williams(D.at((size_t)lo, (size_t)x), D.at((size_t)hi, (size_t)x), x);
    members[(size_t)hi] += members[(size_t)lo];
    unlink(lo);
    steps.push_back(MergeStep{lo, hi, members[(size_t)hi], best});
  }
  std::stable_sort(steps.begin(), steps.end(), [](const MergeStep& p, const MergeStep& q) { return p.diss < q.diss; });
  std::vector<int> up((size_t)(2 * n - 1), -1);
  auto root = [&](int v) { int r = v; while (up[(size_t)r] >= 0) r = up[(size_t)r]; while (up[(size_t)v] >= 0) { const int w = up[(size_t)v]; up[(size_t)v] = r; v = w; } return r; };
  for (size_t i = 0; i < steps.size(); ++i) {
    int p = root(steps[i].a), q = root(steps[i].b);
    if (p > q) std::swap(p, q);
    const int sp = p < n ? 1 : steps[(size_t)(p - n)].size, sq = q < n ? 1 : steps[(size_t)(q - n)].size;
    steps[i].a = p; steps[i].b = q; steps[i].size = sp + sq;
    up[(size_t)p] = up[(size_t)q] = n + (int)i;
  }
  for (auto& s : steps) s.diss = std::sqrt(s.diss);
}

struct ClusterLocus {
  int64_t li = -1;                  // index into the host-path locus list
  int ploidy = 2, n = 0;
  const Seg* trs = nullptr;         // n repeat segments, LocusResult.reads order
  uint64_t blob_off = 0;            // segment i sits at blob_off + seg_off[i] of the batch blob
  std::vector<uint64_t> seg_off;
  std::vector<double> dists;        // condensed; mutated by the linkage exactly as the reference's is
  std::vector<int> group[2];
  int n_groups = 0;
  bool even_odd_redo = false;       // small_group_is_outlier -> the homozygous split (:99-115)
  std::string allele[2];
  uint32_t ci[4] = {0, 0, 0, 0};
  int n_gt = 0;
  std::vector<int8_t> cls;          // per read: 0 / 1 (2 = outlier until assigned)
};

// genotype_cluster::cluster (:154-227)
inline void cluster_groups(int n, std::vector<double>& dists, std::vector<std::vector<int>>& groups, std::vector<MergeStep>& steps) {
  groups.clear();
  if (n == 2) { groups = {{0}, {1}}; return; }
  ward_nnchain(dists.data(), n, steps);
  auto csize = [&](int label) { return label < n ? 1 : steps[(size_t)(label - n)].size; };
  const int min_cluster = std::max(2, (int)std::round(0.01 * (double)n));
  double cutoff = 0.0;
  for (size_t i = steps.size(); i-- > 0;)
    if (std::min(csize(steps[i].a), csize(steps[i].b)) >= min_cluster) { cutoff = steps[i].diss - 0.0001; break; }
  if (cutoff == 0.0) {  // homozygous: split reads across alleles equally
    groups.resize(2);
    for (int i = 0; i < n; ++i) groups[(size_t)(i & 1)].push_back(i);
    return;
  }
  std::vector<int> member((size_t)(2 * n - 1), -1);
  int n_groups = 0;
  for (size_t i = steps.size(); i-- > 0;) {
    if (!(steps[i].diss <= cutoff)) continue;
    int& mine = member[(size_t)n + i];
    if (mine < 0) mine = n_groups++;
    member[(size_t)steps[i].a] = mine; member[(size_t)steps[i].b] = mine;
  }
  for (int i = 0; i < n; ++i) if (member[(size_t)i] < 0) member[(size_t)i] = n_groups++;
  groups.resize((size_t)n_groups);
  for (int i = 0; i < n; ++i) groups[(size_t)member[(size_t)i]].push_back(i);
}

// central_read (:12-39): the member with the smallest distance sum inside the group (first minimum)
inline int central_read(int n, const std::vector<int>& group, const std::vector<double>& dists) {
  if (group.size() <= 2) return group[0];
  std::vector<double> sum(group.size(), 0.0);
  for (size_t i = 0; i + 1 < group.size(); ++i)
    for (size_t j = i + 1; j < group.size(); ++j) {
      const size_t a = (size_t)group[i], b = (size_t)group[j];
      const double v = dists[(size_t)n * a - a * (a + 3) / 2 + b - 1];
      sum[i] += v; sum[j] += v;
    }
  size_t best = 0;
  for (size_t i = 1; i < sum.size(); ++i) if (sum[i] < sum[best]) best = i;
  return group[best];
}

inline int64_t now_ns_cluster() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct ClusterBatch {
  trgt_hip_ctx* c;
  HostPool* pool;
  std::vector<ClusterLocus>& loci;
  std::vector<uint8_t> blob;  // every repeat segment of every cluster locus once; consensus alleles appended later
  int64_t n_ed = 0, n_cons = 0;
  static constexpr uint64_t MAX_OPS = 10000;  // :236

  ClusterBatch(trgt_hip_ctx* c_, HostPool* pool_, std::vector<ClusterLocus>& l) : c(c_), pool(pool_), loci(l) {}

  struct EdRef { double* dst; };
  std::vector<uint64_t> poff, toff; std::vector<uint32_t> plen, tlen; std::vector<EdRef> refs;
  void ed_clear() { poff.clear(); toff.clear(); plen.clear(); tlen.clear(); refs.clear(); }
  // get_dist (:238-248): *dst = sqrt(distance)
  void ed_add(uint64_t a_off, uint32_t a_len, uint64_t b_off, uint32_t b_len, double* dst) {
    if ((uint64_t)a_len * (uint64_t)b_len > MAX_OPS) { *dst = std::sqrt((double)(int32_t)(a_len > b_len ? a_len - b_len : b_len - a_len)); return; }
    poff.push_back(a_off); plen.push_back(a_len); toff.push_back(b_off); tlen.push_back(b_len); refs.push_back({dst});
  }
  int ed_run() {
    if (refs.empty()) return TRGT_OK;
    trgt_wfa_params wp;
    trgt_wfa_default_params(&wp);  // THREAD_WFA_ED: Score scope, MemoryUltraLow, edit, default heuristic (genotype.rs:88-92)
    wp.metric = 1; wp.span = 0; wp.scope = 0; wp.memory_mode = 3; sens_apply(c, wp);
    std::vector<int32_t> score(refs.size());
    const int rc = trgt_wfa_batch(c, &wp, (int64_t)refs.size(), blob.data(), poff.data(), plen.data(), toff.data(), tlen.data(), nullptr,
                                  score.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    if (rc) return rc;
    pool->parallel_for((int64_t)((refs.size() + 4095) / 4096), 1, [&](int64_t blk, int) {  // aligner.score() as f64, then sqrt
      for (size_t j = (size_t)blk * 4096, e = std::min(refs.size(), j + 4096); j < e; ++j) *refs[j].dst = std::sqrt((double)score[j]);
    });
    n_ed += (int64_t)refs.size();
    return TRGT_OK;
  }

  // make_consensus (:41-56) for the listed (locus, group) pairs: one alignment batch, then repair_consensus per group
  int consensus_round(const std::vector<std::pair<int, int>>& todo) {
    if (todo.empty()) return TRGT_OK;
    std::vector<uint64_t> po, to; std::vector<uint32_t> pl, tl;
    std::vector<size_t> first(todo.size() + 1, 0);
    std::vector<int> backbone(todo.size());
    for (size_t t = 0; t < todo.size(); ++t) {
      ClusterLocus& L = loci[(size_t)todo[t].first];
      const std::vector<int>& g = L.group[todo[t].second];
      const int bb = central_read(L.n, g, L.dists);
      backbone[t] = bb;
      for (int i : g) {
        po.push_back(L.blob_off + L.seg_off[(size_t)bb]); pl.push_back(L.trs[bb].n);
        to.push_back(L.blob_off + L.seg_off[(size_t)i]); tl.push_back(L.trs[i].n);
      }
      first[t + 1] = po.size();
    }
    std::vector<std::string> repaired;
    const int rc = consensus_repair_batch(c, (int64_t)po.size(), blob.data(), po.data(), pl.data(), to.data(), tl.data(), first, repaired);
    if (rc) return rc;
    n_cons += (int64_t)po.size();
    for (size_t t = 0; t < todo.size(); ++t) {
      ClusterLocus& L = loci[(size_t)todo[t].first];
      const int gi = todo[t].second;
      uint32_t lo = 0xFFFFFFFFu, hi = 0;
      for (int m : L.group[gi]) { lo = std::min(lo, L.trs[m].n); hi = std::max(hi, L.trs[m].n); }  // get_ci (:229-233)
      L.allele[gi].swap(repaired[t]);
      L.ci[2 * gi] = lo; L.ci[2 * gi + 1] = hi;
    }
    return TRGT_OK;
  }

  int run() {
    if (loci.empty()) return TRGT_OK;
    const bool tl_on = c->knobs.timeline;
    const int64_t tl0 = now_ns_cluster();
#define CTL(name) do { if (tl_on) fprintf(stderr, "[tl]   cluster %-22s +%7.2f ms\n", name, (double)(now_ns_cluster() - tl0) / 1e6); } while (0)
    // the batch blob: all segments once
    {
      uint64_t total = 0;
      for (auto& L : loci) {
        L.blob_off = total; L.seg_off.resize((size_t)L.n);
        for (int i = 0; i < L.n; ++i) { L.seg_off[(size_t)i] = total - L.blob_off; total += L.trs[i].n; }
      }
      blob.resize((size_t)total);
      pool->parallel_for((int64_t)loci.size(), 32, [&](int64_t k, int) {
        const ClusterLocus& L = loci[(size_t)k];
        for (int i = 0; i < L.n; ++i) std::memcpy(blob.data() + L.blob_off + L.seg_off[(size_t)i], L.trs[i].p, L.trs[i].n);
      });
    }
    CTL("blob built");
    // ---- get_dist_matrix for every locus
    ed_clear();
    {
      // the pair list of every locus, written by the host pool into its place in the batch arrays (a 2 000-locus call has 870 k pairs:
      // one thread pushing them one by one took 2 ms): first the number of alignments per locus -- pairs beyond MAX_OPS take the
      // length difference (get_dist :238-248) -- then the entries
      std::vector<uint64_t> first(loci.size() + 1, 0);
      pool->parallel_for((int64_t)loci.size(), 16, [&](int64_t k, int) {
        ClusterLocus& L = loci[(size_t)k];
        L.dists.assign((size_t)L.n * (size_t)(L.n - 1) / 2, 0.0);
        uint64_t n = 0;
        for (int i = 0; i < L.n; ++i)
          for (int j = i + 1; j < L.n; ++j) n += (uint64_t)L.trs[i].n * (uint64_t)L.trs[j].n <= MAX_OPS;
        first[(size_t)k + 1] = n;
      });
      for (size_t k = 0; k < loci.size(); ++k) first[k + 1] += first[k];
      const size_t total = (size_t)first[loci.size()];
      poff.resize(total); plen.resize(total); toff.resize(total); tlen.resize(total); refs.resize(total);
      pool->parallel_for((int64_t)loci.size(), 16, [&](int64_t k, int) {
        ClusterLocus& L = loci[(size_t)k];
        double* d = L.dists.data();
        size_t at = (size_t)first[(size_t)k];
        for (int i = 0; i < L.n; ++i)
          for (int j = i + 1; j < L.n; ++j, ++d) {
            const uint32_t a_len = L.trs[i].n, b_len = L.trs[j].n;
            if ((uint64_t)a_len * (uint64_t)b_len > MAX_OPS) { *d = std::sqrt((double)(int32_t)(a_len > b_len ? a_len - b_len : b_len - a_len)); continue; }
            poff[at] = L.blob_off + L.seg_off[(size_t)i]; plen[at] = a_len; toff[at] = L.blob_off + L.seg_off[(size_t)j]; tlen[at] = b_len; refs[at] = EdRef{d};
            ++at;
          }
      });
    }
    CTL("ed jobs built");
    int rc = ed_run();
    if (rc) return rc;
    CTL("ed batch done");
    // ---- groups
    pool->parallel_for((int64_t)loci.size(), 4, [&](int64_t k, int) {
      ClusterLocus& L = loci[(size_t)k];
      L.cls.assign((size_t)L.n, 0);
      if (L.ploidy == 1 || L.n == 1) {
        L.n_groups = 1; L.group[0].resize((size_t)L.n);
        for (int i = 0; i < L.n; ++i) L.group[0][(size_t)i] = i;
        return;
      }
      std::vector<std::vector<int>> groups; std::vector<MergeStep> steps;
      cluster_groups(L.n, L.dists, groups, steps);
      std::stable_sort(groups.begin(), groups.end(), [](const std::vector<int>& x, const std::vector<int>& y) { return x.size() < y.size(); });
      L.n_groups = 2;
      L.group[0].swap(groups[groups.size() - 1]); L.group[1].swap(groups[groups.size() - 2]);
    });
    CTL("groups");
    std::vector<std::pair<int, int>> todo;
    for (size_t k = 0; k < loci.size(); ++k) for (int g = 0; g < loci[k].n_groups; ++g) todo.push_back({(int)k, g});
    if ((rc = consensus_round(todo))) return rc;
    CTL("consensus round 1");
    // ---- small_group_is_outlier (:89-98) -> redo as the homozygous split
    todo.clear();
    for (size_t k = 0; k < loci.size(); ++k) {
      ClusterLocus& L = loci[k];
      if (L.n_groups != 2) continue;
      const size_t l1 = L.allele[0].size(), l2 = L.allele[1].size(), c1 = L.group[0].size(), c2 = L.group[1].size();
      if ((l1 > l2 ? l1 - l2 : l2 - l1) < 100 && std::min(c1, c2) * 4 < std::max(c1, c2)) {
        L.even_odd_redo = true;
        L.group[0].clear(); L.group[1].clear();
        for (int i = 0; i < L.n; ++i) L.group[(size_t)(i & 1)].push_back(i);
        todo.push_back({(int)k, 0}); todo.push_back({(int)k, 1});
      }
    }
    if ((rc = consensus_round(todo))) return rc;
    CTL("consensus round 2");
    // ---- classification; outlier reads (dropped by cluster()) go to the closer consensus (:117-142)
    ed_clear();
    std::vector<std::array<double, 2>> odist;
    std::vector<std::pair<int, int>> oref;  // (locus, read)
    size_t n_out = 0;
    for (auto& L : loci) {
      if (L.n_groups != 2) continue;
      if (L.even_odd_redo) { for (int i = 0; i < L.n; ++i) L.cls[(size_t)i] = (int8_t)(i & 1); continue; }
      std::fill(L.cls.begin(), L.cls.end(), (int8_t)2);
      for (int i : L.group[0]) L.cls[(size_t)i] = 0;
      for (int i : L.group[1]) L.cls[(size_t)i] = 1;
      for (int i = 0; i < L.n; ++i) n_out += L.cls[(size_t)i] == 2;
    }
    odist.resize(n_out); oref.reserve(n_out);
    if (n_out) {
      for (size_t k = 0; k < loci.size(); ++k) {
        ClusterLocus& L = loci[k];
        if (L.n_groups != 2 || L.even_odd_redo) continue;
        bool any = false;
        for (int i = 0; i < L.n; ++i) any |= L.cls[(size_t)i] == 2;
        if (!any) continue;
        uint64_t aoff[2];
        for (int a = 0; a < 2; ++a) { aoff[a] = blob.size(); blob.insert(blob.end(), L.allele[a].begin(), L.allele[a].end()); }
        for (int i = 0; i < L.n; ++i) {
          if (L.cls[(size_t)i] != 2) continue;
          std::array<double, 2>& d = odist[oref.size()];
          oref.push_back({(int)k, i});
          for (int a = 0; a < 2; ++a) ed_add(L.blob_off + L.seg_off[(size_t)i], L.trs[i].n, aoff[a], (uint32_t)L.allele[a].size(), &d[(size_t)a]);
        }
      }
      if ((rc = ed_run())) return rc;
      for (size_t o = 0; o < oref.size(); ++o) {
        const double d1 = odist[o][0], d2 = odist[o][1];
        // tie_breaker is re-initialised to 1 for every read (:125), so an exact tie always resolves to (1 + 1) % 2 = 0
        loci[(size_t)oref[o].first].cls[(size_t)oref[o].second] = (int8_t)(d1 < d2 ? 0 : (d2 < d1 ? 1 : 0));
      }
    }
    CTL("outliers");
    // ---- Gt / allele order
    for (auto& L : loci) {
      if (L.n_groups == 1) {
        if (L.ploidy == 1) L.n_gt = 1;
        else { L.n_gt = 2; L.allele[1] = L.allele[0]; L.ci[2] = L.ci[0]; L.ci[3] = L.ci[1]; }  // one read, two alleles (:70-72)
        continue;
      }
      L.n_gt = 2;
      if (L.allele[0].size() > L.allele[1].size()) {
        L.allele[0].swap(L.allele[1]); std::swap(L.ci[0], L.ci[2]); std::swap(L.ci[1], L.ci[3]);
        for (auto& v : L.cls) v = (int8_t)(1 - v);
      }
    }
    return TRGT_OK;
  }
};
