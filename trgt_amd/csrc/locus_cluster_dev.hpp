// trgt_amd/csrc/locus_cluster_dev.hpp -- Genotyper::Cluster on the device: what locus_cluster.hpp does on
// host threads between its GPU batches, as kernels between the same alignment launches, so that a cluster locus never leaves the stream.
//
// Replaces, per locus (PacificBiosciences/trgt v3.0.0 src/trgt/genotype/genotype_cluster.rs):
//   get_dist_matrix (:250-286) / get_dist (:236-248)   cluster_front_kernel: one edit-distance job per read pair (or the length
//                                                      difference beyond MAX_OPS) -> ONE score-only BiWFA launch for all loci
//   cluster (:154-227)                                 cluster_ward_kernel: Ward linkage (kodama 0.3.0: Muellner's NN-chain, in-place
//                                                      Lance-Williams updates, stable sort, SciPy labels), cut-off, groups
//   central_read (:12-39), make_consensus (:41-56)     the same kernel: backbone per group, consensus jobs + vote groups
//                                                      -> BiWFA launch (utils::align) + consensus_vote_kernel (repair_consensus)
//   small_group_is_outlier + the homozygous redo       cluster_round2_kernel (second, usually empty, consensus round)
//     (:84-115), outlier reads (:117-142)              and the edit distances of the reads cluster() dropped to both alleles
//   allele order, classifications (:143-151),          cluster_finish_kernel, together with what tr.rs:95-101 does afterwards
//     reference allele first
// One wave per locus everywhere.  The f64 arithmetic that decides anything -- sqrt of the scores, their squares, the Lance-Williams
// update, the sums of central_read, the cut-off -- is written operation for operation like ward_nnchain / central_read in
// locus_cluster.hpp (which the oracle pins; -ffp-contract=off, IEEE division and square root), and the places where the reference's
// sequential scans decide ties are reproduced by construction: "first minimum in index order" is a wave minimum followed by the
// lowest lane that holds it, with the lanes owning consecutive indices.
// A locus that does not fit (more than MAXR reads, no room in the job / CIGAR / vote arenas, a repaired allele beyond its slot) keeps
// need_host = 1 and takes the host path of locus_cluster.hpp, as every cluster locus did before.
#pragma once
#include "locus_gt.hpp"

namespace trgt {
namespace cl {

constexpr uint64_t CL_MAX_OPS = 10000;  // genotype_cluster.rs:236

enum { CC_ED = 0, CC_J1 = 1, CC_G1 = 2, CC_J2 = 3, CC_CIGAR = 4 /* u64 */, CC_OUT = 6 /* u64 */, CC_SCRATCH = 8 /* u64 */, CC_G2 = 10, CC_ED2 = 11, CC_FAILED = 12,
       CC_DONE = 13, CC_REFUSED = 14 /* alignment jobs of the chain the generic kernel refused */, CC_WORDS = 16 };

struct ClRec {  // one cluster locus between the kernels of the chain
  int32_t n;          // spanning reads kept
  int32_t state;      // 0: nothing to genotype (no spanning read), 1: on its way, -1: host path
  int32_t n_groups;   // 1 (haploid, or a single read) or 2
  int32_t redo;       // small_group_is_outlier: the even / odd split replaces the two groups
  int32_t grp[2];     // vote groups of the consensus round that counts (round 1, or round 2 after a redo)
  int32_t gsize[2];   // members
  int32_t cr_eo[2];   // central reads of the even / odd split (on the matrix as the linkage left it)
  uint32_t ci[4];     // get_ci per group
  uint32_t ci_eo[4];
};

struct ClArgs {
  gt::GtArgs g;
  const uint32_t* list; uint32_t n_list;  // the cluster loci of the call
  const uint64_t* mat_off;                // [n_list] first pair slot of a locus (slots: condensed matrix order, n(n-1)/2 of them)
  uint32_t* counts; ClRec* rec;
  int8_t* cls;                            // [reads of the batch] group of a kept read (0 / 1, 2 = dropped by cluster()), at locus_read_begin + rank
  int32_t* escore; double* gmat;          // edit distance per pair slot; the matrix of loci of the large instantiation
  JobDev* ed_jobs;
  JobDev* jobs; uint32_t cap_j;           // consensus jobs: round 1 from 0, round 2 from cap_j
  gt::RGroup* groups; uint32_t cap_g;     // vote groups: round 1 from 0, round 2 from cap_g
  JobDev* ed2_jobs; int32_t* escore2;     // dropped reads against both alleles: slot 2 * (read of the batch) + allele
  const uint8_t* vote_out; const uint32_t* vote_len;
  uint64_t cap_cigar, cap_out, cap_scratch; uint32_t vote_lds_pos;
  uint32_t flags;  // tools/unpinned_sensitivity.py: 1 = nearest-neighbour ties to the last candidate, 2 = Lance-Williams summed in another order
};

template <int MAXR>
struct ClFront {
  uint32_t r_s[MAXR], r_len[MAXR]; uint64_t r_off[MAXR];
  uint32_t s_read[MAXR], s_start[MAXR], s_len[MAXR];
  int n;
};

__device__ __forceinline__ uint32_t pair_idx(uint32_t n, uint32_t i, uint32_t j) { return n * i - i * (i + 1) / 2 + (j - i - 1); }  // i < j
__device__ __forceinline__ uint32_t pair_sym(uint32_t n, uint32_t i, uint32_t j) { return i < j ? pair_idx(n, i, j) : pair_idx(n, j, i); }
__device__ __forceinline__ double wave_min_f64(double v) {
  for (int o = 32; o > 0; o >>= 1) { const double w = __shfl_xor(v, o); v = w < v ? w : v; }
  return v;
}
__device__ __forceinline__ int wave_min_i32(int v) {
  for (int o = 32; o > 0; o >>= 1) { const int w = __shfl_xor(v, o); v = w < v ? w : v; }
  return v;
}
__device__ __forceinline__ unsigned long long lanes_below(int lane) { return lane == 0 ? 0ull : (~0ull >> (64 - lane)); }

// ---- 1. spanning reads, pair list
template <int MAXR>
__global__ void __launch_bounds__(64) cluster_front_kernel(const ClArgs a) {
  __shared__ ClFront<MAXR> sh;
  __shared__ uint32_t s_base;
  const uint32_t k = blockIdx.x;
  if (k >= a.n_list) return;
  const int64_t l = a.list[k];
  const int lane = threadIdx.x;
  const uint64_t r0 = a.g.locus_read_begin[l];
  const int nr = (int)(a.g.locus_read_begin[l + 1] - r0);
  ClRec rec;
  rec.n = 0; rec.state = 0; rec.n_groups = 0; rec.redo = 0;
  for (int q = 0; q < 2; ++q) { rec.grp[q] = -1; rec.gsize[q] = 0; rec.cr_eo[q] = 0; }
  for (int q = 0; q < 4; ++q) { rec.ci[q] = 0; rec.ci_eo[q] = 0; }
  if (a.g.ploidy[l] == 0 || nr == 0 || nr > MAXR) {  // (the genotyper kernel wrote the outputs of such a locus: empty, or need_host = 1)
    rec.state = (a.g.ploidy[l] == 0 || nr == 0) ? 0 : -1;
    if (lane == 0) a.rec[k] = rec;
    return;
  }
  if (lane == 0) sh.n = 0;
  gt::gt_front<MAXR>(sh, a.g, r0, nr, lane);
  const int n = sh.n;
  rec.n = n; rec.state = n > 0 ? 1 : 0;
  if (lane == 0) a.rec[k] = rec;
  if (n < 3) return;  // (no decision reads the matrix of one or two sequences)
  // get_dist_matrix (:250-286): pairs whose length product exceeds MAX_OPS take the length difference (get_dist :238-248)
  uint32_t cnt = 0;
  for (int i = 0; i + 1 < n; ++i) {
    const uint64_t li = sh.s_len[i];
    for (int j = i + 1 + lane; j < n; j += 64) cnt += li * (uint64_t)sh.s_len[j] <= CL_MAX_OPS;
  }
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  if (lane == 0) s_base = atomicAdd(a.counts + CC_ED, cnt);
  __syncthreads();
  uint32_t at = s_base;
  const uint64_t m0 = a.mat_off[k];
  for (int i = 0; i + 1 < n; ++i) {
    const uint32_t li = sh.s_len[i];
    const uint64_t oi = sh.r_off[sh.s_read[i]] + sh.s_start[i];
    for (int jb = i + 1; jb < n; jb += 64) {
      const int j = jb + lane;
      const bool valid = j < n;
      const uint32_t lj = valid ? sh.s_len[j] : 0u;
      const bool job = valid && (uint64_t)li * (uint64_t)lj <= CL_MAX_OPS;
      const unsigned long long mask = __ballot(job);
      if (valid) {
        const uint64_t slot = m0 + pair_idx((uint32_t)n, (uint32_t)i, (uint32_t)j);
        if (job) {
          JobDev jd;
          jd.pat_off = oi; jd.pat_len = li;
          jd.txt_off = sh.r_off[sh.s_read[j]] + sh.s_start[j]; jd.txt_len = lj;
          jd.cigar_off = 0; jd.ops_off = 0; jd.out_index = (uint32_t)slot; jd.pad = 0;
          a.ed_jobs[at + (uint32_t)__popcll(mask & lanes_below(lane))] = jd;
        } else a.escore[slot] = (int32_t)(li > lj ? li - lj : lj - li);
      }
      at += (uint32_t)__popcll(mask);
    }
  }
}

// ---- 2. linkage, groups, backbones, consensus jobs
template <int MAXR, bool LDS_MAT>
struct ClWard {
  ClFront<MAXR> f;
  uint8_t act[MAXR];
  int16_t mem[MAXR], chain[MAXR + 4];
  int16_t st_a[MAXR], st_b[MAXR], st_size[MAXR]; double st_d[MAXR];  // merges in the order they were made
  int16_t so_a[MAXR], so_b[MAXR], so_size[MAXR]; double so_d[MAXR];  // sorted and relabelled
  int16_t up[2 * MAXR], member[2 * MAXR];
  int16_t gsz[MAXR];
  int8_t cls[MAXR];
  int16_t gm[MAXR];  // members of the group at hand
  int rp_ok; uint32_t rp_g0, rp_j0; unsigned long long rp_c0, rp_o0, rp_s0;
  alignas(16) double mat[LDS_MAT ? MAXR * (MAXR - 1) / 2 : 1];
};

// central_read (:12-39): the member with the smallest sum of distances inside the group, first minimum; gm[0 .. cnt) ascending.
// Every member's sum is formed in the order the reference's double loop forms it: the pairs (k, m), k < m, then (m, k), k > m.
template <int MAXR>
__device__ __forceinline__ int central_read_wave(const double* D, uint32_t n, const int16_t* gm, int cnt, int lane) {
  if (cnt <= 2) return gm[0];
  constexpr int T = MAXR / 64;
  double best_v = 0.0; int best_m = -1;
  for (int t = 0; t < T; ++t) {
    const int m = lane * T + t;
    if (m >= cnt) break;
    const uint32_t gmm = (uint32_t)gm[m];
    double sum = 0.0;
    for (int q = 0; q < cnt; ++q) {
      if (q == m) continue;
      const uint32_t gq = (uint32_t)gm[q];
      sum += D[q < m ? pair_idx(n, gq, gmm) : pair_idx(n, gmm, gq)];
    }
    if (best_m < 0 || sum < best_v) { best_v = sum; best_m = m; }
  }
  const double inf = __builtin_huge_val();
  const double vmin = wave_min_f64(best_m >= 0 ? best_v : inf);
  const unsigned long long mask = __ballot(best_m >= 0 && best_v == vmin);
  const int src = mask ? __ffsll((long long)mask) - 1 : 0;
  const int m = __shfl(best_m, src);
  return gm[m < 0 ? 0 : m];
}

template <int MAXR, bool LDS_MAT>
__global__ void __launch_bounds__(64) cluster_ward_kernel(const ClArgs a) {
  __shared__ ClWard<MAXR, LDS_MAT> sh;
  constexpr int T = MAXR / 64;
  const uint32_t k = blockIdx.x;
  if (k >= a.n_list) return;
  ClRec rec = a.rec[k];
  if (rec.state != 1) return;
  const int64_t l = a.list[k];
  const int lane = threadIdx.x;
  const uint64_t r0 = a.g.locus_read_begin[l];
  const int nr = (int)(a.g.locus_read_begin[l + 1] - r0);
  if (lane == 0) sh.f.n = 0;
  gt::gt_front<MAXR>(sh.f, a.g, r0, nr, lane);
  const int n = sh.f.n;
  const uint32_t un = (uint32_t)n;
  const int ploidy = a.g.ploidy[l] == 1 ? 1 : 2;
  const uint64_t m0 = a.mat_off[k];
  double* const D = LDS_MAT ? sh.mat : a.gmat + m0;
  const uint32_t np = un * (un - 1) / 2;
  const bool one_group = ploidy == 1 || n == 1;
  // the matrix: sqrt(score as f64) (get_dist :247); kodama squares it in place (linkage, Method::Ward)
  if (n >= 3)
    for (uint32_t p = (uint32_t)lane; p < np; p += 64) {
      const double d = __builtin_sqrt((double)a.escore[m0 + p]);
      D[p] = one_group ? d : d * d;
    }
  for (int i = lane; i < n; i += 64) sh.cls[i] = 0;
  __syncthreads();
  int n_groups = 1;
  if (!one_group) {
    n_groups = 2;
    if (n == 2) {  // cluster(): [[0], [1]]; the stable sort by size keeps that order, the LAST group is popped first
      if (lane == 0) { sh.cls[0] = 1; sh.cls[1] = 0; }
    } else {
      // ---- nearest-neighbour chain (ward_nnchain of locus_cluster.hpp, operation for operation)
      for (int i = lane; i < n; i += 64) { sh.act[i] = 1; sh.mem[i] = 1; }
      __syncthreads();
      const double inf = __builtin_huge_val();
      // smallest D(i, cur) over the active i != cur, and the smallest such i among equals
      const bool ties_last = (a.flags & 1u) != 0, lw_alt = (a.flags & 2u) != 0;
      auto scan = [&](int cur, double& vmin, int& imin) {
        double bv = inf; int bi = -1;
        for (int t = 0; t < T; ++t) {
          const int i = lane * T + t;
          if (i < n && i != cur && sh.act[i]) { const double v = D[pair_sym(un, (uint32_t)i, (uint32_t)cur)]; if (v < bv || (ties_last && v == bv)) { bv = v; bi = i; } }
        }
        vmin = wave_min_f64(bv);
        const unsigned long long mask = __ballot(bi >= 0 && bv == vmin);
        const int src = mask ? (ties_last ? 63 - (int)__builtin_clzll(mask) : __ffsll((long long)mask) - 1) : 0;
        imin = __shfl(bi, src);
      };
      int chain_len = 0;
      for (int merge = 0; merge + 1 < n; ++merge) {
        int tip, nearest; double best;
        if (chain_len <= 3) {
          int h = 0x7FFFFFFF;
          for (int t = T - 1; t >= 0; --t) { const int i = lane * T + t; if (i < n && sh.act[i]) h = i; }
          tip = wave_min_i32(h);
          sh.chain[0] = (int16_t)tip; chain_len = 1;
          scan(tip, best, nearest);
        } else {
          nearest = sh.chain[chain_len - 3];  // the merged pair and the element before it leave the chain; that element is looked at again
          chain_len -= 3;
          tip = sh.chain[chain_len - 1];
          best = D[pair_sym(un, (uint32_t)tip, (uint32_t)nearest)];
        }
        for (;;) {  // until two clusters are each other's nearest neighbour; ties keep the previous chain element
          const int cur = nearest;
          sh.chain[chain_len++] = (int16_t)cur;
          double vmin; int imin;
          scan(cur, vmin, imin);
          int nn = tip;
          if (imin >= 0 && vmin < best) { best = vmin; nn = imin; }
          tip = cur; nearest = nn;
          if (nearest == sh.chain[chain_len - 2]) break;
        }
        const int lo = tip < nearest ? tip : nearest, hi = tip < nearest ? nearest : tip;
        const double s_lo = (double)sh.mem[lo], s_hi = (double)sh.mem[hi];
        for (int t = 0; t < T; ++t) {  // Lance-Williams, written into the rows of the larger index
          const int x = lane * T + t;
          if (x < n && x != lo && x != hi && sh.act[x]) {
            const double sx = (double)sh.mem[x];
            const double d_lo = D[pair_sym(un, (uint32_t)x, (uint32_t)lo)];
            double* const ph = D + pair_sym(un, (uint32_t)x, (uint32_t)hi);
            const double d_hi = *ph;
            *ph = lw_alt ? (((sx + s_lo) * d_lo) - (sx * best) + ((sx + s_hi) * d_hi)) / (sx + (s_lo + s_hi))
                         : (((sx + s_lo) * d_lo) + ((sx + s_hi) * d_hi) - (sx * best)) / (s_lo + s_hi + sx);
          }
        }
        __syncthreads();
        const int sz = sh.mem[hi] + sh.mem[lo];
        __syncthreads();
        if (lane == 0) {
          sh.mem[hi] = (int16_t)sz; sh.act[lo] = 0;
          sh.st_a[merge] = (int16_t)lo; sh.st_b[merge] = (int16_t)hi; sh.st_size[merge] = (int16_t)sz; sh.st_d[merge] = best;
        }
        __syncthreads();
      }
      // ---- stable sort of the merges by dissimilarity
      const int ns = n - 1;
      for (int t = 0; t < T; ++t) {
        const int s = lane * T + t;
        if (s < ns) {
          const double ds = sh.st_d[s];
          int rk = 0;
          for (int j = 0; j < ns; ++j) { const double dj = sh.st_d[j]; rk += (dj < ds) || (dj == ds && j < s); }
          sh.so_a[rk] = sh.st_a[s]; sh.so_b[rk] = sh.st_b[s]; sh.so_d[rk] = ds;
        }
      }
      for (int i = lane; i < 2 * n - 1; i += 64) { sh.up[i] = -1; sh.member[i] = -1; }
      __syncthreads();
      if (lane == 0) {
        // ---- SciPy labels: the clusters of a merge are named by their roots, the smaller label first
        for (int i = 0; i < ns; ++i) {
          int p = sh.so_a[i], q = sh.so_b[i];
          while (sh.up[p] >= 0) p = sh.up[p];
          while (sh.up[q] >= 0) q = sh.up[q];
          if (p > q) { const int w = p; p = q; q = w; }
          const int sp = p < n ? 1 : sh.so_size[p - n], sq = q < n ? 1 : sh.so_size[q - n];
          sh.so_a[i] = (int16_t)p; sh.so_b[i] = (int16_t)q; sh.so_size[i] = (int16_t)(sp + sq);
          sh.up[p] = sh.up[q] = (int16_t)(n + i);
        }
        for (int i = 0; i < ns; ++i) sh.so_d[i] = __builtin_sqrt(sh.so_d[i]);
        // ---- cluster() (:154-227): the cut-off below the last merge of two clusters of at least min_cluster reads
        auto csize = [&](int label) { return label < n ? 1 : (int)sh.so_size[label - n]; };
        const double mc = __builtin_round(0.01 * (double)n);
        const int min_cluster = mc > 2.0 ? (int)mc : 2;
        double cutoff = 0.0;
        for (int i = ns - 1; i >= 0; --i) {
          const int ca = csize(sh.so_a[i]), cb = csize(sh.so_b[i]);
          if ((ca < cb ? ca : cb) >= min_cluster) { cutoff = sh.so_d[i] - 0.0001; break; }
        }
        int ng = 0;
        if (cutoff == 0.0) {  // homozygous: the reads are split evenly
          for (int i = 0; i < n; ++i) sh.member[i] = (int16_t)(i & 1);
          ng = 2;
        } else {
          for (int i = ns - 1; i >= 0; --i) {
            if (!(sh.so_d[i] <= cutoff)) continue;
            int mine = sh.member[n + i];
            if (mine < 0) { mine = ng++; sh.member[n + i] = (int16_t)mine; }
            sh.member[sh.so_a[i]] = (int16_t)mine; sh.member[sh.so_b[i]] = (int16_t)mine;
          }
          for (int i = 0; i < n; ++i) if (sh.member[i] < 0) sh.member[i] = (int16_t)ng++;
        }
        // ---- the two largest groups; sort_by_key(len) is stable and pop() takes from the end: among equals the later group first
        for (int g = 0; g < ng; ++g) sh.gsz[g] = 0;
        for (int i = 0; i < n; ++i) sh.gsz[sh.member[i]] += 1;
        int g0 = 0;
        for (int g = 1; g < ng; ++g) if (sh.gsz[g] >= sh.gsz[g0]) g0 = g;
        int g1 = -1;
        for (int g = 0; g < ng; ++g) if (g != g0 && (g1 < 0 || sh.gsz[g] >= sh.gsz[g1])) g1 = g;
        for (int i = 0; i < n; ++i) sh.cls[i] = (int8_t)(sh.member[i] == g0 ? 0 : (sh.member[i] == g1 ? 1 : 2));
      }
      __syncthreads();
    }
  }
  // ---- backbones (central_read on the matrix as it is now), intervals, jobs
  int bb[2] = {0, 0}, gcnt[2] = {0, 0};
  uint32_t ci[4] = {0, 0, 0, 0};
  unsigned long long mbytes[2] = {0, 0};
  for (int g = 0; g < n_groups; ++g) {
    __syncthreads();
    if (lane == 0) { int c = 0; for (int i = 0; i < n; ++i) if (sh.cls[i] == g) sh.gm[c++] = (int16_t)i; sh.rp_ok = c; }
    __syncthreads();
    gcnt[g] = sh.rp_ok;
    bb[g] = central_read_wave<MAXR>(D, un, sh.gm, gcnt[g], lane);
    uint32_t lo = 0xFFFFFFFFu, hi = 0;
    for (int q = 0; q < gcnt[g]; ++q) { const uint32_t ln = sh.f.s_len[sh.gm[q]]; lo = ln < lo ? ln : lo; hi = ln > hi ? ln : hi; mbytes[g] += ln; }
    ci[2 * g] = lo; ci[2 * g + 1] = hi;
  }
  if (!one_group && n >= 3) {  // the even / odd split, should round 2 ask for it
    for (int g = 0; g < 2; ++g) {
      __syncthreads();
      if (lane == 0) { int c = 0; for (int i = g; i < n; i += 2) sh.gm[c++] = (int16_t)i; sh.rp_ok = c; }
      __syncthreads();
      rec.cr_eo[g] = central_read_wave<MAXR>(D, un, sh.gm, sh.rp_ok, lane);
      uint32_t lo = 0xFFFFFFFFu, hi = 0;
      for (int i = g; i < n; i += 2) { const uint32_t ln = sh.f.s_len[i]; lo = ln < lo ? ln : lo; hi = ln > hi ? ln : hi; }
      rec.ci_eo[2 * g] = lo; rec.ci_eo[2 * g + 1] = hi;
    }
  }
  // reservations: CIGAR words, result bytes and vote scratch first (a failed reservation must not leave holes in the job list)
  unsigned long long cig[2] = {0, 0}, out_need[2] = {0, 0}, scr_need[2] = {0, 0};
  uint32_t out_cap[2] = {0, 0};
  for (int g = 0; g < n_groups; ++g) {
    const uint32_t bl = sh.f.s_len[bb[g]];
    cig[g] = (unsigned long long)gcnt[g] * ((unsigned long long)bl + 1) + mbytes[g];
    out_cap[g] = (uint32_t)(bl + mbytes[g] + 16);
    out_need[g] = ((unsigned long long)out_cap[g] + 15ull) & ~15ull;
    scr_need[g] = (bl + 1 <= a.vote_lds_pos + 1 ? 0ull : 3ull * ((unsigned long long)bl + 1)) + 3ull * (unsigned long long)gcnt[g];
  }
  __syncthreads();
  if (lane == 0) {
    const unsigned long long cn = cig[0] + cig[1], on = out_need[0] + out_need[1], sn = scr_need[0] + scr_need[1];
    int ok = 1;
    unsigned long long c0 = 0, o0 = 0, s0 = 0;
    c0 = atomicAdd(reinterpret_cast<unsigned long long*>(a.counts + CC_CIGAR), cn);
    if (c0 + cn > a.cap_cigar) ok = 0;
    if (ok) { o0 = atomicAdd(reinterpret_cast<unsigned long long*>(a.counts + CC_OUT), on); if (o0 + on > a.cap_out) ok = 0; }
    if (ok) { s0 = atomicAdd(reinterpret_cast<unsigned long long*>(a.counts + CC_SCRATCH), sn); if (s0 + sn > a.cap_scratch) ok = 0; }
    if (ok) {
      sh.rp_j0 = atomicAdd(a.counts + CC_J1, (uint32_t)(gcnt[0] + gcnt[1]));
      sh.rp_g0 = atomicAdd(a.counts + CC_G1, (uint32_t)n_groups);
    } else atomicAdd(a.counts + CC_FAILED, 1u);
    sh.rp_ok = ok; sh.rp_c0 = c0; sh.rp_o0 = o0; sh.rp_s0 = s0;
  }
  __syncthreads();
  if (!sh.rp_ok) { rec.state = -1; if (lane == 0) a.rec[k] = rec; return; }
  {
    uint32_t g_at = sh.rp_g0, j_at = sh.rp_j0;
    unsigned long long co = sh.rp_c0, oo = sh.rp_o0, so = sh.rp_s0;
    for (int g = 0; g < n_groups; ++g) {
      const int b = bb[g];
      const unsigned long long bb_off = sh.f.r_off[sh.f.s_read[b]] + sh.f.s_start[b];
      const uint32_t bl = sh.f.s_len[b];
      if (lane == 0) {
        gt::RGroup G;
        G.job_first = j_at; G.n_members = (uint32_t)gcnt[g]; G.bb_len = bl; G.out_cap = out_cap[g];
        G.bb_off = bb_off; G.out_off = oo; G.scratch_off = so;
        a.groups[g_at] = G;
      }
      rec.grp[g] = (int32_t)g_at; rec.gsize[g] = gcnt[g];
      uint32_t q = 0;
      for (int i = 0; i < n; ++i) {
        if (sh.cls[i] != g) continue;
        if ((int)(q & 63u) == lane) {
          JobDev jd;
          jd.pat_off = bb_off; jd.pat_len = bl;
          jd.txt_off = sh.f.r_off[sh.f.s_read[i]] + sh.f.s_start[i]; jd.txt_len = sh.f.s_len[i];
          jd.cigar_off = co; jd.ops_off = 0; jd.out_index = j_at + q; jd.pad = 0;
          a.jobs[j_at + q] = jd;
        }
        co += (unsigned long long)bl + sh.f.s_len[i] + 1;
        ++q;
      }
      j_at += (uint32_t)gcnt[g]; oo += out_need[g]; so += scr_need[g]; ++g_at;
    }
  }
  rec.n_groups = n_groups;
  for (int q = 0; q < 4; ++q) rec.ci[q] = ci[q];
  if (lane == 0) a.rec[k] = rec;
  for (int i = lane; i < n; i += 64) a.cls[r0 + i] = sh.cls[i];
}

// ---- 3. behind the first consensus round: the homozygous redo, or the dropped reads against both alleles
template <int MAXR>
__global__ void __launch_bounds__(64) cluster_round2_kernel(const ClArgs a) {
  __shared__ ClFront<MAXR> sh;
  __shared__ int s_ok; __shared__ uint32_t s_g0, s_j0; __shared__ unsigned long long s_c0, s_o0, s_s0;
  const uint32_t k = blockIdx.x;
  if (k >= a.n_list) return;
  ClRec rec = a.rec[k];
  if (rec.state != 1) return;
  const int lane = threadIdx.x;
  for (int g = 0; g < rec.n_groups; ++g)
    if (a.vote_len[rec.grp[g]] == 0xFFFFFFFFu) { rec.state = -1; if (lane == 0) { a.rec[k] = rec; atomicAdd(a.counts + CC_FAILED, 1u); } return; }
  if (rec.n_groups != 2) return;
  const int64_t l = a.list[k];
  const uint64_t r0 = a.g.locus_read_begin[l];
  const int nr = (int)(a.g.locus_read_begin[l + 1] - r0);
  if (lane == 0) sh.n = 0;
  gt::gt_front<MAXR>(sh, a.g, r0, nr, lane);
  const int n = sh.n;
  const uint32_t l1 = a.vote_len[rec.grp[0]], l2 = a.vote_len[rec.grp[1]];
  const uint32_t c1 = (uint32_t)rec.gsize[0], c2 = (uint32_t)rec.gsize[1];
  const uint32_t cmin = c1 < c2 ? c1 : c2, cmax = c1 < c2 ? c2 : c1;
  if ((l1 > l2 ? l1 - l2 : l2 - l1) < 100u && cmin * 4u < cmax) {  // small_group_is_outlier (:84-98): redo the homozygous case
    int gcnt[2] = {(n + 1) / 2, n / 2};
    unsigned long long cig[2], out_need[2], scr_need[2], mbytes[2] = {0, 0};
    uint32_t out_cap[2];
    for (int i = 0; i < n; ++i) mbytes[i & 1] += sh.s_len[i];
    for (int g = 0; g < 2; ++g) {
      const uint32_t bl = sh.s_len[rec.cr_eo[g]];
      cig[g] = (unsigned long long)gcnt[g] * ((unsigned long long)bl + 1) + mbytes[g];
      out_cap[g] = (uint32_t)(bl + mbytes[g] + 16);
      out_need[g] = ((unsigned long long)out_cap[g] + 15ull) & ~15ull;
      scr_need[g] = (bl + 1 <= a.vote_lds_pos + 1 ? 0ull : 3ull * ((unsigned long long)bl + 1)) + 3ull * (unsigned long long)gcnt[g];
    }
    if (lane == 0) {
      const unsigned long long cn = cig[0] + cig[1], on = out_need[0] + out_need[1], sn = scr_need[0] + scr_need[1];
      int ok = 1;
      unsigned long long c0 = 0, o0 = 0, s0 = 0;
      c0 = atomicAdd(reinterpret_cast<unsigned long long*>(a.counts + CC_CIGAR), cn);
      if (c0 + cn > a.cap_cigar) ok = 0;
      if (ok) { o0 = atomicAdd(reinterpret_cast<unsigned long long*>(a.counts + CC_OUT), on); if (o0 + on > a.cap_out) ok = 0; }
      if (ok) { s0 = atomicAdd(reinterpret_cast<unsigned long long*>(a.counts + CC_SCRATCH), sn); if (s0 + sn > a.cap_scratch) ok = 0; }
      if (ok) { s_j0 = atomicAdd(a.counts + CC_J2, (uint32_t)n); s_g0 = atomicAdd(a.counts + CC_G2, 2u); }
      else atomicAdd(a.counts + CC_FAILED, 1u);
      s_ok = ok; s_c0 = c0; s_o0 = o0; s_s0 = s0;
    }
    __syncthreads();
    if (!s_ok) { rec.state = -1; if (lane == 0) a.rec[k] = rec; return; }
    uint32_t g_at = a.cap_g + s_g0, j_at = a.cap_j + s_j0;
    unsigned long long co = s_c0, oo = s_o0, so = s_s0;
    for (int g = 0; g < 2; ++g) {
      const int b = rec.cr_eo[g];
      const unsigned long long bb_off = sh.r_off[sh.s_read[b]] + sh.s_start[b];
      const uint32_t bl = sh.s_len[b];
      if (lane == 0) {
        gt::RGroup G;
        G.job_first = j_at; G.n_members = (uint32_t)gcnt[g]; G.bb_len = bl; G.out_cap = out_cap[g];
        G.bb_off = bb_off; G.out_off = oo; G.scratch_off = so;
        a.groups[g_at] = G;
      }
      rec.grp[g] = (int32_t)g_at; rec.gsize[g] = gcnt[g];
      uint32_t q = 0;
      for (int i = g; i < n; i += 2) {
        if ((int)(q & 63u) == lane) {
          JobDev jd;
          jd.pat_off = bb_off; jd.pat_len = bl;
          jd.txt_off = sh.r_off[sh.s_read[i]] + sh.s_start[i]; jd.txt_len = sh.s_len[i];
          jd.cigar_off = co; jd.ops_off = 0; jd.out_index = j_at + q; jd.pad = 0;
          a.jobs[j_at + q] = jd;
        }
        co += (unsigned long long)bl + sh.s_len[i] + 1;
        ++q;
      }
      j_at += (uint32_t)gcnt[g]; oo += out_need[g]; so += scr_need[g]; ++g_at;
    }
    rec.redo = 1;
    for (int q = 0; q < 4; ++q) rec.ci[q] = rec.ci_eo[q];
    if (lane == 0) a.rec[k] = rec;
    return;
  }
  // the reads cluster() dropped go to the closer consensus (:117-142): their edit distances to both alleles
  const uint64_t aoff[2] = {a.groups[rec.grp[0]].out_off, a.groups[rec.grp[1]].out_off};
  const uint32_t alen[2] = {l1, l2};
  for (int ib = 0; ib < n; ib += 64) {
    const int i = ib + lane;
    const bool out = i < n && a.cls[r0 + i] == 2;
    const uint32_t li = i < n ? sh.s_len[i] : 0u;
    uint32_t want = 0;
    if (out) for (int q = 0; q < 2; ++q) want += (uint64_t)li * (uint64_t)alen[q] <= CL_MAX_OPS;
    uint32_t inc = want;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    const uint32_t total = __shfl(inc, 63);
    uint32_t base = 0;
    if (total) { if (lane == 0) base = atomicAdd(a.counts + CC_ED2, total); base = __shfl(base, 0); }
    if (out) {
      uint32_t at = base + inc - want;
      for (int q = 0; q < 2; ++q) {
        const uint64_t slot = 2 * (r0 + (uint64_t)i) + (uint64_t)q;
        if ((uint64_t)li * (uint64_t)alen[q] <= CL_MAX_OPS) {
          JobDev jd;
          jd.pat_off = sh.r_off[sh.s_read[i]] + sh.s_start[i]; jd.pat_len = li;
          jd.txt_off = aoff[q]; jd.txt_len = alen[q];
          jd.cigar_off = 0; jd.ops_off = 0; jd.out_index = (uint32_t)slot; jd.pad = 0;
          a.ed2_jobs[at++] = jd;
        } else a.escore2[slot] = (int32_t)(li > alen[q] ? li - alen[q] : alen[q] - li);
      }
    }
  }
}

// ---- 4. genotype, classifications, allele order (:143-151, :99-115, :62-73), reference allele first (tr.rs:95-101), outputs
template <int MAXR>
__global__ void __launch_bounds__(64) cluster_finish_kernel(const ClArgs a) {
  __shared__ ClFront<MAXR> sh;
  __shared__ int8_t s_cls[MAXR];
  const uint32_t k = blockIdx.x;
  if (k >= a.n_list) return;
  const ClRec rec = a.rec[k];
  const int64_t l = a.list[k];
  const int lane = threadIdx.x;
  if (rec.state == 0) { if (lane == 0 && a.g.ploidy[l] != 0 && a.g.locus_read_begin[l + 1] != a.g.locus_read_begin[l]) a.g.need_host[l] = 0; return; }  // no spanning read: the empty result stands
  if (rec.state != 1) return;
  for (int g = 0; g < rec.n_groups; ++g)
    if (a.vote_len[rec.grp[g]] == 0xFFFFFFFFu) { if (lane == 0) atomicAdd(a.counts + CC_FAILED, 1u); return; }
  const uint64_t r0 = a.g.locus_read_begin[l];
  const int nr = (int)(a.g.locus_read_begin[l + 1] - r0);
  if (lane == 0) sh.n = 0;
  gt::gt_front<MAXR>(sh, a.g, r0, nr, lane);
  const int n = sh.n;
  const int ploidy = a.g.ploidy[l] == 1 ? 1 : 2;
  const uint8_t* ap[2] = {nullptr, nullptr}; uint32_t aln[2] = {0, 0}; uint32_t civ[4] = {0, 0, 0, 0};
  for (int g = 0; g < rec.n_groups; ++g) {
    ap[g] = a.vote_out + a.groups[rec.grp[g]].out_off; aln[g] = a.vote_len[rec.grp[g]];
    civ[2 * g] = rec.ci[2 * g]; civ[2 * g + 1] = rec.ci[2 * g + 1];
  }
  int n_gt;
  for (int i = lane; i < n; i += 64) {
    int cc = 0;
    if (rec.n_groups == 2) {
      if (rec.redo) cc = i & 1;
      else {
        cc = a.cls[r0 + i];
        if (cc == 2) {  // tie_breaker starts at 1 for every read (:125): an exact tie resolves to (1 + 1) % 2 = 0
          const int32_t d1 = a.escore2[2 * (r0 + (uint64_t)i)], d2 = a.escore2[2 * (r0 + (uint64_t)i) + 1];
          cc = d1 < d2 ? 0 : (d2 < d1 ? 1 : 0);
        }
      }
    }
    s_cls[i] = (int8_t)cc;
  }
  __syncthreads();
  bool swapped = false;
  if (rec.n_groups == 1) {
    if (ploidy == 1) n_gt = 1;
    else { n_gt = 2; ap[1] = ap[0]; aln[1] = aln[0]; civ[2] = civ[0]; civ[3] = civ[1]; }  // one read, two alleles (:70-72)
  } else {
    n_gt = 2;
    if (aln[0] > aln[1]) {
      swapped = true;
      const uint8_t* tp = ap[0]; ap[0] = ap[1]; ap[1] = tp;
      const uint32_t tl = aln[0]; aln[0] = aln[1]; aln[1] = tl;
      uint32_t t0 = civ[0]; civ[0] = civ[2]; civ[2] = t0; t0 = civ[1]; civ[1] = civ[3]; civ[3] = t0;
    }
  }
  int by_hap[2] = {0, 0};
  for (int i = 0; i < n; ++i) { const int cc = swapped ? 1 - s_cls[i] : s_cls[i]; by_hap[cc] += 1; }
  const uint8_t* ref = a.g.tr_blob + a.g.tr_off[l]; const uint32_t refn = a.g.tr_len[l];
  int order[2] = {0, 1}, flip = 0;
  if (n_gt != 1 && !gt::wave_equal(ap[0], aln[0], ref, refn) && gt::wave_equal(ap[1], aln[1], ref, refn)) { order[0] = 1; order[1] = 0; flip = 1; }
  for (int oi = 0; oi < n_gt; ++oi) if (aln[order[oi]] > a.g.allele_cap[l]) { if (lane == 0) atomicAdd(a.counts + CC_FAILED, 1u); return; }  // the host path reports the error
  for (int oi = 0; oi < n_gt; ++oi) {
    const int al = order[oi];
    uint8_t* dst = a.g.allele_blob + a.g.allele_off[2 * l + oi];
    for (uint32_t b = lane; b < aln[al]; b += 64) dst[b] = ap[al][b];
    if (lane == 0) {
      a.g.allele_len[2 * l + oi] = aln[al];
      a.g.ci[4 * l + 2 * oi] = (int32_t)civ[2 * al]; a.g.ci[4 * l + 2 * oi + 1] = (int32_t)civ[2 * al + 1];
      a.g.num_spanning[2 * l + oi] = by_hap[al];
      if (a.g.gt_size) a.g.gt_size[2 * l + oi] = (int32_t)aln[al];  // the cluster genotyper's sizes are its allele lengths
    }
  }
  for (int i = lane; i < n; i += 64) {
    const int cc = swapped ? 1 - s_cls[i] : s_cls[i];
    a.g.classification[r0 + sh.s_read[i]] = flip ? 1 - cc : cc;
    a.g.read_rank[r0 + sh.s_read[i]] = i;
  }
  if (lane == 0) {
    a.g.n_alleles[l] = n_gt; a.g.n_spanning_reads[l] = (uint32_t)n; a.g.flipped[l] = (uint8_t)flip; a.g.need_host[l] = 0;
    atomicAdd(a.counts + CC_DONE, 1u);
  }
}

}  // namespace cl
}  // namespace trgt
