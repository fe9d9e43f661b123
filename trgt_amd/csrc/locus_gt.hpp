// trgt_amd/csrc/locus_gt.hpp -- the host glue of analyze_tr on the device, for the common case:
//   get_spanning_reads  (src/trgt/workflows/tr.rs:111-184)  filter, stable sort by span length, uniform downsample
//   genotype_size::genotype (genotype_size.rs:6-64) with diploid::genotype (diploid.rs:5-103) / haploid::genotype
//   (haploid.rs:3-30), consensus::get_consensus (consensus.rs:113-154), the read classification (genotype_size.rs:42-61)
//   and "reference allele first" (tr.rs:95-101).
// One wavefront per locus; everything the decisions depend on (span lengths, the repeat segments themselves) is staged in
// LDS.  A locus is handed back to the host path (need_host = 1) when it is out of this kernel's envelope: more than
// GT_MAX_READS reads, more than GT_SEG_LDS segment bytes, or an allele without majority support, which needs the consensus
// alignments of stage B (repair_consensus, consensus.rs:5-111).  The arithmetic (f64 penalties, tie-breaks) is written
// operation for operation like the host version in locus.hip, which the oracle pins.
#pragma once
#include "common.hpp"
#include "wfa_host.hpp"

namespace trgt {
namespace gt {

// Two instantiations: <256, 16 KB> and, for batches whose loci all have at most 64 reads, <64, 8 KB> -- 11 instead of 27 KB of LDS
// per locus, i.e. 14 instead of 5 loci in flight per CU for a kernel that is all latency (0.53 -> 0.3 ms on the 10k-locus batch).
constexpr int GT_MAX_READS = 256;      // reads of a locus (and therefore kept spanning reads) handled by the large instantiation
constexpr int GT_SEG_LDS = 16 * 1024;  // bytes of repeat segments staged per locus (large instantiation)

// ---- consensus repair on the device (genotype_size.rs:32-37 -> utils::align -> repair_consensus, consensus.rs:5-111): a locus whose pick
// lacks majority support no longer goes back to the host.  The genotyper writes the consensus alignments it needs (backbone = the pick,
// texts = the unique sequences of the allele's group, all of them segments of the read blob) into a job list, one vote group per allele
// and a record of what it had decided; behind it run the alignment kernel over that list, the column voting and repair_finish_kernel,
// which classifies the reads against the repaired alleles and writes the locus out.  Space is handed out with atomic counters; a locus
// that finds no room (or is out of the envelope: a segment longer than max_seg) takes the host path as before.
struct RGroup {  // = vote::Group (consensus_vote.hpp; the layouts are asserted equal in locus.hip)
  uint32_t job_first, n_members, bb_len, out_cap;
  uint64_t bb_off, out_off, scratch_off;
};
struct RepairPend {  // what the genotyper had decided for a locus that waits for its repaired alleles
  int32_t n_gt, n_pick; uint32_t size[2]; int32_t civ[4]; int32_t rep[2] /* rank of the pick */; int32_t grp[2] /* vote group, -1: the pick stands */;
};
enum { RC_GROUPS = 0, RC_JOBS = 1, RC_LOCI = 2, RC_FAILED = 3, RC_CIGAR = 4 /* u64 */, RC_OUT = 6 /* u64 */, RC_SCRATCH = 8 /* u64 */, RC_REFUSED = 10 /* alignment jobs of the chain the generic kernel refused (beyond its planned workspace) */, RC_WORDS = 16 };
struct RepairBufs {
  uint32_t* counts;  // [RC_WORDS]; nullptr: no device-side repair (every such locus takes the host path)
  RGroup* groups; JobDev* jobs; uint32_t* loci; RepairPend* pend;
  uint32_t cap_groups, cap_jobs, max_seg, vote_lds_pos;
  uint64_t cap_cigar, cap_out, cap_scratch;
};

struct GtArgs {
  const uint8_t* reads; const uint64_t* read_off; const uint32_t* read_len; const uint64_t* locus_read_begin;
  const int32_t* span_start; const int32_t* span_end;
  const uint8_t* ploidy; const uint8_t* tr_blob; const uint64_t* tr_off; const uint32_t* tr_len;
  const uint64_t* allele_off; const uint32_t* allele_cap;
  int64_t n_loci; int32_t flank_len, max_depth;
  uint8_t* need_host; int32_t* n_alleles; uint8_t* allele_blob; uint32_t* allele_len; int32_t* ci; int32_t* num_spanning;
  int32_t* classification; int32_t* read_rank; uint32_t* n_spanning_reads;
  uint8_t* flipped;  // per locus: the two alleles were swapped to put the reference allele first
  int32_t* gt_size;  // [2 n_loci] TrSize::size of the genotype behind every allele (optional)
  const uint8_t* genotyper;  // optional [n_loci]: 1 = Genotyper::Cluster, a locus the host genotypes (need_host = 1 at once)
  uint8_t* skip_b;   // [n_loci] 0 for the loci repair_finish_kernel completed (their HMM batch runs behind it), else 1
  int32_t finish_clears_need;  // 1: repair_finish_kernel also sets need_host = 0 (nothing reads it concurrently: one HMM batch behind the repair)
  RepairBufs rp;
};

template <int MAXR, int SEG>
struct GtShared {
  uint32_t r_s[MAXR], r_len[MAXR];     // per read of the locus: span start / span length (0xFFFFFFFF start = not kept)
  uint64_t r_off[MAXR];                        // per read: byte offset of the read in the blob
  uint32_t s_read[MAXR], s_start[MAXR], s_len[MAXR];  // kept reads in LocusResult.reads order
  uint16_t s_loff[MAXR];                                              // offset of the segment bytes in `bytes` (4-aligned)
  uint16_t u_rep[MAXR], u_cnt[MAXR];  // unique sequences (lexicographic order): representative, multiplicity
  uint32_t ulen[MAXR], ucnt[MAXR];    // unique lengths ascending, multiplicities
  int8_t cls[MAXR];
  int n, n_sizes, bail, fits;
  // decisions of lane 0, written out by the whole wave
  int res_n_gt, res_flip, res_rep[2], res_ci[4], res_hap[2];
  // device-side repair: reservations of lane 0
  int rp_ok; uint32_t rp_g0, rp_j0; unsigned long long rp_c0, rp_o0, rp_s0;
  uint32_t ref_off;                                    // the reference repeat staged behind the segments
  alignas(16) uint8_t bytes[SEG];
};

__device__ __forceinline__ uint32_t adiff_u(uint32_t a, uint32_t b) { return a > b ? a - b : b - a; }

// cmp_seg of locus.hip (memcmp over the common prefix, then the shorter one first) on 4-aligned LDS copies, by the whole wave:
// lane i compares dword i (64 dwords per round), a ballot finds the first difference; byte order is memory order, so the
// deciding dwords are compared byte-swapped.  Uniform arguments, uniform result.
__device__ __forceinline__ int cmp_lds(const uint8_t* a, uint32_t na, const uint8_t* b, uint32_t nb) {
  const int lane = threadIdx.x & 63;
  const uint32_t m = na < nb ? na : nb;
  const uint32_t* A = reinterpret_cast<const uint32_t*>(a); const uint32_t* B = reinterpret_cast<const uint32_t*>(b);
  const uint32_t nw = (m + 3u) >> 2;
  for (uint32_t base = 0; base < nw; base += 64) {
    const uint32_t w = base + (uint32_t)lane;
    uint32_t x = 0, y = 0;
    if (w < nw) {
      x = A[w]; y = B[w];
      if (w == nw - 1 && (m & 3u)) { const uint32_t mask = (1u << (8 * (m & 3u))) - 1u; x &= mask; y &= mask; }
    }
    const unsigned long long ne = __ballot(x != y);
    if (ne) {
      const int f = __ffsll((long long)ne) - 1;
      const uint32_t xf = (uint32_t)__builtin_amdgcn_readlane((int)x, f), yf = (uint32_t)__builtin_amdgcn_readlane((int)y, f);
      return __builtin_bswap32(xf) < __builtin_bswap32(yf) ? -1 : 1;
    }
  }
  return na < nb ? -1 : (na > nb ? 1 : 0);
}

// the same order on byte strings anywhere (global memory, any alignment): lane i assembles dword i from four byte loads.  For the
// loci whose repeat segments do not fit the LDS staging area (long alleles, deep loci): a few per cent of a genome-wide catalog.
__device__ __forceinline__ int cmp_bytes(const uint8_t* __restrict__ a, uint32_t na, const uint8_t* __restrict__ b, uint32_t nb) {
  const int lane = threadIdx.x & 63;
  const uint32_t m = na < nb ? na : nb;
  for (uint32_t base = 0; base < m; base += 256) {
    uint32_t x = 0, y = 0;  // big-endian: the first byte is the most significant
    for (uint32_t q = 0; q < 4; ++q) {
      const uint32_t i = base + 4u * (uint32_t)lane + q;
      if (i < m) { x |= (uint32_t)a[i] << (24 - 8 * q); y |= (uint32_t)b[i] << (24 - 8 * q); }
    }
    const unsigned long long ne = __ballot(x != y);
    if (ne) {
      const int f = __ffsll((long long)ne) - 1;
      const uint32_t xf = (uint32_t)__builtin_amdgcn_readlane((int)x, f), yf = (uint32_t)__builtin_amdgcn_readlane((int)y, f);
      return xf < yf ? -1 : 1;
    }
  }
  return na < nb ? -1 : (na > nb ? 1 : 0);
}

// 16 lanes copy n bytes from global (any alignment) to a 4-aligned LDS destination, 16 bytes per lane and round
__device__ __forceinline__ void copy16(uint8_t* dst, const uint8_t* __restrict__ src, uint32_t n, int sub) {
  const uint32_t full = n >> 4;
  for (uint32_t w = sub; w < full; w += 16) {
    uint4 v; __builtin_memcpy(&v, src + 16 * w, 16);
    uint32_t* d = reinterpret_cast<uint32_t*>(dst) + 4 * w;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  for (uint32_t b = full * 16 + (uint32_t)sub; b < n; b += 16) dst[b] = src[b];
}

// get_spanning_reads (tr.rs:111-184) for one locus by one wave: filter, stable sort by span length, uniform downsample.  On return
// sh.n kept reads sit in sh.s_read / s_start / s_len in LocusResult.reads order (sh.r_off: blob offsets of all reads of the locus).
template <int MAXR, class SH>
__device__ __forceinline__ void gt_front(SH& sh, const GtArgs& a, uint64_t r0, int nr, int lane) {
  const int F = a.flank_len;
  for (int i = lane; i < nr; i += 64) {
    const int32_t s = a.span_start[r0 + i], e = a.span_end[r0 + i];
    const bool keep = s >= 0 && s >= F && (int64_t)a.read_len[r0 + i] - e >= F;  // filter of get_spanning_reads (tr.rs:139-145)
    sh.r_s[i] = keep ? (uint32_t)s : 0xFFFFFFFFu; sh.r_len[i] = keep ? (uint32_t)(e - s) : 0u;
    sh.r_off[i] = a.read_off[r0 + i];
  }
  __syncthreads();
  if (lane == 0) {  // in read order
    int n = 0;
    for (int i = 0; i < nr; ++i)
      if (sh.r_s[i] != 0xFFFFFFFFu) { sh.s_read[n] = (uint32_t)i; sh.s_start[n] = sh.r_s[i]; sh.s_len[n] = sh.r_len[i]; ++n; }
    sh.n = n;
  }
  __syncthreads();
  const int n = sh.n;
  if (n > 0) {
    // ---- stable sort by span length (:157): rank = #{shorter} + #{equal and earlier}; every lane owns elements lane, lane+64, ...
    {
      uint32_t rd[MAXR / 64], st[MAXR / 64], ln[MAXR / 64]; int rk[MAXR / 64];
      for (int t = 0; t < MAXR / 64; ++t) {
        const int i = lane + 64 * t;
        rk[t] = -1;
        if (i < n) {
          rd[t] = sh.s_read[i]; st[t] = sh.s_start[i]; ln[t] = sh.s_len[i];
          int r = 0;
          for (int j = 0; j < n; ++j) { const uint32_t lj = sh.s_len[j]; r += (lj < ln[t]) || (lj == ln[t] && j < i); }
          rk[t] = r;
        }
      }
      __syncthreads();
      for (int t = 0; t < MAXR / 64; ++t)
        if (rk[t] >= 0) { sh.s_read[rk[t]] = rd[t]; sh.s_start[rk[t]] = st[t]; sh.s_len[rk[t]] = ln[t]; }
      __syncthreads();
    }
    if (lane == 0) {
      // ---- uniform downsample (:172-184), sequential swaps exactly as written there
      if (n > a.max_depth) {
        const double step = (double)n / (double)a.max_depth;
        double fast = 0.0;
        for (int i = 0; i < a.max_depth; ++i) {
          const int ind = (int)floor(fast);
          if (ind != i) {
            const uint32_t t0 = sh.s_read[i], t1 = sh.s_start[i], t2 = sh.s_len[i];
            sh.s_read[i] = sh.s_read[ind]; sh.s_start[i] = sh.s_start[ind]; sh.s_len[i] = sh.s_len[ind];
            sh.s_read[ind] = t0; sh.s_start[ind] = t1; sh.s_len[ind] = t2;
          }
          fast += step;
        }
        sh.n = a.max_depth;
      }
    }
    __syncthreads();
  }
}

template <int MAXR, int SEG>
__global__ void __launch_bounds__(64) locus_genotype_kernel(const GtArgs a) {  // exactly one wave per locus: the lane-0 sections and the ballots rely on it
  __shared__ GtShared<MAXR, SEG> sh;
  const int64_t l = blockIdx.x;
  if (l >= a.n_loci) return;
  const int lane = threadIdx.x;
  const uint64_t r0 = a.locus_read_begin[l], r1 = a.locus_read_begin[l + 1];
  const int nr = (int)(r1 - r0);
  if (lane == 0) {
    sh.n = 0; sh.bail = 0; sh.fits = 1; sh.res_n_gt = 0;
    if (a.gt_size) a.gt_size[2 * l] = a.gt_size[2 * l + 1] = 0;
    if (a.skip_b) a.skip_b[l] = 1;
    a.need_host[l] = 0; a.n_alleles[l] = 0; a.n_spanning_reads[l] = 0; a.flipped[l] = 0;
    a.allele_len[2 * l] = a.allele_len[2 * l + 1] = 0; a.num_spanning[2 * l] = a.num_spanning[2 * l + 1] = 0;
  }
  const bool cluster = a.genotyper && a.genotyper[l] == 1;
  if (a.ploidy[l] == 0 || nr == 0 || nr > MAXR || cluster) {  // Ploidy::Zero -> LocusResult::empty (tr.rs:29-31); oversized, cluster genotyper -> host
    for (int i = lane; i < nr; i += 64) { a.classification[r0 + i] = -1; a.read_rank[r0 + i] = -1; }
    if (lane == 0 && (nr > MAXR || cluster) && a.ploidy[l] != 0 && nr != 0) a.need_host[l] = 1;
    return;
  }
  gt_front<MAXR>(sh, a, r0, nr, lane);
  const uint32_t refn = a.tr_len[l];
  const uint64_t refo = a.tr_off[l];
  int n = sh.n;
  if (n > 0) {
    if (lane == 0) {
      // ---- LDS layout of the repeat segments (4-aligned) and of the reference repeat behind them
      uint32_t o = 0;
      const int nn = sh.n;
      //      (segments that do not fit are compared where they lie, in the read blob: sh.fits = 0)
      for (int i = 0; i < nn; ++i) { sh.s_loff[i] = (uint16_t)o; o += (sh.s_len[i] + 3u) & ~3u; if (o > (uint32_t)SEG) { sh.fits = 0; break; } }
      sh.ref_off = o;
      if (o + ((refn + 3u) & ~3u) > (uint32_t)SEG) sh.fits = 0;
    }
    __syncthreads();
    n = sh.n;
    if (sh.fits) {
      const int grp = lane >> 4, sub = lane & 15;
      for (int i = grp; i < n; i += 4) copy16(sh.bytes + sh.s_loff[i], a.reads + sh.r_off[sh.s_read[i]] + sh.s_start[i], sh.s_len[i], sub);
      if (grp == 0) copy16(sh.bytes + sh.ref_off, a.tr_blob + refo, refn, sub);
    }
    __syncthreads();
  }
  const bool fits = sh.fits != 0;
  auto seg_glob = [&](int i) { return a.reads + sh.r_off[sh.s_read[i]] + sh.s_start[i]; };
  auto cmp_seg = [&](int i, int j) {  // uniform arguments, uniform result
    return fits ? cmp_lds(sh.bytes + sh.s_loff[i], sh.s_len[i], sh.bytes + sh.s_loff[j], sh.s_len[j]) : cmp_bytes(seg_glob(i), sh.s_len[i], seg_glob(j), sh.s_len[j]);
  };
  if (n > 0 && !sh.bail) {
    // ---- unique lengths / counts ascending (sorted(lens) of genotype_size_front; after a downsample the list is not sorted)
    if (lane == 0) {  // (one lane builds the histogram: read-modify-writes of LDS by 64 lanes at once only work by lockstep)
      int u = 0;
      for (int i = 0; i < n; ++i) {
        const uint32_t v = sh.s_len[i];
        int p = 0;
        while (p < u && sh.ulen[p] < v) ++p;
        if (p < u && sh.ulen[p] == v) { sh.ucnt[p] += 1; continue; }
        for (int q = u; q > p; --q) { sh.ulen[q] = sh.ulen[q - 1]; sh.ucnt[q] = sh.ucnt[q - 1]; }
        sh.ulen[p] = v; sh.ucnt[p] = 1; ++u;
      }
      sh.n_sizes = u;
    }
    __syncthreads();
    const int u = sh.n_sizes;
    const int ploidy = a.ploidy[l] == 1 ? 1 : 2;
    // ---- diploid::genotype / haploid::genotype: candidates spread over the lanes, first minimum wins
    double best_pen = 0.0; int best_p = 0x7FFFFFFF; bool have = false;
    if (ploidy == 2) {
      int p = 0;
      for (int si = 0; si < u; ++si)
        for (int li = si; li < u; ++li, ++p) {
          if ((p & 63) != lane) continue;
          const uint32_t sa = sh.ulen[si], la = sh.ulen[li];
          const double max_frac = adiff_u(sa, la) <= 100 ? 0.25 : 0.05;
          double pen = 0.0;
          for (int i = 0; i < u; ++i) {
            const uint32_t st = sh.ulen[i] != sa ? 10 + 2 * adiff_u(sa, sh.ulen[i]) : 0, lt = sh.ulen[i] != la ? 10 + 2 * adiff_u(la, sh.ulen[i]) : 0;
            const double term = (double)(st < lt ? st : lt) + max_frac * (double)(st > lt ? st : lt);
            pen += term * (double)sh.ucnt[i];
          }
          if (!have || pen < best_pen) { have = true; best_pen = pen; best_p = p; }
        }
    } else {
      for (int c = lane; c < u; c += 64) {
        double pen = 0.0;
        for (int i = 0; i < u; ++i) {
          const double term = sh.ulen[i] != sh.ulen[c] ? 10.0 + 2.0 * (double)adiff_u(sh.ulen[c], sh.ulen[i]) : 0.0;
          pen += term * (double)sh.ucnt[i];
        }
        if (!have || pen < best_pen) { have = true; best_pen = pen; best_p = c; }
      }
    }
    for (int o = 32; o > 0; o >>= 1) {  // (penalty, candidate index) minimum over the wave: the first minimum in candidate order
      const double op = __shfl_xor(best_pen, o); const int oi = __shfl_xor(best_p, o); const int oh = __shfl_xor((int)have, o);
      if (oh && (!have || op < best_pen || (op == best_pen && oi < best_p))) { have = true; best_pen = op; best_p = oi; }
    }
    // ---- the decisions are sequential and small: every lane takes them redundantly (uniform control flow, identical LDS
    //      writes), which lets the sequence comparisons use the whole wave
    {
      uint32_t size[2] = {0, 0}, civ[4] = {0, 0, 0, 0};
      int n_gt;
      if (ploidy == 2) {
        int si = 0, li = 0, p = 0;
        for (int x = 0; x < u; ++x) for (int y = x; y < u; ++y, ++p) if (p == best_p) { si = x; li = y; }
        const uint32_t bs = sh.ulen[si], bl = sh.ulen[li];
        uint32_t short_size = bs < bl ? bs : bl, long_size = bs > bl ? bs : bl;
        if (short_size != long_size && u >= 2) {
          uint64_t coverage = 0; int top = 0;
          for (int i = 0; i < u; ++i) { coverage += sh.ucnt[i]; if (sh.ucnt[i] > sh.ucnt[top]) top = i; }  // stable desc sort, first
          const double top_frac = (double)sh.ucnt[top] / (double)coverage;
          const uint32_t range = sh.ulen[u - 1] - sh.ulen[0];
          if (top_frac > 0.60 && range <= 6) short_size = long_size = sh.ulen[top];
        }
        n_gt = 2; size[0] = short_size; size[1] = long_size;
        civ[0] = civ[1] = short_size; civ[2] = civ[3] = long_size;
        for (int i = 0; i < u; ++i) {
          const uint32_t s = sh.ulen[i];
          if (adiff_u(s, short_size) <= adiff_u(s, long_size)) { civ[0] = civ[0] < s ? civ[0] : s; civ[1] = civ[1] > s ? civ[1] : s; }
          else { civ[2] = civ[2] < s ? civ[2] : s; civ[3] = civ[3] > s ? civ[3] : s; }
        }
      } else {
        n_gt = 1; size[0] = sh.ulen[best_p];
        civ[0] = sh.ulen[0]; civ[1] = sh.ulen[u - 1];
      }
      // ---- get_seq_hist: unique sequences in byte-lexicographic order (binary-search insertion into a sorted unique list)
      int nu = 0;
      for (int i = 0; i < n; ++i) {
        int lo = 0, hi = nu, eq = -1;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          const int r = sh.u_rep[mid];
          const int c = cmp_seg(i, r);
          if (c == 0) { eq = mid; break; }
          if (c < 0) hi = mid; else lo = mid + 1;
        }
        // (the list is written by lane 0 only; the comparisons of the next round read it back in program order: one wave, one LDS queue)
        if (eq >= 0) { if (lane == 0) sh.u_cnt[eq] += 1; continue; }
        if (lane == 0) {
          for (int q = nu; q > lo; --q) { sh.u_rep[q] = sh.u_rep[q - 1]; sh.u_cnt[q] = sh.u_cnt[q - 1]; }
          sh.u_rep[lo] = (uint16_t)i; sh.u_cnt[lo] = 1;
        }
        ++nu;
      }
      auto ulen_of = [&](int q) { return sh.s_len[sh.u_rep[q]]; };
      auto closest = [&](uint32_t target) { uint32_t c = ulen_of(0); for (int q = 0; q < nu; ++q) if (adiff_u(c, target) > adiff_u(ulen_of(q), target)) c = ulen_of(q); return c; };
      auto most_frequent = [&](uint32_t len) { int best = -1; for (int q = 0; q < nu; ++q) if (ulen_of(q) == len && (best < 0 || sh.u_cnt[q] >= sh.u_cnt[best])) best = q; return best; };
      int pick[2] = {most_frequent(closest(size[0])), -1};
      int n_pick = 1;
      if (n_gt != 1 && size[0] != size[1]) { pick[1] = most_frequent(closest(size[1])); n_pick = 2; }
      bool majority = true, lacks[2] = {false, false};
      auto in_group = [&](int q, int al) {
        if (n_gt == 1) return true;
        const uint32_t d1 = adiff_u(ulen_of(q), size[0]), d2 = adiff_u(ulen_of(q), size[1]);
        return al == 0 ? d1 <= d2 : d2 < d1;
      };
      for (int al = 0; al < n_pick; ++al) {  // split(): majority support of the pick inside its group, else stage B is needed
        uint64_t coverage = 0, ref_count = 0;
        for (int q = 0; q < nu; ++q) {
          if (!in_group(q, al)) continue;
          coverage += sh.u_cnt[q];
          if (q == pick[al]) ref_count = sh.u_cnt[q];
        }
        if (!(2 * ref_count >= coverage)) { majority = false; lacks[al] = true; }
      }
      if (!majority) {
        // ---- stage B on the device: one vote group per allele without majority support, one alignment job per unique sequence of
        //      its group against the pick (the members of make_consensus, genotype_size.rs:32-37), all segments of the read blob
        const RepairBufs& rp = a.rp;
        bool can = rp.counts != nullptr;
        uint32_t nm[2] = {0, 0}; unsigned long long mbytes[2] = {0, 0}, cig[2] = {0, 0};
        if (can)
          for (int al = 0; al < n_pick; ++al) {
            if (!lacks[al]) continue;
            const uint32_t bb = ulen_of(pick[al]);
            if (bb > rp.max_seg) can = false;
            for (int q = 0; q < nu; ++q) {
              if (!in_group(q, al)) continue;
              const uint32_t ln = ulen_of(q);
              if (ln > rp.max_seg) can = false;
              nm[al] += 1; mbytes[al] += ln; cig[al] += (unsigned long long)bb + ln + 1;
            }
          }
        unsigned long long out_need[2] = {0, 0}, scr_need[2] = {0, 0};
        uint32_t out_cap[2] = {0, 0};
        if (can) {
          for (int al = 0; al < n_pick; ++al) {
            if (!lacks[al]) continue;
            const uint32_t bb = ulen_of(pick[al]);
            // at most one base per backbone position plus the insertions taken, each of which is a piece of some member
            out_cap[al] = (uint32_t)(bb + mbytes[al] + 16);
            out_need[al] = ((unsigned long long)out_cap[al] + 15ull) & ~15ull;
            scr_need[al] = (bb + 1 <= rp.vote_lds_pos + 1 ? 0ull : 3ull * ((unsigned long long)bb + 1)) + 3ull * nm[al];
          }
          if (lane == 0) {  // cigar words, result bytes and vote scratch first: a failed reservation must not leave holes in the job list
            const unsigned long long cn = cig[0] + cig[1], on = out_need[0] + out_need[1], sn = scr_need[0] + scr_need[1];
            int ok = 1;
            unsigned long long c0 = 0, o0 = 0, s0 = 0;
            c0 = atomicAdd(reinterpret_cast<unsigned long long*>(rp.counts + RC_CIGAR), cn);
            if (c0 + cn > rp.cap_cigar) ok = 0;
            if (ok) { o0 = atomicAdd(reinterpret_cast<unsigned long long*>(rp.counts + RC_OUT), on); if (o0 + on > rp.cap_out) ok = 0; }
            if (ok) { s0 = atomicAdd(reinterpret_cast<unsigned long long*>(rp.counts + RC_SCRATCH), sn); if (s0 + sn > rp.cap_scratch) ok = 0; }
            if (ok) {
              sh.rp_j0 = atomicAdd(rp.counts + RC_JOBS, nm[0] + nm[1]);
              sh.rp_g0 = atomicAdd(rp.counts + RC_GROUPS, (uint32_t)lacks[0] + (uint32_t)lacks[1]);
              rp.loci[atomicAdd(rp.counts + RC_LOCI, 1u)] = (uint32_t)l;
              if (sh.rp_j0 + nm[0] + nm[1] > rp.cap_jobs || sh.rp_g0 + 2 > rp.cap_groups) ok = 0;  // (cannot happen: the caps are the read and locus counts)
            } else atomicAdd(rp.counts + RC_FAILED, 1u);
            sh.rp_ok = ok; sh.rp_c0 = c0; sh.rp_o0 = o0; sh.rp_s0 = s0;
          }
          __syncthreads();
          can = sh.rp_ok != 0;
        }
        if (!can) sh.bail = 1;
        else {
          uint32_t g = sh.rp_g0, j = sh.rp_j0;
          unsigned long long co = sh.rp_c0, oo = sh.rp_o0, so = sh.rp_s0;
          RepairPend pd;
          pd.n_gt = n_gt; pd.n_pick = n_pick; pd.size[0] = size[0]; pd.size[1] = size[1];
          for (int k = 0; k < 4; ++k) pd.civ[k] = (int32_t)civ[k];
          pd.rep[0] = pd.rep[1] = -1; pd.grp[0] = pd.grp[1] = -1;
          for (int al = 0; al < n_pick; ++al) {
            const int rep = sh.u_rep[pick[al]];
            pd.rep[al] = rep;
            if (!lacks[al]) continue;
            const unsigned long long bb_off = sh.r_off[sh.s_read[rep]] + sh.s_start[rep];
            const uint32_t bb = sh.s_len[rep];
            if (lane == 0) {
              RGroup G;
              G.job_first = j; G.n_members = nm[al]; G.bb_len = bb; G.out_cap = out_cap[al];
              G.bb_off = bb_off; G.out_off = oo; G.scratch_off = so;
              rp.groups[g] = G;
            }
            pd.grp[al] = (int32_t)g;
            uint32_t k = 0;
            for (int q = 0; q < nu; ++q) {
              if (!in_group(q, al)) continue;
              const int rq = sh.u_rep[q];
              if ((int)(k & 63u) == lane) {
                JobDev jd;
                jd.pat_off = bb_off; jd.pat_len = bb;
                jd.txt_off = sh.r_off[sh.s_read[rq]] + sh.s_start[rq]; jd.txt_len = sh.s_len[rq];
                jd.cigar_off = co; jd.ops_off = 0; jd.out_index = j + k; jd.pad = 0;
                rp.jobs[j + k] = jd;
              }
              co += (unsigned long long)bb + sh.s_len[rq] + 1;
              ++k;
            }
            j += nm[al]; oo += out_need[al]; so += scr_need[al]; ++g;
          }
          if (lane == 0) rp.pend[l] = pd;
          sh.bail = 2;  // the locus waits for repair_finish_kernel
        }
      }
      else {
        // ---- classification (genotype_size.rs:42-61), reference allele first (tr.rs:95-101)
        int rep[2]; uint32_t aln[2];
        for (int al = 0; al < n_pick; ++al) { rep[al] = sh.u_rep[pick[al]]; aln[al] = sh.s_len[rep[al]]; }
        int n_al = n_pick;
        if (ploidy == 2 && n_al == 1) { rep[1] = rep[0]; aln[1] = aln[0]; n_al = 2; }
        int by_hap[2] = {0, 0}, tie = 1;
        for (int i = 0; i < n; ++i) {
          int cc = 0;
          if (n_al == 2) {
            const uint32_t d1 = adiff_u(sh.s_len[i], aln[0]), d2 = adiff_u(sh.s_len[i], aln[1]);
            if (d1 < d2) cc = 0; else if (d1 > d2) cc = 1; else { tie = (tie + 1) % 2; cc = tie; }
          }
          sh.cls[i] = (int8_t)cc; by_hap[cc] += 1;
        }
        auto eq_ref = [&](int al) {
          return fits ? cmp_lds(sh.bytes + sh.s_loff[rep[al]], aln[al], sh.bytes + sh.ref_off, refn) == 0 : cmp_bytes(seg_glob(rep[al]), aln[al], a.tr_blob + refo, refn) == 0;
        };
        int order[2] = {0, 1}; int flip = 0;
        if (n_gt != 1 && !eq_ref(0) && eq_ref(1)) { order[0] = 1; order[1] = 0; flip = 1; }
        for (int oi = 0; oi < n_gt; ++oi)
          if (aln[order[oi]] > a.allele_cap[l]) sh.bail = 1;  // the host path reports the error
        sh.res_n_gt = n_gt; sh.res_flip = flip;
        for (int oi = 0; oi < n_gt; ++oi) {
          const int al = order[oi];
          sh.res_rep[oi] = rep[al]; sh.res_ci[2 * oi] = (int)civ[2 * al]; sh.res_ci[2 * oi + 1] = (int)civ[2 * al + 1]; sh.res_hap[oi] = by_hap[al];
        }
      }
    }
    __syncthreads();
  }
  // ---- outputs, by the whole wave
  const bool done = n > 0 && !sh.bail;
  for (int i = lane; i < nr; i += 64) { a.classification[r0 + i] = -1; a.read_rank[r0 + i] = -1; }
  if (n > 0 && sh.bail) { if (lane == 0) a.need_host[l] = (uint8_t)(sh.bail == 2 ? 2 : 1); return; }
  if (!done) return;
  __syncthreads();  // the -1 defaults above are ordered before the entries of the kept reads (same wave, same addresses)
  const int n_gt = sh.res_n_gt;
  for (int oi = 0; oi < n_gt; ++oi) {
    const int rep = sh.res_rep[oi];
    const uint32_t len = sh.s_len[rep];
    const uint8_t* src = fits ? sh.bytes + sh.s_loff[rep] : seg_glob(rep);
    uint8_t* dst = a.allele_blob + a.allele_off[2 * l + oi];
    for (uint32_t b = lane; b < len; b += 64) dst[b] = src[b];
    if (lane == 0) {
      a.allele_len[2 * l + oi] = len;
      a.ci[4 * l + 2 * oi] = sh.res_ci[2 * oi]; a.ci[4 * l + 2 * oi + 1] = sh.res_ci[2 * oi + 1];
      a.num_spanning[2 * l + oi] = sh.res_hap[oi];
      if (a.gt_size) a.gt_size[2 * l + oi] = (int32_t)len;  // (a pick with majority support has the genotype's size)
    }
  }
  for (int i = lane; i < n; i += 64) {
    a.classification[r0 + sh.s_read[i]] = sh.res_flip ? 1 - sh.cls[i] : sh.cls[i];
    a.read_rank[r0 + sh.s_read[i]] = i;
  }
  if (lane == 0) { a.n_alleles[l] = n_gt; a.n_spanning_reads[l] = (uint32_t)n; a.flipped[l] = (uint8_t)sh.res_flip; }
}

// ---- behind the consensus alignments and the column voting of the loci that waited for a repair: the rest of genotype_size::genotype
// (classification of the reads against the repaired alleles, genotype_size.rs:42-61), reference allele first (tr.rs:95-101), outputs.
// One wave per waiting locus (list in rp.loci).  A locus whose repaired allele does not fit (vote overflow, allele_cap) goes to the
// host path after all (need_host = 1); the others leave with need_host = 0.
struct FinishArgs { const uint8_t* vote_out; const uint32_t* vote_len; };
template <int MAXR>
struct FinShared {
  uint32_t r_s[MAXR], r_len[MAXR]; uint64_t r_off[MAXR];
  uint32_t s_read[MAXR], s_start[MAXR], s_len[MAXR];
  int8_t cls[MAXR];
  int n;
};
// equality of two byte strings in global memory, by the whole wave (uniform arguments, uniform result)
__device__ __forceinline__ bool wave_equal(const uint8_t* __restrict__ p, uint32_t n, const uint8_t* __restrict__ q, uint32_t m) {
  if (n != m) return false;
  bool diff = false;
  for (uint32_t i = threadIdx.x & 63; i < n; i += 64) diff = diff || p[i] != q[i];
  return __ballot(diff) == 0ull;
}
template <int MAXR>
__global__ void __launch_bounds__(64) repair_finish_kernel(const GtArgs a, const FinishArgs f) {
  __shared__ FinShared<MAXR> sh;
  const RepairBufs& rp = a.rp;
  if (blockIdx.x >= rp.counts[RC_LOCI]) return;
  const int64_t l = rp.loci[blockIdx.x];
  const int lane = threadIdx.x;
  const uint64_t r0 = a.locus_read_begin[l];
  const int nr = (int)(a.locus_read_begin[l + 1] - r0);
  if (lane == 0) sh.n = 0;
  gt_front<MAXR>(sh, a, r0, nr, lane);
  const int n = sh.n;
  const RepairPend pd = rp.pend[l];
  const int ploidy = a.ploidy[l] == 1 ? 1 : 2;
  // the alleles: the repaired sequence of a group, or the pick that had majority support
  const uint8_t* ap[2] = {nullptr, nullptr}; uint32_t aln[2] = {0, 0};
  bool fail = n == 0;
  for (int al = 0; al < pd.n_pick && !fail; ++al) {
    if (pd.grp[al] >= 0) {
      const uint32_t len = f.vote_len[pd.grp[al]];
      if (len == 0xFFFFFFFFu) { fail = true; break; }
      ap[al] = f.vote_out + rp.groups[pd.grp[al]].out_off; aln[al] = len;
    } else {
      const int rep = pd.rep[al];
      ap[al] = a.reads + sh.r_off[sh.s_read[rep]] + sh.s_start[rep]; aln[al] = sh.s_len[rep];
    }
  }
  int n_al = pd.n_pick;
  if (!fail && ploidy == 2 && n_al == 1) { ap[1] = ap[0]; aln[1] = aln[0]; n_al = 2; }
  int by_hap[2] = {0, 0}, order[2] = {0, 1}, flip = 0;
  if (!fail) {
    int tie = 1;  // (every lane walks the reads: uniform, and the list is short)
    for (int i = 0; i < n; ++i) {
      int cc = 0;
      if (n_al == 2) {
        const uint32_t d1 = adiff_u(sh.s_len[i], aln[0]), d2 = adiff_u(sh.s_len[i], aln[1]);
        if (d1 < d2) cc = 0; else if (d1 > d2) cc = 1; else { tie = (tie + 1) % 2; cc = tie; }
      }
      if (lane == 0) sh.cls[i] = (int8_t)cc;
      by_hap[cc] += 1;
    }
    const uint8_t* ref = a.tr_blob + a.tr_off[l]; const uint32_t refn = a.tr_len[l];
    if (pd.n_gt != 1 && !wave_equal(ap[0], aln[0], ref, refn) && wave_equal(ap[1], aln[1], ref, refn)) { order[0] = 1; order[1] = 0; flip = 1; }
    for (int oi = 0; oi < pd.n_gt; ++oi) if (aln[order[oi]] > a.allele_cap[l]) fail = true;  // the host path reports the error
  }
  __syncthreads();
  if (fail) { if (lane == 0) a.need_host[l] = 1; return; }
  for (int oi = 0; oi < pd.n_gt; ++oi) {
    const int al = order[oi];
    uint8_t* dst = a.allele_blob + a.allele_off[2 * l + oi];
    for (uint32_t b = lane; b < aln[al]; b += 64) dst[b] = ap[al][b];
    if (lane == 0) {
      a.allele_len[2 * l + oi] = aln[al];
      a.ci[4 * l + 2 * oi] = pd.civ[2 * al]; a.ci[4 * l + 2 * oi + 1] = pd.civ[2 * al + 1];
      a.num_spanning[2 * l + oi] = by_hap[al];
      if (a.gt_size) a.gt_size[2 * l + oi] = (int32_t)pd.size[al];
    }
  }
  for (int i = lane; i < n; i += 64) {
    a.classification[r0 + sh.s_read[i]] = flip ? 1 - sh.cls[i] : sh.cls[i];
    a.read_rank[r0 + sh.s_read[i]] = i;
  }
  // (need_host stays 2: the HMM batch of the loci settled by the genotyper may be reading it right now; the host turns 2 into skip_b)
  if (lane == 0) { a.n_alleles[l] = pd.n_gt; a.n_spanning_reads[l] = (uint32_t)n; a.flipped[l] = (uint8_t)flip; a.skip_b[l] = 0; if (a.finish_clears_need) a.need_host[l] = 0; }
}

}  // namespace gt
}  // namespace trgt
