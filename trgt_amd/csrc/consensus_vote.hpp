// trgt_amd/csrc/consensus_vote.hpp -- repair_consensus (src/trgt/genotype/consensus.rs:5-111) on the device, included by locus.hip
// only (inside its anonymous namespace).
//
// A consensus "group" is a backbone sequence and the members aligned against it (BiWFA, gap-affine 2,5,1: the alignments of
// trgt_wfa_batch, whose run-length CIGARs stay in HBM).  Per backbone position the members vote A / T / C / G / deleted (the LAST
// maximum wins, Iterator::max_by_key), an insertion in front of a position is taken when more than half of the members have
// one there and its most frequent string (the first of the sorted strings among equals) is carried by more members than have no
// insertion there.  One workgroup per group:
//   1. every wave walks the CIGARs of its members run by run, the lanes of a wave the positions of a run: votes are 16-bit counters
//      packed in pairs, bumped with LDS (backbones up to VOTE_LDS_POS bases) or global atomics;
//   2. per position: the winning base, and for the few positions with enough insertions the members' CIGARs are walked once more
//      for the inserted strings (lane = member), counted against each other, and the winner kept as (member, offset, length);
//   3. an exclusive scan over the emitted lengths, then the bytes.
#pragma once
#include "common.hpp"
#include "wfa_host.hpp"

namespace vote {

constexpr int VOTE_THREADS = 256;
constexpr int VOTE_LDS_POS = 4000;  // backbone positions (+ 1) whose votes fit the LDS budget below (3 words each)

struct Group {  // one backbone and its members = jobs [job_first, job_first + n_members) of the alignment batch
  uint32_t job_first, n_members, bb_len, out_cap;
  uint64_t bb_off;       // backbone bytes in the sequence blob
  uint64_t out_off;      // result bytes
  uint64_t scratch_off;  // in 32-bit words: 3 (bb_len + 1) vote words (when they do not fit LDS) | 3 n_members words of insert records
};
struct VoteArgs {
  const Group* groups; uint32_t n_groups; const uint32_t* n_groups_dev;  // n_groups_dev != nullptr: the count is read there (the grid is an upper bound)
  const uint8_t* seqs; const trgt::JobDev* jobs; const uint32_t* cigar; const uint32_t* cigar_len;
  uint32_t* scratch; uint8_t* out; uint32_t* out_len;  // out_len: 0xFFFFFFFF when the result does not fit out_cap
};

__device__ __forceinline__ int base_index(uint8_t b) { return b == 'A' ? 0 : b == 'T' ? 1 : b == 'C' ? 2 : 3; }  // consensus.rs:74-92: anything else votes G

// memcmp-then-length order of two byte strings (std::string / Rust String ordering)
__device__ __forceinline__ int cmp_str(const uint8_t* a, uint32_t na, const uint8_t* b, uint32_t nb) {
  const uint32_t m = na < nb ? na : nb;
  for (uint32_t i = 0; i < m; ++i) if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
  return na < nb ? -1 : (na > nb ? 1 : 0);
}

__global__ void __launch_bounds__(VOTE_THREADS) consensus_vote_kernel(const VoteArgs a) {
  __shared__ uint32_t l_votes[3 * (VOTE_LDS_POS + 1)];
  __shared__ uint32_t l_scan[VOTE_THREADS / 64 + 1];
  __shared__ uint32_t l_carry, l_top_member, l_top_count, l_k;
  const uint32_t g = blockIdx.x;
  if (g >= (a.n_groups_dev ? *a.n_groups_dev : a.n_groups)) return;
  const Group grp = a.groups[g];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t L = grp.bb_len, n = grp.n_members;
  const bool in_lds = L + 1 <= (uint32_t)VOTE_LDS_POS + 1;
  uint32_t* const votes = in_lds ? l_votes : a.scratch + grp.scratch_off;             // [3 (L + 1)]: A|T, C|G, deleted|insertions
  uint32_t* const ins_rec = a.scratch + grp.scratch_off + (in_lds ? 0 : 3 * ((size_t)L + 1));  // [3 n]: x, length, count of equals
  for (uint32_t i = (uint32_t)tid; i < 3 * (L + 1); i += VOTE_THREADS) votes[i] = 0;
  __syncthreads();
  // ---- 1. votes
  for (uint32_t m = (uint32_t)wave; m < n; m += VOTE_THREADS / 64) {
    const trgt::JobDev job = a.jobs[grp.job_first + m];
    const uint32_t* cg = a.cigar + job.cigar_off;
    const uint32_t nc = a.cigar_len[job.out_index];  // 0: the alignment failed -- the member counts, its votes do not (consensus.rs:28-31)
    const uint8_t* sq = a.seqs + job.txt_off;
    uint32_t x = 0, y = 0;
    for (uint32_t r = 0; r < nc; ++r) {
      const uint32_t e = cg[r], len = e >> 4, code = e & 0xFu;
      if (code == 7 || code == 8 || code == 0) {
        for (uint32_t i = (uint32_t)lane; i < len; i += 64) {
          if (y + i >= L || x + i >= job.txt_len) break;
          const int bi = base_index(sq[x + i]);
          atomicAdd(&votes[3 * (y + i) + (bi >> 1)], 1u << (16 * (bi & 1)));
        }
        x += len; y += len;
      } else if (code == 2) {
        for (uint32_t i = (uint32_t)lane; i < len; i += 64) { if (y + i >= L) break; atomicAdd(&votes[3 * (y + i) + 2], 1u); }
        y += len;
      } else if (code == 1) {
        if (lane == 0 && y <= L) atomicAdd(&votes[3 * y + 2], 1u << 16);
        x += len;
      }
    }
  }
  __syncthreads();
  // ---- 2. per position: winning base (word 0), insertion taken in front of it (word 1: member, word 2: length; 0 = none)
  for (uint32_t p0 = 0; p0 < L; p0 += VOTE_THREADS) {
    const uint32_t p = p0 + (uint32_t)tid;
    uint32_t n_ins = 0;
    if (p < L) {
      const uint32_t w0 = votes[3 * p], w1 = votes[3 * p + 1], w2 = votes[3 * p + 2];
      const uint32_t v[5] = {w0 & 0xFFFFu, w0 >> 16, w1 & 0xFFFFu, w1 >> 16, w2 & 0xFFFFu};
      int best = 0;
      for (int i = 1; i < 5; ++i) if (v[i] >= v[best]) best = i;  // max_by_key: the last maximum
      n_ins = w2 >> 16;
      votes[3 * p] = (uint32_t)best; votes[3 * p + 1] = 0; votes[3 * p + 2] = 0;
    }
    const bool cand = p < L && n_ins > n / 2;
    // the candidates of this block of positions, one after the other (uniform loop over the waves' ballots)
    for (int w = 0; w < VOTE_THREADS / 64; ++w) {
      __syncthreads();
      if (wave == w) { const unsigned long long bm = __ballot(cand); if (lane == 0) { l_scan[0] = (uint32_t)bm; l_scan[1] = (uint32_t)(bm >> 32); } }
      __syncthreads();
      unsigned long long todo = ((unsigned long long)l_scan[1] << 32) | l_scan[0];
      while (todo) {
        const int cl = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const uint32_t cp = p0 + 64u * (uint32_t)w + (uint32_t)cl;  // the position
        // inserted strings at cp: member m's is query[x .. x + len) of the insertion run met at backbone offset cp
        for (uint32_t m = (uint32_t)tid; m < n; m += VOTE_THREADS) {
          const trgt::JobDev job = a.jobs[grp.job_first + m];
          const uint32_t* cg = a.cigar + job.cigar_off;
          const uint32_t nc = a.cigar_len[job.out_index];
          uint32_t x = 0, y = 0, fx = 0, fl = 0;
          for (uint32_t r = 0; r < nc && y <= cp; ++r) {
            const uint32_t e = cg[r], len = e >> 4, code = e & 0xFu;
            if (code == 7 || code == 8 || code == 0) { x += len; y += len; }
            else if (code == 2) y += len;
            else if (code == 1) { if (y == cp) { fx = x; fl = len; } x += len; }
          }
          ins_rec[3 * m] = fx; ins_rec[3 * m + 1] = fl; ins_rec[3 * m + 2] = 0;
        }
        __syncthreads();
        for (uint32_t m = (uint32_t)tid; m < n; m += VOTE_THREADS) {  // how many members insert the same string
          const uint32_t fl = ins_rec[3 * m + 1];
          if (!fl) continue;
          const uint8_t* sm = a.seqs + a.jobs[grp.job_first + m].txt_off + ins_rec[3 * m];
          uint32_t eq = 0;
          for (uint32_t j = 0; j < n; ++j) {
            const uint32_t fj = ins_rec[3 * j + 1];
            if (fj == fl && cmp_str(sm, fl, a.seqs + a.jobs[grp.job_first + j].txt_off + ins_rec[3 * j], fj) == 0) ++eq;
          }
          ins_rec[3 * m + 2] = eq;
        }
        __syncthreads();
        if (tid == 0) {  // the most frequent string, the smallest one among equals (first of the sorted strings); members without one
          uint32_t k = 0, top = 0xFFFFFFFFu, top_count = 0;
          for (uint32_t m = 0; m < n; ++m) {
            const uint32_t fl = ins_rec[3 * m + 1], eq = ins_rec[3 * m + 2];
            if (!fl) continue;
            ++k;
            bool better = eq > top_count;
            if (!better && eq == top_count && top != 0xFFFFFFFFu)
              better = cmp_str(a.seqs + a.jobs[grp.job_first + m].txt_off + ins_rec[3 * m], fl,
                               a.seqs + a.jobs[grp.job_first + top].txt_off + ins_rec[3 * top], ins_rec[3 * top + 1]) < 0;
            if (better) { top = m; top_count = eq; }
          }
          l_top_member = top; l_top_count = top_count; l_k = k;
        }
        __syncthreads();
        if (tid == 0 && l_top_member != 0xFFFFFFFFu && l_top_count > n - l_k) {  // more members carry it than have no insertion here
          votes[3 * cp + 1] = l_top_member + 1;
          votes[3 * cp + 2] = ins_rec[3 * l_top_member + 1];
        }
        __syncthreads();
      }
    }
  }
  __syncthreads();
  // ---- 3. exclusive scan of the emitted lengths, then the bytes
  if (tid == 0) l_carry = 0;
  __syncthreads();
  uint8_t* const out = a.out + grp.out_off;
  for (uint32_t p0 = 0; p0 < L; p0 += VOTE_THREADS) {
    const uint32_t p = p0 + (uint32_t)tid;
    uint32_t emit = 0, best = 4, im = 0, il = 0;
    if (p < L) { best = votes[3 * p]; im = votes[3 * p + 1]; il = im ? votes[3 * p + 2] : 0; emit = il + (best != 4 ? 1u : 0u); }
    // inclusive scan inside the wave, then across the waves
    uint32_t inc = emit;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) l_scan[wave] = inc;
    __syncthreads();
    uint32_t before = l_carry;
    for (int w = 0; w < wave; ++w) before += l_scan[w];
    const uint32_t at = before + inc - emit;
    if (p < L && at + emit <= grp.out_cap) {
      uint32_t o = at;
      if (il) {
        // the inserted string: walked again for its offset (only the winner's length was kept)
        const trgt::JobDev job = a.jobs[grp.job_first + (im - 1)];
        const uint32_t* cg = a.cigar + job.cigar_off;
        const uint32_t nc = a.cigar_len[job.out_index];
        uint32_t x = 0, y = 0, fx = 0;
        for (uint32_t r = 0; r < nc && y <= p; ++r) {
          const uint32_t e = cg[r], len = e >> 4, code = e & 0xFu;
          if (code == 7 || code == 8 || code == 0) { x += len; y += len; }
          else if (code == 2) y += len;
          else if (code == 1) { if (y == p) fx = x; x += len; }
        }
        const uint8_t* s = a.seqs + job.txt_off + fx;
        for (uint32_t i = 0; i < il; ++i) out[o++] = s[i];
      }
      if (best != 4) out[o] = (uint8_t)("ATCG"[best]);
    }
    __syncthreads();
    if (tid == VOTE_THREADS - 1) l_carry = before + inc;
    __syncthreads();
  }
  if (tid == 0) a.out_len[g] = l_carry <= grp.out_cap ? l_carry : 0xFFFFFFFFu;
}

}  // namespace vote
