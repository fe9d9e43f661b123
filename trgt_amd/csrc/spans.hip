// trgt_amd/csrc/spans.hip -- trgt_find_spans_batch: locate the repeat inside every read.
//
// Replaces find_spans / find_tr_spans of PacificBiosciences/trgt v3.0.0
// (src/trgt/genotype/span_locater.rs:7-68): per read and per flank piece, the leftmost exact
// occurrence (windows().position(), :10-12), else a gap-affine ends-free wavefront alignment of the
// piece against the whole read (THREAD_WFA_FLANK, src/commands/genotype.rs:66-80; :14-26) accepted
// when count_matches() >= flank_len * min_flank_id_frac; the two flank hits are combined into the
// repeat span (:52-66).
//
// Launches on the ctx stream, no host round trip in between:
//   1. flank_scan_wide_kernel  one wavefront per read, both pieces at once: 1024 read bytes per round, 16 candidate
//                          windows per lane from one 24-byte load, 8-byte filter, cooperative verify in increasing
//                          position -> leftmost hit.  Misses are appended to a device-side job list (atomic counter).
//                          (flank_scan_kernel, one wavefront per (read, side), remains for flank_len < 4.)
//   2. wfa_fast_kernel     (wfa_fast.hpp) persistent workgroups drain that list: first the alignments of the reads too short to
//                          span their locus; then (flank_window_kernel, window_check_kernel) the other alignments on a seeded
//                          window of the read where one can be proven sufficient; then the rest against the whole read.
//   3. span_combine_kernel per read: threshold test and (lf.end, rf.start) combination.
#include <algorithm>
#include <cmath>

#include "wfa_host.hpp"

namespace trgt {

// The counter block of a find_spans call (16 words in HBM, cleared in front of the scan): list lengths and tallies shared by its kernels.
enum SpanCount : int {
  SC_HEAVY = 0,     // expensive fallback alignments: the front of the two-ended job list
  SC_LONG = 1,      // reads beyond the dedicated kernels' texts (second list)
  SC_LIGHT = 2,     // the other fallback alignments: the back of the two-ended list
  SC_ZERO = 3,      // stays 0 (the "front" count of a launch that takes the back of the list only)
  SC_WIN = 4,       // alignments on a seeded window
  SC_REST = 5,      // alignments against the whole read (no window, or a window that did not stand)
  SC_KEEP = 6,      // what the pre-filter keeps for the back-tracing launch
  SC_SHORTCUT = 7,  // alignments settled by the substitution / one-base-gap shortcuts
  SC_NOSEED = 8,    // expensive alignments without seeds inside the read: the pre-filter's list
  SC_GAPS = 9,      // of SC_SHORTCUT: one-base gaps
  SC_BAND = 10,     // of SC_KEEP: back-traced inside the band the pre-filter's penalty and end diagonal allow
  SC_HREST = 11,    // of SC_KEEP: the others (and the banded ones that did not stand), back-traced over the whole read
  SC_BANDFAIL = 12, // banded runs that did not come out with the pre-filter's penalty (the argument says: none)
  SC_LBAND = 13,    // long reads (SC_LONG) whose windows named the penalty and the end diagonal: back-traced inside a band
  SC_LREST = 14,    // ... the other long reads the window filter keeps: back-traced over the whole read
  SC_LNOSEED = 15,  // long reads' alignments the seed search could neither settle nor window (or whose window did not stand): the window filter's list
  SC_WORDS = 16
};

struct ScanArgs {
  const uint8_t* flank_blob; const uint8_t* read_blob;
  const uint64_t* piece_off;   // [2 * n_loci] left piece, right piece (offsets into flank_blob)
  const uint64_t* read_off; const uint32_t* read_len; const uint32_t* read_locus;
  uint64_t n_jobs;             // 2 * n_reads
  int32_t flank_len;
  int32_t* pos;                // [n_jobs] leftmost exact start or -1
  int32_t* n_match;            // [n_jobs] initialised to -1 here (the alignment kernels fill in the jobs they run)
  JobDev* wfa_jobs; uint32_t* wfa_count;  // fallback alignments: two-ended job list.  Reads shorter than heavy_len[locus] cannot hold
                                          // both flanks: their alignments run to high scores and cost 10-100x the others, so they
                                          // are appended from the front (wfa_count[0]) and drained first; the rest from the back
                                          // (wfa_jobs[jobs_cap - 1 - k], wfa_count[2]).  Longest-first keeps the tail of the
                                          // persistent alignment kernel short (12.45 vs 13.0 ms on the 10k-locus batch).
  const uint32_t* heavy_len; uint32_t jobs_cap;
  JobDev* wfa_jobs_long; uint32_t long_tlen;  // reads longer than long_tlen go to a second list (wfa_count[1]): they would not fit the
                                              // LDS budget of the dedicated kernel and must not drag the whole batch onto the generic one
};

__device__ __forceinline__ uint32_t load_u32(const uint8_t* p) {
  uint32_t v;
  __builtin_memcpy(&v, p, 4);
  return v;
}

__global__ void __launch_bounds__(256) flank_scan_kernel(const ScanArgs a) {
  const uint64_t j = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= a.n_jobs) return;  // whole wave exits together
  const int lane = threadIdx.x & 63;
  const uint64_t r = j >> 1;
  const int side = (int)(j & 1);
  const int F = a.flank_len, n = (int)a.read_len[r];
  const uint8_t* __restrict__ piece = a.flank_blob + a.piece_off[2 * (uint64_t)a.read_locus[r] + side];
  const uint8_t* __restrict__ read = a.read_blob + a.read_off[r];
  int found = -1;
  if (n >= F) {
    const int last = n - F;  // last candidate start
    const uint32_t head = F >= 4 ? load_u32(piece) : 0;
    for (int base = 0; base <= last && found < 0; base += 64) {
      const int p = base + lane;
      bool hit = false;
      if (p <= last) {
        hit = F < 4 || load_u32(read + p) == head;
        if (hit) {
          int i = F >= 4 ? 4 : 0;
          for (; i + 4 <= F; i += 4)
            if (load_u32(read + p + i) != load_u32(piece + i)) { hit = false; break; }
          if (hit)
            for (; i < F; ++i)
              if (read[p + i] != piece[i]) { hit = false; break; }
        }
      }
      const unsigned long long m = __ballot(hit);
      if (m) found = base + (__ffsll((long long)m) - 1);
    }
  }
  if (lane == 0) {
    a.pos[j] = found; a.n_match[j] = -1;
    if (found < 0) {  // fall back to the wavefront aligner (span_locater.rs:13-26)
      const bool lng = (uint32_t)n > a.long_tlen;
      const uint32_t slot = atomicAdd(a.wfa_count + (lng ? SC_LONG : SC_HEAVY), 1u);
      JobDev jd;
      jd.pat_off = a.piece_off[2 * (uint64_t)a.read_locus[r] + side]; jd.txt_off = a.read_off[r];
      jd.cigar_off = 0; jd.ops_off = 0; jd.pat_len = (uint32_t)F; jd.txt_len = (uint32_t)n; jd.out_index = (uint32_t)j; jd.pad = 0;
      (lng ? a.wfa_jobs_long : a.wfa_jobs)[slot] = jd;
    }
  }
}

// ---- Seeded windows for the fallback alignments (an implementation shortcut with identical results, checked after the fact) ----
// The ends-free alignment of a flank piece against a whole read keeps every diagonal of the read alive at every score level,
// although the alignment it ends up with sits on a handful of diagonals.  When the optimal penalty s* is small this is provable
// up front.  Cut the piece into win_m segments of win_q bases.  With the conditions checked by window_plan() every mismatch or gap
// spoils at most as many segments as its penalty / x, so an alignment of penalty <= S0 = x * win_m - 1 leaves at least one segment
// matched exactly and gap-free: an exact occurrence of that segment at read position t pins the alignment to the diagonal
// k = t - (segment offset), and its path strays from k by at most G = (S0 - o) / e diagonals (its total gap length).  A wavefront
// cell (s, k) depends only on cells (s', k') with |k' - k| <= (s - s') / e.  So the alignment of the piece against the read window
// [kmin - G - C, kmax + G + C + F), C = S0 / e, kmin / kmax over ALL exact segment occurrences in the read, computes every cell the
// full run's termination test and back-trace would read -- bit for bit -- PROVIDED its penalty turns out <= S0 (window_check_kernel;
// otherwise, or when the occurrences are spread too widely or there are none, the whole read is aligned as before).  Offsets of a
// windowed run never exceed those of the full run (fewer sources under the max), so it cannot end early either.
// One wavefront; every lane returns the same kmin / kmax (kmin > kmax: no occurrence).
constexpr int WIN_SEGMENTS_DEFAULT = 8;  // (4, 6 and 8 are instantiated; TRGT_WIN_SEGMENTS picks another one.  Measured on the 10k-locus
                                         // batch, search + alignments: 4 -> 3.19 ms, 6 -> 3.00, 8 -> 2.90, 10 -> 2.97, 12 -> 3.02)
// An "occurrence" is a match of the first TWELVE bases of a segment (all from registers: no memory round trip per candidate).  That
// is a superset of the exact occurrences, which is all the argument needs -- a chance match (4^-12 per position) can only widen
// the window or make the spread test fail.
template <int WIN_SEGMENTS>
__device__ __forceinline__ void piece_window_heads(const uint8_t* __restrict__ read, int n, const uint64_t (&h)[WIN_SEGMENTS], const uint32_t (&h3)[WIN_SEGMENTS], int q, int lane,
                                                   int& kmin_out, int& kmax_out);
template <int WIN_SEGMENTS>
__device__ __forceinline__ void piece_window(const uint8_t* __restrict__ read, int n, const uint8_t* __restrict__ piece, int q, int lane,
                                             int& kmin_out, int& kmax_out) {
  uint64_t h[WIN_SEGMENTS]; uint32_t h3[WIN_SEGMENTS];
#pragma unroll
  for (int i = 0; i < WIN_SEGMENTS; ++i) {  // (all loads in flight together)
    const uint8_t* __restrict__ seg = piece + i * q;
    h[i] = (uint64_t)load_u32(seg + 4) << 32 | load_u32(seg);
    h3[i] = load_u32(seg + 8);
  }
  piece_window_heads<WIN_SEGMENTS>(read, n, h, h3, q, lane, kmin_out, kmax_out);
}
// ... with the twelve head bytes of every segment already in registers (flank_window_kernel fetches them for all jobs of a wave up front)
template <int WIN_SEGMENTS>
__device__ __forceinline__ void piece_window_heads(const uint8_t* __restrict__ read, int n, const uint64_t (&h)[WIN_SEGMENTS], const uint32_t (&h3)[WIN_SEGMENTS], int q, int lane,
                                                   int& kmin_out, int& kmax_out) {
  int kmin = 0x7FFFFFFF, kmax = -0x7FFFFFFF;
  const int last = n - 12;  // last start of a twelve-base match
  for (int base = 0; base <= last; base += 1024) {
    const int off = base + 16 * lane;
    uint32_t w[7] = {0, 0, 0, 0, 0, 0, 0};
    if (off + 28 <= n) {
      uint4 v; uint2 v2; uint32_t v3;
      __builtin_memcpy(&v, read + off, 16); __builtin_memcpy(&v2, read + off + 16, 8); __builtin_memcpy(&v3, read + off + 24, 4);
      w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; w[4] = v2.x; w[5] = v2.y; w[6] = v3;
    } else if (off < n) {
      // the lane that crosses the end of the read (the whole wave walks this branch with it): whole dwords while they fit, then the last
      // one to three bytes out of the dword that ENDS with the read -- seven independent loads and a shift (byte by byte it was 28
      // dependent iterations, a fifth of the kernel's instructions)
      const int rem = n - off, full = rem >> 2, tail = rem & 3;
#pragma unroll
      for (int k = 0; k < 7; ++k)
        if (k < full) w[k] = load_u32(read + off + 4 * k);
      if (tail && full < 7) {
        const uint32_t t = n >= 4 ? load_u32(read + n - 4) >> (8 * (4 - tail)) : 0u;  // bytes n - tail .. n - 1, zero above
        if (n >= 4) {
#pragma unroll
          for (int k = 0; k < 7; ++k) if (k == full) w[k] = t;
        } else {
          for (int b = 4 * full; b < rem; ++b) w[b >> 2] |= (uint32_t)read[off + b] << (8 * (b & 3));
        }
      }
    }
    // The first FOUR bytes at every candidate start decide whether a segment is looked at more closely in this round (a 32-bit
    // compare per position and segment; the other eight bytes are formed only then).  This kernel is bound by instruction issue:
    // twelve bytes at every position and 64-bit compares were 48 byte-aligns and 128 double-width compares per round.
    uint32_t lo[16];
#pragma unroll
    for (int s16 = 0; s16 < 16; ++s16) {
      const int d = s16 >> 2, sh = s16 & 3;
      lo[s16] = sh == 0 ? w[d] : __builtin_amdgcn_alignbyte(w[d + 1], w[d], sh);
    }
#pragma unroll
    for (int i = 0; i < WIN_SEGMENTS; ++i) {
      const uint32_t h_lo = (uint32_t)h[i], h_hi = (uint32_t)(h[i] >> 32);
      bool any = false;
#pragma unroll
      for (int s16 = 0; s16 < 16; ++s16) any |= lo[s16] == h_lo;
      if (__ballot(any) == 0ull) continue;  // (uniform; taken but for the round that holds an occurrence of this segment, or four of its bases by chance)
#pragma unroll
      for (int s16 = 0; s16 < 16; ++s16) {
        if (lo[s16] == h_lo) {
          const int d = s16 >> 2, sh = s16 & 3;
          const uint32_t hi = sh == 0 ? w[d + 1] : __builtin_amdgcn_alignbyte(w[d + 2], w[d + 1], sh);
          const uint32_t w3 = sh == 0 ? w[d + 2] : __builtin_amdgcn_alignbyte(w[d + 3], w[d + 2], sh);
          if (hi == h_hi && w3 == h3[i] && off + s16 <= last) {
            const int k = off + s16 - i * q;
            kmin = k < kmin ? k : kmin; kmax = k > kmax ? k : kmax;
          }
        }
      }
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const int a = __shfl_xor(kmin, d), b = __shfl_xor(kmax, d);
    kmin = a < kmin ? a : kmin; kmax = b > kmax ? b : kmax;
  }
  kmin_out = kmin; kmax_out = kmax;
}

// One wavefront per fallback alignment of the light list (the back of wfa_jobs): jobs that get a window go to win_jobs (count[4],
// JobDev::pad = first text base of the window), the others to rest_jobs (count[5]).  A workgroup takes WIN_JOBS_PER_WG jobs and
// reserves its output slots with one atomic per list.
struct WindowArgs {
  const uint8_t* flank_blob; const uint8_t* read_blob;
  const JobDev* wfa_jobs; uint32_t jobs_cap; uint32_t* count;
  JobDev* win_jobs; JobDev* rest_jobs;
  int32_t flank_len, q, margin, spread, tbf;
  int32_t front;                       // 1: the jobs at the FRONT of wfa_jobs (count[0], the expensive ones); the jobs without a window then go to rest_jobs under count[8]
  const JobDev* long_jobs; JobDev* long_rest;  // not NULL (with front == 0): the long reads' list (count[SC_LONG]) behind the light one; its jobs without a window go to long_rest under count[SC_LNOSEED]
  int32_t hamming_max;                 // > 0: the substitution-only shortcut below, for up to this many mismatches
  int32_t indel_ok;                    // 1: the one-base-gap shortcut (penalties 2,5,1)
  int32_t* n_match; uint32_t* span4;   // per (read, side): what the alignment kernels would have written for such a job
};
constexpr int WIN_JOBS_PER_WG = 64;
template <int WIN_SEGMENTS>
__global__ void __launch_bounds__(256) flank_window_kernel(const WindowArgs a) {
  __shared__ JobDev l_out[WIN_JOBS_PER_WG];  // windowed jobs from the front, the others from the back
  __shared__ JobDev l_long[WIN_JOBS_PER_WG]; // long reads' jobs without a window
  __shared__ uint32_t l_nw, l_nr, l_bw, l_br, l_ns, l_ni, l_nl, l_bl;
  // (the length of the list this launch walks; with long_jobs the long reads' list behind it -- ONE launch: a second one would wait behind the
  //  pre-filter's persistent workgroups of the other stream, 1.6 ms for 9 k jobs on the catalog mix)
  const uint32_t n_first = a.count[a.front ? SC_HEAVY : SC_LIGHT], n_light = n_first + (a.long_jobs ? a.count[SC_LONG] : 0u);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // jobs per workgroup and round: 64, fewer when the list is short (a job is a dependent chain of loads: 17 k jobs in rounds of 64 kept
  // 270 workgroups busy for 0.39 ms; spread over all of them they take a fifth of that)
  const uint32_t per_wg = min((uint32_t)WIN_JOBS_PER_WG, max(4u, ((n_light + gridDim.x - 1u) / gridDim.x + 3u) & ~3u));
  for (uint32_t c0 = blockIdx.x * per_wg; c0 < n_light; c0 += gridDim.x * per_wg) {
    if (threadIdx.x == 0) { l_nw = 0; l_nr = 0; l_ns = 0; l_ni = 0; l_nl = 0; }
    __syncthreads();
    const uint32_t c1 = c0 + per_wg < n_light ? c0 + per_wg : n_light;
    // (a job is a chain of dependent loads -- the job, the heads of its piece's segments, the read -- and a wave walks up to sixteen of
    //  them: lane j fetches the first two links for the wave's j-th job up front, as the scan does, round 5)
    uint64_t pf_pat = 0, pf_txt = 0, pf_cig = 0, pf_ops = 0;
    uint32_t pf_plen = 0, pf_tlen = 0, pf_oi = 0, pf_pad = 0, pf_hlo[WIN_SEGMENTS], pf_hhi[WIN_SEGMENTS], pf_h3[WIN_SEGMENTS];
#pragma unroll
    for (int sg = 0; sg < WIN_SEGMENTS; ++sg) { pf_hlo[sg] = 0; pf_hhi[sg] = 0; pf_h3[sg] = 0; }
    {
      const uint32_t ij = c0 + (uint32_t)wave + 4u * (uint32_t)lane;
      if (lane < WIN_JOBS_PER_WG / 4 && ij < c1) {
        const JobDev pj = ij >= n_first ? a.long_jobs[ij - n_first] : a.front ? a.wfa_jobs[ij] : a.wfa_jobs[a.jobs_cap - 1u - ij];
        pf_pat = pj.pat_off; pf_txt = pj.txt_off; pf_cig = pj.cigar_off; pf_ops = pj.ops_off; pf_plen = pj.pat_len; pf_tlen = pj.txt_len; pf_oi = pj.out_index; pf_pad = pj.pad;
        if (pj.txt_len >= 12u) {
#pragma unroll
          for (int sg = 0; sg < WIN_SEGMENTS; ++sg) {
            const uint8_t* __restrict__ seg = a.flank_blob + pj.pat_off + sg * a.q;
            pf_hlo[sg] = load_u32(seg); pf_hhi[sg] = load_u32(seg + 4); pf_h3[sg] = load_u32(seg + 8);
          }
        }
      }
    }
    auto lane32 = [](uint32_t v, int j) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)v, j); };
    auto lane64 = [&](uint64_t v, int j) -> uint64_t { return (uint64_t)lane32((uint32_t)v, j) | ((uint64_t)lane32((uint32_t)(v >> 32), j) << 32); };
    int slot = 0;
    for (uint32_t i = c0 + (uint32_t)wave; i < c1; i += 4, ++slot) {
      const bool is_long = i >= n_first;
      JobDev jd;
      jd.pat_off = lane64(pf_pat, slot); jd.txt_off = lane64(pf_txt, slot); jd.cigar_off = lane64(pf_cig, slot); jd.ops_off = lane64(pf_ops, slot);
      jd.pat_len = lane32(pf_plen, slot); jd.txt_len = lane32(pf_tlen, slot); jd.out_index = lane32(pf_oi, slot); jd.pad = lane32(pf_pad, slot);
      const int n = (int)jd.txt_len, F = a.flank_len;
      int kmin = 1, kmax = 0;
      if (n >= 12) {
        uint64_t h[WIN_SEGMENTS]; uint32_t h3[WIN_SEGMENTS];
#pragma unroll
        for (int sg = 0; sg < WIN_SEGMENTS; ++sg) { h[sg] = (uint64_t)lane32(pf_hlo[sg], slot) | ((uint64_t)lane32(pf_hhi[sg], slot) << 32); h3[sg] = lane32(pf_h3[sg], slot); }
        piece_window_heads<WIN_SEGMENTS>(a.read_blob + jd.txt_off, n, h, h3, a.q, lane, kmin, kmax);
      }
      // ---- The alignment of a piece that differs from the read by one or two substitutions, without aligning.  All seeds on ONE
      //      diagonal k with the piece inside the read there, d <= hamming_max = min(segments - 1, (o + e - 1) / x, 4) mismatches on it:
      //      every alignment of penalty <= x d < o + e is gap-free, i.e. a diagonal k' with at most d mismatches; those spoil at most
      //      d segments, the others occur exactly on k' and their heads are among the seeds, so k' = k.  The optimal alignment is
      //      therefore unique -- diagonal k, penalty x d -- and what the reference reads off it (span_locater.rs:14-26) is
      //      count_matches() = F - d and the text span [k, k + F).  (d = 0 cannot happen: the exact scan would have found it.)
      bool solved = false;
      if (a.hamming_max > 0 && kmin == kmax && kmin >= 0 && kmin + F <= n) {
        const uint8_t* __restrict__ t = a.read_blob + jd.txt_off + kmin;
        const uint8_t* __restrict__ pz = a.flank_blob + jd.pat_off;
        auto differing_bytes = [](uint32_t x) -> uint32_t { return (uint32_t)__builtin_popcount((x | ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu)) & 0x80808080u); };
        uint32_t cnt = 0;
        for (int i = lane; i < (F >> 2); i += 64) cnt += differing_bytes(load_u32(t + 4 * i) ^ load_u32(pz + 4 * i));
        if (lane == 63 && (F & 3)) cnt += differing_bytes((load_u32(t + F - 4) ^ load_u32(pz + F - 4)) >> (8 * (4 - (F & 3))));  // the last one to three bytes
        // (the sum over the lanes from four ballots: a shuffle reduction is six LDS round trips per job, as much as the search itself)
        const unsigned long long b1 = __ballot(cnt >= 1u);
        uint32_t total = (uint32_t)__popcll(b1);
        if (total <= (uint32_t)a.hamming_max) total += (uint32_t)(__popcll(__ballot(cnt >= 2u)) + __popcll(__ballot(cnt >= 3u)) + __popcll(__ballot(cnt >= 4u)) + __popcll(__ballot(cnt >= 5u)));
        cnt = total;  // (uniform; lanes hold at most one dword and the tail for flanks up to 255 bases, more only beyond -- then cnt >= 5 ends the shortcut)
        if (cnt >= 1u && cnt <= (uint32_t)a.hamming_max) {
          solved = true;
          if (lane == 0) {
            const uint64_t j = jd.out_index;
            a.n_match[j] = F - (int)cnt;
            a.span4[4 * j] = 0u; a.span4[4 * j + 1] = (uint32_t)F; a.span4[4 * j + 2] = (uint32_t)kmin; a.span4[4 * j + 3] = (uint32_t)(kmin + F);
            atomicAdd(&l_ns, 1u);  // (one global atomic per workgroup below: 60k of them on one address doubled this kernel's time)
          }
        }
      }
      // ---- ... and of a piece that differs from the read by ONE inserted or deleted base (penalties 2,5,1 only: o + e = 6).  All seeds on
      //      two neighbouring diagonals, both inside the read.  An alignment of penalty <= 6 is gap-free with <= 3 mismatches, or has one
      //      gap of one base and no mismatch.  The former leaves five segments exact, so it lies on a seeded diagonal: excluded when both
      //      have >= 4 mismatches -- and with 2 and 4 excluded as well, the optimal penalty is 6.  The latter leaves seven segments exact;
      //      the part on either side of the gap that holds a whole segment head lies on a seeded diagonal.  Both parts seeded: prefix on
      //      one, suffix on the other of the two diagonals -- feasible iff the exact prefix on the first reaches the exact suffix on the
      //      second (lcp / sfx below), and required to be feasible in ONE order only.  One part without a head (fewer than 12 bases in
      //      front of the gap, or a suffix that starts beyond the last head): then the other part is exact from within 13 bases of the
      //      start, or up to the last head, on a seeded diagonal -- excluded by the two margin tests.  So every optimal alignment runs
      //      along the first diagonal, takes the one-base gap somewhere in the overlap of prefix and suffix (a homopolymer leaves the
      //      place open, nothing else) and ends on the second: count_matches() and the span are the same for all of them --
      //      F matches over F + 1 bases for an inserted base, F - 1 over F - 1 for a deleted one (span_locater.rs:14-26).
      if (!solved && a.indel_ok && kmax - kmin == 1 && kmin >= 0 && kmax + F <= n && F <= 255) {
        const uint8_t* __restrict__ t = a.read_blob + jd.txt_off + kmin;
        const uint8_t* __restrict__ pz = a.flank_blob + jd.pat_off;
        const int nd = F >> 2, rem = F & 3;
        uint32_t pw = 0, t0 = 0, t1 = 0;  // lane i: bytes 4 i .. 4 i + 3 of the piece and of the read on the two diagonals; lane nd: the last one to three
        if (lane < nd) { pw = load_u32(pz + 4 * lane); t0 = load_u32(t + 4 * lane); t1 = load_u32(t + 1 + 4 * lane); }
        else if (lane == nd && rem) { const int sh = 8 * (4 - rem); pw = load_u32(pz + F - 4) >> sh; t0 = load_u32(t + F - 4) >> sh; t1 = load_u32(t + 1 + F - 4) >> sh; }
        auto byte_flags = [](uint32_t x) -> uint32_t { return (x | ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu)) & 0x80808080u; };  // bit 8 b + 7: byte b differs
        const uint32_t m0 = byte_flags(pw ^ t0), m1 = byte_flags(pw ^ t1);
        const unsigned long long b0 = __ballot(m0 != 0u), b1 = __ballot(m1 != 0u);
        auto lcp = [&](unsigned long long b, uint32_t m) -> int {  // bases of the piece that match from its start
          if (!b) return F;
          const int l = (int)__builtin_ctzll(b);
          return 4 * l + (__builtin_ctz((uint32_t)__builtin_amdgcn_readlane((int)m, l)) >> 3);
        };
        auto sfx = [&](unsigned long long b, uint32_t m) -> int {  // first base of the piece's exact suffix
          if (!b) return 0;
          const int l = 63 - (int)__builtin_clzll(b);
          return 4 * l + ((31 - __builtin_clz((uint32_t)__builtin_amdgcn_readlane((int)m, l))) >> 3) + 1;
        };
        auto at_least_4 = [&](unsigned long long b, uint32_t m) -> bool {  // mismatches on the diagonal
          const int lanes = __popcll(b);
          if (lanes >= 4) return true;
          const uint32_t c = (uint32_t)__builtin_popcount(m);
          return lanes + __popcll(__ballot(c >= 2u)) + __popcll(__ballot(c >= 3u)) + __popcll(__ballot(c >= 4u)) >= 4;
        };
        const int lcp0 = lcp(b0, m0), lcp1 = lcp(b1, m1), sfx0 = sfx(b0, m0), sfx1 = sfx(b1, m1);
        const int last_head = (WIN_SEGMENTS - 1) * a.q;
        const bool margins = sfx0 > 13 && sfx1 > 13 && lcp0 < last_head - 1 && lcp1 < last_head - 1;
        const bool ins_ok = sfx1 <= lcp0;                                  // prefix on kmin, one read base skipped, suffix on kmin + 1
        const bool del_ok = max(sfx0 - 1, 0) <= min(lcp1, F - 1);          // prefix on kmin + 1, one base of the piece skipped, suffix on kmin
        if (margins && ins_ok != del_ok && at_least_4(b0, m0) && at_least_4(b1, m1)) {
          solved = true;
          if (lane == 0) {
            const uint64_t j = jd.out_index;
            const int start = ins_ok ? kmin : kmax, len = ins_ok ? F + 1 : F - 1;
            a.n_match[j] = ins_ok ? F : F - 1;
            a.span4[4 * j] = 0u; a.span4[4 * j + 1] = (uint32_t)F; a.span4[4 * j + 2] = (uint32_t)start; a.span4[4 * j + 3] = (uint32_t)(start + len);
            atomicAdd(&l_ns, 1u);
            atomicAdd(&l_ni, 1u);
          }
        }
      }
      if (solved) continue;
      int w0 = 0, wl = 0;
      // (front list: only a piece that lies inside the read whole.  A read that ends within the piece has seeds too, but its alignment
      //  deletes the rest of the piece -- far beyond the bound of the window argument: it would fail the check and be aligned against
      //  the whole read without the pre-filter in front; measured, that put 6 ms of alignments behind a 1.4-ms filter pass)
      if (kmin <= kmax && kmax - kmin <= a.spread && (!a.front || (kmin >= 0 && kmax + F <= n))) {
        // diagonals [kmin - margin, kmax + margin] start the alignment (text_begin_free = 2 margin + spread of the windowed launch, counted
        // from the window start) and run through at most flank_len more bases of the read
        const int lo = kmin - a.margin;
        w0 = lo > 0 ? lo : 0;
        wl = n - w0 < a.tbf + F ? n - w0 : a.tbf + F;
      }
      if (lane == 0) {
        if (wl > 0) { jd.txt_off += (uint64_t)w0; jd.txt_len = (uint32_t)wl; jd.pad = (uint32_t)w0; l_out[atomicAdd(&l_nw, 1u)] = jd; }
        else if (is_long) l_long[atomicAdd(&l_nl, 1u)] = jd;
        else l_out[WIN_JOBS_PER_WG - 1 - atomicAdd(&l_nr, 1u)] = jd;
      }
    }
    __syncthreads();
    if (threadIdx.x == 0 && l_nw) l_bw = atomicAdd(a.count + SC_WIN, l_nw);
    if (threadIdx.x == 64 && l_nr) l_br = atomicAdd(a.count + (a.front ? SC_NOSEED : SC_REST), l_nr);
    if (threadIdx.x == 65 && l_nl) l_bl = atomicAdd(a.count + SC_LNOSEED, l_nl);
    if (threadIdx.x == 128 && l_ns) atomicAdd(a.count + SC_SHORTCUT, l_ns);
    if (threadIdx.x == 192 && l_ni) atomicAdd(a.count + SC_GAPS, l_ni);  // (of those: one-base gaps)
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < l_nw; i += blockDim.x) a.win_jobs[l_bw + i] = l_out[i];
    for (uint32_t i = threadIdx.x; i < l_nr; i += blockDim.x) a.rest_jobs[l_br + i] = l_out[WIN_JOBS_PER_WG - 1 - i];
    for (uint32_t i = threadIdx.x; i < l_nl; i += blockDim.x) a.long_rest[l_bl + i] = l_long[i];
    __syncthreads();
  }
}

// One wavefront per READ, both flank pieces at once (flank_len >= 4).  A round covers 1024 read bytes: every lane loads 24
// consecutive bytes (16 + 8 of overlap), forms its 16 candidate windows with byte-align shifts and compares each with the first
// EIGHT bytes of the two pieces (four bases match by chance once in 256 positions: four false candidates per piece and read, each a
// dependent round trip to verify -- most of this kernel's time; eight bases: one in 65536); candidates are then verified in
// increasing position, cooperatively (lane i compares dword i of
// the piece), so the first full match is the leftmost occurrence, exactly windows().position() (span_locater.rs:10-12).
// A workgroup walks SCAN_READS_PER_WG reads (one per wave at a time) and collects its fallback jobs in LDS; one global atomic
// per workgroup reserves their slots in the job list (a per-job atomic on a single counter was most of this kernel's time).
constexpr int SCAN_READS_PER_WG = 64;
__global__ void __launch_bounds__(256) flank_scan_wide_kernel(const ScanArgs a) {
  __shared__ JobDev l_jobs[2 * SCAN_READS_PER_WG];  // expensive jobs fill it from the front, the others from the back
  __shared__ uint32_t l_n, l_n2, l_base, l_base2;
  if (threadIdx.x == 0) { l_n = 0; l_n2 = 0; }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const uint64_t n_reads = a.n_jobs >> 1;
  const uint64_t r_begin = (uint64_t)blockIdx.x * SCAN_READS_PER_WG, r_end = r_begin + SCAN_READS_PER_WG < n_reads ? r_begin + SCAN_READS_PER_WG : n_reads;
  // What a read's search needs before its first byte can be looked at -- length, offset, locus, the locus' two piece offsets, the
  // pieces' first eight bytes -- is a chain of four dependent loads, and with one read at a time per wave that chain (not instruction
  // issue, not bandwidth) was most of this kernel's time: 300 k reads x ~5 us / 4 k resident waves.  Round 5: lane j fetches it all for
  // the wave's j-th read up front (four rounds of loads per wave instead of per read); the loop takes it from that lane.
  const int F = a.flank_len;
  const bool h8 = F >= 8;
  uint32_t pf_n = 0, pf_h0 = 0, pf_h1 = 0, pf_h20 = 0, pf_h21 = 0, pf_heavy = 0;
  uint64_t pf_roff = 0, pf_po0 = 0, pf_po1 = 0;
  {
    const uint64_t rj = r_begin + (threadIdx.x >> 6) + 4ull * (uint64_t)lane;
    if (lane < SCAN_READS_PER_WG / 4 && rj < r_end) {
      pf_n = a.read_len[rj]; pf_roff = a.read_off[rj];
      const uint64_t loc = a.read_locus[rj];
      pf_po0 = a.piece_off[2 * loc]; pf_po1 = a.piece_off[2 * loc + 1];
      if (a.heavy_len) pf_heavy = a.heavy_len[loc];
      if ((int)pf_n >= F) {
        pf_h0 = load_u32(a.flank_blob + pf_po0); pf_h1 = load_u32(a.flank_blob + pf_po1);
        if (h8) { pf_h20 = load_u32(a.flank_blob + pf_po0 + 4); pf_h21 = load_u32(a.flank_blob + pf_po1 + 4); }
      }
    }
  }
  auto from_lane32 = [](uint32_t v, int j) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)v, j); };
  auto from_lane64 = [&](uint64_t v, int j) -> uint64_t { return (uint64_t)from_lane32((uint32_t)v, j) | ((uint64_t)from_lane32((uint32_t)(v >> 32), j) << 32); };
  int slot = 0;
  for (uint64_t r = r_begin + (threadIdx.x >> 6); r < r_end; r += 4, ++slot) {
  const int n = (int)from_lane32(pf_n, slot);
  const uint64_t po0 = from_lane64(pf_po0, slot), po1 = from_lane64(pf_po1, slot);
  const uint8_t* __restrict__ piece[2] = {a.flank_blob + po0, a.flank_blob + po1};
  const uint64_t r_off = from_lane64(pf_roff, slot);
  const uint32_t heavy_r = from_lane32(pf_heavy, slot);
  const uint8_t* __restrict__ read = a.read_blob + r_off;
  int found[2] = {-1, -1};
  if (n >= F) {
    const int last = n - F;  // last candidate start
    const uint32_t head[2] = {from_lane32(pf_h0, slot), from_lane32(pf_h1, slot)};
    const uint32_t head2[2] = {from_lane32(pf_h20, slot), from_lane32(pf_h21, slot)};
    const int nd = F >> 2;   // full dwords of a piece
    for (int base = 0; base <= last && (found[0] < 0 || found[1] < 0); base += 1024) {
      const int off = base + 16 * lane;
      uint32_t w[6] = {0, 0, 0, 0, 0, 0};
      if (off + 24 <= n) {
        uint4 q; uint2 q2;
        __builtin_memcpy(&q, read + off, 16); __builtin_memcpy(&q2, read + off + 16, 8);
        w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w; w[4] = q2.x; w[5] = q2.y;
      } else if (off <= last) {
        // (a lane whose 24 bytes cross the end of the read AND that still holds a candidate start: only with pieces shorter than 24
        //  bases.  With TRGT's 250 the lanes at the end of a read start beyond the last candidate, their bytes are never looked at --
        //  and this byte loop, which the whole wave walks when one lane enters it, was a third of the kernel's instructions)
        for (int b = 0; b < 24 && off + b < n; ++b) w[b >> 2] |= (uint32_t)read[off + b] << (8 * (b & 3));
      }
      uint32_t m[2] = {0, 0};  // bit s: candidate start off + s matches the head of piece 0 / 1
      // (eight head bytes as one 64-bit compare per position and piece; the positions beyond the last candidate start are masked once
      //  per round, not tested per position: this kernel is bound by instruction issue)
      const uint64_t h64[2] = {(uint64_t)head[0] | ((uint64_t)head2[0] << 32), (uint64_t)head[1] | ((uint64_t)head2[1] << 32)};
#pragma unroll
      for (int s16 = 0; s16 < 16; ++s16) {
        const uint32_t win = (s16 & 3) == 0 ? w[s16 >> 2] : __builtin_amdgcn_alignbyte(w[(s16 >> 2) + 1], w[s16 >> 2], s16 & 3);
        if (h8) {
          const uint32_t win2 = (s16 & 3) == 0 ? w[(s16 >> 2) + 1] : __builtin_amdgcn_alignbyte(w[(s16 >> 2) + 2], w[(s16 >> 2) + 1], s16 & 3);
          const uint64_t w64 = (uint64_t)win | ((uint64_t)win2 << 32);
          m[0] |= (uint32_t)(w64 == h64[0]) << s16;
          m[1] |= (uint32_t)(w64 == h64[1]) << s16;
        } else {
          m[0] |= (uint32_t)(win == head[0]) << s16;
          m[1] |= (uint32_t)(win == head[1]) << s16;
        }
      }
      {
        const int room = last - off;  // candidate starts off .. off + room are inside the read
        const uint32_t vmask = room >= 15 ? 0xFFFFu : room < 0 ? 0u : (1u << (room + 1)) - 1u;
        m[0] &= vmask; m[1] &= vmask;
      }
#pragma unroll
      for (int side = 0; side < 2; ++side) {
        if (found[side] >= 0) continue;
        uint32_t mm = m[side];
        for (;;) {
          const unsigned long long lanes = __ballot(mm != 0u);
          if (!lanes) break;
          const int lq = __ffsll((long long)lanes) - 1;
          const uint32_t mq = (uint32_t)__builtin_amdgcn_readlane((int)mm, lq);
          const int sq = __ffs((int)mq) - 1;
          const int p = base + 16 * lq + sq;
          bool ok = true;  // lane i verifies dwords i, i + 64, ... and the last lane the tail bytes
          for (int i = lane; i < nd; i += 64) ok = ok && load_u32(read + p + 4 * i) == load_u32(piece[side] + 4 * i);
          if (lane == 63 && (F & 3)) ok = ok && load_u32(read + p + F - 4) == load_u32(piece[side] + F - 4);  // the last one to three bytes: the dword that ends the piece (F >= 4)
          if (__ballot(!ok) == 0ull) { found[side] = p; break; }
          if (lane == lq) mm &= mm - 1u;  // drop this candidate
        }
      }
    }
  }
  if (lane == 0) {
    for (int side = 0; side < 2; ++side) {
      const uint64_t j = 2 * r + side;
      a.pos[j] = found[side]; a.n_match[j] = -1;
      if (found[side] < 0) {  // fall back to the wavefront aligner (span_locater.rs:13-26)
        JobDev jd;
        jd.pat_off = side ? po1 : po0; jd.txt_off = r_off;
        jd.cigar_off = 0; jd.ops_off = 0; jd.pat_len = (uint32_t)F; jd.txt_len = (uint32_t)n; jd.out_index = (uint32_t)j; jd.pad = 0;
        if ((uint32_t)n > a.long_tlen) a.wfa_jobs_long[atomicAdd(a.wfa_count + SC_LONG, 1u)] = jd;  // rare: straight to the second list
        else if (!a.heavy_len || (uint32_t)n < heavy_r) l_jobs[atomicAdd(&l_n, 1u)] = jd;
        else l_jobs[2 * SCAN_READS_PER_WG - 1 - atomicAdd(&l_n2, 1u)] = jd;
      }
    }
  }
  }  // reads of this workgroup
  __syncthreads();
  if (threadIdx.x == 0 && l_n) l_base = atomicAdd(a.wfa_count + SC_HEAVY, l_n);
  if (threadIdx.x == 64 && l_n2) l_base2 = atomicAdd(a.wfa_count + SC_LIGHT, l_n2);
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < l_n; i += blockDim.x) a.wfa_jobs[l_base + i] = l_jobs[i];
  for (uint32_t i = threadIdx.x; i < l_n2; i += blockDim.x) a.wfa_jobs[a.jobs_cap - 1u - (l_base2 + i)] = l_jobs[2 * SCAN_READS_PER_WG - 1 - i];
}

// After the windowed launch: a windowed alignment whose penalty is within the bound of piece_window's argument IS the alignment of
// the whole read, shifted by the window start; any other goes to the back of the job list and is aligned against the whole read.
struct WinCheckArgs {
  const JobDev* win_jobs; const uint32_t* n_win; const int32_t* score; uint32_t* span4; int32_t* n_match; int32_t s0;
  JobDev* wfa_jobs; uint32_t* wfa_count; uint32_t jobs_cap; const uint64_t* read_off; const uint32_t* read_len;
  JobDev* long_rest; uint32_t long_tlen;  // a long read's alignment whose window did not stand: to the long reads' list (NULL: there is none)
};
__global__ void window_check_kernel(const WinCheckArgs a) {
  const uint32_t n = *a.n_win;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    JobDev jd = a.win_jobs[i];
    const uint32_t j = jd.out_index;
    const int32_t sc = a.score[j];  // -penalty, INT32_MIN when the alignment did not complete
    if (sc != INT32_MIN && -sc <= a.s0 && a.n_match[j] > 0) {
      a.span4[4 * (uint64_t)j + 2] += jd.pad; a.span4[4 * (uint64_t)j + 3] += jd.pad;
    } else {
      a.n_match[j] = -1;
      jd.txt_off = a.read_off[j >> 1]; jd.txt_len = a.read_len[j >> 1]; jd.pad = 0;
      if (a.long_rest && jd.txt_len > a.long_tlen) a.long_rest[atomicAdd(a.wfa_count + SC_LNOSEED, 1u)] = jd;
      else a.wfa_jobs[atomicAdd(a.wfa_count + SC_REST, 1u)] = jd;  // (wfa_jobs: the rest list here)
    }
  }
}

// ---- The back-trace of what the pre-filter keeps, confined to a band ----
// The pre-filter hands over more than a verdict: the optimal penalty s* and the diagonal k_end the run ends on (the first one with
// the pattern consumed at level s*).  A wavefront cell (s, k) depends only on cells (s', k') with |k' - k| <= (s - s') / e, and the
// back-trace from (s*, k_end) reads nothing but such cells: everything it touches lies on the diagonals [k_end - s*, k_end + s*],
// text positions [k_end - s*, k_end + s* + plen).  The alignment of the piece against THAT window, started on those diagonals only,
// computes these cells bit for bit; its other cells can only be smaller than in the full run (fewer sources under the max), so it can
// neither end at a lower level nor, at level s*, on a diagonal below k_end.  band_check_kernel compares the penalty anyway and
// sends what differs to the whole-read launch.  A full run spends its time on the ~tlen + 2 s diagonals of every level; the banded
// one on 2 s* + 2 s + 1.
struct BandArgs {
  const JobDev* keep_jobs; const uint32_t* n_keep; JobDev* band_jobs; JobDev* rest_jobs; uint32_t* count; int32_t s_max;
  const int32_t* score; uint32_t* span4; int32_t* n_match; const uint64_t* read_off; const uint32_t* read_len;
  int32_t i_band, i_rest;  // which counters of the block (SC_BAND / SC_HREST, or SC_LBAND / SC_LREST for the long reads)
};
__global__ void heavy_band_kernel(const BandArgs a) {
  const uint32_t n = *a.n_keep;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    JobDev jd = a.keep_jobs[i];
    const uint32_t pad = jd.pad;
    const int s = (int)((pad >> 16) & 0x7FFFu), plen = (int)jd.pat_len, tlen = (int)jd.txt_len, k_end = (int)(pad & 0xFFFFu) - plen;
    const int t0 = max(0, k_end - s), k_hi = min(k_end + s, tlen), t_end = min(tlen, k_end + s + plen);
    if ((pad >> 31) && s <= a.s_max && k_hi >= t0 && t_end > t0) {
      jd.txt_off += (uint64_t)t0; jd.txt_len = (uint32_t)(t_end - t0); jd.pad = (uint32_t)t0;
      jd.ops_off = (uint64_t)(k_hi - t0);  // the free text start of this job (wfa_fast_kernel<64, 3>)
      jd.cigar_off = (uint64_t)s;          // the penalty it must come out with
      a.band_jobs[atomicAdd(a.count + a.i_band, 1u)] = jd;
    } else {
      jd.pad = 0;
      a.rest_jobs[atomicAdd(a.count + a.i_rest, 1u)] = jd;
    }
  }
}
__global__ void band_check_kernel(const BandArgs a) {
  const uint32_t n = a.count[a.i_band];
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    JobDev jd = a.band_jobs[i];
    const uint32_t j = jd.out_index;
    const int32_t sc = a.score[j];  // -penalty, INT32_MIN when the alignment did not complete
    if (sc != INT32_MIN && -sc == (int32_t)jd.cigar_off && a.n_match[j] > 0) {
      a.span4[4 * (uint64_t)j + 2] += jd.pad; a.span4[4 * (uint64_t)j + 3] += jd.pad;
    } else {
      a.n_match[j] = -1;
      jd.txt_off = a.read_off[j >> 1]; jd.txt_len = a.read_len[j >> 1]; jd.pad = 0; jd.ops_off = 0; jd.cigar_off = 0;
      a.rest_jobs[atomicAdd(a.count + a.i_rest, 1u)] = jd;
      atomicAdd(a.count + SC_BANDFAIL, 1u);
    }
  }
}

struct CombineArgs {
  uint64_t n_reads; int32_t flank_len; double threshold;
  const int32_t* pos; const int32_t* n_match; const uint32_t* span4;
  int32_t* span_start; int32_t* span_end; uint8_t* lf_hit; uint8_t* rf_hit;
  const uint32_t* count; unsigned long long* cells;  // count[7] (alignments settled by the substitution shortcut) -> cells[2], next to the offset counters
};

__global__ void span_combine_kernel(const CombineArgs a) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r == 0 && a.cells) a.cells[2] = a.count[SC_SHORTCUT];
  if (r >= a.n_reads) return;
  int s[2], e[2], hit[2];
  for (int side = 0; side < 2; ++side) {
    const uint64_t j = 2 * r + side;
    s[side] = e[side] = -1; hit[side] = 0;
    if (a.pos[j] >= 0) { s[side] = a.pos[j]; e[side] = a.pos[j] + a.flank_len; hit[side] = 1; }
    else if ((double)(uint64_t)(uint32_t)a.n_match[j] >= a.threshold && a.n_match[j] >= 0) {  // span_locater.rs:18-22
      s[side] = (int)a.span4[4 * j + 2]; e[side] = (int)a.span4[4 * j + 3]; hit[side] = 2;
    }
  }
  int ss = -1, ee = -1;
  if (hit[0] && hit[1] && e[0] <= s[1]) { ss = e[0]; ee = s[1]; }  // :59-65
  a.span_start[r] = ss; a.span_end[r] = ee;
  if (a.lf_hit) a.lf_hit[r] = (uint8_t)hit[0];
  if (a.rf_hit) a.rf_hit[r] = (uint8_t)hit[1];
}

__global__ void cells_merge_kernel(unsigned long long* total2, const unsigned long long* heavy) { total2[1] = heavy[0]; total2[0] += heavy[0]; }

// ---- reads beyond the dedicated kernels' texts (alleles of kilobases): the pre-filter judges them WINDOW BY WINDOW.
// An optimal ends-free alignment of a flank piece costs at most o + plen e (the piece deleted as a whole), so it inserts at most that
// many bases and covers at most span_max = 2 plen + o text bases: with windows of wl bases every `step` = wl - span_max bases (and one
// flush with the end of the read) every alignment the whole read admits lies inside a window, with the same penalty and the same
// matches.  If the filter rejects every window -- no alignment of the window that can still be optimal has min_matches matches -- then
// no alignment of the read has, whichever of the co-optimal ones WFA2-lib would return: the job is dropped as the filter drops a
// short one.  A job with a window that is kept (or could not be judged) goes to the exact kernel with its whole read, as before.
struct LongWinArgs {
  const JobDev* jobs; const uint32_t* n_jobs;  // the long list and its length
  JobDev* sub; uint32_t* parent; uint8_t* sub_keep; uint32_t* n_sub; uint32_t cap;
  uint8_t* job_keep;                           // per long job: 1 = to the exact kernel
  JobDev* kept; uint32_t* n_kept;
  int32_t wl, step;
  // The windows also say WHERE the alignment of a kept job is (FilterArgs::band): every alignment the read admits lies inside a window
  // with the same penalty, so the read's optimal penalty is the smallest one a window reports and its end diagonal the smallest among
  // those windows (a terminating diagonal of a window's run terminates in the whole read's run too, and the first one of the read is
  // the first one of the window that holds its alignment).  That holds when every window either completed or was rejected at a level
  // above that penalty; then the kept job carries penalty and diagonal for the banded back-trace (heavy_band_kernel), else 0.
  const uint32_t* sub_band; uint32_t* job_best; uint32_t* job_rej;  // per long job: min (penalty << 16 | biased end diagonal in the read) | min level of a rejected window (0: a window was not judged)
};
__device__ __forceinline__ uint32_t long_windows_of(uint32_t tlen, int wl, int step) {
  return tlen <= (uint32_t)wl ? 1u : (tlen - (uint32_t)wl + (uint32_t)step - 1u) / (uint32_t)step + 1u;
}
// The windows of a job take a range of the list reserved with one atomic (their order in the list does not matter).  A job whose range
// passes the end of the list is kept unseen; the part of its range that is inside the list is filled with copies of its first window,
// so that the list [0, min(n_sub, cap)) holds valid jobs only.
__global__ void long_windows_kernel(const LongWinArgs a) {
  const uint32_t n = *a.n_jobs;
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    const JobDev job = a.jobs[j];
    const uint32_t tlen = job.txt_len, nw = long_windows_of(tlen, a.wl, a.step);
    const uint32_t off = atomicAdd(a.n_sub, nw);
    const bool fits = off <= a.cap && nw <= a.cap - off;
    a.job_keep[j] = fits ? 0 : 1;
    if (a.job_best) { a.job_best[j] = 0xFFFFFFFFu; a.job_rej[j] = fits ? 0xFFFFFFFFu : 0u; }
    for (uint32_t w = 0; w < nw && off + w < a.cap; ++w) {
      uint32_t start = fits ? w * (uint32_t)a.step : 0u;
      if (tlen > (uint32_t)a.wl && start + (uint32_t)a.wl > tlen) start = tlen - (uint32_t)a.wl;  // the last window: flush with the end
      JobDev sj = job;
      sj.txt_off = job.txt_off + start; sj.txt_len = tlen > (uint32_t)a.wl ? (uint32_t)a.wl : tlen; sj.out_index = off + w;
      a.sub[off + w] = sj; a.parent[off + w] = j;
    }
  }
}
__global__ void long_verdict_kernel(const LongWinArgs a) {  // a kept window keeps its job
  const uint32_t n = min(*a.n_sub, a.cap);
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
    const uint32_t p = a.parent[s];
    if (a.sub_keep[s]) a.job_keep[p] = 1;
    if (a.sub_band) {
      const uint32_t b = a.sub_band[s];
      if (b >> 31) {
        const uint64_t kbg = (uint64_t)(b & 0xFFFFu) + (a.sub[s].txt_off - a.jobs[p].txt_off);  // the diagonal counted from the start of the read
        if (kbg < 0x10000u) atomicMin(a.job_best + p, (b & 0x7FFF0000u) | (uint32_t)kbg);
        else atomicMin(a.job_rej + p, 0u);
      } else atomicMin(a.job_rej + p, (b >> 30) ? (b & 0x3FFFFFFFu) : 0u);
    }
  }
}
__global__ void long_kept_kernel(const LongWinArgs a) {
  const uint32_t n = *a.n_jobs;
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
    if (a.job_keep[j]) {
      JobDev jd = a.jobs[j];
      jd.pad = 0;
      if (a.job_best) {
        const uint32_t best = a.job_best[j];
        if (best != 0xFFFFFFFFu && (best >> 16) < a.job_rej[j]) jd.pad = 0x80000000u | best;
      }
      a.kept[atomicAdd(a.n_kept, 1u)] = jd;
    }
}

// Device-side part shared with trgt_locus_batch: everything already resident, results left on the device.
int find_spans_device(trgt_hip_ctx* c, const trgt_span_params& p, int64_t n_loci, int64_t n_reads, const uint8_t* d_flank,
                      const uint64_t* d_piece_off, const uint8_t* d_reads, const uint64_t* d_read_off, const uint32_t* d_read_len,
                      const uint32_t* d_read_locus, uint32_t max_read_len, int32_t* d_span_start, int32_t* d_span_end,
                      uint8_t* d_lf_hit, uint8_t* d_rf_hit, const uint32_t* d_heavy_len, uint32_t heavy_tlen_max) {
  const uint64_t n_jobs = 2ull * (uint64_t)n_reads;
  void *d_pos = nullptr, *d_wjobs = nullptr, *d_count = nullptr, *d_span4 = nullptr, *d_nmatch = nullptr;
  int rc;
  if ((rc = dev_get(c, S_FS_POS, n_jobs * 4, &d_pos)) || (rc = dev_get(c, S_FS_WFAJOBS, n_jobs * sizeof(JobDev), &d_wjobs)) ||
      (rc = dev_get_zeroed(c, S_FS_COUNT, 4 * SC_WORDS, &d_count, c->stream)) || (rc = dev_get(c, S_FS_SPAN, n_jobs * 16, &d_span4)) ||
      (rc = dev_get(c, S_FS_NMATCH, n_jobs * 4, &d_nmatch)))
    return rc;
  ScanArgs sa;
  sa.flank_blob = d_flank; sa.read_blob = d_reads; sa.piece_off = d_piece_off; sa.read_off = d_read_off; sa.read_len = d_read_len;
  sa.read_locus = d_read_locus; sa.n_jobs = n_jobs; sa.flank_len = p.flank_len; sa.pos = (int32_t*)d_pos; sa.n_match = (int32_t*)d_nmatch;
  sa.wfa_jobs = (JobDev*)d_wjobs; sa.wfa_count = (uint32_t*)d_count;
  sa.heavy_len = d_heavy_len; sa.jobs_cap = (uint32_t)n_jobs;
  // reads up to long_tlen keep the dedicated kernel at 4 workgroups per CU (LDS: ring + windows <= ~39 KB per alignment)
  const int ring_slots = std::max(p.mism, p.gapo + p.gape) + 1 + 2 * (p.gape + 1);
  const int64_t fit = 39000 / (2 * (int64_t)ring_slots + 4) - p.flank_len - 16;
  const uint32_t long_tlen = (uint32_t)std::max<int64_t>(fit, 64);
  const bool has_long = max_read_len > long_tlen;
  void* d_wjobs_long = nullptr;
  if (has_long && (rc = dev_get(c, S_FS_WFAJOBS_LONG, n_jobs * sizeof(JobDev), &d_wjobs_long))) return rc;
  sa.wfa_jobs_long = (JobDev*)d_wjobs_long; sa.long_tlen = has_long ? long_tlen : 0xFFFFFFFFu;
  // Seeded windows (piece_window): eight segments by default; the conditions make a mismatch the cheapest way to spoil a segment
  // (a gap inside one segment costs o + e, a deletion across t segments o + ((t - 2) q + 2) e >= t x), so that S0 = (segments) x - 1.
  int32_t win_s0 = 0, win_q = 0, win_margin = 0, win_spread = 0, win_m = WIN_SEGMENTS_DEFAULT; uint32_t win_tlen = 0;
  void *d_winjobs = nullptr, *d_restjobs = nullptr, *d_score = nullptr;
  {
    int m = c->knobs.win_segments;
    if (m != 4 && m != 6 && m != 8) m = WIN_SEGMENTS_DEFAULT;
    win_m = m;
    const int q = p.flank_len / m, x = p.mism, o = p.gapo, e = p.gape;
    const bool two_launches = d_heavy_len && heavy_tlen_max > 0 && !c->knobs.one_launch;
    const bool ok = two_launches && !c->knobs.no_window && q >= 12 && x >= 1 && e >= 1 && o >= 0 &&
                    q * e >= x && o + 2 * e >= 2 * x && o + e >= x;
    if (ok) {
      win_s0 = x * m - 1;
      const int G = win_s0 >= o + e ? (win_s0 - o) / e : 0, C = win_s0 / e;
      win_q = q; win_margin = G + C; win_spread = 2 * G;
      win_tlen = (uint32_t)(p.flank_len + 2 * win_margin + win_spread);
      if (win_tlen + 64 >= max_read_len) win_q = 0;  // reads hardly longer than a window
    }
    if (win_q > 0 && ((rc = dev_get(c, S_FS_WINJOBS, n_jobs * sizeof(JobDev), &d_winjobs)) || (rc = dev_get(c, S_FS_RESTJOBS, n_jobs * sizeof(JobDev), &d_restjobs)) ||
                      (rc = dev_get(c, S_FS_SCORE, n_jobs * 4, &d_score))))
      return rc;
  }
  {
    KTimer t(c, TRGT_K_FLANK_SCAN);
    if (p.flank_len >= 4) hipLaunchKernelGGL(flank_scan_wide_kernel, dim3((unsigned)((n_reads + SCAN_READS_PER_WG - 1) / SCAN_READS_PER_WG)), dim3(256), 0, c->stream, sa);
    else hipLaunchKernelGGL(flank_scan_kernel, dim3((unsigned)((n_jobs + 3) / 4)), dim3(256), 0, c->stream, sa);
    TRGT_HIP_TRY(c, hipGetLastError());
    t.stop(0);
  }
  tl_mark(c, "scan launched");
  trgt_wfa_params wp;
  trgt_wfa_default_params(&wp);  // THREAD_WFA_FLANK (genotype.rs:66-80)
  wp.metric = 3; wp.mismatch = p.mism; wp.gap_open1 = p.gapo; wp.gap_ext1 = p.gape;
  wp.span = 1; wp.pattern_begin_free = 0; wp.pattern_end_free = 0; wp.text_begin_free = -1; wp.text_end_free = -1;
  wp.scope = 1; wp.memory_mode = 0; wp.heuristic = 0;
  // The number of fallback alignments is known only on the device: the kernel reads it there (n_jobs_dev), the planner sizes
  // the workspace for the upper bound (every job falls back), and nothing here waits for the GPU -- trgt_locus_batch enqueues
  // this function chunk after chunk.
  c->last_wfa_cells_dev = nullptr;
  WfaLaunch L;
  L.jobs_dev = (const JobDev*)d_wjobs; L.n_jobs_host = (int64_t)n_jobs; L.n_jobs_dev = (const uint32_t*)d_count;
  L.n_jobs2_dev = (const uint32_t*)d_count + SC_LIGHT; L.jobs_cap = (uint32_t)n_jobs;
  L.pat_base = d_flank; L.txt_base = d_reads;
  const uint32_t short_max = has_long ? long_tlen : max_read_len;
  L.max_plen = p.flank_len; L.max_tlen = short_max; L.max_sum = (int64_t)p.flank_len + short_max;
  L.threads = c->knobs.flank_threads;
  L.timer_slot = TRGT_K_WFA_FLANK;
  L.n_match = (int32_t*)d_nmatch; L.span4 = (uint32_t*)d_span4;
  // Two launches.  The front of the list (reads too short to span their locus: 95 % of the wavefront offsets) holds short texts
  // only, so its launch is planned for heavy_tlen_max: a smaller LDS ring per alignment and one more resident workgroup per CU
  // (occupancy is what this latency-bound kernel lives on: 16.9 -> 15.1 ms from 4 to 5 per CU).  The rest follows at the full size.
  // (a catalog with a few long-read loci has heavy_tlen_max beyond the dedicated kernel's texts: those reads are on the long list
  //  anyway, the launch over the expensive alignments is planned for what is left)
  const bool split = d_heavy_len && heavy_tlen_max > 0 && !c->knobs.one_launch;
  heavy_tlen_max = std::min(heavy_tlen_max, short_max);
  bool heavy_window = false;  // the seed search runs over the expensive list as well (below)
  bool heavy_band = false;    // ... and what the pre-filter keeps of it is back-traced inside a band (BandArgs)
  auto window_args = [&]() {
    WindowArgs wa;
    wa.flank_blob = d_flank; wa.read_blob = d_reads; wa.wfa_jobs = (const JobDev*)d_wjobs; wa.jobs_cap = (uint32_t)n_jobs; wa.count = (uint32_t*)d_count;
    wa.win_jobs = (JobDev*)d_winjobs; wa.rest_jobs = (JobDev*)d_restjobs; wa.flank_len = p.flank_len; wa.q = win_q; wa.margin = win_margin; wa.spread = win_spread; wa.tbf = 2 * win_margin + win_spread;
    wa.front = 0; wa.long_jobs = nullptr; wa.long_rest = nullptr;
    wa.hamming_max = c->knobs.no_hamming ? 0 : std::min(std::min(win_m - 1, (p.gapo + p.gape - 1) / p.mism), 4);  // (4: the kernel's count is exact up to there)
    wa.indel_ok = !c->knobs.no_hamming && !c->knobs.no_indel_shortcut && p.mism == 2 && p.gapo == 5 && p.gape == 1 && wa.hamming_max == 2 ? 1 : 0;
    wa.n_match = (int32_t*)d_nmatch; wa.span4 = (uint32_t*)d_span4;
    return wa;
  };
  auto launch_window = [&](const WindowArgs& wa) {  // on the current stream
    const dim3 wgrid((unsigned)std::max<int64_t>(1, std::min<int64_t>((int64_t)c->num_cus * 8, (int64_t)((n_jobs + WIN_JOBS_PER_WG - 1) / WIN_JOBS_PER_WG))));
    if (win_m == 4) hipLaunchKernelGGL(flank_window_kernel<4>, wgrid, dim3(256), 0, c->stream, wa);
    else if (win_m == 6) hipLaunchKernelGGL(flank_window_kernel<6>, wgrid, dim3(256), 0, c->stream, wa);
    else hipLaunchKernelGGL(flank_window_kernel<8>, wgrid, dim3(256), 0, c->stream, wa);
  };
  void* d_long_noseed = nullptr;                                   // the long reads' list behind the seed search (when that runs over it)
  const JobDev* long_in = (const JobDev*)d_wjobs_long; int long_in_count = SC_LONG;
  c->last_filter_cells_dev = nullptr;
  bool heavy_join = false;         // the expensive alignments run on the second stream: wait for it before the spans are combined
  void* heavy_cells_dev = nullptr; // ... and their offset counter lives in workspace set 1
  if (split) {
    WfaLaunch LH = L;
    LH.n_jobs2_dev = nullptr; LH.jobs_cap = 0;
    LH.max_tlen = heavy_tlen_max; LH.max_sum = (int64_t)p.flank_len + heavy_tlen_max;
    // The expensive alignments first meet the register-resident pre-filter (wfa_reg.hip): exact penalty and an upper bound on
    // count_matches() without history or back-trace; only those whose bound reaches the threshold (a quarter of the jobs, 5 % of
    // the wavefront offsets on the bench workload) are aligned again by the back-tracing kernel.  span_locater.rs:18-22 does
    // nothing with the others but drop them.
    const double thr = (double)(uint64_t)p.flank_len * p.min_flank_id_frac;
    int64_t min_matches = thr > 0 ? (int64_t)std::ceil(thr) : 0;
    while (min_matches > 0 && (double)(min_matches - 1) >= thr) --min_matches;
    while ((double)min_matches < thr) ++min_matches;
    // (texts beyond the filter's diagonals are kept unseen, job by job: a batch with a few long reads still filters the others)
    const int64_t flt_tlen = flank_filter_max_tlen(p.flank_len);
    const bool filter_pen = (p.mism == 2 && p.gapo == 5 && p.gape == 1) || (p.mism == 1 && p.gapo == 0 && p.gape == 1);  // wgs / targeted presets (cli.rs:271-280)
    const bool use_filter = filter_pen && flt_tlen >= 2 * (int64_t)p.flank_len &&
                            heavy_tlen_max >= (uint32_t)p.flank_len && min_matches <= 254 && !c->knobs.no_filter;
    // Two streams: the expensive alignments (pre-filter, then the back-tracing kernel over what it keeps) run on the second stream
    // NEXT TO the other fallback alignments (segment search, windowed launch, whole-read launch) instead of in front of them.  The
    // filter is bound by VALU issue at three waves per SIMD, the light launches by latency and by their job-claim atomics: they
    // fill each other's gaps (measured: see DESIGN.md).  Workspace set 1 for the back-tracing launch of that stream.
    const bool two_streams = use_filter && !c->knobs.one_stream;
    void* cells_heavy = nullptr;
    if (two_streams) {
      if (!c->stream2) TRGT_HIP_TRY(c, trgt::make_stream(c, &c->stream2));
      if (!c->ev_scan) TRGT_HIP_TRY(c, hipEventCreateWithFlags(&c->ev_scan, hipEventDisableTiming));
      if (!c->ev_heavy) TRGT_HIP_TRY(c, hipEventCreateWithFlags(&c->ev_heavy, hipEventDisableTiming));
      TRGT_HIP_TRY(c, hipEventRecord(c->ev_scan, c->stream));
      TRGT_HIP_TRY(c, hipStreamWaitEvent(c->stream2, c->ev_scan, 0));
      std::swap(c->stream, c->stream2);
    }
    struct StreamBack { trgt_hip_ctx* c; bool on; ~StreamBack() { if (on) std::swap(c->stream, c->stream2); } } stream_back{c, two_streams};
    // The expensive list first meets the seed search too.  "Too short to span the locus" says nothing about the flank the read DOES
    // hold: a fifth of these alignments have a penalty below 8 (tools/filter_hist.py) -- the pre-filter let them through after a few
    // levels and the back-tracing kernel aligned them a second time, at the end of this stream's chain.  With seeds they are settled
    // by the substitution shortcut or join the windowed launch of the other stream; the pre-filter sees the jobs WITHOUT seeds only.
    heavy_window = use_filter && two_streams && win_q > 0 && !c->knobs.no_heavy_window;
    void* d_noseed = nullptr;
    if (heavy_window) {
      if ((rc = dev_get(c, S_FS_NOSEED, n_jobs * sizeof(JobDev), &d_noseed))) return rc;
      if (!c->ev_hwin) TRGT_HIP_TRY(c, hipEventCreateWithFlags(&c->ev_hwin, hipEventDisableTiming));
      WindowArgs wh = window_args();
      wh.front = 1; wh.rest_jobs = (JobDev*)d_noseed;
      KTimer t(c, TRGT_K_FLANK_WINDOW);
      launch_window(wh);
      TRGT_HIP_TRY(c, hipGetLastError());
      t.stop(0);
      TRGT_HIP_TRY(c, hipEventRecord(c->ev_hwin, c->stream));
    }
    if (use_filter) {
      void* d_keepjobs = nullptr;
      if ((rc = dev_get(c, S_FS_KEEPJOBS, n_jobs * sizeof(JobDev), &d_keepjobs))) return rc;
      FilterLaunch FL;
      FL.jobs_dev = heavy_window ? (const JobDev*)d_noseed : (const JobDev*)d_wjobs; FL.n_jobs_host = (int64_t)n_jobs;
      FL.n_jobs_dev = (const uint32_t*)d_count + (heavy_window ? SC_NOSEED : SC_HEAVY);
      FL.pat_base = d_flank; FL.txt_base = d_reads; FL.max_plen = p.flank_len; FL.max_tlen = std::min<int64_t>(heavy_tlen_max, flt_tlen);
      FL.mism = p.mism; FL.gapo = p.gapo; FL.gape = p.gape; FL.count_offsets = c->timing; FL.min_matches = (int32_t)min_matches; FL.early_reject = !c->knobs.no_early; FL.keep_jobs = (JobDev*)d_keepjobs; FL.keep_count = (uint32_t*)d_count + SC_KEEP;
      if ((rc = flank_filter_launch(c, FL))) return rc;
      tl_mark(c, "filter launched");
      LH.jobs_dev = (const JobDev*)d_keepjobs; LH.n_jobs_dev = (const uint32_t*)d_count + SC_KEEP;
      heavy_band = c->knobs.heavy_band > 0 && p.mism == 2 && p.gapo == 5 && p.gape == 1 && !c->knobs.no_spec && !c->knobs.skip_bt && !c->knobs.wfa_no_stage;  // (the banded launch exists as the LDS kernel of TRGT's configuration only)
    }
    if (heavy_band) {  // (see BandArgs) the kept alignments inside their band, one wave each; then whatever is left over the whole read
      void *d_band = nullptr, *d_hrest = nullptr, *d_bscore = nullptr;
      if ((rc = dev_get(c, S_FS_BANDJOBS, n_jobs * sizeof(JobDev), &d_band)) || (rc = dev_get(c, S_FS_HRESTJOBS, n_jobs * sizeof(JobDev), &d_hrest)) ||
          (rc = dev_get(c, S_FS_BSCORE, n_jobs * 4, &d_bscore)))
        return rc;
      const int s_max = c->knobs.heavy_band;
      BandArgs ba;
      ba.keep_jobs = LH.jobs_dev; ba.n_keep = LH.n_jobs_dev; ba.band_jobs = (JobDev*)d_band; ba.rest_jobs = (JobDev*)d_hrest; ba.count = (uint32_t*)d_count;
      ba.s_max = s_max; ba.score = (const int32_t*)d_bscore; ba.span4 = (uint32_t*)d_span4; ba.n_match = (int32_t*)d_nmatch;
      ba.read_off = d_read_off; ba.read_len = d_read_len; ba.i_band = SC_BAND; ba.i_rest = SC_HREST;
      hipLaunchKernelGGL(heavy_band_kernel, dim3(64), dim3(256), 0, c->stream, ba);
      TRGT_HIP_TRY(c, hipGetLastError());
      WfaLaunch LB = LH;
      LB.jobs_dev = (const JobDev*)d_band; LB.n_jobs_dev = (const uint32_t*)d_count + SC_BAND;
      LB.max_tlen = (int64_t)p.flank_len + 2 * s_max; LB.max_sum = (int64_t)p.flank_len + LB.max_tlen;
      LB.score = (int32_t*)d_bscore; LB.kernel_tag = 3; LB.max_score = s_max; LB.threads = c->knobs.band_threads;
      if (two_streams) LB.buffer_set = 1;
      trgt_wfa_params wpb = wp;
      wpb.text_begin_free = 2 * s_max;
      if ((rc = wfa_launch(c, wpb, LB))) return rc;
      hipLaunchKernelGGL(band_check_kernel, dim3(64), dim3(256), 0, c->stream, ba);
      TRGT_HIP_TRY(c, hipGetLastError());
      LH.jobs_dev = (const JobDev*)d_hrest; LH.n_jobs_dev = (const uint32_t*)d_count + SC_HREST; LH.keep_cells = true;
    }
    // three waves per alignment here: the wavefronts of these short texts are narrow (on average less than one 128-diagonal strip per
    // wave and level), and a wave without a strip still pays the per-level prologue and barrier (7.39 -> 7.21 ms)
    LH.threads = c->knobs.heavy_threads > 0 ? c->knobs.heavy_threads : (L.threads == 256 ? 192 : L.threads);
    if (two_streams) LH.buffer_set = 1;
    if ((rc = wfa_launch(c, wp, LH))) return rc;
    tl_mark(c, "heavy launch");
    cells_heavy = c->last_wfa_cells_dev;
    if (two_streams) {
      TRGT_HIP_TRY(c, hipEventRecord(c->ev_heavy, c->stream));
      std::swap(c->stream, c->stream2);
      stream_back.on = false;
      heavy_join = true;
    } else {
      // offsets of the first launch, kept next to the running total (cells[1]): the roofline of the dominant launch counts its own
      TRGT_HIP_TRY(c, hipMemcpyAsync((uint8_t*)c->last_wfa_cells_dev + 8, c->last_wfa_cells_dev, 8, hipMemcpyDeviceToDevice, c->stream));
    }
    heavy_cells_dev = cells_heavy;
    L.n_jobs_dev = (const uint32_t*)d_count + SC_ZERO;  // always 0: this launch takes the back part of the list only
    L.keep_cells = !two_streams; L.timer_slot = TRGT_K_WFA_FLANK_REST;  // (two streams: the first launch of THIS stream resets the counter of set 0)
    if (win_q > 0) {  // the alignments with a seeded window: short texts, more of them per CU; then sort out which of them stand
      // (the other stream's seed search appends to the same windowed list; this stream's search waits for it rather than running next to
      //  it: the pre-filter behind that one is the critical path of a call, this stream has 0.5 ms of slack -- side by side the short
      //  search took 0.34 instead of 0.08 ms)
      if (heavy_window) TRGT_HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_hwin, 0));
      {
        KTimer t(c, TRGT_K_FLANK_WINDOW);
        WindowArgs wl2 = window_args();
        if (has_long && !c->knobs.no_long_window) {  // the long reads' alignments meet the seed search too: shortcuts and windows do not care how long the read is
          if ((rc = dev_get(c, S_FS_LONGNOSEED, n_jobs * sizeof(JobDev), &d_long_noseed))) return rc;
          wl2.long_jobs = (const JobDev*)d_wjobs_long; wl2.long_rest = (JobDev*)d_long_noseed;
          long_in = (const JobDev*)d_long_noseed; long_in_count = SC_LNOSEED;
        }
        launch_window(wl2);
        TRGT_HIP_TRY(c, hipGetLastError());
        t.stop(0);
      }
      WfaLaunch LW = L;
      LW.jobs_dev = (const JobDev*)d_winjobs; LW.n_jobs_dev = (const uint32_t*)d_count + SC_WIN; LW.n_jobs2_dev = nullptr; LW.jobs_cap = 0;
      LW.max_tlen = win_tlen; LW.max_sum = (int64_t)p.flank_len + win_tlen;
      LW.score = (int32_t*)d_score; LW.kernel_tag = 2; LW.max_score = win_s0;
      // only the diagonals that can matter start the alignment: a wavefront of 2 margin + spread + 1 diagonals instead of one per base
      // of the window, i.e. one strip of one wave per level
      LW.threads = c->knobs.win_threads;
      trgt_wfa_params wpw = wp;
      wpw.text_begin_free = 2 * win_margin + win_spread;
      if ((rc = wfa_launch(c, wpw, LW))) return rc;
      tl_mark(c, "window launch");
      L.keep_cells = true;
      WinCheckArgs wc;
      wc.win_jobs = (const JobDev*)d_winjobs; wc.n_win = (const uint32_t*)d_count + SC_WIN; wc.score = (const int32_t*)d_score;
      wc.span4 = (uint32_t*)d_span4; wc.n_match = (int32_t*)d_nmatch; wc.s0 = win_s0;
      wc.wfa_jobs = (JobDev*)d_restjobs; wc.wfa_count = (uint32_t*)d_count; wc.jobs_cap = (uint32_t)n_jobs;
      wc.read_off = d_read_off; wc.read_len = d_read_len; wc.long_rest = (JobDev*)d_long_noseed; wc.long_tlen = long_tlen;
      hipLaunchKernelGGL(window_check_kernel, dim3(256), dim3(256), 0, c->stream, wc);
      TRGT_HIP_TRY(c, hipGetLastError());
      L.jobs_dev = (const JobDev*)d_restjobs; L.n_jobs_dev = (const uint32_t*)d_count + SC_REST; L.n_jobs2_dev = nullptr; L.jobs_cap = 0;
    }
  }
  if ((rc = wfa_launch(c, wp, L))) return rc;
  tl_mark(c, "rest launch");
  if (!split) TRGT_HIP_TRY(c, hipMemcpyAsync((uint8_t*)c->last_wfa_cells_dev + 8, c->last_wfa_cells_dev, 8, hipMemcpyDeviceToDevice, c->stream));
  if (has_long) {  // the long reads: same parameters, workspace and kernel choice planned for their size
    WfaLaunch L2 = L;
    L2.jobs_dev = long_in; L2.n_jobs_dev = (const uint32_t*)d_count + long_in_count; L2.n_jobs2_dev = nullptr; L2.jobs_cap = 0;
    L2.max_tlen = max_read_len; L2.max_sum = (int64_t)p.flank_len + max_read_len;
    L2.keep_cells = true; L2.timer_slot = TRGT_K_WFA_FLANK_REST;
    // the pre-filter over windows of the long reads (see LongWinArgs): what it rejects never reaches the exact kernel
    {
      const double thr = (double)(uint64_t)p.flank_len * p.min_flank_id_frac;
      int64_t min_matches = thr > 0 ? (int64_t)std::ceil(thr) : 0;
      while (min_matches > 0 && (double)(min_matches - 1) >= thr) --min_matches;
      while ((double)min_matches < thr) ++min_matches;
      const int64_t wl = flank_filter_max_tlen(p.flank_len);
      const int64_t span_max = 2 * (int64_t)p.flank_len + p.gapo + 8, step = wl - span_max;
      // It pays where the exact kernel is the generic one (wavefronts in HBM): texts beyond what the LDS kernel of wfa_launch takes
      // (ring of 11 levels x 2 B + 4 B of windows per diagonal in 96 KB: about 3 500 bases for 250-base pieces).  Reads just above the
      // dedicated launches' length (cfg4: up to 1 300 bases) go to that LDS kernel, and the extra filter launch cost 5 % there.
      const int64_t lds_kernel_tlen = (96 * 1024) / (2 * ring_slots + 4) - p.flank_len - 32;
      if (((p.mism == 2 && p.gapo == 5 && p.gape == 1) || (p.mism == 1 && p.gapo == 0 && p.gape == 1)) && min_matches >= 1 && min_matches <= 254 && step >= 256 && (int64_t)max_read_len > lds_kernel_tlen &&
          !c->knobs.no_filter && !c->knobs.no_long_filter) {
        const uint64_t w_max = (uint64_t)((int64_t)max_read_len > wl ? ((int64_t)max_read_len - wl + step - 1) / step + 1 : 1);
        const uint64_t cap = std::min<uint64_t>((uint64_t)n_jobs * w_max, 1ull << 21);
        void *d_sub = nullptr, *d_parent = nullptr, *d_subkeep = nullptr, *d_jobkeep = nullptr, *d_kept = nullptr, *d_lwc = nullptr;
        if ((rc = dev_get(c, S_LW_SUB, cap * sizeof(JobDev), &d_sub)) ||
            (rc = dev_get(c, S_LW_PARENT, cap * 4, &d_parent)) || (rc = dev_get(c, S_LW_SUBKEEP, cap, &d_subkeep)) ||
            (rc = dev_get(c, S_LW_JOBKEEP, n_jobs, &d_jobkeep)) || (rc = dev_get(c, S_LW_KEPT, n_jobs * sizeof(JobDev), &d_kept)) ||
            (rc = dev_get_zeroed(c, S_LW_COUNT, 16, &d_lwc, c->stream)))
          return rc;
        LongWinArgs lw;
        lw.jobs = long_in; lw.n_jobs = (const uint32_t*)d_count + long_in_count; 
        lw.sub = (JobDev*)d_sub; lw.parent = (uint32_t*)d_parent; lw.sub_keep = (uint8_t*)d_subkeep; lw.n_sub = (uint32_t*)d_lwc; lw.cap = (uint32_t)cap;
        lw.job_keep = (uint8_t*)d_jobkeep; lw.kept = (JobDev*)d_kept; lw.n_kept = (uint32_t*)d_lwc + 1; lw.wl = (int32_t)wl; lw.step = (int32_t)step;
        const bool long_band = c->knobs.heavy_band > 0 && p.mism == 2 && p.gapo == 5 && p.gape == 1 && !c->knobs.no_spec && !c->knobs.skip_bt && !c->knobs.wfa_no_stage && max_read_len < 0xF000u;
        void *d_sband = nullptr, *d_jbest = nullptr, *d_jrej = nullptr;
        if (long_band && ((rc = dev_get(c, S_LW_SUBBAND, cap * 4, &d_sband)) || (rc = dev_get(c, S_LW_JOBBEST, n_jobs * 4, &d_jbest)) || (rc = dev_get(c, S_LW_JOBREJ, n_jobs * 4, &d_jrej)))) return rc;
        lw.sub_band = (const uint32_t*)d_sband; lw.job_best = (uint32_t*)d_jbest; lw.job_rej = (uint32_t*)d_jrej;
        const dim3 g((unsigned)c->num_cus * 2), b(256);
        hipLaunchKernelGGL(long_windows_kernel, g, b, 0, c->stream, lw);
        TRGT_HIP_TRY(c, hipGetLastError());
        FilterLaunch FW;
        FW.jobs_dev = (const JobDev*)d_sub; FW.n_jobs_host = (int64_t)cap; FW.n_jobs_dev = (const uint32_t*)d_lwc;
        FW.pat_base = d_flank; FW.txt_base = d_reads; FW.max_plen = p.flank_len; FW.max_tlen = wl;
        FW.mism = p.mism; FW.gapo = p.gapo; FW.gape = p.gape; FW.min_matches = (int32_t)min_matches; FW.early_reject = !c->knobs.no_early; FW.keep = (uint8_t*)d_subkeep; FW.band = (uint32_t*)d_sband; FW.set = 1;
        if ((rc = flank_filter_launch(c, FW))) return rc;
        hipLaunchKernelGGL(long_verdict_kernel, g, b, 0, c->stream, lw);
        hipLaunchKernelGGL(long_kept_kernel, g, b, 0, c->stream, lw);
        TRGT_HIP_TRY(c, hipGetLastError());
        L2.jobs_dev = (const JobDev*)d_kept; L2.n_jobs_dev = (const uint32_t*)d_lwc + 1;
        if (long_band) {  // the kept reads whose windows named penalty and end diagonal: inside their band, as the short reads (BandArgs)
          void *d_lband = nullptr, *d_lrest = nullptr, *d_lscore = nullptr;
          if ((rc = dev_get(c, S_LW_BANDJOBS, n_jobs * sizeof(JobDev), &d_lband)) || (rc = dev_get(c, S_LW_RESTJOBS, n_jobs * sizeof(JobDev), &d_lrest)) ||
              (rc = dev_get(c, S_LW_BSCORE, n_jobs * 4, &d_lscore)))
            return rc;
          const int s_max = c->knobs.heavy_band;
          BandArgs ba;
          ba.keep_jobs = (const JobDev*)d_kept; ba.n_keep = (const uint32_t*)d_lwc + 1; ba.band_jobs = (JobDev*)d_lband; ba.rest_jobs = (JobDev*)d_lrest; ba.count = (uint32_t*)d_count;
          ba.s_max = s_max; ba.score = (const int32_t*)d_lscore; ba.span4 = (uint32_t*)d_span4; ba.n_match = (int32_t*)d_nmatch;
          ba.read_off = d_read_off; ba.read_len = d_read_len; ba.i_band = SC_LBAND; ba.i_rest = SC_LREST;
          hipLaunchKernelGGL(heavy_band_kernel, dim3(64), dim3(256), 0, c->stream, ba);
          TRGT_HIP_TRY(c, hipGetLastError());
          WfaLaunch LB = L2;
          LB.jobs_dev = (const JobDev*)d_lband; LB.n_jobs_dev = (const uint32_t*)d_count + SC_LBAND;
          LB.max_tlen = (int64_t)p.flank_len + 2 * s_max; LB.max_sum = (int64_t)p.flank_len + LB.max_tlen;
          LB.score = (int32_t*)d_lscore; LB.kernel_tag = 3; LB.max_score = s_max; LB.threads = c->knobs.band_threads;
          trgt_wfa_params wpb = wp;
          wpb.text_begin_free = 2 * s_max;
          if ((rc = wfa_launch(c, wpb, LB))) return rc;
          hipLaunchKernelGGL(band_check_kernel, dim3(64), dim3(256), 0, c->stream, ba);
          TRGT_HIP_TRY(c, hipGetLastError());
          L2.jobs_dev = (const JobDev*)d_lrest; L2.n_jobs_dev = (const uint32_t*)d_count + SC_LREST;
        }
      }
    }
    if ((rc = wfa_launch(c, wp, L2))) return rc;
  }
  if (heavy_join) {
    TRGT_HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_heavy, 0));
    // one counter pair again: [0] all flank alignments, [1] those of the launch over the expensive ones
    hipLaunchKernelGGL(cells_merge_kernel, dim3(1), dim3(1), 0, c->stream, (unsigned long long*)c->last_wfa_cells_dev, (const unsigned long long*)heavy_cells_dev);
    TRGT_HIP_TRY(c, hipGetLastError());
  }
  CombineArgs ca;
  ca.n_reads = (uint64_t)n_reads; ca.flank_len = p.flank_len;
  ca.threshold = (double)(uint64_t)p.flank_len * p.min_flank_id_frac;  // span_locater.rs:46
  ca.pos = (const int32_t*)d_pos; ca.n_match = (const int32_t*)d_nmatch; ca.span4 = (const uint32_t*)d_span4;
  ca.span_start = d_span_start; ca.span_end = d_span_end; ca.lf_hit = d_lf_hit; ca.rf_hit = d_rf_hit;
  ca.count = (const uint32_t*)d_count; ca.cells = (unsigned long long*)c->last_wfa_cells_dev;
  hipLaunchKernelGGL(span_combine_kernel, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, c->stream, ca);
  TRGT_HIP_TRY(c, hipGetLastError());
  if (c->knobs.debug) {  // (synchronises: developer output only)
    uint32_t h[16];
    TRGT_HIP_TRY(c, trgt::stream_wait(c, c->stream));
    TRGT_HIP_TRY(c, hipMemcpy(h, d_count, 64, hipMemcpyDeviceToHost));
    if (win_q > 0)
      fprintf(stderr, "[spans] fallback alignments: first launch %u, long reads %u, light %u -> windowed %u, whole read %u (that is %u without seeds + %u windows that did not stand), settled by the shortcuts %u (one-base gaps: %u)\n",
              h[SC_HEAVY], h[SC_LONG], h[SC_LIGHT], h[SC_WIN], h[SC_REST], h[SC_LIGHT] - h[SC_WIN] - h[SC_SHORTCUT], h[SC_REST] - (h[SC_LIGHT] - h[SC_WIN] - h[SC_SHORTCUT]), h[SC_SHORTCUT], h[SC_GAPS]);
    if (win_q <= 0) fprintf(stderr, "[spans] fallback alignments: first launch %u, long reads %u, light %u (no seeded windows for this configuration)\n", h[SC_HEAVY], h[SC_LONG], h[SC_LIGHT]);
    if (win_q > 0 && heavy_window) fprintf(stderr, "[spans+] (the seed search ran over the first launch's list too: %u of its %u alignments had no seeds and met the pre-filter; the counts of the windowed list and of the shortcut include the others)\n", h[SC_NOSEED], h[SC_HEAVY]);
    if (has_long) fprintf(stderr, "[spans+] long reads kept by the window filter: back-traced inside a band %u, over the whole read %u\n", h[SC_LBAND], h[SC_LREST]);
    if (heavy_band) fprintf(stderr, "[spans+] kept by the pre-filter: %u -> back-traced inside a band %u (did not stand: %u), over the whole read %u\n", h[SC_KEEP], h[SC_BAND], h[SC_BANDFAIL], h[SC_HREST]);
  }
  (void)n_loci;
  return TRGT_OK;
}

}  // namespace trgt

using namespace trgt;

extern "C" int trgt_find_spans_batch(trgt_hip_ctx* c, const trgt_span_params* p, int64_t n_loci, const uint8_t* flank_blob,
                                     const uint64_t* lf_off, const uint32_t* lf_len, const uint64_t* rf_off,
                                     const uint32_t* rf_len, const uint64_t* locus_read_begin, const uint8_t* read_blob,
                                     const uint64_t* read_off, const uint32_t* read_len, int32_t* span_start,
                                     int32_t* span_end, uint8_t* lf_hit, uint8_t* rf_hit) {
  if (!c) return TRGT_ERR_INVALID;
  if (!p || n_loci < 0 || (n_loci > 0 && (!flank_blob || !lf_off || !lf_len || !rf_off || !rf_len || !locus_read_begin ||
                                          !read_blob || !read_off || !read_len || !span_start || !span_end)))
    return fail(c, TRGT_ERR_INVALID, "trgt_find_spans_batch: null argument");
  if (n_loci == 0) return TRGT_OK;
  if (p->flank_len <= 0) return fail(c, TRGT_ERR_INVALID, "trgt_find_spans_batch: flank_len must be positive");
  TRGT_HIP_TRY(c, hipSetDevice(c->device));
  const int64_t n_reads = (int64_t)locus_read_begin[n_loci];
  if (n_reads == 0) return TRGT_OK;
  if (2 * n_reads > 0xFFFFFFF0ll) return fail(c, TRGT_ERR_UNSUPPORTED, "trgt_find_spans_batch: too many reads in one call");
  std::vector<uint64_t> piece_off(2 * (size_t)n_loci);
  std::vector<uint32_t> read_locus((size_t)n_reads), heavy_len((size_t)n_loci);
  uint64_t flank_total = 0, read_total = 0;
  uint32_t max_read_len = 0, heavy_tlen_max = 0;
  for (int64_t l = 0; l < n_loci; ++l) {
    if ((int64_t)lf_len[l] < p->flank_len || (int64_t)rf_len[l] < p->flank_len)
      return fail(c, TRGT_ERR_INVALID, "trgt_find_spans_batch: locus %lld flank shorter than flank_len", (long long)l);
    piece_off[2 * l] = lf_off[l] + lf_len[l] - (uint64_t)p->flank_len;  // lf[lf.len()-F..]  (span_locater.rs:38)
    piece_off[2 * l + 1] = rf_off[l];                                    // rf[..F]           (:39)
    flank_total = std::max<uint64_t>(flank_total, std::max(lf_off[l] + lf_len[l], rf_off[l] + rf_len[l]));
    uint32_t ml = 0;
    for (uint64_t r = locus_read_begin[l]; r < locus_read_begin[l + 1]; ++r) { read_locus[r] = (uint32_t)l; ml = std::max(ml, read_len[r]); }
    heavy_len[(size_t)l] = heavy_read_len(ml, p->flank_len);
    heavy_tlen_max = std::max(heavy_tlen_max, heavy_len[(size_t)l]);
  }
  for (int64_t r = 0; r < n_reads; ++r) {
    read_total = std::max<uint64_t>(read_total, read_off[r] + read_len[r]);
    max_read_len = std::max(max_read_len, read_len[r]);
  }
  int rc;
  const uint8_t *d_flank = nullptr, *d_reads = nullptr;
  const uint64_t *d_piece = nullptr, *d_roff = nullptr;
  const uint32_t *d_rlen = nullptr, *d_rloc = nullptr, *d_heavy = nullptr;
  if ((rc = dev_in(c, S_FS_HEAVY, heavy_len.data(), heavy_len.size(), &d_heavy)) ||
      (rc = dev_in(c, S_FS_FLANK, flank_blob, (size_t)flank_total, &d_flank)) ||
      (rc = dev_in(c, S_FS_READS, read_blob, (size_t)read_total, &d_reads)) ||
      (rc = dev_in(c, S_FS_JOBS, piece_off.data(), piece_off.size(), &d_piece)) ||
      (rc = dev_in(c, S_FS_LIST, read_off, (size_t)n_reads, &d_roff)) ||
      (rc = dev_in(c, S_FS_OUT0, read_len, (size_t)n_reads, &d_rlen)) ||
      (rc = dev_in(c, S_FS_OUT1, read_locus.data(), (size_t)n_reads, &d_rloc)))
    return rc;
  DevOut<int32_t> o_s, o_e; DevOut<uint8_t> o_l, o_r;
  if ((rc = o_s.init(c, S_LOCUS_0, span_start, (size_t)n_reads)) || (rc = o_e.init(c, S_LOCUS_1, span_end, (size_t)n_reads)) ||
      (rc = o_l.init(c, S_FS_HIT0, lf_hit, (size_t)n_reads)) || (rc = o_r.init(c, S_FS_HIT1, rf_hit, (size_t)n_reads)))
    return rc;
  if ((rc = find_spans_device(c, *p, n_loci, n_reads, d_flank, d_piece, d_reads, d_roff, d_rlen, d_rloc, max_read_len, o_s.dev,
                              o_e.dev, o_l.dev, o_r.dev, d_heavy, heavy_tlen_max > 0 ? heavy_tlen_max - 1 : 0)))
    return rc;
  if ((rc = o_s.finish(c)) || (rc = o_e.finish(c)) || (rc = o_l.finish(c)) || (rc = o_r.finish(c))) return rc;
  unsigned long long cells[2] = {0, 0}, fcells[2] = {0, 0};  // total, first launch; pre-filter: offsets, alignments kept
  if (c->last_wfa_cells_dev) { const int d2h_rc = trgt::d2h(c, cells, c->last_wfa_cells_dev, 16, c->stream); if (d2h_rc) return d2h_rc; }
  if (c->last_filter_cells_dev) { const int d2h_rc = trgt::d2h(c, fcells, c->last_filter_cells_dev, 16, c->stream); if (d2h_rc) return d2h_rc; }
  TRGT_HIP_TRY(c, trgt::stream_wait(c, c->stream));
  if (c->timing) {
    c->k_cells[TRGT_K_WFA_FLANK] += (int64_t)cells[1]; c->k_cells[TRGT_K_WFA_FLANK_REST] += (int64_t)(cells[0] - cells[1]);
    c->k_cells[TRGT_K_WFA_FILTER] += (int64_t)fcells[0];
  }
  return TRGT_OK;
}
