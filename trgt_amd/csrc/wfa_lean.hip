// trgt_amd/csrc/wfa_lean.hip -- register-resident BiWFA for the small end-to-end alignments of the locus path.
//
// What it replaces: the per-alignment WFA2-lib calls of PacificBiosciences/trgt v3.0.0 behind
//   utils::align            src/utils/align.rs:14-28          BiWFA, gap-affine (2,5,1), default heuristic  (THREAD_WFA_CONSENSUS,
//                                                              src/commands/genotype.rs:82-86) -> run-length CIGAR
//   get_dist / get_dist_matrix  src/trgt/genotype/genotype_cluster.rs:236-286   score-only BiWFA, edit distance, default heuristic
//                                                              (THREAD_WFA_ED, genotype.rs:88-92)
// i.e. a read of an allele against the central read / the consensus of its cluster, or two reads of one locus against each other:
// near-identical sequences of tens to hundreds of bases whose penalties stay in the low tens.
//
// The generic engine (wfa_engine.hpp / wfa.hip) keeps wavefronts in an HBM arena behind descriptors that one thread maintains: a
// score level costs it 15-26 k cycles whatever the width of the wavefront.  Here ONE WAVE owns an alignment and every
// wavefront of the last max(x, o + e) + 1 levels is a REGISTER: lane l holds diagonal k = kbase + l (64 diagonals around 0 and
// tlen - plen; an alignment that needs more is handed to the generic kernel through the retry list, which redoes it from scratch, so
// results do not depend on where an alignment ran).  The neighbours k - 1 / k + 1 of the recurrences are DPP wave shifts, trimming
// and termination are ballots, the forward and the reverse front of the breakpoint search are two register sets that meet through
// one lane permutation (k_reverse = tlen - plen - k_forward is a reversal of the lanes).  LDS holds only what the back-trace of a
// base alignment needs: 16-bit offsets of the computed range of every level, bump-allocated, and the run-length operations.
//
// Semantics are those of oracle/wfa.cpp (the restatement of WFA2-lib that the KATs pin): recurrences,
// trimming, the null-step bookkeeping, wfadaptive(10, 50, 1) with its equate step, termination, back-trace priorities, the two
// phases of the breakpoint search with the extra levels of the second, the bialign_min_length / bialign_min_score base cases and
// the component hand-over (begin / end in M, I or D) between the halves of a split.  Anything this kernel does not finish with
// "completed" -- including alignments it could finish but whose status would be "unattainable" -- goes to the retry list.
#include "wfa_host.hpp"

namespace trgt {
namespace lean {

constexpr int NUL = INT32_MIN / 2;  // WAVEFRONT_OFFSET_NULL
enum { CM = 0, CI = 1, CD = 3 };    // component codes of the generic engine (M, I1, D1)
enum { ST_OK = 0, ST_END_REACHED = 1, ST_END_UNREACHABLE = 2, ST_NOFIT = 3 };
// why an alignment went on to the generic kernel (developer statistics, TRGT_WFA_DEBUG)
enum { WHY_LEN = 0, WHY_WINDOW = 1, WHY_RANGE = 2, WHY_HIST_LEVELS = 3, WHY_HIST_CELLS = 4, WHY_RLE = 5, WHY_STACK = 6, WHY_STATUS = 7 };
constexpr int HIST_CELLS = 2816;   // 16-bit offsets of a base alignment's history
constexpr int HIST_LEVELS = 96;
constexpr int RLE_CAP = 80;
constexpr int STACK = 12;
constexpr int MAX_LEN = 32000;      // offsets + 1 are stored in 16 bits (insertion offsets may pass tlen by the score)

__device__ __forceinline__ int from_below(int x) { return __builtin_amdgcn_update_dpp(NUL, x, 0x138 /* wave_shr:1 */, 0xF, 0xF, false); }  // lane l <- lane l - 1 (diagonal k - 1)
__device__ __forceinline__ int from_above(int x) { return __builtin_amdgcn_update_dpp(NUL, x, 0x130 /* wave_shl:1 */, 0xF, 0xF, false); }  // lane l <- lane l + 1 (diagonal k + 1)
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
// Behind a branch on the lane number: the two paths meet HERE, before any scalar value is merged.  (Without it the optimiser may send
// the paths of `if (lane == 0) {...} continue;` to the loop header separately, which makes every value carried around that loop
// "divergent" for the register allocator: ranges and scores in VGPRs, uniform branches through exec masks.)
#define JOIN() __builtin_amdgcn_wave_barrier()

struct Range { int lo, hi; };  // lo > hi: ->null
__device__ __forceinline__ bool is_null(const Range& r) { return r.lo > r.hi; }
__device__ __forceinline__ Range canon(const Range& r) { Range o; const bool n = r.lo > r.hi; o.lo = n ? 1 : r.lo; o.hi = n ? -1 : r.hi; return o; }  // wavefront_compute_get_*wavefront: ->null reads as (1, -1)

// One unidirectional aligner: the live wavefronts.  [0] = the current level s, [i] = level s - i; lanes outside a wavefront's range hold NUL.
template <int NL, int NC> struct Front {
  int M[NL]; Range rM[NL];
  int I[NC > 1 ? NL : 1]; Range rI[NC > 1 ? NL : 1];
  int D[NC > 1 ? NL : 1]; Range rD[NC > 1 ? NL : 1];
  int s, num_null, steps_wait, m_exists, status, end_score;
};

struct Seqs { const uint8_t* p; const uint8_t* t; int plen, tlen, rev, kbase, ak; };

struct Breakpoint { int score, score_f, score_r, k_f, off_f, comp; };

struct Shared {
  uint16_t hist[HIST_CELLS];
  uint32_t hdesc[HIST_LEVELS];  // base | (lo - kbase) << 16 | width << 24 (width 0: nothing allocated at this level)
  uint32_t rle_tmp[RLE_CAP], rle_out[RLE_CAP];
  int stack[STACK][8];
};

__device__ __forceinline__ uint8_t seq_at(const uint8_t* p, int len, int rev, int i) { return p[rev ? len - 1 - i : i]; }

// bases that match from (v, h) on, looking at eight of them at most (fewer at the end of a sequence)
__device__ __forceinline__ int match8(const Seqs& q, int v, int h) {
  const int rem = min(q.plen - v, q.tlen - h);
  if (rem >= 8) {
    uint64_t a, b;
    if (!q.rev) { __builtin_memcpy(&a, q.p + v, 8); __builtin_memcpy(&b, q.t + h, 8); }
    else { __builtin_memcpy(&a, q.p + (q.plen - v - 8), 8); __builtin_memcpy(&b, q.t + (q.tlen - h - 8), 8); }
    const uint64_t x = a ^ b;
    if (!x) return 8;
    return (!q.rev ? __builtin_ctzll(x) : __builtin_clzll(x)) >> 3;
  }
  int n = 0;
  while (n < rem && seq_at(q.p, q.plen, q.rev, v + n) == seq_at(q.t, q.tlen, q.rev, h + n)) ++n;
  return n;
}

// wavefront_extend_matches_packed: every lane eight bases of its own cell first; a cell that matched all eight is finished by the whole
// wave, 512 bases per round trip (the idea of wfa_engine.hpp::extend_wave: these are alignments of near-identical sequences).
__device__ __forceinline__ int extend_cells(const Seqs& q, int k, int off, bool on) {
  int v = on ? off - k : 0, h = on ? off : 0;
  bool going = false;
  if (on) { const int n = match8(q, v, h); v += n; h += n; going = n == 8; }
  unsigned long long m = __ballot(going);
  const int lane = lane_id();
  while (m) {
    const int j = (int)__builtin_ctzll(m);
    m &= m - 1ull;
    int vj = __builtin_amdgcn_readlane(v, j), hj = __builtin_amdgcn_readlane(h, j);
    for (;;) {
      const int n = match8(q, vj + 8 * lane, hj + 8 * lane);  // (beyond either sequence: nothing matches)
      const unsigned long long stop = __ballot(n < 8);
      if (stop) {
        const int js = (int)__builtin_ctzll(stop);
        const int ext = 8 * js + __builtin_amdgcn_readlane(n, js);
        vj += ext; hj += ext;
        break;
      }
      vj += 512; hj += 512;
    }
    if (lane == j) { v = vj; h = hj; }
  }
  return h;
}

__device__ __forceinline__ bool in_bounds(int off, int k, int plen, int tlen) { return (uint32_t)off <= (uint32_t)tlen && (uint32_t)(off - k) <= (uint32_t)plen; }

// wavefront_compute_trim_ends on a register: range of the in-bounds cells inside [lo, hi] (nothing in bounds: (lo, lo - 1), as the
// two loops of the reference leave it), lanes outside the result are NUL afterwards
__device__ __forceinline__ Range trim(int& val, int k, const Range& c, const Seqs& q) {
  const bool valid = k >= c.lo && k <= c.hi && in_bounds(val, k, q.plen, q.tlen);
  const unsigned long long m = __ballot(valid);
  Range r;
  if (m) { r.lo = q.kbase + (int)__builtin_ctzll(m); r.hi = q.kbase + 63 - (int)__builtin_clzll(m); }
  else { r.lo = c.lo; r.hi = c.lo - 1; }
  if (k < r.lo || k > r.hi) val = NUL;
  return r;
}

// wave-wide minimum / maximum without LDS traffic: inclusive scan inside the rows of 16 lanes (row_shr 1, 2, 4, 8), then the row totals
// across the rows (row_bcast:15, row_bcast:31); lane 63 holds the result
template <bool MAX> __device__ __forceinline__ int wave_reduce(int v) {
  constexpr int ident = MAX ? INT32_MIN : INT32_MAX;
  auto op = [](int a, int b) { return MAX ? max(a, b) : min(a, b); };
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x111 /* row_shr:1 */, 0xF, 0xF, false));
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x112 /* row_shr:2 */, 0xF, 0xF, false));
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x114 /* row_shr:4 */, 0xF, 0xF, false));
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x118 /* row_shr:8 */, 0xF, 0xF, false));
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x142 /* row_bcast:15 */, 0xA, 0xF, false));
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x143 /* row_bcast:31 */, 0xC, 0xF, false));
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_min(int v) { return wave_reduce<false>(v); }
__device__ __forceinline__ int wave_max(int v) { return wave_reduce<true>(v); }

struct Heur { int on, min_len, max_dist, steps; };

template <int NL, int NC>
__device__ __forceinline__ void rotate(Front<NL, NC>& f) {
#pragma unroll
  for (int i = NL - 1; i > 0; --i) {
    f.M[i] = f.M[i - 1]; f.rM[i] = f.rM[i - 1];
    if constexpr (NC > 1) { f.I[i] = f.I[i - 1]; f.rI[i] = f.rI[i - 1]; f.D[i] = f.D[i - 1]; f.rD[i] = f.rD[i - 1]; }
  }
}

// wavefront_unialign_init (end-to-end): one cell at diagonal 0 in the begin component
template <int NL, int NC>
__device__ __forceinline__ void front_init(Front<NL, NC>& f, const Seqs& q, int cb, const Heur& hp, unsigned long long& cells) {
  const int k = q.kbase + lane_id();
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    f.M[i] = NUL; f.rM[i] = Range{1, -1};
    if constexpr (NC > 1) { f.I[i] = NUL; f.rI[i] = Range{1, -1}; f.D[i] = NUL; f.rD[i] = Range{1, -1}; }
  }
  const int cell = k == 0 ? 0 : NUL;
  if (NC == 1 || cb == CM) { f.M[0] = cell; f.rM[0] = Range{0, 0}; }
  if constexpr (NC > 1) {
    if (cb == CI) { f.I[0] = cell; f.rI[0] = Range{0, 0}; }
    else if (cb == CD) { f.D[0] = cell; f.rD[0] = Range{0, 0}; }
  }
  f.s = 0; f.num_null = 0; f.steps_wait = hp.steps; f.m_exists = cb == CM; f.status = ST_OK; f.end_score = -1;
  cells += 1;
}

// wavefront_compute_{edit,affine} for level f.s + 1 (gap-affine with x = 2, o = 5, e = 1: M of levels s - 2 and s - 6, I / D of s - 1).
// Returns false when the computed range leaves the 64 diagonals of the wave.
template <int METRIC, int NL, int NC>
__device__ __forceinline__ bool front_compute(Front<NL, NC>& f, const Seqs& q, unsigned long long& cells) {
  const int k = q.kbase + lane_id();
  const int wlo = q.kbase, whi = q.kbase + 63;
  f.s += 1;
  if constexpr (METRIC == 1) {  // wavefront_compute_edit: the previous wavefront as it is (even ->null)
    Range prev; prev.lo = f.m_exists ? f.rM[0].lo : 1; prev.hi = f.m_exists ? f.rM[0].hi : -1;
    const Range c{prev.lo - 1, prev.hi + 1};
    if (c.lo <= c.hi && (c.lo < wlo || c.hi > whi)) return false;
    const int src = f.M[0];
    const int ins = from_below(src), del = from_above(src);
    int mx = max(del, max(ins, src) + 1);
    if (k < c.lo || k > c.hi || !in_bounds(mx, k, q.plen, q.tlen)) mx = NUL;
    cells += (unsigned long long)max(0, c.hi - c.lo + 1);
    rotate(f);
    f.M[0] = mx; f.rM[0] = trim(f.M[0], k, c, q);
    f.m_exists = 1;
    if (is_null(f.rM[0])) f.num_null = INT32_MAX;
    return true;
  } else {
    constexpr int X = 2, OE = 6, E = 1;
    const Range z{1, -1};
    // (field by field: a conditional between two Range objects is a choice between two addresses, which keeps the whole Front in scratch)
    const Range r_mis = canon(f.rM[X - 1]), r_open = canon(f.rM[OE - 1]), r_i = canon(f.rI[E - 1]), r_d = canon(f.rD[E - 1]);
    const bool n_mis = is_null(r_mis), n_open = is_null(r_open), n_i = is_null(r_i), n_d = is_null(r_d);
    if (n_mis && n_open && n_i && n_d) {  // wavefront_compute_allocate_output_null
      rotate(f);
      f.M[0] = NUL; f.I[0] = NUL; f.D[0] = NUL; f.rM[0] = z; f.rI[0] = z; f.rD[0] = z;
      f.num_null += 1; f.m_exists = 0;
      return true;
    }
    f.num_null = 0;
    Range c{r_mis.lo, r_mis.hi};  // wavefront_compute_limits_input (a null wavefront counts as lo = 1, hi = -1)
    c.lo = min(c.lo, r_open.lo - 1); c.hi = max(c.hi, r_open.hi + 1);
    c.lo = min(c.lo, r_i.lo + 1); c.hi = max(c.hi, r_i.hi + 1);
    c.lo = min(c.lo, r_d.lo - 1); c.hi = max(c.hi, r_d.hi - 1);
    if (c.lo <= c.hi && (c.lo < wlo || c.hi > whi)) return false;
    const int m_mis = f.M[X - 1], m_open = f.M[OE - 1], i_ext = f.I[E - 1], d_ext = f.D[E - 1];
    int ins = max(from_below(m_open), from_below(i_ext)) + 1;
    int del = max(from_above(m_open), from_above(d_ext));
    const int mis = m_mis + 1;
    int mx = max(del, max(mis, ins));
    const bool act = k >= c.lo && k <= c.hi;
    if (!act || !in_bounds(mx, k, q.plen, q.tlen)) mx = NUL;
    if (!act) { ins = NUL; del = NUL; }
    cells += 3ull * (unsigned long long)max(0, c.hi - c.lo + 1);
    rotate(f);
    f.M[0] = mx; f.rM[0] = trim(f.M[0], k, c, q);
    const bool has_i = !n_open || !n_i, has_d = !n_open || !n_d;
    if (has_i) { f.I[0] = ins; f.rI[0] = trim(f.I[0], k, c, q); } else { f.I[0] = NUL; f.rI[0] = z; }
    if (has_d) { f.D[0] = del; f.rD[0] = trim(f.D[0], k, c, q); } else { f.D[0] = NUL; f.rD[0] = z; }
    f.m_exists = 1;
    return true;
  }
}

// wavefront_heuristic_cufoff (wfadaptive) on the current level
template <int METRIC, int NL, int NC>
__device__ __forceinline__ void front_heuristic(Front<NL, NC>& f, const Seqs& q, const Heur& hp) {
  if (!f.m_exists || is_null(f.rM[0])) return;
  const int k = q.kbase + lane_id();
  f.steps_wait -= 1;
  if (f.steps_wait <= 0) {
    const Range m = f.rM[0];
    if (m.hi - m.lo + 1 >= hp.min_len) {
      const int off = f.M[0];
      const bool in = k >= m.lo && k <= m.hi;
      const int d = (!in || off < 0) ? -NUL : max(q.plen - (off - k), q.tlen - off);  // wf_distance_end2end
      const int min_d = min(max(q.plen, q.tlen), wave_min(d));
      const bool keep = in && d - min_d <= hp.max_dist;
      // wf_heuristic_wfadaptive_reduce (the target diagonal tlen - plen is preserved)
      const int top_limit = min(q.ak, m.hi);
      int lo_red = m.lo;
      if (top_limit > m.lo) {
        const unsigned long long c = __ballot(keep && k < top_limit);
        lo_red = c ? q.kbase + (int)__builtin_ctzll(c) : top_limit;
      }
      const int bottom_limit = max(q.ak, lo_red);
      int hi_red = m.hi;
      if (m.hi > bottom_limit) {
        const unsigned long long c = __ballot(keep && k > bottom_limit);
        hi_red = c ? q.kbase + 63 - (int)__builtin_clzll(c) : bottom_limit;
      }
      f.rM[0] = Range{lo_red, hi_red};
      if (k < lo_red || k > hi_red) f.M[0] = NUL;
      f.steps_wait = hp.steps;
    }
  }
  if constexpr (NC > 1) {  // wavefront_heuristic_equate (a ->null wavefront stays one)
    const Range m = f.rM[0];
    if (!is_null(f.rI[0])) { f.rI[0].lo = max(f.rI[0].lo, m.lo); f.rI[0].hi = min(f.rI[0].hi, m.hi); if (k < f.rI[0].lo || k > f.rI[0].hi) f.I[0] = NUL; }
    if (!is_null(f.rD[0])) { f.rD[0].lo = max(f.rD[0].lo, m.lo); f.rD[0].hi = min(f.rD[0].hi, m.hi); if (k < f.rD[0].lo || k > f.rD[0].hi) f.D[0] = NUL; }
  }
}

// wavefront_extend_end2end(_max) on the current level.  Returns 1 when the alignment is done (f.status says how).
template <int METRIC, int NL, int NC>
__device__ __forceinline__ int front_extend(Front<NL, NC>& f, const Seqs& q, int ce, const Heur& hp, bool act_on_end, bool want_ak, int& max_ak) {
  max_ak = 0;
  const bool m_null = is_null(f.rM[0]);
  if (!f.m_exists || m_null) {
    if (!f.m_exists || METRIC <= 1) {
      if (f.num_null > (METRIC == 1 ? 2 : 7)) { f.status = ST_END_UNREACHABLE; f.end_score = f.s; return 1; }
    }
    if (!f.m_exists) return 0;
  }
  const int k = q.kbase + lane_id();
  const int off0 = f.M[0];
  const bool on = off0 >= 0;
  const int ext = extend_cells(q, k, off0, on);
  if (on) f.M[0] = ext;
  // wavefront_termination_end2end: the end component's cell on diagonal tlen - plen
  int endv = NUL;
  {
    int src = f.M[0];
    if constexpr (NC > 1) { if (ce == CI) src = f.I[0]; else if (ce == CD) src = f.D[0]; }
    endv = __builtin_amdgcn_readlane(src, q.ak - q.kbase);
  }
  if (endv >= q.tlen && act_on_end) { f.status = ST_END_REACHED; f.end_score = f.s; return 1; }
  if (want_ak) max_ak = wave_max(on ? 2 * ext - k : 0);
  if (hp.on) front_heuristic<METRIC>(f, q, hp);
  return 0;
}

// ---------------------------------------------------------------- history + back-trace of a base alignment (LDS)
template <int NL, int NC>
__device__ __forceinline__ bool hist_store(Shared& S, const Front<NL, NC>& f, const Seqs& q, int& bump, int& why) {
  const int s = f.s;
  if (s >= HIST_LEVELS) { why = WHY_HIST_LEVELS; return false; }
  // the computed range is not kept: the union of the three final ranges holds every cell the back-trace may read
  int lo = INT32_MAX, hi = INT32_MIN;
  if (!is_null(f.rM[0])) { lo = min(lo, f.rM[0].lo); hi = max(hi, f.rM[0].hi); }
  if constexpr (NC > 1) {
    if (!is_null(f.rI[0])) { lo = min(lo, f.rI[0].lo); hi = max(hi, f.rI[0].hi); }
    if (!is_null(f.rD[0])) { lo = min(lo, f.rD[0].lo); hi = max(hi, f.rD[0].hi); }
  }
  const int lane = lane_id();
  if (lo > hi) { if (lane == 0) S.hdesc[s] = 0; JOIN(); return true; }
  const int w = hi - lo + 1;
  if (bump + NC * w > HIST_CELLS) { why = WHY_HIST_CELLS; return false; }
  const int k = q.kbase + lane;
  if (k >= lo && k <= hi) {
    const int i = bump + (k - lo);
    S.hist[i] = (uint16_t)max(f.M[0] + 1, 0);
    if constexpr (NC > 1) { S.hist[i + w] = (uint16_t)max(f.I[0] + 1, 0); S.hist[i + 2 * w] = (uint16_t)max(f.D[0] + 1, 0); }
  }
  JOIN();
  if (lane == 0) S.hdesc[s] = (uint32_t)bump | ((uint32_t)(lo - q.kbase) << 16) | ((uint32_t)w << 24);
  JOIN();
  bump += NC * w;
  return true;
}

__device__ __forceinline__ void rle_push(uint32_t* buf, int& n, uint32_t code, int len, bool& overflow) {
  if (len <= 0) return;
  // (LDS reads are uniform here -- every lane walks the same path -- but the compiler cannot know: readfirstlane keeps what is derived
  //  from them in scalar registers and the control flow of the kernel uniform)
  if (n > 0) { const uint32_t last = (uint32_t)uni((int)buf[n - 1]); if ((last & 0xF) == code) { buf[n - 1] = last + ((uint32_t)len << 4); return; } }
  if (n < RLE_CAP) buf[n++] = ((uint32_t)len << 4) | code; else overflow = true;
}

// (offset + add) << 4 | type of the history cell (component plane c, level s, diagonal k); NUL when there is none
template <int NC>
__device__ __forceinline__ long long bt_cand(const Shared& S, const Seqs& q, int plane, int s, int k, int add, int type) {
  if (s < 0) return (long long)NUL;
  const uint32_t d = (uint32_t)uni((int)S.hdesc[s]);
  const int w = (int)(d >> 24), lo = q.kbase + (int)((d >> 16) & 0xFF);
  if (w == 0 || k < lo || k >= lo + w) return (long long)NUL;
  const int raw = uni((int)S.hist[(d & 0xFFFF) + (uint32_t)(plane * w + (k - lo))]);
  if (raw == 0) return (long long)NUL;
  return (((long long)(raw - 1 + add)) << 4) | type;
}

// wavefront_backtrace_{linear,affine}: every lane walks the same path (uniform values, LDS broadcasts); the
// operations land in S.rle_tmp in reverse order
template <int METRIC, int NC>
__device__ __forceinline__ void backtrace(Shared& S, const Seqs& q, int ce, int end_score, int& ntmp, bool& overflow) {
  constexpr int X = METRIC == 1 ? 1 : 2, O = 5, E = 1;
  const int plen = q.plen, tlen = q.tlen;
  int mt = ce, s = end_score, k = q.ak, off = tlen;
  int h = off, v = off - k;
  ntmp = 0;
  if (ce == CM) { rle_push(S.rle_tmp, ntmp, 2u, plen - v, overflow); rle_push(S.rle_tmp, ntmp, 1u, tlen - h, overflow); }
  while (v > 0 && h > 0 && s > 0) {
    long long best = (long long)NUL;
    if (METRIC == 1) {
      best = max(best, bt_cand<NC>(S, q, 0, s - 1, k, +1, 9));
      best = max(best, bt_cand<NC>(S, q, 0, s - 1, k - 1, +1, 1));
      best = max(best, bt_cand<NC>(S, q, 0, s - 1, k + 1, 0, 5));
    } else {
      if (mt == CM) best = max(best, bt_cand<NC>(S, q, 0, s - X, k, +1, 9));
      if (mt == CM || mt == CD) { best = max(best, bt_cand<NC>(S, q, 2, s - E, k + 1, 0, 6)); best = max(best, bt_cand<NC>(S, q, 0, s - O - E, k + 1, 0, 5)); }
      if (mt == CM || mt == CI) { best = max(best, bt_cand<NC>(S, q, 1, s - E, k - 1, +1, 2)); best = max(best, bt_cand<NC>(S, q, 0, s - O - E, k - 1, +1, 1)); }
    }
    if (best < 0) break;
    const int best_off = (int)(best >> 4), type = (int)(best & 0xF);
    if (mt == CM) {
      rle_push(S.rle_tmp, ntmp, 7u, off - best_off, overflow);
      off = best_off; h = off; v = off - k;
      if (v <= 0 || h <= 0) break;
    }
    switch (type) {
      case 9: rle_push(S.rle_tmp, ntmp, 8u, 1, overflow); s -= X; mt = CM; --off; break;
      case 1: rle_push(S.rle_tmp, ntmp, 1u, 1, overflow); s -= METRIC == 1 ? 1 : (O + E); mt = CM; --k; --off; break;
      case 2: rle_push(S.rle_tmp, ntmp, 1u, 1, overflow); s -= E; mt = CI; --k; --off; break;
      case 5: rle_push(S.rle_tmp, ntmp, 2u, 1, overflow); s -= METRIC == 1 ? 1 : (O + E); mt = CM; ++k; break;
      default: rle_push(S.rle_tmp, ntmp, 2u, 1, overflow); s -= E; mt = CD; ++k; break;
    }
    h = off; v = off - k;
  }
  if (mt == CM && v > 0 && h > 0) {
    const int n = min(v, h);
    rle_push(S.rle_tmp, ntmp, 7u, n, overflow);
    v -= n; h -= n;
  }
  rle_push(S.rle_tmp, ntmp, 2u, v, overflow);
  rle_push(S.rle_tmp, ntmp, 1u, h, overflow);
}

// ---------------------------------------------------------------- breakpoint detection
// wavefront_bialign_breakpoint_{indel2indel,m2m}: w0 = a wavefront of the front that just advanced (level s0), w1 = one of the other front
// (level s1); the other front's diagonal k1 = ak - k0 sits in lane c_rev - lane.
__device__ __forceinline__ void bp_check(int w0, const Range& r0, int w1, const Range& r1, const Seqs& q, bool fwd, int s0, int s1, int comp, int gap_open,
                                         Breakpoint& bp) {
  const int ak = q.ak;
  const int lo1 = ak - r1.hi, hi1 = ak - r1.lo;
  if (hi1 < r0.lo || r0.hi < lo1) return;
  if (!(s0 + s1 - gap_open < bp.score)) return;
  const int lane = lane_id();
  const int k0 = q.kbase + lane, k1 = ak - k0;
  const int src_lane = k1 - q.kbase;
  int h1 = __shfl(w1, src_lane & 63);
  if (src_lane < 0 || src_lane > 63) h1 = NUL;
  const int h0 = w0;
  const int min_hi = min(r0.hi, hi1), max_lo = max(r0.lo, lo1);
  bool ok = k0 >= max_lo && k0 <= min_hi && h0 + h1 >= q.tlen;
  if (ok) { const int hh = fwd ? h0 : h1, kk = fwd ? k0 : k1; ok = !((hh - kk) > q.plen || hh > q.tlen); }
  const unsigned long long m = __ballot(ok);
  if (!m) return;
  const int j = (int)__builtin_ctzll(m);
  const int kf0 = q.kbase + j, kf1 = ak - kf0;
  const int hf0 = __builtin_amdgcn_readlane(h0, j), hf1 = __builtin_amdgcn_readlane(h1, j);
  if (fwd) { bp.score_f = s0; bp.score_r = s1; bp.k_f = kf0; bp.off_f = hf0; }
  else { bp.score_f = s1; bp.score_r = s0; bp.k_f = kf1; bp.off_f = hf1; }
  bp.score = s0 + s1 - gap_open; bp.comp = comp;
}

// wavefront_bialign_overlap: a0 just advanced to s0; a1's levels s1, s1 - 1, ... are a1.X[0], a1.X[1], ...
template <int METRIC, int NL, int NC>
__device__ __forceinline__ void overlap(const Front<NL, NC>& a0, const Front<NL, NC>& a1, const Seqs& q, int s0, int s1, bool fwd, Breakpoint& bp) {
  if (!a0.m_exists) return;
  constexpr int O = 5;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int si = s1 - i;
    if (si >= 0) {  // (no break: the loop must unroll, the wavefronts are registers)
      if constexpr (NC > 1) {
        if (s0 + si - O < bp.score) {
          bp_check(a0.D[0], a0.rD[0], a1.D[i], a1.rD[i], q, fwd, s0, si, CD, O, bp);
          bp_check(a0.I[0], a0.rI[0], a1.I[i], a1.rI[i], q, fwd, s0, si, CI, O, bp);
        }
      }
      if (s0 + si < bp.score) bp_check(a0.M[0], a0.rM[0], a1.M[i], a1.rM[i], q, fwd, s0, si, CM, 0, bp);
    }
  }
}

struct Args {
  int scope_alignment, bi_min_score, bi_min_length;
  Heur heur;
  const JobDev* jobs; const uint32_t* n_jobs_dev; uint32_t n_jobs;
  const uint8_t* pat_base; const uint8_t* txt_base;
  unsigned int* counter;
  int32_t* status; int32_t* score; int32_t* n_match; uint32_t* span4; uint32_t* cigar; uint32_t* cigar_len; uint32_t* ops_len;
  unsigned long long* cells_out;
  JobDev* retry_jobs; unsigned int* retry_count; uint32_t retry_cap;
  unsigned int* retry_lost;  // jobs that found the retry list full (must stay 0: the list is sized for every job)
  unsigned int* why_hist;    // developer statistics: 8 counters, see WHY_* (NULL: none)
};

// window of 64 diagonals for sequences of these lengths: false when 0 and tlen - plen do not fit with a margin
__device__ __forceinline__ bool set_window(Seqs& q) {
  q.ak = q.tlen - q.plen;
  const int span = q.ak < 0 ? -q.ak : q.ak;
  if (span > 63 - 8) return false;
  q.kbase = min(0, q.ak) - (63 - span) / 2;
  return true;
}

// wavefront_bialign_find_breakpoint for the segment (pattern pb .. pb + pl, text tb .. tb + tl).  ST_OK: bp is set.
template <int METRIC, int NL, int NC>
__device__ __forceinline__ int find_breakpoint(const uint8_t* P, const uint8_t* T, int pl, int tl, int cb, int ce, const Heur& hp, Breakpoint& bp,
                                               unsigned long long& cells, int& why) {
  Seqs qf{P, T, pl, tl, 0, 0, 0}, qr{P, T, pl, tl, 1, 0, 0};
  if (!set_window(qf)) { why = WHY_WINDOW; return ST_NOFIT; }
  qr.kbase = qf.kbase; qr.ak = qf.ak;
  Front<NL, NC> F, R;
  front_init(F, qf, cb, hp, cells);
  front_init(R, qr, ce, hp, cells);
  const int max_ad = pl + tl - 1;
  constexpr int scope = METRIC == 1 ? 2 : 7, gap_opening = METRIC == 1 ? 0 : 5;
  bp.score = INT32_MAX;
  int fak = 0, rak = 0, mak = 0;
  bool last_forward = false;
  // pc: 0 / 1 the two extensions of level 0; 2 / 3 phase one (forward / reverse step); 5 / 6 phase two
  int pc = 0;
  for (;;) {
    bool do_f;        // which front advances now
    bool compute = true, act = true, want_ak = true;
    if (pc == 0) { do_f = true; compute = false; }
    else if (pc == 1) { do_f = false; compute = false; }
    else {
      if ((pc == 2 || pc == 3) && fak + rak >= max_ad) pc = 5;
      if (pc == 2) do_f = true;
      else if (pc == 3) do_f = false;
      else {
        act = false; want_ak = false;
        if (pc == 5 && !last_forward) pc = 6;
        if (pc == 5) {
          const int min_sr = (R.s > scope - 1) ? R.s - (scope - 1) : 0;
          if (F.s + min_sr - gap_opening >= bp.score) break;
          overlap<METRIC>(F, R, qf, F.s, R.s, true, bp);
          do_f = false;
        } else {
          const int min_sf = (F.s > scope - 1) ? F.s - (scope - 1) : 0;
          if (min_sf + R.s - gap_opening >= bp.score) break;
          overlap<METRIC>(R, F, qr, R.s, F.s, false, bp);
          do_f = true;
        }
      }
    }
    int done;
    if (do_f) {
      if (compute && !front_compute<METRIC>(F, qf, cells)) { why = WHY_RANGE; return ST_NOFIT; }
      done = front_extend<METRIC>(F, qf, CM, hp, act, want_ak, mak);
      if (done) return F.status;
    } else {
      if (compute && !front_compute<METRIC>(R, qr, cells)) { why = WHY_RANGE; return ST_NOFIT; }
      done = front_extend<METRIC>(R, qr, CM, hp, act, want_ak, mak);
      if (done) return R.status;
    }
    switch (pc) {
      case 0: fak = mak; pc = 1; break;
      case 1: rak = mak; pc = 2; break;
      case 2: if (fak < mak) fak = mak; last_forward = true; pc = 3; break;
      case 3: if (rak < mak) rak = mak; last_forward = false; pc = 2; break;
      case 5: pc = 6; break;
      default: last_forward = true; pc = 5; break;
    }
  }
  return ST_OK;
}

// wavefront_unialign of a segment (wavefront_bialign_base): ST_END_REACHED with end_score set, history in LDS when want_hist
template <int METRIC, int NL, int NC>
__device__ __forceinline__ int base_align(Shared& S, const uint8_t* P, const uint8_t* T, int pl, int tl, int cb, int ce, const Heur& hp, bool want_hist,
                                          int& end_score, Seqs& q, unsigned long long& cells, int& why) {
  q = Seqs{P, T, pl, tl, 0, 0, 0};
  if (!set_window(q)) { why = WHY_WINDOW; return ST_NOFIT; }
  Front<NL, NC> U;
  front_init(U, q, cb, hp, cells);
  int bump = 0;
  bool compute = false;
  for (;;) {
    if (compute && !front_compute<METRIC>(U, q, cells)) { why = WHY_RANGE; return ST_NOFIT; }
    compute = true;
    int unused_ak = 0;
    const int done = front_extend<METRIC>(U, q, ce, hp, true, false, unused_ak);
    if (want_hist && !hist_store(S, U, q, bump, why)) return ST_NOFIT;
    if (done) { end_score = U.end_score; return U.status; }
  }
}

template <int METRIC>
__global__ void __launch_bounds__(64) wfa_lean_kernel(const Args a) {
  constexpr int NL = METRIC == 1 ? 2 : 7, NC = METRIC == 1 ? 1 : 3;
  __shared__ Shared S;
  const int lane = lane_id();
  const uint32_t n_jobs = a.n_jobs_dev ? *a.n_jobs_dev : a.n_jobs;
  unsigned long long cells_acc = 0;
  for (;;) {
    uint32_t j = 0;
    if (lane == 0) j = atomicAdd(a.counter, 1u);
    JOIN();
    j = (uint32_t)uni((int)j);
    if (j >= n_jobs) break;
    const JobDev job = a.jobs[j];
    const int plen = (int)job.pat_len, tlen = (int)job.txt_len;
    const uint8_t* P = a.pat_base + job.pat_off;
    const uint8_t* T = a.txt_base + job.txt_off;
    const uint32_t o = job.out_index;
    const bool aln = a.scope_alignment != 0;
    // ---- identical sequences: one run of matches, penalty 0 (what the searches below return for them; see wfa.hip)
    if (plen == tlen && plen > 0) {
      uint64_t diff = 0;
      int i = 8 * lane;
      for (; i + 8 <= plen; i += 512) { uint64_t x, y; __builtin_memcpy(&x, P + i, 8); __builtin_memcpy(&y, T + i, 8); diff |= x ^ y; }
      if (i < plen && i + 8 > plen) for (int b = i; b < plen; ++b) diff |= (uint64_t)(P[b] ^ T[b]);
      if (!__ballot(diff != 0ull)) {
        if (lane == 0) {
          if (a.status) a.status[o] = TRGT_WF_COMPLETED;
          if (a.score) a.score[o] = aln ? INT32_MIN : 0;
          if (a.n_match) a.n_match[o] = aln ? plen : 0;
          if (a.span4) { a.span4[4 * o + 0] = 0; a.span4[4 * o + 1] = (uint32_t)plen; a.span4[4 * o + 2] = 0; a.span4[4 * o + 3] = (uint32_t)tlen; }
          if (a.cigar_len) a.cigar_len[o] = aln ? 1u : 0u;
          if (a.ops_len) a.ops_len[o] = aln ? (uint32_t)plen : 0u;
          if (aln && a.cigar) a.cigar[job.cigar_off] = ((uint32_t)plen << 4) | 7u;
        }
        JOIN();
        cells_acc += 2ull;
        continue;
      }
    }
    bool fit = plen <= MAX_LEN && tlen <= MAX_LEN && plen > 0 && tlen > 0;
    int status = TRGT_WF_COMPLETED, score = INT32_MIN, rle_n = 0;
    unsigned long long cells = 0;
    bool overflow = false;
    int why = fit ? WHY_STATUS : WHY_LEN;
    if (fit && !aln) {
      // ---- wavefront_bialign_compute_score
      Breakpoint bp;
      const int st = find_breakpoint<METRIC, NL, NC>(P, T, plen, tlen, CM, CM, a.heur, bp, cells, why);
      if (st == ST_END_REACHED) {
        int es = 0; Seqs q;
        const int s2 = base_align<METRIC, NL, NC>(S, P, T, plen, tlen, CM, CM, a.heur, false, es, q, cells, why);
        if (s2 == ST_END_REACHED) score = METRIC == 1 ? es : -es; else fit = false;
      } else if (st == ST_OK) score = METRIC == 1 ? bp.score : -bp.score;
      else fit = false;
    } else if (fit) {
      // ---- wavefront_bialign_alignment: the recursion as an explicit stack, left half first
      int sp = 0;
      if (lane == 0) {
        int* e = S.stack[0];
        e[0] = 0; e[1] = plen; e[2] = 0; e[3] = tlen; e[4] = CM; e[5] = CM; e[6] = max(plen, tlen) <= a.bi_min_length ? 0 : INT32_MAX; e[7] = 1;
      }
      JOIN();
      sp = 1;
      while (sp > 0 && fit) {
        __syncthreads();  // (one wave: orders lane 0's stack writes before everybody's reads)
        const int* e = S.stack[sp - 1];
        const int pb = uni(e[0]), pl = uni(e[1]), tb = uni(e[2]), tl = uni(e[3]), cb = uni(e[4]), ce = uni(e[5]), rem = uni(e[6]), top = uni(e[7]);
        sp -= 1;
        if (tl == 0) { rle_push(S.rle_out, rle_n, 2u, pl, overflow); continue; }
        if (pl == 0) { rle_push(S.rle_out, rle_n, 1u, tl, overflow); continue; }
        bool base = rem <= a.bi_min_score;
        Breakpoint bp;
        if (!base) {
          const int st = find_breakpoint<METRIC, NL, NC>(P + pb, T + tb, pl, tl, cb, ce, a.heur, bp, cells, why);
          if (st == ST_END_REACHED) base = true;
          else if (st != ST_OK) { fit = false; break; }
        }
        if (base) {
          int es = 0; Seqs q;
          const int st = base_align<METRIC, NL, NC>(S, P + pb, T + tb, pl, tl, cb, ce, a.heur, true, es, q, cells, why);
          if (st != ST_END_REACHED) { fit = false; break; }
          int nt = 0;
          __syncthreads();  // (the history written lane by lane is read by every lane)
          backtrace<METRIC, NC>(S, q, ce, es, nt, overflow);
          for (int i = nt - 1; i >= 0; --i) { const uint32_t e2 = (uint32_t)uni((int)S.rle_tmp[i]); rle_push(S.rle_out, rle_n, e2 & 0xF, (int)(e2 >> 4), overflow); }
          continue;
        }
        const int bh = bp.off_f, bv = bp.off_f - bp.k_f;
        if (top) score = METRIC == 1 ? bp.score : -bp.score;
        if (sp + 2 > STACK) { fit = false; why = WHY_STACK; break; }
        if (lane == 0) {
          int* r = S.stack[sp]; int* l = S.stack[sp + 1];
          r[0] = pb + bv; r[1] = pl - bv; r[2] = tb + bh; r[3] = tl - bh; r[4] = bp.comp; r[5] = ce; r[6] = bp.score_r; r[7] = 0;
          l[0] = pb; l[1] = bv; l[2] = tb; l[3] = bh; l[4] = cb; l[5] = bp.comp; l[6] = bp.score_f; l[7] = 0;
        }
        JOIN();
        sp += 2;
      }
      if (overflow) { fit = false; why = WHY_RLE; }
    }
    if (!fit) {  // to the generic kernel, from scratch
      if (lane == 0) {
        const uint32_t r = atomicAdd(a.retry_count, 1u);
        if (r < a.retry_cap) a.retry_jobs[r] = job; else atomicAdd(a.retry_lost, 1u);
        if (a.why_hist) atomicAdd(a.why_hist + why, 1u);
      }
      JOIN();
      continue;
    }
    // ---- per-job epilogue (as wfa_kernel_body's): status, score, count_matches, alignment span, CIGAR
    uint32_t nm = 0, total = 0;
    for (int r = 0; r < rle_n; ++r) { const uint32_t e = (uint32_t)uni((int)S.rle_out[r]); total += e >> 4; if ((e & 0xF) == 7u) nm += e >> 4; }
    if (lane == 0) {
      if (a.status) a.status[o] = status;
      if (a.score) a.score[o] = score;
      if (a.n_match) a.n_match[o] = (int32_t)nm;
      if (a.span4) { a.span4[4 * o + 0] = 0; a.span4[4 * o + 1] = (uint32_t)plen; a.span4[4 * o + 2] = 0; a.span4[4 * o + 3] = (uint32_t)tlen; }
      if (a.cigar_len) a.cigar_len[o] = (uint32_t)rle_n;
      if (a.ops_len) a.ops_len[o] = total;
    }
    JOIN();
    if (a.cigar) for (int r = lane; r < rle_n; r += 64) a.cigar[job.cigar_off + r] = S.rle_out[r];
    JOIN();
    cells_acc += cells;
  }
  if (lane == 0 && a.cells_out && cells_acc) atomicAdd(a.cells_out, cells_acc);
}

}  // namespace lean

// Enqueue the lean kernel over a job list; alignments it does not take are appended to retry_jobs (count at retry_count, both in
// device memory) for the generic kernel.  metric 1 (edit) or 3 (gap-affine 2,5,1), end-to-end, BiWFA, Heuristic::None or WFadaptive.
int wfa_lean_launch(trgt_hip_ctx* c, const trgt_wfa_params& p, const WfaLaunch& L, JobDev* retry_jobs, unsigned int* retry_count, uint32_t retry_cap,
                    unsigned int* retry_lost, unsigned int* counter, unsigned long long* cells_out, unsigned int* why_hist) {
  lean::Args a;
  std::memset(&a, 0, sizeof a);
  a.scope_alignment = p.scope != 0; a.bi_min_score = p.bialign_min_score; a.bi_min_length = p.bialign_min_length;
  a.heur.on = p.heuristic != 0; a.heur.min_len = p.h_min_wavefront_length; a.heur.max_dist = p.h_max_distance_threshold; a.heur.steps = p.h_steps_between_cutoffs;
  a.jobs = L.jobs_dev; a.n_jobs_dev = L.n_jobs_dev; a.n_jobs = (uint32_t)L.n_jobs_host;
  a.pat_base = L.pat_base; a.txt_base = L.txt_base; a.counter = counter;
  a.status = L.status; a.score = L.score; a.n_match = L.n_match; a.span4 = L.span4; a.cigar = L.cigar; a.cigar_len = L.cigar_len; a.ops_len = L.ops_len;
  a.cells_out = cells_out; a.retry_jobs = retry_jobs; a.retry_count = retry_count; a.retry_cap = retry_cap; a.retry_lost = retry_lost; a.why_hist = why_hist;
  void (*const fn)(const lean::Args) = p.metric == 1 ? lean::wfa_lean_kernel<1> : lean::wfa_lean_kernel<3>;
  int occ = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, 64, 0) != hipSuccess || occ < 1) { (void)hipGetLastError(); occ = 8; }
  const int64_t bound = L.jobs_bound > 0 ? L.jobs_bound : L.n_jobs_host;
  const int64_t grid = std::max<int64_t>(1, std::min<int64_t>(bound, (int64_t)c->num_cus * occ));
  if (c->knobs.debug) fprintf(stderr, "[wfa] lean kernel: metric %d, at most %lld jobs, occupancy %d, grid %lld\n", p.metric, (long long)bound, occ, (long long)grid);
  hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(64), 0, c->stream, a);
  const hipError_t le = hipGetLastError();
  if (le != hipSuccess) return fail(c, TRGT_ERR_HIP, "lean alignment kernel launch failed: %s", hipGetErrorString(le));
  return TRGT_OK;
}

}  // namespace trgt
