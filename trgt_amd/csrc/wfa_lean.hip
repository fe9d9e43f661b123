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
// Semantics are those of the generic engine and of the test suite's CPU restatement of WFA2-lib that the KATs pin: recurrences,
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
// why an alignment went on to the next kernel (developer statistics, TRGT_WFA_DEBUG)
enum { WHY_LEN = 0, WHY_WINDOW = 1, WHY_RANGE = 2, WHY_HIST_LEVELS = 3, WHY_HIST_CELLS = 4, WHY_RLE = 5, WHY_STACK = 6, WHY_STATUS = 7 };
constexpr int RLE_CAP = 80;
constexpr int STACK = 12;
constexpr int MAX_LEN = 32000;      // offsets + 1 are stored in 16 bits (insertion offsets may pass tlen by the score)

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
// Behind a branch on the lane number: the two paths meet HERE, before any scalar value is merged.  (Without it the optimiser may send
// the paths of `if (lane == 0) {...} continue;` to the loop header separately, which makes every value carried around that loop
// "divergent" for the register allocator: ranges and scores in VGPRs, uniform branches through exec masks.)
#define JOIN() __builtin_amdgcn_wave_barrier()

#ifdef TRGT_LEAN_PROF
// developer build (make EXTRA=-DTRGT_LEAN_PROF): shader-clock time of the phases of a level, summed over all waves
// [0] compute [1] extend [2] heuristic [3] swap [4] overlap [5] history [6] back-trace [7] job set-up + epilogue [8] levels
__device__ unsigned long long g_lean_prof[16];
__shared__ unsigned long long l_lean_prof[16];  // (per wave; flushed when the wave ends: a global atomic per mark would be what the next memory wait measures)
#define LP_DECL unsigned long long lp_t = __builtin_readcyclecounter()
#define LP_MARK(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); if (lane_id() == 0) l_lean_prof[i] += n_ - lp_t; lp_t = n_; } while (0)
#define LP_COUNT(i) do { if (lane_id() == 0) l_lean_prof[i] += 1ull; } while (0)
#else
#define LP_DECL
#define LP_MARK(i)
#define LP_COUNT(i)
#endif

struct Range { int lo, hi; };  // lo > hi: ->null
__device__ __forceinline__ bool is_null(const Range& r) { return r.lo > r.hi; }
__device__ __forceinline__ Range canon(const Range& r) { Range o; const bool n = r.lo > r.hi; o.lo = n ? 1 : r.lo; o.hi = n ? -1 : r.hi; return o; }  // wavefront_compute_get_*wavefront: ->null reads as (1, -1)

// A wavefront: NS strips of 64 diagonals, strip t of lane l = diagonal kbase + 64 t + l.  Lanes outside the wavefront's range hold NUL.
template <int NS> struct Wf { int v[NS]; };

// the wavefront moved by one diagonal: result[k] = x[k - 1] / x[k + 1] (NUL beyond the window).  Inside a strip a DPP wave shift; the lane
// at the end of a strip takes the neighbouring strip's end lane (wave rotate of that strip, kept by the shift as its `old` operand).
template <int NS> __device__ __forceinline__ Wf<NS> from_below(const Wf<NS>& x) {
  Wf<NS> r;
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    int carry = NUL;
    if (t > 0) carry = __builtin_amdgcn_update_dpp(0, x.v[t - 1], 0x13C /* wave_ror:1 */, 0xF, 0xF, false);
    r.v[t] = __builtin_amdgcn_update_dpp(carry, x.v[t], 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
  }
  return r;
}
template <int NS> __device__ __forceinline__ Wf<NS> from_above(const Wf<NS>& x) {
  Wf<NS> r;
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    int carry = NUL;
    if (t + 1 < NS) carry = __builtin_amdgcn_update_dpp(0, x.v[t + 1], 0x134 /* wave_rol:1 */, 0xF, 0xF, false);
    r.v[t] = __builtin_amdgcn_update_dpp(carry, x.v[t], 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
  }
  return r;
}
// A strip chosen by a (uniform) run-time index must be chosen among VALUES: written plainly, the optimiser turns the choice into a choice
// of addresses inside the Front object, which then stays in scratch memory as a whole (1 KB per lane, and every value read back from it
// counts as divergent).  An empty asm statement makes each candidate a register value first.
__device__ __forceinline__ int keep(int x) { asm volatile("" : "+v"(x)); return x; }
template <int NS> __device__ __forceinline__ int below_strip(const Wf<NS>& x, int t) {  // strip t of from_below(x); t a compile-time constant after unrolling
  int carry = NUL;
  if (t > 0) carry = __builtin_amdgcn_update_dpp(0, x.v[t > 0 ? t - 1 : 0], 0x13C /* wave_ror:1 */, 0xF, 0xF, false);
  return __builtin_amdgcn_update_dpp(carry, x.v[t], 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}
template <int NS> __device__ __forceinline__ int above_strip(const Wf<NS>& x, int t) {
  int carry = NUL;
  if (t + 1 < NS) carry = __builtin_amdgcn_update_dpp(0, x.v[t + 1 < NS ? t + 1 : t], 0x134 /* wave_rol:1 */, 0xF, 0xF, false);
  return __builtin_amdgcn_update_dpp(carry, x.v[t], 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
}
// the cell of window position g (uniform), to every lane
template <int NS> __device__ __forceinline__ int cell_at(const Wf<NS>& x, int g) {
  int src = keep(x.v[0]);
#pragma unroll
  for (int t = 1; t < NS; ++t) { const int c = keep(x.v[t]); if ((g >> 6) == t) src = c; }
  return __builtin_amdgcn_readlane(src, g & 63);
}

// One unidirectional aligner: the live wavefronts.  [0] = the current level s, [i] = level s - i.
template <int NL, int NC, int NS> struct Front {
  Wf<NS> M[NL]; Range rM[1];
  int HM;  // ranges of M of the levels before the current one, lo | hi << 16, level s - i in LANE i of this register (i >= 1): rotating them
           // is one DPP shift and reading one is one v_readlane, where seven scalar registers per front would be moved at every level --
           // the scalar unit, one per CU and shared by its sixteen waves, is what bounds this kernel
  // (ranges of I / D: the current level's only.  The next level's recurrences read them; the breakpoint detection, which looks at older
  //  levels, needs no range at all -- cells outside a wavefront's range are NUL and fail its test by themselves -- and every range kept is
  //  two scalar registers rotated at every level, of which there are not enough: 84 for two fronts, against ~100)
  Wf<NS> I[NC > 1 ? NL : 1]; Range rI[1];
  Wf<NS> D[NC > 1 ? NL : 1]; Range rD[1];
  int s, num_null, steps_wait, m_exists, status, end_score, rev;
};

__device__ __forceinline__ int pack(const Range& r) { return (r.lo & 0xFFFF) | (int)((uint32_t)r.hi << 16); }
__device__ __forceinline__ Range unpack(int p) { Range r; r.lo = (int)(int16_t)(p & 0xFFFF); r.hi = p >> 16; return r; }
template <int NL, int NC, int NS> __device__ __forceinline__ Range level_range(const Front<NL, NC, NS>& f, int i) {  // i: a compile-time constant after unrolling
  if (i == 0) return f.rM[0];
  return unpack(__builtin_amdgcn_readlane(f.HM, i));
}

struct Seqs { const uint8_t* p; const uint8_t* t; int plen, tlen, rev, kbase, ak; };

struct Breakpoint { int score, score_f, score_r, k_f, off_f, comp; };

// what the per-job epilogue needs of the kernel argument: read from LDS there instead of being carried in 26 scalar registers through the
// level loops (which have none to spare)
struct Outs {
  int32_t* status; int32_t* score; int32_t* n_match; uint32_t* span4; uint32_t* cigar; uint32_t* cigar_len; uint32_t* ops_len;
  JobDev* retry_jobs; unsigned int* retry_count; unsigned int* retry_lost; unsigned int* why_hist; uint32_t retry_cap, pad;
};
template <class T> __device__ __forceinline__ T* uni_ptr(T* const& p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return reinterpret_cast<T*>(((uint64_t)hi << 32) | lo);
}

template <int HC, int HL, int SC> struct Shared {
  Outs out;
  // (the staged sequences must NOT sit at the very start of the workgroup's LDS: they are read through generic pointers, and a flat
  //  instruction whose register address is a few bytes below the LDS aperture -- the compiler may split `p + (plen - v - 8)` into a base
  //  below p and a positive immediate offset -- is classified by that base alone and faults: HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION)
  int stack[STACK][8];
  uint32_t rle_tmp[RLE_CAP], rle_out[RLE_CAP];
  uint32_t hdesc[HL];     // base | (lo - kbase) << 15 | width << 23 (width 0: nothing allocated at this level)
  alignas(16) uint8_t seq[SC];  // pattern | text of the current alignment when they fit (the extension then reads LDS, ~5 x less latency than L2)
  uint16_t hist[HC];      // 16-bit offsets + 1 of a base alignment's history, bump-allocated level by level
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

// wavefront_extend_matches_packed on one strip: every lane eight bases of its own cell first; a cell that matched all eight is finished by
// the whole wave, 512 bases per round trip (the idea of wfa_engine.hpp::extend_wave: these are alignments of near-identical sequences).
__device__ __forceinline__ int extend_cells(const Seqs& q, int k, int off, bool on) {
  int v = on ? off - k : 0, h = on ? off : 0;
  bool going = false;
  if (on) { const int n = match8(q, v, h); v += n; h += n; going = n == 8; }
  unsigned long long m = __ballot(going);
  // Repeats: a diagonal a multiple of the motif length away from the alignment's own matches for tens or hundreds of bases too, and in a
  // tandem repeat there are dozens of such lanes per level (measured: the one-at-a-time finish below was 90-99 % of a level's time on the
  // cluster genotyper's and the VNTR batches).  While more than a few cells are still running every lane steps its own cell, eight bases
  // per round trip; the stragglers -- the alignment's own diagonal, mostly -- are then finished by the whole wave.
  while (__builtin_popcountll(m) > 3) {
    if (going) { const int n = match8(q, v, h); v += n; h += n; going = n == 8; }
    m = __ballot(going);
  }
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

// first / last window position whose lane votes yes, over the strips (-1: none)
template <int NS> __device__ __forceinline__ void first_last(const bool (&vote)[NS], int& first, int& last) {
  first = -1; last = -1;
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    const unsigned long long m = __ballot(vote[t]);
    if (m) { if (first < 0) first = 64 * t + (int)__builtin_ctzll(m); last = 64 * t + 63 - (int)__builtin_clzll(m); }
  }
}

// wavefront_compute_trim_ends on a wavefront: range of the in-bounds cells inside [lo, hi] (nothing in bounds: (lo, lo - 1), as the
// two loops of the reference leave it), lanes outside the result are NUL afterwards
template <int NS> __device__ __forceinline__ Range trim(Wf<NS>& w, const Range& c, const Seqs& q) {
  const int k0 = q.kbase + lane_id();
  bool valid[NS];
#pragma unroll
  for (int t = 0; t < NS; ++t) { const int k = k0 + 64 * t; valid[t] = k >= c.lo && k <= c.hi && in_bounds(w.v[t], k, q.plen, q.tlen); }
  int first = -1, last = -1;
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    if (NS > 1 && (q.kbase + 64 * t > c.hi || q.kbase + 64 * t + 63 < c.lo)) continue;  // (uniform: no cell of the level in this strip)
    const unsigned long long m = __ballot(valid[t]);
    if (m) { if (first < 0) first = 64 * t + (int)__builtin_ctzll(m); last = 64 * t + 63 - (int)__builtin_clzll(m); }
  }
  Range r;
  if (first >= 0) { r.lo = q.kbase + first; r.hi = q.kbase + last; }
  else { r.lo = c.lo; r.hi = c.lo - 1; }
#pragma unroll
  for (int t = 0; t < NS; ++t) { const int k = k0 + 64 * t; if (k < r.lo || k > r.hi) w.v[t] = NUL; }
  return r;
}
template <int NS> __device__ __forceinline__ void clip(Wf<NS>& w, const Range& r, const Seqs& q) {
  const int k0 = q.kbase + lane_id();
#pragma unroll
  for (int t = 0; t < NS; ++t) { const int k = k0 + 64 * t; if (k < r.lo || k > r.hi) w.v[t] = NUL; }
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

template <int NL, int NC, int NS>
__device__ __forceinline__ void rotate(Front<NL, NC, NS>& f) {
  f.HM = __builtin_amdgcn_update_dpp(f.HM, f.HM, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
  f.HM = lane_id() == 1 ? pack(f.rM[0]) : f.HM;
#pragma unroll
  for (int i = NL - 1; i > 0; --i) {
    f.M[i] = f.M[i - 1];
    if constexpr (NC > 1) { f.I[i] = f.I[i - 1]; f.D[i] = f.D[i - 1]; }
  }
}
template <int NS> __device__ __forceinline__ void fill(Wf<NS>& w, int v) {
#pragma unroll
  for (int t = 0; t < NS; ++t) w.v[t] = v;
}

// wavefront_unialign_init (end-to-end): one cell at diagonal 0 in the begin component
template <int NL, int NC, int NS>
__device__ __forceinline__ void front_init(Front<NL, NC, NS>& f, const Seqs& q, int cb, const Heur& hp, unsigned long long& cells) {
  const int k0 = q.kbase + lane_id();
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    fill(f.M[i], NUL);
    if constexpr (NC > 1) { fill(f.I[i], NUL); fill(f.D[i], NUL); }
  }
  if constexpr (NC > 1) { f.rI[0] = Range{1, -1}; f.rD[0] = Range{1, -1}; }
  Wf<NS> cell;
#pragma unroll
  for (int t = 0; t < NS; ++t) cell.v[t] = k0 + 64 * t == 0 ? 0 : NUL;
  f.rM[0] = Range{1, -1}; f.HM = pack(Range{1, -1});
  if (NC == 1 || cb == CM) { f.M[0] = cell; f.rM[0] = Range{0, 0}; }
  if constexpr (NC > 1) {
    if (cb == CI) { f.I[0] = cell; f.rI[0] = Range{0, 0}; }
    else if (cb == CD) { f.D[0] = cell; f.rD[0] = Range{0, 0}; }
  }
  f.s = 0; f.num_null = 0; f.steps_wait = hp.steps; f.m_exists = cb == CM; f.status = ST_OK; f.end_score = -1;
  cells += 1;
}

// wavefront_compute_{edit,affine} for level f.s + 1 (gap-affine with x = 2, o = 5, e = 1: M of levels s - 2 and s - 6, I / D of s - 1).
// Returns false when the computed range leaves the window of the wave.
template <int METRIC, int NL, int NC, int NS>
__device__ __forceinline__ bool front_compute(Front<NL, NC, NS>& f, const Seqs& q, unsigned long long& cells) {
  const int k0 = q.kbase + lane_id();
  const int wlo = q.kbase, whi = q.kbase + 64 * NS - 1;
  f.s += 1;
  if constexpr (METRIC == 1) {  // wavefront_compute_edit: the previous wavefront as it is (even ->null)
    Range prev; prev.lo = f.m_exists ? f.rM[0].lo : 1; prev.hi = f.m_exists ? f.rM[0].hi : -1;
    const Range c{prev.lo - 1, prev.hi + 1};
    if (c.lo <= c.hi && (c.lo < wlo || c.hi > whi)) return false;
    Wf<NS> mx;
#pragma unroll
    for (int t = 0; t < NS; ++t) {
      if (NS > 1 && (q.kbase + 64 * t > c.hi || q.kbase + 64 * t + 63 < c.lo)) { mx.v[t] = NUL; continue; }
      const int k = k0 + 64 * t;
      int m = max(above_strip(f.M[0], t), max(below_strip(f.M[0], t), f.M[0].v[t]) + 1);
      if (k < c.lo || k > c.hi || !in_bounds(m, k, q.plen, q.tlen)) m = NUL;
      mx.v[t] = m;
    }
    cells += (unsigned long long)max(0, c.hi - c.lo + 1);
    rotate(f);
    f.M[0] = mx; f.rM[0] = trim(f.M[0], c, q);
    f.m_exists = 1;
    if (is_null(f.rM[0])) f.num_null = INT32_MAX;
    return true;
  } else {
    constexpr int X = 2, OE = 6, E = 1;
    const Range z{1, -1};
    // (field by field: a conditional between two Range objects is a choice between two addresses, which keeps the whole Front in scratch)
    const Range r_mis = canon(level_range(f, X - 1)), r_open = canon(level_range(f, OE - 1)), r_i = canon(f.rI[E - 1]), r_d = canon(f.rD[E - 1]);
    const bool n_mis = is_null(r_mis), n_open = is_null(r_open), n_i = is_null(r_i), n_d = is_null(r_d);
    if (n_mis && n_open && n_i && n_d) {  // wavefront_compute_allocate_output_null
      rotate(f);
      fill(f.M[0], NUL); fill(f.I[0], NUL); fill(f.D[0], NUL); f.rM[0] = z; f.rI[0] = z; f.rD[0] = z;
      f.num_null += 1; f.m_exists = 0;
      return true;
    }
    f.num_null = 0;
    Range c{r_mis.lo, r_mis.hi};  // wavefront_compute_limits_input (a null wavefront counts as lo = 1, hi = -1)
    c.lo = min(c.lo, r_open.lo - 1); c.hi = max(c.hi, r_open.hi + 1);
    c.lo = min(c.lo, r_i.lo + 1); c.hi = max(c.hi, r_i.hi + 1);
    c.lo = min(c.lo, r_d.lo - 1); c.hi = max(c.hi, r_d.hi - 1);
    if (c.lo <= c.hi && (c.lo < wlo || c.hi > whi)) return false;
    Wf<NS> mx, ins, del;
#pragma unroll
    for (int t = 0; t < NS; ++t) {
      // (a strip the level does not reach -- a window of 256 diagonals is mostly wider than the wavefront -- costs three moves)
      if (NS > 1 && (q.kbase + 64 * t > c.hi || q.kbase + 64 * t + 63 < c.lo)) { mx.v[t] = NUL; ins.v[t] = NUL; del.v[t] = NUL; continue; }
      const int k = k0 + 64 * t;
      int i = max(below_strip(f.M[OE - 1], t), below_strip(f.I[E - 1], t)) + 1;
      int d = max(above_strip(f.M[OE - 1], t), above_strip(f.D[E - 1], t));
      int m = max(d, max(f.M[X - 1].v[t] + 1, i));
      const bool act = k >= c.lo && k <= c.hi;
      if (!act || !in_bounds(m, k, q.plen, q.tlen)) m = NUL;
      if (!act) { i = NUL; d = NUL; }
      mx.v[t] = m; ins.v[t] = i; del.v[t] = d;
    }
    cells += 3ull * (unsigned long long)max(0, c.hi - c.lo + 1);
    rotate(f);
    f.M[0] = mx; f.rM[0] = trim(f.M[0], c, q);
    const bool has_i = !n_open || !n_i, has_d = !n_open || !n_d;
    if (has_i) { f.I[0] = ins; f.rI[0] = trim(f.I[0], c, q); } else { fill(f.I[0], NUL); f.rI[0] = z; }
    if (has_d) { f.D[0] = del; f.rD[0] = trim(f.D[0], c, q); } else { fill(f.D[0], NUL); f.rD[0] = z; }
    f.m_exists = 1;
    return true;
  }
}

// wavefront_heuristic_cufoff (wfadaptive) on the current level
template <int METRIC, int NL, int NC, int NS>
__device__ __forceinline__ void front_heuristic(Front<NL, NC, NS>& f, const Seqs& q, const Heur& hp) {
  if (!f.m_exists || is_null(f.rM[0])) return;
  const int k0 = q.kbase + lane_id();
  f.steps_wait -= 1;
  if (f.steps_wait <= 0) {
    const Range m = f.rM[0];
    if (m.hi - m.lo + 1 >= hp.min_len) {
      int d[NS];
      int dmin = INT32_MAX;
#pragma unroll
      for (int t = 0; t < NS; ++t) {
        const int k = k0 + 64 * t, off = f.M[0].v[t];
        const bool in = k >= m.lo && k <= m.hi;
        d[t] = (!in || off < 0) ? -NUL : max(q.plen - (off - k), q.tlen - off);  // wf_distance_end2end
        dmin = min(dmin, d[t]);
      }
      const int min_d = min(max(q.plen, q.tlen), wave_min(dmin));
      // wf_heuristic_wfadaptive_reduce (the target diagonal tlen - plen is preserved)
      const int top_limit = min(q.ak, m.hi);
      bool keep_lo[NS], keep_hi[NS];
#pragma unroll
      for (int t = 0; t < NS; ++t) {
        const int k = k0 + 64 * t;
        const bool keep = k >= m.lo && k <= m.hi && d[t] - min_d <= hp.max_dist;
        keep_lo[t] = keep && k < top_limit;
        keep_hi[t] = keep;
      }
      int lo_red = m.lo;
      if (top_limit > m.lo) { int first, last; first_last<NS>(keep_lo, first, last); lo_red = first >= 0 ? q.kbase + first : top_limit; }
      const int bottom_limit = max(q.ak, lo_red);
      int hi_red = m.hi;
      if (m.hi > bottom_limit) {
#pragma unroll
        for (int t = 0; t < NS; ++t) keep_hi[t] = keep_hi[t] && k0 + 64 * t > bottom_limit;
        int first, last; first_last<NS>(keep_hi, first, last);
        hi_red = last >= 0 ? q.kbase + last : bottom_limit;
      }
      f.rM[0] = Range{lo_red, hi_red};
      clip(f.M[0], f.rM[0], q);
      f.steps_wait = hp.steps;
    }
  }
  if constexpr (NC > 1) {  // wavefront_heuristic_equate (a ->null wavefront stays one)
    const Range m = f.rM[0];
    if (!is_null(f.rI[0])) { f.rI[0].lo = max(f.rI[0].lo, m.lo); f.rI[0].hi = min(f.rI[0].hi, m.hi); clip(f.I[0], f.rI[0], q); }
    if (!is_null(f.rD[0])) { f.rD[0].lo = max(f.rD[0].lo, m.lo); f.rD[0].hi = min(f.rD[0].hi, m.hi); clip(f.D[0], f.rD[0], q); }
  }
}

// wavefront_extend_end2end(_max) on the current level.  Returns 1 when the alignment is done (f.status says how).
template <int METRIC, int NL, int NC, int NS>
__device__ __forceinline__ int front_extend(Front<NL, NC, NS>& f, const Seqs& q, int ce, const Heur& hp, bool act_on_end, bool want_ak, int& max_ak) {
  max_ak = 0;
  const bool m_null = is_null(f.rM[0]);
  if (!f.m_exists || m_null) {
    if (!f.m_exists || METRIC <= 1) {
      if (f.num_null > (METRIC == 1 ? 2 : 7)) { f.status = ST_END_UNREACHABLE; f.end_score = f.s; return 1; }
    }
    if (!f.m_exists) return 0;
  }
  const int k0 = q.kbase + lane_id();
  int ad = 0;
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    // (a strip without a live cell costs one ballot)
    const int k = k0 + 64 * t, off0 = f.M[0].v[t];
    const bool on = off0 >= 0;
    if (__ballot(on)) {
      const int ext = extend_cells(q, k, off0, on);
      if (on) { f.M[0].v[t] = ext; ad = max(ad, 2 * ext - k); }
    }
  }
  // wavefront_termination_end2end: the end component's cell on diagonal tlen - plen
  int endv;
  {
    const int g = q.ak - q.kbase;
    endv = cell_at(f.M[0], g);
    if constexpr (NC > 1) { if (ce == CI) endv = cell_at(f.I[0], g); else if (ce == CD) endv = cell_at(f.D[0], g); }
  }
  if (endv >= q.tlen && act_on_end) { f.status = ST_END_REACHED; f.end_score = f.s; return 1; }
  if (want_ak) max_ak = wave_max(ad);
  if (hp.on) front_heuristic<METRIC>(f, q, hp);
  return 0;
}

// ---------------------------------------------------------------- history + back-trace of a base alignment (LDS)
template <int HC, int HL, int SC, int NL, int NC, int NS>
__device__ __forceinline__ bool hist_store(Shared<HC, HL, SC>& S, const Front<NL, NC, NS>& f, const Seqs& q, int& bump, int& why) {
  const int s = f.s;
  if (s >= HL) { why = WHY_HIST_LEVELS; return false; }
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
  if (bump + NC * w > HC) { why = WHY_HIST_CELLS; return false; }
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    const int k = q.kbase + 64 * t + lane;
    if (k >= lo && k <= hi) {
      const int i = bump + (k - lo);
      S.hist[i] = (uint16_t)max(f.M[0].v[t] + 1, 0);
      if constexpr (NC > 1) { S.hist[i + w] = (uint16_t)max(f.I[0].v[t] + 1, 0); S.hist[i + 2 * w] = (uint16_t)max(f.D[0].v[t] + 1, 0); }
    }
    JOIN();
  }
  if (lane == 0) S.hdesc[s] = (uint32_t)bump | ((uint32_t)(lo - q.kbase) << 15) | ((uint32_t)w << 23);
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
template <int HC, int HL, int SC>
__device__ __forceinline__ long long bt_cand(const Shared<HC, HL, SC>& S, const Seqs& q, int plane, int s, int k, int add, int type) {
  if (s < 0) return (long long)NUL;
  const uint32_t d = (uint32_t)uni((int)S.hdesc[s]);
  const int w = (int)(d >> 23), lo = q.kbase + (int)((d >> 15) & 0xFF);
  if (w == 0 || k < lo || k >= lo + w) return (long long)NUL;
  const int raw = uni((int)S.hist[(d & 0x7FFF) + (uint32_t)(plane * w + (k - lo))]);
  if (raw == 0) return (long long)NUL;
  return (((long long)(raw - 1 + add)) << 4) | type;
}

// wavefront_backtrace_{linear,affine}: every lane walks the same path (uniform values, LDS broadcasts); the
// operations land in S.rle_tmp in reverse order
template <int METRIC, int HC, int HL, int SC>
__device__ __forceinline__ void backtrace(Shared<HC, HL, SC>& S, const Seqs& q, int ce, int end_score, int& ntmp, bool& overflow) {
  constexpr int X = METRIC == 1 ? 1 : 2, O = 5, E = 1;
  const int plen = q.plen, tlen = q.tlen;
  int mt = ce, s = end_score, k = q.ak, off = tlen;
  int h = off, v = off - k;
  ntmp = 0;
  if (ce == CM) { rle_push(S.rle_tmp, ntmp, 2u, plen - v, overflow); rle_push(S.rle_tmp, ntmp, 1u, tlen - h, overflow); }
  while (v > 0 && h > 0 && s > 0) {
    long long best = (long long)NUL;
    if (METRIC == 1) {
      best = max(best, bt_cand(S, q, 0, s - 1, k, +1, 9));
      best = max(best, bt_cand(S, q, 0, s - 1, k - 1, +1, 1));
      best = max(best, bt_cand(S, q, 0, s - 1, k + 1, 0, 5));
    } else {
      if (mt == CM) best = max(best, bt_cand(S, q, 0, s - X, k, +1, 9));
      if (mt == CM || mt == CD) { best = max(best, bt_cand(S, q, 2, s - E, k + 1, 0, 6)); best = max(best, bt_cand(S, q, 0, s - O - E, k + 1, 0, 5)); }
      if (mt == CM || mt == CI) { best = max(best, bt_cand(S, q, 1, s - E, k - 1, +1, 2)); best = max(best, bt_cand(S, q, 0, s - O - E, k - 1, +1, 1)); }
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

template <int NS> __device__ __forceinline__ void swap_wf(Wf<NS>& a, Wf<NS>& b) {
#pragma unroll
  for (int t = 0; t < NS; ++t) { const int x = a.v[t]; a.v[t] = b.v[t]; b.v[t] = x; }
}
// The front of a breakpoint search that is not advancing: its wavefronts in registers like the other's, its scalar state -- ranges of the
// current level, score, counters -- in the lanes of ONE register (PS_*), so that changing places costs vector moves and lane accesses,
// not scalar instructions, and the waiting front holds no scalar registers.
enum { PS_MLO = 0, PS_MHI, PS_ILO, PS_IHI, PS_DLO, PS_DHI, PS_S, PS_NULLS, PS_WAIT, PS_MEX, PS_REV };
template <int NL, int NC, int NS> struct Parked {
  Wf<NS> M[NL]; Wf<NS> I[NC > 1 ? NL : 1]; Wf<NS> D[NC > 1 ? NL : 1];
  int HM, PS;
};
__device__ __forceinline__ void swap_lane(int& ps, int& x, int lane) {
  const int t = __builtin_amdgcn_readlane(ps, lane);
  ps = lane_id() == lane ? x : ps;  // (v_writelane: no builtin in this compiler; a compare against a constant lane and a select)
  x = t;
}
template <int NL, int NC, int NS> __device__ __forceinline__ void swap_fronts(Front<NL, NC, NS>& a, Parked<NL, NC, NS>& b) {
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    swap_wf(a.M[i], b.M[i]);
    if constexpr (NC > 1) { swap_wf(a.I[i], b.I[i]); swap_wf(a.D[i], b.D[i]); }
  }
  { const int x = a.HM; a.HM = b.HM; b.HM = x; }
  swap_lane(b.PS, a.rM[0].lo, PS_MLO); swap_lane(b.PS, a.rM[0].hi, PS_MHI);
  if constexpr (NC > 1) { swap_lane(b.PS, a.rI[0].lo, PS_ILO); swap_lane(b.PS, a.rI[0].hi, PS_IHI); swap_lane(b.PS, a.rD[0].lo, PS_DLO); swap_lane(b.PS, a.rD[0].hi, PS_DHI); }
  swap_lane(b.PS, a.s, PS_S); swap_lane(b.PS, a.num_null, PS_NULLS); swap_lane(b.PS, a.steps_wait, PS_WAIT); swap_lane(b.PS, a.m_exists, PS_MEX); swap_lane(b.PS, a.rev, PS_REV);
}

// ---------------------------------------------------------------- breakpoint detection
// wavefront_bialign_breakpoint_{indel2indel,m2m}: w0 = a wavefront of the front that just advanced (level s0), w1 = one of the other front
// (level s1).  The other front's diagonal k1 = ak - k0 sits at window position crev - (window position of k0): the window reversed.
template <int NS>
__device__ __forceinline__ void bp_check(const Wf<NS>& w0, const Range& r0, const Wf<NS>& w1, const Range& r1, const Seqs& q, bool fwd, int s0, int s1, int comp,
                                         int gap_open, Breakpoint& bp) {
  const int ak = q.ak;
  const int lo1 = ak - r1.hi, hi1 = ak - r1.lo;
  if (hi1 < r0.lo || r0.hi < lo1) return;
  if (!(s0 + s1 - gap_open < bp.score)) return;
  const int lane = lane_id();
  const int min_hi = min(r0.hi, hi1), max_lo = max(r0.lo, lo1);
  const int crev = ak - 2 * q.kbase;
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    if (q.kbase + 64 * t > min_hi || q.kbase + 64 * t + 63 < max_lo) continue;  // (uniform: the strip holds no candidate diagonal)
    const int k0 = q.kbase + 64 * t + lane, k1 = ak - k0;
    const int g1 = crev - 64 * t - lane;  // window position of k1
    // the strip of w1 that holds g1: one of two neighbours, the boundary between them at lane (crev - 64 t) & 63
    const int cb = crev - 64 * t, ta = cb >> 6;
    int va = NUL, vb = NUL;
#pragma unroll
    for (int u = 0; u < NS; ++u) { const int c = keep(w1.v[u]); if (u == ta) va = c; if (u == ta - 1) vb = c; }
    const int sa = __shfl(va, g1 & 63), sb = __shfl(vb, g1 & 63);
    int h1 = (g1 >> 6) == ta ? sa : sb;
    if (g1 < 0 || g1 >= 64 * NS) h1 = NUL;
    const int h0 = w0.v[t];
    bool ok = k0 >= max_lo && k0 <= min_hi && h0 + h1 >= q.tlen;
    if (ok) { const int hh = fwd ? h0 : h1, kk = fwd ? k0 : k1; ok = !((hh - kk) > q.plen || hh > q.tlen); }
    const unsigned long long m = __ballot(ok);
    if (!m) continue;
    const int j = (int)__builtin_ctzll(m);
    const int kf0 = q.kbase + 64 * t + j, kf1 = ak - kf0;
    const int hf0 = __builtin_amdgcn_readlane(h0, j), hf1 = __builtin_amdgcn_readlane(h1, j);
    if (fwd) { bp.score_f = s0; bp.score_r = s1; bp.k_f = kf0; bp.off_f = hf0; }
    else { bp.score_f = s1; bp.score_r = s0; bp.k_f = kf1; bp.off_f = hf1; }
    bp.score = s0 + s1 - gap_open; bp.comp = comp;
    return;
  }
}

// wavefront_bialign_overlap: a0 just advanced to s0; a1's levels s1, s1 - 1, ... are a1.X[0], a1.X[1], ...
template <int METRIC, int NL, int NC, int NS>
__device__ __forceinline__ void overlap(const Front<NL, NC, NS>& a0, const Parked<NL, NC, NS>& a1, const Seqs& q, int s0, int s1, bool fwd, Breakpoint& bp) {
  if (!a0.m_exists) return;
  constexpr int O = 5;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int si = s1 - i;
    if (si >= 0) {  // (no break: the loop must unroll, the wavefronts are registers)
      if constexpr (NC > 1) {
        if (s0 + si - O < bp.score) {
          const Range whole{q.kbase, q.kbase + 64 * NS - 1};
          bp_check(a0.D[0], a0.rD[0], a1.D[i], whole, q, fwd, s0, si, CD, O, bp);
          bp_check(a0.I[0], a0.rI[0], a1.I[i], whole, q, fwd, s0, si, CI, O, bp);
        }
      }
      if (s0 + si < bp.score) bp_check(a0.M[0], a0.rM[0], a1.M[i], i == 0 ? Range{__builtin_amdgcn_readlane(a1.PS, PS_MLO), __builtin_amdgcn_readlane(a1.PS, PS_MHI)} : unpack(__builtin_amdgcn_readlane(a1.HM, i)), q, fwd, s0, si, CM, 0, bp);
    }
  }
}

struct Args {
  int scope_alignment, bi_min_score, bi_min_length, stage, chunk;
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

// window of 64 NS diagonals for sequences of these lengths: false when 0 and tlen - plen do not fit with a margin
template <int NS> __device__ __forceinline__ bool set_window(Seqs& q) {
  q.ak = q.tlen - q.plen;
  const int span = q.ak < 0 ? -q.ak : q.ak;
  constexpr int W = 64 * NS;
  if (span > W - 1 - 8) return false;
  q.kbase = min(0, q.ak) - (W - 1 - span) / 2;
  return true;
}

// One segment (pattern P .. P + pl, text T .. T + tl) in one of three modes:
//   0  wavefront_bialign_find_breakpoint: forward and reverse front in turn until they overlap (phase one: by antidiagonals) and the
//      best breakpoint is certain (phase two: extra levels); ST_OK: bp is set, ST_END_REACHED: a front reached the other end first
//   1  wavefront_unialign with the history in LDS (wavefront_bialign_base before its back-trace)      } ST_END_REACHED: end_score set
//   2  the same without history (score only)                                                           }
// ONE copy of the level code serves the forward front, the reverse front and the base alignment: the front that advances is always `A`,
// and the two fronts of a breakpoint search change places before every step (register swaps: cheaper than two more copies of the code
// in a 64-KB instruction cache that sixteen waves per CU share).
template <int METRIC, int NL, int NC, int NS, int HC, int HL, int SC>
__device__ __forceinline__ int engine(Shared<HC, HL, SC>& S, const uint8_t* P, const uint8_t* T, int pl, int tl, int cb, int ce, const Heur& hp, int mode,
                                      Breakpoint& bp, int& end_score, Seqs& q, unsigned long long& cells, int& why) {
  q = Seqs{P, T, pl, tl, 0, 0, 0};
  if (!set_window<NS>(q)) { why = WHY_WINDOW; return ST_NOFIT; }
  Front<NL, NC, NS> A;
  Parked<NL, NC, NS> B;
  front_init(A, q, ce, hp, cells); A.rev = 1;  // the reverse front first (it begins in the segment's end component) ...
  B.PS = 0; B.HM = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) { fill(B.M[i], NUL); if constexpr (NC > 1) { fill(B.I[i], NUL); fill(B.D[i], NUL); } }
  swap_fronts(A, B);                           // ... parked ...
  front_init(A, q, cb, hp, cells); A.rev = 0;  // ... then the forward one
  if (mode != 0) cells -= 1;  // (no reverse front)
  const int max_ad = pl + tl - 1;
  constexpr int scope = METRIC == 1 ? 2 : 7, gap_opening = METRIC == 1 ? 0 : 5;
  bp.score = INT32_MAX;
  int fak = 0, rak = 0, mak = 0, bump = 0;
  bool last_forward = false, a_fwd = true, first = true;
  // pc: 0 / 1 the two extensions of level 0; 2 / 3 phase one (forward / reverse step); 5 phase two (the fronts keep alternating)
  int pc = 0;
  LP_DECL;
  for (;;) {
    LP_COUNT(8);
    bool compute = !first, act = true, want_ak = mode == 0, swap_first = mode == 0 && !first;
    if (mode == 0) {
      if (pc == 1) compute = false;
      if ((pc == 2 || pc == 3) && fak + rak >= max_ad) pc = 5;
      if (pc == 5) {  // A advanced last
        act = false; want_ak = false;
        const int sb = __builtin_amdgcn_readlane(B.PS, PS_S);
        const int min_sb = (sb > scope - 1) ? sb - (scope - 1) : 0;
        if (A.s + min_sb - gap_opening >= bp.score) break;
        overlap<METRIC>(A, B, q, A.s, sb, a_fwd, bp);
      }
    }
    LP_MARK(4);
    if (swap_first) { swap_fronts(A, B); a_fwd = !a_fwd; }
    q.rev = A.rev;
    LP_MARK(3);
    if (compute && !front_compute<METRIC>(A, q, cells)) { why = WHY_RANGE; return ST_NOFIT; }
    LP_MARK(0);
    const int done = front_extend<METRIC>(A, q, mode == 0 ? (int)CM : ce, hp, act, want_ak, mak);
    LP_MARK(1);
    if (mode == 1 && !hist_store(S, A, q, bump, why)) return ST_NOFIT;
    LP_MARK(5);
    if (done) { end_score = A.end_score; q.rev = 0; return A.status; }
    first = false;
    if (mode == 0) {
      if (pc == 0) { fak = mak; pc = 1; }
      else if (pc == 1) { rak = mak; pc = 2; }
      else if (pc == 2) { if (fak < mak) fak = mak; last_forward = true; pc = 3; }
      else if (pc == 3) { if (rak < mak) rak = mak; last_forward = false; pc = 2; }
    }
  }
  (void)last_forward;
  return ST_OK;
}

// NS strips of 64 diagonals per wavefront, HC history cells and HL history levels in LDS.  The first tier (NS = 1, 7 KB of LDS, four to six
// waves per SIMD) takes nearly every alignment; the second (NS = 4: 256 diagonals, 56 KB) takes what the first hands on -- reads of two
// different alleles against each other, mostly -- with few waves resident, which is all those few alignments need.
template <int METRIC, int NS, int HC, int HL, int SC>
__global__ void __launch_bounds__(64) wfa_lean_kernel(const Args a) {
  constexpr int NL = METRIC == 1 ? 2 : 7, NC = METRIC == 1 ? 1 : 3;
  __shared__ Shared<HC, HL, SC> S;
  const int lane = lane_id();
#ifdef TRGT_LEAN_PROF
  if (lane < 16) l_lean_prof[lane] = 0;
  __syncthreads();
#endif
  if (lane == 0) {
    Outs& o = S.out;
    o.status = a.status; o.score = a.score; o.n_match = a.n_match; o.span4 = a.span4; o.cigar = a.cigar; o.cigar_len = a.cigar_len; o.ops_len = a.ops_len;
    o.retry_jobs = a.retry_jobs; o.retry_count = a.retry_count; o.retry_lost = a.retry_lost; o.why_hist = a.why_hist; o.retry_cap = a.retry_cap; o.pad = 0;
  }
  JOIN();
  __syncthreads();
  const uint32_t n_jobs = a.n_jobs_dev ? *a.n_jobs_dev : a.n_jobs;
  const uint32_t n_claims = (n_jobs + (uint32_t)a.chunk - 1u) / (uint32_t)a.chunk;
  unsigned long long cells_acc = 0;
  for (;;) {
    // Several jobs per claim: every wave of the grid bumps ONE counter, and an atomic on one address completes every ~10 ns -- one claim
    // per alignment bounded the first tier at ~0.5 ms per 57 k alignments, several times what their instructions take.  The jobs of a
    // claim are spread over the list (claim c takes c, c + n_claims, c + 2 n_claims, ...): neighbours in the list belong to one locus and
    // cost alike, and a wave that drew eight expensive ones in a row would be the tail of the launch.
    uint32_t c0 = 0;
    if (lane == 0) c0 = atomicAdd(a.counter, 1u);
    JOIN();
    c0 = (uint32_t)uni((int)c0);
    if (c0 >= n_claims) break;
    for (uint32_t j = c0; j < n_jobs; j += n_claims) {
    const JobDev job = a.jobs[j];
    const int plen = (int)job.pat_len, tlen = (int)job.txt_len;
    const uint8_t* P = a.pat_base + job.pat_off;
    const uint8_t* T = a.txt_base + job.txt_off;
    const bool aln = a.scope_alignment != 0;
    // ---- identical sequences: one run of matches, penalty 0 (what the searches below return for them; see wfa.hip)
    if (plen == tlen && plen > 0) {
      uint64_t diff = 0;
      int i = 8 * lane;
      for (; i + 8 <= plen; i += 512) { uint64_t x, y; __builtin_memcpy(&x, P + i, 8); __builtin_memcpy(&y, T + i, 8); diff |= x ^ y; }
      if (i < plen && i + 8 > plen) for (int b = i; b < plen; ++b) diff |= (uint64_t)(P[b] ^ T[b]);
      if (!__ballot(diff != 0ull)) {
        if (lane == 0) {
          const Outs& O = S.out;
          const uint32_t o = job.out_index;
          if (O.status) O.status[o] = TRGT_WF_COMPLETED;
          if (O.score) O.score[o] = aln ? INT32_MIN : 0;
          if (O.n_match) O.n_match[o] = aln ? plen : 0;
          if (O.span4) { O.span4[4 * o + 0] = 0; O.span4[4 * o + 1] = (uint32_t)plen; O.span4[4 * o + 2] = 0; O.span4[4 * o + 3] = (uint32_t)tlen; }
          if (O.cigar_len) O.cigar_len[o] = aln ? 1u : 0u;
          if (O.ops_len) O.ops_len[o] = aln ? (uint32_t)plen : 0u;
          if (aln && O.cigar) O.cigar[job.cigar_off] = ((uint32_t)plen << 4) | 7u;
        }
        JOIN();
        cells_acc += 2ull;
        continue;
      }
    }
    bool fit = plen <= MAX_LEN && tlen <= MAX_LEN && plen > 0 && tlen > 0;
    {  // the two sequences into LDS when they fit (read through generic pointers from here on: unaligned eight-byte loads)
      const int t_at = (plen + 15) & ~15;
      if (fit && a.stage && t_at + tlen <= SC) {
        __syncthreads();  // (one wave: the previous alignment's reads of the staging area are done)
        for (int i = 16 * lane; i < plen; i += 1024) {
          if (i + 16 <= plen) { uint4 w; __builtin_memcpy(&w, P + i, 16); *reinterpret_cast<uint4*>(S.seq + i) = w; }
          else for (int b = i; b < plen; ++b) S.seq[b] = P[b];
        }
        for (int i = 16 * lane; i < tlen; i += 1024) {
          if (i + 16 <= tlen) { uint4 w; __builtin_memcpy(&w, T + i, 16); *reinterpret_cast<uint4*>(S.seq + t_at + i) = w; }
          else for (int b = i; b < tlen; ++b) S.seq[t_at + b] = T[b];
        }
        JOIN();
        __syncthreads();
        P = S.seq; T = S.seq + t_at;
      }
    }
    int score = INT32_MIN, rle_n = 0;
    unsigned long long cells = 0;
    bool overflow = false;
    int why = fit ? WHY_STATUS : WHY_LEN;
    if (fit) {
      // ---- wavefront_bialign_alignment (the recursion as an explicit stack, left half first) and wavefront_bialign_compute_score (its
      //      first level only) through ONE call site of the engine
      int sp = 0;
      if (lane == 0) {
        int* e = S.stack[0];
        e[0] = 0; e[1] = plen; e[2] = 0; e[3] = tlen; e[4] = CM; e[5] = CM; e[6] = (aln && max(plen, tlen) <= a.bi_min_length) ? 0 : INT32_MAX; e[7] = 1;
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
        int mode = rem <= a.bi_min_score ? (aln ? 1 : 2) : 0;
        Breakpoint bp;
        int es = 0, st; Seqs q;
        for (;;) {
          st = engine<METRIC, NL, NC, NS>(S, P + pb, T + tb, pl, tl, cb, ce, a.heur, mode, bp, es, q, cells, why);
          if (mode == 0 && st == ST_END_REACHED) { mode = aln ? 1 : 2; continue; }  // a front reached the other end: the base alignment
          break;
        }
        if (mode != 0) {
          if (st != ST_END_REACHED) { fit = false; break; }
          if (!aln) { score = METRIC == 1 ? es : -es; break; }
          int nt = 0;
          __syncthreads();  // (the history written lane by lane is read by every lane)
          backtrace<METRIC>(S, q, ce, es, nt, overflow);
          for (int i = nt - 1; i >= 0; --i) { const uint32_t e2 = (uint32_t)uni((int)S.rle_tmp[i]); rle_push(S.rle_out, rle_n, e2 & 0xF, (int)(e2 >> 4), overflow); }
          continue;
        }
        if (st != ST_OK) { fit = false; break; }
        if (top) score = METRIC == 1 ? bp.score : -bp.score;
        if (!aln) break;
        const int bh = bp.off_f, bv = bp.off_f - bp.k_f;
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
    // (the job record again: it was not carried through the level loops)
    const JobDev jb = a.jobs[j];
    if (!fit) {  // to the next kernel, from scratch
      if (lane == 0) {
        const Outs& O = S.out;
        const uint32_t r = atomicAdd(O.retry_count, 1u);
        if (r < O.retry_cap) { JobDev* dst = O.retry_jobs + r; dst->pat_off = jb.pat_off; dst->txt_off = jb.txt_off; dst->cigar_off = jb.cigar_off; dst->ops_off = jb.ops_off; dst->pat_len = jb.pat_len; dst->txt_len = jb.txt_len; dst->out_index = jb.out_index; dst->pad = 0; }
        else atomicAdd(O.retry_lost, 1u);
        if (O.why_hist) atomicAdd(O.why_hist + why, 1u);
      }
      JOIN();
      continue;
    }
    // ---- per-job epilogue (as wfa_kernel_body's): status, score, count_matches, alignment span, CIGAR
    uint32_t nm = 0, total = 0;
    for (int r = 0; r < rle_n; ++r) { const uint32_t e = (uint32_t)uni((int)S.rle_out[r]); total += e >> 4; if ((e & 0xF) == 7u) nm += e >> 4; }
    uint32_t* const cigar = uni_ptr(S.out.cigar);
    if (lane == 0) {
      const Outs& O = S.out;
      const uint32_t o = jb.out_index;
      if (O.status) O.status[o] = TRGT_WF_COMPLETED;
      if (O.score) O.score[o] = score;
      if (O.n_match) O.n_match[o] = (int32_t)nm;
      if (O.span4) { O.span4[4 * o + 0] = 0; O.span4[4 * o + 1] = jb.pat_len; O.span4[4 * o + 2] = 0; O.span4[4 * o + 3] = jb.txt_len; }
      if (O.cigar_len) O.cigar_len[o] = (uint32_t)rle_n;
      if (O.ops_len) O.ops_len[o] = total;
    }
    JOIN();
    if (cigar) for (int r = lane; r < rle_n; r += 64) cigar[jb.cigar_off + r] = S.rle_out[r];
    JOIN();
    cells_acc += cells;
    }
  }
  if (lane == 0 && a.cells_out && cells_acc) atomicAdd(a.cells_out, cells_acc);
#ifdef TRGT_LEAN_PROF
  __syncthreads();
  if (lane < 16 && l_lean_prof[lane]) atomicAdd(&g_lean_prof[lane], l_lean_prof[lane]);
#endif
}

}  // namespace lean

// Enqueue the lean kernels over a job list: tier one (64 diagonals per wavefront) over the list, tier two (256 diagonals, a large history)
// over what tier one hands on; what tier two does not take either is appended to retry_jobs (count at retry_count, both in device
// memory) for the generic kernel.  metric 1 (edit) or 3 (gap-affine 2,5,1), end-to-end, BiWFA, Heuristic::None or WFadaptive.
// mid_jobs / counters: the list between the tiers and five words (job counter and hand-over count / lost count of either tier).
int wfa_lean_launch(trgt_hip_ctx* c, const trgt_wfa_params& p, const WfaLaunch& L, JobDev* mid_jobs, JobDev* retry_jobs, unsigned int* retry_count, uint32_t retry_cap,
                    unsigned int* retry_lost, unsigned int* counters, unsigned long long* cells_out, unsigned int* why_hist) {
  lean::Args a;
  std::memset(&a, 0, sizeof a);
  static const int no_stage = [] { const char* e = TRGT_DEV_ENV("TRGT_WFA_LEAN_NO_STAGE"); return e ? atoi(e) : 0; }();  // developer switch: 1 / 2 / 3 = tier one / two / both read the sequences from global memory
  a.stage = !(no_stage & 1);
  // jobs per claim: consensus alignments differ in cost by two orders of magnitude (a wave that draws several expensive ones is the tail of
  // the launch: measured 0.47 ms with one job per claim against 0.9 ms with eight), edit distances of read pairs are alike
  // (round 5, cfg5: two per claim measure the same as one for a lone context -- 10.1 ms per call -- and the contexts of a pool, whose claims
  //  meet on the GPU's atomic units, gain 5-8 % with four: 400 -> 440 k loci/s)
  a.chunk = p.metric == 1 ? 8 : c->in_pool ? 4 : 2;
  if (c->knobs.lean_chunk > 0 && p.metric != 1) a.chunk = c->knobs.lean_chunk;  // (developer switch: jobs per claim of the consensus alignments)
  a.scope_alignment = p.scope != 0; a.bi_min_score = p.bialign_min_score; a.bi_min_length = p.bialign_min_length;
  a.heur.on = p.heuristic != 0; a.heur.min_len = p.h_min_wavefront_length; a.heur.max_dist = p.h_max_distance_threshold; a.heur.steps = p.h_steps_between_cutoffs;
  a.jobs = L.jobs_dev; a.n_jobs_dev = L.n_jobs_dev; a.n_jobs = (uint32_t)L.n_jobs_host;
  a.pat_base = L.pat_base; a.txt_base = L.txt_base; a.counter = counters;
  a.status = L.status; a.score = L.score; a.n_match = L.n_match; a.span4 = L.span4; a.cigar = L.cigar; a.cigar_len = L.cigar_len; a.ops_len = L.ops_len;
  a.cells_out = cells_out; a.retry_cap = retry_cap; a.retry_lost = retry_lost; a.why_hist = why_hist;
  // Up to three tiers: 64 diagonals per wavefront (NS = 1: nearly every alignment), optionally 128 (NS = 2, round 5), 256 (NS = 4).
  // mid_jobs holds the two lists between them (retry_cap entries each); counters: [0] tier-one job counter, [1] jobs handed to tier two,
  // [2] its counter, [3] jobs handed to tier three, [4] its counter.
  const bool more_tiers = mid_jobs != nullptr && !c->knobs.lean_one_tier;
  // (the middle tier is OFF unless TRGT_WFA_LEAN_MID_TIER=1.  Measured on cfg5, round 5: a launch over what tier one hands on is as long
  //  as its slowest alignment -- 0.5 to 0.9 ms for a read of one allele against the backbone of the other, whatever the number of
  //  strips -- and the tiers run one behind the other, so a third kernel ADDS its tail: one-context call 10.1 -> 10.8 ms.)
  const bool mid_tier = more_tiers && c->knobs.lean_mid_tier;
  JobDev* const mid2_jobs = mid_jobs ? mid_jobs + retry_cap : nullptr;
  a.retry_jobs = more_tiers ? mid_jobs : retry_jobs; a.retry_count = more_tiers ? counters + 1 : retry_count;
  constexpr int HC1 = 2816, HL1 = 96, SC1 = 1280, HCM = 8 * 1024, HLM = 192, SCM = 3072, HC2 = 25 * 1024, HL2 = 320, SC2 = 6144, HCME = 2048, SCME = 2048, HC2E = 4096, SC2E = 2048;  // (edit distances are score-only on the locus path: no history)
  const int64_t bound = L.jobs_bound > 0 ? L.jobs_bound : L.n_jobs_host;
  auto launch = [&](void (*fn)(const lean::Args), const lean::Args& args, int occ_default, const char* what) -> int {
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, 64, 0) != hipSuccess || occ < 1) { (void)hipGetLastError(); occ = occ_default; }
    const int64_t grid = std::max<int64_t>(1, std::min<int64_t>(bound, (int64_t)c->num_cus * occ));
    if (c->knobs.debug) fprintf(stderr, "[wfa] lean kernel (%s): metric %d, at most %lld jobs, occupancy %d, grid %lld\n", what, p.metric, (long long)bound, occ, (long long)grid);
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(64), 0, c->stream, args);
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) return fail(c, TRGT_ERR_HIP, "lean alignment kernel (%s) launch failed: %s", what, hipGetErrorString(le));
    return TRGT_OK;
  };
  if (int rc = launch(p.metric == 1 ? lean::wfa_lean_kernel<1, 1, HC1, HL1, SC1> : lean::wfa_lean_kernel<3, 1, HC1, HL1, SC1>, a, 8, "64 diagonals")) return rc;
  if (more_tiers) {
    lean::Args b = a;
    b.jobs = mid_jobs; b.n_jobs_dev = counters + 1; b.n_jobs = 0; b.counter = counters + 2;
    b.stage = !(no_stage & 2); b.chunk = 1;
    if (mid_tier) {
      b.retry_jobs = mid2_jobs; b.retry_count = counters + 3; b.why_hist = why_hist ? why_hist + 8 : nullptr;
      if (int rc = launch(p.metric == 1 ? lean::wfa_lean_kernel<1, 2, HCME, HLM, SCME> : lean::wfa_lean_kernel<3, 2, HCM, HLM, SCM>, b, 4, "128 diagonals")) return rc;
      b.jobs = mid2_jobs; b.n_jobs_dev = counters + 3; b.counter = counters + 4;
    }
    b.retry_jobs = retry_jobs; b.retry_count = retry_count; b.why_hist = why_hist ? why_hist + 16 : nullptr;
    if (int rc = launch(p.metric == 1 ? lean::wfa_lean_kernel<1, 4, HC2E, HL2, SC2E> : lean::wfa_lean_kernel<3, 4, HC2, HL2, SC2>, b, 2, "256 diagonals")) return rc;
  }
#ifdef TRGT_LEAN_PROF
  {
    unsigned long long h[16], z[16] = {0};
    (void)hipStreamSynchronize(c->stream);
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(lean::g_lean_prof), sizeof h);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(lean::g_lean_prof), z, sizeof h);
    const double lv = (double)h[8] + 1e-9;
    fprintf(stderr, "[lean prof] metric %d levels %llu | cycles per level: compute %.0f extend %.0f swap %.0f overlap %.0f history %.0f\n", p.metric, h[8], h[0] / lv, h[1] / lv, h[3] / lv, h[4] / lv, h[5] / lv);
  }
#endif
  return TRGT_OK;
}

}  // namespace trgt
