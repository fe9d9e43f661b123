// trgt_amd/csrc/hmm_ppl.hpp -- the Viterbi FILL of the motif HMM with one lane per MOTIF POSITION (round 5), included by hmm.hip.
//
// Replaces the fill part of Hmm::label (hmm_model.rs:54-114: generate_mats / calc_viterbi_score) for motif sets whose positions fit
// one wave; the back-pointers it writes (1 B per (column, state), the layout of hmm_viterbi_kernel) are traced back by the kernels
// that were there before (hmm_viterbi_kernel with its fill skipped, hmm_traceback_long_kernel).
//
// Why another layout.  With one lane per STATE (hmm_viterbi_kernel) a column costs the wave ~20 ds_bpermute_b32 -- every state fetches
// its predecessors from arbitrary lanes -- and a single STR motif (17-26 states) leaves most of the wave idle: two alleles per wave at
// best.  The model is regular (builder.rs:80-173): position k of a motif block owns a match, an insertion and a deletion state, and
//   m_k <- {m_(k-1), block start, i_(k-1), d_(k-2)}     i_k <- {i_k, m_k}     d_k <- {m_k, d_(k-1)}     block end <- {m_(n-1), i_(n-1), d_(n-2)}
// so with lane = position (the three states of a position in ONE lane, the block end in the last position's deletion slot, the skip
// block as one more position) every predecessor is in the lane itself or the lane before it: DPP wave_shr:1 moves, no crossbar.  The
// run end is a butterfly maximum over the lanes of the job (DPP quad_perm / row_mirror), the only crossbar fetch left per column is
// the own block's end.  A job takes G = 8 / 16 / 32 / 64 lanes (positions + 1 <= G), so a wave fills 8 alleles of a 3- to 7-base
// motif side by side.  Every f64 sum is formed as the reference forms it -- (score + ln p) + emission -- candidates are compared in
// predecessor-list order with strict '>', and the transition / emission terms are read from the same tables (host libm logarithms).
#pragma once

namespace ppl {

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(x);
  const int lo = (int)(u & 0xFFFFFFFFull), hi = (int)(u >> 32);
  // (bound_ctrl: a lane without a source reads 0 -- only lane 0 of a wave shift, whose value no position uses -- so the move needs no
  //  "old" operand and the compiler no copy in front of it)
  const int slo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  const int shi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __longlong_as_double((long long)(((unsigned long long)(unsigned)shi << 32) | (unsigned)slo));
}
__device__ __forceinline__ double shr1(double x) { return dpp_f64<0x138>(x); }  // the value of the lane before (wave_shr:1; lane 0: 0.0)

// a DPP move that leaves the lanes of the rows outside ROWS (and lanes without a source) at `ident`
template <int CTRL, int ROWS>
__device__ __forceinline__ double dpp_f64_rows(double x, double ident) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(x), d = (unsigned long long)__double_as_longlong(ident);
  const int slo = __builtin_amdgcn_update_dpp((int)(d & 0xFFFFFFFFull), (int)(u & 0xFFFFFFFFull), CTRL, ROWS, 0xF, false);
  const int shi = __builtin_amdgcn_update_dpp((int)(d >> 32), (int)(u >> 32), CTRL, ROWS, 0xF, false);
  return __longlong_as_double((long long)(((unsigned long long)(unsigned)shi << 32) | (unsigned)slo));
}
__device__ __forceinline__ double readlane_f64(double x, int l) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(x);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(u & 0xFFFFFFFFull), l), hi = (unsigned)__builtin_amdgcn_readlane((int)(u >> 32), l);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// maximum over the G lanes of a job (G a power of two, jobs aligned to G): xor-1, xor-2 inside a quad, mirror inside 8 and 16 lanes by
// DPP -- every lane of a row of sixteen then holds the row's maximum.  Jobs of 32 / 64 lanes: the rows are combined by row_bcast:15
// (lane 15 of a row to the row behind it) and row_bcast:31, and the last row's value comes back to everybody through a scalar register
// (v_readlane).  No crossbar: the two ds_bpermute round trips of the butterfly's last steps were ~250 cycles of a 64-lane column,
// twice (this and the minimum below).
template <int G>
__device__ __forceinline__ double group_max(double v, int lane) {
  v = max_f64(v, dpp_f64<0xB1>(v));    // quad_perm [1,0,3,2]
  v = max_f64(v, dpp_f64<0x4E>(v));    // quad_perm [2,3,0,1]
  v = max_f64(v, dpp_f64<0x141>(v));   // row_half_mirror
  if (G >= 16) v = max_f64(v, dpp_f64<0x140>(v));  // row_mirror
  if constexpr (G >= 32) {
    const double ninf = -__builtin_huge_val();
    v = max_f64(v, dpp_f64_rows<0x142, 0xA>(v, ninf));  // row_bcast:15 into rows 1 and 3: row 1 = rows 0-1, row 3 = rows 2-3
    if constexpr (G >= 64) {
      v = max_f64(v, dpp_f64_rows<0x143, 0xC>(v, ninf));  // row_bcast:31 into rows 2 and 3: row 3 = the wave
      return readlane_f64(v, 63);
    } else {
      const double a = readlane_f64(v, 31), b = readlane_f64(v, 63);
      return lane < 32 ? a : b;
    }
  }
  return v;
}

// ... and the minimum of a small integer over the same lanes
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int x) { return __builtin_amdgcn_mov_dpp(x, CTRL, 0xF, 0xF, true); }
template <int G>
__device__ __forceinline__ int group_min_i32(int v, int lane) {
  v = min(v, dpp_i32<0xB1>(v));
  v = min(v, dpp_i32<0x4E>(v));
  v = min(v, dpp_i32<0x141>(v));
  if (G >= 16) v = min(v, dpp_i32<0x140>(v));
  if constexpr (G >= 32) {  // (the rows combined as in group_max)
    v = min(v, __builtin_amdgcn_update_dpp(0x7FFFFFFF, v, 0x142, 0xA, 0xF, false));
    if constexpr (G >= 64) {
      v = min(v, __builtin_amdgcn_update_dpp(0x7FFFFFFF, v, 0x143, 0xC, 0xF, false));
      return __builtin_amdgcn_readlane(v, 63);
    } else {
      const int a = __builtin_amdgcn_readlane(v, 31), b = __builtin_amdgcn_readlane(v, 63);
      return lane < 32 ? a : b;
    }
  }
  return v;
}

// maximum / minimum of an integer over the wave, to a scalar: row scan by DPP, the rows combined by row_bcast:15 / :31, lane 63 read back
template <bool MAX>
__device__ __forceinline__ int wave_reduce_i32(int v) {
  constexpr int ident = MAX ? (int)0x80000000 : 0x7FFFFFFF;
  auto op = [](int a, int b) { return MAX ? max(a, b) : min(a, b); };
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x111 /* row_shr:1 */, 0xF, 0xF, false));
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x112 /* row_shr:2 */, 0xF, 0xF, false));
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x114 /* row_shr:4 */, 0xF, 0xF, false));
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x118 /* row_shr:8 */, 0xF, 0xF, false));
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x142 /* row_bcast:15 */, 0xA, 0xF, false));
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x143 /* row_bcast:31 */, 0xC, 0xF, false));
  return __builtin_amdgcn_readlane(v, 63);
}

__host__ __device__ inline int lanes_for(uint32_t positions) { return positions == 0 || positions > 64 ? 0 : positions <= 8 ? 8 : positions <= 16 ? 16 : positions <= 32 ? 32 : 64; }

constexpr int ROW_CHAIN_MIN = 16;  // deletion chains of this many steps and more (jobs of 32 / 64 lanes) go row by row; shorter ones stay a fixed-point walk
constexpr int CODE_WIN = 256;   // columns of symbol codes per window (+ 2 look-ahead)
constexpr int CODE_ROW = 272;

// The job slots of a launch: n_seg segments of one list, segment k = slots [begin[k], begin[k + 1]) of which the first counts[k] hold
// jobs (counts NULL: all of them).  One class of the host-built list is one segment; the device-resolved list of a locus batch has
// one segment per class (capacity: the candidates of the class; count: what the resolve kernel kept), filled by ONE launch per width.
struct PplSegs { uint32_t begin[9]; uint32_t n_seg; const uint32_t* counts; uint32_t first_slot[4], end_slot[4]; };  // [first_slot[g], end_slot[g]): the slots a launch of width 8 << g looks at (the segments that hold such jobs)

// PACK (round 6): ONE byte per lane and column -- the back-pointers of the position's match (bits 0-1), insertion (2) and deletion /
// block-end slot (3-4), of its block's start (5; the block's first position) and two bits of the job's run start / run end (6-7: lane 0
// the run start, lanes 1-3 bits 0-1 / 2-3 / 4-5 of the run end's block index) -- rows of G bytes: one store per lane and column instead
// of five byte stores into rows of one byte per STATE (three states per position), a third of the bytes written and read back.  The
// trace-backs translate (state -> lane, shift, mask: bp_loc in hmm.hip).  PACK = false keeps the round-5 layout (TRGT_HMM_PPL_WIDE).
template <int G, bool PACK>
__global__ void __launch_bounds__(64, (G <= 16 ? 3 : 2)) hmm_fill_ppl_kernel(const HmmJobDev* __restrict__ jobs, const HmmSetDev* __restrict__ sets, const uint8_t* __restrict__ model,
                                                          const uint8_t* __restrict__ seq_blob, uint8_t* __restrict__ bp_ws, const PplSegs segs) {
  constexpr int JPW = 64 / G;
  __shared__ double l_em[64 * 10];            // per lane: emission terms of its match state [5], of its insertion state [5]
  __shared__ uint8_t l_code[JPW][CODE_ROW];   // per job: window of symbol codes
  const int lane = (int)threadIdx.x, grp = lane / G, pos = lane % G, lane_base = grp * G;
  constexpr int GI = G == 8 ? 0 : G == 16 ? 1 : G == 32 ? 2 : 3;
  const uint32_t jidx = segs.first_slot[GI] + blockIdx.x * (uint32_t)JPW + (uint32_t)grp;
  const double NINF = -__builtin_huge_val();
  // the segment of the job list this slot lies in, and whether the segment holds a job there (a choice among values: an index into the
  // kernel argument would move it to scratch memory)
  uint32_t sb = segs.begin[0], se = segs.begin[1], sk = 0;
#pragma unroll
  for (int t = 1; t < 8; ++t) if ((uint32_t)t < segs.n_seg && jidx >= segs.begin[t]) { sb = segs.begin[t]; se = segs.begin[t + 1]; sk = (uint32_t)t; }
  uint32_t seg_jobs = se - sb;
  if (segs.counts) seg_jobs = min(seg_jobs, segs.counts[sk]);
  bool has = jidx >= sb && jidx - sb < seg_jobs && jidx < segs.end_slot[GI];
  HmmJobDev job{}; HmmSetDev set{};
  job.bp_off = 16; set.S = 8; set.n_blocks = 1;
  if (has) {
    job = jobs[jidx];
    set = sets[job.set];
    has = lanes_for(set.ppl_lanes) == G && job.seq_len > 0;
  }
  if (!__any(has)) return;
  const int S = (int)set.S, nb = (int)set.n_blocks, Spad = (S + 15) & ~15;
  const int Lj = has ? (int)job.seq_len + 2 : 0;
  const double* __restrict__ g_inlp = reinterpret_cast<const double*>(model + set.off_inlp);
  const double* __restrict__ g_em = reinterpret_cast<const double*>(model + set.off_em);
  const uint32_t* __restrict__ g_blocks = reinterpret_cast<const uint32_t*>(model + set.off_blocks);
  // ---- my position: block b, position k of its n; the skip block is one position behind the motifs'
  int kind = 0, k = 0, n = 0, ms_st = 0, end_pos = 0;
  unsigned long long ends_mask = 0ull;  // positions (relative to the job) that hold a block end, in block order
  if (has) {
    int acc = 0;
    for (int b = 0; b + 1 < nb; ++b) {
      const int ml = (int)g_blocks[2 * nb + b];
      if (pos >= acc && pos < acc + ml) { kind = 1; k = pos - acc; n = ml; ms_st = (int)g_blocks[0 * nb + b]; end_pos = acc + ml - 1; }
      acc += ml;
      ends_mask |= 1ull << (acc - 1);
    }
    if (pos == acc) { kind = 2; k = 0; n = 1; ms_st = (int)g_blocks[0 * nb + nb - 1]; end_pos = acc; }
    ends_mask |= 1ull << acc;
  }
  const bool is_pos = kind == 1, is_skip = kind == 2, is_end = (is_pos && k == n - 1) || is_skip;
  const int st_m = ms_st + 1 + k, st_i = st_m + n, st_d = is_skip ? ms_st + 2 : st_m + 2 * n;
  auto LP = [&](int slot, int st) -> double { return g_inlp[(size_t)slot * S + st]; };
  // match state (skip lane: the skip state, its self loop in the insertion slot C with its own score as the source)
  double lpA = NINF, lpB = NINF, lpC = NINF, lpD = NINF, lpI0 = NINF, lpI1 = NINF, lpO0 = NINF, lpO1 = NINF, lp_step = NINF, lpS0 = NINF, lpS1 = NINF;
  int adj = 0, chain_slot = 1;
  if (is_pos) {
    if (k == 0) { lpB = LP(0, st_m); adj = 1; }
    else { lpA = LP(0, st_m); lpB = LP(1, st_m); lpC = LP(2, st_m); if (k >= 2) lpD = LP(3, st_m); }
    lpI0 = LP(0, st_i); lpI1 = LP(1, st_i);
    if (k < n - 1) { lpO0 = LP(0, st_d); if (k >= 1) { lp_step = LP(1, st_d); chain_slot = 1; } }
    else { lpO0 = LP(0, st_d); lpO1 = LP(1, st_d); if (n > 1) { lp_step = LP(2, st_d); chain_slot = 2; } }
    lpS0 = LP(0, ms_st); lpS1 = LP(1, ms_st);
  } else if (is_skip) {
    lpB = LP(0, st_m); lpC = LP(1, st_m); adj = 1;
    lpO0 = LP(0, st_d);
    lpS0 = LP(0, ms_st); lpS1 = LP(1, ms_st);
  }
  const double lp_re = has ? LP(0, S - 2) : NINF, lp_rs0 = has ? LP(0, 1) : NINF, lp_rs1 = has ? LP(1, 1) : NINF;
  const double em_start0 = has ? g_em[0] : NINF;                                  // the start state's term for '#'
  // emission tables of my two emitting states -> LDS (all lanes: idle ones hold -inf, so that nothing undefined is ever summed)
  for (int c = 0; c < 5; ++c) {
    l_em[lane * 10 + c] = kind ? g_em[(size_t)c * S + st_m] : NINF;
    l_em[lane * 10 + 5 + c] = is_pos ? g_em[(size_t)c * S + st_i] : NINF;
  }
  for (int q = pos; q < CODE_ROW; q += G) l_code[grp][q] = 0;
  // ---- where my states' back-pointers go: five byte stores per column; a role a lane does not have stores to the job's dump slot
  //      (the 16 bytes in front of its back-pointer rows).  What is written for a state WITHOUT a score (-inf: no predecessor has
  //      one) is some valid predecessor slot, not the 0xFF of hmm_viterbi_kernel: the trace-back only ever reads the back-pointers of
  //      states on the best path, which all have scores, and the chunk maps of the long trace-back walk any byte safely (slots are
  //      taken modulo 4, block indices clamped).  That saves a compare and a select per state and column.
  uint8_t* const bp0 = bp_ws + job.bp_off;
  uint8_t* const dump = bp0 - 16 + (lane & 15);
  uint8_t* p_m = kind ? bp0 + st_m : dump;
  uint8_t* p_i = is_pos ? bp0 + st_i : dump;
  uint8_t* p_d = kind ? bp0 + st_d : dump;
  uint8_t* p_s = kind && k == 0 ? bp0 + ms_st : dump;
  const int aux_st = pos == 0 ? 1 : pos == 1 ? S - 2 : pos == 2 ? 0 : S - 1;   // run start | run end | start | end
  uint8_t* p_x = has && pos < 4 ? bp0 + aux_st : dump;
  int inc_m = kind ? Spad : 0, inc_i = is_pos ? Spad : 0, inc_s = kind && k == 0 ? Spad : 0, inc_x = has && pos < 4 ? Spad : 0;  // (0 once the job's last column is written: see run)
  if constexpr (PACK) { p_m = has ? bp0 + pos : dump; inc_m = has ? G : 0; }  // the lane's one byte of a row of G (every lane of a job has one)
  const int re_shift = pos >= 1 && pos <= 3 ? 2 * (pos - 1) : 31;  // which two bits of the run end's block index this lane keeps (31: none)
  // constant back-pointers of the aux states from column 1 on: run start <- run end (slot 1); start: none; end <- run end (slot 0)
  const int aux_const = pos == 0 ? 1 : pos == 2 ? 0xFF : 0;
  const bool aux_is_re = pos == 1;
  // what a winning candidate of my match state is called in the reference's predecessor list (position 0 and the skip state have
  // no "match before": their list starts with the block start)
  const int kA = 0, kB = 1 - adj, kC = 2 - adj, kD = 3 - adj;
  const int my_blk_or_none = is_end ? (int)__builtin_popcountll(ends_mask & ((1ull << pos) - 1ull)) : 255;  // my block's index (block ends only)
  const double lp_re_lane = is_end ? lp_re : NINF;  // only block ends are candidates of the run end
  const int a_end = (lane_base + end_pos) << 2;  // my block's end: its last position's deletion slot
  const uint8_t* __restrict__ seq = seq_blob + job.seq_off;
  // scalar loop bounds: the longest allele of the wave, the longest deletion chain of the wave
  int Lw = Lj, steps = is_pos ? n - 1 : 0;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { Lw = max(Lw, __shfl_xor(Lw, o)); steps = max(steps, __shfl_xor(steps, o)); }
  Lw = __builtin_amdgcn_readfirstlane(Lw); steps = __builtin_amdgcn_readfirstlane(steps);
  int Lmin = has ? Lj : 0x7FFFFFFF;  // the shortest allele of the wave: up to there no store needs a test of its own
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) Lmin = min(Lmin, __shfl_xor(Lmin, o));
  Lmin = __builtin_amdgcn_readfirstlane(Lmin);
  // Deletion chains of jobs of 8 / 16 lanes (a job lies inside one row of sixteen lanes): chain state k is max(own_k, d_(k-1) + c_k), i.e.
  // max over j <= k of own_j + c_(j+1) + ... + c_k summed in that order.  Instead of walking the lanes (a DPP move, an add and a max per
  // step, each waiting for the one before), lane j adds its OWN way down the chain -- T(t) = T(t-1) + c_(j+t), t additions that depend on
  // nothing but each other -- and lane k takes T(t) of the lane t places before it by a row rotation, all rotations independent of one
  // another.  Same sums in the same order (the rounding of an addition is monotone, so the maximum may be taken before or after), same
  // strict comparison against the own candidates.  c_(j+t) is -inf from the first lane of the next block / job on, so sums that would
  // leave the block -- and the values a rotation wraps around the row -- are -inf.
  constexpr int CT = G <= 8 ? 6 : G <= 16 ? 14 : 0;   // chain steps a job of G lanes can need (motif length - 1)
  double cstep[CT > 0 ? CT : 1];
  if constexpr (CT > 0) {
#pragma unroll
    for (int t = 1; t <= CT; ++t) {
      const double v = __longlong_as_double(((long long)__shfl_down((int)((unsigned long long)__double_as_longlong(lp_step) >> 32), t) << 32) |
                                            (unsigned)__shfl_down((int)((unsigned long long)__double_as_longlong(lp_step) & 0xFFFFFFFFull), t));
      cstep[t - 1] = lane + t < 64 ? v : NINF;
    }
  }
  // Jobs of 32 / 64 lanes with long motifs (VNTR loci: the chain of a column really is as long as the motif -- every state off the
  // alignment's own diagonal is best reached by deleting on from it): the chain in two phases.  (A) inside the rows of sixteen lanes,
  // the rotations of the narrow jobs (sums that would leave the ROW are -inf); (B) row after row, what enters a row from the last
  // lane of the row before: every lane adds ITS prefix of the row's steps to that value, one addition per step and no choice --
  // the steps behind its own are +0.0, which changes no sum (the scores are sums of logarithms: never -0.0).  Same sums in the same
  // order as the walk along the lanes: the value a row passes on is a maximum of such sums, and rounding is monotone.
  constexpr bool ROWS = CT == 0;
  double crow_a[ROWS ? 15 : 1], crow_b[ROWS ? 16 : 1];
  if constexpr (ROWS) {
    if (steps >= ROW_CHAIN_MIN) {
      auto lane_f64 = [&](double x, int src) { return bperm_f64(src << 2, x); };
#pragma unroll
      for (int t = 1; t <= 15; ++t) { const double v = lane_f64(lp_step, min(lane + t, 63)); crow_a[t - 1] = (lane & 15) + t <= 15 ? v : NINF; }
#pragma unroll
      for (int u = 0; u < 16; ++u) { const double v = lane_f64(lp_step, (lane & ~15) + u); crow_b[u] = u <= (lane & 15) ? v : 0.0; }
    } else {
#pragma unroll
      for (int t = 0; t < 15; ++t) crow_a[t] = NINF;
#pragma unroll
      for (int u = 0; u < 16; ++u) crow_b[u] = 0.0;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");

  // scores of the column before (final): my match / insertion state, my block's start
  double m = NINF, iv = NINF, msv = NINF;
  double mA = NINF, iC = NINF, dD = NINF;  // the lane before's match and insertion state, the deletion state two positions back
  const uint8_t* const codes = l_code[grp];
  const double* const my_em = l_em + lane * 10;
  double em_m = NINF, em_i = NINF;
  int sym1 = 0;
  auto refill = [&](const int i) {  // next window of symbol codes (hmm_code: '#' + allele + '#', invalid bases replaced); i a multiple of CODE_WIN
    for (int q = pos; q < CODE_WIN + 2 && i + q < Lj; q += G) l_code[grp][q] = (uint8_t)hmm_code(seq, i + q, Lj);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    const int sym = codes[0];
    sym1 = codes[1];
    em_m = my_em[sym]; em_i = my_em[5 + sym];
  };
  // One column.  FIRST: column 0; STEPS: the wave's deletion-chain steps as a compile-time number (8- and 16-lane jobs: the column is
  // then ONE basic block the scheduler can fill) or -1 (a loop); GUARD: stores tested against the job's own length (the columns beyond
  // the wave's shortest allele)
  auto column = [&](const int i, auto first_tag, auto steps_tag, auto guard_tag) {
    constexpr bool FIRST = decltype(first_tag)::value;
    constexpr int STEPS = decltype(steps_tag)::value;
    constexpr bool GUARD = decltype(guard_tag)::value;
    const int w = i & (CODE_WIN - 1);
    const double e_m = em_m, e_i = em_i;
    // (terms of the next column, symbol of the one after: their LDS round trips are not waited for in this column)
    em_m = my_em[sym1]; em_i = my_em[5 + sym1];
    sym1 = codes[w + 2];
    // ---- emitting states (from the column before): first strict maximum in list order = a tournament that keeps the earlier on ties
    double m_new, i_new;
    int bp_m, bp_i;
    if constexpr (FIRST) {
      m_new = NINF; i_new = NINF; bp_m = 0; bp_i = 0;  // index 0: states with predecessors have no score (hmm_model.rs:69-71)
    } else {
      const double vA = (mA + lpA) + e_m, vB = (msv + lpB) + e_m, vC = (iC + lpC) + e_m, vD = (dD + lpD) + e_m;
      const double ab = max_f64(vA, vB), cd = max_f64(vC, vD);
      const int c_ab = vB > vA ? kB : kA, c_cd = vD > vC ? kD : kC;
      m_new = max_f64(ab, cd);
      bp_m = cd > ab ? c_cd : c_ab;
      const double w0 = (iv + lpI0) + e_i, w1 = (m + lpI1) + e_i;
      i_new = max_f64(w0, w1);
      bp_i = w1 > w0 ? 1 : 0;
    }
    // ---- silent states of this column: deletion chain and block end (own candidates, then the lane before, step by step)
    const double o0 = (m_new + lpO0), o1 = (i_new + lpO1);
    const double own = max_f64(o0, o1);
    int bp_d = o1 > o0 ? 1 : 0;
    double val = own, cand = NINF;
    if constexpr (CT > 0) {
      double T = own;
      auto step = [&](auto tc) {
        constexpr int t = decltype(tc)::value;
        if constexpr (t <= STEPS) {
          T = (T + cstep[t - 1]);
          cand = max_f64(cand, dpp_f64<0x120 + t>(T));   // row_ror:t -- the lane t places before me, inside my row
        }
      };
      step(std::integral_constant<int, 1>()); step(std::integral_constant<int, 2>()); step(std::integral_constant<int, 3>());
      step(std::integral_constant<int, 4>()); step(std::integral_constant<int, 5>()); step(std::integral_constant<int, 6>());
      step(std::integral_constant<int, 7>()); step(std::integral_constant<int, 8>()); step(std::integral_constant<int, 9>());
      step(std::integral_constant<int, 10>()); step(std::integral_constant<int, 11>()); step(std::integral_constant<int, 12>());
      step(std::integral_constant<int, 13>()); step(std::integral_constant<int, 14>());
      val = max_f64(cand, own);
    } else if constexpr (STEPS == -2) {  // (jobs of 32 / 64 lanes, long motifs: rows of sixteen, see crow_a / crow_b)
      double T = own;
      auto stepA = [&](auto tc) {
        constexpr int t = decltype(tc)::value;
        T = (T + crow_a[t - 1]);
        cand = max_f64(cand, dpp_f64<0x120 + t>(T));   // row_ror:t
      };
      stepA(std::integral_constant<int, 1>()); stepA(std::integral_constant<int, 2>()); stepA(std::integral_constant<int, 3>());
      stepA(std::integral_constant<int, 4>()); stepA(std::integral_constant<int, 5>()); stepA(std::integral_constant<int, 6>());
      stepA(std::integral_constant<int, 7>()); stepA(std::integral_constant<int, 8>()); stepA(std::integral_constant<int, 9>());
      stepA(std::integral_constant<int, 10>()); stepA(std::integral_constant<int, 11>()); stepA(std::integral_constant<int, 12>());
      stepA(std::integral_constant<int, 13>()); stepA(std::integral_constant<int, 14>()); stepA(std::integral_constant<int, 15>());
      val = max_f64(cand, own);
      auto rowB = [&](auto mask_tag) {
        constexpr int MASK = decltype(mask_tag)::value;
        double carry = dpp_f64_rows<0x142, MASK>(val, NINF);  // row_bcast:15: the (final) value of the last lane of the row before
#pragma unroll
        for (int u = 0; u < 16; ++u) carry = (carry + crow_b[u]);
        cand = max_f64(cand, carry);
        val = max_f64(val, carry);
      };
      if constexpr (G == 32) rowB(std::integral_constant<int, 0xA>());  // (two jobs per wave: rows 1 and 3 are their second rows)
      else { rowB(std::integral_constant<int, 0x2>()); rowB(std::integral_constant<int, 0x4>()); rowB(std::integral_constant<int, 0x8>()); }
    } else if constexpr (STEPS >= 0) {  // (jobs of 32 / 64 lanes, short motifs: the walk along the lanes, its length known when compiled)
#pragma unroll
      for (int t = 0; t < STEPS; ++t) {
        cand = (shr1(val) + lp_step);
        val = max_f64(cand, own);
      }
    } else {
      // long motifs: the chain is a fixed-point iteration (every lane takes max(own, the lane before + ln p)): once an iteration changes
      // no lane of the wave, the values -- and the candidates of that iteration -- are the final ones.
      // (four steps per convergence test: the test and its branch cost as much as a step)
      for (int t = 0; t < steps; t += 4) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          cand = (shr1(val) + lp_step);
          val = max_f64(cand, own);
        }
        cand = (shr1(val) + lp_step);
        const double nv = max_f64(cand, own);
        const bool changed = nv != val;
        val = nv;
        if (!__any(changed)) break;
      }
    }
    bp_d = cand > own ? chain_slot : bp_d;
    const double d_new = val;
    // ---- run end: the block ends in block order, first strict maximum; every lane of the job has it
    const double v_re = (d_new + lp_re_lane);
    const double re_new = group_max<G>(v_re, lane);
    const double my_end = bperm_f64(a_end, d_new);
    const int bp_re = group_min_i32<G>(v_re == re_new ? my_blk_or_none : 255, lane);  // the first block that attains it
    // ---- run start {start state, run end}; the start state has a score in column 0 only
    double rs; int bp_rs;
    if constexpr (FIRST) {
      const double v0 = (em_start0 + lp_rs0), v1 = (re_new + lp_rs1);
      rs = max_f64(v0, v1);
      bp_rs = v1 > v0 ? 1 : 0;
    } else {
      rs = (re_new + lp_rs1);
      bp_rs = 1;
    }
    // ---- my block's start {run start, own block end}
    const double s0 = (rs + lpS0), s1 = (my_end + lpS1);
    const double ms_new = max_f64(s0, s1);
    const int bp_s = s1 > s0 ? 1 : 0;
    // ---- run start / run end / start / end state: one store by the first four lanes of the job
    int bp_x;
    if constexpr (FIRST) bp_x = pos == 0 ? bp_rs : pos == 1 ? bp_re : pos == 2 ? 0xFE : 0xFF;
    else bp_x = aux_is_re ? bp_re : aux_const;
    if constexpr (PACK) {
      const int aux = pos == 0 ? bp_rs : (bp_re >> re_shift) & 3;
      const int packed = bp_m | (bp_i << 2) | (bp_d << 3) | (bp_s << 5) | (aux << 6);
      if (!GUARD || i < Lj) *p_m = (uint8_t)packed;
      p_m += inc_m;
    } else {
      if (!GUARD || i < Lj) {
        *p_m = (uint8_t)bp_m; *p_i = (uint8_t)bp_i; *p_d = (uint8_t)bp_d; *p_s = (uint8_t)bp_s; *p_x = (uint8_t)bp_x;
      }
      p_m += inc_m; p_i += inc_i; p_d += inc_m; p_s += inc_s; p_x += inc_x;
    }
    // ---- what the next column's emitting states take from the lane before
    m = m_new; iv = i_new; msv = ms_new;
    mA = shr1(m_new);
    const double i_sh = shr1(i_new);
    iC = is_skip ? m_new : i_sh;
    dD = shr1(shr1(d_new));
  };
  // The columns in segments of SEG: before each one the wave looks at the jobs that still have columns.  Its chain length is THEIR
  // longest motif's, the stores need no test up to THEIR shortest allele -- a finished job's lanes store to its dump slot from then
  // on.  (Decided once per wave, a 10-kb allele of a 3-base motif next to a short allele of a 12-base motif paid fourteen rotations
  // and the guarded -- slower -- column for all of its columns: the 16-lane class of cfg3 3.0 ms.)
  constexpr int SEG = 128;
  static_assert(CODE_WIN % SEG == 0, "a segment lies inside one window of symbol codes");
  int Lmin_live = Lmin;
  auto run = [&](auto steps_tag, int i, const int seg_end) {
    const int safe_end = min(seg_end, Lmin_live);
    if (i == 0) { column(0, std::true_type(), steps_tag, std::true_type()); i = 1; }
    for (; i < safe_end; ++i) column(i, std::false_type(), steps_tag, std::false_type());
    for (; i < seg_end; ++i) column(i, std::false_type(), steps_tag, std::true_type());
  };
  for (int i = 0; i < Lw;) {
    if ((i & (CODE_WIN - 1)) == 0) refill(i);
    const int seg_end = min(Lw, (i | (SEG - 1)) + 1);
    const bool live = has && i < Lj;
    if (!live) { p_m = dump; p_i = dump; p_d = dump; p_s = dump; p_x = dump; inc_m = 0; inc_i = 0; inc_s = 0; inc_x = 0; }
    const int st = wave_reduce_i32<true>(live && is_pos ? n - 1 : 0);
    Lmin_live = wave_reduce_i32<false>(live ? Lj : 0x7FFFFFFF);
    if constexpr (CT == 0) {
      switch (st) {  // (short motifs: one loop copy per chain length; longer ones: the fixed-point loop or the rows)
        case 0: run(std::integral_constant<int, 0>(), i, seg_end); break;
        case 1: run(std::integral_constant<int, 1>(), i, seg_end); break;
        case 2: run(std::integral_constant<int, 2>(), i, seg_end); break;
        case 3: run(std::integral_constant<int, 3>(), i, seg_end); break;
        case 4: run(std::integral_constant<int, 4>(), i, seg_end); break;
        case 5: run(std::integral_constant<int, 5>(), i, seg_end); break;
        case 6: run(std::integral_constant<int, 6>(), i, seg_end); break;
        default:
          if (st >= ROW_CHAIN_MIN) run(std::integral_constant<int, -2>(), i, seg_end);
          else run(std::integral_constant<int, -1>(), i, seg_end);
          break;
      }
    } else {
      // (one copy of the loop per chain length: the step count is the wave's, a scalar)
      switch (st) {
        case 0: run(std::integral_constant<int, 0>(), i, seg_end); break;
        case 1: run(std::integral_constant<int, 1>(), i, seg_end); break;
        case 2: run(std::integral_constant<int, 2>(), i, seg_end); break;
        case 3: run(std::integral_constant<int, 3>(), i, seg_end); break;
        case 4: run(std::integral_constant<int, 4>(), i, seg_end); break;
        case 5: run(std::integral_constant<int, 5>(), i, seg_end); break;
        case 6: run(std::integral_constant<int, 6>(), i, seg_end); break;
        default:
          if constexpr (CT > 6) {
            if (st <= 8) run(std::integral_constant<int, 8>(), i, seg_end);
            else if (st <= 10) run(std::integral_constant<int, 10>(), i, seg_end);
            else run(std::integral_constant<int, 14>(), i, seg_end);
          }
          break;
      }
    }
    i = seg_end;
  }
}

}  // namespace ppl
