// trgt_amd/csrc/wfa_reg.hip -- register-resident wavefront pre-filter for TRGT's flank fallback alignments.
//
// find_spans (src/trgt/genotype/span_locater.rs:14-26) aligns a flank piece against a whole read (gap-affine 2,5,1, pattern
// global, text free at both ends: THREAD_WFA_FLANK, src/commands/genotype.rs:66-80), and then looks at TWO things only:
// count_matches() >= flank_len * min_flank_id_frac, and -- if so -- the text span.  The alignments that cost the time (reads cut
// short of a flank: 12 % of the fallback alignments, > 90 % of the wavefront offsets, penalties of 100-255) are almost all
// rejected by that test.  This kernel computes, for every alignment, the exact optimal penalty and an UPPER BOUND on the number
// of matches of the alignment the reference's back-trace would return -- without a wavefront history, without a back-trace,
// without touching HBM inside the level loop.  Alignments whose bound is below the threshold are rejected right here (exact:
// the true count is no larger); the others (5 % of the offsets on the bench workload) go on to the exact kernel
// (wfa_fast.hpp) which back-traces them.
//
// How the bound rides along for free.  With the pattern global (v = offset - k in [0, plen], plen <= 254) a wavefront cell is
// stored as v + 1 in eight bits (0 = NULL) instead of the offset in sixteen; the other eight bits of the 16-bit cell carry a
// match count.  In the v-domain the gap-affine recurrences are
//     I[s][k] = max(M[s-6][k-1], I[s-1][k-1])          (v unchanged)
//     D[s][k] = max(M[s-6][k+1], D[s-1][k+1]) + 1      (v + 1)
//     M[s][k] = max(M[s-2][k] + 1, I[s][k], D[s][k])   then extended along the diagonal by n matches: v += n, count += n
// and a plain unsigned 16-bit max over (v + 1) << 8 | count picks the furthest-reaching source exactly as WFA2 does, and among
// sources that tie on the offset the one with the LARGEST count.  The reference's back-trace breaks such ties by operation type
// (SURVEY.md Appendix A.6), so the count carried here is not its count_matches() -- but by induction over the cells it is never
// smaller, and offsets, ranges, penalty and terminating cell are untouched by the payload: they are the reference's, bit for bit.
//
// Where the wavefronts live.  One 64-lane wave owns one alignment; the live wavefronts -- M of the last six levels, I and D of
// the last one -- sit in VGPRs as packed pairs of 16-bit cells: lane l of strip t owns the 2*B consecutive diagonals
// kb = t*128*B + l*2*B + 2*j + {0, 1} (kb = k + plen).  Neighbouring diagonals are the other half of the same register, the
// neighbouring register, or -- at the two ends of a lane's block -- the neighbouring lane (one DPP wave shift per level, source
// and direction).  No LDS for the wavefronts, no barrier, no per-wave redundancy; LDS only holds the two sequences as 4-byte
// sliding windows (one aligned dword covers four bases: extension = xor + v_ffbl), padded with sentinels so that the ends of
// the sequences and NULL cells need no test at all (a NULL cell reads the all-ones window and extends by nothing).
//
// Exactness of the ranges (wavefront_compute_limits_input / wavefront_compute_trim_ends of WFA2-lib).  Before termination
// v <= plen holds for every cell, D cells are always in bounds, I cells satisfy v <= plen - 1, and an offset can exceed tlen
// only on the diagonals k > tlen - plen.  So trimming is arithmetic everywhere but at the upper end of M and I (found with
// ballots over one strip), M needs its out-of-bounds test only in the strips that reach above tlen - plen, and the I cells the
// reference trims away (non-NULL, out of bounds, beyond the last valid one) are zeroed so that NULL == 0 == "outside the
// trimmed range" holds for every register.  The number of offsets computed per level therefore equals the reference's
// (tests/test_filter_gpu.py compares it with the count of the CPU restatement).
#include <algorithm>
#include <type_traits>

#include "wfa_host.hpp"

#ifndef TRGT_FLT_W4
#define TRGT_FLT_W4 4  // waves per SIMD the four-strip instantiation is compiled for
#endif

namespace trgt {
namespace wfa {

namespace {

__device__ __forceinline__ uint32_t rpk_min(uint32_t a, uint32_t b) {
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  union U { uint32_t u; us2 v; } x, y, r;
  x.u = a; y.u = b; r.v = __builtin_elementwise_min(x.v, y.v);
  return r.u;
}
__device__ __forceinline__ uint32_t rpk_max(uint32_t a, uint32_t b) {
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  union U { uint32_t u; us2 v; } x, y, r;
  x.u = a; y.u = b; r.v = __builtin_elementwise_max(x.v, y.v);
  return r.u;
}
__device__ __forceinline__ uint32_t rpk_add_sat(uint32_t a, uint32_t b) {  // v_pk_add_u16 ... clamp
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  union U { uint32_t u; us2 v; } x, y, r;
  x.u = a; y.u = b; r.v = __builtin_elementwise_add_sat(x.v, y.v);
  return r.u;
}
__device__ __forceinline__ uint32_t rpk_sub(uint32_t a, uint32_t b) {
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  union U { uint32_t u; us2 v; } x, y, r;
  x.u = a; y.u = b; r.v = x.v - y.v;
  return r.u;
}
__device__ __forceinline__ uint32_t rpk_add(uint32_t a, uint32_t b) {
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  union U { uint32_t u; us2 v; } x, y, r;
  x.u = a; y.u = b; r.v = x.v + y.v;
  return r.u;
}
// v + 1 on both halves unless NULL (a non-NULL cell is >= 0x0100)
__device__ __forceinline__ uint32_t inc_v_nz(uint32_t x) { return rpk_add(x, rpk_min(x, 0x01000100u)); }
__device__ __forceinline__ uint32_t ffbl_or_m1(uint32_t v) {  // v_ffbl_b32: -1 for 0
  uint32_t r;
  asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(v));
  return r;
}
__device__ __forceinline__ int rfl_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
// A copy the optimiser must take element by element: without it the rotation of the wavefront ring is recognised as a block copy,
// the rows become ten-wide vector values, and every branch that touches ONE element copies the whole row.
__device__ __forceinline__ uint32_t opaque(uint32_t x) { asm volatile("" : "+v"(x)); return x; }
// lane i <- own[i - 1]; lane 0 <- below[63]
__device__ __forceinline__ uint32_t shr_from(uint32_t own, uint32_t below) {
  const int t = __builtin_amdgcn_update_dpp(0, (int)below, 0x13C /* wave_ror:1 */, 0xF, 0xF, false);
  return (uint32_t)__builtin_amdgcn_update_dpp(t, (int)own, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}
// lane i <- own[i + 1]; lane 63 <- above[0]
__device__ __forceinline__ uint32_t shl_from(uint32_t own, uint32_t above) {
  const int t = __builtin_amdgcn_update_dpp(0, (int)above, 0x134 /* wave_rol:1 */, 0xF, 0xF, false);
  return (uint32_t)__builtin_amdgcn_update_dpp(t, (int)own, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
}

constexpr uint32_t PAT_PAD = 0x01010101u, TXT_PAD = 0x02020202u, NULL_WIN = 0xFFFFFFFFu;
// TRGT_FLT_PAD (an A/B build, not the default: DESIGN.md 5): the text windows padded by one dword per 32, so that lanes reading at equal v --
// four dwords apart -- fall into 32 different banks instead of eight.  Costs two or three VALU instructions per read.
#ifdef TRGT_FLT_PAD
__device__ __forceinline__ int tw_phys(int i) { return i + (i >> 5); }
#else
__device__ __forceinline__ int tw_phys(int i) { return i; }
#endif
constexpr int PWN = 264;       // pattern windows: index v + 1 (0: the NULL window), plen <= 254
constexpr int TW_EXTRA = 264;  // text windows: index (v + 1) + kb, i.e. text position + plen + 1
constexpr int SMAX = 300;      // more levels than any flank alignment can take (a pattern of <= 254 bases is deleted for <= 5 + 254)

// bytes that collide with the sentinels make the job "dirty": it is handed to the exact kernel unseen
__device__ __forceinline__ uint32_t has_zero_byte(uint32_t v) { return (v - 0x01010101u) & ~v & 0x80808080u; }
__device__ __forceinline__ uint32_t dirty_dword(uint32_t x) {
  return has_zero_byte(x ^ PAT_PAD) | has_zero_byte(x ^ TXT_PAD) | has_zero_byte(~x);
}

}  // namespace

// NS strips of 128 * B diagonals; lane l of strip t owns the 2 * B consecutive diagonals kb = t * 128 * B + l * 2 * B + 2 * j + {0, 1},
// j = 0 .. B - 1 (kb = k + plen): B packed registers per strip and component.  A strip's cells follow the diagonal order lane by
// lane, so "the last in-bounds cell" is a few ballots over ONE strip, the out-of-bounds test of M is needed only in the strips
// that reach above tlen - plen, and strips outside the level's limits are skipped.
// X = mismatch penalty, OE = gap_open + gap_extend (gap_extend = 1): M[s - X] and M[s - OE] are the sources of a level, the ring keeps
// R = max(X, OE) M levels (wgs preset 2,5,1: X = 2, OE = 6; targeted preset 1,0,1 of cli.rs:271-280: X = OE = 1, one level, updated in place).
template <int NS, int B, int X = 2, int OE = 6>
__global__ void __launch_bounds__(64, (NS * B <= 8 ? TRGT_FLT_W4 : NS * B <= 10 ? 3 : 2)) wfa_filter_kernel(const FilterArgs a) {
  constexpr int NP = NS * B, SW = 128 * B, D = NS * SW, TWN = D + TW_EXTRA, LW = 2 * B, R = X > OE ? X : OE;
  // The M levels a level reads, s - X and s - OE, are in ITS OWN residue class modulo G = gcd(X, OE) (2,5,1: even levels read even levels,
  // odd ones odd ones).  So the ring of the last R levels is G rings of R / G: a level rotates only the ring of its class -- two register
  // moves per pair of cells and level instead of five -- and the level loop is unrolled over the classes, which makes the ring a level
  // uses a compile-time choice.
  constexpr int G = (X == OE) ? X : ((X % 2 == 0 && OE % 2 == 0) ? 2 : 1), DEP = R / G, IX = X / G - 1, IO = OE / G - 1;
  static_assert(R % G == 0 && X % G == 0 && OE % G == 0 && G <= 2, "ring classes");
#ifdef TRGT_FLT_PAD
  __shared__ uint32_t lds[PWN + TWN + TWN / 32 + 2];
#else
  __shared__ uint32_t lds[PWN + TWN];
#endif
  uint32_t* const Pw = lds;
  uint32_t* const Tw = lds + PWN;
  const int lane = (int)threadIdx.x;
  const uint32_t* const twl = Tw + lane * LW;  // + (v + 1) + C_p: the window of diagonal kb = C_p + lane * LW at offset v + k
#ifdef TRGT_FLT_PAD
#define TWR(x) Tw[tw_phys(lane * LW + (int)(x))]
#else
#define TWR(x) twl[x]
#endif
  const uint32_t n_jobs = a.n_jobs_dev ? min(*a.n_jobs_dev, a.n_jobs) : a.n_jobs;  // (never past the list the host sized)
  unsigned long long cells_acc = 0, kept_acc = 0;

  // 4-byte sliding windows of `len` bytes at src into W[i] (i = 0 .. n_win - 1), bytes beyond the sequence = pad
  auto stage = [&](const uint8_t* __restrict__ src, int len, uint32_t* __restrict__ W, int n_win, uint32_t pad, uint32_t& dirty, int w0, bool text) {  // (text windows: W = Tw, window i at tw_phys(w0 + i))
    for (int i0 = 4 * lane; i0 < n_win; i0 += 256) {
      uint32_t d0 = pad, d1 = pad;
      if (i0 + 8 <= len) {
        __builtin_memcpy(&d0, src + i0, 4); __builtin_memcpy(&d1, src + i0 + 4, 4);
        dirty |= dirty_dword(d0) | dirty_dword(d1);
      } else if (i0 < len) {
        for (int b = 0; b < 8; ++b)
          if (i0 + b < len) {
            const uint32_t c = src[i0 + b];
            if (c == 1u || c == 2u || c == 255u) dirty |= 1u;
            if (b < 4) d0 = (d0 & ~(0xFFu << (8 * b))) | (c << (8 * b));
            else d1 = (d1 & ~(0xFFu << (8 * (b - 4)))) | (c << (8 * (b - 4)));
          }
      }
      auto at = [&](int i) -> uint32_t& { return W[text ? tw_phys(w0 + i) : w0 + i]; };
      at(i0) = d0;
      if (i0 + 1 < n_win) at(i0 + 1) = __builtin_amdgcn_alignbyte(d1, d0, 1);
      if (i0 + 2 < n_win) at(i0 + 2) = __builtin_amdgcn_alignbyte(d1, d0, 2);
      if (i0 + 3 < n_win) at(i0 + 3) = __builtin_amdgcn_alignbyte(d1, d0, 3);
    }
  };

  for (;;) {
    uint32_t j = 0;
    if (lane == 0) j = atomicAdd(a.counter, 1u);
    j = (uint32_t)rfl_i((int)j);
    if (j >= n_jobs) break;
    // (the job's fields as scalars: held as a struct its twelve dwords are loaded per lane, live across the level loop for the epilogue and
    //  spilled there -- with the per-lane counters below that was the kernel's 88 bytes of scratch per lane, half of its HBM traffic)
    const JobDev* const jp = a.jobs + j;
    const int plen = rfl_i((int)jp->pat_len), tlen = rfl_i((int)jp->txt_len);
    const uint64_t pat_off = ((uint64_t)(uint32_t)rfl_i((int)(jp->pat_off >> 32)) << 32) | (uint32_t)rfl_i((int)(uint32_t)jp->pat_off);
    const uint64_t txt_off = ((uint64_t)(uint32_t)rfl_i((int)(jp->txt_off >> 32)) << 32) | (uint32_t)rfl_i((int)(uint32_t)jp->txt_off);
    if (plen + tlen + 1 <= a.diag_lo || plen + tlen + 1 > a.diag_hi) continue;  // another launch's job
    int keep = 0, score_out = INT32_MIN, bound_out = -1;
    uint32_t band = 0;  // (kept jobs: bit 31 | penalty << 16 | biased end diagonal when the run completed)
    const bool fits = plen >= 1 && plen <= 254 && tlen >= plen && tlen + plen + 1 <= D;
    uint32_t dirty = 0;
    if (fits) {
      // ---- the two sequences as sliding windows (one wave: its LDS operations execute in order, no barrier needed)
      if (lane == 0) Pw[0] = NULL_WIN;
      stage(a.pat_base + pat_off, plen, Pw, PWN - 1, PAT_PAD, dirty, 1, false);
      for (int i = lane; i < plen + 1; i += 64) Tw[tw_phys(i)] = TXT_PAD;  // text positions < 0: only NULL cells look there
      stage(a.txt_base + txt_off, tlen, Tw, TWN - (plen + 1), TXT_PAD, dirty, plen + 1, true);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    }
    const bool is_dirty = __builtin_amdgcn_ballot_w64(dirty != 0u) != 0ull;
    if (!fits || is_dirty) keep = 1;
    else {
      // one step of the extension of both cells of a packed pair: the number of matching bases (0 .. 4) inside the next window,
      // n_a | n_b << 16.  A NULL cell (v + 1 = 0) reads the all-ones pattern window: no match, it stays NULL.
      auto window_step = [&](uint32_t key, int C) __attribute__((always_inline)) -> uint32_t {
        const uint32_t va = (key >> 8) & 0xFFu, vb = key >> 24;
        // the four LDS reads of the pair go out together, one wait for all of them
        const uint32_t pa = Pw[va], ta = TWR(va + C), pb = Pw[vb], tb = TWR(vb + C + 1);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);  // 4 DS reads
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);  // then the VALU work
        const uint32_t na = min(ffbl_or_m1(pa ^ ta) >> 3, 4u);
        const uint32_t nb = min(ffbl_or_m1(pb ^ tb) >> 3, 4u);
        return na | (nb << 16);
      };
      // Extension of the level in Mx: first window of every cell straight-line (all LDS reads in flight together), then, pair by
      // pair, the cells that matched a whole window go on (a few per level: random bases agree on four in a row once in 256).
      // Returns the packed maximum over the level (termination test).
      auto extend_level = [&](uint32_t (&Mx)[NP]) __attribute__((always_inline)) -> uint32_t {
        static_assert(NP <= 14, "the continuation flags of a level share one register");
        uint32_t cflag = 0;  // bit 2 + (NP - 1 - p): the A cell of pair p matched a whole window, bit 18 + (NP - 1 - p): its B cell
#pragma unroll
        for (int t = 0; t < NS; ++t) {
          // the LDS reads of a whole strip (4 B of them) go out together and are waited for once (sched_group_barrier: left alone, the
          // compiler -- short of registers -- read, waited and compared cell by cell: 67 waits for 128 reads; 2.26 -> 2.08 ms and
          // 1.07 -> 0.81 ms for the two launches of the 10k-locus batch)
          uint32_t pw[2 * B], tw[2 * B];
#pragma unroll
          for (int jj = 0; jj < B; ++jj) {
            const uint32_t key = Mx[t * B + jj];
            const uint32_t va = (key >> 8) & 0xFFu, vb = key >> 24;
            const int C = t * SW + 2 * jj;
            pw[2 * jj] = Pw[va]; tw[2 * jj] = TWR(va + C); pw[2 * jj + 1] = Pw[vb]; tw[2 * jj + 1] = TWR(vb + C + 1);
          }
          __builtin_amdgcn_sched_group_barrier(0x100, 4 * B, 0);
#pragma unroll
          for (int jj = 0; jj < B; ++jj) {
            const int p = t * B + jj;
            const uint32_t na = min(ffbl_or_m1(pw[2 * jj] ^ tw[2 * jj]) >> 3, 4u), nb = min(ffbl_or_m1(pw[2 * jj + 1] ^ tw[2 * jj + 1]) >> 3, 4u);
            const uint32_t n = na | (nb << 16);
            Mx[p] += n * 0x0101u;
            cflag = (cflag << 1) | (n & 0x00040004u);
          }
          __builtin_amdgcn_sched_barrier(0);  // one strip's LDS reads in flight at a time: all of them together cost a third of the register file
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          uint32_t c = (cflag >> (NP - 1 - p)) & 0x00040004u;
          if (__builtin_amdgcn_ballot_w64(c != 0u)) {
            uint32_t m = Mx[p];  // (the loop carries a scalar copy: an array element updated inside it would drag the array along)
            do {
              if (c != 0u) {
                const uint32_t n2 = window_step(m, (p / B) * SW + 2 * (p % B));
                const uint32_t add = ((c & 0xFFFFu) ? n2 & 0xFFFFu : 0u) | ((c >> 16) ? n2 & 0xFFFF0000u : 0u);
                m += add * 0x0101u;
                c = add & 0x00040004u;
              }
            } while (__builtin_amdgcn_ballot_w64(c != 0u));
            Mx[p] = m;
          }
        }
        uint32_t t = 0;
#pragma unroll
        for (int p = 0; p < NP; ++p) t = rpk_max(t, Mx[p]);
        return t;
      };
      uint32_t Mr[G][DEP][NP], Ir[NP], Dr[NP];  // Mr[c][d]: level s - G (d + 1) of class c = s mod G, for the level s being computed
#pragma unroll
      for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int d = 0; d < DEP; ++d) Mr[g][d][p] = 0u;
        Ir[p] = 0u; Dr[p] = 0u;
      }
      // trimmed ranges of the live wavefronts in k (not biased); null = (1, -1) as in the library
      int mlo[G][DEP], mhi[G][DEP], ilo = 1, ihi = -1, dlo = 1, dhi = -1;
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int d = 0; d < DEP; ++d) { mlo[g][d] = 1; mhi[g][d] = -1; }
      const int lane_kb0 = lane * LW;
      const int term_v = plen + 1;
      const int lane_bnd0 = tlen + 1 + plen - lane_kb0;  // v + 1 <= lane_bnd - C  <=>  offset <= tlen on diagonal kb = C + lane_kb
      // ---- level 0: M[0][k] = k for k in [0, tlen] (v = 0), then extended
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int kbA = (p / B) * SW + 2 * (p % B) + lane_kb0;
        Mr[0][0][p] = ((unsigned)(kbA - plen) <= (unsigned)tlen ? 0x0100u : 0u) | ((unsigned)(kbA + 1 - plen) <= (unsigned)tlen ? 0x01000000u : 0u);
      }
      uint32_t tmax = extend_level(Mr[0][0]);
      mlo[0][0] = 0; mhi[0][0] = tlen;
      unsigned long long cells = (unsigned long long)tlen + 1ull;
      int s = 0, num_null = 0;
      int next_check = 16;  // early rejection: the next level at which it is tried (every 16th, sooner when the best cell is close to the limit)
      bool done = false, bail = false, early = false;
      // one level of class RC (= s mod G once s is counted up); false: the alignment is over (done / bail / early say how)
      auto level = [&](auto rc_) __attribute__((always_inline)) -> bool {
        constexpr int RC = decltype(rc_)::value;
        uint32_t (&Ms)[DEP][NP] = Mr[RC];
        int (&slo)[DEP] = mlo[RC];
        int (&shi)[DEP] = mhi[RC];
        // ---- termination (wavefront_termination_endsfree with pattern_end_free = 0, text_end_free = tlen): v == plen
        const bool tA = ((tmax >> 8) & 0xFFu) == (uint32_t)term_v, tB = (tmax >> 24) == (uint32_t)term_v;
        if (__builtin_amdgcn_ballot_w64(tA || tB)) { done = true; return false; }
        ++s;
        if (s > SMAX) { bail = true; return false; }
        const bool n_mm = slo[IX] > shi[IX], n_mo = slo[IO] > shi[IO], n_ie = ilo > ihi, n_de = dlo > dhi;
        // (per-lane constants go through opaque() once per level: hoisted out of the loop, their per-pair variants -- bounds, diagonal
        //  numbers -- would sit in two dozen registers for the whole alignment)
        const int lane_bnd = (int)opaque((uint32_t)lane_bnd0), lane_kb = (int)opaque((uint32_t)lane_kb0);
        uint32_t (&Mn)[NP] = Ms[DEP - 1];  // the new level replaces M[s - R] in place (its last use: the first pass below for the gap-open
                                         // source, the read of the same element in the third pass when it is also the mismatch source)
        int nmlo = 1, nmhi = -1, nilo = 1, nihi = -1, ndlo = 1, ndhi = -1;
        if (n_mm && n_mo && n_ie && n_de) {
          if (++num_null > R + 1) { bail = true; return false; }  // (cannot happen with a free text; the exact kernel decides)
#pragma unroll
          for (int p = 0; p < NP; ++p) Mn[p] = 0u;  // (I and D are NULL already: their sources were)
          tmax = 0;
        } else {
          num_null = 0;
          // wavefront_compute_limits_input (null wavefronts take part with lo = 1, hi = -1, as in the library)
          const int lo = min(min(slo[IX], slo[IO] - 1), min(ilo + 1, dlo - 1));
          const int hi = max(max(shi[IX], shi[IO] + 1), max(ihi + 1, dhi - 1));
          cells += 3ull * (unsigned long long)max(0, hi - lo + 1);
          // ---- the recurrences.  ins[k] = max(Mo, Ie)[k - 1] and del[k] = (max(Mo, De) + 1)[k + 1]: ONE diagonal shift per
          //      component, taken after the maximum.  In place: first I <- max(Mo, I), D <- max(Mo, D) + 1, then the shifts
          //      (I from the top strip down, D from the bottom up, so that the neighbour's unshifted value is still there).
#pragma unroll
          for (int p = 0; p < NP; ++p) {
            const uint32_t mo = Ms[IO][p];
            Ir[p] = rpk_max(mo, Ir[p]);
            Dr[p] = inc_v_nz(rpk_max(mo, Dr[p]));
          }
#pragma unroll
          for (int t = NS - 1; t >= 0; --t) {
            // lane i <- lane i - 1's last pair; lane 0 <- the last pair of lane 63 of the strip below
            const uint32_t edge = shr_from(Ir[t * B + B - 1], t > 0 ? Ir[(t - 1) * B + B - 1] : 0u);
#pragma unroll
            for (int jj = B - 1; jj >= 0; --jj) Ir[t * B + jj] = __builtin_amdgcn_alignbit(Ir[t * B + jj], jj ? Ir[t * B + jj - 1] : edge, 16);
          }
#pragma unroll
          for (int t = 0; t < NS; ++t) {
            const uint32_t edge = shl_from(Dr[t * B], t < NS - 1 ? Dr[(t + 1) * B] : 0u);
#pragma unroll
            for (int jj = 0; jj < B; ++jj) Dr[t * B + jj] = __builtin_amdgcn_alignbit(jj < B - 1 ? Dr[t * B + jj + 1] : edge, Dr[t * B + jj], 16);
          }
#pragma unroll
          for (int t = 0; t < NS; ++t) {
            const bool top = t * SW + SW - 1 > tlen;  // (uniform) diagonals k > tlen - plen: an offset can pass the end of the text
#pragma unroll
            for (int jj = 0; jj < B; ++jj) {
              const int p = t * B + jj;
              uint32_t mxp = rpk_max(Dr[p], rpk_max(inc_v_nz(Ms[IX][p]), Ir[p]));
              if (top) {  // "adjust offset out of boundaries": offset > tlen -> NULL
                const int bA = lane_bnd - (t * SW + 2 * jj);
                const bool okA = (int)((mxp >> 8) & 0xFFu) <= bA, okB = (int)(mxp >> 24) <= bA - 1;
                mxp = (okA ? mxp & 0xFFFFu : 0u) | (okB ? mxp & 0xFFFF0000u : 0u);
              }
              Mn[p] = mxp;
            }
          }
          tmax = extend_level(Mn);
          // ---- wavefront_compute_trim_ends, restated: D is always in bounds; I and M are, below k = tlen - plen (+ 1)
          constexpr int INF = 1 << 20;
          const int so_lo = n_mo ? INF : slo[IO], so_hi = n_mo ? -INF : shi[IO];
          const bool has_d = !n_mo || !n_de, has_i = !n_mo || !n_ie;
          if (has_d) { ndlo = min(so_lo, n_de ? INF : dlo) - 1; ndhi = max(so_hi, n_de ? -INF : dhi) - 1; }
          // last / first cell of a component for which pred holds, looking at the strips from biased kb_hi downwards / kb_lo upwards
          auto find_last = [&](const uint32_t (&R)[NP], int kb_lo, int kb_hi, auto pred) __attribute__((always_inline)) -> int {
            int best = -1;
#pragma unroll
            for (int t = NS - 1; t >= 0; --t) {
              if (best < 0 && t * SW <= kb_hi && t * SW + SW - 1 >= kb_lo) {
#pragma unroll
                for (int jj = 0; jj < B; ++jj) {
                  bool va, vb;
                  pred(R[t * B + jj], t * SW + 2 * jj, va, vb);
                  const unsigned long long mA = __builtin_amdgcn_ballot_w64(va), mB = __builtin_amdgcn_ballot_w64(vb);
                  if (mA) best = max(best, (63 - (int)__builtin_clzll(mA)) * LW + t * SW + 2 * jj);
                  if (mB) best = max(best, (63 - (int)__builtin_clzll(mB)) * LW + t * SW + 2 * jj + 1);
                }
              }
            }
            return best;
          };
          auto find_first = [&](const uint32_t (&R)[NP], int kb_lo, int kb_hi, auto pred) __attribute__((always_inline)) -> int {
            int best = INF;
#pragma unroll
            for (int t = 0; t < NS; ++t) {
              if (best == INF && t * SW <= kb_hi && t * SW + SW - 1 >= kb_lo) {
#pragma unroll
                for (int jj = 0; jj < B; ++jj) {
                  bool va, vb;
                  pred(R[t * B + jj], t * SW + 2 * jj, va, vb);
                  const unsigned long long mA = __builtin_amdgcn_ballot_w64(va), mB = __builtin_amdgcn_ballot_w64(vb);
                  if (mA) best = min(best, (int)__builtin_ctzll(mA) * LW + t * SW + 2 * jj);
                  if (mB) best = min(best, (int)__builtin_ctzll(mB) * LW + t * SW + 2 * jj + 1);
                }
              }
            }
            return best == INF ? -1 : best;
          };
          auto nonzero = [&](uint32_t x, int C, bool& va, bool& vb) { (void)C; va = (x & 0xFFFFu) != 0u; vb = (x >> 16) != 0u; };
          auto i_valid = [&](uint32_t x, int C, bool& va, bool& vb) {
            const int bA = lane_bnd - C, ea = (int)((x >> 8) & 0xFFu), eb = (int)(x >> 24);
            va = ea != 0 && ea <= bA; vb = eb != 0 && eb <= bA - 1;
          };
          int ic_lo = INF, ic_hi = -INF;  // non-NULL I cells
          if (has_i) {
            ic_lo = min(so_lo, n_ie ? INF : ilo) + 1; ic_hi = max(so_hi, n_ie ? -INF : ihi) + 1;
            nilo = ic_lo; nihi = ic_hi;
            if (ic_hi > tlen - plen + 1) {
              const int f = find_last(Ir, ic_lo + plen, ic_hi + plen, i_valid);
              if (f < 0) { nilo = 1; nihi = -1; }
              else {
                nihi = f - plen;
                if (ic_lo > tlen - plen + 1) nilo = find_first(Ir, ic_lo + plen, ic_hi + plen, i_valid) - plen;
              }
              if (nilo > nihi || nilo != ic_lo || nihi != ic_hi) {
                // the cells the reference trims away are not NULL: zero them (NULL == outside the trimmed range, for every register)
                const int zlo = nilo > nihi ? INF : nilo + plen, zhi = nilo > nihi ? -INF : nihi + plen;
#pragma unroll
                for (int t = 0; t < NS; ++t) {
                  if (t * SW + SW - 1 >= ic_lo + plen && t * SW <= ic_hi + plen && (t * SW < zlo || t * SW + SW - 1 > zhi)) {
#pragma unroll
                    for (int jj = 0; jj < B; ++jj) {
                      const int kbA = t * SW + 2 * jj + lane_kb;
                      const bool kA = kbA >= zlo && kbA <= zhi, kB = kbA + 1 >= zlo && kbA + 1 <= zhi;
                      Ir[t * B + jj] &= (kA ? 0xFFFFu : 0u) | (kB ? 0xFFFF0000u : 0u);
                    }
                  }
                }
              }
            }
          }
          {
            int mc_lo = n_mm ? INF : slo[IX], mc_hi = n_mm ? -INF : shi[IX];
            if (has_i) { mc_lo = min(mc_lo, ic_lo); mc_hi = max(mc_hi, ic_hi); }
            if (has_d) { mc_lo = min(mc_lo, ndlo); mc_hi = max(mc_hi, ndhi); }
            nmlo = mc_lo; nmhi = mc_hi;
            if (mc_hi > tlen - plen) {
              const int f = find_last(Mn, mc_lo + plen, mc_hi + plen, nonzero);
              if (f < 0) { nmlo = 1; nmhi = -1; }
              else {
                nmhi = f - plen;
                if (mc_lo > tlen - plen) nmlo = find_first(Mn, mc_lo + plen, mc_hi + plen, nonzero) - plen;
              }
            }
          }
        }
        // ---- rotate the ring of this class: the new level (in the slot of M[s - R]) becomes its newest.  Element by element through
        //      opaque(): see there.
        if constexpr (DEP > 1) {
#pragma unroll
          for (int p = 0; p < NP; ++p) {
            const uint32_t newest = opaque(Ms[DEP - 1][p]);
#pragma unroll
            for (int d = DEP - 1; d >= 1; --d) Ms[d][p] = opaque(Ms[d - 1][p]);
            Ms[0][p] = newest;
          }
#pragma unroll
          for (int d = DEP - 1; d >= 1; --d) { slo[d] = slo[d - 1]; shi[d] = shi[d - 1]; }
        }
        slo[0] = nmlo; shi[0] = nmhi; ilo = nilo; ihi = nihi; dlo = ndlo; dhi = ndhi;
        // ---- early rejection (at the levels where it can first succeed, see next_check).  A cell (v bases of the pattern consumed, at most c of them matched, text position
        //      h = v + k) can end in an alignment of at most c + min(plen - v, tlen - h) matches -- a match needs a base of both -- and
        //      no step raises that number: a match raises c, v and h together, a mismatch v and h, a deleted base v, an inserted base h.
        //      Every alignment that ends beyond this level continues from a cell of the last six M levels or of the current I / D: if
        //      none of them can still reach min_matches, the optimal alignment -- whichever it is -- has fewer matches than the caller
        //      asks for, and the remaining levels (the widest ones: the work of a level grows with the score) need not be computed.
        //      Per 16-bit cell: deficit = (v - c) + max(0, k - (tlen - plen)), 0xFFFF for NULL; at most plen - min_matches to go on.
        if (a.early_reject && s >= next_check) {
          uint32_t dmin = 0xFFFFFFFFu;
#pragma unroll
          for (int t = 0; t < NS; ++t) {
            const bool top = t * SW + SW - 1 > tlen;  // (uniform) diagonals above tlen - plen: the text ends before the pattern does
#pragma unroll
            for (int jj = 0; jj < B; ++jj) {
              const int p = t * B + jj;
              uint32_t dm = 0xFFFFFFFFu;
              auto take = [&](uint32_t x) {
                const uint32_t cnt = x & 0x00FF00FFu, v1 = (x >> 8) & 0x00FF00FFu;
                dm = rpk_min(dm, rpk_sub(rpk_sub(v1, cnt), 0x00010001u));
              };
#pragma unroll
              for (int g = 0; g < G; ++g)
#pragma unroll
                for (int d = 0; d < DEP; ++d) take(Mr[g][d][p]);
              take(Ir[p]); take(Dr[p]);
              if (top) {
                const int over = t * SW + 2 * jj + lane_kb - tlen;  // biased diagonal of the low half, minus tlen
                dm = rpk_add_sat(dm, (uint32_t)min(max(over, 0), 0x7FFF) | ((uint32_t)min(max(over + 1, 0), 0x7FFF) << 16));
              }
              dmin = rpk_min(dmin, dm);
            }
          }
          uint32_t best = min(dmin & 0xFFFFu, dmin >> 16);  // the smallest deficit of this lane's cells
          if (!__builtin_amdgcn_ballot_w64((int)best <= plen - a.min_matches)) { early = true; return false; }
          // When to look again: in 16 levels, or -- when the smallest deficit m is close to the limit -- after the X (limit - m + 1)
          // levels the cheapest way of spoiling bases (mismatches, one every X levels) takes to push that cell over it.  (Only the
          // schedule rests on this estimate; a test is exact whenever it runs.)  A text without the piece is given up at level 52
          // with TRGT's 2,5,1 and 90 % instead of at level 64.
          if (a.early_reject == 2) {
#pragma unroll
            for (int dd = 32; dd >= 1; dd >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, dd));
          }
          next_check = s + (a.early_reject == 2 ? max(2, min(16, X * (plen - a.min_matches - rfl_i((int)best) + 1))) : 16);  // (uniform: an SGPR -- as a vector value it was spilled, a scratch load per level)
        }
        return true;
      };
      for (;;) {  // levels 1, 2, 3, ...: class 1, class 0, class 1, ... (one class when G = 1)
        if constexpr (G == 2) { if (!level(std::integral_constant<int, 1>())) break; }
        if (!level(std::integral_constant<int, 0>())) break;
      }
      cells_acc += cells;
      if (early) { keep = 0; score_out = INT32_MIN + 1; bound_out = a.min_matches - 1; band = 0x40000000u | (uint32_t)min(s, 0x3FFFFFFF); }  // (no alignment of penalty <= s exists: the run had not ended)
      else if (bail || !done) keep = 1;
      else {
        // the first terminating diagonal (wavefront_extend walks k upwards and stops at the first one)
        int best = 1 << 20; uint32_t best_reg = 0;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const int C = (p / B) * SW + 2 * (p % B);
          // (the newest level: of the class of s)
          const uint32_t x = (G == 2 && (s & 1)) ? opaque(Mr[G - 1][0][p]) : opaque(Mr[0][0][p]);
          const unsigned long long mA = __builtin_amdgcn_ballot_w64(((x >> 8) & 0xFFu) == (uint32_t)term_v);
          const unsigned long long mB = __builtin_amdgcn_ballot_w64((x >> 24) == (uint32_t)term_v);
          if (mA) {
            const int l = (int)__builtin_ctzll(mA), kb = l * LW + C;
            if (kb < best) { best = kb; best_reg = (uint32_t)__builtin_amdgcn_readlane((int)x, l) & 0xFFFFu; }
          }
          if (mB) {
            const int l = (int)__builtin_ctzll(mB), kb = l * LW + C + 1;
            if (kb < best) { best = kb; best_reg = (uint32_t)__builtin_amdgcn_readlane((int)x, l) >> 16; }
          }
        }
        score_out = -s;
        bound_out = (int)(best_reg & 0xFFu);
        keep = bound_out >= a.min_matches ? 1 : 0;
        // for the back-tracing launch over what is kept: the penalty and the (biased) diagonal the alignment ends on are all it takes
        // to confine that run to a band (heavy_band_kernel, spans.hip)
        if (s < 0x8000 && best < 0x10000) band = 0x80000000u | ((uint32_t)s << 16) | (uint32_t)best;
      }
    }
    if (lane == 0) {
      const uint32_t o = jp->out_index;
      if (a.score) a.score[o] = score_out;
      if (a.bound) a.bound[o] = bound_out;
      if (a.keep) a.keep[o] = (uint8_t)keep;
      if (a.band) a.band[o] = band;
      if (keep && a.keep_jobs) { JobDev kj = *jp; kj.pad = (band >> 31) ? band : 0u; a.keep_jobs[atomicAdd(a.keep_count, 1u)] = kj; }
    }
    kept_acc += (unsigned long long)keep;  // (uniform, like cells_acc: a scalar)
  }
  if (lane == 0 && a.cells_out && cells_acc) atomicAdd(a.cells_out, cells_acc);
  if (lane == 0 && a.cells_out && kept_acc) atomicAdd(a.cells_out + 1, kept_acc);  // [1]: alignments kept
}

}  // namespace wfa

int flank_filter_max_tlen(int flank_len) {
  const int64_t t = 12 * 128 - (int64_t)flank_len - 1;  // the largest instantiation: 1536 diagonals
  return flank_len >= 1 && flank_len <= 254 ? (int)std::max<int64_t>(t, 0) : 0;
}

int flank_filter_launch(trgt_hip_ctx* c, const FilterLaunch& L) {
  using namespace wfa;
  if (L.max_plen < 1 || L.max_plen > 254) return fail(c, TRGT_ERR_UNSUPPORTED, "flank filter: pattern length %lld outside 1..254", (long long)L.max_plen);
  FilterArgs a;
  std::memset(&a, 0, sizeof a);
  a.jobs = L.jobs_dev; a.n_jobs_dev = L.n_jobs_dev; a.n_jobs = (uint32_t)L.n_jobs_host;
  a.pat_base = L.pat_base; a.txt_base = L.txt_base;
  a.min_matches = L.min_matches; a.early_reject = L.early_reject ? (c->knobs.early_adaptive ? 2 : 1) : 0;
  a.keep_jobs = L.keep_jobs; a.keep_count = L.keep_count;
  a.score = L.score; a.bound = L.bound; a.keep = L.keep; a.band = L.band;
  void* d_counter = nullptr; void* d_cells = nullptr;
  int rc;
  if ((rc = dev_get_zeroed(c, L.set ? S_FLT_COUNTER_B : S_FLT_COUNTER, 16, &d_counter, c->stream)) || (rc = dev_get_zeroed(c, L.set ? S_FLT_CELLS_B : S_FLT_CELLS, 16, &d_cells, c->stream))) return rc;
  a.counter = (unsigned int*)d_counter; a.cells_out = (unsigned long long*)d_cells;
  a.diag_lo = INT32_MIN; a.diag_hi = INT32_MAX;
  // instantiation by the number of diagonals the longest text of the launch needs (jobs that do not fit are kept unseen)
  const int64_t diag = L.max_plen + L.max_tlen + 1;
  // (nine strips of 128 diagonals for texts that need up to 1152: 8 % faster there than the five strips of 256, which carry 128 dead ones)
  const bool targeted = L.mism == 1 && L.gapo == 0 && L.gape == 1;  // the other preset: 1,0,1 (cli.rs:271-280)
  if (!targeted && !(L.mism == 2 && L.gapo == 5 && L.gape == 1)) return fail(c, TRGT_ERR_UNSUPPORTED, "flank filter: no instantiation for penalties %d,%d,%d", L.mism, L.gapo, L.gape);
  void (*fn)(const FilterArgs) = diag <= 4 * 256 ? wfa_filter_kernel<4, 2> : diag <= 9 * 128 ? wfa_filter_kernel<9, 1> : diag <= 5 * 256 ? wfa_filter_kernel<5, 2> : wfa_filter_kernel<6, 2>;
  void (*fn4)(const FilterArgs) = wfa_filter_kernel<4, 2>;
  if (targeted) { fn = diag <= 4 * 256 ? wfa_filter_kernel<4, 2, 1, 1> : diag <= 5 * 256 ? wfa_filter_kernel<5, 2, 1, 1> : wfa_filter_kernel<6, 2, 1, 1>; fn4 = wfa_filter_kernel<4, 2, 1, 1>; }
  if (const int f = targeted ? 0 : c->knobs.filter_force) {  // developer probe (tools/filter_inst_probe.py; TRGT_FILTER_FORCE in `make DEV=1` builds): one instantiation for the whole launch
    fn = f == 71 ? wfa_filter_kernel<7, 1> : f == 91 ? wfa_filter_kernel<9, 1> : f == 42 ? wfa_filter_kernel<4, 2> : f == 52 ? wfa_filter_kernel<5, 2> : fn;
  }
  // One-wave workgroups: resident waves per CU from the kernel's own register and LDS footprint (the occupancy query answers
  // per SIMD for 64-thread blocks; measured: it said 3 where 12 waves fit a CU)
  auto grid_for = [&](void (*f)(const FilterArgs)) -> int64_t {
    int occ = 8;
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, (const void*)f) == hipSuccess) {
      const int alloc = std::max(8, (fa.numRegs + 7) / 8 * 8);
      const int per_simd = std::max(1, std::min(8, 512 / alloc));
      const int by_lds = fa.sharedSizeBytes > 0 ? (int)((160u * 1024u) / (unsigned)fa.sharedSizeBytes) : 32;
      occ = std::max(1, std::min(std::min(4 * per_simd, by_lds), 32));
    } else (void)hipGetLastError();
    if (c->knobs.filter_per_cu > 0) occ = c->knobs.filter_per_cu;
    if (c->knobs.debug) fprintf(stderr, "[filter] diagonals %lld occupancy %d\n", (long long)diag, occ);
    return std::max<int64_t>(1, std::min<int64_t>((int64_t)c->num_cus * occ, L.n_jobs_host));
  };
  const int64_t grid = grid_for(fn);
  KTimer t(c, L.timer_slot);
  // Two launches over the same list when the longest text needs more than four strips: most texts are shorter than the longest (on
  // the 10k-locus batch 80 % of the expensive alignments fit 1024 diagonals, the rest needs up to 1280), and a job in the four-strip
  // instantiation runs a fifth fewer instructions per level than in the five-strip one.  (Skipping the dead strips inside one kernel
  // was slower: DESIGN.md 5.)  The long ones first; each launch claims the whole list with a counter of its own and passes over the
  // other's jobs.
  const bool split = diag > 4 * 256 && L.n_jobs_host >= 1024 && !c->knobs.filter_one_launch && !c->knobs.filter_force;
  if (split) {
    // (a ladder of four instantiations -- 1536 / 1280 / 1152 / 1024 diagonals -- was no better than these two on the catalog mix:
    //  every launch has a tail)
    FilterArgs b = a;
    b.diag_lo = 4 * 256;
    FilterArgs s4 = a;
    s4.diag_hi = 4 * 256; s4.counter = a.counter + 1;
    // The two launches NEXT TO each other (round 3): the long texts are few (a fifth of the jobs: fewer than the launch has
    // workgroups, so it lasts as long as its longest alignment and leaves most of the GPU idle); the launch over the others fills it.
    // (the timer of the pass brackets both: its stop event follows the join on this stream.  Not in a pool: with several contexts on
    //  the GPU it is full anyway, and the extra stream cost 2-3 % of the pool's throughput -- 2.37 against 2.30 M loci/s)
    const bool side = !c->knobs.filter_serial && (!c->in_pool || c->knobs.filter_side);
    if (side) {
      if (!c->stream_flt) TRGT_HIP_TRY(c, trgt::make_side_stream(c, &c->stream_flt));
      if (!c->ev_flt_a) TRGT_HIP_TRY(c, hipEventCreateWithFlags(&c->ev_flt_a, hipEventDisableTiming));
      if (!c->ev_flt_b) TRGT_HIP_TRY(c, hipEventCreateWithFlags(&c->ev_flt_b, hipEventDisableTiming));
      TRGT_HIP_TRY(c, hipEventRecord(c->ev_flt_a, c->stream));           // the job list and the cleared counters
      TRGT_HIP_TRY(c, hipStreamWaitEvent(c->stream_flt, c->ev_flt_a, 0));
      hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(64), 0, c->stream_flt, b);
      TRGT_HIP_TRY(c, hipEventRecord(c->ev_flt_b, c->stream_flt));
      hipLaunchKernelGGL(fn4, dim3((unsigned)grid_for(fn4)), dim3(64), 0, c->stream, s4);
      TRGT_HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_flt_b, 0));
    } else {
      hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(64), 0, c->stream, b);
      hipLaunchKernelGGL(fn4, dim3((unsigned)grid_for(fn4)), dim3(64), 0, c->stream, s4);
    }
  } else hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(64), 0, c->stream, a);
  TRGT_HIP_TRY(c, hipGetLastError());
  t.stop(0);
  if (!L.set) c->last_filter_cells_dev = d_cells;
  return TRGT_OK;
}

}  // namespace trgt

using namespace trgt;

extern "C" int trgt_flank_filter_batch(trgt_hip_ctx* c, const trgt_span_params* p, int64_t n_jobs, const uint8_t* seqs,
                                       const uint64_t* pat_off, const uint32_t* pat_len, const uint64_t* txt_off,
                                       const uint32_t* txt_len, int32_t min_matches, int32_t early_reject, int32_t* score,
                                       int32_t* match_bound, uint8_t* keep, int64_t* offsets_computed) {
  if (!c) return TRGT_ERR_INVALID;
  if (!p || n_jobs < 0 || (n_jobs > 0 && (!seqs || !pat_off || !pat_len || !txt_off || !txt_len)))
    return fail(c, TRGT_ERR_INVALID, "trgt_flank_filter_batch: null argument");
  if (!(p->mism == 2 && p->gapo == 5 && p->gape == 1) && !(p->mism == 1 && p->gapo == 0 && p->gape == 1))
    return fail(c, TRGT_ERR_UNSUPPORTED, "trgt_flank_filter_batch: only --aln-scoring 2,5,1 and 1,0,1 have a filter kernel");
  if (offsets_computed) *offsets_computed = 0;
  if (n_jobs == 0) return TRGT_OK;
  if (n_jobs > 0xFFFFFFF0ll) return fail(c, TRGT_ERR_UNSUPPORTED, "trgt_flank_filter_batch: too many jobs");
  TRGT_HIP_TRY(c, hipSetDevice(c->device));
  std::vector<JobDev> jobs((size_t)n_jobs);
  FilterLaunch L;
  uint64_t seq_total = 0;
  for (int64_t j = 0; j < n_jobs; ++j) {
    JobDev& jd = jobs[(size_t)j];
    jd.pat_off = pat_off[j]; jd.txt_off = txt_off[j]; jd.pat_len = pat_len[j]; jd.txt_len = txt_len[j]; jd.out_index = (uint32_t)j;
    jd.cigar_off = 0; jd.ops_off = 0; jd.pad = 0;
    // (the launch is planned for the jobs the kernel can judge; the others are kept unseen)
    if (pat_len[j] >= 1 && pat_len[j] <= 254 && (int64_t)txt_len[j] <= flank_filter_max_tlen((int)pat_len[j])) {
      L.max_plen = std::max<int64_t>(L.max_plen, pat_len[j]); L.max_tlen = std::max<int64_t>(L.max_tlen, txt_len[j]);
    }
    seq_total = std::max<uint64_t>(seq_total, std::max(pat_off[j] + pat_len[j], txt_off[j] + txt_len[j]));
  }
  if (L.max_plen < 1) L.max_plen = 1;
  int rc;
  const uint8_t* d_seq = nullptr;
  if ((rc = dev_in(c, S_FLT_SEQ, seqs, (size_t)seq_total, &d_seq))) return rc;
  void* d_jobs = nullptr;
  if ((rc = dev_get(c, S_FLT_JOBS, jobs.size() * sizeof(JobDev), &d_jobs))) return rc;
  TRGT_HIP_TRY(c, hipMemcpyAsync(d_jobs, jobs.data(), jobs.size() * sizeof(JobDev), hipMemcpyHostToDevice, c->stream));
  DevOut<int32_t> o_score, o_bound; DevOut<uint8_t> o_keep;
  if ((rc = o_score.init(c, S_FLT_SCORE, score, (size_t)n_jobs)) || (rc = o_bound.init(c, S_FLT_BOUND, match_bound, (size_t)n_jobs)) ||
      (rc = o_keep.init(c, S_FLT_KEEP, keep, (size_t)n_jobs)))
    return rc;
  L.jobs_dev = (const JobDev*)d_jobs; L.n_jobs_host = n_jobs; L.pat_base = d_seq; L.txt_base = d_seq;
  L.count_offsets = offsets_computed != nullptr || c->timing; L.mism = p->mism; L.gapo = p->gapo; L.gape = p->gape;
  L.min_matches = min_matches; L.early_reject = early_reject != 0; L.score = o_score.dev; L.bound = o_bound.dev; L.keep = o_keep.dev;
  if ((rc = flank_filter_launch(c, L))) return rc;
  if ((rc = o_score.finish(c)) || (rc = o_bound.finish(c)) || (rc = o_keep.finish(c))) return rc;
  unsigned long long cells = 0;
  { const int d2h_rc = trgt::d2h(c, &cells, c->last_filter_cells_dev, 8, c->stream); if (d2h_rc) return d2h_rc; }
  TRGT_HIP_TRY(c, trgt::stream_wait(c, c->stream));
  if (offsets_computed) *offsets_computed = (int64_t)cells;
  if (c->timing) c->k_cells[TRGT_K_WFA_FILTER] += (int64_t)cells;
  return TRGT_OK;
}
