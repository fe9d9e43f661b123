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
// neighbouring register, or -- at the two ends of a lane's block -- the neighbouring lane (one DPP wave shift per strip, source
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

#include "wfa_host.hpp"

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
constexpr int PWN = 264;       // pattern windows: index v + 1 (0: the NULL window), plen <= 254
constexpr int TW_EXTRA = 264;  // text windows: index (v + 1) + kb, i.e. text position + plen + 1
constexpr int SMAX = 300;      // more levels than any flank alignment can take (a pattern of <= 254 bases is deleted for <= 5 + 254)

// bytes that collide with the sentinels make the job "dirty": it is handed to the exact kernel unseen
__device__ __forceinline__ uint32_t has_zero_byte(uint32_t v) { return (v - 0x01010101u) & ~v & 0x80808080u; }
__device__ __forceinline__ uint32_t dirty_dword(uint32_t x) {
  return has_zero_byte(x ^ PAT_PAD) | has_zero_byte(x ^ TXT_PAD) | has_zero_byte(~x);
}

}  // namespace

template <int NS, int B>
__global__ void __launch_bounds__(64) wfa_filter_kernel(const FilterArgs a) {
  constexpr int NP = NS * B, SW = 128 * B, D = NS * SW, TWN = D + TW_EXTRA;
  __shared__ uint32_t lds[PWN + TWN];
  uint32_t* const Pw = lds;
  uint32_t* const Tw = lds + PWN;
  const int lane = (int)threadIdx.x;
  const uint32_t* const twl = Tw + lane * 2 * B;  // + (v + 1) + C_p: the window of diagonal kb = C_p + lane * 2B at offset v + k
  const uint32_t n_jobs = a.n_jobs_dev ? *a.n_jobs_dev : a.n_jobs;
  unsigned long long cells_acc = 0, kept_acc = 0;

  // 4-byte sliding windows of `len` bytes at src into W[base + i] (i = 0 .. n_win - 1), bytes beyond the sequence = pad
  auto stage = [&](const uint8_t* __restrict__ src, int len, uint32_t* __restrict__ W, int n_win, uint32_t pad, uint32_t& dirty) {
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
      W[i0] = d0;
      if (i0 + 1 < n_win) W[i0 + 1] = __builtin_amdgcn_alignbyte(d1, d0, 1);
      if (i0 + 2 < n_win) W[i0 + 2] = __builtin_amdgcn_alignbyte(d1, d0, 2);
      if (i0 + 3 < n_win) W[i0 + 3] = __builtin_amdgcn_alignbyte(d1, d0, 3);
    }
  };

  for (;;) {
    uint32_t j = 0;
    if (lane == 0) j = atomicAdd(a.counter, 1u);
    j = (uint32_t)rfl_i((int)j);
    if (j >= n_jobs) break;
    const JobDev job = a.jobs[j];
    const int plen = rfl_i((int)job.pat_len), tlen = rfl_i((int)job.txt_len);
    int keep = 0, score_out = INT32_MIN, bound_out = -1;
    const bool fits = plen >= 1 && plen <= 254 && tlen >= plen && tlen + plen + 1 <= D;
    uint32_t dirty = 0;
    if (fits) {
      // ---- the two sequences as sliding windows
      __builtin_amdgcn_s_barrier();  // (one wave: orders the LDS reads of the previous job before these writes)
      if (lane == 0) Pw[0] = NULL_WIN;
      stage(a.pat_base + job.pat_off, plen, Pw + 1, PWN - 1, PAT_PAD, dirty);
      for (int i = lane; i < plen + 1; i += 64) Tw[i] = TXT_PAD;  // text positions < 0: only NULL cells look there
      stage(a.txt_base + job.txt_off, tlen, Tw + plen + 1, TWN - (plen + 1), TXT_PAD, dirty);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    }
    const bool is_dirty = __builtin_amdgcn_ballot_w64(dirty != 0u) != 0ull;
    if (!fits || is_dirty) keep = 1;
    else {
      // extension of both cells of a packed pair; C = the pair's compile-time diagonal offset inside the lane's window pointer
      auto extend_pair = [&](uint32_t key, int C) -> uint32_t {
        uint32_t va = (key >> 8) & 0xFFu, vb = key >> 24;
        uint32_t na = min(ffbl_or_m1(Pw[va] ^ twl[va + C]) >> 3, 4u);
        uint32_t nb = min(ffbl_or_m1(Pw[vb] ^ twl[vb + C + 1]) >> 3, 4u);
        uint32_t ta = na, tb = nb;
        bool ca = na == 4u, cb = nb == 4u;
        while (__builtin_amdgcn_ballot_w64(ca || cb)) {
          if (ca) { na = min(ffbl_or_m1(Pw[va + ta] ^ twl[va + ta + C]) >> 3, 4u); ta += na; ca = na == 4u; }
          if (cb) { nb = min(ffbl_or_m1(Pw[vb + tb] ^ twl[vb + tb + C + 1]) >> 3, 4u); tb += nb; cb = nb == 4u; }
        }
        return key + (ta | (tb << 16)) * 0x0101u;
      };
      uint32_t Mr[6][NP], Ir[NP], Dr[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int d = 0; d < 6; ++d) Mr[d][p] = 0u;
        Ir[p] = 0u; Dr[p] = 0u;
      }
      // trimmed ranges of the live wavefronts in k (not biased); null = (1, -1) as in the library
      int mlo[6], mhi[6], ilo = 1, ihi = -1, dlo = 1, dhi = -1;
#pragma unroll
      for (int d = 0; d < 6; ++d) { mlo[d] = 1; mhi[d] = -1; }
      const int lane_kb = lane * 2 * B;
      const int term_v = plen + 1;
      // ---- level 0: M[0][k] = k for k in [0, tlen] (v = 0), then extended
      uint32_t tmax = 0;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int C = (p / B) * SW + 2 * (p % B);
        const int kbA = C + lane_kb;
        uint32_t key = ((unsigned)(kbA - plen) <= (unsigned)tlen ? 0x0100u : 0u) | ((unsigned)(kbA + 1 - plen) <= (unsigned)tlen ? 0x01000000u : 0u);
        key = extend_pair(key, C);
        Mr[0][p] = key;
        tmax = rpk_max(tmax, key);
      }
      mlo[0] = 0; mhi[0] = tlen;
      unsigned long long cells = (unsigned long long)tlen + 1ull;
      int s = 0, num_null = 0;
      bool done = false, bail = false;
      for (;;) {
        // ---- termination (wavefront_termination_endsfree with pattern_end_free = 0, text_end_free = tlen): v == plen
        const bool tA = ((tmax >> 8) & 0xFFu) == (uint32_t)term_v, tB = (tmax >> 24) == (uint32_t)term_v;
        if (__builtin_amdgcn_ballot_w64(tA || tB)) { done = true; break; }
        ++s;
        if (s > SMAX) { bail = true; break; }
        const bool n_mm = mlo[1] > mhi[1], n_mo = mlo[5] > mhi[5], n_ie = ilo > ihi, n_de = dlo > dhi;
        uint32_t Mn[NP], In[NP], Dn[NP];
        int nmlo = 1, nmhi = -1, nilo = 1, nihi = -1, ndlo = 1, ndhi = -1;
        tmax = 0;
        if (n_mm && n_mo && n_ie && n_de) {
          ++num_null;
          if (num_null > 7) { bail = true; break; }  // (cannot happen with a free text; the exact kernel decides)
#pragma unroll
          for (int p = 0; p < NP; ++p) { Mn[p] = 0u; In[p] = 0u; Dn[p] = 0u; }
        } else {
          num_null = 0;
          // wavefront_compute_limits_input (null wavefronts take part with lo = 1, hi = -1, as in the library)
          const int lo = min(min(mlo[1], mlo[5] - 1), min(ilo + 1, dlo - 1));
          const int hi = max(max(mhi[1], mhi[5] + 1), max(ihi + 1, dhi - 1));
          cells += 3ull * (unsigned long long)max(0, hi - lo + 1);
          const int lob = lo + plen, hib = hi + plen;  // biased
          const int lane_bnd = tlen + 1 + plen - lane_kb;  // v + 1 <= lane_bnd - C  <=>  offset <= tlen on diagonal kb = C + lane_kb
#pragma unroll
          for (int t = 0; t < NS; ++t) {
            const int k0 = t * SW, k1 = k0 + SW - 1;
            if (k1 < lob || k0 > hib) {  // (uniform) nothing of this strip is inside the limits: every source is NULL there
#pragma unroll
              for (int jj = 0; jj < B; ++jj) { Mn[t * B + jj] = 0u; In[t * B + jj] = 0u; Dn[t * B + jj] = 0u; }
              continue;
            }
            const uint32_t mo_pe = shr_from(Mr[5][t * B + B - 1], t > 0 ? Mr[5][(t - 1) * B + B - 1] : 0u);
            const uint32_t ie_pe = shr_from(Ir[t * B + B - 1], t > 0 ? Ir[(t - 1) * B + B - 1] : 0u);
            const uint32_t mo_ne = shl_from(Mr[5][t * B], t < NS - 1 ? Mr[5][(t + 1) * B] : 0u);
            const uint32_t de_ne = shl_from(Dr[t * B], t < NS - 1 ? Dr[(t + 1) * B] : 0u);
            const bool top = k1 > tlen;  // (uniform) diagonals k > tlen - plen: an offset can pass the end of the text
#pragma unroll
            for (int jj = 0; jj < B; ++jj) {
              const int p = t * B + jj, C = t * SW + 2 * jj;
              const uint32_t mo_c = Mr[5][p], mo_p = jj ? Mr[5][p - 1] : mo_pe, mo_n = jj < B - 1 ? Mr[5][p + 1] : mo_ne;
              const uint32_t ie_c = Ir[p], ie_p = jj ? Ir[p - 1] : ie_pe;
              const uint32_t de_c = Dr[p], de_n = jj < B - 1 ? Dr[p + 1] : de_ne;
              const uint32_t ins = rpk_max(__builtin_amdgcn_alignbit(mo_c, mo_p, 16), __builtin_amdgcn_alignbit(ie_c, ie_p, 16));
              const uint32_t del = inc_v_nz(rpk_max(__builtin_amdgcn_alignbit(mo_n, mo_c, 16), __builtin_amdgcn_alignbit(de_n, de_c, 16)));
              const uint32_t mis = inc_v_nz(Mr[1][p]);
              uint32_t mxp = rpk_max(del, rpk_max(mis, ins));
              if (top) {  // "adjust offset out of boundaries": offset > tlen -> NULL
                const int bA = lane_bnd - C;
                const bool okA = (int)((mxp >> 8) & 0xFFu) <= bA, okB = (int)(mxp >> 24) <= bA - 1;
                mxp = (okA ? mxp & 0xFFFFu : 0u) | (okB ? mxp & 0xFFFF0000u : 0u);
              }
              const uint32_t mq = extend_pair(mxp, C);
              Mn[p] = mq; In[p] = ins; Dn[p] = del;
              tmax = rpk_max(tmax, mq);
            }
          }
          // ---- wavefront_compute_trim_ends, restated: D is always in bounds; I and M are, below k = tlen - plen (+ 1)
          constexpr int INF = 1 << 20;
          const int so_lo = n_mo ? INF : mlo[5], so_hi = n_mo ? -INF : mhi[5];
          const bool has_d = !n_mo || !n_de, has_i = !n_mo || !n_ie;
          if (has_d) { ndlo = min(so_lo, n_de ? INF : dlo) - 1; ndhi = max(so_hi, n_de ? -INF : dhi) - 1; }
          int ic_lo = INF, ic_hi = -INF;  // non-NULL I cells
          // first / last cell of a component for which pred holds, inside biased [kb_lo, kb_hi]; -1: none
          auto find_last = [&](const uint32_t (&R)[NP], int kb_lo, int kb_hi, auto pred) -> int {
            int best = -1;
#pragma unroll
            for (int t = NS - 1; t >= 0; --t) {
              if (best < 0 && t * SW <= kb_hi && t * SW + SW - 1 >= kb_lo) {
#pragma unroll
                for (int jj = 0; jj < B; ++jj) {
                  bool va, vb;
                  pred(R[t * B + jj], t * SW + 2 * jj, va, vb);
                  const unsigned long long mA = __builtin_amdgcn_ballot_w64(va), mB = __builtin_amdgcn_ballot_w64(vb);
                  if (mA) best = max(best, (63 - (int)__builtin_clzll(mA)) * 2 * B + t * SW + 2 * jj);
                  if (mB) best = max(best, (63 - (int)__builtin_clzll(mB)) * 2 * B + t * SW + 2 * jj + 1);
                }
              }
            }
            return best;
          };
          auto find_first = [&](const uint32_t (&R)[NP], int kb_lo, int kb_hi, auto pred) -> int {
            int best = INF;
#pragma unroll
            for (int t = 0; t < NS; ++t) {
              if (best == INF && t * SW <= kb_hi && t * SW + SW - 1 >= kb_lo) {
#pragma unroll
                for (int jj = 0; jj < B; ++jj) {
                  bool va, vb;
                  pred(R[t * B + jj], t * SW + 2 * jj, va, vb);
                  const unsigned long long mA = __builtin_amdgcn_ballot_w64(va), mB = __builtin_amdgcn_ballot_w64(vb);
                  if (mA) best = min(best, (int)__builtin_ctzll(mA) * 2 * B + t * SW + 2 * jj);
                  if (mB) best = min(best, (int)__builtin_ctzll(mB) * 2 * B + t * SW + 2 * jj + 1);
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
          if (has_i) {
            ic_lo = min(so_lo, n_ie ? INF : ilo) + 1; ic_hi = max(so_hi, n_ie ? -INF : ihi) + 1;
            nilo = ic_lo; nihi = ic_hi;
            if (ic_hi > tlen - plen + 1) {
              const int f = find_last(In, ic_lo + plen, ic_hi + plen, i_valid);
              if (f < 0) { nilo = 1; nihi = -1; }
              else {
                nihi = f - plen;
                if (ic_lo > tlen - plen + 1) nilo = find_first(In, ic_lo + plen, ic_hi + plen, i_valid) - plen;
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
                      In[t * B + jj] &= (kA ? 0xFFFFu : 0u) | (kB ? 0xFFFF0000u : 0u);
                    }
                  }
                }
              }
            }
          }
          {
            int mc_lo = n_mm ? INF : mlo[1], mc_hi = n_mm ? -INF : mhi[1];
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
        // ---- rotate
#pragma unroll
        for (int d = 5; d >= 1; --d) {
          mlo[d] = mlo[d - 1]; mhi[d] = mhi[d - 1];
#pragma unroll
          for (int p = 0; p < NP; ++p) Mr[d][p] = Mr[d - 1][p];
        }
        mlo[0] = nmlo; mhi[0] = nmhi; ilo = nilo; ihi = nihi; dlo = ndlo; dhi = ndhi;
#pragma unroll
        for (int p = 0; p < NP; ++p) { Mr[0][p] = Mn[p]; Ir[p] = In[p]; Dr[p] = Dn[p]; }
      }
      cells_acc += cells;
      if (bail || !done) keep = 1;
      else {
        // the first terminating diagonal (wavefront_extend walks k upwards and stops at the first one)
        int best = 1 << 20; uint32_t best_reg = 0;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const int C = (p / B) * SW + 2 * (p % B);
          const uint32_t x = Mr[0][p];
          const unsigned long long mA = __builtin_amdgcn_ballot_w64(((x >> 8) & 0xFFu) == (uint32_t)term_v);
          const unsigned long long mB = __builtin_amdgcn_ballot_w64((x >> 24) == (uint32_t)term_v);
          if (mA) {
            const int l = (int)__builtin_ctzll(mA), kb = l * 2 * B + C;
            if (kb < best) { best = kb; best_reg = (uint32_t)__builtin_amdgcn_readlane((int)x, l) & 0xFFFFu; }
          }
          if (mB) {
            const int l = (int)__builtin_ctzll(mB), kb = l * 2 * B + C + 1;
            if (kb < best) { best = kb; best_reg = (uint32_t)__builtin_amdgcn_readlane((int)x, l) >> 16; }
          }
        }
        score_out = -s;
        bound_out = (int)(best_reg & 0xFFu);
        keep = bound_out >= a.min_matches ? 1 : 0;
      }
    }
    if (lane == 0) {
      const uint32_t o = job.out_index;
      if (a.score) a.score[o] = score_out;
      if (a.bound) a.bound[o] = bound_out;
      if (a.keep) a.keep[o] = (uint8_t)keep;
      if (keep && a.keep_jobs) a.keep_jobs[atomicAdd(a.keep_count, 1u)] = job;
      kept_acc += (unsigned long long)keep;
    }
  }
  if (lane == 0 && a.cells_out && cells_acc) atomicAdd(a.cells_out, cells_acc);
  if (lane == 0 && a.cells_out && kept_acc) atomicAdd(a.cells_out + 1, kept_acc);  // [1]: alignments kept
}

}  // namespace wfa

int flank_filter_max_tlen(int flank_len) {
  const int64_t t = 6 * 256 - (int64_t)flank_len - 1;  // the largest instantiation: 1536 diagonals
  return flank_len >= 1 && flank_len <= 254 ? (int)std::max<int64_t>(t, 0) : 0;
}

int flank_filter_launch(trgt_hip_ctx* c, const FilterLaunch& L) {
  using namespace wfa;
  if (L.max_plen < 1 || L.max_plen > 254) return fail(c, TRGT_ERR_UNSUPPORTED, "flank filter: pattern length %lld outside 1..254", (long long)L.max_plen);
  FilterArgs a;
  std::memset(&a, 0, sizeof a);
  a.jobs = L.jobs_dev; a.n_jobs_dev = L.n_jobs_dev; a.n_jobs = (uint32_t)L.n_jobs_host;
  a.pat_base = L.pat_base; a.txt_base = L.txt_base;
  a.min_matches = L.min_matches;
  a.keep_jobs = L.keep_jobs; a.keep_count = L.keep_count;
  a.score = L.score; a.bound = L.bound; a.keep = L.keep;
  void* d_counter = nullptr; void* d_cells = nullptr;
  int rc;
  if ((rc = dev_get(c, S_FLT_COUNTER, 16, &d_counter)) || (rc = dev_get(c, S_FLT_CELLS, 16, &d_cells))) return rc;
  TRGT_HIP_TRY(c, hipMemsetAsync(d_counter, 0, 16, c->stream));
  TRGT_HIP_TRY(c, hipMemsetAsync(d_cells, 0, 16, c->stream));
  a.counter = (unsigned int*)d_counter; a.cells_out = (unsigned long long*)d_cells;
  // instantiation by the number of diagonals the longest text of the launch needs (jobs that do not fit are kept unseen)
  const int64_t diag = L.max_plen + L.max_tlen + 1;
  void (*fn)(const FilterArgs) = diag <= 4 * 256 ? wfa_filter_kernel<4, 2> : diag <= 5 * 256 ? wfa_filter_kernel<5, 2> : wfa_filter_kernel<6, 2>;
  int occ = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, 64, 0) != hipSuccess || occ < 1) { (void)hipGetLastError(); occ = 8; }
  if (c->knobs.filter_per_cu > 0) occ = c->knobs.filter_per_cu;
  const int64_t grid = std::max<int64_t>(1, std::min<int64_t>((int64_t)c->num_cus * occ, L.n_jobs_host));
  if (c->knobs.debug) fprintf(stderr, "[filter] diagonals %lld occupancy %d grid %lld\n", (long long)diag, occ, (long long)grid);
  KTimer t(c, L.timer_slot);
  hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(64), 0, c->stream, a);
  TRGT_HIP_TRY(c, hipGetLastError());
  t.stop(0);
  c->last_filter_cells_dev = d_cells;
  return TRGT_OK;
}

}  // namespace trgt

using namespace trgt;

extern "C" int trgt_flank_filter_batch(trgt_hip_ctx* c, const trgt_span_params* p, int64_t n_jobs, const uint8_t* seqs,
                                       const uint64_t* pat_off, const uint32_t* pat_len, const uint64_t* txt_off,
                                       const uint32_t* txt_len, int32_t min_matches, int32_t* score, int32_t* match_bound,
                                       uint8_t* keep, int64_t* offsets_computed) {
  if (!c) return TRGT_ERR_INVALID;
  if (!p || n_jobs < 0 || (n_jobs > 0 && (!seqs || !pat_off || !pat_len || !txt_off || !txt_len)))
    return fail(c, TRGT_ERR_INVALID, "trgt_flank_filter_batch: null argument");
  if (p->mism != 2 || p->gapo != 5 || p->gape != 1)
    return fail(c, TRGT_ERR_UNSUPPORTED, "trgt_flank_filter_batch: only --aln-scoring 2,5,1 has a filter kernel");
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
  L.min_matches = min_matches; L.score = o_score.dev; L.bound = o_bound.dev; L.keep = o_keep.dev;
  if ((rc = flank_filter_launch(c, L))) return rc;
  if ((rc = o_score.finish(c)) || (rc = o_bound.finish(c)) || (rc = o_keep.finish(c))) return rc;
  unsigned long long cells = 0;
  TRGT_HIP_TRY(c, hipMemcpyAsync(&cells, c->last_filter_cells_dev, 8, hipMemcpyDeviceToHost, c->stream));
  TRGT_HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (offsets_computed) *offsets_computed = (int64_t)cells;
  if (c->timing) c->k_cells[TRGT_K_WFA_FILTER] += (int64_t)cells;
  return TRGT_OK;
}
