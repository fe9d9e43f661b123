// trgt_amd/csrc/wfa_fast.hpp -- LDS-resident fast path of the wavefront aligner for the case that
// dominates TRGT's batches: exact (Heuristic::None) unidirectional gap-affine alignment, i.e.
// THREAD_WFA_FLANK (src/commands/genotype.rs:66-80): a 250-bp flank piece against a ~1.2-kb read
// with both text ends free, ~1450 live diagonals per score level.
//
// What changes relative to the generic engine (results are identical, see tests/test_wfa_gpu.py):
//   * the live wavefronts -- M of the last max(x, o+e)+1 levels, I and D of the last e+1 levels --
//     sit in an LDS ring as 16-bit "offset+1" values (0 = NULL), indexed directly by diagonal, so the
//     recurrences read LDS, never HBM; pattern and text bytes are read from LDS too;
//   * the wavefront history needed by the back-trace is streamed to the HBM arena write-only
//     (coalesced 4-byte stores, one per offset: the 4*W term of the roofline model);
//   * ONE workgroup barrier per score level: every thread derives the trimmed descriptor of the level
//     it just helped to compute from a triple-buffered reduction block, so there is no serial
//     "thread 0 publishes, everyone waits" section in the loop.
#pragma once
#include "wfa_engine.hpp"

namespace trgt {
namespace wfa {

// Explicit global address space for the write-only history stream: a generic (flat) store would bump lgkmcnt as well,
// and every LDS wait in front of the per-level barrier would then also wait for the HBM stores to retire.
typedef __attribute__((address_space(1))) int32_t g_i32;

__device__ __forceinline__ uint16_t enc16(int32_t off) { return off < 0 ? (uint16_t)0 : (uint16_t)(off + 1); }
__device__ __forceinline__ int32_t dec16(uint16_t e) { return e ? (int32_t)e - 1 : OFF_NULL; }
__device__ __forceinline__ int32_t ring_get(const uint16_t* arr, int wcap, int koff, const WfDesc& d, int slot, int k) {
  return (k >= d.lo && k <= d.hi) ? dec16(arr[slot * wcap + k + koff]) : OFF_NULL;
}

__device__ __forceinline__ void fred_reset(FastRed& r) {
  r.lo[0] = r.lo[1] = r.lo[2] = INT32_MAX; r.hi[0] = r.hi[1] = r.hi[2] = INT32_MIN;
  r.term_key = ~0ull; r.end_val = OFF_NULL;
}

// P4, T4: LDS 4-byte sliding windows of pattern / text.  ring: (RM + 2*RI) * wcap uint16 in LDS.  A: history arena (HBM).
// All threads; returns ST_*.
__device__ __forceinline__ int wf_run_lds_affine(const KParams& kp, const uint32_t* P4, const uint32_t* T4, uint16_t* ring, int wcap,
                                                 int32_t* __restrict__ A) {
  Inst& I = sh.inst[I_UNI];
  const int tid = threadIdx.x, nT = blockDim.x;
  const int plen = I.plen, tlen = I.tlen, ak = tlen - plen, koff = plen + 2;  // one pad cell each side: k-1 / k+1 reads never leave the slot
  const int x = kp.pen.x, oe = kp.pen.o1 + kp.pen.e1, e = kp.pen.e1, scope = kp.pen.scope;
  const int RM = max(x, oe) + 1, RI = e + 1;
  uint16_t* Mr = ring;
  uint16_t* Ir = ring + RM * wcap;
  uint16_t* Dr = Ir + RI * wcap;
  const uint32_t cap = I.arena_cap;
  const int n_slots = I.n_slots, span = I.span, pef = I.pef, tef = I.tef, pbf = I.pbf, tbf = I.tbf;
  WfDesc* const gdesc = I.gdesc;
  WfDesc* const lring = sh.ring[I_UNI];

  // Extension over 4-byte sliding windows: P4[i] = pattern bytes i..i+3 (zero padded), T4 likewise.  One aligned LDS
  // dword per sequence covers four bases; most diagonals stop inside the first window, so the common case is straight-line.
  auto extend = [&](int k, int32_t off, FastRed& red) -> int32_t {
    int v = off - k, h = off;
    int n;
    do {
      const uint32_t xw = P4[v] ^ T4[h];
      n = xw ? (__builtin_ctz(xw) >> 3) : 4;
      n = min(n, min(plen - v, tlen - h));
      v += n; h += n;
    } while (n == 4);
    if (span == 1) {
      if ((h >= tlen && plen - v <= pef) || (v >= plen && tlen - h <= tef))
        atomicMin(&red.term_key, ((unsigned long long)(unsigned)(k + KBIAS) << 32) | (unsigned)h);
    } else if (k == ak) red.end_val = h;
    return h;
  };

  __syncthreads();
  if (tid == 0) {
    for (int r = 0; r < 3; ++r) fred_reset(sh.fred[r]);
    I.status = ST_OK; I.end_score = -1; I.num_null_steps = 0;
  }
  __syncthreads();
  // ---- score 0 (wavefront_unialign_init + first extension)
  WfDesc lastM, lastI = null_desc(), lastD = null_desc();
  lastM.lo = lastM.lo_alloc = span ? -pbf : 0;
  lastM.hi = span ? tbf : 0;
  lastM.base = 0;
  uint32_t bump = (uint32_t)(lastM.hi - lastM.lo + 1);
  unsigned long long cells = bump;
  int status = ST_OK, num_null = 0, s = 0;
  int slM = 0, slI = 0;  // s % RM, s % RI kept incrementally
  bool computed = true;
  if (bump > cap || n_slots < 1) status = ST_OOM;
  else {
    FastRed& red = sh.fred[0];
    for (int kb = lastM.lo; kb <= lastM.hi; kb += nT) {
      const int k = kb + tid;
      if (k <= lastM.hi) {
        int32_t off = span ? (k > 0 ? k : 0) : 0;
        off = extend(k, off, red);
        Mr[k + koff] = enc16(off);
        ((g_i32*)A)[(uint32_t)(k - lastM.lo)] = off;
      }
    }
  }
  while (status == ST_OK) {
    __syncthreads();  // the one barrier per level: level s is complete in LDS, its reductions are final
    // ---- everything the next level needs from LDS is fetched up front (one wait): this level's reductions and the
    //      descriptors of the older levels that feed level s+1
    const FastRed R = sh.fred[s % 3];
    const int sn = s + 1;
    const WfDesc r_mm = lring[((sn - x) & (RING - 1)) * 5 + CM], r_mo = lring[((sn - oe) & (RING - 1)) * 5 + CM];
    const WfDesc r_ie = lring[((sn - e) & (RING - 1)) * 5 + CI1], r_de = lring[((sn - e) & (RING - 1)) * 5 + CD1];
    if (s > 0 && computed) {  // wavefront_compute_trim_ends, derived redundantly by every thread
      if (R.lo[0] == INT32_MAX) lastM.hi = lastM.lo - 1; else { lastM.lo = R.lo[0]; lastM.hi = R.hi[0]; }
      if (lastI.base != NOBASE) { if (R.lo[1] == INT32_MAX) lastI.hi = lastI.lo - 1; else { lastI.lo = R.lo[1]; lastI.hi = R.hi[1]; } }
      if (lastD.base != NOBASE) { if (R.lo[2] == INT32_MAX) lastD.hi = lastD.lo - 1; else { lastD.lo = R.lo[2]; lastD.hi = R.hi[2]; } }
    }
    bool end_reached = false;
    int end_k = 0, end_off = 0;
    if (lastM.base != NOBASE) {
      if (span == 1) {
        if (R.term_key != ~0ull) { end_reached = true; end_k = (int)(R.term_key >> 32) - KBIAS; end_off = (int)(R.term_key & 0xFFFFFFFFu); }
      } else if (ak >= lastM.lo && ak <= lastM.hi && R.end_val >= tlen) { end_reached = true; end_k = ak; end_off = tlen; }
    }
    if (tid == 0) {  // publish level s: LDS mirror for the recurrences of later levels, HBM copy for the back-trace
      WfDesc* lr = lring + (s & (RING - 1)) * 5;
      lr[CM] = lastM; lr[CI1] = lastI; lr[CD1] = lastD;
      if (s < n_slots) { WfDesc* g = gdesc + (size_t)s * 5; g[CM] = lastM; g[CI1] = lastI; g[CD1] = lastD; }
      fred_reset(sh.fred[(s + 2) % 3]);
      if (end_reached) { I.end_score = s; I.end_k = end_k; I.end_off = end_off; }
    }
    if (end_reached) { status = ST_END_REACHED; break; }
    if (lastM.base == NOBASE && num_null > scope) { status = ST_END_UNREACHABLE; break; }
    // ---- next level
    ++s;
    slM = slM + 1 == RM ? 0 : slM + 1; slI = slI + 1 == RI ? 0 : slI + 1;
    auto pick = [&](const WfDesc& ringd, const WfDesc& last, int lvl) -> WfDesc {  // branch-free select + null normalisation
      WfDesc d = lvl == s - 1 ? last : ringd;
      const bool bad = lvl < 0 || d.base == NOBASE || d.lo > d.hi;
      d.lo = bad ? 1 : d.lo; d.hi = bad ? -1 : d.hi; d.lo_alloc = bad ? 1 : d.lo_alloc; d.base = bad ? NOBASE : d.base;
      return d;
    };
    const WfDesc m_mis = pick(r_mm, lastM, s - x), m_o = pick(r_mo, lastM, s - oe), ie = pick(r_ie, lastI, s - e), de = pick(r_de, lastD, s - e);
    if (m_mis.base == NOBASE && m_o.base == NOBASE && ie.base == NOBASE && de.base == NOBASE) {
      ++num_null; computed = false;
      lastM = null_desc(); lastI = null_desc(); lastD = null_desc();
      continue;
    }
    num_null = 0; computed = true;
    int lo = m_mis.lo, hi = m_mis.hi;
    lim(m_o, -1, +1, lo, hi); lim(ie, +1, +1, lo, hi); lim(de, -1, -1, lo, hi);
    const uint32_t w = (uint32_t)max(0, hi - lo + 1);
    if (s >= n_slots || (unsigned long long)bump + 3ull * w > cap) { status = ST_OOM; break; }
    const uint32_t bM = bump, bI = bump + w, bD = bump + 2 * w;
    bump += 3 * w; cells += 3ull * w;
    const bool has_i = m_o.base != NOBASE || ie.base != NOBASE, has_d = m_o.base != NOBASE || de.base != NOBASE;
    lastM.lo = lastM.lo_alloc = lo; lastM.hi = hi; lastM.base = bM;
    lastI = lastM; lastI.base = has_i ? bI : NOBASE;
    lastD = lastM; lastD.base = has_d ? bD : NOBASE;
    if (!has_i) { lastI.lo = lastI.lo_alloc = 1; lastI.hi = -1; }
    if (!has_d) { lastD.lo = lastD.lo_alloc = 1; lastD.hi = -1; }
    // Branch-light strip loop in the encoded domain (enc = offset + 1, 0 = NULL):
    //   ins = max(Mo[k-1], Ie[k-1]) (+1 if non-NULL), del = max(Mo[k+1], De[k+1]), mis = Mm[k] (+1 if non-NULL)
    const int slMo = slM - oe < 0 ? slM - oe + RM : slM - oe, slMm = slM - x < 0 ? slM - x + RM : slM - x;  // oe, x < RM
    const int slIe = slI - e < 0 ? slI - e + RI : slI - e;
    const uint16_t* pMo = Mr + slMo * wcap + koff;
    const uint16_t* pMm = Mr + slMm * wcap + koff;
    const uint16_t* pIe = Ir + slIe * wcap + koff;
    const uint16_t* pDe = Dr + slIe * wcap + koff;
    uint16_t* qM = Mr + slM * wcap + koff;
    uint16_t* qI = Ir + slI * wcap + koff;
    uint16_t* qD = Dr + slI * wcap + koff;
    const int lo_mo = m_o.lo, lo_mm = m_mis.lo, lo_ie = ie.lo, lo_de = de.lo;
    const unsigned n_mo = m_o.base == NOBASE ? 0u : (unsigned)(m_o.hi - m_o.lo + 1), n_mm = m_mis.base == NOBASE ? 0u : (unsigned)(m_mis.hi - m_mis.lo + 1);
    const unsigned n_ie = ie.base == NOBASE ? 0u : (unsigned)(ie.hi - ie.lo + 1), n_de = de.base == NOBASE ? 0u : (unsigned)(de.hi - de.lo + 1);
    g_i32* __restrict__ hM = (g_i32*)(A + bM - lo);
    g_i32* __restrict__ hI = (g_i32*)(A + bI - lo);
    g_i32* __restrict__ hD = (g_i32*)(A + bD - lo);
    FastRed& rn = sh.fred[s % 3];
    // per-lane first / last in-bounds diagonal of M, I, D as biased 16-bit values (kb = k + koff); "last" is stored
    // complemented so that a single packed unsigned min reduces everything: t0 = (fM, fI), t1 = (fD, ~lM), t2 = (~lI, ~lD)
    unsigned fM = 0xFFFFu, fI = 0xFFFFu, fD = 0xFFFFu, nlM = 0xFFFFu, nlI = 0xFFFFu, nlD = 0xFFFFu;
    for (int k = lo + tid; k <= hi; k += nT) {
      unsigned a = pMo[k - 1], b = pIe[k - 1], c = pMo[k + 1], d = pDe[k + 1], m = pMm[k];
      a = (unsigned)(k - 1 - lo_mo) < n_mo ? a : 0u;
      b = (unsigned)(k - 1 - lo_ie) < n_ie ? b : 0u;
      c = (unsigned)(k + 1 - lo_mo) < n_mo ? c : 0u;
      d = (unsigned)(k + 1 - lo_de) < n_de ? d : 0u;
      m = (unsigned)(k - lo_mm) < n_mm ? m : 0u;
      const unsigned mi = max(a, b);
      const unsigned ins = mi + (mi != 0u), del = max(c, d), mis = m + (m != 0u);
      unsigned mx = max(del, max(mis, ins));
      int32_t off = (int32_t)mx - 1;
      const bool okM = (uint32_t)off <= (uint32_t)tlen && (uint32_t)(off - k) <= (uint32_t)plen;
      if (okM) { off = extend(k, off, rn); mx = (unsigned)off + 1u; } else { mx = 0u; off = OFF_NULL; }
      qI[k] = (uint16_t)ins; qD[k] = (uint16_t)del; qM[k] = (uint16_t)mx;
      const int32_t vi = ins ? (int32_t)ins - 1 : OFF_NULL, vd = del ? (int32_t)del - 1 : OFF_NULL;
      hI[k] = vi; hD[k] = vd; hM[k] = off;  // history for the back-trace: written once, never re-read by this loop
      // wavefront_compute_trim_ends bookkeeping: k only grows per lane, so "first" is set once and "last" overwritten
      const bool okI = in_bounds(vi, k, plen, tlen), okD = in_bounds(vd, k, plen, tlen);
      const unsigned kb = (unsigned)(k + koff), nkb = 0xFFFFu - kb;
      fM = okM ? min(fM, kb) : fM; nlM = okM ? nkb : nlM;
      fI = okI ? min(fI, kb) : fI; nlI = okI ? nkb : nlI;
      fD = okD ? min(fD, kb) : fD; nlD = okD ? nkb : nlD;
    }
    {
      typedef unsigned short us2 __attribute__((ext_vector_type(2)));
      union U { unsigned u; us2 v; };
      U t0, t1, t2;
      t0.u = fM | (fI << 16); t1.u = fD | (nlM << 16); t2.u = nlI | (nlD << 16);
      for (int o = 32; o > 0; o >>= 1) {
        U a0, a1, a2;
        a0.u = __shfl_xor(t0.u, o); a1.u = __shfl_xor(t1.u, o); a2.u = __shfl_xor(t2.u, o);
        t0.v = __builtin_elementwise_min(t0.v, a0.v); t1.v = __builtin_elementwise_min(t1.v, a1.v); t2.v = __builtin_elementwise_min(t2.v, a2.v);
      }
      if ((tid & 63) == 0) {
        const unsigned rfM = t0.u & 0xFFFFu, rfI = t0.u >> 16, rfD = t1.u & 0xFFFFu, rlM = 0xFFFFu - (t1.u >> 16), rlI = 0xFFFFu - (t2.u & 0xFFFFu), rlD = 0xFFFFu - (t2.u >> 16);
        if (rfM != 0xFFFFu) { atomicMin(&rn.lo[0], (int)rfM - koff); atomicMax(&rn.hi[0], (int)rlM - koff); }
        if (rfI != 0xFFFFu) { atomicMin(&rn.lo[1], (int)rfI - koff); atomicMax(&rn.hi[1], (int)rlI - koff); }
        if (rfD != 0xFFFFu) { atomicMin(&rn.lo[2], (int)rfD - koff); atomicMax(&rn.hi[2], (int)rlD - koff); }
      }
    }
  }
  __syncthreads();
  if (tid == 0) { I.status = status; sh.cells += cells; }
  __syncthreads();
  return status;
}

// ------------------------------------------------------------------------------------------------
// Back-trace of the fast path (gap-affine, single piece).  Same candidate encoding and priorities as
// wf_backtrace (wavefront_backtrace_affine: (offset << 4 | type), maximum wins), but executed by the 64 lanes of
// wave 0 with the level descriptors in LDS, and with gap runs resolved 64 steps per memory round trip: along a pure
// gap-extension chain the next cells are known in advance -- (s - j*e, k + j) for deletions, (s - j*e, k - j) with the
// offset falling by one for insertions -- so lane j fetches step j's two candidates and a ballot finds where the
// chain stops.  Long deletion runs are what the back-trace of a read that lacks the flank consists of.
// ld: LDS copy of the descriptors, ld[s * 3 + {0: M, 1: I1, 2: D1}].  Runs are pushed (reversed) by lane 0.
__device__ __forceinline__ long long bt_cand_lds(const WfDesc* ld, const int32_t* __restrict__ A, int cidx, int s, int k, int add, int type) {
  if (s < 0) return (long long)OFF_NULL;
  const WfDesc d = ld[s * 3 + cidx];
  if (d.base == NOBASE || k < d.lo || k > d.hi) return (long long)OFF_NULL;
  return (((long long)(A[d.base + (uint32_t)(k - d.lo_alloc)] + add)) << 4) | type;
}

__device__ __forceinline__ void wf_backtrace_fast_affine(const KParams& kp, const WfDesc* ld, const int32_t* __restrict__ A,
                                                         uint32_t* tmp, int& ntmp_out, uint32_t cap) {
  const Inst& I = sh.inst[I_UNI];
  const int lane = threadIdx.x & 63;
  const int plen = I.plen, tlen = I.tlen;
  const int x = kp.pen.x, oe = kp.pen.o1 + kp.pen.e1, e = kp.pen.e1;
  int mt = CM, s = I.end_score, k = I.end_k, off = I.end_off;
  int h = off, v = off - k, nt = 0;
  if (lane == 0) { rle_push(tmp, nt, cap, 2u, plen - v); rle_push(tmp, nt, cap, 1u, tlen - h); }
  while (v > 0 && h > 0 && s > 0) {
    if (mt == CM) {
      long long best = bt_cand_lds(ld, A, 0, s - x, k, +1, 9);
      best = max(best, bt_cand_lds(ld, A, 2, s - e, k + 1, 0, 6));
      best = max(best, bt_cand_lds(ld, A, 0, s - oe, k + 1, 0, 5));
      best = max(best, bt_cand_lds(ld, A, 1, s - e, k - 1, +1, 2));
      best = max(best, bt_cand_lds(ld, A, 0, s - oe, k - 1, +1, 1));
      if (best < 0) break;
      const int best_off = (int)(best >> 4), type = (int)(best & 0xF);
      if (lane == 0) rle_push(tmp, nt, cap, 7u, off - best_off);
      off = best_off; h = off; v = off - k;
      if (v <= 0 || h <= 0) break;
      switch (type) {
        case 9: if (lane == 0) rle_push(tmp, nt, cap, 8u, 1); s -= x; --off; break;
        case 1: if (lane == 0) rle_push(tmp, nt, cap, 1u, 1); s -= oe; --k; --off; break;
        case 2: if (lane == 0) rle_push(tmp, nt, cap, 1u, 1); s -= e; mt = CI1; --k; --off; break;
        case 5: if (lane == 0) rle_push(tmp, nt, cap, 2u, 1); s -= oe; ++k; break;
        default: if (lane == 0) rle_push(tmp, nt, cap, 2u, 1); s -= e; mt = CD1; ++k; break;
      }
      h = off; v = off - k;
    } else {
      const bool del = mt == CD1;
      // ---- lane j inspects step j of the gap-extension chain
      const int sj = s - lane * e, kj = del ? k + lane : k - lane;
      const bool alive = (del ? (v - lane > 0 && h > 0) : (h - lane > 0 && v > 0)) && sj > 0;
      long long ce = (long long)OFF_NULL, co = (long long)OFF_NULL;
      if (alive) {
        ce = del ? bt_cand_lds(ld, A, 2, sj - e, kj + 1, 0, 6) : bt_cand_lds(ld, A, 1, sj - e, kj - 1, +1, 2);
        co = del ? bt_cand_lds(ld, A, 0, sj - oe, kj + 1, 0, 5) : bt_cand_lds(ld, A, 0, sj - oe, kj - 1, +1, 1);
      }
      const bool cont = alive && ce >= 0 && ce > co;  // the extension candidate is the maximum
      const unsigned long long stop = __ballot(!cont);
      const int j = stop ? __ffsll((long long)stop) - 1 : 64;
      if (j > 0) {  // j pure extension steps
        if (lane == 0) rle_push(tmp, nt, cap, del ? 2u : 1u, j);
        s -= j * e;
        if (del) { k += j; } else { k -= j; off -= j; }
        h = off; v = off - k;
        continue;
      }
      // ---- the chain stops right here: one ordinary step (gap open, or no source at all)
      const long long cext = del ? bt_cand_lds(ld, A, 2, s - e, k + 1, 0, 6) : bt_cand_lds(ld, A, 1, s - e, k - 1, +1, 2);
      const long long copn = del ? bt_cand_lds(ld, A, 0, s - oe, k + 1, 0, 5) : bt_cand_lds(ld, A, 0, s - oe, k - 1, +1, 1);
      const long long best = max(cext, copn);
      if (best < 0) break;
      if (lane == 0) rle_push(tmp, nt, cap, del ? 2u : 1u, 1);
      if (best == cext) s -= e; else { s -= oe; mt = CM; }
      if (del) ++k; else { --k; --off; }
      h = off; v = off - k;
    }
  }
  if (lane == 0) {
    if (mt == CM && v > 0 && h > 0) { const int n = min(v, h); rle_push(tmp, nt, cap, 7u, n); v -= n; h -= n; }
    rle_push(tmp, nt, cap, 2u, v);
    rle_push(tmp, nt, cap, 1u, h);
    ntmp_out = nt;
  }
}

}  // namespace wfa
}  // namespace trgt
