// trgt_amd/csrc/wfa_fast.hpp -- dedicated LDS-resident kernel for the alignments that dominate TRGT's batches:
// exact (Heuristic::None) unidirectional gap-affine WFA, i.e. THREAD_WFA_FLANK (src/commands/genotype.rs:66-80):
// a 250-bp flank piece against a ~1-kb read with both text ends free, ~1300 live diagonals per score level.
// Included by wfa.hip after the generic engine; the host planner there launches it whenever a whole batch qualifies.
//
// What differs from the generic engine (results are identical, see tests/test_wfa_gpu.py):
//   * the live wavefronts -- M of the last max(x, o+e)+1 levels, I and D of the last e+1 levels --
//     sit in an LDS ring as 16-bit "offset+1" values (0 = NULL), indexed directly by the biased diagonal
//     kb = k + plen + 2, so the recurrences read LDS, never HBM; pattern and text are read from LDS too;
//   * the wavefront history needed by the back-trace is streamed to the HBM arena write-only, in the same
//     16-bit encoding (coalesced 2-byte stores, one per offset: the 2*W term of the roofline model);
//   * ONE workgroup barrier per score level and no atomics on the per-level path: every wave finds the first / last
//     in-bounds diagonal of its strips with ballots on the scalar unit and leaves one 16-byte record in LDS; after
//     the barrier every wave folds the records (packed 16-bit min) and derives the next level's descriptors itself,
//     on the scalar unit, from (lo | hi << 16) words -- there is no "thread 0 publishes, everyone waits" section;
//   * strips that lie inside all four source ranges (the common case) run a body without range masks;
//   * it is a kernel of its own: nothing of the BiWFA / heuristic machinery is live, so the per-level state stays in
//     SGPRs and the strip loop in registers (inside the generic kernel the same code spilled, and every spill reload in
//     the strip loop is an s_waitcnt vmcnt(0) that also waits for the history stores in flight).
#pragma once
#include <type_traits>
#include "wfa_engine.hpp"

namespace trgt {
namespace wfa {

// Explicit global address space for the write-only history stream: a generic (flat) store would bump lgkmcnt as well,
// and every LDS wait in front of the per-level barrier would then also wait for the HBM stores to retire.
typedef __attribute__((address_space(1))) uint16_t g_u16;

#ifdef TRGT_WFA_PROF
// Developer-only build (make PROF=1): thread 0 of every workgroup splits its shader-clock time over the phases of a job
// (g_wfa_prof) and over the phases of a score level (g_wfa_lvprof).
__device__ unsigned long long g_wfa_prof[32];
__device__ unsigned long long g_wfa_lvprof[32];  // [wave][8]: lane 0 of every wave
#define PROF_DECL unsigned long long pf_t = clock64(), pf_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PROF_MARK(i) do { const unsigned long long n_ = clock64(); pf_acc[i] += n_ - pf_t; pf_t = n_; } while (0)
#define LV_DECL unsigned long long lv_t = clock64(), lv_acc[5] = {0, 0, 0, 0, 0}
#define LV_MARK(i) do { const unsigned long long n_ = clock64(); lv_acc[i] += n_ - lv_t; lv_t = n_; } while (0)
#define LV_FLUSH do { if ((threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 5; ++i_) atomicAdd(&g_wfa_lvprof[(threadIdx.x >> 6) * 8 + i_], lv_acc[i_]); } while (0)
#else
#define PROF_DECL
#define PROF_MARK(i)
#define LV_DECL
#define LV_MARK(i)
#define LV_FLUSH
#endif

// Packed descriptor of one wavefront component: biased lo | biased hi << 16; lo > hi = null.
__device__ __forceinline__ int pd_lo(uint32_t d) { return (int)(d & 0xFFFFu); }
__device__ __forceinline__ int pd_hi(uint32_t d) { return (int)(d >> 16); }
__device__ __forceinline__ bool pd_null(uint32_t d) { return pd_lo(d) > pd_hi(d); }
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  union U { uint32_t u; us2 v; } x, y, r;
  x.u = a; y.u = b; r.v = __builtin_elementwise_min(x.v, y.v);
  return r.u;
}

__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  union U { uint32_t u; us2 v; } x, y, r;
  x.u = a; y.u = b; r.v = __builtin_elementwise_max(x.v, y.v);
  return r.u;
}
__device__ __forceinline__ uint32_t pk_add_u16(uint32_t a, uint32_t b) {
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  union U { uint32_t u; us2 v; } x, y, r;
  x.u = a; y.u = b; r.v = x.v + y.v;
  return r.u;
}
// (x + (x != 0)) on both halves: the "+1 unless NULL" of the encoded domain
__device__ __forceinline__ uint32_t pk_inc_nz(uint32_t x) { return pk_add_u16(x, pk_min_u16(x, 0x00010001u)); }

// v_ffbl_b32 returns -1 for a zero input, which is exactly what the extension wants ("no mismatch in this window" -> a byte
// index >= 4 after the shift); __builtin_ctz would be undefined there and __ffs costs a compare + select.
__device__ __forceinline__ uint32_t ffbl_raw(uint32_t v) {
  uint32_t r;
  asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(v));
  return r;
}
// History stream: raw buffer stores (resource + per-lane byte offset + scalar byte offset) need no per-store address
// arithmetic.  The resource base sits HIST_BIAS elements below the arena so that the scalar offset of a level whose lowest
// biased diagonal exceeds its bump offset stays non-negative.
constexpr int HIST_BIAS = 1 << 16;
__device__ __forceinline__ void hist_store(__amdgpu_buffer_rsrc_t rs, uint32_t kb2, uint32_t soff, uint32_t enc) {
  __builtin_amdgcn_raw_buffer_store_b16((short)enc, rs, (int)kb2, (int)soff, 0);
}

// History descriptor of one level in HBM: FD_STRIDE dwords {M, I, D packed ranges (trimmed), base, lo_alloc | width << 16}.
// Offsets of component c (0 M, 1 I, 2 D) of diagonal kb sit at A16[base + c * width + (kb - lo_alloc)].
constexpr int FD_STRIDE = 8, FD_LDS_STRIDE = 5;

struct FastTerm {  // termination record of one score level (triple-buffered)
  unsigned long long term_key;  // (k + KBIAS) << 32 | offset, minimum = first terminating diagonal
  int end_val, pad;
};

struct FastShared {
  uint4 fdesc[RING];    // packed descriptors {M, I, D, history base} of the last RING levels
  uint32_t flaw[RING];  // ... and their lo_alloc | width << 16: a back-trace of fewer than RING levels needs nothing from HBM but offsets
  uint4 wred[2][16];    // per-wave trim records, double-buffered
  FastTerm fterm[3];
  int job, slot, rle_n, total_ops, chunk_end;
};
__shared__ FastShared g_fsh;

struct FastJob { int plen, tlen, span, pbf, pef, tbf, tef, n_slots, koff; uint32_t cap; };  // all wave-uniform
struct FastEnd { int status, score, k, off; unsigned long long cells; };

// P4, T4: LDS 4-byte sliding windows of pattern / text.  ring: (RM + 2*RI) * wcap uint16 in LDS.  A16: history arena,
// gd: history descriptors (both HBM, uniform pointers).  All threads (blockDim.x a multiple of 64).
// SPEC: the configuration TRGT's flank location always runs -- gap-affine (x, o, e) = (2, 5, 1), ends-free with the whole text
// free at both ends and the pattern not at all (span_locater.rs:17, genotype.rs:66-80) -- as compile-time constants: the ring
// geometry, the source-level selects and the termination test fold away, and with them a third of the scalar registers
// (the general instantiation spills SGPRs to VGPR lanes in the per-level prologue).
template <int SPEC, bool WIN = false>  // SPEC 0: general; else the compile-time thread count of TRGT's flank configuration (256, 192, 64).
                                       // WIN: the windowed launch -- the same constants but text_begin_free and the ring bias from the job
__device__ __forceinline__ FastEnd wf_run_lds_affine(const Pen& pen, const FastJob& J, const uint32_t* P4, const uint32_t* T4, uint16_t* ring,
                                                     int wcap, g_u16* A16, uint32_t* gd) {
  FastShared& fs = g_fsh;
  const int tid = threadIdx.x, nT = SPEC ? SPEC : (int)blockDim.x, lane = tid & 63;  // SPEC is launched with SPEC threads only
  const int wave = rfl(tid >> 6), nW = nT >> 6;
  const int plen = J.plen, tlen = J.tlen, koff = SPEC && !WIN ? plen + 2 : J.koff;  // plen + 2 (one pad cell each side: kb-1 / kb+1 reads never leave
                                                                              // the slot) unless the launch bounds the penalty (KArgs::fast_koff)
  const int ak_b = tlen - plen + koff;
  const int x = SPEC ? 2 : pen.x, oe = SPEC ? 6 : pen.o1 + pen.e1, e = SPEC ? 1 : pen.e1, scope = SPEC ? 7 : pen.scope;
  const int RM = max(x, oe) + 1, RI = e + 1;
  uint16_t* const Mr = ring;
  uint16_t* const Ir = ring + RM * wcap;
  uint16_t* const Dr = Ir + RI * wcap;
  const uint32_t cap = J.cap;
  const int n_slots = J.n_slots, span = SPEC ? 1 : J.span, pef = SPEC ? 0 : J.pef, tef = SPEC ? tlen : J.tef, pbf = SPEC ? 0 : J.pbf,
            tbf = SPEC && !WIN ? tlen : J.tbf;
  const uint32_t PDN = (uint32_t)(koff + 1) | ((uint32_t)(koff - 1) << 16);  // the canonical null wavefront (lo = 1, hi = -1)
  const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc((void*)(A16 - HIST_BIAS), 0, -1, 0x00020000);
  // which sources are "the level just finished" (taken from registers) rather than an older one (taken from the LDS ring)
  const bool mis_cur = x == 1, o_cur = oe == 1, e_cur = e == 1;
  const int W2 = wcap * 2, RMW = RM * W2, RIW = RI * W2, XW = x * W2, OEW = oe * W2, EW = e * W2;  // ring geometry in bytes

  // Extension over 4-byte sliding windows: P4[i] = pattern bytes i..i+3 (zero padded), T4 likewise.  One aligned LDS
  // dword per sequence covers four bases; most diagonals stop inside the first window, so the common case is straight-line.
  // The rest of a run of matches, by the whole wave (uniform control flow; `c`: this lane's cell matched its windows so far and goes on
  // at (v, h)).  A cell that is still going after two windows is almost surely on the diagonal of the alignment itself -- the piece
  // matches the read there for dozens or hundreds of bases -- and one lane stepping four bases at a time kept the other 63 waiting
  // (a quarter of the windowed launch's instructions).  Lane i compares the window 4 i bases further on: 256 bases per round trip; the
  // first lane whose window does not match whole ends the run, exactly where the one-lane loop would have ended it.
  auto finish_runs = [&](bool c, int& v, int& h) {
    unsigned long long m = __ballot(c);
    while (m) {
      const int j = (int)__builtin_ctzll(m);
      m &= m - 1ull;
      int vj = __builtin_amdgcn_readlane(v, j), hj = __builtin_amdgcn_readlane(h, j);
      for (;;) {
        const int pv = vj + 4 * lane, ph = hj + 4 * lane;
        const int rem = min(plen - pv, tlen - ph);  // (<= 0 beyond either sequence: that lane ends the run with nothing)
        const uint32_t xw = P4[min(pv, plen)] ^ T4[min(ph, tlen)];
        const uint32_t n = rem > 0 ? min(min(ffbl_raw(xw) >> 3, 4u), (uint32_t)rem) : 0u;
        const unsigned long long stop = __ballot(n < 4u);
        if (stop) {
          const int js = (int)__builtin_ctzll(stop);
          const int ext = 4 * js + __builtin_amdgcn_readlane((int)n, js);
          vj += ext; hj += ext;
          break;
        }
        vj += 256; hj += 256;
      }
      if (lane == j) { v = vj; h = hj; }
    }
  };
  // extend() for a whole wave in uniform control flow (`on`: this lane has a cell): two windows per lane, then finish_runs
  auto extend_u = [&](int k, int32_t off, bool on, FastTerm& tm) -> int32_t {
    int v = on ? off - k : 0, h = on ? off : 0;
    bool c = on;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      if (c) {
        const uint32_t xw = P4[v] ^ T4[h];
        const uint32_t n = min(min(ffbl_raw(xw) >> 3, 4u), (uint32_t)min(plen - v, tlen - h));
        v += (int)n; h += (int)n;
        c = n == 4u;
      }
    }
    finish_runs(c, v, h);
    if (on) {
      if (span == 1) {
        if ((h >= tlen && plen - v <= pef) || (v >= plen && tlen - h <= tef))
          atomicMin(&tm.term_key, ((unsigned long long)(unsigned)(k + KBIAS) << 32) | (unsigned)h);
      } else if (k + koff == ak_b) tm.end_val = h;
    }
    return h;
  };

  __syncthreads();
  if (tid < RING) fs.fdesc[tid] = make_uint4(PDN, PDN, PDN, 0u);  // levels "-1 .. -RING" are null: no lvl < 0 tests later
  if (tid == 0)
    for (int r = 0; r < 3; ++r) { fs.fterm[r].term_key = ~0ull; fs.fterm[r].end_val = OFF_NULL; }
  __syncthreads();
  // ---- score 0 (wavefront_unialign_init + first extension); level 0 is never trimmed
  uint32_t cM, cI = PDN, cD = PDN;      // packed (trimmed) descriptors of the current level s
  uint32_t lo_c, w_c, base_c = 0;       // computed range / history base of the current level
  {
    const int lo0 = span ? -pbf : 0, hi0 = span ? tbf : 0;
    cM = (uint32_t)(lo0 + koff) | ((uint32_t)(hi0 + koff) << 16);
    lo_c = (uint32_t)(lo0 + koff); w_c = (uint32_t)(hi0 - lo0 + 1);
  }
  uint32_t bump = w_c;
  unsigned long long cells = bump;
  int status = ST_OK, num_null = 0, s = 0;
  int oM = 0, oI = 0, s3 = 0;  // byte offsets of the slots of level s in the M / I,D rings; s % 3
  bool computed = true;
  if (bump > cap || n_slots < 1) status = ST_OOM;
  else {
    FastTerm& tm = fs.fterm[0];
    const int hi_b = pd_hi(cM);
    const uint32_t so0 = 2u * (uint32_t)(HIST_BIAS - (int)lo_c);
    for (int kb0 = (int)lo_c + wave * 64; kb0 <= hi_b; kb0 += nT) {
      const int kb = kb0 + lane;
      const bool on = kb <= hi_b;
      const int k = kb - koff;
      int32_t off = span ? (k > 0 ? k : 0) : 0;
      off = extend_u(k, off, on, tm);
      if (on) {
        Mr[kb] = (uint16_t)(off + 1);
        hist_store(hrs, 2u * (uint32_t)kb, so0, (uint32_t)(off + 1));
      }
    }
  }
  int end_k = 0, end_off = 0;
  LV_DECL;
  while (status == ST_OK) {
    LV_MARK(3);
    __syncthreads();  // the one barrier per level: level s is complete in LDS, its trim / termination records are final
    LV_MARK(0);
    // ---- everything the next level needs from LDS is fetched up front (one wait): this level's records and the
    //      descriptors of the older levels that feed level s+1
    const int sn = s + 1;
    const FastTerm Tm = fs.fterm[s3];
    const uint4 fa = fs.fdesc[(sn - x) & (RING - 1)], fb = fs.fdesc[(sn - oe) & (RING - 1)], fc = fs.fdesc[(sn - e) & (RING - 1)];
    if (s > 0 && computed) {  // wavefront_compute_trim_ends from the per-wave records
      uint4 r = fs.wred[s & 1][0];
      for (int w = 1; w < nW; ++w) {
        const uint4 q = fs.wred[s & 1][w];
        r.x = pk_min_u16(r.x, q.x); r.y = pk_min_u16(r.y, q.y); r.z = pk_min_u16(r.z, q.z);
      }
      const uint32_t rx = rfl(r.x), ry = rfl(r.y), rz = rfl(r.z);
      cM = (rx & 0xFFFFu) == 0xFFFFu ? PDN : rx ^ 0xFFFF0000u;
      cI = (ry & 0xFFFFu) == 0xFFFFu ? PDN : ry ^ 0xFFFF0000u;
      cD = (rz & 0xFFFFu) == 0xFFFFu ? PDN : rz ^ 0xFFFF0000u;
    }
    bool end_reached = false;
    if (computed) {
      if (span == 1) {
        const uint32_t tk_hi = rfl((uint32_t)(Tm.term_key >> 32)), tk_lo = rfl((uint32_t)Tm.term_key);
        if ((tk_hi & tk_lo) != 0xFFFFFFFFu) { end_reached = true; end_k = (int)tk_hi - KBIAS; end_off = (int)tk_lo; }
      } else if (ak_b >= pd_lo(cM) && ak_b <= pd_hi(cM) && rfl(Tm.end_val) >= tlen) { end_reached = true; end_k = ak_b - koff; end_off = tlen; }
    }
    if (tid == 0) {  // publish level s: LDS ring for the recurrences of later levels, HBM copy for the back-trace
      fs.fdesc[s & (RING - 1)] = make_uint4(cM, cI, cD, base_c);
      fs.flaw[s & (RING - 1)] = lo_c | (w_c << 16);
      if (s < n_slots) {
        uint32_t* g = gd + (size_t)s * FD_STRIDE;
        *reinterpret_cast<uint4*>(g) = make_uint4(cM, cI, cD, base_c);
        g[4] = lo_c | (w_c << 16);
      }
      FastTerm& nx = fs.fterm[s3 == 0 ? 2 : s3 - 1];  // (s + 2) % 3
      nx.term_key = ~0ull; nx.end_val = OFF_NULL;
    }  // (nothing in this block may depend on end_reached: a shared condition lets the compiler thread the uniform exit
       //  below through this divergent branch, and the whole per-level state then counts as divergent -> VGPRs)
    if (end_reached) { status = ST_END_REACHED; break; }
    if (!computed && num_null > scope) { status = ST_END_UNREACHABLE; break; }
    // ---- next level
    ++s;
    oM = oM + W2 == RMW ? 0 : oM + W2; oI = oI + W2 == RIW ? 0 : oI + W2; s3 = s3 == 2 ? 0 : s3 + 1;
    // wavefront_compute_get_*: the ring holds canonical descriptors (a NULL or ->null wavefront is PDN, and so is every
    // level below 0), so a source is a plain select between registers and the ring word fetched above
    const uint32_t m_mis = mis_cur ? cM : rfl(fa.x), m_o = o_cur ? cM : rfl(fb.x), ie = e_cur ? cI : rfl(fc.y), de = e_cur ? cD : rfl(fc.z);
    if ((((m_mis ^ PDN) | (m_o ^ PDN)) | ((ie ^ PDN) | (de ^ PDN))) == 0u) {
      ++num_null; computed = false;
      cM = cI = cD = PDN; lo_c = 0; w_c = 0; base_c = 0;
      continue;
    }
    num_null = 0; computed = true;
    // wavefront_compute_limits_input (null wavefronts take part with lo = 1, hi = -1, exactly as in the library)
    const int lo = min(min(pd_lo(m_mis), pd_lo(m_o) - 1), min(pd_lo(ie) + 1, pd_lo(de) - 1));
    const int hi = max(max(pd_hi(m_mis), pd_hi(m_o) + 1), max(pd_hi(ie) + 1, pd_hi(de) - 1));
    const uint32_t w = (uint32_t)max(0, hi - lo + 1);
    // history placement: component stride and (base - lo) are kept even, so that the pair of diagonals a lane owns in the packed
    // strips below is one aligned dword in LDS and in HBM alike
    const uint32_t wp = w + (w & 1u);
    if (s >= n_slots || 3u * wp + 1u > cap - bump) { status = ST_OOM; break; }  // bump <= cap always
    const uint32_t bM = bump + ((bump ^ (uint32_t)lo) & 1u);
    bump = bM + 3 * wp; cells += 3ull * w;
    lo_c = (uint32_t)lo; w_c = wp; base_c = bM;
    const int oMo = oM - OEW < 0 ? oM - OEW + RMW : oM - OEW, oMm = oM - XW < 0 ? oM - XW + RMW : oM - XW;  // oe, x < RM
    const int oIe = oI - EW < 0 ? oI - EW + RIW : oI - EW;
    const uint16_t* const pMo = reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(Mr) + oMo);
    const uint16_t* const pMm = reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(Mr) + oMm);
    const uint16_t* const pIe = reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(Ir) + oIe);
    const uint16_t* const pDe = reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(Dr) + oIe);
    uint16_t* const qM = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(Mr) + oM);
    uint16_t* const qI = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(Ir) + oI);
    uint16_t* const qD = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(Dr) + oI);
    const int lo_mo = pd_lo(m_o), hi_mo = pd_hi(m_o), lo_mm = pd_lo(m_mis), hi_mm = pd_hi(m_mis);
    const int lo_ie = pd_lo(ie), hi_ie = pd_hi(ie), lo_de = pd_lo(de), hi_de = pd_hi(de);
    const unsigned n_mo = (unsigned)max(0, hi_mo - lo_mo + 1), n_mm = (unsigned)max(0, hi_mm - lo_mm + 1);
    const unsigned n_ie = (unsigned)max(0, hi_ie - lo_ie + 1), n_de = (unsigned)max(0, hi_de - lo_de + 1);
    const uint32_t soM = 2u * (uint32_t)((int)bM - lo + HIST_BIAS), soI = soM + 2u * wp, soD = soI + 2u * wp;  // history: scalar byte offsets
    FastTerm& tn = fs.fterm[s3];
    // first / last in-bounds diagonal of M, I, D seen by this wave (biased), scalar registers updated from ballots
    int fM = 0xFFFF, fI = 0xFFFF, fD = 0xFFFF, lM = 0, lI = 0, lD = 0;
    auto note = [](unsigned long long m, int kbase, int& f, int& l) {  // ballot over 64 consecutive diagonals
      if (m) { f = min(f, kbase + (int)__builtin_ctzll(m)); l = max(l, kbase + 63 - (int)__builtin_clzll(m)); }
    };
    auto note2 = [](unsigned long long mA, unsigned long long mB, int kbase, int& f, int& l) {  // even / odd diagonals of a 128-wide strip
      if ((mA & mB) == ~0ull) { f = min(f, kbase); l = max(l, kbase + 127); return; }
      if (mA) { f = min(f, kbase + 2 * (int)__builtin_ctzll(mA)); l = max(l, kbase + 2 * (63 - (int)__builtin_clzll(mA))); }
      if (mB) { f = min(f, kbase + 1 + 2 * (int)__builtin_ctzll(mB)); l = max(l, kbase + 1 + 2 * (63 - (int)__builtin_clzll(mB))); }
    };
    // strips whose diagonals (and their k-1 / k+1 neighbours) lie inside all four source ranges need no masks
    const int in_lo = max(max(lo_mo + 1, lo_ie + 1), max(lo_de - 1, lo_mm));
    const int in_hi = min(min(hi_mo - 1, hi_ie + 1), min(hi_de - 1, hi_mm));
    LV_MARK(1);
    // One diagonal per lane, every source masked by its range: the strips at the two ends of a wavefront.
    // Encoded domain (enc = offset + 1, 0 = NULL): ins = max(Mo[k-1], Ie[k-1]) + 1, del = max(Mo[k+1], De[k+1]), mis = Mm[k] + 1.
    auto edge = [&](int kb, bool& okM, int32_t& off_e, unsigned& ins_e, unsigned& del_e) {
      unsigned a = pMo[kb - 1], b = pIe[kb - 1], c = pMo[kb + 1], d = pDe[kb + 1], m = pMm[kb];
      a = (unsigned)(kb - 1 - lo_mo) < n_mo ? a : 0u;
      b = (unsigned)(kb - 1 - lo_ie) < n_ie ? b : 0u;
      c = (unsigned)(kb + 1 - lo_mo) < n_mo ? c : 0u;
      d = (unsigned)(kb + 1 - lo_de) < n_de ? d : 0u;
      m = (unsigned)(kb - lo_mm) < n_mm ? m : 0u;
      const unsigned mi = max(a, b);
      const unsigned ins = mi + (mi != 0u), del = max(c, d), mis = m + (m != 0u);
      unsigned mx = max(del, max(mis, ins));
      const int k = kb - koff;
      off_e = (int32_t)mx - 1;
      okM = (uint32_t)off_e <= (uint32_t)tlen && (uint32_t)(off_e - k) <= (uint32_t)plen;
      ins_e = ins; del_e = del;
    };
    // ... and, after the extension (taken by the whole wave together: extend_u), what is stored of the cell
    auto edge_store = [&](int kb, bool okM, int32_t off, unsigned ins, unsigned del, bool& okI, bool& okD) {
      const int k = kb - koff;
      const unsigned mx = okM ? (unsigned)off + 1u : 0u;
      qI[kb] = (uint16_t)ins; qD[kb] = (uint16_t)del; qM[kb] = (uint16_t)mx;
      const uint32_t kb2 = 2u * (uint32_t)kb;
      hist_store(hrs, kb2, soI, ins); hist_store(hrs, kb2, soD, del); hist_store(hrs, kb2, soM, mx);  // history: written once
      okI = (ins - 1u) <= (uint32_t)tlen && (ins - 1u - (uint32_t)k) <= (uint32_t)plen;
      okD = (del - 1u) <= (uint32_t)tlen && (del - 1u - (uint32_t)k) <= (uint32_t)plen;
    };
    // wave w takes the 128-diagonal strips nW-1-w, 2nW-1-w, ...: the ragged last strip then falls on the last wave, not on wave 0,
    // which also carries thread 0's publishing work.  Strips start on even diagonals.
    // (Per wave, make PROF=1: strip work is 33 / 42 / 58 / 69 % of the level time on waves 0..3 -- wave 3 owns the first strip, an end
    //  strip on the two-pass path, and in a level of 4n+1 strips the last one too.  Contiguous blocks of strips per wave, dealing the
    //  strips by cost, or moving only that last strip all balance the waves (barrier wait 9-31 % instead of 5-36 %) and were all
    //  SLOWER, 7.33 -> 7.75 / 8.47 / 7.83 ms: the strip work itself grows by ~11 % -- adjacent strips processed at the same time
    //  write adjacent history segments, and any other loop shape also costs the compiler's schedule of this body.)
    for (int kb0 = (lo & ~1) + (nW - 1 - wave) * 128; kb0 <= hi; kb0 += 2 * nT) {
      if (kb0 >= in_lo && kb0 + 127 <= in_hi) {
        // ---- interior strip: a lane owns the diagonals kbA = kb0 + 2*lane and kbA + 1; recurrences on packed 16-bit pairs,
        //      the two extension chains are independent (their LDS round trips overlap)
        const int kbA = kb0 + 2 * lane;
        const uint32_t* const dMo = reinterpret_cast<const uint32_t*>(pMo + kbA);
        const uint32_t mo0 = dMo[-1], mo1 = dMo[0], mo2 = dMo[1];   // Mo[kbA-2 .. kbA+3]
        const uint32_t* const dIe = reinterpret_cast<const uint32_t*>(pIe + kbA);
        const uint32_t ie0 = dIe[-1], ie1 = dIe[0];                 // Ie[kbA-2 .. kbA+1]
        const uint32_t* const dDe = reinterpret_cast<const uint32_t*>(pDe + kbA);
        const uint32_t de1 = dDe[0], de2 = dDe[1];                  // De[kbA .. kbA+3]
        const uint32_t mm1 = *reinterpret_cast<const uint32_t*>(pMm + kbA);  // Mm[kbA], Mm[kbA+1]
        const uint32_t insS = pk_max_u16(__builtin_amdgcn_alignbit(mo1, mo0, 16), __builtin_amdgcn_alignbit(ie1, ie0, 16));  // sources at k-1
        const uint32_t del = pk_max_u16(__builtin_amdgcn_alignbit(mo2, mo1, 16), __builtin_amdgcn_alignbit(de2, de1, 16));   // sources at k+1
        const uint32_t ins = pk_inc_nz(insS), mis = pk_inc_nz(mm1);
        const uint32_t mxp = pk_max_u16(del, pk_max_u16(mis, ins));
        const int kA = kbA - koff, kB = kA + 1;
        int32_t offA = (int32_t)(mxp & 0xFFFFu) - 1, offB = (int32_t)(mxp >> 16) - 1;
        const bool okMA = (uint32_t)offA <= (uint32_t)tlen && (uint32_t)(offA - kA) <= (uint32_t)plen;
        const bool okMB = (uint32_t)offB <= (uint32_t)tlen && (uint32_t)(offB - kB) <= (uint32_t)plen;
        // first window of both chains (cells that are not in bounds read window 0 and discard it)
        int vA = okMA ? offA - kA : 0, hA = okMA ? offA : 0, vB = okMB ? offB - kB : 0, hB = okMB ? offB : 0;
        const uint32_t xA = P4[vA] ^ T4[hA], xB = P4[vB] ^ T4[hB];
        uint32_t nA = min(min(ffbl_raw(xA) >> 3, 4u), (uint32_t)min(plen - vA, tlen - hA));
        uint32_t nB = min(min(ffbl_raw(xB) >> 3, 4u), (uint32_t)min(plen - vB, tlen - hB));
        vA += (int)nA; hA += (int)nA; vB += (int)nB; hB += (int)nB;
        bool cA = okMA && nA == 4u, cB = okMB && nB == 4u;
        if (__ballot(cA || cB)) {  // rare: a run of 4+ matches -- one more window per lane, then the wave together (finish_runs)
          if (cA) { const uint32_t xw = P4[vA] ^ T4[hA]; nA = min(min(ffbl_raw(xw) >> 3, 4u), (uint32_t)min(plen - vA, tlen - hA)); vA += (int)nA; hA += (int)nA; cA = nA == 4u; }
          if (cB) { const uint32_t xw = P4[vB] ^ T4[hB]; nB = min(min(ffbl_raw(xw) >> 3, 4u), (uint32_t)min(plen - vB, tlen - hB)); vB += (int)nB; hB += (int)nB; cB = nB == 4u; }
          finish_runs(cA, vA, hA);
          finish_runs(cB, vB, hB);
        }
        if (span == 1) {  // wavefront_termination_endsfree
          if (okMA && ((hA >= tlen && plen - vA <= pef) || (vA >= plen && tlen - hA <= tef)))
            atomicMin(&tn.term_key, ((unsigned long long)(unsigned)(kA + KBIAS) << 32) | (unsigned)hA);
          if (okMB && ((hB >= tlen && plen - vB <= pef) || (vB >= plen && tlen - hB <= tef)))
            atomicMin(&tn.term_key, ((unsigned long long)(unsigned)(kB + KBIAS) << 32) | (unsigned)hB);
        } else {
          if (okMA && kbA == ak_b) tn.end_val = hA;
          if (okMB && kbA + 1 == ak_b) tn.end_val = hB;
        }
        const uint32_t mq = (okMA ? (uint32_t)hA + 1u : 0u) | ((okMB ? (uint32_t)hB + 1u : 0u) << 16);
        *reinterpret_cast<uint32_t*>(qI + kbA) = ins; *reinterpret_cast<uint32_t*>(qD + kbA) = del; *reinterpret_cast<uint32_t*>(qM + kbA) = mq;
        const uint32_t kb2 = 2u * (uint32_t)kbA;
        __builtin_amdgcn_raw_buffer_store_b32((int)ins, hrs, (int)kb2, (int)soI, 0);
        __builtin_amdgcn_raw_buffer_store_b32((int)del, hrs, (int)kb2, (int)soD, 0);
        __builtin_amdgcn_raw_buffer_store_b32((int)mq, hrs, (int)kb2, (int)soM, 0);
        const uint32_t iA = (ins & 0xFFFFu) - 1u, iB = (ins >> 16) - 1u, dA = (del & 0xFFFFu) - 1u, dB = (del >> 16) - 1u;
        const bool okIA = iA <= (uint32_t)tlen && (iA - (uint32_t)kA) <= (uint32_t)plen, okIB = iB <= (uint32_t)tlen && (iB - (uint32_t)kB) <= (uint32_t)plen;
        const bool okDA = dA <= (uint32_t)tlen && (dA - (uint32_t)kA) <= (uint32_t)plen, okDB = dB <= (uint32_t)tlen && (dB - (uint32_t)kB) <= (uint32_t)plen;
        note2(__ballot(okMA), __ballot(okMB), kb0, fM, lM);
        note2(__ballot(okIA), __ballot(okIB), kb0, fI, lI);
        note2(__ballot(okDA), __ballot(okDB), kb0, fD, lD);
      } else {
        // ---- edge strip: two passes of 64 consecutive diagonals, one per lane, every source masked
        for (int half = 0; half < 2; ++half) {
          const int kbs = kb0 + 64 * half, kb = kbs + lane;
          if (kbs > hi) break;
          bool okM = false, okI = false, okD = false;
          const bool inr = kb >= lo && kb <= hi;
          int32_t off_e = 0; unsigned ins_e = 0, del_e = 0;
          if (inr) edge(kb, okM, off_e, ins_e, del_e);
          off_e = extend_u(kb - koff, off_e, okM, tn);
          if (inr) edge_store(kb, okM, off_e, ins_e, del_e, okI, okD);
          note(__ballot(okM), kbs, fM, lM); note(__ballot(okI), kbs, fI, lI); note(__ballot(okD), kbs, fD, lD);
        }
      }
    }
    const uint32_t nlM = 0xFFFFu - (uint32_t)lM, nlI = 0xFFFFu - (uint32_t)lI, nlD = 0xFFFFu - (uint32_t)lD;  // "last" complemented: one packed min folds a record
    LV_MARK(2);
    if (lane == 0) fs.wred[s & 1][wave] = make_uint4((uint32_t)fM | (nlM << 16), (uint32_t)fI | (nlI << 16), (uint32_t)fD | (nlD << 16), 0u);
    // keeps the join of this divergent branch out of the loop's latch block: the uniformity analysis taints every phi of a
    // block where divergent paths join, and the latch block holds the phis of the whole (uniform) per-level state
    asm volatile("" ::: "memory");
  }
  LV_FLUSH;
  FastEnd E;
  E.status = status; E.score = s; E.k = end_k; E.off = end_off; E.cells = cells;
  return E;
}

// ------------------------------------------------------------------------------------------------
// Back-trace of the fast path (gap-affine, single piece).  Same candidate encoding and priorities as
// wf_backtrace (wavefront_backtrace_affine: (offset << 4 | type), maximum wins), but executed by the 64 lanes of
// wave 0 with the level descriptors in LDS, and with gap runs resolved 64 steps per memory round trip: along a pure
// gap-extension chain the next cells are known in advance -- (s - j*e, k + j) for deletions, (s - j*e, k - j) with the
// offset falling by one for insertions -- so lane j fetches step j's two candidates and a ballot finds where the
// chain stops.  Long deletion runs are what the back-trace of a read that lacks the flank consists of.
// ld: descriptors, ld[s * STRIDE + {0: M, 1: I1, 2: D1, 3: base, 4: lo_alloc | width << 16}].  Runs are pushed (reversed) by lane 0.
// A candidate is located first (descriptor lookups, no memory access to the history) and its offset fetched afterwards, so that
// the loads of all candidates of a step go out back to back and the step costs ONE memory round trip, not one per candidate.
struct BtLoc { uint32_t idx; bool ok; };
template <int STRIDE>
__device__ __forceinline__ BtLoc bt_locate(const uint32_t* ld, int cidx, int s, int kb) {
  BtLoc r; r.idx = 0; r.ok = false;
  if (s < 0) return r;
  const uint32_t* d = ld + s * STRIDE;
  const uint32_t rg = d[cidx];
  if (kb < pd_lo(rg) || kb > pd_hi(rg)) return r;
  const uint32_t base = d[3], law = STRIDE == 4 ? g_fsh.flaw[s] : d[4];  // STRIDE 4: the fdesc / flaw rings themselves (score < RING)
  r.idx = base + (uint32_t)cidx * (law >> 16) + (uint32_t)(kb - (int)(law & 0xFFFFu));
  r.ok = true;
  return r;
}
__device__ __forceinline__ long long bt_value(const BtLoc& l, uint32_t enc, int add, int type) {
  if (!l.ok || enc == 0u) return (long long)OFF_NULL;
  return (((long long)((int)enc - 1 + add)) << 4) | type;
}

template <int STRIDE>
__device__ __forceinline__ int wf_backtrace_fast_affine(const Pen& pen, int plen, int tlen, const FastEnd& E, const uint32_t* ld,
                                                        const uint16_t* __restrict__ A16, uint32_t* tmp, uint32_t cap, uint32_t* lruns, uint32_t lcap, int koff) {
  const int lane = threadIdx.x & 63;
  const int x = pen.x, oe = pen.o1 + pen.e1, e = pen.e1;
  int mt = CM, s = E.score, k = E.k, off = E.off;
  int h = off, v = off - k, nt = 0;
  // run-length builder with the open run in registers: the run list in HBM is write-only (rle_push would read the last entry
  // back, a memory round trip per operation); uniform values, lane 0 stores
  uint32_t cur_code = 0; int cur_len = 0;
  auto push = [&](uint32_t code, int len) {
    if (len <= 0) return;
    if (cur_len > 0 && code == cur_code) { cur_len += len; return; }
    if (cur_len > 0) {
      const uint32_t run = ((uint32_t)cur_len << 4) | cur_code;
      if (lane == 0 && (uint32_t)nt < cap) { tmp[nt] = run; if ((uint32_t)nt < lcap) lruns[nt] = run; }  // HBM copy: write-only; LDS copy: what the epilogue reads
      if ((uint32_t)nt < cap) ++nt;
    }
    cur_code = code; cur_len = len;
  };
  auto loc = [&](int cidx, int ss, int kk) { return bt_locate<STRIDE>(ld, cidx, ss, kk + koff); };
  push(2u, plen - v); push(1u, tlen - h);
  while (v > 0 && h > 0 && s > 0) {
    if (mt == CM) {
      const BtLoc l0 = loc(0, s - x, k), l1 = loc(2, s - e, k + 1), l2 = loc(0, s - oe, k + 1), l3 = loc(1, s - e, k - 1), l4 = loc(0, s - oe, k - 1);
      const uint32_t e0 = A16[l0.idx], e1 = A16[l1.idx], e2 = A16[l2.idx], e3 = A16[l3.idx], e4 = A16[l4.idx];  // index 0 when not located
      long long best = bt_value(l0, e0, +1, 9);
      best = max(best, bt_value(l1, e1, 0, 6));
      best = max(best, bt_value(l2, e2, 0, 5));
      best = max(best, bt_value(l3, e3, +1, 2));
      best = max(best, bt_value(l4, e4, +1, 1));
      if (best < 0) break;
      const int best_off = (int)(best >> 4), type = (int)(best & 0xF);
      push(7u, off - best_off);
      off = best_off; h = off; v = off - k;
      if (v <= 0 || h <= 0) break;
      switch (type) {
        case 9: push(8u, 1); s -= x; --off; break;
        case 1: push(1u, 1); s -= oe; --k; --off; break;
        case 2: push(1u, 1); s -= e; mt = CI1; --k; --off; break;
        case 5: push(2u, 1); s -= oe; ++k; break;
        default: push(2u, 1); s -= e; mt = CD1; ++k; break;
      }
      h = off; v = off - k;
    } else {
      const bool del = mt == CD1;
      // ---- lane j inspects step j of the gap-extension chain
      const int sj = s - lane * e, kj = del ? k + lane : k - lane;
      const bool alive = (del ? (v - lane > 0 && h > 0) : (h - lane > 0 && v > 0)) && sj > 0;
      long long ce = (long long)OFF_NULL, co = (long long)OFF_NULL;
      {
        BtLoc le = del ? loc(2, sj - e, kj + 1) : loc(1, sj - e, kj - 1), lo = del ? loc(0, sj - oe, kj + 1) : loc(0, sj - oe, kj - 1);
        if (!alive) { le.ok = false; le.idx = 0; lo.ok = false; lo.idx = 0; }
        const uint32_t ee = A16[le.idx], eo = A16[lo.idx];
        ce = del ? bt_value(le, ee, 0, 6) : bt_value(le, ee, +1, 2);
        co = del ? bt_value(lo, eo, 0, 5) : bt_value(lo, eo, +1, 1);
      }
      const bool cont = alive && ce >= 0 && ce > co;  // the extension candidate is the maximum
      const unsigned long long stop = __ballot(!cont);
      const int j = stop ? __ffsll((long long)stop) - 1 : 64;
      if (j > 0) {  // j pure extension steps
        push(del ? 2u : 1u, j);
        s -= j * e;
        if (del) { k += j; } else { k -= j; off -= j; }
        h = off; v = off - k;
        continue;
      }
      // ---- the chain stops right here: one ordinary step (gap open, or no source at all)
      const BtLoc lx = del ? loc(2, s - e, k + 1) : loc(1, s - e, k - 1), ln = del ? loc(0, s - oe, k + 1) : loc(0, s - oe, k - 1);
      const uint32_t ex = A16[lx.idx], en = A16[ln.idx];
      const long long cext = del ? bt_value(lx, ex, 0, 6) : bt_value(lx, ex, +1, 2);
      const long long copn = del ? bt_value(ln, en, 0, 5) : bt_value(ln, en, +1, 1);
      const long long best = max(cext, copn);
      if (best < 0) break;
      push(del ? 2u : 1u, 1);
      if (best == cext) s -= e; else { s -= oe; mt = CM; }
      if (del) ++k; else { --k; --off; }
      h = off; v = off - k;
    }
  }
  if (mt == CM && v > 0 && h > 0) { const int n = min(v, h); push(7u, n); v -= n; h -= n; }
  push(2u, v);
  push(1u, h);
  if (cur_len > 0) {
    const uint32_t run = ((uint32_t)cur_len << 4) | cur_code;
    if (lane == 0 && (uint32_t)nt < cap) { tmp[nt] = run; if ((uint32_t)nt < lcap) lruns[nt] = run; }
    if ((uint32_t)nt < cap) ++nt;
  }
  return nt;  // uniform
}

// ------------------------------------------------------------------------------------------------
// The kernel: persistent workgroups, one alignment at a time per workgroup (job cost varies by 100x), workspace slot
// acquired per resident workgroup exactly as in wfa_kernel.  LDS (dynamic): ring | pattern windows | text windows.
// TAG names the instantiation (0: the first / only launch of a batch, 1: the launch over the remaining flank alignments, 2: the
// launch over the flank alignments with a seeded window, 3: the launch over the band the pre-filter found -- as 2, but one job per
// claim and the free text start of each job in its JobDev::ops_off), so that a kernel trace tells the launches of
// trgt_find_spans_batch apart.
template <int SPEC, int TAG>
__global__ void __launch_bounds__(256, 4) wfa_fast_kernel(const KArgs a) {
  extern __shared__ unsigned char lds_dyn[];
  FastShared& fs = g_fsh;
  const int tid = threadIdx.x, T = blockDim.x;
  const Pen pen = a.kp.pen;
  if (tid == 0) {
    uint32_t i = (blockIdx.x * 2654435761u) % a.n_slots_ws;
    while (atomicCAS(&a.slot_flags[i], 0u, 1u) != 0u) i = i + 1 == a.n_slots_ws ? 0 : i + 1;
    fs.slot = (int)i;
  }
  __syncthreads();
  const uint32_t ws_slot = rfl((uint32_t)fs.slot);
  uint8_t* const wsb = a.ws + (size_t)ws_slot * a.ws_per_block;
  uint32_t* const gd = reinterpret_cast<uint32_t*>(wsb + a.off_gdesc);
  uint16_t* const A16g = reinterpret_cast<uint16_t*>(wsb + a.off_arena_u);
  uint32_t* const rle_tmp = reinterpret_cast<uint32_t*>(wsb + a.off_rle_tmp);
  uint32_t* const rle_out = reinterpret_cast<uint32_t*>(wsb + a.off_rle_out);
  uint32_t* const run_start = reinterpret_cast<uint32_t*>(wsb + a.off_run_start);
  uint16_t* const ring = reinterpret_cast<uint16_t*>(lds_dyn);
  uint32_t* const P4 = reinterpret_cast<uint32_t*>(lds_dyn + a.fast_ring_bytes);
  const uint32_t n_front = a.n_jobs_dev ? *a.n_jobs_dev : a.n_jobs;
  const uint32_t n_jobs = n_front + (a.n_jobs2_dev ? *a.n_jobs2_dev : 0u);
  unsigned long long cells_acc = 0;
  PROF_DECL;
  // 4-byte sliding windows straight from global memory: a thread turns two (unaligned) dwords into the windows of four positions
  auto stage = [&](const uint8_t* __restrict__ src, int len, uint32_t* __restrict__ W, int t, int nt) {
    for (int i0 = 4 * t; i0 <= len; i0 += 4 * nt) {
      uint32_t d0 = 0, d1 = 0;
      if (i0 + 8 <= len) { __builtin_memcpy(&d0, src + i0, 4); __builtin_memcpy(&d1, src + i0 + 4, 4); }
      else if (len >= 4) {
        // the last positions (the whole wave walks this branch with the one or two threads that take it): a whole dword if it fits, then
        // the one to three bytes left out of the dword that ENDS with the sequence -- two loads and a shift, not eight byte loads
        const int rem = len - i0, tail = rem & 3;
        uint32_t t = 0;
        if (tail) { __builtin_memcpy(&t, src + len - 4, 4); t >>= 8 * (4 - tail); }  // bytes len - tail .. len - 1, zero above
        if (rem >= 4) { __builtin_memcpy(&d0, src + i0, 4); d1 = t; } else d0 = t;
      } else
        for (int b = 0; b < 8; ++b)
          if (i0 + b < len) { if (b < 4) d0 |= (uint32_t)src[i0 + b] << (8 * b); else d1 |= (uint32_t)src[i0 + b] << (8 * (b - 4)); }
      W[i0] = d0; W[i0 + 1] = __builtin_amdgcn_alignbyte(d1, d0, 1); W[i0 + 2] = __builtin_amdgcn_alignbyte(d1, d0, 2);
      W[i0 + 3] = __builtin_amdgcn_alignbyte(d1, d0, 3);
    }
  };
  auto job_at = [&](uint32_t j) { return a.jobs[j < n_front ? j : a.jobs_cap - 1u - (j - n_front)]; };
  // Jobs are claimed one ahead: the index of the next job is fetched while the current one runs, so that the waves that idle during
  // the back-trace of a light alignment (one wave's work) can already build the windows of the next one.
  // (the launch over the windowed alignments takes four jobs per atomic -- 4: 1.22 ms, 8: 1.25, 16: 1.33, 32: 1.43 for the two light launches
  //  together: 128 k three-microsecond jobs on one counter came out at 12.7 ns
  //  apiece whatever the kernel did in between -- the rate of same-address atomics, not of alignments)
  constexpr int CLAIM = TAG == 2 ? 4 : 1;
  auto claim = [&]() {  // thread 0
    if (CLAIM == 1) { fs.job = (int)atomicAdd(a.counter, 1u); return; }
    int nx = fs.job + 1;
    if (nx >= fs.chunk_end) { nx = (int)atomicAdd(a.counter, (unsigned)CLAIM); fs.chunk_end = nx + CLAIM; }
    fs.job = nx;
  };
  __syncthreads();
  if (tid == 0) { fs.job = -1; fs.chunk_end = 0; claim(); }
  __syncthreads();
  uint32_t j = rfl((uint32_t)fs.job);
  bool staged = false;  // windows of job j already built (by waves 1.. during the previous back-trace)
  while (j < n_jobs) {
    PROF_MARK(5);
    PROF_MARK(0);
    const JobDev job = job_at(j);
    const int plen = (int)job.pat_len, tlen = (int)job.txt_len;
    const uint8_t* const P = a.pat_base + job.pat_off;
    const uint8_t* const Tx = a.txt_base + job.txt_off;
    uint32_t* const T4 = P4 + plen + 4;  // (three entries of slack behind each window array: the unguarded tail writes of stage())
    if (!staged) { stage(P, plen, P4, tid, T); stage(Tx, tlen, T4, tid, T); }
    staged = false;
    // the next job: claimed before the level loop in the launch over the light alignments (its latency hides behind the loop, the
    // job is a few microseconds of work), behind the loop in the launch over the expensive ones (a workgroup sitting on a claimed
    // 400-microsecond job while others run dry lengthened that launch by 6 %)
    if (TAG >= 1 && tid == 0) claim();
    FastJob J;
    {
      const int sp = a.kp.span;
      auto fr = [](int v, int len) { return v < 0 || v > len ? len : v; };  // (a free length beyond the sequence means all of it)
      J.plen = plen; J.tlen = tlen; J.span = sp;
      J.pbf = sp ? fr(a.kp.pbf, plen) : 0; J.pef = sp ? fr(a.kp.pef, plen) : 0;
      J.tbf = sp ? fr(a.kp.tbf, tlen) : 0; J.tef = sp ? fr(a.kp.tef, tlen) : 0;
      J.n_slots = (int)a.uni_slots; J.cap = a.arena_uni_cap;
      J.koff = a.fast_koff ? (int)a.fast_koff : plen + 2;
      if (TAG == 3) J.tbf = min(J.tbf, (int)job.ops_off);
    }
    PROF_MARK(1);
    const FastEnd E = wf_run_lds_affine<SPEC, TAG >= 2>(pen, J, P4, T4, ring, (int)a.fast_wcap, (g_u16*)A16g, gd);
#ifdef TRGT_WFA_PROF
    const unsigned long long pf_t2 = pf_t;
#endif
    PROF_MARK(2);
#ifdef TRGT_WFA_PROF
    if (tid == 0) {
      pf_acc[6] += 1; pf_acc[7] += (unsigned long long)E.score;
      if (E.score > 40) { atomicAdd(&g_wfa_prof[16], 1ull); atomicAdd(&g_wfa_prof[17], (unsigned long long)E.score); atomicAdd(&g_wfa_prof[18], pf_t - pf_t2); }
    }
#endif
    cells_acc += E.cells;
    if (TAG == 0 && tid == 0) claim();
    uint32_t j_next = TAG >= 1 ? rfl((uint32_t)fs.job) : 0u;  // TAG == 0: read behind the next barrier
    const bool ok = E.status == ST_END_REACHED;
    int nrun = 0;
    // The run list of the back-trace (reversed) lives in the idle ring area of LDS, behind the staged descriptors, with the run
    // start positions behind it; the HBM copies are only read when a list does not fit there.
    uint32_t* lruns = reinterpret_cast<uint32_t*>(ring);
    uint32_t lcap = 0;
    if (ok && a.kp.scope_alignment && !a.fast_dbg) {
      // back-trace by wave 0; the level descriptors are staged in the (now idle) ring area of LDS when they fit
      const bool in_ring = E.score < RING;  // every level's descriptor is still in the LDS rings: nothing to stage
      const uint32_t ld_bytes = in_ring ? 0u : ((uint32_t)(E.score + 1) * FD_LDS_STRIDE * 4u + 15u) & ~15u;
      const bool fits = ld_bytes <= a.fast_ring_bytes;
      const uint32_t run_base = fits ? ld_bytes : 0u;
      lruns = reinterpret_cast<uint32_t*>(lds_dyn + run_base);
      lcap = (a.fast_ring_bytes - run_base) / 8u;  // runs [0, lcap), run starts [lcap, 2 lcap)
      __syncthreads();  // every wave has left the level loop (ring idle), thread 0's descriptor stores are done
      if (TAG == 0) j_next = rfl((uint32_t)fs.job);
      int nt = 0;
      if (in_ring) {
        if (tid < 64) nt = wf_backtrace_fast_affine<4>(pen, plen, tlen, E, reinterpret_cast<const uint32_t*>(fs.fdesc), A16g, rle_tmp, a.rle_cap, lruns, lcap, J.koff);
        else if (j_next < n_jobs) {  // meanwhile: the windows of the next job (the back-trace touches neither them nor the sequences)
          const JobDev nj = job_at(j_next);
          stage(a.pat_base + nj.pat_off, (int)nj.pat_len, P4, tid - 64, T - 64);
          stage(a.txt_base + nj.txt_off, (int)nj.txt_len, P4 + nj.pat_len + 4, tid - 64, T - 64);
        }
        staged = j_next < n_jobs && T > 64;
      } else if (fits) {
        uint32_t* ld = reinterpret_cast<uint32_t*>(ring);
        for (int i = tid; i < (E.score + 1) * FD_LDS_STRIDE; i += T) {
          const int lvl = i / FD_LDS_STRIDE, c5 = i - lvl * FD_LDS_STRIDE;
          ld[i] = gd[(size_t)lvl * FD_STRIDE + c5];
        }
        __syncthreads();
        if (tid < 64) nt = wf_backtrace_fast_affine<FD_LDS_STRIDE>(pen, plen, tlen, E, ld, A16g, rle_tmp, a.rle_cap, lruns, lcap, J.koff);
      } else if (tid < 64) {
        nt = wf_backtrace_fast_affine<FD_STRIDE>(pen, plen, tlen, E, gd, A16g, rle_tmp, a.rle_cap, lruns, lcap, J.koff);
      }
      if (tid == 0) fs.rle_n = nt;
      __syncthreads();
      nrun = rfl(fs.rle_n);
      if ((uint32_t)nrun > lcap) for (int r = tid; r < nrun; r += T) rle_out[r] = rle_tmp[nrun - 1 - r];  // forward order, through HBM
      PROF_MARK(3);
    }
    const bool lds_runs = (uint32_t)nrun <= lcap;
    uint32_t* const lstart = lruns + lcap;
    auto run_at = [&](int r) -> uint32_t { return lds_runs ? lruns[nrun - 1 - r] : rle_out[r]; };  // forward order
    __syncthreads();
    if (TAG == 0) j_next = rfl((uint32_t)fs.job);
    PROF_MARK(4);
    // ---- per-job epilogue: status, score, count_matches, alignment span, CIGAR, expanded operations (as in wfa_kernel)
    if (tid == 0) {
      const uint32_t o = job.out_index;
      if (a.status) a.status[o] = ok ? TRGT_WF_COMPLETED : (E.status == ST_OOM ? TRGT_WF_OOM : TRGT_WF_UNATTAINABLE);
      if (a.score) a.score[o] = ok ? -E.score : INT32_MIN;
      uint32_t pi = 0, ti = 0, ps = 0, pe = 0, ts = 0, te = 0, nm = 0, total = 0;
      bool started = false;
      for (int r = 0; r < nrun; ++r) {
        const uint32_t en = run_at(r), len = en >> 4, code = en & 0xF;
        if (a.ops) { if (lds_runs) lstart[r] = total; else run_start[r] = total; }
        total += len;
        if (code == 1u) ti += len;
        else if (code == 2u) pi += len;
        else { if (!started) { ps = pi; ts = ti; started = true; } pi += len; ti += len; pe = pi; te = ti; if (code == 7u) nm += len; }
      }
      if (a.kp.span == 0) { ps = 0; pe = (uint32_t)plen; ts = 0; te = (uint32_t)tlen; }
      if (a.n_match) a.n_match[o] = (int32_t)nm;
      if (a.span4) { a.span4[4 * o + 0] = ps; a.span4[4 * o + 1] = pe; a.span4[4 * o + 2] = ts; a.span4[4 * o + 3] = te; }
      if (a.cigar_len) a.cigar_len[o] = (uint32_t)nrun;
      if (a.ops_len) a.ops_len[o] = total;
      fs.total_ops = (int)total;
    }
    if (a.cigar) for (int r = tid; r < nrun; r += T) a.cigar[job.cigar_off + r] = run_at(r);
    if (a.ops && nrun > 0) {
      __syncthreads();
      const uint32_t total = (uint32_t)fs.total_ops;
      for (uint32_t p = tid; p < total; p += T) {
        int lo = 0, hi = nrun - 1;  // last run whose start <= p
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((lds_runs ? lstart[mid] : run_start[mid]) <= p) lo = mid; else hi = mid - 1; }
        const uint32_t code = run_at(lo) & 0xF;
        a.ops[job.ops_off + p] = code == 7u ? 'M' : code == 8u ? 'X' : code == 1u ? 'I' : 'D';
      }
    }
    j = j_next;
  }
  if (tid == 0 && a.cells_out && cells_acc) atomicAdd(a.cells_out, cells_acc);
#ifdef TRGT_WFA_PROF
  PROF_MARK(5);
  if (tid == 0) for (int i = 0; i < 8; ++i) atomicAdd(&g_wfa_prof[i], pf_acc[i]);
#endif
  __syncthreads();
  if (tid == 0) { __threadfence(); atomicExch(&a.slot_flags[ws_slot], 0u); }
}

}  // namespace wfa
}  // namespace trgt
