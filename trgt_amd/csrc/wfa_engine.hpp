// trgt_amd/csrc/wfa_engine.hpp -- device-side wavefront alignment engine for gfx950.
//
// One workgroup works on one alignment at a time (persistent workgroups pull jobs from a global
// counter because job cost varies by 100x).  Diagonals k of a wavefront are mapped to consecutive
// lanes, T = blockDim.x diagonals per strip, so every wavefront read / write is a coalesced
// 4-byte-per-lane access and the trim / termination reductions are one wave ballot + one LDS
// atomic per wave.  The wavefront history needed by the back-trace is written ONCE to the
// workgroup's HBM workspace (4 bytes per offset: the "4*W" term of the roofline model); the
// descriptors (lo, hi, base) of the last 32 score levels are mirrored in LDS so that the
// recurrences never wait on a descriptor load.  Sequences are staged into LDS when they fit.
//
// Semantics restated from WFA2-lib (un-vendored dependency of the reference):
// recurrences / trimming / termination / back-trace priorities / wfadaptive cut-off / BiWFA
// breakpoint search exactly as SURVEY.md Appendix A describes them and as the reference's
// known-answer tests (src/wfaligner.rs:1136-1828) pin them.
#pragma once
#include "common.hpp"
#include "wfa_host.hpp"

namespace trgt {
namespace wfa {

constexpr int32_t OFF_NULL = INT32_MIN / 2;  // WAVEFRONT_OFFSET_NULL
constexpr uint32_t NOBASE = 0xFFFFFFFFu;
constexpr int RING = 32;                      // LDS mirror depth of wavefront descriptors (>= max_score_scope)
constexpr int KBIAS = 1 << 30;
enum { CM = 0, CI1 = 1, CI2 = 2, CD1 = 3, CD2 = 4 };
enum { ST_OK = 0, ST_END_REACHED = 1, ST_END_UNREACHABLE = 2, ST_OOM = 3 };
enum { M_INDEL = 0, M_EDIT = 1, M_LINEAR = 2, M_AFFINE = 3, M_AFFINE2P = 4 };
enum { I_FWD = 0, I_REV = 1, I_UNI = 2 };

struct WfDesc { int lo, hi, lo_alloc; uint32_t base; };  // base == NOBASE: wavefront pointer is NULL; lo > hi: ->null

struct Pen { int metric, x, o1, e1, o2, e2, scope, ncomp; };

struct KParams {  // by-value kernel argument
  Pen pen;
  int span, pbf, pef, tbf, tef;  // -1 = sequence length
  int scope_alignment, biwfa, heuristic, h_min_len, h_max_dist, h_steps, bi_min_score, bi_min_length;
};

struct KArgs {
  KParams kp;
  const JobDev* jobs; const uint32_t* n_jobs_dev; uint32_t n_jobs;
  const uint32_t* n_jobs2_dev; uint32_t jobs_cap;   // optional second part of the list, stored downwards from jobs[jobs_cap - 1]
  const uint8_t* pat_base; const uint8_t* txt_base;
  unsigned int* counter;
  unsigned int* slot_flags; uint32_t n_slots_ws, jobs_per_block;  // workspace slots are acquired per resident workgroup
  uint8_t* ws; uint64_t ws_per_block;
  uint64_t off_gdesc, off_arena_u, off_arena_f, off_arena_r, off_rle_tmp, off_rle_out, off_run_start;
  uint32_t uni_slots, arena_uni_cap, ring_stride, rle_cap, lds_seq_cap, fast_wcap, fast_ring_bytes, fast_dbg, fast_koff /* bias of the diagonal index in the ring; 0 = pattern length + 2 */;
  int32_t* status; int32_t* score; int32_t* n_match; uint32_t* span4; uint32_t* cigar; uint32_t* cigar_len; uint8_t* ops; uint32_t* ops_len;
  unsigned long long* cells_out;
  unsigned int* refused;  // WfaLaunch::refused (may be null)
  // LDS-arena variant: byte offsets inside the dynamic LDS and capacities; jobs it cannot hold go to the retry list (the HBM variant)
  uint32_t la_rle_tmp, la_rle_out, la_rle_cap, la_region, la_region_bytes, la_gdesc_slots, la_seq_max;
  JobDev* retry_jobs; unsigned int* retry_count;
  uint32_t ring_off, ring_mask;  // descriptor rings in the dynamic LDS (behind the sequences): ring_mask + 1 levels (a power of two >= the score scope)
};

struct Inst {  // one unidirectional aligner (forward / reverse / base)
  const uint8_t* pp; const uint8_t* tp;
  int plen, tlen, rev;
  int span, pbf, pef, tbf, tef, cb, ce;
  int modular;
  WfDesc* gdesc; int n_slots;           // global descriptor history (full-history instance only)
  int32_t* arena; uint32_t arena_cap, bump, stride;
  int num_null_steps, steps_wait, status, end_score, end_k, end_off, cur;
  uint32_t arena_lds, pp_lds, tp_lds;   // LDS-arena variant (LA): byte offsets of the arena / the two sequences in the dynamic LDS
};

struct Red {
  int lo[5], hi[5];
  unsigned long long term_key;  // (k + KBIAS) << 32 | offset, minimum = first terminating diagonal
  int end_val, max_ak, min_dist, cand_lo, cand_hi, bp_k, flag, oom;
};


struct Breakpoint { int score, score_f, score_r, k_f, off_f, comp; };
struct Seg { int pb, pl, tb, tl, cb, ce, rem, top; };

constexpr int BI_STACK = 48;  // explicit recursion stack of the BiWFA driver (two entries per level of the split tree)
struct Shared {
  Inst inst[3];
  Red red;
  Breakpoint bp;
  Seg stack[BI_STACK];
  int sp, job, status, score, top_bp, rle_n, rle_tmp_n;
  unsigned long long cells;
  // the kernel argument, copied once per workgroup: the engine functions read it here (a reference to the kernel argument itself handed to
  // a non-inlined function forces a private copy of all of it, and every field read becomes a scratch load)
  KArgs args;
  uint32_t ring_off, ring_mask;  // descriptor rings of the three instances in the dynamic LDS: [3][ring_mask + 1][5] WfDesc at lds_dyn + ring_off
};

// Workgroup state lives in one file-scope LDS object so that the (non-inlined) engine functions address it as LDS.
__shared__ Shared g_sh;
#define sh g_sh

// Dynamic LDS of wfa_kernel: [pattern | text] staged per job, and -- LDS-arena variant (template parameter LA of everything below that
// touches wavefront offsets) -- the run-length buffers and the wavefront arenas of the current alignment.  With LA the arena and the
// sequences are addressed as offsets into this array, so that the compiler emits LDS instructions (a pointer kept in the Inst record
// would make every access a flat one); without it they are the HBM workspace pointers of the Inst record.
extern __shared__ unsigned char lds_dyn[];
#ifdef TRGT_WFA_PROF
// engine profile (make PROF=1): thread-0 clocks per section of a score level -- [0] fetch of the source descriptors, [1] thread 0's
// allocation / descriptor block, [2] strips (loads, recurrences, extension, stores, range reductions), [3] closing descriptors,
// [4] post_extend, [5] heuristic cut-off, [6] bialign overlap, [7] init, [8] extend_only, [9] back-trace, [10] levels counted
__device__ unsigned long long g_wfa_eprof[16];
#define EP_DECL unsigned long long ep_t = clock64()
#define EP_MARK(i) do { const unsigned long long n_ = clock64(); if (threadIdx.x == 0) atomicAdd(&g_wfa_eprof[i], n_ - ep_t); ep_t = n_; } while (0)
#define EP_COUNT(i) do { if (threadIdx.x == 0) atomicAdd(&g_wfa_eprof[i], 1ull); } while (0)
#else
#define EP_DECL
#define EP_MARK(i)
#define EP_COUNT(i)
#endif
__device__ __forceinline__ WfDesc& ring_at(int ii, int s, int c) {
  return reinterpret_cast<WfDesc*>(lds_dyn + sh.ring_off)[((uint32_t)ii * (sh.ring_mask + 1u) + ((uint32_t)s & sh.ring_mask)) * 5u + (uint32_t)c];
}
#define KP (sh.args.kp)
template <bool LA> __device__ __forceinline__ int32_t* arena_of(const Inst& I) {
  if constexpr (LA) return reinterpret_cast<int32_t*>(lds_dyn + I.arena_lds); else return I.arena;
}
template <bool LA> __device__ __forceinline__ const uint8_t* pat_of(const Inst& I) { if constexpr (LA) return lds_dyn + I.pp_lds; else return I.pp; }
template <bool LA> __device__ __forceinline__ const uint8_t* txt_of(const Inst& I) { if constexpr (LA) return lds_dyn + I.tp_lds; else return I.tp; }

// run-length CIGAR building block (len << 4 | code), merging equal neighbours
__device__ __forceinline__ void rle_push(uint32_t* buf, int& n, uint32_t cap, uint32_t code, int len) {
  if (len <= 0) return;
  if (n > 0 && (buf[n - 1] & 0xF) == code) { buf[n - 1] += (uint32_t)len << 4; return; }
  if ((uint32_t)n < cap) buf[n++] = ((uint32_t)len << 4) | code;
}

__device__ __forceinline__ uint8_t seq_at(const uint8_t* p, int len, int rev, int i) { return p[rev ? len - 1 - i : i]; }

__device__ __forceinline__ WfDesc null_desc() { WfDesc d; d.lo = 1; d.hi = -1; d.lo_alloc = 1; d.base = NOBASE; return d; }

// wavefront_compute_get_*wavefront: NULL pointer or ->null are replaced by the canonical null wavefront
__device__ __forceinline__ WfDesc fetch(int ii, int c, int s) {
  if (s < 0) return null_desc();
  WfDesc d = ring_at(ii, s, c);
  if (d.base == NOBASE || d.lo > d.hi) return null_desc();
  return d;
}
// raw pointer semantics (may be ->null); valid for s within the last RING levels
__device__ __forceinline__ WfDesc fetch_raw(int ii, int c, int s) {
  if (s < 0) return null_desc();
  return ring_at(ii, s, c);
}
__device__ __forceinline__ int32_t wf_get(const int32_t* __restrict__ arena, const WfDesc& d, int k) {
  return (k >= d.lo && k <= d.hi) ? arena[d.base + (uint32_t)(k - d.lo_alloc)] : OFF_NULL;
}
__device__ __forceinline__ bool in_bounds(int32_t off, int k, int plen, int tlen) {
  return (uint32_t)off <= (uint32_t)tlen && (uint32_t)(off - k) <= (uint32_t)plen;
}
__device__ __forceinline__ void lim(const WfDesc& w, int dlo, int dhi, int& lo, int& hi) {
  if (lo > w.lo + dlo) lo = w.lo + dlo;
  if (hi < w.hi + dhi) hi = w.hi + dhi;
}
// wave-level "first / last lane with valid" -> LDS min/max (lanes hold ascending k)
__device__ __forceinline__ void red_range(bool valid, int k, int* lo, int* hi) {
  const unsigned long long m = __ballot(valid);
  if (m) {
    const int lane = threadIdx.x & 63;
    if (lane == __ffsll((long long)m) - 1) atomicMin(lo, k);
    if (lane == 63 - __clzll((long long)m)) atomicMax(hi, k);
  }
}
__device__ __forceinline__ void red_first(bool valid, int k, int* lo) {
  const unsigned long long m = __ballot(valid);
  if (m && (int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) atomicMin(lo, k);
}
__device__ __forceinline__ void red_last(bool valid, int k, int* hi) {
  const unsigned long long m = __ballot(valid);
  if (m && (int)(threadIdx.x & 63) == 63 - __clzll((long long)m)) atomicMax(hi, k);
}
__device__ __forceinline__ int wave_max(int v) {
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ int wave_min(int v) {
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
  return v;
}

__device__ __forceinline__ void red_reset(Red& r) {
  for (int c = 0; c < 5; ++c) { r.lo[c] = INT32_MAX; r.hi[c] = INT32_MIN; }
  r.term_key = ~0ull; r.end_val = OFF_NULL; r.max_ak = 0; r.min_dist = INT32_MAX; r.cand_lo = INT32_MAX; r.cand_hi = INT32_MIN;
  r.bp_k = INT32_MAX; r.oom = 0;
}

// Extension of the M cells of a wave (wavefront_extend_matches_packed_*), in uniform control flow: `on` = this lane has a cell at
// offset `off` on diagonal k.  Every lane first compares eight bases of its own (most cells of a wavefront stop inside them); a cell
// that matched all eight is almost surely on the diagonal of the alignment itself -- these are alignments of near-identical sequences,
// a read of an allele against its consensus -- and is finished by the whole wave: lane i compares the eight bases 8 i further on, 512
// bases per memory round trip (a lane stepping along such a diagonal alone took one dependent round trip per eight bases -- ~1 500
// cycles each, whether the sequences sit in LDS behind a generic pointer or in global memory -- with the other 63 lanes waiting: most
// of this kernel's time, tools/biwfa_probe.py).  Reverse sequences (BiWFA's backward fronts) read the eight bytes that END at the
// position and count equal bytes from the top.  Returns the extended offset (h); same result as a base-by-base walk.
template <bool LA>
__device__ __forceinline__ int32_t extend_wave(const Inst& I, int k, int32_t off, bool on) {
  const int plen = I.plen, tlen = I.tlen, rev = I.rev;
  const uint8_t* pp = pat_of<LA>(I); const uint8_t* tp = txt_of<LA>(I);
  // bases that match from (v, h) on, looking at eight of them at most (fewer at the end of a sequence)
  auto chunk = [&](int v, int h) -> int {
    const int rem = min(plen - v, tlen - h);
    if (rem >= 8) {
      uint64_t a, b;
      if (!rev) { __builtin_memcpy(&a, pp + v, 8); __builtin_memcpy(&b, tp + h, 8); }
      else { __builtin_memcpy(&a, pp + (plen - v - 8), 8); __builtin_memcpy(&b, tp + (tlen - h - 8), 8); }
      const uint64_t x = a ^ b;
      if (!x) return 8;
      return (!rev ? __builtin_ctzll(x) : __builtin_clzll(x)) >> 3;
    }
    int n = 0;
    while (n < rem && seq_at(pp, plen, rev, v + n) == seq_at(tp, tlen, rev, h + n)) ++n;
    return n;  // (< 8: the run ends here, by a mismatch or with the sequence)
  };
  int v = on ? off - k : 0, h = on ? off : 0;
  bool going = false;
  if (on) { const int n = chunk(v, h); v += n; h += n; going = n == 8; }
  unsigned long long m = __ballot(going);
  const int lane = (int)(threadIdx.x & 63u);
  while (m) {
    const int j = (int)__builtin_ctzll(m);
    m &= m - 1ull;
    int vj = __builtin_amdgcn_readlane(v, j), hj = __builtin_amdgcn_readlane(h, j);
    for (;;) {
      const int n = chunk(vj + 8 * lane, hj + 8 * lane);  // (beyond either sequence: rem <= 0, nothing matches)
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
// ... and what the wavefront's reductions take from an extended cell (termination, end cell, antidiagonal)
__device__ __forceinline__ void extended_cell(const Inst& I, Red& red, int k, int32_t off, bool want_ak, int ak) {
  if (I.span == 1) {  // wavefront_termination_endsfree
    const int h = off, v = off - k;
    if ((h >= I.tlen && I.plen - v <= I.pef) || (v >= I.plen && I.tlen - h <= I.tef))
      atomicMin(&red.term_key, ((unsigned long long)(unsigned)(k + KBIAS) << 32) | (unsigned)off);
  }
  if (k == ak && I.ce == CM) red.end_val = off;
  if (want_ak) atomicMax(&red.max_ak, 2 * off - k);
}

// thread 0: publish a finished descriptor (LDS mirror + global history)
__device__ __forceinline__ void put_desc(int ii, int c, int s, const WfDesc& d) {
  ring_at(ii, s, c) = d;
  Inst& I = sh.inst[ii];
  if (!I.modular && s < I.n_slots) I.gdesc[(size_t)s * 5 + c] = d;
}

// thread 0: allocate `width` offsets for (score s, component slot ci of ncomp)
__device__ __forceinline__ uint32_t wf_alloc(Inst& I, Red& red, int s, int ci, int ncomp, int scope, int width) {
  if (width < 0) width = 0;
  if (I.modular) {
    if ((uint32_t)width > I.stride) { red.oom = 1; return 0; }
    return (uint32_t)((s % scope) * ncomp + ci) * I.stride;
  }
  if (s >= I.n_slots || (unsigned long long)I.bump + (unsigned)width > I.arena_cap) { red.oom = 1; return 0; }
  const uint32_t b = I.bump;
  I.bump += (uint32_t)width;
  return b;
}

// ------------------------------------------------------------------------------------------------
// wavefront_unialign_init: wavefront zero (+ heuristic clear).  All threads.
template <bool LA>
__device__ __noinline__ void wf_init(int ii) {
  const KParams& kp = KP;
  Inst& I = sh.inst[ii];
  int32_t* const AR = arena_of<LA>(I);
  const int tid = threadIdx.x, T = blockDim.x;
  EP_DECL;
  __syncthreads();
  if (tid == 0) {
    I.num_null_steps = 0; I.status = ST_OK; I.end_score = -1; I.steps_wait = kp.h_steps; I.bump = 0; I.cur = 0;
    red_reset(sh.red);
    for (int c = 0; c < 5; ++c) put_desc(ii, c, 0, null_desc());
    const int ncomp = kp.pen.ncomp;
    WfDesc d;
    if (I.span == 0) {
      d.lo = d.hi = d.lo_alloc = 0;
      static const int slot_of[5] = {0, 1, 3, 2, 4};  // component -> allocation slot (M, I1, D1, I2, D2)
      d.base = wf_alloc(I, sh.red, 0, slot_of[I.cb], ncomp, kp.pen.scope, 1);
      if (!sh.red.oom) { AR[d.base] = 0; put_desc(ii, I.cb, 0, d); }
      sh.cells += 1;
    } else {
      d.lo = d.lo_alloc = -I.pbf; d.hi = I.tbf;
      d.base = wf_alloc(I, sh.red, 0, 0, ncomp, kp.pen.scope, I.tbf + I.pbf + 1);
      if (!sh.red.oom) put_desc(ii, CM, 0, d);
      sh.cells += (unsigned long long)(I.tbf + I.pbf + 1);
    }
  }
  __syncthreads();
  if (sh.red.oom) { if (tid == 0) I.status = ST_OOM; __syncthreads(); return; }
  if (I.span == 1) {
    const WfDesc d = ring_at(ii, 0, CM);
    for (int k = d.lo + tid; k <= d.hi; k += T) AR[d.base + (uint32_t)(k - d.lo_alloc)] = k > 0 ? k : 0;
  }
  __syncthreads();
  EP_MARK(7);
}

// Extension of an existing M wavefront (score 0).  All threads.
template <bool LA>
__device__ __noinline__ void wf_extend_only(int ii, int s, bool want_ak) {
  Inst& I = sh.inst[ii];
  int32_t* const AR = arena_of<LA>(I);
  const int tid = threadIdx.x, T = blockDim.x;
  const WfDesc d = fetch_raw(ii, CM, s);
  EP_DECL;
  if (tid == 0) red_reset(sh.red);
  __syncthreads();
  if (d.base != NOBASE) {
    const int ak = I.tlen - I.plen;
    for (int kb = d.lo; kb <= d.hi; kb += T) {  // (whole waves: the extension is a wave's joint work)
      const int k = kb + tid;
      int32_t off = k <= d.hi ? AR[d.base + (uint32_t)(k - d.lo_alloc)] : OFF_NULL;
      const bool on = off >= 0;
      off = extend_wave<LA>(I, k, off, on);
      if (on) {
        extended_cell(I, sh.red, k, off, want_ak, ak);
        AR[d.base + (uint32_t)(k - d.lo_alloc)] = off;
      }
    }
    if (I.ce != CM && tid == 0) {  // end component other than M at score 0: its single cell
      const WfDesc e = fetch_raw(ii, I.ce, s);
      if (e.base != NOBASE && ak >= e.lo && ak <= e.hi) sh.red.end_val = AR[e.base + (uint32_t)(ak - e.lo_alloc)];
    }
  }
  __syncthreads();
  EP_MARK(8);
}

// ------------------------------------------------------------------------------------------------
// wavefront_compute_{edit,linear,affine,affine2p} fused with the extension of the new M wavefront.
// All threads.  On return the descriptors of score s are published and the reductions hold the
// termination / antidiagonal data of M[s].
template <int METRIC, bool LA>
__device__ __noinline__ void wf_compute_extend(int ii, int s, bool want_ak) {
  const KParams& kp = KP;
  Inst& I = sh.inst[ii];
  Red& red = sh.red;
  const Pen& pen = kp.pen;
  const int tid = threadIdx.x, T = blockDim.x;
  const int plen = I.plen, tlen = I.tlen, ak = tlen - plen;
  constexpr int NCOMP = METRIC <= M_LINEAR ? 1 : (METRIC == M_AFFINE ? 3 : 5);
  EP_DECL;
  EP_COUNT(10);
  // ---- inputs (uniform)
  WfDesc m_mis = null_desc(), m_o1 = null_desc(), m_o2 = null_desc(), i1e = null_desc(), d1e = null_desc(), i2e = null_desc(), d2e = null_desc();
  int lo, hi;
  bool all_null;
  if (METRIC <= M_EDIT) {
    m_mis = fetch_raw(ii, CM, s - 1);  // raw: wavefront_compute_edit uses the previous wavefront as is
    if (m_mis.base == NOBASE) m_mis = null_desc();
    lo = m_mis.lo - 1; hi = m_mis.hi + 1;
    all_null = false;
  } else if (METRIC == M_LINEAR) {
    m_mis = fetch(ii, CM, s - pen.x); m_o1 = fetch(ii, CM, s - pen.o1);
    all_null = m_mis.base == NOBASE && m_o1.base == NOBASE;
    lo = m_mis.lo; hi = m_mis.hi; lim(m_o1, -1, +1, lo, hi);
  } else {
    m_mis = fetch(ii, CM, s - pen.x); m_o1 = fetch(ii, CM, s - pen.o1 - pen.e1);
    i1e = fetch(ii, CI1, s - pen.e1); d1e = fetch(ii, CD1, s - pen.e1);
    all_null = m_mis.base == NOBASE && m_o1.base == NOBASE && i1e.base == NOBASE && d1e.base == NOBASE;
    lo = m_mis.lo; hi = m_mis.hi;
    lim(m_o1, -1, +1, lo, hi); lim(i1e, +1, +1, lo, hi); lim(d1e, -1, -1, lo, hi);
    if (METRIC == M_AFFINE2P) {
      m_o2 = fetch(ii, CM, s - pen.o2 - pen.e2); i2e = fetch(ii, CI2, s - pen.e2); d2e = fetch(ii, CD2, s - pen.e2);
      all_null = all_null && m_o2.base == NOBASE && i2e.base == NOBASE && d2e.base == NOBASE;
      lim(m_o2, -1, +1, lo, hi); lim(i2e, +1, +1, lo, hi); lim(d2e, -1, -1, lo, hi);
    }
  }
  __syncthreads();  // everyone has read the ring before thread 0 overwrites the slot of score s
  EP_MARK(0);
  if (all_null) {  // wavefront_compute_allocate_output_null
    if (tid == 0) {
      I.num_null_steps += 1;
      for (int c = 0; c < 5; ++c) put_desc(ii, c, s, null_desc());
      red_reset(red);
      I.cur = s;
    }
    __syncthreads();
    return;
  }
  const bool has_i1 = m_o1.base != NOBASE || i1e.base != NOBASE, has_d1 = m_o1.base != NOBASE || d1e.base != NOBASE;
  const bool has_i2 = m_o2.base != NOBASE || i2e.base != NOBASE, has_d2 = m_o2.base != NOBASE || d2e.base != NOBASE;
  const int width = hi - lo + 1;
  if (tid == 0) {
    I.num_null_steps = 0;
    red_reset(red);
    for (int c = 0; c < 5; ++c) put_desc(ii, c, s, null_desc());
    WfDesc d; d.lo = d.lo_alloc = lo; d.hi = hi;
    d.base = wf_alloc(I, red, s, 0, NCOMP, pen.scope, width); ring_at(ii, s, CM) = d;
    if (NCOMP >= 3) {
      d.base = wf_alloc(I, red, s, 1, NCOMP, pen.scope, width); ring_at(ii, s, CI1) = d;
      d.base = wf_alloc(I, red, s, 2, NCOMP, pen.scope, width); ring_at(ii, s, CD1) = d;
    }
    if (NCOMP == 5) {
      d.base = wf_alloc(I, red, s, 3, NCOMP, pen.scope, width); ring_at(ii, s, CI2) = d;
      d.base = wf_alloc(I, red, s, 4, NCOMP, pen.scope, width); ring_at(ii, s, CD2) = d;
    }
    sh.cells += (unsigned long long)(width > 0 ? width : 0) * NCOMP;
  }
  __syncthreads();
  EP_MARK(1);
  if (red.oom) { if (tid == 0) { I.status = ST_OOM; I.cur = s; } __syncthreads(); return; }
  const uint32_t bM = ring_at(ii, s, CM).base;
  const uint32_t bI1 = ring_at(ii, s, CI1).base, bD1 = ring_at(ii, s, CD1).base;
  const uint32_t bI2 = ring_at(ii, s, CI2).base, bD2 = ring_at(ii, s, CD2).base;
  int32_t* __restrict__ A = arena_of<LA>(I);
  // ---- strips of T diagonals
  for (int kb = lo; kb <= hi; kb += T) {
    const int k = kb + tid;
    const bool act = k <= hi;
    const uint32_t idx = (uint32_t)(k - lo);
    int32_t mx = OFF_NULL, ins1 = OFF_NULL, del1 = OFF_NULL, ins2 = OFF_NULL, del2 = OFF_NULL;
    if (act) {
      if (METRIC == M_EDIT) {
        const int32_t ins = wf_get(A, m_mis, k - 1), del = wf_get(A, m_mis, k + 1), mis = wf_get(A, m_mis, k);
        mx = max(del, max(ins, mis) + 1);
      } else if (METRIC == M_INDEL) {
        const int32_t ins = wf_get(A, m_mis, k - 1), del = wf_get(A, m_mis, k + 1);
        mx = max(del, ins + 1);
      } else if (METRIC == M_LINEAR) {
        const int32_t ins = wf_get(A, m_o1, k - 1), del = wf_get(A, m_o1, k + 1), mis = wf_get(A, m_mis, k);
        mx = max(del, max(mis, ins) + 1);
      } else {
        ins1 = max(wf_get(A, m_o1, k - 1), wf_get(A, i1e, k - 1)) + 1;
        del1 = max(wf_get(A, m_o1, k + 1), wf_get(A, d1e, k + 1));
        const int32_t mis = wf_get(A, m_mis, k) + 1;
        int32_t ins = ins1, del = del1;
        if (METRIC == M_AFFINE2P) {
          ins2 = max(wf_get(A, m_o2, k - 1), wf_get(A, i2e, k - 1)) + 1;
          del2 = max(wf_get(A, m_o2, k + 1), wf_get(A, d2e, k + 1));
          ins = max(ins1, ins2); del = max(del1, del2);
        }
        mx = max(del, max(mis, ins));
        A[bI1 + idx] = ins1; A[bD1 + idx] = del1;
        if (METRIC == M_AFFINE2P) { A[bI2 + idx] = ins2; A[bD2 + idx] = del2; }
      }
      if (!in_bounds(mx, k, plen, tlen)) mx = OFF_NULL;  // "adjust offset out of boundaries"
    }
    EP_MARK(11);
    {
      const bool on = act && mx >= 0;
      const int32_t ext = extend_wave<LA>(I, k, mx, on);
      EP_MARK(12);
      if (on) { mx = ext; extended_cell(I, red, k, mx, want_ak, ak); }
    }
    EP_MARK(13);
    if (act) {
      A[bM + idx] = mx;
      if (NCOMP >= 3 && k == ak && I.ce != CM)
        red.end_val = I.ce == CI1 ? ins1 : I.ce == CD1 ? del1 : I.ce == CI2 ? ins2 : del2;
    }
    // wavefront_compute_trim_ends: first / last in-bounds diagonal of every component
    red_range(act && mx >= 0, k, &red.lo[CM], &red.hi[CM]);
    if (NCOMP >= 3) {
      red_range(act && in_bounds(ins1, k, plen, tlen), k, &red.lo[CI1], &red.hi[CI1]);
      red_range(act && in_bounds(del1, k, plen, tlen), k, &red.lo[CD1], &red.hi[CD1]);
    }
    if (NCOMP == 5) {
      red_range(act && in_bounds(ins2, k, plen, tlen), k, &red.lo[CI2], &red.hi[CI2]);
      red_range(act && in_bounds(del2, k, plen, tlen), k, &red.lo[CD2], &red.hi[CD2]);
    }
    EP_MARK(14);
  }
  __syncthreads();
  EP_MARK(2);
  if (tid == 0) {
    auto fin = [&](int c, bool exists) {
      WfDesc d = ring_at(ii, s, c);
      if (!exists) d = null_desc();
      else if (red.lo[c] == INT32_MAX) { d.hi = d.lo - 1; }  // nothing in bounds: ->null (lo > hi)
      else { d.lo = red.lo[c]; d.hi = red.hi[c]; }
      put_desc(ii, c, s, d);
    };
    fin(CM, true);
    if (NCOMP >= 3) { fin(CI1, has_i1); fin(CD1, has_d1); }
    if (NCOMP == 5) { fin(CI2, has_i2); fin(CD2, has_d2); }
    if (METRIC <= M_EDIT && red.lo[CM] == INT32_MAX) I.num_null_steps = INT32_MAX;
    I.cur = s;
  }
  __syncthreads();
  EP_MARK(3);
}

// wf_distance_end2end / _endsfree for the wfadaptive cut-off
__device__ __forceinline__ int wf_dist(const Inst& I, int32_t off, int k) {
  if (off < 0) return -OFF_NULL;
  const int left_v = I.plen - (off - k), left_h = I.tlen - off;
  if (I.span == 0) return max(left_v, left_h);
  return min(max(left_h, left_v - I.pef), max(left_v, left_h - I.tef));
}

// wavefront_heuristic_cufoff (wfadaptive).  All threads.
template <bool LA>
__device__ __noinline__ void wf_heuristic_cutoff(int ii, int s) {
  const KParams& kp = KP;
  Inst& I = sh.inst[ii];
  Red& red = sh.red;
  const int tid = threadIdx.x, T = blockDim.x;
  const WfDesc m = fetch_raw(ii, CM, s);
  if (m.base == NOBASE || m.lo > m.hi) return;  // uniform
  EP_DECL;
  __syncthreads();
  if (tid == 0) { I.steps_wait -= 1; red.min_dist = max(I.plen, I.tlen); red.cand_lo = INT32_MAX; red.cand_hi = INT32_MIN; }
  __syncthreads();
  const bool run = I.steps_wait <= 0 && (m.hi - m.lo + 1) >= kp.h_min_len;
  if (run) {
    const int32_t* __restrict__ A = arena_of<LA>(I);
    for (int kb = m.lo; kb <= m.hi; kb += T) {
      const int k = kb + tid;
      int d = INT32_MAX;
      if (k <= m.hi) d = wf_dist(I, A[m.base + (uint32_t)(k - m.lo_alloc)], k);
      d = wave_min(d);
      if ((tid & 63) == 0) atomicMin(&red.min_dist, d);
    }
    __syncthreads();
    const int min_d = red.min_dist, ak = I.tlen - I.plen, thr = kp.h_max_dist;
    const int top_limit = min(ak, m.hi), bottom_limit = max(ak, m.lo);  // wf_heuristic_wfadaptive_reduce
    for (int kb = m.lo; kb <= m.hi; kb += T) {
      const int k = kb + tid;
      bool keep = false;
      if (k <= m.hi) keep = wf_dist(I, A[m.base + (uint32_t)(k - m.lo_alloc)], k) - min_d <= thr;
      red_first(keep && k < top_limit, k, &red.cand_lo);
      red_last(keep && k > bottom_limit, k, &red.cand_hi);
    }
    __syncthreads();
  }
  if (tid == 0) {
    WfDesc d = m;
    if (run) {
      const int ak = I.tlen - I.plen;
      const int top_limit = min(ak, m.hi);
      int lo_red = m.lo;
      if (top_limit > m.lo) lo_red = red.cand_lo != INT32_MAX ? red.cand_lo : top_limit;
      d.lo = lo_red;
      const int bottom_limit = max(ak, d.lo);
      int hi_red = m.hi;
      if (m.hi > bottom_limit) hi_red = (red.cand_hi != INT32_MIN && red.cand_hi > bottom_limit) ? red.cand_hi : bottom_limit;
      d.hi = hi_red;
      I.steps_wait = kp.h_steps;
    }
    put_desc(ii, CM, s, d);
    if (kp.pen.metric > M_LINEAR) {  // wavefront_heuristic_equate
      auto equate = [&](int c) {
        WfDesc e = ring_at(ii, s, c);
        if (e.base == NOBASE) return;
        if (d.lo > e.lo) e.lo = d.lo;
        if (d.hi < e.hi) e.hi = d.hi;
        put_desc(ii, c, s, e);
      };
      equate(CI1); equate(CD1);
      if (kp.pen.metric == M_AFFINE2P) { equate(CI2); equate(CD2); }
    }
  }
  __syncthreads();
  EP_MARK(5);
}

// Post-extension part of wavefront_extend_{end2end,end2end_max,endsfree}.  All threads; returns 1 when done.
template <bool LA>
__device__ __noinline__ int wf_post_extend(int ii, int s, bool act_on_end, int* max_ak) {
  const KParams& kp = KP;
  Inst& I = sh.inst[ii];
  Red& red = sh.red;
  const int tid = threadIdx.x;
  EP_DECL;
  __syncthreads();
  if (tid == 0) {
    int done = 0, cont_heur = 0;
    const WfDesc m = fetch_raw(ii, CM, s);
    const bool m_ptr = m.base != NOBASE, m_null = m.lo > m.hi;
    bool stop_here = false;
    if (I.status == ST_OOM) { done = 1; stop_here = true; }
    if (!stop_here && (!m_ptr || m_null)) {
      if (!m_ptr || kp.pen.metric <= M_EDIT) {
        if (I.num_null_steps > kp.pen.scope) { I.status = ST_END_UNREACHABLE; I.end_score = s; done = 1; stop_here = true; }
      }
      if (!stop_here && !m_ptr) stop_here = true;  // not done, nothing to extend
    }
    if (!stop_here) {
      bool end_reached = false;
      if (I.span == 1) {
        if (red.term_key != ~0ull) {
          end_reached = true;
          I.end_score = s; I.end_k = (int)(red.term_key >> 32) - KBIAS; I.end_off = (int)(red.term_key & 0xFFFFFFFFu);
        }
      } else {
        const int ak = I.tlen - I.plen;
        const WfDesc e = fetch_raw(ii, I.ce, s);
        if (e.base != NOBASE && ak >= e.lo && ak <= e.hi && red.end_val >= I.tlen) {
          end_reached = true; I.end_score = s; I.end_k = ak; I.end_off = I.tlen;
        }
      }
      if (end_reached && act_on_end) { I.status = ST_END_REACHED; done = 1; }
      else cont_heur = kp.heuristic != 0;
    }
    red.flag = done | (cont_heur << 1);
  }
  __syncthreads();
  const int f = red.flag;
  const int mak = red.max_ak;
  EP_MARK(4);
  if (f & 2) wf_heuristic_cutoff<LA>(ii, s);
  if (max_ak) *max_ak = (f & 1) ? 0 : mak;
  return f & 1;
}

// One score step: compute + extend + post.  All threads.
template <int METRIC, bool LA>
__device__ __forceinline__ int wf_step(int ii, int s, bool act_on_end, int* max_ak) {
  wf_compute_extend<METRIC, LA>(ii, s, max_ak != nullptr);
  return wf_post_extend<LA>(ii, s, act_on_end, max_ak);
}

// wavefront_unialign.  All threads; returns status.
template <int METRIC, bool LA>
__device__ int wf_run(int ii) {
  wf_init<LA>(ii);
  if (sh.inst[ii].status == ST_OOM) return ST_OOM;
  wf_extend_only<LA>(ii, 0, false);
  if (wf_post_extend<LA>(ii, 0, true, nullptr)) return sh.inst[ii].status;
  for (int s = 1;; ++s)
    if (wf_step<METRIC, LA>(ii, s, true, nullptr)) return sh.inst[ii].status;
}

}  // namespace wfa
}  // namespace trgt
