// trgt_amd/csrc/wfa.hip -- batch wavefront alignment kernel (WFA / BiWFA) and trgt_wfa_batch.
//
// Replaces the per-alignment WFA2-lib calls behind src/wfaligner.rs of PacificBiosciences/trgt
// v3.0.0 (wavefront_align :492-498/:519-525, cigar_get_CIGAR :944-949, cigar_count_matches :999,
// get_alignment_span :864-908) with one launch per batch.  See wfa_engine.hpp for the wavefront
// machinery; this file holds the back-trace, the BiWFA driver (breakpoint search + recursion as
// an explicit stack), the per-job epilogue and the host-side planner.
#include <algorithm>
#include <chrono>

#include "wfa_engine.hpp"

namespace trgt {
namespace wfa {

// ------------------------------------------------------------------ back-trace (thread 0)
struct RleOut { uint32_t* buf; uint32_t cap; };

__device__ __forceinline__ uint32_t op_code(char op) { return op == 'M' ? 7u : op == 'X' ? 8u : op == 'I' ? 1u : 2u; }


template <bool LA>
__device__ __forceinline__ long long bt_cand(const Inst& I, int c, int s, int k, int add, int type) {
  if (s < 0 || s >= I.n_slots) return (long long)OFF_NULL;
  const WfDesc d = I.gdesc[(size_t)s * 5 + c];
  if (d.base == NOBASE || k < d.lo || k > d.hi) return (long long)OFF_NULL;
  return (((long long)(arena_of<LA>(I)[d.base + (uint32_t)(k - d.lo_alloc)] + add)) << 4) | type;  // BACKTRACE_TYPE_BITS_SET
}

// wavefront_backtrace_{linear,affine}: candidates encoded (offset << 4 | type), maximum wins.
// Emits the operations in reverse order, run-length encoded, into tmp[0..*ntmp).
template <bool LA>
__device__ __noinline__ void wf_backtrace(const Inst& I, uint32_t* tmp, int& ntmp, uint32_t cap) {
  const Pen& pen = KP.pen;
  const int plen = I.plen, tlen = I.tlen, metric = pen.metric;
  const int x = pen.x, o1 = pen.o1, e1 = pen.e1, o2 = pen.o2, e2 = pen.e2;
  const bool lin = metric <= M_LINEAR;
  int mt = I.ce, s = I.end_score, k = I.end_k, off = I.end_off;
  int h = off, v = off - k;
  ntmp = 0;
  if (I.ce == CM) {  // ending deletions / insertions of an ends-free alignment
    rle_push(tmp, ntmp, cap, 2u, plen - v);
    rle_push(tmp, ntmp, cap, 1u, tlen - h);
  }
  while (v > 0 && h > 0 && s > 0) {
    long long best = (long long)OFF_NULL;
    if (lin) {
      if (metric != M_INDEL) best = max(best, bt_cand<LA>(I, CM, s - x, k, +1, 9));
      best = max(best, bt_cand<LA>(I, CM, s - o1, k - 1, +1, 1));
      best = max(best, bt_cand<LA>(I, CM, s - o1, k + 1, 0, 5));
    } else {
      if (mt == CM) best = max(best, bt_cand<LA>(I, CM, s - x, k, +1, 9));
      if (mt == CM || mt == CD1) { best = max(best, bt_cand<LA>(I, CD1, s - e1, k + 1, 0, 6)); best = max(best, bt_cand<LA>(I, CM, s - o1 - e1, k + 1, 0, 5)); }
      if (mt == CM || mt == CI1) { best = max(best, bt_cand<LA>(I, CI1, s - e1, k - 1, +1, 2)); best = max(best, bt_cand<LA>(I, CM, s - o1 - e1, k - 1, +1, 1)); }
      if (metric == M_AFFINE2P) {
        if (mt == CM || mt == CD2) { best = max(best, bt_cand<LA>(I, CD2, s - e2, k + 1, 0, 8)); best = max(best, bt_cand<LA>(I, CM, s - o2 - e2, k + 1, 0, 7)); }
        if (mt == CM || mt == CI2) { best = max(best, bt_cand<LA>(I, CI2, s - e2, k - 1, +1, 4)); best = max(best, bt_cand<LA>(I, CM, s - o2 - e2, k - 1, +1, 3)); }
      }
    }
    if (best < 0) break;
    const int best_off = (int)(best >> 4), type = (int)(best & 0xF);
    if (mt == CM) {
      rle_push(tmp, ntmp, cap, 7u, off - best_off);
      off = best_off; h = off; v = off - k;
      if (v <= 0 || h <= 0) break;
    }
    switch (type) {
      case 9: rle_push(tmp, ntmp, cap, 8u, 1); s -= x; mt = CM; --off; break;
      case 1: rle_push(tmp, ntmp, cap, 1u, 1); s -= lin ? o1 : (o1 + e1); mt = CM; --k; --off; break;
      case 2: rle_push(tmp, ntmp, cap, 1u, 1); s -= e1; mt = CI1; --k; --off; break;
      case 3: rle_push(tmp, ntmp, cap, 1u, 1); s -= o2 + e2; mt = CM; --k; --off; break;
      case 4: rle_push(tmp, ntmp, cap, 1u, 1); s -= e2; mt = CI2; --k; --off; break;
      case 5: rle_push(tmp, ntmp, cap, 2u, 1); s -= lin ? o1 : (o1 + e1); mt = CM; ++k; break;
      case 6: rle_push(tmp, ntmp, cap, 2u, 1); s -= e1; mt = CD1; ++k; break;
      case 7: rle_push(tmp, ntmp, cap, 2u, 1); s -= o2 + e2; mt = CM; ++k; break;
      default: rle_push(tmp, ntmp, cap, 2u, 1); s -= e2; mt = CD2; ++k; break;
    }
    h = off; v = off - k;
  }
  if (mt == CM && v > 0 && h > 0) {
    const int n = min(v, h);
    rle_push(tmp, ntmp, cap, 7u, n);
    v -= n; h -= n;
  }
  rle_push(tmp, ntmp, cap, 2u, v);
  rle_push(tmp, ntmp, cap, 1u, h);
}

// append the reversed run list to the job's forward CIGAR (merging equal neighbours like cigar_get_CIGAR does)
__device__ __forceinline__ void rle_append_reversed(uint32_t* out, int& n, uint32_t cap, const uint32_t* tmp, int ntmp) {
  for (int i = ntmp - 1; i >= 0; --i) rle_push(out, n, cap, tmp[i] & 0xF, (int)(tmp[i] >> 4));
}

// ------------------------------------------------------------------ BiWFA
// wavefront_bialign_breakpoint_{indel2indel,m2m}.  All threads.
template <bool LA>
__device__ __noinline__ void bp_check(int i0, int i1, bool fwd, int s0, int s1, const WfDesc& w0, const WfDesc& w1, int comp,
                         int gap_open) {
  const Inst& A0 = sh.inst[i0];
  const Inst& A1 = sh.inst[i1];
  const int32_t* const R0 = arena_of<LA>(A0); const int32_t* const R1 = arena_of<LA>(A1);
  const int plen = A0.plen, tlen = A0.tlen, tid = threadIdx.x, T = blockDim.x;
  const int lo0 = w0.lo, hi0 = w0.hi, lo1 = (tlen - plen) - w1.hi, hi1 = (tlen - plen) - w1.lo;
  if (hi1 < lo0 || hi0 < lo1) return;
  if (!(s0 + s1 - gap_open < sh.bp.score)) return;
  const int min_hi = min(hi0, hi1), max_lo = max(lo0, lo1);
  __syncthreads();
  if (tid == 0) sh.red.bp_k = INT32_MAX;
  __syncthreads();
  for (int kb = max_lo; kb <= min_hi; kb += T) {
    const int k0 = kb + tid;
    bool ok = false;
    if (k0 <= min_hi) {
      const int k1 = (tlen - plen) - k0;
      const int32_t h0 = R0[w0.base + (uint32_t)(k0 - w0.lo_alloc)], h1 = R1[w1.base + (uint32_t)(k1 - w1.lo_alloc)];
      if (h0 + h1 >= tlen) {
        const int hh = fwd ? h0 : h1, kk = fwd ? k0 : k1;
        ok = !((hh - kk) > plen || hh > tlen);
      }
    }
    red_first(ok, k0, &sh.red.bp_k);
  }
  __syncthreads();
  if (tid == 0 && sh.red.bp_k != INT32_MAX) {
    const int k0 = sh.red.bp_k, k1 = (tlen - plen) - k0;
    const int32_t h0 = R0[w0.base + (uint32_t)(k0 - w0.lo_alloc)], h1 = R1[w1.base + (uint32_t)(k1 - w1.lo_alloc)];
    Breakpoint& bp = sh.bp;
    if (fwd) { bp.score_f = s0; bp.score_r = s1; bp.k_f = k0; bp.off_f = h0; }
    else { bp.score_f = s1; bp.score_r = s0; bp.k_f = k1; bp.off_f = h1; }
    bp.score = s0 + s1 - gap_open; bp.comp = comp;
  }
  __syncthreads();
}

// wavefront_bialign_overlap.  All threads.
template <bool LA>
__device__ __noinline__ void bi_overlap(int i0, int i1, int s0, int s1, bool fwd) {
  const Pen& pen = KP.pen;
  const WfDesc m0 = fetch_raw(i0, CM, s0);
  if (m0.base == NOBASE) return;
  WfDesc d10 = null_desc(), i10 = null_desc(), d20 = null_desc(), i20 = null_desc();
  if (pen.metric >= M_AFFINE) { d10 = fetch_raw(i0, CD1, s0); i10 = fetch_raw(i0, CI1, s0); }
  if (pen.metric == M_AFFINE2P) { d20 = fetch_raw(i0, CD2, s0); i20 = fetch_raw(i0, CI2, s0); }
  for (int i = 0; i < pen.scope; ++i) {
    const int si = s1 - i;
    if (si < 0) break;
    if (pen.metric == M_AFFINE2P && s0 + si - pen.o2 < sh.bp.score) {
      const WfDesc d21 = fetch_raw(i1, CD2, si);
      if (d20.base != NOBASE && d21.base != NOBASE) bp_check<LA>(i0, i1, fwd, s0, si, d20, d21, CD2, pen.o2);
      const WfDesc i21 = fetch_raw(i1, CI2, si);
      if (i20.base != NOBASE && i21.base != NOBASE) bp_check<LA>(i0, i1, fwd, s0, si, i20, i21, CI2, pen.o2);
    }
    if (pen.metric >= M_AFFINE && s0 + si - pen.o1 < sh.bp.score) {
      const WfDesc d11 = fetch_raw(i1, CD1, si);
      if (d10.base != NOBASE && d11.base != NOBASE) bp_check<LA>(i0, i1, fwd, s0, si, d10, d11, CD1, pen.o1);
      const WfDesc i11 = fetch_raw(i1, CI1, si);
      if (i10.base != NOBASE && i11.base != NOBASE) bp_check<LA>(i0, i1, fwd, s0, si, i10, i11, CI1, pen.o1);
    }
    if (s0 + si >= sh.bp.score) continue;
    const WfDesc m1 = fetch_raw(i1, CM, si);
    if (m1.base != NOBASE) bp_check<LA>(i0, i1, fwd, s0, si, m0, m1, CM, 0);
  }
}

struct BlockWs {  // carved from the workgroup's HBM workspace
  WfDesc* gdesc; int32_t* arena_u; int32_t* arena_f; int32_t* arena_r; uint32_t* rle_tmp; uint32_t* rle_out; uint32_t* run_start;
};

// P / T: sequences of the (sub-)alignment; p_lds / t_lds: their byte offsets in the dynamic LDS (LDS-arena variant)
template <bool LA>
__device__ __forceinline__ void setup_inst(int ii, const BlockWs& ws, const uint8_t* P, uint32_t p_lds, int pl,
                                           const uint8_t* T, uint32_t t_lds, int tl, int rev, int span, int pbf, int pef, int tbf, int tef,
                                           int cb, int ce) {
  Inst& I = sh.inst[ii];
  const KArgs& a = sh.args;
  I.pp = P; I.tp = T; I.pp_lds = p_lds; I.tp_lds = t_lds; I.plen = pl; I.tlen = tl; I.rev = rev;
  I.span = span; I.pbf = pbf; I.pef = pef; I.tbf = tbf; I.tef = tef; I.cb = cb; I.ce = ce;
  I.modular = ii != I_UNI;
  I.bump = 0;
  if constexpr (LA) {
    // One LDS region serves the two phases of an alignment, which never overlap: the forward / reverse rings of the breakpoint search
    // (half of it each: stride = what fits scope x ncomp wavefronts), or the descriptor history + the bump arena of a base alignment.
    const uint32_t half = (a.la_region_bytes / 2) & ~15u;
    const uint32_t per_dir = (uint32_t)(a.kp.pen.scope * a.kp.pen.ncomp);
    if (ii == I_UNI) {
      const uint32_t gd_bytes = a.la_gdesc_slots * 5u * (uint32_t)sizeof(WfDesc);
      I.gdesc = reinterpret_cast<WfDesc*>(lds_dyn + a.la_region); I.n_slots = (int)a.la_gdesc_slots;
      I.arena_lds = a.la_region + gd_bytes; I.arena_cap = (a.la_region_bytes - gd_bytes) / 4; I.stride = 0; I.arena = nullptr;
    } else {
      I.gdesc = nullptr; I.n_slots = INT32_MAX;
      I.arena_lds = a.la_region + (ii == I_FWD ? 0u : half); I.arena_cap = 0; I.stride = half / 4 / per_dir; I.arena = nullptr;
    }
  } else {
    I.gdesc = ws.gdesc; I.n_slots = ii == I_UNI ? (int)a.uni_slots : INT32_MAX;
    I.arena = ii == I_UNI ? ws.arena_u : (ii == I_FWD ? ws.arena_f : ws.arena_r);
    I.arena_cap = a.arena_uni_cap; I.stride = a.ring_stride; I.arena_lds = 0;
  }
}

// wavefront_bialign_find_breakpoint.  All threads.
template <int METRIC, bool LA>
__device__ __noinline__ int bi_find_breakpoint(const BlockWs& ws, const uint8_t* P, uint32_t p_lds, const uint8_t* T, uint32_t t_lds, const Seg seg) {
  const KParams& kp = KP;
  __syncthreads();
  if (threadIdx.x == 0) {
    setup_inst<LA>(I_FWD, ws, P + seg.pb, p_lds + (uint32_t)seg.pb, seg.pl, T + seg.tb, t_lds + (uint32_t)seg.tb, seg.tl, 0, 0, 0, 0, 0, 0, seg.cb, CM);
    setup_inst<LA>(I_REV, ws, P + seg.pb, p_lds + (uint32_t)seg.pb, seg.pl, T + seg.tb, t_lds + (uint32_t)seg.tb, seg.tl, 1, 0, 0, 0, 0, 0, seg.ce, CM);
    sh.bp.score = INT32_MAX;
  }
  __syncthreads();
  const int max_antidiagonal = seg.pl + seg.tl - 1;
  int sf = 0, sr = 0, fak = 0, rak = 0, mak = 0;
  wf_init<LA>(I_FWD);
  if (sh.inst[I_FWD].status == ST_OOM) return ST_OOM;
  wf_extend_only<LA>(I_FWD, 0, true);
  if (wf_post_extend<LA>(I_FWD, 0, true, &fak)) return sh.inst[I_FWD].status;
  wf_init<LA>(I_REV);
  if (sh.inst[I_REV].status == ST_OOM) return ST_OOM;
  wf_extend_only<LA>(I_REV, 0, true);
  if (wf_post_extend<LA>(I_REV, 0, true, &rak)) return sh.inst[I_REV].status;
  bool last_forward = false;
  while (true) {
    if (fak + rak >= max_antidiagonal) break;
    ++sf;
    if (wf_step<METRIC, LA>(I_FWD, sf, true, &mak)) return sh.inst[I_FWD].status;
    if (fak < mak) fak = mak;
    last_forward = true;
    if (fak + rak >= max_antidiagonal) break;
    ++sr;
    if (wf_step<METRIC, LA>(I_REV, sr, true, &mak)) return sh.inst[I_REV].status;
    if (rak < mak) rak = mak;
    last_forward = false;
  }
  const int scope = kp.pen.scope;
  const int gap_opening = METRIC == M_AFFINE ? kp.pen.o1 : (METRIC == M_AFFINE2P ? max(kp.pen.o1, kp.pen.o2) : 0);
  while (true) {
    if (last_forward) {
      const int min_sr = (sr > scope - 1) ? sr - (scope - 1) : 0;
      if (sf + min_sr - gap_opening >= sh.bp.score) break;
      bi_overlap<LA>(I_FWD, I_REV, sf, sr, true);
      ++sr;
      if (wf_step<METRIC, LA>(I_REV, sr, false, nullptr)) return sh.inst[I_REV].status;  // only a dead front ends phase 2
    }
    const int min_sf = (sf > scope - 1) ? sf - (scope - 1) : 0;
    if (min_sf + sr - gap_opening >= sh.bp.score) break;
    bi_overlap<LA>(I_REV, I_FWD, sr, sf, false);
    ++sf;
    if (wf_step<METRIC, LA>(I_FWD, sf, false, nullptr)) return sh.inst[I_FWD].status;
    last_forward = true;
  }
  return ST_OK;
}

__device__ __forceinline__ int classic_score(int metric, int s) { return metric <= M_EDIT ? s : -s; }

// wavefront_bialign_base.  All threads.  Returns false on failure (sh.status set).
template <int METRIC, bool LA>
__device__ __noinline__ bool bi_base(const BlockWs& ws, const uint8_t* P, uint32_t p_lds, const uint8_t* T, uint32_t t_lds, const Seg seg, bool want_cigar, uint32_t rle_cap) {
  __syncthreads();
  if (threadIdx.x == 0) setup_inst<LA>(I_UNI, ws, P + seg.pb, p_lds + (uint32_t)seg.pb, seg.pl, T + seg.tb, t_lds + (uint32_t)seg.tb, seg.tl, 0, 0, 0, 0, 0, 0, seg.cb, seg.ce);
  __syncthreads();
  const int st = wf_run<METRIC, LA>(I_UNI);
  if (threadIdx.x == 0) {
    if (st != ST_END_REACHED) sh.status = st == ST_OOM ? TRGT_WF_OOM : TRGT_WF_UNATTAINABLE;
    else if (want_cigar) {
      int nt = 0;
      wf_backtrace<LA>(sh.inst[I_UNI], ws.rle_tmp, nt, rle_cap);
      rle_append_reversed(ws.rle_out, sh.rle_n, rle_cap, ws.rle_tmp, nt);
      if (LA && ((uint32_t)nt >= rle_cap || (uint32_t)sh.rle_n >= rle_cap)) sh.status = TRGT_WF_OOM;
    }
  }
  __syncthreads();
  return st == ST_END_REACHED;
}

// ------------------------------------------------------------------ kernel
#ifdef TRGT_WFA_PROF
__device__ unsigned long long g_wfa_gprof[8];  // generic kernel, thread 0: claim + staging, breakpoint searches, base alignments, epilogue, jobs
#define GP_DECL unsigned long long gp_t = clock64()
#define GP_MARK(i) do { const unsigned long long n_ = clock64(); if (threadIdx.x == 0) atomicAdd(&g_wfa_gprof[i], n_ - gp_t); gp_t = n_; } while (0)
#else
#define GP_DECL
#define GP_MARK(i)
#endif
// LA = false: wavefronts in the workgroup's HBM workspace (any size).  LA = true (the LDS-arena variant, one wave per alignment):
// sequences, run-length buffers and wavefronts of the current alignment live in the dynamic LDS; an alignment that does not fit -- a
// wavefront wider than the ring stride, a base alignment whose history outgrows the arena or its descriptor slots, more runs than
// the buffers hold, sequences beyond the staging area -- is appended to the retry list and redone by the HBM variant (launched right
// behind over that list): the same code on the same inputs, so the results do not depend on where an alignment ran.
template <int METRIC, bool LA>
__device__ __forceinline__ void wfa_kernel_body(const KArgs& a);
template <int METRIC, bool LA>
__global__ void __launch_bounds__(256, 3) wfa_kernel(const KArgs a) { wfa_kernel_body<METRIC, LA>(a); }
// One wave per alignment (consensus alignments, edit distances: every batch of the locus path but the flank location): the same body
// compiled for 128 VGPRs, i.e. four waves per SIMD instead of three.  The kernel waits on dependent memory round trips most of the
// time: resident waves are what its throughput is made of.
template <int METRIC>
__global__ void __launch_bounds__(64, 4) wfa_kernel_wave(const KArgs a) { wfa_kernel_body<METRIC, false>(a); }
template <int METRIC>
__global__ void __launch_bounds__(64, 4) wfa_kernel_wave_la(const KArgs a) { wfa_kernel_body<METRIC, true>(a); }
template <int METRIC, bool LA>
__device__ __forceinline__ void wfa_kernel_body(const KArgs& a) {
  const int tid = threadIdx.x, T = blockDim.x;
  {  // the argument block into LDS (what the engine functions read) and the place of the descriptor rings in the dynamic LDS
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&a); uint32_t* dst = reinterpret_cast<uint32_t*>(&sh.args);
    for (uint32_t i = tid; i < sizeof(KArgs) / 4; i += T) dst[i] = src[i];
    if (tid == 0) { sh.ring_off = a.ring_off; sh.ring_mask = a.ring_mask; }
  }
  __syncthreads();
  const KParams& kp = KP;
  // The HBM workspace belongs to a *slot* acquired for the lifetime of the workgroup (not to blockIdx), which keeps the
  // door open for grids larger than the slot count; today the grid equals the slot count and workgroups are persistent.
  uint32_t ws_slot = 0;
  BlockWs ws;
  if constexpr (!LA) {
    if (tid == 0) {
      uint32_t i = (blockIdx.x * 2654435761u) % a.n_slots_ws;
      while (atomicCAS(&a.slot_flags[i], 0u, 1u) != 0u) i = i + 1 == a.n_slots_ws ? 0 : i + 1;
      sh.top_bp = (int)i;
    }
    __syncthreads();
    ws_slot = (uint32_t)sh.top_bp;
    __syncthreads();
    uint8_t* base = a.ws + (size_t)ws_slot * a.ws_per_block;
    ws.gdesc = reinterpret_cast<WfDesc*>(base + a.off_gdesc);
    ws.arena_u = reinterpret_cast<int32_t*>(base + a.off_arena_u);
    ws.arena_f = reinterpret_cast<int32_t*>(base + a.off_arena_f);
    ws.arena_r = reinterpret_cast<int32_t*>(base + a.off_arena_r);
    ws.rle_tmp = reinterpret_cast<uint32_t*>(base + a.off_rle_tmp);
    ws.rle_out = reinterpret_cast<uint32_t*>(base + a.off_rle_out);
    ws.run_start = reinterpret_cast<uint32_t*>(base + a.off_run_start);
  } else {
    ws.gdesc = nullptr; ws.arena_u = ws.arena_f = ws.arena_r = nullptr; ws.run_start = nullptr;
    ws.rle_tmp = reinterpret_cast<uint32_t*>(lds_dyn + a.la_rle_tmp);
    ws.rle_out = reinterpret_cast<uint32_t*>(lds_dyn + a.la_rle_out);
  }
  const uint32_t rle_cap = LA ? a.la_rle_cap : a.rle_cap;
  const uint32_t n_front = a.n_jobs_dev ? *a.n_jobs_dev : a.n_jobs;
  const uint32_t n_jobs = n_front + (a.n_jobs2_dev ? *a.n_jobs2_dev : 0u);
  unsigned long long cells_acc = 0;
  GP_DECL;
  for (uint32_t jb = 0; jb < a.jobs_per_block; ++jb) {
    __syncthreads();
    GP_MARK(3);
    if (tid == 0) sh.job = (int)atomicAdd(a.counter, 1u);
    __syncthreads();
    const uint32_t j = (uint32_t)sh.job;
    if (j >= n_jobs) break;
    const JobDev job = a.jobs[j < n_front ? j : a.jobs_cap - 1u - (j - n_front)];
    const int plen = (int)job.pat_len, tlen = (int)job.txt_len;
    const uint8_t* P = a.pat_base + job.pat_off;
    const uint8_t* Tx = a.txt_base + job.txt_off;
    // The workspace is planned from the batch maxima the caller states (rings of ring_stride = max_sum + 4 offsets): a job beyond them
    // would write past its workgroup's slice.  Refuse it loudly (TRGT_WF_OOM, outputs cleared) instead of trusting device-built lists.
    if ((uint32_t)plen + (uint32_t)tlen + 4u > a.ring_stride) {
      if (tid == 0) {
        const uint32_t o = job.out_index;
        if (a.refused) atomicAdd(a.refused, 1u);
        if (a.status) a.status[o] = TRGT_WF_OOM;
        if (a.score) a.score[o] = INT32_MIN;
        if (a.n_match) a.n_match[o] = 0;
        if (a.cigar_len) a.cigar_len[o] = 0;
        if (a.ops_len) a.ops_len[o] = 0;
      }
      continue;
    }
    // ---- Identical sequences (end-to-end): a read of an allele against the central read / the consensus of its cluster, most of the
    //      time.  The result needs no alignment: one run of matches, penalty 0 -- what the code below returns for such a pair after a
    //      forward extension over the whole length, a base alignment and its back-trace (BiWFA: the breakpoint search ends with "end
    //      reached" at score 0 and hands over to the base alignment, so the top-level score is never set; score-only: 0).  Checked
    //      eight bytes per thread and step, straight from global memory, before anything is staged.
    if (kp.span == 0 && plen == tlen && plen > 0 && !LA) {
      uint64_t diff = 0;
      int i = 8 * tid;
      for (; i + 8 <= plen; i += 8 * T) { uint64_t x, y; __builtin_memcpy(&x, P + i, 8); __builtin_memcpy(&y, Tx + i, 8); diff |= x ^ y; }
      if (i < plen && i + 8 > plen) for (int b = i; b < plen; ++b) diff |= (uint64_t)(P[b] ^ Tx[b]);
      if (!__syncthreads_or(diff != 0ull)) {
        if (tid == 0) {
          const uint32_t o = job.out_index;
          const bool aln = kp.scope_alignment != 0;
          if (a.status) a.status[o] = TRGT_WF_COMPLETED;
          if (a.score) a.score[o] = kp.biwfa && aln ? INT32_MIN : 0;
          if (a.n_match) a.n_match[o] = aln ? plen : 0;
          if (a.span4) { a.span4[4 * o + 0] = 0; a.span4[4 * o + 1] = (uint32_t)plen; a.span4[4 * o + 2] = 0; a.span4[4 * o + 3] = (uint32_t)tlen; }
          if (a.cigar_len) a.cigar_len[o] = aln ? 1u : 0u;
          if (a.ops_len) a.ops_len[o] = aln ? (uint32_t)plen : 0u;
          if (aln && a.cigar) a.cigar[job.cigar_off] = ((uint32_t)plen << 4) | 7u;
          cells_acc += kp.biwfa ? 2ull : 1ull;  // (the level-0 cells of the runs this stands for)
        }
        if (kp.scope_alignment && a.ops) for (int p = tid; p < plen; p += T) a.ops[job.ops_off + p] = 'M';
        continue;
      }
    }
    // stage the two sequences in LDS when they fit (extension = byte compares against LDS)
    const uint32_t pl_pad = ((uint32_t)plen + 15u) & ~15u;
    const bool staged = pl_pad + (uint32_t)tlen <= a.lds_seq_cap;
    if (LA && !staged) {  // too long for this variant: to the HBM one
      if (tid == 0) a.retry_jobs[atomicAdd(a.retry_count, 1u)] = job;
      continue;
    }
    if (staged) {
      for (int i = tid; i < plen; i += T) lds_dyn[i] = P[i];
      for (int i = tid; i < tlen; i += T) lds_dyn[pl_pad + i] = Tx[i];
      P = lds_dyn; Tx = lds_dyn + pl_pad;
    }
    const uint32_t p_lds = 0, t_lds = pl_pad;
    if (tid == 0) { sh.status = TRGT_WF_COMPLETED; sh.score = INT32_MIN; sh.rle_n = 0; sh.sp = 0; sh.cells = 0; sh.top_bp = 0; }
    __syncthreads();
    GP_MARK(0);
    if (!kp.biwfa) {
      // ---- unidirectional (wavefront_unialign): MemoryHigh / Med / Low
      if (tid == 0) {
        const int sp = kp.span;
        auto fr = [](int v, int len) { return v < 0 ? len : v; };
        setup_inst<LA>(I_UNI, ws, P, p_lds, plen, Tx, t_lds, tlen, 0, sp, sp ? fr(kp.pbf, plen) : 0, sp ? fr(kp.pef, plen) : 0,
                       sp ? fr(kp.tbf, tlen) : 0, sp ? fr(kp.tef, tlen) : 0, CM, CM);
      }
      __syncthreads();
      const int st = wf_run<METRIC, LA>(I_UNI);
      if (tid == 0) {
        if (st != ST_END_REACHED) sh.status = st == ST_OOM ? TRGT_WF_OOM : TRGT_WF_UNATTAINABLE;
        else {
          sh.score = classic_score(METRIC, sh.inst[I_UNI].end_score);
          if (kp.scope_alignment) {
            int nt = 0;
            wf_backtrace<LA>(sh.inst[I_UNI], ws.rle_tmp, nt, rle_cap);
            rle_append_reversed(ws.rle_out, sh.rle_n, rle_cap, ws.rle_tmp, nt);
            if (LA && ((uint32_t)nt >= rle_cap || (uint32_t)sh.rle_n >= rle_cap)) sh.status = TRGT_WF_OOM;
          }
        }
      }
    } else if (!kp.scope_alignment) {
      // ---- BiWFA score only (wavefront_bialign_compute_score)
      Seg seg; seg.pb = 0; seg.pl = plen; seg.tb = 0; seg.tl = tlen; seg.cb = CM; seg.ce = CM; seg.rem = INT32_MAX; seg.top = 1;
      const int st = bi_find_breakpoint<METRIC, LA>(ws, P, p_lds, Tx, t_lds, seg);
      if (st == ST_END_REACHED) {
        if (bi_base<METRIC, LA>(ws, P, p_lds, Tx, t_lds, seg, false, rle_cap) && tid == 0) sh.score = classic_score(METRIC, sh.inst[I_UNI].end_score);
      } else if (tid == 0) {
        if (st != ST_OK) sh.status = st == ST_OOM ? TRGT_WF_OOM : TRGT_WF_UNATTAINABLE;
        else sh.score = classic_score(METRIC, sh.bp.score);
      }
    } else {
      // ---- BiWFA alignment (wavefront_bialign_alignment), recursion as an explicit stack, left half first
      if (tid == 0) {
        Seg s0; s0.pb = 0; s0.pl = plen; s0.tb = 0; s0.tl = tlen; s0.cb = CM; s0.ce = CM;
        s0.rem = max(plen, tlen) <= kp.bi_min_length ? 0 : INT32_MAX; s0.top = 1;
        sh.stack[0] = s0; sh.sp = 1;
      }
      __syncthreads();
      while (true) {
        __syncthreads();
        if (sh.sp == 0 || sh.status != TRGT_WF_COMPLETED) break;
        const Seg seg = sh.stack[sh.sp - 1];
        __syncthreads();
        if (tid == 0) sh.sp -= 1;
        __syncthreads();
        if (seg.tl == 0) { if (tid == 0) rle_push(ws.rle_out, sh.rle_n, rle_cap, 2u, seg.pl); continue; }
        if (seg.pl == 0) { if (tid == 0) rle_push(ws.rle_out, sh.rle_n, rle_cap, 1u, seg.tl); continue; }
        GP_MARK(3);
        if (seg.rem <= kp.bi_min_score) { bi_base<METRIC, LA>(ws, P, p_lds, Tx, t_lds, seg, true, rle_cap); GP_MARK(2); continue; }
        const int st = bi_find_breakpoint<METRIC, LA>(ws, P, p_lds, Tx, t_lds, seg);
        GP_MARK(1);
        if (st == ST_END_REACHED) { bi_base<METRIC, LA>(ws, P, p_lds, Tx, t_lds, seg, true, rle_cap); GP_MARK(2); continue; }
        if (st != ST_OK) { if (tid == 0) sh.status = st == ST_OOM ? TRGT_WF_OOM : TRGT_WF_UNATTAINABLE; continue; }
        if (tid == 0) {
          const Breakpoint bp = sh.bp;
          const int bh = bp.off_f, bv = bp.off_f - bp.k_f;
          if (seg.top) { sh.top_bp = 1; sh.score = classic_score(METRIC, bp.score); }
          if (sh.sp + 2 > BI_STACK) sh.status = TRGT_WF_OOM;
          else {
            Seg r; r.pb = seg.pb + bv; r.pl = seg.pl - bv; r.tb = seg.tb + bh; r.tl = seg.tl - bh; r.cb = bp.comp; r.ce = seg.ce; r.rem = bp.score_r; r.top = 0;
            Seg l; l.pb = seg.pb; l.pl = bv; l.tb = seg.tb; l.tl = bh; l.cb = seg.cb; l.ce = bp.comp; l.rem = bp.score_f; l.top = 0;
            sh.stack[sh.sp++] = r; sh.stack[sh.sp++] = l;
          }
        }
      }
      if (LA && tid == 0 && (uint32_t)sh.rle_n >= rle_cap) sh.status = TRGT_WF_OOM;
    }
    __syncthreads();
    GP_MARK(3);
    if (LA && sh.status == TRGT_WF_OOM) {  // did not fit the LDS budget somewhere: the HBM variant redoes it from scratch
      if (tid == 0) a.retry_jobs[atomicAdd(a.retry_count, 1u)] = job;  // (its offsets are counted where it completes)
      continue;
    }
    // ---- per-job epilogue: status, score, count_matches, alignment span, CIGAR, expanded operations
    const int ok = sh.status == TRGT_WF_COMPLETED;
    const int nrun = ok ? sh.rle_n : 0;
    if (tid == 0) {
      const uint32_t o = job.out_index;
      if (a.status) a.status[o] = sh.status;
      if (a.score) a.score[o] = ok ? sh.score : INT32_MIN;
      uint32_t pi = 0, ti = 0, ps = 0, pe = 0, ts = 0, te = 0, nm = 0, total = 0;
      bool started = false;
      for (int r = 0; r < nrun; ++r) {
        const uint32_t e = ws.rle_out[r], len = e >> 4, code = e & 0xF;
        if (!LA) ws.run_start[r] = total;
        total += len;
        if (code == 1u) ti += len;
        else if (code == 2u) pi += len;
        else { if (!started) { ps = pi; ts = ti; started = true; } pi += len; ti += len; pe = pi; te = ti; if (code == 7u) nm += len; }
      }
      if (kp.span == 0) { ps = 0; pe = (uint32_t)plen; ts = 0; te = (uint32_t)tlen; }
      if (a.n_match) a.n_match[o] = (int32_t)nm;
      if (a.span4) { a.span4[4 * o + 0] = ps; a.span4[4 * o + 1] = pe; a.span4[4 * o + 2] = ts; a.span4[4 * o + 3] = te; }
      if (a.cigar_len) a.cigar_len[o] = (uint32_t)nrun;
      if (a.ops_len) a.ops_len[o] = total;
      sh.top_bp = (int)total;
      cells_acc += sh.cells;
    }
    __syncthreads();
    if (a.cigar) for (int r = tid; r < nrun; r += T) a.cigar[job.cigar_off + r] = ws.rle_out[r];
    if (!LA && a.ops && nrun > 0) {
      const uint32_t total = (uint32_t)sh.top_bp;
      for (uint32_t p = tid; p < total; p += T) {
        int lo = 0, hi = nrun - 1;  // last run whose start <= p
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (ws.run_start[mid] <= p) lo = mid; else hi = mid - 1; }
        const uint32_t code = ws.rle_out[lo] & 0xF;
        a.ops[job.ops_off + p] = code == 7u ? 'M' : code == 8u ? 'X' : code == 1u ? 'I' : 'D';
      }
    }
  }
  if (tid == 0 && a.cells_out && cells_acc) atomicAdd(a.cells_out, cells_acc);
  if constexpr (!LA) {
    __syncthreads();
    if (tid == 0) { __threadfence(); atomicExch(&a.slot_flags[ws_slot], 0u); }
  }
}

}  // namespace wfa
}  // namespace trgt

#include "wfa_fast.hpp"  // the dedicated LDS-resident kernel for exact unidirectional gap-affine batches

namespace trgt {

// ------------------------------------------------------------------ host planner
static int gap_cost(const trgt_wfa_params& p, int64_t len) {
  if (len <= 0) return 0;
  int64_t c;
  switch (p.metric) {
    case 0: case 1: c = len; break;
    case 2: c = (int64_t)p.gap_ext1 * len; break;
    case 3: c = p.gap_open1 + (int64_t)p.gap_ext1 * len; break;
    default: c = std::min<int64_t>(p.gap_open1 + (int64_t)p.gap_ext1 * len, p.gap_open2 + (int64_t)p.gap_ext2 * len); break;
  }
  return (int)std::min<int64_t>(c, 1 << 28);
}

int wfa_launch(trgt_hip_ctx* c, const trgt_wfa_params& p, const WfaLaunch& L) {
  using namespace wfa;
  if (p.metric < 0 || p.metric > 4) return fail(c, TRGT_ERR_INVALID, "wfa: bad metric %d", p.metric);
  if (p.heuristic != 0 && p.heuristic != 1) return fail(c, TRGT_ERR_UNSUPPORTED, "wfa: only Heuristic::None and WFadaptive are implemented");
  if (p.memory_mode == 3 && p.span != 0) return fail(c, TRGT_ERR_UNSUPPORTED, "wfa: BiWFA is end-to-end only (as used by TRGT)");
  if (p.metric >= 2 && (p.mismatch <= 0 || p.gap_ext1 <= 0 || (p.metric >= 3 && p.gap_open1 < 0)))
    return fail(c, TRGT_ERR_INVALID, "wfa: penalties must be positive");
  KArgs a;
  std::memset(&a, 0, sizeof a);
  Pen& pen = a.kp.pen;
  pen.metric = p.metric;
  switch (p.metric) {
    case 0: pen.x = -1; pen.o1 = 1; pen.e1 = -1; pen.o2 = pen.e2 = -1; pen.scope = 2; pen.ncomp = 1; break;
    case 1: pen.x = 1; pen.o1 = 1; pen.e1 = -1; pen.o2 = pen.e2 = -1; pen.scope = 2; pen.ncomp = 1; break;
    case 2: pen.x = p.mismatch; pen.o1 = p.gap_ext1; pen.e1 = -1; pen.o2 = pen.e2 = -1; pen.scope = std::max(pen.x, pen.o1) + 1; pen.ncomp = 1; break;
    case 3: pen.x = p.mismatch; pen.o1 = p.gap_open1; pen.e1 = p.gap_ext1; pen.o2 = pen.e2 = -1; pen.scope = std::max(pen.x, pen.o1 + pen.e1) + 1; pen.ncomp = 3; break;
    default: pen.x = p.mismatch; pen.o1 = p.gap_open1; pen.e1 = p.gap_ext1; pen.o2 = p.gap_open2; pen.e2 = p.gap_ext2;
      pen.scope = std::max(pen.x, std::max(pen.o1 + pen.e1, pen.o2 + pen.e2)) + 1; pen.ncomp = 5; break;
  }
  if (pen.scope > RING) return fail(c, TRGT_ERR_UNSUPPORTED, "wfa: penalties need a score scope of %d (> %d)", pen.scope, RING);
  a.kp.span = p.span; a.kp.pbf = p.pattern_begin_free; a.kp.pef = p.pattern_end_free; a.kp.tbf = p.text_begin_free; a.kp.tef = p.text_end_free;
  a.kp.scope_alignment = p.scope != 0; a.kp.biwfa = p.memory_mode == 3; a.kp.heuristic = p.heuristic;
  a.kp.h_min_len = p.h_min_wavefront_length; a.kp.h_max_dist = p.h_max_distance_threshold; a.kp.h_steps = p.h_steps_between_cutoffs;
  a.kp.bi_min_score = p.bialign_min_score; a.kp.bi_min_length = p.bialign_min_length;
  // ---- workspace plan from the batch maxima
  const int64_t mp = L.max_plen, mt = L.max_tlen, msum = L.max_sum;
  int64_t score_bound;
  const bool text_free = p.span == 1 && (p.text_begin_free < 0) && (p.text_end_free < 0);
  if (text_free) score_bound = gap_cost(p, mp) + 2;          // delete the whole pattern anywhere in the free text
  else score_bound = (int64_t)gap_cost(p, mp) + gap_cost(p, mt) + 2;
  if (a.kp.biwfa && a.kp.scope_alignment) {                   // base cases: score <= 250 unless the heuristic misleads; min-length fallback
    const int64_t small = (int64_t)gap_cost(p, std::min<int64_t>(mp, p.bialign_min_length)) + gap_cost(p, std::min<int64_t>(mt, p.bialign_min_length)) + 2;
    score_bound = std::min<int64_t>(score_bound, std::max<int64_t>(small, 4 * (int64_t)p.bialign_min_score + 64));
  }
  score_bound = std::min<int64_t>(score_bound, 1 << 20);
  const bool capped = L.max_score > 0 && p.span == 1 && p.pattern_begin_free == 0 && p.text_begin_free >= 0;
  if (capped) score_bound = std::min<int64_t>(score_bound, L.max_score + 1);
  const uint64_t per_level = (uint64_t)pen.ncomp * (uint64_t)std::min<int64_t>(msum + 3, 2 * score_bound + 3 + (p.span ? msum : 0));
  uint64_t arena_ints = std::min<uint64_t>((uint64_t)(score_bound + 1) * per_level + 64, (1ull << 30) / 4);  // <= 1 GiB per workgroup
  const uint64_t ring_stride = (uint64_t)msum + 4;
  const uint64_t ring_ints = a.kp.biwfa ? (uint64_t)pen.scope * pen.ncomp * ring_stride : 0;
  auto al = [](uint64_t v) { return (v + 255) & ~255ull; };
  uint64_t off = 0;
  a.uni_slots = (uint32_t)(score_bound + 1);
  a.off_gdesc = off; off = al(off + (uint64_t)a.uni_slots * 5 * sizeof(WfDesc));
  a.off_arena_u = off; off = al(off + arena_ints * 4);
  a.off_arena_f = off; off = al(off + ring_ints * 4);
  a.off_arena_r = off; off = al(off + ring_ints * 4);
  a.rle_cap = (uint32_t)(msum + 4);
  a.off_rle_tmp = off; off = al(off + (uint64_t)a.rle_cap * 4);
  a.off_rle_out = off; off = al(off + (uint64_t)a.rle_cap * 4);
  a.off_run_start = off; off = al(off + (uint64_t)a.rle_cap * 4);
  a.ws_per_block = off;
  a.arena_uni_cap = (uint32_t)std::min<uint64_t>(arena_ints, 0xFFFFFFF0ull);
  a.ring_stride = (uint32_t)ring_stride;
  const int threads = L.threads > 0 ? L.threads : (p.span == 1 ? 256 : 64);
  int64_t blocks = std::min<int64_t>(L.n_jobs_host, (int64_t)c->num_cus * (threads >= 512 ? 4 : threads >= 256 ? 8 : 16));
  blocks = std::min<int64_t>(blocks, (int64_t)(c->ws_limit / std::max<uint64_t>(a.ws_per_block, 1)));
  if (L.ws_budget > 0 && blocks > 1) blocks = std::max<int64_t>(1, std::min<int64_t>(blocks, (int64_t)(L.ws_budget / std::max<uint64_t>(a.ws_per_block, 1))));
  if (blocks < 1) {
    // shrink the per-workgroup arena to what the limit allows; overflowing jobs report TRGT_WF_OOM
    const uint64_t fixed = a.ws_per_block - al(arena_ints * 4);
    if (c->ws_limit <= fixed + 4096) return fail(c, TRGT_ERR_NOMEM, "wfa: workspace limit %llu B too small", (unsigned long long)c->ws_limit);
    arena_ints = (c->ws_limit - fixed - 4096) / 4;
    return fail(c, TRGT_ERR_NOMEM, "wfa: one workgroup needs %llu B of workspace, limit is %llu B", (unsigned long long)a.ws_per_block,
                (unsigned long long)c->ws_limit);
  }
  void* d_ws = nullptr; void* d_counter = nullptr; void* d_cells = nullptr;
  int rc;
  if ((rc = dev_get(c, L.buffer_set == 2 ? S_WFA_WS_C : L.buffer_set ? S_WFA_WS_B : S_WFA_WS, (size_t)(a.ws_per_block * (uint64_t)blocks), &d_ws))) return rc;
  // job counters | slot flags | 24 words of the lean kernels (statistics, their counters); the offset counter of the batch.  Cleared
  // pieces of the call's zero arena (inside trgt_locus_batch: no hipMemsetAsync per launch), else pool slots cleared here
  const size_t counter_bytes = 16 + 4 * (size_t)blocks + 160;  // (+ 40 words of the lean kernels: 3 x 8 statistics, 5 counters)
  const int bset = L.buffer_set == 2 ? 2 : L.buffer_set ? 1 : 0;
  if ((rc = dev_get_zeroed(c, bset == 2 ? S_WFA_COUNTER_C : bset ? S_WFA_COUNTER_B : S_WFA_COUNTER, counter_bytes, &d_counter, c->stream))) return rc;
  // (buffer set 2 = the device-side chains of a locus call -- consensus repair, cluster genotyper: their launches count into one piece,
  //  read back once per call for the roofline block of the consensus alignments)
  const bool chain_keep = bset == 2 && c->zero_on && c->wfa_cells_cur[2] != nullptr;
  if (!L.keep_cells && !chain_keep) {
    if ((rc = dev_get_zeroed(c, bset == 2 ? S_WFA_CELLS_C : bset ? S_WFA_CELLS_B : S_WFA_CELLS, 32, &d_cells, c->stream))) return rc;
    c->wfa_cells_cur[bset] = d_cells;
  } else if (c->wfa_cells_cur[bset]) d_cells = c->wfa_cells_cur[bset];  // a later launch of the same logical batch: keeps counting where the first one did
  else if ((rc = dev_get(c, bset == 2 ? S_WFA_CELLS_C : bset ? S_WFA_CELLS_B : S_WFA_CELLS, 32, &d_cells))) return rc;
  a.slot_flags = (unsigned int*)d_counter + 4; a.n_slots_ws = (uint32_t)blocks; a.jobs_per_block = 0xFFFFFFFFu;  // persistent workgroups: measured 15-45 % faster than short-lived ones
  a.ws = (uint8_t*)d_ws; a.counter = (unsigned int*)d_counter; a.cells_out = (unsigned long long*)d_cells; a.refused = L.refused;
  a.jobs = L.jobs_dev; a.n_jobs = (uint32_t)L.n_jobs_host; a.n_jobs_dev = L.n_jobs_dev;
  a.n_jobs2_dev = L.n_jobs2_dev; a.jobs_cap = L.jobs_cap;
  a.pat_base = L.pat_base; a.txt_base = L.txt_base;
  a.status = L.status; a.score = L.score; a.n_match = L.n_match; a.span4 = L.span4; a.cigar = L.cigar; a.cigar_len = L.cigar_len;
  a.ops = L.ops; a.ops_len = L.ops_len;
  a.retry_jobs = nullptr; a.retry_count = nullptr;
  const uint64_t seq_need = ((uint64_t)mp + 15) / 16 * 16 + (uint64_t)mt + 16;
  a.lds_seq_cap = (uint32_t)((std::min<uint64_t>(seq_need, 32 * 1024) + 15) & ~15ull);
  if (seq_need > 32 * 1024 || c->knobs.wfa_no_stage) a.lds_seq_cap = 0;  // too long: extend straight from global memory (L1/L2 cached)
  size_t lds = a.lds_seq_cap;
  a.fast_wcap = 0; a.fast_ring_bytes = 0; a.fast_koff = 0;
  a.fast_dbg = c->knobs.skip_bt ? 1 : 0;
  if (p.metric == 3 && p.heuristic == 0 && !a.kp.biwfa && a.lds_seq_cap > 0 && mt + score_bound < 65000 && threads % 64 == 0 && threads <= 256) {
    // LDS fast path (wfa_fast.hpp): ring of the live wavefronts as 16-bit offsets
    uint64_t wcap = ((uint64_t)msum + 8 + 7) & ~7ull;
    if (capped) {  // levels 0 .. max_score + 1: diagonals -(max_score + 1) .. text_begin_free + max_score + 1
      a.fast_koff = (uint32_t)(L.max_score + 4);
      wcap = std::min<uint64_t>(wcap, ((uint64_t)a.fast_koff + (uint64_t)p.text_begin_free + (uint64_t)L.max_score + 12 + 7) & ~7ull);
    }
    const uint64_t ring_bytes = (uint64_t)(std::max(pen.x, pen.o1 + pen.e1) + 1 + 2 * (pen.e1 + 1)) * wcap * 2;
    const uint64_t win_bytes = 4ull * (uint64_t)(mp + mt + 8);
    // (the byte copies of the two sequences are staged in the ring area, which is idle until level 0 is written)
    if (ring_bytes + win_bytes <= 96 * 1024 && (seq_need <= ring_bytes || capped)) { a.fast_wcap = (uint32_t)wcap; a.fast_ring_bytes = (uint32_t)((ring_bytes + 15) & ~15ull); lds = (size_t)a.fast_ring_bytes + (size_t)win_bytes; }
  }
  // TRGT's flank-location configuration has an instantiation of its own (wfa_fast.hpp, SPEC)
  const int tag = L.kernel_tag >= 0 ? L.kernel_tag : (L.timer_slot == TRGT_K_WFA_FLANK_REST ? 1 : 0);
  const bool flank_pen = pen.x == 2 && pen.o1 == 5 && pen.e1 == 1 && a.kp.span == 1 && a.kp.pbf == 0 && a.kp.pef == 0 && a.kp.tef < 0 && !c->knobs.no_spec;
  const bool fast_spec = flank_pen && a.fast_koff == 0 && a.kp.tbf < 0 && (threads == 256 || threads == 192);
  const bool win_spec = flank_pen && a.fast_koff != 0 && (tag == 2 || tag == 3) && threads == 64;  // the windowed launches of trgt_find_spans_batch
  // (TRGT_BAND_THREADS=256 / 128: the banded back-trace of what the pre-filter keeps with four / two waves per alignment -- its launch is as
  //  long as its slowest alignment, and a band of 2 s* + 2 s + 1 diagonals is several strips of a wave)
  const bool band_wide = flank_pen && a.fast_koff != 0 && tag == 3 && (threads == 256 || threads == 128);
  if (tag == 3 && !win_spec && !band_wide) return fail(c, TRGT_ERR_INVALID, "wfa: the banded launch exists for the flank configuration only");
  void (*const spec_fn[2][3])(const KArgs) = {{wfa_fast_kernel<256, 0>, wfa_fast_kernel<256, 1>, wfa_fast_kernel<256, 2>},
                                              {wfa_fast_kernel<192, 0>, wfa_fast_kernel<192, 1>, wfa_fast_kernel<192, 2>}};
  void (*const gen_fn[3])(const KArgs) = {wfa_fast_kernel<0, 0>, wfa_fast_kernel<0, 1>, wfa_fast_kernel<0, 2>};
  void (*const fast_fn)(const KArgs) = band_wide ? (threads == 256 ? wfa_fast_kernel<256, 3> : wfa_fast_kernel<128, 3>) : win_spec ? (tag == 3 ? wfa_fast_kernel<64, 3> : wfa_fast_kernel<64, 2>) : fast_spec ? spec_fn[threads == 192 ? 1 : 0][tag] : gen_fn[tag];
  KTimer t(c, L.timer_slot);
  // `blocks` bounds how many workgroups can be resident (one workspace slot each); the grid covers all jobs
  int64_t grid_blocks = std::max<int64_t>(1, std::min<int64_t>(blocks, L.n_jobs_host));
  if (a.fast_wcap > 0) {
    // Persistent workgroups: launch exactly as many as can be resident (LDS- and register-bound, 4 per CU at most).  Workgroups
    // waiting for dispatch would do no work anyway, and while they wait the dispatcher keeps kernels of other streams (the
    // gather / HMM kernels of an earlier chunk) from starting.
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fast_fn, threads, lds) != hipSuccess || occ < 1) { (void)hipGetLastError(); occ = 4; }
    int64_t per_cu = occ;
    if (c->knobs.grid_per_cu > 0) per_cu = c->knobs.grid_per_cu;
    if (c->knobs.debug) fprintf(stderr, "[wfa] fast kernel lds=%zu occupancy=%d per_cu=%lld threads=%d\n", lds, occ, (long long)per_cu, threads);
    grid_blocks = std::min<int64_t>(grid_blocks, (int64_t)c->num_cus * per_cu);
  }
  const dim3 grid((unsigned)grid_blocks), block((unsigned)threads);
  if (a.fast_wcap == 0) {  // the generic kernel keeps the descriptor rings of its three instances behind the staged sequences
    a.ring_mask = RING - 1; a.ring_off = (a.lds_seq_cap + 15u) & ~15u;
    lds = (size_t)a.ring_off + 3 * RING * 5 * sizeof(WfDesc);
    if (lds > 48 * 1024)
      for (const void* fn : {(const void*)wfa_kernel<0, false>, (const void*)wfa_kernel<1, false>, (const void*)wfa_kernel<2, false>, (const void*)wfa_kernel<3, false>, (const void*)wfa_kernel<4, false>,
                             (const void*)wfa_kernel_wave<1>, (const void*)wfa_kernel_wave<3>})
        TRGT_HIP_TRY(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  // ---- BiWFA batches of one wave per alignment (consensus alignments, edit distances): first the LDS-arena variant over the whole list,
  //      then the HBM variant over what did not fit (its job list and count are written by the first launch)
  // ---- BiWFA batches of one wave per alignment (consensus alignments, edit distances): first the register-resident kernel (wfa_lean.hip) over
  //      the whole list, then this kernel over what it did not take (job list and count written by the first launch)
  const int64_t jobs_bound = L.jobs_bound > 0 ? L.jobs_bound : L.n_jobs_host;  // (n_jobs_host may only bound the workgroups: the retry list holds every job)
  const bool one_wave_biwfa = a.kp.biwfa && p.span == 0 && threads == 64 && !L.ops && a.fast_wcap == 0 && !L.n_jobs2_dev;
  const bool lean_ok = one_wave_biwfa && !c->knobs.no_lean && (p.metric == 1 || (p.metric == 3 && pen.x == 2 && pen.o1 == 5 && pen.e1 == 1));
  if (lean_ok) {
    // (with a device-side job count the retry list must hold every job the lean kernels may hand on -- n_jobs_host only
    //  bounds the workgroups then -- so the caller has to say how many there can be; a list that is too short would drop alignments)
    if (L.n_jobs_dev && L.jobs_bound <= 0) return fail(c, TRGT_ERR_INVALID, "wfa: a launch with a device-side job count needs jobs_bound (the capacity of the hand-over lists)");
    void* d_retry = nullptr;
    if ((rc = dev_get(c, L.buffer_set == 2 ? S_WFA_RETRY_C : L.buffer_set ? S_WFA_RETRY_B : S_WFA_RETRY, (size_t)jobs_bound * sizeof(JobDev), &d_retry))) return rc;
    void* d_mid = nullptr;
    if ((rc = dev_get(c, L.buffer_set == 2 ? S_WFA_MID_C : L.buffer_set ? S_WFA_MID_B : S_WFA_MID, 2 * (size_t)jobs_bound * sizeof(JobDev), &d_mid))) return rc;  // (two lists: between tiers one / two and two / three)
    unsigned int* const lean_words = (unsigned int*)d_counter + 4 + blocks;  // [0..23] statistics of the three tiers, [24..28] their counters
    if ((rc = wfa_lean_launch(c, p, L, (JobDev*)d_mid, (JobDev*)d_retry, (unsigned int*)d_counter + 1, (uint32_t)std::min<int64_t>(jobs_bound, 0xFFFFFFF0ll), (unsigned int*)d_counter + 3,
                              lean_words + 24, (unsigned long long*)d_cells, c->knobs.debug ? lean_words : nullptr)))
      return rc;
    a.jobs = (const JobDev*)d_retry; a.n_jobs_dev = (const uint32_t*)d_counter + 1; a.n_jobs2_dev = nullptr; a.jobs_cap = 0;
    a.counter = (unsigned int*)d_counter + 2;
  }
  const bool la_ok = !lean_ok && one_wave_biwfa && (p.metric == 1 || p.metric == 3) && !c->knobs.no_lds_wfa;
  if (la_ok) {
    void* d_retry = nullptr;
    if ((rc = dev_get(c, L.buffer_set == 2 ? S_WFA_RETRY_C : L.buffer_set ? S_WFA_RETRY_B : S_WFA_RETRY, (size_t)jobs_bound * sizeof(JobDev), &d_retry))) return rc;
    KArgs la = a;
    // dynamic LDS of a workgroup: sequences | two run-length buffers | descriptor rings (as many levels as the score scope needs) | the
    // region of the wavefronts.  Defaults sized for 12 workgroups per CU (one wave each; the registers allow no more): this kernel is all
    // latency, and the alignments that do not fit are few and go to the HBM variant.
    const uint32_t seq_cap = (uint32_t)std::max(256, c->knobs.lds_wfa_seq), rle_cap = 96, region = (uint32_t)std::max(2, c->knobs.lds_wfa_kb) * 1024u;
    uint32_t ring_levels = 2;
    while ((int)ring_levels < pen.scope) ring_levels *= 2;
    la.lds_seq_cap = seq_cap; la.la_rle_cap = rle_cap; la.la_rle_tmp = seq_cap; la.la_rle_out = seq_cap + 4 * rle_cap;
    la.ring_off = seq_cap + 8 * rle_cap; la.ring_mask = ring_levels - 1;
    la.la_region = la.ring_off + 3 * ring_levels * 5 * (uint32_t)sizeof(WfDesc); la.la_region_bytes = region;
    la.la_gdesc_slots = std::min<uint32_t>(40, (region / 3) / (5 * (uint32_t)sizeof(WfDesc)));
    la.retry_jobs = (JobDev*)d_retry; la.retry_count = (unsigned int*)d_counter + 1;
    const size_t la_lds = (size_t)la.la_region + region;
    void (*const la_fn)(const KArgs) = c->knobs.wfa_no_wave_variant ? (p.metric == 1 ? wfa_kernel<1, true> : wfa_kernel<3, true>) : (p.metric == 1 ? wfa_kernel_wave_la<1> : wfa_kernel_wave_la<3>);
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, la_fn, 64, la_lds) != hipSuccess || occ < 1) { (void)hipGetLastError(); occ = 4; }
    const int64_t la_grid = std::max<int64_t>(1, std::min<int64_t>(L.n_jobs_host, (int64_t)c->num_cus * occ));
    if (c->knobs.debug) fprintf(stderr, "[wfa] LDS-arena variant: lds=%zu (+ static) occupancy=%d grid=%lld ring levels %u gdesc slots %u\n", la_lds, occ, (long long)la_grid, ring_levels, la.la_gdesc_slots);
    hipLaunchKernelGGL(la_fn, dim3((unsigned)la_grid), dim3(64), la_lds, c->stream, la);
    TRGT_HIP_TRY(c, hipGetLastError());
    a.jobs = (const JobDev*)d_retry; a.n_jobs_dev = (const uint32_t*)d_counter + 1; a.n_jobs2_dev = nullptr; a.jobs_cap = 0;
    a.counter = (unsigned int*)d_counter + 2;
  }
  const bool wave_variant = threads == 64 && !c->knobs.wfa_no_wave_variant;  // (edit and gap-affine have one: what the locus path uses)
  if (a.fast_wcap > 0) {
    // every job of this batch qualifies for the dedicated LDS-resident kernel (wfa_fast.hpp)
    if (lds > 48 * 1024) TRGT_HIP_TRY(c, hipFuncSetAttribute((const void*)fast_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(fast_fn, grid, block, lds, c->stream, a);
  } else
  switch (p.metric) {
    case 0: hipLaunchKernelGGL((wfa_kernel<0, false>), grid, block, lds, c->stream, a); break;
    case 1: if (wave_variant) hipLaunchKernelGGL((wfa_kernel_wave<1>), grid, block, lds, c->stream, a); else hipLaunchKernelGGL((wfa_kernel<1, false>), grid, block, lds, c->stream, a); break;
    case 2: hipLaunchKernelGGL((wfa_kernel<2, false>), grid, block, lds, c->stream, a); break;
    case 3: if (wave_variant) hipLaunchKernelGGL((wfa_kernel_wave<3>), grid, block, lds, c->stream, a); else hipLaunchKernelGGL((wfa_kernel<3, false>), grid, block, lds, c->stream, a); break;
    default: hipLaunchKernelGGL((wfa_kernel<4, false>), grid, block, lds, c->stream, a); break;
  }
  { const hipError_t le = hipGetLastError();
    if (le != hipSuccess) return fail(c, TRGT_ERR_HIP, "alignment kernel launch failed: %s (buffer set %d, at most %lld jobs, metric %d, %s, lds %zu, grid %lld x %d)", hipGetErrorString(le),
                                      L.buffer_set, (long long)L.n_jobs_host, p.metric, a.fast_wcap ? "dedicated kernel" : "generic kernel", lds, (long long)grid_blocks, threads); }
  t.stop(0);
  if (lean_ok && c->knobs.debug) {  // (synchronises: developer output only)
    unsigned int h[4] = {0, 0, 0, 0};
    TRGT_HIP_TRY(c, trgt::stream_wait(c, c->stream));
    TRGT_HIP_TRY(c, hipMemcpy(h, d_counter, 16, hipMemcpyDeviceToHost));
    unsigned int w[40];
    TRGT_HIP_TRY(c, hipMemcpy(w, (unsigned int*)d_counter + 4 + blocks, 160, hipMemcpyDeviceToHost));
    fprintf(stderr, "[wfa] lean kernels: metric %d, %lld jobs at most, %u went on to the second tier, %u to the third, %u to the generic kernel (%u lost)\n", p.metric, (long long)jobs_bound, w[25], w[27], h[1], h[3]);
    for (int t = 0; t < 3; ++t)
      fprintf(stderr, "[wfa]   tier %d handed on: lengths %u, window %u, range %u, history levels %u, history cells %u, runs %u, stack %u, status %u\n", t + 1,
              w[8 * t + 0], w[8 * t + 1], w[8 * t + 2], w[8 * t + 3], w[8 * t + 4], w[8 * t + 5], w[8 * t + 6], w[8 * t + 7]);
  }
  if (la_ok && c->knobs.debug) {  // (synchronises: developer output only)
    unsigned int h[4] = {0, 0, 0, 0};
    TRGT_HIP_TRY(c, trgt::stream_wait(c, c->stream));
    TRGT_HIP_TRY(c, hipMemcpy(h, d_counter, 16, hipMemcpyDeviceToHost));
    fprintf(stderr, "[wfa] LDS-arena variant: metric %d, %lld jobs at most, %u went on to the HBM variant\n", p.metric, (long long)L.n_jobs_host, h[1]);
  }
#ifdef TRGT_WFA_PROF
  if (a.fast_wcap == 0) {
    unsigned long long h[8], z[8] = {0};
    TRGT_HIP_TRY(c, trgt::stream_wait(c, c->stream));
    TRGT_HIP_TRY(c, hipMemcpyFromSymbol(h, HIP_SYMBOL(wfa::g_wfa_gprof), sizeof h));
    TRGT_HIP_TRY(c, hipMemcpyToSymbol(HIP_SYMBOL(wfa::g_wfa_gprof), z, sizeof h));
    const double tot = (double)(h[0] + h[1] + h[2] + h[3]) + 1e-9;
    fprintf(stderr, "[wfa gprof] jobs=%lld grid=%lld thr=%d | claim+stage %.1f%% breakpoint %.1f%% base %.1f%% other %.1f%% | Mcycles total %.1f\n",
            (long long)L.n_jobs_host, (long long)grid_blocks, threads, 100 * h[0] / tot, 100 * h[1] / tot, 100 * h[2] / tot, 100 * h[3] / tot, tot / 1e6);
    unsigned long long e[16], ze[16] = {0};
    TRGT_HIP_TRY(c, hipMemcpyFromSymbol(e, HIP_SYMBOL(wfa::g_wfa_eprof), sizeof e));
    TRGT_HIP_TRY(c, hipMemcpyToSymbol(HIP_SYMBOL(wfa::g_wfa_eprof), ze, sizeof e));
    const double lv = (double)e[10] + 1e-9;
    fprintf(stderr, "[wfa eprof] levels %llu (%.1f per job) | cycles per level: fetch %.0f alloc %.0f strips %.0f close %.0f post %.0f heuristic %.0f | per job: init %.0f extend_only %.0f\n",
            e[10], lv / (double)std::max<int64_t>(1, L.n_jobs_host), e[0] / lv, e[1] / lv, e[2] / lv, e[3] / lv, e[4] / lv, e[5] / lv,
            (double)e[7] / (double)std::max<int64_t>(1, L.n_jobs_host), (double)e[8] / (double)std::max<int64_t>(1, L.n_jobs_host));
    fprintf(stderr, "[wfa eprof] inside the strips, cycles per level: loads + recurrences %.0f, extension %.0f, termination / antidiagonal atomics %.0f, stores + range reductions %.0f\n",
            e[11] / lv, e[12] / lv, e[13] / lv, e[14] / lv);
  }
  if (a.fast_wcap > 0) {
    unsigned long long h[32], lv[32], z[32] = {0};
    TRGT_HIP_TRY(c, trgt::stream_wait(c, c->stream));
    TRGT_HIP_TRY(c, hipMemcpyFromSymbol(h, HIP_SYMBOL(wfa::g_wfa_prof), sizeof h));
    TRGT_HIP_TRY(c, hipMemcpyToSymbol(HIP_SYMBOL(wfa::g_wfa_prof), z, sizeof h));
    TRGT_HIP_TRY(c, hipMemcpyFromSymbol(lv, HIP_SYMBOL(wfa::g_wfa_lvprof), sizeof lv));
    TRGT_HIP_TRY(c, hipMemcpyToSymbol(HIP_SYMBOL(wfa::g_wfa_lvprof), z, sizeof lv));
    const double lt = (double)(lv[0] + lv[1] + lv[2] + lv[3]) + 1e-9;
    const double tot = (double)(h[0] + h[1] + h[2] + h[3] + h[4] + h[5]) + 1e-9;
    fprintf(stderr, "[wfa prof] jobs=%lld grid=%lld thr=%d | fetch %.1f%% stage %.1f%% levels %.1f%% backtrace %.1f%% sync %.1f%% epilogue+idle %.1f%% | "
            "mean end_score %.1f | long(>40) %llu mean score %.1f, %.1f%% of level time | Mcycles/wg %.1f | level loop (thread 0): barrier %.1f%% "
            "prologue %.1f%% strips %.1f%% record %.1f%%\n",
            (long long)L.n_jobs_host, (long long)grid_blocks, threads, 100 * h[0] / tot, 100 * h[1] / tot, 100 * h[2] / tot, 100 * h[3] / tot,
            100 * h[4] / tot, 100 * h[5] / tot, h[6] ? (double)h[7] / h[6] : 0.0, h[16], h[16] ? (double)h[17] / h[16] : 0.0,
            h[2] ? 100.0 * h[18] / h[2] : 0.0, tot / 1e6 / (double)grid_blocks, 100 * lv[0] / lt, 100 * lv[1] / lt, 100 * lv[2] / lt, 100 * lv[3] / lt);
    for (int w = 1; w < 4; ++w) {  // the other waves' view of the level loop
      const unsigned long long* q = lv + 8 * w;
      const double t = (double)(q[0] + q[1] + q[2] + q[3]) + 1e-9;
      fprintf(stderr, "[wfa prof]   wave %d: barrier %.1f%% prologue %.1f%% strips %.1f%% record %.1f%%\n", w, 100 * q[0] / t, 100 * q[1] / t, 100 * q[2] / t, 100 * q[3] / t);
    }
  }
#endif
  c->last_wfa_cells_dev = d_cells;
  return TRGT_OK;
}

}  // namespace trgt

using namespace trgt;

namespace trgt {
namespace {
// run-length CIGARs of a batch, gathered into one dense buffer on the device: only the entries actually produced cross PCIe
__global__ void cigar_pack_kernel(const uint32_t* __restrict__ cigar, const JobDev* __restrict__ jobs, const uint32_t* __restrict__ clen,
                                  const uint64_t* __restrict__ dst_off, uint32_t* __restrict__ packed, uint64_t n) {
  const uint64_t j = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= n) return;
  const uint32_t* src = cigar + jobs[j].cigar_off;
  uint32_t* dst = packed + dst_off[j];
  for (uint32_t i = threadIdx.x & 63; i < clen[j]; i += 64) dst[i] = src[i];
}
}  // namespace

// trgt_wfa_batch proper.  packed != nullptr: the caller wants the run-length CIGARs as one dense array (job j =
// packed->data[packed->off[j] .. packed->off[j + 1])) instead of the worst-case-sized slots of the public ABI -- the internal
// callers (consensus alignments of trgt_locus_batch) read a few runs per job, not plen + tlen + 1.
int wfa_batch_impl(trgt_hip_ctx* c, const trgt_wfa_params* p, int64_t n_jobs, const uint8_t* seqs, const uint64_t* pat_off,
                   const uint32_t* pat_len, const uint64_t* txt_off, const uint32_t* txt_len, int32_t* status, int32_t* score,
                   int32_t* n_match, uint32_t* span4, uint32_t* cigar, const uint64_t* cigar_off, uint32_t* cigar_len, uint8_t* ops,
                   const uint64_t* ops_off, uint32_t* ops_len, PackedCigars* packed, const std::function<int()>* while_running,
                   WfaOnDevice* on_device) {
  if (!c) return TRGT_ERR_INVALID;
  if (!p || n_jobs < 0 || (n_jobs > 0 && (!seqs || !pat_off || !pat_len || !txt_off || !txt_len)))
    return fail(c, TRGT_ERR_INVALID, "trgt_wfa_batch: null argument");
  if ((cigar && (!cigar_off || !cigar_len)) || (ops && (!ops_off || !ops_len)))
    return fail(c, TRGT_ERR_INVALID, "trgt_wfa_batch: cigar/ops need their offset and length arrays");
  if (packed) { packed->data.clear(); packed->off.assign((size_t)n_jobs + 1, 0); }
  if (n_jobs == 0) return TRGT_OK;
  const bool tl_on = c->knobs.timeline;
  const auto tl0 = std::chrono::steady_clock::now();
#define WTL(name) do { if (tl_on) fprintf(stderr, "[tl]   wfa_batch %-18s +%6.2f ms\n", name, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tl0).count()); } while (0)
  if (n_jobs > 0xFFFFFFF0ll) return fail(c, TRGT_ERR_UNSUPPORTED, "trgt_wfa_batch: too many jobs");
  TRGT_HIP_TRY(c, hipSetDevice(c->device));
  std::vector<JobDev> jobs((size_t)n_jobs);
  WfaLaunch L;
  uint64_t seq_total = 0, cigar_total = 0, ops_total = 0;
  for (int64_t j = 0; j < n_jobs; ++j) {
    JobDev& jd = jobs[(size_t)j];
    jd.pat_off = pat_off[j]; jd.txt_off = txt_off[j]; jd.pat_len = pat_len[j]; jd.txt_len = txt_len[j]; jd.out_index = (uint32_t)j;
    jd.cigar_off = cigar ? cigar_off[j] : 0; jd.ops_off = ops ? ops_off[j] : 0;
    if (packed || on_device) { jd.cigar_off = cigar_total; cigar_total += (uint64_t)pat_len[j] + txt_len[j] + 1; }
    L.max_plen = std::max<int64_t>(L.max_plen, pat_len[j]); L.max_tlen = std::max<int64_t>(L.max_tlen, txt_len[j]);
    L.max_sum = std::max<int64_t>(L.max_sum, (int64_t)pat_len[j] + txt_len[j]);
    seq_total = std::max<uint64_t>(seq_total, std::max(pat_off[j] + pat_len[j], txt_off[j] + txt_len[j]));
    if (cigar) cigar_total = std::max<uint64_t>(cigar_total, cigar_off[j] + pat_len[j] + txt_len[j] + 1);
    if (ops) ops_total = std::max<uint64_t>(ops_total, ops_off[j] + pat_len[j] + txt_len[j]);
  }
  if (L.max_sum > (1 << 27)) return fail(c, TRGT_ERR_UNSUPPORTED, "trgt_wfa_batch: sequences too long");
  int rc;
  const uint8_t* d_seq = nullptr;
  WTL("jobs built");
  if ((rc = dev_in(c, S_WFA_SEQ, seqs, (size_t)seq_total, &d_seq))) return rc;
  WTL("sequences uploaded");
  void* d_jobs = nullptr;
  if ((rc = dev_get(c, S_WFA_JOBS, jobs.size() * sizeof(JobDev), &d_jobs))) return rc;
  if ((rc = h2d_small(c, d_jobs, jobs.data(), jobs.size() * sizeof(JobDev), c->stream, S_WFA_JOBS))) return rc;
  DevOut<int32_t> o_status, o_score, o_nm; DevOut<uint32_t> o_span, o_cigar, o_clen, o_olen; DevOut<uint8_t> o_ops;
  std::vector<uint32_t> h_clen;
  if (packed || on_device) {  // device-only CIGAR slots; the lengths come back first (packed) or not at all (on_device)
    void* d = nullptr;
    if ((rc = dev_get(c, S_WFA_CIGAR, (size_t)cigar_total * 4, &d))) return rc;
    o_cigar.dev = (uint32_t*)d;
    if (packed) { h_clen.resize((size_t)n_jobs); cigar_len = h_clen.data(); }
  }
  if ((rc = o_status.init(c, S_WFA_STATUS, status, (size_t)n_jobs)) || (rc = o_score.init(c, S_WFA_SCORE, score, (size_t)n_jobs)) ||
      (rc = o_nm.init(c, S_WFA_NMATCH, n_match, (size_t)n_jobs)) || (rc = o_span.init(c, S_WFA_SPAN, span4, (size_t)n_jobs * 4)) ||
      (!packed && !on_device && (rc = o_cigar.init(c, S_WFA_CIGAR, cigar, (size_t)cigar_total))) || (rc = o_clen.init(c, S_WFA_CLEN, cigar_len, (size_t)n_jobs)) ||
      (rc = o_ops.init(c, S_WFA_OPS, ops, (size_t)ops_total)) || (rc = o_olen.init(c, S_WFA_OLEN, ops_len, (size_t)n_jobs)))
    return rc;
  if (on_device && !o_clen.dev) {  // (no host destination: a device-only array)
    void* d = nullptr;
    if ((rc = dev_get(c, S_WFA_CLEN, (size_t)n_jobs * 4, &d))) return rc;
    o_clen.dev = (uint32_t*)d;
  }
  if (on_device) { on_device->jobs = (const JobDev*)d_jobs; on_device->cigar = o_cigar.dev; on_device->cigar_len = o_clen.dev; on_device->seqs = d_seq; }
  L.jobs_dev = (const JobDev*)d_jobs; L.n_jobs_host = n_jobs; L.n_jobs_dev = nullptr;
  L.buffer_set = 1;  // trgt_locus_batch runs consensus alignments next to a flank-location launch (set 0) of a later chunk
  L.pat_base = d_seq; L.txt_base = d_seq;
  L.status = o_status.dev; L.score = o_score.dev; L.n_match = o_nm.dev; L.span4 = o_span.dev; L.cigar = o_cigar.dev;
  L.cigar_len = o_clen.dev; L.ops = o_ops.dev; L.ops_len = o_olen.dev;
  WTL("buffers ready");
  if ((rc = wfa_launch(c, *p, L))) return rc;
  WTL("launched");
  // host work of the caller goes here: the copies below end in pageable memory, i.e. they block until the kernel is done
  if (while_running && *while_running) { const int wrc = (*while_running)(); if (wrc) { (void)trgt::stream_wait(c, c->stream); return wrc; } }
  if ((rc = o_status.finish(c)) || (rc = o_score.finish(c)) || (rc = o_nm.finish(c)) || (rc = o_span.finish(c)) ||
      (rc = o_cigar.finish(c)) || (rc = o_clen.finish(c)) || (rc = o_ops.finish(c)) || (rc = o_olen.finish(c)))
    return rc;
  unsigned long long cells = 0;
  void* const cells_dev = c->last_wfa_cells_dev;  // (the callback may launch alignments of its own)
  hipStream_t const my_stream = c->stream;
  { const int d2h_rc = trgt::d2h(c, &cells, cells_dev, 8, my_stream); if (d2h_rc) return d2h_rc; }
  TRGT_HIP_TRY(c, trgt::stream_wait(c, c->stream));
  if (c->timing) c->k_cells[TRGT_K_WFA] += (int64_t)cells;
  if (packed) {
    uint64_t total = 0;
    for (int64_t j = 0; j < n_jobs; ++j) { packed->off[(size_t)j] = total; total += h_clen[(size_t)j]; }
    packed->off[(size_t)n_jobs] = total;
    packed->data.resize((size_t)total);
    if (total) {
      void *d_poff = nullptr, *d_packed = nullptr;
      if ((rc = dev_get(c, S_WFA_POFF, (size_t)n_jobs * 8, &d_poff)) || (rc = dev_get(c, S_WFA_PACKED, (size_t)total * 4, &d_packed))) return rc;
      if ((rc = h2d_small(c, d_poff, packed->off.data(), (size_t)n_jobs * 8, c->stream, S_WFA_POFF))) return rc;
      hipLaunchKernelGGL(cigar_pack_kernel, dim3((unsigned)((n_jobs + 3) / 4)), dim3(256), 0, c->stream, (const uint32_t*)o_cigar.dev,
                         (const JobDev*)d_jobs, (const uint32_t*)o_clen.dev, (const uint64_t*)d_poff, (uint32_t*)d_packed, (uint64_t)n_jobs);
      TRGT_HIP_TRY(c, hipGetLastError());
      { const int d2h_rc = trgt::d2h(c, packed->data.data(), d_packed, (size_t)total * 4, c->stream); if (d2h_rc) return d2h_rc; }
      TRGT_HIP_TRY(c, trgt::stream_wait(c, c->stream));
    }
  }
  return TRGT_OK;
}
}  // namespace trgt

extern "C" int trgt_wfa_batch(trgt_hip_ctx* c, const trgt_wfa_params* p, int64_t n_jobs, const uint8_t* seqs,
                              const uint64_t* pat_off, const uint32_t* pat_len, const uint64_t* txt_off,
                              const uint32_t* txt_len, int32_t* status, int32_t* score, int32_t* n_match, uint32_t* span4,
                              uint32_t* cigar, const uint64_t* cigar_off, uint32_t* cigar_len, uint8_t* ops,
                              const uint64_t* ops_off, uint32_t* ops_len) {
  return trgt::wfa_batch_impl(c, p, n_jobs, seqs, pat_off, pat_len, txt_off, txt_len, status, score, n_match, span4, cigar, cigar_off,
                              cigar_len, ops, ops_off, ops_len, nullptr);
}
