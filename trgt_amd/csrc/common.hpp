// trgt_amd/csrc/common.hpp -- shared host-side plumbing of libtrgt_hip.so:
// the opaque ctx, error propagation across the C ABI, host/device pointer
// classification, a small device-buffer pool and per-kernel HIP-event timing.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/trgt_hip.h"

// Planner knobs.  In a release build none of them changes a result: they are read from the environment ONCE, when the context is
// created (trgt_hip_create), so that the parity tests can pin every planner path against the oracle.  The switches that DO change
// results -- TRGT_SENS_* (the un-pinned choices flipped, tools/unpinned_sensitivity.py) and TRGT_DBG_SKIP_BT -- are read only under
// `make DEV=1` (TRGT_DEV_BUILD); in the default build their fields keep the defaults below and the names are not in the binary
// (tests/test_abi_exports.py checks).
// Developer switches (settled A/Bs, probes) are read only in `make DEV=1` builds: in the release library the expression is a null pointer
// and the name is not in the binary.
#ifdef TRGT_DEV_BUILD
#define TRGT_DEV_ENV(name) getenv(name)
#else
#define TRGT_DEV_ENV(name) ((const char*)nullptr)
#endif

struct trgt_knobs {
  int filter_force = 0;      // TRGT_FILTER_FORCE (DEV): one instantiation of the pre-filter for the whole launch (42 / 52 / 71 / 91), developer probe
  int flank_threads = 256;   // TRGT_FLANK_THREADS: threads per flank alignment of the back-tracing kernel
  int heavy_band = 96;       // TRGT_HEAVY_BAND: the back-trace of what the pre-filter keeps runs inside the band its penalty allows when that is at most this (0: off)
  int heavy_threads = 0;     // TRGT_HEAVY_THREADS: ... of its launch over the expensive alignments (0: 192 when flank_threads == 256)
  int band_threads = 64;     // TRGT_BAND_THREADS: ... of the banded back-trace of what the pre-filter keeps (64, 128 or 256)
  int win_threads = 64;      // TRGT_WIN_THREADS: ... of the windowed launch
  int win_segments = 8;      // TRGT_WIN_SEGMENTS: 4 / 6 / 8 segments for the window search
  int grid_per_cu = 0;       // TRGT_WFA_GRID_PER_CU: persistent workgroups per CU of the dedicated kernel (0: occupancy query)
  int filter_per_cu = 0;     // TRGT_FILTER_PER_CU: persistent waves per CU of the pre-filter (0: occupancy query)
  bool one_launch = false;   // TRGT_WFA_ONE_LAUNCH: all flank alignments in one launch
  bool no_spec = false;      // TRGT_WFA_NO_SPEC: general instantiation of the dedicated kernel
  bool no_window = false;    // TRGT_WFA_NO_WINDOW: no seeded windows
  bool early_adaptive = true;   // TRGT_EARLY_ADAPTIVE=0: the pre-filter's early-rejection test every 16th level (else scheduled by the smallest deficit seen)
  bool no_heavy_window = false;  // TRGT_NO_HEAVY_WINDOW: the expensive fallback alignments go to the pre-filter without the seed search first
  bool no_indel_shortcut = false;  // TRGT_NO_INDEL_SHORTCUT: one-base gaps are aligned (the substitution shortcut stays)
  bool no_hamming = false;   // TRGT_NO_HAMMING: no substitution-only shortcut in the window search (every light fallback is aligned)
  bool no_filter = false;    // TRGT_WFA_NO_FILTER: no pre-filter in front of the expensive alignments
  bool one_stream = false;   // TRGT_FLANK_ONE_STREAM: the expensive flank alignments in front of the others instead of next to them
  bool host_genotyper = false;  // TRGT_HOST_GENOTYPER: host glue for every locus
  bool no_early = false;        // TRGT_WFA_NO_EARLY: the flank pre-filter runs every alignment to its end
  bool stage_lock = false;      // TRGT_STAGE_LOCK: only one context per device in its flank-location stage at a time
  bool host_hmm_lists = false;  // TRGT_HOST_HMM_LISTS: stage C job lists built by the host after the genotyper (not resolved on the device)
  bool debug = false;        // TRGT_WFA_DEBUG: launch plans on stderr (synchronises)
  bool hmm_resolve_one_wg = false;   // TRGT_HMM_RESOLVE_ONE_WG: the stage-C job list by the one-workgroup kernel, class after class
  bool wfa_no_wave_variant = false;  // TRGT_WFA_NO_WAVE_VARIANT: one-wave batches of the generic kernel take the 168-register build (three waves per SIMD)
  bool wfa_no_stage = false;       // TRGT_WFA_NO_STAGE: the generic kernel extends from global memory, sequences are not staged in LDS
  bool filter_side = false;        // TRGT_FILTER_SIDE: ... next to each other in the contexts of a pool too
  bool filter_serial = false;      // TRGT_FILTER_SERIAL: the pre-filter's two launches one after the other on one stream
  bool filter_one_launch = false;  // TRGT_FILTER_ONE_LAUNCH: the pre-filter in one launch whatever the text lengths
  bool no_long_window = false;  // TRGT_NO_LONG_WINDOW: the long reads' alignments skip the seed search (shortcuts, seeded windows)
  bool no_long_filter = false;  // TRGT_NO_LONG_FILTER: long reads straight to the exact kernel (no window-by-window pre-filter)
  bool no_lean = false;      // TRGT_WFA_NO_LEAN: consensus alignments / edit distances straight to the generic kernel (no register-resident BiWFA kernel in front)
  bool lean_mid_tier = false;  // TRGT_WFA_LEAN_MID_TIER: a 128-diagonal tier between the 64- and the 256-diagonal lean kernels (measured slower on cfg5: the tiers' tails add up)
  int lean_chunk = 0;          // TRGT_LEAN_CHUNK: alignments per claim of the lean kernels' job counter (0: 2, 4 in a pool; edit distances always 8)
  bool no_zero_arena = false;  // TRGT_NO_ZERO_ARENA: counters and small lists cleared by a hipMemsetAsync each (round 4) instead of one arena clear per call
  bool hmm_ppl_per_class = false;  // TRGT_HMM_PPL_PER_CLASS: the position-per-lane fills of a locus batch launched class by class (until late in round 5) instead of once per group width over all classes
  bool hmm_ppl_wide = false;   // TRGT_HMM_PPL_WIDE: the position-per-lane fill writes the round-5 rows of one byte per state (default: one byte per lane)
  bool hmm_ppl_serial = false; // TRGT_HMM_PPL_SERIAL: the position-per-lane fills of one class one after the other on the class's stream (no side streams)
  bool lean_one_tier = false;  // TRGT_WFA_LEAN_ONE_TIER: no second tier (256 diagonals) between the register-resident kernel and the generic one
  bool no_lds_wfa = true;    // TRGT_WFA_LDS=1 turns the LDS-arena variant of the BiWFA kernel on (in front of the HBM-arena one).  Off by default:
                             // measured on cfg5 it is no faster -- the generic engine spends its time in instructions, not in HBM latency
  int lds_wfa_kb = 7;        // TRGT_WFA_LDS_KB: LDS of the LDS-arena variant for the wavefronts of one alignment
  int lds_wfa_seq = 768;     // TRGT_WFA_LDS_SEQ: ... for its two sequences (padded pattern + text)
  // tools/unpinned_sensitivity.py (make DEV=1 only): the decisions inside un-vendored dependencies that no reference test pins, flipped one at a time
  int sens_bialign_min_len = -1;  // TRGT_SENS_BIALIGN_MIN_LEN: bialign_min_length of the consensus alignments / edit distances (default 100; SURVEY A.7 read literally: 0)
  bool sens_cons_unidir = false;  // TRGT_SENS_CONS_UNIDIR: consensus alignments back-traced unidirectionally (MemoryHigh) instead of by BiWFA
  bool sens_ward_ties = false;    // TRGT_SENS_WARD_TIES: nearest-neighbour ties of the Ward linkage go to the LAST candidate instead of the first
  bool sens_lw_order = false;     // TRGT_SENS_LW_ORDER: the Lance-Williams update summed in another order (last bits of the matrix central_read reads)
  bool hmm_no_dedupe = false;   // TRGT_HMM_NO_DEDUPE: the second allele of a homozygous locus is labelled by an HMM job of its own (as the reference does) instead of taking the first one's results
  bool hmm_no_long_tb = false;  // TRGT_HMM_NO_LONG_TB: alleles of 1 536 columns and more are traced back by the fill kernel's one lane too (not by hmm_traceback_long_kernel)
  int hmm_long_wgs = 4;  // TRGT_HMM_LONG_WGS: workgroups per long allele in the chunk-map and re-walk passes of the long trace-back (1: one workgroup does everything in one launch)
  bool hmm_no_ppl = false;  // TRGT_HMM_NO_PPL: no position-per-lane fill (hmm_ppl.hpp) in front of the HMM kernels: every set is filled by hmm_viterbi_kernel with one lane per state, as in round 4
  bool hmm_four_rounds = false;  // TRGT_HMM_FOUR_ROUNDS: the register fill fetches across lanes once per pass of a column (four rounds) instead of twice per column
  bool hmm_lds_fill = false;  // TRGT_HMM_LDS_FILL: one-wave motif sets fill their Viterbi columns through LDS like the larger ones (not in registers)
  int cluster_arena_kb = 0;  // TRGT_CLUSTER_ARENA_KB: developer switch -- the CIGAR / result / scratch arenas of the device-side cluster genotyper capped at this many KB (loci that find no room take the host path: the mixed case of the tests)
  bool host_cluster = false; // TRGT_HOST_CLUSTER: Genotyper::Cluster loci take the host path (linkage, groups and round sequencing on host threads, locus_cluster.hpp)
  bool host_repair = false;  // TRGT_HOST_REPAIR: loci whose pick lacks majority support go back to the host (no device-side consensus repair)
  bool split_hmm = false;    // TRGT_SPLIT_HMM: the HMM of the loci the genotyper settles next to the device-side repair of the others, a second batch behind it
  int repair_blocks = 2048;  // TRGT_REPAIR_BLOCKS: workgroups (and workspaces) of the alignment kernel of the device-side repair
  int repair_max_seg = 0;     // TRGT_REPAIR_MAX_SEG: longest repeat segment the device-side repair takes; 0 = by the batch: what its longest read can hold, between 1 024 and 16 384 bases
  bool timeline = false;     // TRGT_TIMELINE: host-side timeline of a call on stderr
  bool skip_bt = false;      // TRGT_DBG_SKIP_BT (make DEV=1 only): skip back-traces -- timing experiments, results are wrong
};

struct trgt_hip_ctx {
  trgt_knobs knobs;
  int device = -1;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int stream_priority = 0;  // of every stream the context creates (0: the default)
  bool in_pool = false;     // created by trgt_hip_pool_create
  std::string err;
  uint64_t ws_limit = 32ull << 30;
  int num_cus = 256;
  // cached device buffers, indexed by slot
  struct Buf { void* p = nullptr; size_t cap = 0; };
  std::vector<Buf> pool;
  // timing
  bool timing = false;
  double k_ms[TRGT_K_COUNT] = {};
  int64_t k_launches[TRGT_K_COUNT] = {};
  int64_t k_cells[TRGT_K_COUNT] = {};
  struct Pending { int k; hipEvent_t a, b; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> retired_events;  // resolved timing events: neither destroyed nor re-recorded while calls are being timed (see resolve_timing)
  // zero arena: the small device counters a call needs cleared (job counters, offset counters, count blocks) are carved out of ONE
  // buffer that one kernel clears when the call starts (trgt::zero_begin) -- they used to be 15-25 hipMemsetAsync dispatches per call
  void* zero_arena = nullptr; size_t zero_cap = 0, zero_used = 0, zero_dirty = 0; bool zero_on = false;
  void* wfa_cells_cur[3] = {nullptr, nullptr, nullptr};  // offset counter of the current logical batch of each buffer set (wfa_launch, keep_cells)
  void* last_wfa_cells_dev = nullptr;
  void* last_filter_cells_dev = nullptr;
  int64_t dbg_ns[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int64_t tl_t0 = 0;  // TRGT_TIMELINE: start of the call being traced (steady clock, ns)  // host-side phase timers of the last call (diagnostics)
  // trgt_locus_batch pipelines chunks of loci: stage A of chunk k+1 runs on `stream` while the host glue of chunk k and its
  // small transfers / gather kernel use `stream2`; pinned host buffers (slot-indexed like `pool`) make those copies asynchronous
  hipStream_t stream2 = nullptr;
  struct PinBuf { void* p = nullptr; size_t cap = 0; };
  std::vector<PinBuf> pinned;
  std::vector<PinBuf> h2d_stage;  // pinned staging of the small uploads, one per device slot (h2d_small)
  // downloads into pageable memory go through pinned chunks and reach their destination when the stream is waited for (trgt::d2h)
  struct D2hChunk { void* p = nullptr; size_t cap = 0, used = 0; };
  struct D2hPending { void* dst; const void* staged; size_t bytes; hipStream_t stream; };
  std::vector<D2hChunk> d2h_chunks;
  std::vector<D2hPending> d2h_pending;
  void* host_pool = nullptr;  // trgt::HostPool*, created on first use
  int host_pool_threads = 0;
  // trgt_locus_batch_submit / _wait: two staging sets for the read and flank bytes of batches whose upload runs on `stream_copy`
  // next to the kernels of the batch before
  hipStream_t stream_copy = nullptr;
  hipEvent_t ev_upload = nullptr;  // behind the small uploads of a call that go through stream_copy (HMM models, candidate job table)
  struct Staged {
    bool in_use = false; int64_t ticket = 0;
    trgt_locus_params params; const trgt_locus_batch_in* in = nullptr; trgt_locus_batch_out* out = nullptr;
    const uint8_t* d_reads = nullptr; const uint8_t* d_flank = nullptr;  // staged copies (nullptr: the caller's pointer is used as it is)
    uint64_t read_bytes = 0, flank_bytes = 0;
    bool copy_pending = false;  // the copy is issued from inside the wait of the batch before (see issue_pending_uploads)
    hipEvent_t ready = nullptr;
  } staged[2];
  int64_t next_ticket = 1;
  // side streams for the launches of one HMM batch (one per workgroup-size class: they run next to each other, not one behind the
  // other's tail), with the events that fork them off the batch's stream and join them back
  hipStream_t stream_flt = nullptr;  // the pre-filter's launch over the long texts, next to the one over the others
  hipEvent_t ev_flt_a = nullptr, ev_flt_b = nullptr;
  hipStream_t stream_hmm = nullptr;  // the first device-resolved HMM batch of a call, when the device-side repair runs next to it
  hipEvent_t ev_gt = nullptr, ev_rp = nullptr;  // fork / join of the device-side consensus repair (second stream) next to the first HMM batch
  hipEvent_t ev_scan = nullptr, ev_heavy = nullptr, ev_hwin = nullptr;  // find_spans_device: fork / join of the stream with the expensive flank alignments
  // side streams of the HMM launches: [0..2] of buffer set 0, [3..5] of buffer set 1 (the second batch of a call runs next to the first)
  hipStream_t hmm_side[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t hmm_fork[2] = {nullptr, nullptr}, hmm_join[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  // the position-per-lane fills of ONE class with several group widths run next to each other too (each is as long as its longest allele):
  // [buffer set][class slot][extra launch] streams and join events, [buffer set][class slot] fork events
  hipStream_t hmm_ppl_side[2][4][3] = {};
  hipEvent_t hmm_ppl_join[2][4][3] = {}, hmm_ppl_fork[2][4] = {};
};

namespace trgt {

inline int64_t wall_ns() {
  timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
  return (int64_t)ts.tv_sec * 1000000000ll + ts.tv_nsec;
}

inline int fail(trgt_hip_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf;
  return code;
}

#define TRGT_HIP_TRY(ctx, expr)                                                                       \
  do {                                                                                                \
    hipError_t e__ = (expr);                                                                          \
    if (e__ != hipSuccess)                                                                            \
      return trgt::fail((ctx), TRGT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                        __FILE__, __LINE__);                                                          \
  } while (0)

// Waiting for a stream or an event: the runtime's own wait by default; TRGT_POLL_WAIT=1 polls the completion state instead (spins
// for the first 2 ms, then every 20 us), which was written while hunting the stalls that turned out to be malloc's (see ctx.hip) and
// measures the same since.
// TRGT_POLL_SPIN_US: how long a polling wait spins before it starts to nap (default 2000); TRGT_POLL_NAP_US: the nap (default 20).  A host
// whose cgroup grants fewer CPUs than ranks x contexts want to spin on (bench.py sets these then) gives the waits up sooner.
inline long poll_knob(const char* name, long dflt) { const char* e = getenv(name); if (!e || !*e) return dflt; const long v = std::atol(e); return v >= 0 ? v : dflt; }
inline long poll_spin_ns() { static const long v = poll_knob("TRGT_POLL_SPIN_US", 2000) * 1000; return v; }
inline long poll_nap_ns() { static const long v = std::max(1l, poll_knob("TRGT_POLL_NAP_US", 20)) * 1000; return v; }
inline bool poll_wait_knob() { static const bool on = [] { const char* e = getenv("TRGT_POLL_WAIT"); return e && *e && std::strcmp(e, "0") != 0; }(); return on; }
template <class Query>
inline hipError_t poll_until_ready(Query q) {
  timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
  for (unsigned it = 0;; ++it) {
    const hipError_t e = q();
    if (e != hipErrorNotReady) {
      if (it) (void)hipGetLastError();  // (the "not ready" answers of this thread must not surface in a later hipGetLastError())
      return e;
    }
    if ((it & 63) == 63) {
      timespec t; clock_gettime(CLOCK_MONOTONIC, &t);
      if ((t.tv_sec - t0.tv_sec) * 1000000000ll + (t.tv_nsec - t0.tv_nsec) > (long long)poll_spin_ns()) { const timespec nap{0, poll_nap_ns()}; nanosleep(&nap, nullptr); }
    }
    __builtin_ia32_pause();
  }
}
inline hipError_t stream_wait(hipStream_t s) {
  if (!poll_wait_knob()) return hipStreamSynchronize(s);
  return poll_until_ready([s] { return hipStreamQuery(s); });
}
inline hipError_t event_wait(hipEvent_t ev) {
  if (!poll_wait_knob()) return hipEventSynchronize(ev);
  return poll_until_ready([ev] { return hipEventQuery(ev); });
}

// TRGT_TIMELINE: a mark of the host-side timeline from inside the enqueue path (ms since the call started)
inline void tl_mark(const trgt_hip_ctx* c, const char* name) {
  if (!c->knobs.timeline || !c->tl_t0) return;
  timespec t; clock_gettime(CLOCK_MONOTONIC, &t);
  fprintf(stderr, "[tl]   . %-26s %7.3f ms\n", name, (double)((int64_t)t.tv_sec * 1000000000ll + t.tv_nsec - c->tl_t0) / 1e6);
}

inline bool is_device_ptr(const void* p) {
  if (p == nullptr) return false;
  hipPointerAttribute_t at;
  hipError_t e = hipPointerGetAttributes(&at, p);
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }
  return at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged;
}

// pool slot ids (each call site owns a range so buffers are reused call to call)
enum Slot {
  S_HMM_SEQ = 0, S_HMM_DESC, S_HMM_MODEL, S_HMM_JOBS, S_HMM_BP, S_HMM_PATH, S_HMM_SPANS, S_HMM_NSP, S_HMM_CNT, S_HMM_PUR,
  S_HMM_EDIT, S_HMM_MAXD, S_HMM_PLEN, S_HMM_VISITS, S_HMM_LONG, S_HMM_MOTIFS,
  S_HMM_B_BASE, S_HMM_B_LAST = S_HMM_B_BASE + (S_HMM_MOTIFS - S_HMM_SEQ),  // second set of the HMM slots: two batches in flight
  S_WFA_SEQ, S_WFA_JOBS, S_WFA_WS, S_WFA_STATUS, S_WFA_SCORE, S_WFA_NMATCH, S_WFA_SPAN, S_WFA_CIGAR, S_WFA_CLEN, S_WFA_OPS,
  S_WFA_OLEN, S_WFA_COUNTER, S_WFA_CELLS, S_WFA_WS_B, S_WFA_COUNTER_B, S_WFA_CELLS_B, S_WFA_POFF, S_WFA_PACKED, S_WFA_RETRY, S_WFA_RETRY_B, S_WFA_WS_C, S_WFA_COUNTER_C, S_WFA_CELLS_C, S_WFA_RETRY_C, S_WFA_MID, S_WFA_MID_B, S_WFA_MID_C,
  S_FS_FLANK, S_FS_READS, S_FS_JOBS, S_FS_POS, S_FS_LIST, S_FS_COUNT, S_FS_OUT0, S_FS_OUT1, S_FS_HIT0, S_FS_HIT1,
  S_FS_WFAJOBS, S_FS_WFAJOBS_LONG, S_FS_KEEPJOBS, S_PF_READS0, S_PF_READS1, S_PF_FLANK0, S_PF_FLANK1, S_READS_PACKED, S_READS_EXPANDED, S_FLT_COUNTER, S_FLT_CELLS, S_FLT_COUNTER_B, S_FLT_CELLS_B, S_LW_FIRST, S_LW_SUB, S_LW_PARENT, S_LW_SUBKEEP, S_LW_JOBKEEP, S_LW_KEPT, S_LW_COUNT, S_FLT_SEQ, S_FLT_JOBS, S_FLT_SCORE, S_FLT_BOUND, S_FLT_KEEP, S_FS_WINJOBS, S_FS_RESTJOBS, S_FS_SCORE, S_FS_SPAN, S_FS_NMATCH, S_FS_HEAVY, S_FS_NOSEED, S_FS_BANDJOBS, S_FS_HRESTJOBS, S_FS_BSCORE, S_FS_LONGNOSEED, S_LW_SUBBAND, S_LW_JOBBEST, S_LW_JOBREJ, S_LW_BANDJOBS, S_LW_RESTJOBS, S_LW_BSCORE,
  S_LOCUS_0, S_LOCUS_1, S_LOCUS_2, S_LOCUS_3, S_LOCUS_4, S_LOCUS_5, S_LOCUS_6, S_LOCUS_7,
  S_GT_LRB, S_GT_PLOIDY, S_GT_TR, S_GT_TROFF, S_GT_TRLEN, S_GT_ALOFF, S_GT_ALCAP, S_GT_NEED, S_GT_NAL, S_GT_BLOB, S_GT_ALEN, S_GT_CI, S_GT_NSP,
  S_GT_CLS, S_GT_RANK, S_GT_NSPAN, S_GT_TOFF, S_GT_PACKED, S_GT_GENO,
  S_HMM_BUILD,  // inputs of the device-side model builder (one slab)
  S_VOTE_GROUPS, S_VOTE_SCRATCH, S_VOTE_OUT, S_VOTE_LEN,  // consensus column voting (consensus_vote.hpp)
  S_RP_GROUPS, S_RP_JOBS, S_RP_LOCI, S_RP_PEND, S_RP_CIGAR, S_RP_CLEN, S_RP_VOUT, S_RP_VLEN, S_RP_VSCR, S_RP_COUNTS,  // device-side consensus repair (locus_gt.hpp)
  S_CL_LIST, S_CL_MOFF, S_CL_COUNTS, S_CL_REC, S_CL_CLS, S_CL_ESCORE, S_CL_GMAT, S_CL_EDJOBS, S_CL_JOBS, S_CL_GROUPS, S_CL_ED2JOBS, S_CL_ESCORE2, S_CL_CIGAR, S_CL_CLEN,
  S_CL_VOUT, S_CL_VLEN, S_CL_VSCR,  // device-side cluster genotyper (locus_cluster_dev.hpp)
  S_INF_SRC, S_INF_DESC, S_INF_DST, S_INF_STATUS, S_INF_COUNTER,  // device-side BGZF inflate (inflate_dev.hip)
  S_ZERO_ARENA,  // trgt::zero_begin / zero_take
  S_DEFL_SCRATCH,  // device-side BGZF deflate (deflate_dev.hip): the lanes' bit strings
  S_COUNT
};
// pinned host buffer slots
enum PinSlot { P_SPAN_S = 0, P_SPAN_E, P_HIT_L, P_HIT_R, P_CELLS, P_HMM_SEQ, P_HMM_SEQ_B, P_HMM_JOBS, P_HMM_JOBS_B, P_SEG0, P_SEG_META, P_GT_NEED, P_GT_NAL, P_GT_ALEN, P_GT_CI, P_GT_NSP, P_GT_CLS,
               P_GT_RANK, P_GT_NSPAN, P_GT_TOFF, P_GT_PACKED, P_HMM_BUILD, P_COUNT };

inline int dev_get(trgt_hip_ctx* c, int slot, size_t bytes, void** out) {
  if ((int)c->pool.size() < S_COUNT) c->pool.resize(S_COUNT);
  auto& b = c->pool[slot];
  if (bytes == 0) bytes = 16;
  if (b.cap < bytes) {
    if (b.p) { TRGT_HIP_TRY(c, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    size_t want = bytes + bytes / 8 + 256;
    if (c->knobs.debug && want > ((size_t)256 << 20)) fprintf(stderr, "[mem] slot %d: %.2f GB\n", slot, (double)want / (double)(1ull << 30));
    // A large buffer must leave the runtime room for what it allocates itself when a kernel is launched (scratch for the kernels with
    // spills, queue and signal memory): with a few dozen MB of HBM left, ROCm aborts the whole process from a queue callback
    // (HSA_STATUS_ERROR_OUT_OF_RESOURCES) instead of failing a call.  So the call fails here, with a message, while that is still possible.
    if (want >= ((size_t)64 << 20)) {
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const size_t reserve = (size_t)1 << 30;
        if (free_b < want || free_b - want < reserve)
          return fail(c, TRGT_ERR_NOMEM, "%zu bytes of device memory wanted, %zu free: less than 1 GB would be left for the runtime (fewer contexts per GPU, or a smaller trgt_hip_set_workspace_limit)", want, free_b);
      } else (void)hipGetLastError();
    }
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      b.p = nullptr;
      return fail(c, TRGT_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
    }
    b.cap = want;
  }
  *out = b.p;
  return TRGT_OK;
}

inline int pin_get(trgt_hip_ctx* c, int slot, size_t bytes, void** out) {
  if ((int)c->pinned.size() < P_COUNT) c->pinned.resize(P_COUNT);
  auto& b = c->pinned[slot];
  if (bytes == 0) bytes = 16;
  if (b.cap < bytes) {
    if (b.p) { TRGT_HIP_TRY(c, hipHostFree(b.p)); b.p = nullptr; b.cap = 0; }
    const size_t want = bytes + bytes / 4 + 4096;
    hipError_t e = hipHostMalloc(&b.p, want, hipHostMallocDefault);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      b.p = nullptr;
      return fail(c, TRGT_ERR_NOMEM, "hipHostMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
    }
    b.cap = want;
  }
  *out = b.p;
  return TRGT_OK;
}

// ---- zero arena.  zero_begin (start of trgt_locus_batch, on the call's stream) clears what the call before handed out; zero_take
// hands out cleared, 64-byte aligned pieces until the call ends (zero_end).  Every stream a call uses forks off its main stream behind
// this point, so a piece is clear wherever it is first touched.  Outside a call, or when the arena is used up, zero_take returns nullptr
// and the caller clears a buffer of its own with hipMemsetAsync as before.
constexpr size_t ZERO_ARENA_BYTES = 2u << 20;
static __global__ void zero_arena_kernel(uint4* __restrict__ p, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(0u, 0u, 0u, 0u);
}
inline int zero_begin(trgt_hip_ctx* c) {
  c->zero_on = false;
  if (c->knobs.no_zero_arena) return TRGT_OK;
  void* p = nullptr;
  if (int rc = dev_get(c, S_ZERO_ARENA, ZERO_ARENA_BYTES, &p)) return rc;
  const bool fresh = p != c->zero_arena;
  c->zero_arena = p; c->zero_cap = ZERO_ARENA_BYTES;
  const size_t dirty = fresh ? ZERO_ARENA_BYTES : std::min(ZERO_ARENA_BYTES, (std::max(c->zero_dirty, c->zero_used) + 4095) & ~(size_t)4095);
  if (dirty) {
    const size_t n16 = dirty / 16;
    hipLaunchKernelGGL(zero_arena_kernel, dim3((unsigned)std::min<size_t>(64, (n16 + 255) / 256)), dim3(256), 0, c->stream, (uint4*)p, n16);
    TRGT_HIP_TRY(c, hipGetLastError());
  }
  c->zero_used = 0; c->zero_dirty = 0; c->zero_on = true;
  c->wfa_cells_cur[0] = c->wfa_cells_cur[1] = c->wfa_cells_cur[2] = nullptr;
  return TRGT_OK;
}
inline void zero_end(trgt_hip_ctx* c) {
  if (c->zero_on) { c->zero_dirty = std::max(c->zero_dirty, c->zero_used); c->zero_on = false; }
  c->wfa_cells_cur[0] = c->wfa_cells_cur[1] = c->wfa_cells_cur[2] = nullptr;  // (pieces of the arena: not valid beyond the call)
}
inline void* zero_take(trgt_hip_ctx* c, size_t bytes) {
  if (!c->zero_on) return nullptr;
  const size_t need = (bytes + 63) & ~(size_t)63;
  if (c->zero_used + need > c->zero_cap) return nullptr;
  void* p = (uint8_t*)c->zero_arena + c->zero_used;
  c->zero_used += need;
  return p;
}
// a cleared device buffer of `bytes`: from the arena inside a call, else the pool slot cleared by hipMemsetAsync on `stream`
inline int dev_get_zeroed(trgt_hip_ctx* c, int slot, size_t bytes, void** out, hipStream_t stream) {
  if (void* z = zero_take(c, bytes)) { *out = z; return TRGT_OK; }
  if (int rc = dev_get(c, slot, bytes, out)) return rc;
  TRGT_HIP_TRY(c, hipMemsetAsync(*out, 0, bytes, stream));
  return TRGT_OK;
}

// Small host -> device uploads (offset tables, job lists, model inputs: KB to a few MB) go through a KERNEL that reads pinned host
// memory, not through hipMemcpyAsync: the copy engine serves the host-to-device copies of ALL streams and contexts in issue order, and
// a table queued behind another call's 360-MB read upload waits milliseconds for it (several contexts per GPU, or the pipelined
// submit / wait entry points, then run at 70 % of the link rate).  Bulk data (the read and flank blobs) stays on the copy engine.
constexpr size_t H2D_KERNEL_MAX = 8u << 20;
static __global__ void h2d_copy_kernel(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, size_t bytes) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, n = (size_t)gridDim.x * blockDim.x;
  if ((((uintptr_t)dst | (uintptr_t)src) & 15u) == 0) {
    const size_t n16 = bytes >> 4;
    for (size_t k = i; k < n16; k += n) reinterpret_cast<uint4*>(dst)[k] = reinterpret_cast<const uint4*>(src)[k];
    for (size_t k = (n16 << 4) + i; k < bytes; k += n) dst[k] = src[k];
  } else for (size_t k = i; k < bytes; k += n) dst[k] = src[k];
}
inline bool is_pinned_host_ptr(const void* p) {
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return at.type == hipMemoryTypeHost;
}
// dst (host) <- src (device) on `stream`.  A download into pageable memory is a synchronous call in the HIP runtime (it returns when
// the stream has reached it); so pageable destinations are served from pinned chunks: the copy lands there and is moved to dst by
// the stream_wait(c, stream) that follows, and the host is free until then.  dst must stay valid until that wait.
inline int d2h(trgt_hip_ctx* c, void* dst, const void* src, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return TRGT_OK;
  if (is_pinned_host_ptr(dst)) { TRGT_HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream)); return TRGT_OK; }
  const size_t need = (bytes + 63) & ~(size_t)63;
  trgt_hip_ctx::D2hChunk* ch = nullptr;
  for (auto& k : c->d2h_chunks) if (k.cap - k.used >= need) { ch = &k; break; }
  if (!ch) {
    trgt_hip_ctx::D2hChunk k;
    k.cap = std::max<size_t>(need, (size_t)4 << 20);
    if (hipHostMalloc(&k.p, k.cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return fail(c, TRGT_ERR_NOMEM, "hipHostMalloc(%zu bytes) failed", k.cap); }
    c->d2h_chunks.push_back(k);
    ch = &c->d2h_chunks.back();
  }
  void* staged = (uint8_t*)ch->p + ch->used;
  ch->used += need;
  TRGT_HIP_TRY(c, hipMemcpyAsync(staged, src, bytes, hipMemcpyDeviceToHost, stream));
  c->d2h_pending.push_back({dst, staged, bytes, stream});
  return TRGT_OK;
}
// waits for `stream`, then delivers the downloads d2h() staged on it
inline hipError_t stream_wait(trgt_hip_ctx* c, hipStream_t stream) {
  const hipError_t e = stream_wait(stream);
  if (c->d2h_pending.empty()) return e;
  size_t kept = 0;
  for (auto& q : c->d2h_pending) {
    if (q.stream == stream) { if (e == hipSuccess) std::memcpy(q.dst, q.staged, q.bytes); }
    else c->d2h_pending[kept++] = q;
  }
  c->d2h_pending.resize(kept);
  if (kept == 0) {  // nothing in flight: the chunks are free again (one chunk of the total size replaces several)
    size_t total = 0;
    for (auto& k : c->d2h_chunks) { total += k.used; k.used = 0; }
    if (c->d2h_chunks.size() > 1) {
      for (auto& k : c->d2h_chunks) (void)hipHostFree(k.p);
      c->d2h_chunks.clear();
      trgt_hip_ctx::D2hChunk k; k.cap = total + total / 4 + 4096;
      if (hipHostMalloc(&k.p, k.cap, hipHostMallocDefault) == hipSuccess) c->d2h_chunks.push_back(k); else (void)hipGetLastError();
    }
  }
  return e;
}

// dst (device) <- src (host) on `stream`.  stage_slot >= 0: src may be pageable, it is copied into that slot's pinned staging first (the
// caller may then reuse src at once); < 0: src is pinned and stays untouched until the stream has passed the copy.
inline int h2d_small(trgt_hip_ctx* c, void* dst, const void* src, size_t bytes, hipStream_t stream, int stage_slot) {
  if (bytes == 0) return TRGT_OK;
  static const size_t kmax = [] { const char* e = TRGT_DEV_ENV("TRGT_H2D_KERNEL_MAX"); return e && *e ? (size_t)atoll(e) : H2D_KERNEL_MAX; }();
  if (bytes > kmax) { TRGT_HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream)); return TRGT_OK; }
  const void* from = src;
  if (stage_slot >= 0 && !is_pinned_host_ptr(src)) {
    if ((int)c->h2d_stage.size() < S_COUNT) c->h2d_stage.resize(S_COUNT);
    auto& b = c->h2d_stage[(size_t)stage_slot];
    if (b.cap < bytes) {
      if (b.p) { TRGT_HIP_TRY(c, hipHostFree(b.p)); b.p = nullptr; b.cap = 0; }
      const size_t want = bytes + bytes / 4 + 4096;
      if (hipHostMalloc(&b.p, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); b.p = nullptr; return fail(c, TRGT_ERR_NOMEM, "hipHostMalloc(%zu bytes) failed", want); }
      b.cap = want;
    }
    std::memcpy(b.p, src, bytes);
    from = b.p;
  }
  const unsigned blocks = (unsigned)std::min<size_t>(256, (bytes / 16 + 255) / 256 + 1);
  hipLaunchKernelGGL(h2d_copy_kernel, dim3(blocks), dim3(256), 0, stream, (uint8_t*)dst, (const uint8_t*)from, bytes);
  TRGT_HIP_TRY(c, hipGetLastError());
  return TRGT_OK;
}

// Input that may live on the host or on the device.  dev() returns a device pointer valid until the next
// use of the same slot (uploading on the ctx stream when the caller's pointer is a host pointer).
template <typename T>
inline int dev_in(trgt_hip_ctx* c, int slot, const T* p, size_t count, const T** out) {
  if (is_device_ptr(p)) { *out = p; return TRGT_OK; }
  void* d = nullptr;
  int rc = dev_get(c, slot, count * sizeof(T), &d);
  if (rc) return rc;
  if (count && (rc = h2d_small(c, d, p, count * sizeof(T), c->stream, slot))) return rc;
  *out = (const T*)d;
  return TRGT_OK;
}

// Several small uploads / fills as ONE dispatch (every dispatch of a call costs 5-30 us of queue time, and a dozen offset tables in front
// of stage A were 0.35 ms of its critical path): segments are collected with add() / add_fill() and flush() launches one kernel over
// all of them (blockIdx.y = segment).  Host sources that are not pinned are staged in the slot's pinned buffer first (stage()).
struct MultiSeg { void* dst; const void* src; size_t bytes; uint32_t fill; };  // src == nullptr: fill with the byte `fill`
constexpr int MULTI_SEG_MAX = 20;
struct MultiSegs { MultiSeg s[MULTI_SEG_MAX]; int n; };
static __global__ void __launch_bounds__(256) multi_copy_kernel(const MultiSegs m) {
  const MultiSeg g = m.s[blockIdx.y];
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, n = (size_t)gridDim.x * blockDim.x;
  uint8_t* __restrict__ dst = (uint8_t*)g.dst;
  if (g.src == nullptr) {
    const uint32_t f4 = g.fill * 0x01010101u;
    if (((uintptr_t)dst & 15u) == 0) {
      const size_t n16 = g.bytes >> 4;
      for (size_t k = i; k < n16; k += n) reinterpret_cast<uint4*>(dst)[k] = make_uint4(f4, f4, f4, f4);
      for (size_t k = (n16 << 4) + i; k < g.bytes; k += n) dst[k] = (uint8_t)g.fill;
    } else for (size_t k = i; k < g.bytes; k += n) dst[k] = (uint8_t)g.fill;
    return;
  }
  const uint8_t* __restrict__ src = (const uint8_t*)g.src;
  if ((((uintptr_t)dst | (uintptr_t)src) & 15u) == 0) {
    const size_t n16 = g.bytes >> 4;
    for (size_t k = i; k < n16; k += n) reinterpret_cast<uint4*>(dst)[k] = reinterpret_cast<const uint4*>(src)[k];
    for (size_t k = (n16 << 4) + i; k < g.bytes; k += n) dst[k] = src[k];
  } else for (size_t k = i; k < g.bytes; k += n) dst[k] = src[k];
}
struct UploadBatch {
  trgt_hip_ctx* c; hipStream_t stream; MultiSegs m;
  struct Staging { void* pinned; const void* src; size_t bytes; };
  std::vector<Staging> staging;  // host memcpys still to do (run_staging: the caller may spread them over its threads)
  std::vector<int> used_slots;   // staging slots holding a source of this batch
  UploadBatch(trgt_hip_ctx* c_, hipStream_t s) : c(c_), stream(s) { m.n = 0; }
  int flush() {
    run_staging();
    used_slots.clear();
    if (m.n == 0) return TRGT_OK;
    size_t most = 0;
    for (int i = 0; i < m.n; ++i) most = std::max(most, m.s[i].bytes);
    const unsigned bx = (unsigned)std::min<size_t>(128, most / (16 * 256) + 1);
    hipLaunchKernelGGL(multi_copy_kernel, dim3(bx, (unsigned)m.n), dim3(256), 0, stream, m);
    m.n = 0;
    TRGT_HIP_TRY(c, hipGetLastError());
    return TRGT_OK;
  }
  void run_staging() { for (auto& g : staging) std::memcpy(g.pinned, g.src, g.bytes); staging.clear(); }
  int add_fill(void* dst, size_t bytes, uint8_t value) {
    if (bytes == 0) return TRGT_OK;
    if (m.n == MULTI_SEG_MAX) { const int rc = flush(); if (rc) return rc; }
    m.s[m.n++] = MultiSeg{dst, nullptr, bytes, value};
    return TRGT_OK;
  }
  // src: host memory (pinned, or pageable with stage_slot >= 0: copied into that slot's pinned staging by run_staging / flush)
  int add(void* dst, const void* src, size_t bytes, int stage_slot) {
    if (bytes == 0) return TRGT_OK;
    if (bytes > H2D_KERNEL_MAX) { TRGT_HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream)); return TRGT_OK; }
    const void* from = src;
    if (stage_slot >= 0 && !is_pinned_host_ptr(src)) {
      // (a slot's pinned staging holds ONE source until the batch is flushed: a second use of the slot flushes first)
      for (const auto& u : used_slots) if (u == stage_slot) { const int rc = flush(); if (rc) return rc; break; }
      used_slots.push_back(stage_slot);
      if ((int)c->h2d_stage.size() < S_COUNT) c->h2d_stage.resize(S_COUNT);
      auto& b = c->h2d_stage[(size_t)stage_slot];
      if (b.cap < bytes) {
        if (b.p) { TRGT_HIP_TRY(c, hipHostFree(b.p)); b.p = nullptr; b.cap = 0; }
        const size_t want = bytes + bytes / 4 + 4096;
        if (hipHostMalloc(&b.p, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); b.p = nullptr; return fail(c, TRGT_ERR_NOMEM, "hipHostMalloc(%zu bytes) failed", want); }
        b.cap = want;
      }
      staging.push_back({b.p, src, bytes});
      from = b.p;
    }
    if (m.n == MULTI_SEG_MAX) { const int rc = flush(); if (rc) return rc; }
    m.s[m.n++] = MultiSeg{dst, from, bytes, 0};
    return TRGT_OK;
  }
};
// dev_in whose upload joins a batch (valid once the batch is flushed)
template <typename T>
inline int dev_in(trgt_hip_ctx* c, int slot, const T* p, size_t count, const T** out, UploadBatch* ub) {
  if (is_device_ptr(p)) { *out = p; return TRGT_OK; }
  void* d = nullptr;
  int rc = dev_get(c, slot, count * sizeof(T), &d);
  if (rc) return rc;
  if (count && (rc = ub->add(d, p, count * sizeof(T), slot))) return rc;
  *out = (const T*)d;
  return TRGT_OK;
}

// Output that may live on the host or on the device.
template <typename T>
struct DevOut {
  T* user = nullptr; T* dev = nullptr; size_t count = 0; bool staged = false;
  int init(trgt_hip_ctx* c, int slot, T* p, size_t n) {
    user = p; count = n; staged = false; dev = nullptr;
    if (p == nullptr) return TRGT_OK;
    if (is_device_ptr(p)) { dev = p; return TRGT_OK; }
    void* d = nullptr;
    int rc = dev_get(c, slot, n * sizeof(T), &d);
    if (rc) return rc;
    dev = (T*)d; staged = true;
    return TRGT_OK;
  }
  int finish(trgt_hip_ctx* c) {
    if (staged && count) return d2h(c, user, dev, count * sizeof(T), c->stream);
    return TRGT_OK;
  }
};

// ---- kernel timing: events on the ctx stream, resolved lazily ---------------------------------
struct KTimer {
  trgt_hip_ctx* c; int k; hipEvent_t a = nullptr, b = nullptr; bool on; hipStream_t s;
  KTimer(trgt_hip_ctx* c_, int k_, hipStream_t stream = nullptr) : c(c_), k(k_), on(c_->timing), s(stream ? stream : c_->stream) {
    if (on) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, s); }
  }
  void stop(int64_t cells) {
    if (on) { (void)hipEventRecord(b, s); c->pending.push_back({k, a, b}); c->k_launches[k] += 1; c->k_cells[k] += cells; }
  }
};
// streams of a context (ctx.hip)
hipError_t make_stream(trgt_hip_ctx* c, hipStream_t* s);
hipError_t make_side_stream(trgt_hip_ctx* c, hipStream_t* s);
void ctx_next_stream_priority(int p);
void ctx_next_in_pool(bool on);

inline void resolve_timing(trgt_hip_ctx* c) {
  for (auto& p : c->pending) {
    float ms = 0;
    (void)trgt::event_wait(p.b);
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) c->k_ms[p.k] += ms;
    // destroying (or re-recording) events here makes one of the next few calls 3-7 ms slower now and then (measured, ROCm 7.2): they
    // are parked and destroyed with the ctx, or in bulk once there are very many
    c->retired_events.push_back(p.a); c->retired_events.push_back(p.b);
  }
  c->pending.clear();
  if (c->retired_events.size() > (1u << 18)) { for (hipEvent_t e : c->retired_events) (void)hipEventDestroy(e); c->retired_events.clear(); }
}

}  // namespace trgt

// entry points implemented in other translation units
namespace trgt {
struct HmmHostModel;
}
