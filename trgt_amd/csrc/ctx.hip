// trgt_amd/csrc/ctx.hip -- ctx lifetime, stream binding, timing accessors.
#include "common.hpp"
#include "host_pool.hpp"

static std::string g_create_err;

// glibc hands blocks above its mmap threshold (128 KB at first) straight from mmap and returns them with munmap, and trims the top of
// the heap when enough of it is free.  Every such unmap runs the MMU notifiers of the process, the GPU driver's among them, and on this
// stack (ROCm 7.2, kernel 6.18, HZ=100) the work submitted to the GPU next does not start before the driver's restore timer fires,
// one to three scheduler ticks of 10 ms later: config 5 spent 50 ms per call instead of 24 because of the per-call scratch vectors
// of the host stages (and numpy temporaries of the caller do the same).  So the first context of a process tells malloc to keep
// what it has: blocks below 32 MB come from the heap, the heap is not trimmed.  TRGT_MALLOC_TUNE=0 leaves malloc alone,
// =2 also keeps the blocks above 32 MB on the heap (M_MMAP_MAX 0).
#include <malloc.h>
static void tune_malloc_once() {
  static std::once_flag once;
  std::call_once(once, [] {
    const char* e = getenv("TRGT_MALLOC_TUNE");
    const int mode = e && *e ? atoi(e) : 1;
    if (mode <= 0) return;
    (void)mallopt(M_MMAP_THRESHOLD, 32 << 20);
    (void)mallopt(M_TRIM_THRESHOLD, 1 << 30);
    (void)mallopt(M_TOP_PAD, 64 << 20);
    if (mode >= 2) (void)mallopt(M_MMAP_MAX, 0);
  });
}

namespace trgt {
// Streams of a context are created where they are first needed, all with the context's priority.  A pool gives its contexts
// different priorities (trgt_hip_pool_create): the runtime keeps a set of hardware queues per priority, so the contexts of a pool
// share fewer queues, and when their kernels meet on the GPU the more urgent one finishes first instead of all of them finishing
// together (config 2: 1.74 -> 1.82 M loci/s with four contexts, config 4: 0.88 -> 0.92 M).
static thread_local int g_next_stream_priority = 0;
static thread_local bool g_next_in_pool = false;
void ctx_next_stream_priority(int p) { g_next_stream_priority = p; }
void ctx_next_in_pool(bool on) { g_next_in_pool = on; }
hipError_t make_stream(trgt_hip_ctx* c, hipStream_t* s) {
  return c->stream_priority ? hipStreamCreateWithPriority(s, hipStreamNonBlocking, c->stream_priority) : hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}
// The side streams of the HMM launches take a priority of their own: the runtime maps the streams of a process onto a few hardware
// queues PER PRIORITY, round robin in creation order, and streams on one queue run one after the other.  With one priority for all
// nine streams of a context a side stream lands on the queue of the main stream -- and the two longest HMM launches of a config-3
// call, made to run next to each other, ran one behind the other (44 ms per call in a fresh context, 30 ms in one whose streams
// happened to fall differently).
hipError_t make_side_stream(trgt_hip_ctx* c, hipStream_t* s) {
  // (the contexts of a pool already sit on different priorities, context by context, and measured better with all streams of a context
  //  on its own: 8.3 against 7.1 k loci/s on config 3, 1.11 against 1.05 M on config 4)
  if (c->in_pool) return make_stream(c, s);
  int lo = 0, hi = 0;  // (lo = least urgent, numerically larger)
  if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess || lo <= hi) { (void)hipGetLastError(); return make_stream(c, s); }
  const int base = c->stream_priority;  // 0 = default
  const int side = base != hi ? hi : hi + 1;
  return hipStreamCreateWithPriority(s, hipStreamNonBlocking, side);
}
}  // namespace trgt

extern "C" {

int trgt_hip_abi_version(void) { return TRGT_HIP_ABI_VERSION; }

int trgt_hip_create(int device, trgt_hip_ctx** out) {
  if (!out) return TRGT_ERR_INVALID;
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    g_create_err = "no HIP device visible (libtrgt_hip has no CPU fallback)";
    return TRGT_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= n) { g_create_err = "device ordinal out of range"; return TRGT_ERR_INVALID; }
  e = hipSetDevice(device);
  if (e != hipSuccess) { g_create_err = hipGetErrorString(e); return TRGT_ERR_HIP; }
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) { g_create_err = hipGetErrorString(e); return TRGT_ERR_HIP; }
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    g_create_err = std::string("device is ") + prop.gcnArchName + ", this library carries gfx950 code only";
    return TRGT_ERR_NO_DEVICE;
  }
  tune_malloc_once();
  trgt_hip_ctx* c = new trgt_hip_ctx();
  {
    trgt_knobs& k = c->knobs;
    auto num = [](const char* name, int dflt) { const char* e = getenv(name); return e && *e ? atoi(e) : dflt; };
    auto flag = [](const char* name) { const char* e = getenv(name); return e != nullptr && *e != 0 && std::strcmp(e, "0") != 0; };
    // DEV_FLAG / DEV_NUM: switches whose A/B is settled (tools/README.md) -- read in `make DEV=1` builds only; the release library keeps
    // their code paths at the defaults and does not carry their names (tests/test_abi_exports.py counts what it carries)
#ifdef TRGT_DEV_BUILD
#define DEV_FLAG(name) flag(name)
#define DEV_NUM(name, dflt) num(name, dflt)
#else
#define DEV_FLAG(name) false
#define DEV_NUM(name, dflt) (dflt)
#endif
    (void)num; (void)flag;
    k.flank_threads = DEV_NUM("TRGT_FLANK_THREADS", k.flank_threads); k.heavy_threads = DEV_NUM("TRGT_HEAVY_THREADS", 0); k.heavy_band = std::max(0, std::min(num("TRGT_HEAVY_BAND", k.heavy_band), 256));
    k.win_threads = DEV_NUM("TRGT_WIN_THREADS", k.win_threads); { const int bt = DEV_NUM("TRGT_BAND_THREADS", k.band_threads); k.band_threads = bt == 128 || bt == 256 ? bt : 64; } k.win_segments = DEV_NUM("TRGT_WIN_SEGMENTS", k.win_segments);
    k.grid_per_cu = DEV_NUM("TRGT_WFA_GRID_PER_CU", 0); k.filter_per_cu = DEV_NUM("TRGT_FILTER_PER_CU", 0);
    k.one_launch = DEV_FLAG("TRGT_WFA_ONE_LAUNCH"); k.no_spec = DEV_FLAG("TRGT_WFA_NO_SPEC"); k.no_window = flag("TRGT_WFA_NO_WINDOW"); k.no_hamming = flag("TRGT_NO_HAMMING"); k.no_indel_shortcut = flag("TRGT_NO_INDEL_SHORTCUT"); k.no_heavy_window = DEV_FLAG("TRGT_NO_HEAVY_WINDOW"); { const char* e = TRGT_DEV_ENV("TRGT_EARLY_ADAPTIVE"); if (e && *e) k.early_adaptive = std::strcmp(e, "0") != 0; }
    k.no_filter = flag("TRGT_WFA_NO_FILTER"); k.one_stream = DEV_FLAG("TRGT_FLANK_ONE_STREAM"); k.host_genotyper = flag("TRGT_HOST_GENOTYPER"); k.host_hmm_lists = DEV_FLAG("TRGT_HOST_HMM_LISTS"); k.no_early = flag("TRGT_WFA_NO_EARLY"); k.stage_lock = flag("TRGT_STAGE_LOCK"); k.no_long_filter = flag("TRGT_NO_LONG_FILTER"); k.no_long_window = DEV_FLAG("TRGT_NO_LONG_WINDOW"); k.filter_force = DEV_NUM("TRGT_FILTER_FORCE", 0); k.filter_one_launch = DEV_FLAG("TRGT_FILTER_ONE_LAUNCH"); k.filter_serial = DEV_FLAG("TRGT_FILTER_SERIAL"); k.filter_side = DEV_FLAG("TRGT_FILTER_SIDE"); k.wfa_no_stage = DEV_FLAG("TRGT_WFA_NO_STAGE"); k.wfa_no_wave_variant = DEV_FLAG("TRGT_WFA_NO_WAVE_VARIANT"); k.hmm_resolve_one_wg = DEV_FLAG("TRGT_HMM_RESOLVE_ONE_WG"); k.debug = flag("TRGT_WFA_DEBUG");
    k.timeline = flag("TRGT_TIMELINE");
    k.host_repair = flag("TRGT_HOST_REPAIR"); k.host_cluster = flag("TRGT_HOST_CLUSTER"); k.cluster_arena_kb = num("TRGT_CLUSTER_ARENA_KB", 0); k.hmm_lds_fill = DEV_FLAG("TRGT_HMM_LDS_FILL"); k.hmm_four_rounds = DEV_FLAG("TRGT_HMM_FOUR_ROUNDS"); k.hmm_no_ppl = flag("TRGT_HMM_NO_PPL"); k.hmm_long_wgs = std::max(1, std::min(DEV_NUM("TRGT_HMM_LONG_WGS", k.hmm_long_wgs), 16)); k.hmm_no_long_tb = flag("TRGT_HMM_NO_LONG_TB"); k.hmm_no_dedupe = flag("TRGT_HMM_NO_DEDUPE"); k.repair_max_seg = num("TRGT_REPAIR_MAX_SEG", k.repair_max_seg); k.split_hmm = DEV_FLAG("TRGT_SPLIT_HMM"); k.repair_blocks = DEV_NUM("TRGT_REPAIR_BLOCKS", k.repair_blocks);
    k.no_lean = flag("TRGT_WFA_NO_LEAN"); k.lean_one_tier = DEV_FLAG("TRGT_WFA_LEAN_ONE_TIER"); k.lean_mid_tier = DEV_FLAG("TRGT_WFA_LEAN_MID_TIER"); k.lean_chunk = DEV_NUM("TRGT_LEAN_CHUNK", 0); k.no_zero_arena = flag("TRGT_NO_ZERO_ARENA"); k.hmm_ppl_serial = DEV_FLAG("TRGT_HMM_PPL_SERIAL"); k.hmm_ppl_wide = DEV_FLAG("TRGT_HMM_PPL_WIDE"); k.hmm_ppl_per_class = DEV_FLAG("TRGT_HMM_PPL_PER_CLASS");
    k.no_lds_wfa = !DEV_FLAG("TRGT_WFA_LDS"); k.lds_wfa_kb = DEV_NUM("TRGT_WFA_LDS_KB", k.lds_wfa_kb); k.lds_wfa_seq = DEV_NUM("TRGT_WFA_LDS_SEQ", k.lds_wfa_seq);
#ifdef TRGT_DEV_BUILD
    // switches that CHANGE results exist only in `make DEV=1` builds (tools/unpinned_sensitivity.py builds one for itself)
    k.sens_bialign_min_len = num("TRGT_SENS_BIALIGN_MIN_LEN", -1); k.sens_cons_unidir = flag("TRGT_SENS_CONS_UNIDIR"); k.sens_ward_ties = flag("TRGT_SENS_WARD_TIES"); k.sens_lw_order = flag("TRGT_SENS_LW_ORDER");
    k.skip_bt = flag("TRGT_DBG_SKIP_BT");
#endif
  }
  c->device = device;
  c->num_cus = prop.multiProcessorCount;
  c->stream_priority = trgt::g_next_stream_priority;
  c->in_pool = trgt::g_next_in_pool;
  e = trgt::make_stream(c, &c->stream);
  if (e != hipSuccess) { g_create_err = hipGetErrorString(e); delete c; return TRGT_ERR_HIP; }
  c->own_stream = true;
  c->pool.resize(trgt::S_COUNT);
  *out = c;
  return TRGT_OK;
}

void trgt_hip_destroy(trgt_hip_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)trgt::stream_wait(c, c->stream);
  if (c->stream2) { (void)trgt::stream_wait(c, c->stream2); (void)hipStreamDestroy(c->stream2); }
  if (c->ev_upload) (void)hipEventDestroy(c->ev_upload);
  if (c->stream_copy) { (void)trgt::stream_wait(c, c->stream_copy); (void)hipStreamDestroy(c->stream_copy); }
  for (auto& st : c->staged) if (st.ready) (void)hipEventDestroy(st.ready);
  for (int i = 0; i < 6; ++i) {
    if (c->hmm_side[i]) { (void)trgt::stream_wait(c, c->hmm_side[i]); (void)hipStreamDestroy(c->hmm_side[i]); }
    if (c->hmm_join[i]) (void)hipEventDestroy(c->hmm_join[i]);
  }
  for (int i = 0; i < 2; ++i) if (c->hmm_fork[i]) (void)hipEventDestroy(c->hmm_fork[i]);
  for (int b = 0; b < 2; ++b)
    for (int k = 0; k < 4; ++k) {
      for (int g = 0; g < 3; ++g) {
        if (c->hmm_ppl_side[b][k][g]) { (void)trgt::stream_wait(c, c->hmm_ppl_side[b][k][g]); (void)hipStreamDestroy(c->hmm_ppl_side[b][k][g]); }
        if (c->hmm_ppl_join[b][k][g]) (void)hipEventDestroy(c->hmm_ppl_join[b][k][g]);
      }
      if (c->hmm_ppl_fork[b][k]) (void)hipEventDestroy(c->hmm_ppl_fork[b][k]);
    }
  if (c->ev_scan) (void)hipEventDestroy(c->ev_scan);
  if (c->ev_heavy) (void)hipEventDestroy(c->ev_heavy);
  if (c->ev_hwin) (void)hipEventDestroy(c->ev_hwin);
  if (c->stream_hmm) (void)hipStreamDestroy(c->stream_hmm);
  if (c->stream_flt) (void)hipStreamDestroy(c->stream_flt);
  if (c->ev_flt_a) (void)hipEventDestroy(c->ev_flt_a);
  if (c->ev_flt_b) (void)hipEventDestroy(c->ev_flt_b);
  if (c->ev_gt) (void)hipEventDestroy(c->ev_gt);
  if (c->ev_rp) (void)hipEventDestroy(c->ev_rp);
  delete static_cast<trgt::HostPool*>(c->host_pool);
  for (auto& b : c->h2d_stage) if (b.p) (void)hipHostFree(b.p);
  for (auto& k : c->d2h_chunks) if (k.p) (void)hipHostFree(k.p);
  for (auto& b : c->pinned)
    if (b.p) (void)hipHostFree(b.p);
  trgt::resolve_timing(c);
  for (auto& b : c->pool)
    if (b.p) (void)hipFree(b.p);
  trgt::resolve_timing(c);
  for (hipEvent_t e : c->retired_events) (void)hipEventDestroy(e);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char* trgt_hip_last_error(const trgt_hip_ctx* c) { return c ? c->err.c_str() : g_create_err.c_str(); }

int trgt_hip_set_stream(trgt_hip_ctx* c, void* s) {
  if (!c) return TRGT_ERR_INVALID;
  (void)hipSetDevice(c->device);
  TRGT_HIP_TRY(c, trgt::stream_wait(c, c->stream));
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  if (s == nullptr) {
    TRGT_HIP_TRY(c, trgt::make_stream(c, &c->stream));
    c->own_stream = true;
  } else {
    c->stream = (hipStream_t)s;
    c->own_stream = false;
  }
  return TRGT_OK;
}

int trgt_hip_set_workspace_limit(trgt_hip_ctx* c, uint64_t bytes) {
  if (!c) return TRGT_ERR_INVALID;
  c->ws_limit = bytes ? bytes : (32ull << 30);
  return TRGT_OK;
}

int trgt_hip_timing_enable(trgt_hip_ctx* c, int on) {
  if (!c) return TRGT_ERR_INVALID;
  c->timing = on != 0;
  return TRGT_OK;
}
int trgt_hip_timing_reset(trgt_hip_ctx* c) {
  if (!c) return TRGT_ERR_INVALID;
  trgt::resolve_timing(c);
  for (int k = 0; k < TRGT_K_COUNT; ++k) { c->k_ms[k] = 0; c->k_launches[k] = 0; c->k_cells[k] = 0; }
  return TRGT_OK;
}
int trgt_hip_timing_get(trgt_hip_ctx* c, int k, double* ms, int64_t* launches, int64_t* cells) {
  if (!c || k < 0 || k >= TRGT_K_COUNT) return TRGT_ERR_INVALID;
  trgt::resolve_timing(c);
  if (ms) *ms = c->k_ms[k];
  if (launches) *launches = c->k_launches[k];
  if (cells) *cells = c->k_cells[k];
  return TRGT_OK;
}

void trgt_wfa_default_params(trgt_wfa_params* p) {
  std::memset(p, 0, sizeof(*p));
  p->metric = 3; p->mismatch = 4; p->gap_open1 = 6; p->gap_ext1 = 2; p->gap_open2 = 24; p->gap_ext2 = 1;
  p->span = 0; p->scope = 1; p->memory_mode = 0;
  p->heuristic = 1; p->h_min_wavefront_length = 10; p->h_max_distance_threshold = 50; p->h_steps_between_cutoffs = 1;
  p->bialign_min_score = 250; p->bialign_min_length = 100;
}

}  // extern "C"
