// trgt_amd/csrc/ctx.hip -- ctx lifetime, stream binding, timing accessors.
#include "common.hpp"
#include "host_pool.hpp"

static std::string g_create_err;

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
  trgt_hip_ctx* c = new trgt_hip_ctx();
  {
    trgt_knobs& k = c->knobs;
    auto num = [](const char* name, int dflt) { const char* e = getenv(name); return e && *e ? atoi(e) : dflt; };
    auto flag = [](const char* name) { const char* e = getenv(name); return e != nullptr && *e != 0 && std::strcmp(e, "0") != 0; };
    k.flank_threads = num("TRGT_FLANK_THREADS", k.flank_threads); k.heavy_threads = num("TRGT_HEAVY_THREADS", 0);
    k.win_threads = num("TRGT_WIN_THREADS", k.win_threads); k.win_segments = num("TRGT_WIN_SEGMENTS", k.win_segments);
    k.grid_per_cu = num("TRGT_WFA_GRID_PER_CU", 0); k.filter_per_cu = num("TRGT_FILTER_PER_CU", 0);
    k.one_launch = flag("TRGT_WFA_ONE_LAUNCH"); k.no_spec = flag("TRGT_WFA_NO_SPEC"); k.no_window = flag("TRGT_WFA_NO_WINDOW");
    k.no_filter = flag("TRGT_WFA_NO_FILTER"); k.one_stream = flag("TRGT_FLANK_ONE_STREAM"); k.host_genotyper = flag("TRGT_HOST_GENOTYPER"); k.host_hmm_lists = flag("TRGT_HOST_HMM_LISTS"); k.no_early = flag("TRGT_WFA_NO_EARLY"); k.stage_lock = flag("TRGT_STAGE_LOCK"); k.debug = flag("TRGT_WFA_DEBUG");
    k.timeline = flag("TRGT_TIMELINE");
#ifdef TRGT_DEV_BUILD
    k.skip_bt = flag("TRGT_DBG_SKIP_BT");
#endif
  }
  c->device = device;
  c->num_cus = prop.multiProcessorCount;
  e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) { g_create_err = hipGetErrorString(e); delete c; return TRGT_ERR_HIP; }
  c->own_stream = true;
  c->pool.resize(trgt::S_COUNT);
  *out = c;
  return TRGT_OK;
}

void trgt_hip_destroy(trgt_hip_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->stream2) { (void)hipStreamSynchronize(c->stream2); (void)hipStreamDestroy(c->stream2); }
  if (c->ev_upload) (void)hipEventDestroy(c->ev_upload);
  if (c->stream_copy) { (void)hipStreamSynchronize(c->stream_copy); (void)hipStreamDestroy(c->stream_copy); }
  for (auto& st : c->staged) if (st.ready) (void)hipEventDestroy(st.ready);
  for (int i = 0; i < 3; ++i) {
    if (c->hmm_side[i]) { (void)hipStreamSynchronize(c->hmm_side[i]); (void)hipStreamDestroy(c->hmm_side[i]); }
    if (c->hmm_join[i]) (void)hipEventDestroy(c->hmm_join[i]);
  }
  if (c->hmm_fork) (void)hipEventDestroy(c->hmm_fork);
  if (c->ev_scan) (void)hipEventDestroy(c->ev_scan);
  if (c->ev_heavy) (void)hipEventDestroy(c->ev_heavy);
  delete static_cast<trgt::HostPool*>(c->host_pool);
  for (auto& b : c->h2d_stage) if (b.p) (void)hipHostFree(b.p);
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
  TRGT_HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  if (s == nullptr) {
    TRGT_HIP_TRY(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
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
