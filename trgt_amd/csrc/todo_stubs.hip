// TEMPORARY: entry points not implemented yet fail loudly (no fallback).  Removed as each lands.
#include "common.hpp"
extern "C" {
int trgt_find_spans_batch(trgt_hip_ctx* c, const trgt_span_params*, int64_t, const uint8_t*, const uint64_t*, const uint32_t*, const uint64_t*, const uint32_t*, const uint64_t*, const uint8_t*, const uint64_t*, const uint32_t*, int32_t*, int32_t*, uint8_t*, uint8_t*) { return trgt::fail(c, TRGT_ERR_UNSUPPORTED, "trgt_find_spans_batch: not built yet"); }
int trgt_locus_batch(trgt_hip_ctx* c, const trgt_locus_params*, const trgt_locus_batch_in*, trgt_locus_batch_out*) { return trgt::fail(c, TRGT_ERR_UNSUPPORTED, "trgt_locus_batch: not built yet"); }
void trgt_synth_default_params(trgt_synth_params*, int) {}
int trgt_synth_sizes_for(const trgt_synth_params*, int64_t, int64_t, trgt_synth_sizes*) { return TRGT_ERR_UNSUPPORTED; }
int trgt_synth_fill(const trgt_synth_params*, int64_t, int64_t, uint8_t*, uint64_t*, uint32_t*, uint64_t*, uint32_t*, uint8_t*, uint64_t*, uint32_t*, uint8_t*, uint32_t*, uint32_t*, uint8_t*, uint64_t*, uint8_t*, uint64_t*, uint32_t*) { return TRGT_ERR_UNSUPPORTED; }
}
