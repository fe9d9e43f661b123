// trgt_amd/csrc/hmm_host.hpp -- host-side model container of the motif HMM, shared by hmm.hip and locus.hip.
#pragma once
#include <string>
#include <vector>

#include "common.hpp"

namespace trgt {

struct HmmSetDev {  // device-visible descriptor of one motif set (one locus)
  uint32_t S, n_blocks, chain_rounds, max_mlen;  // chain_rounds: waves the longest deletion chain spans
  uint64_t off_inlp;    // f64 [4][S]   ln transition probabilities, predecessor-list order of the reference
  uint64_t off_em;      // f64 [5][S]   ln emissions over # A T C G
  uint64_t off_inst;    // u16 [4][S]   predecessor states
  uint64_t off_block;   // i16 [S]      motif block of the state (-1 outside)
  uint64_t off_nin;     // u8  [S]      #predecessors (0xFF: run-end state, predecessors = block ends)
  uint64_t off_level;   // u8  [S]      0 emitting, >=1 silent evaluation level
  uint64_t off_flags;   // u8  [S]      bit0 any finite emission, bit1 emits a base
  uint64_t off_blocks;  // u32 [4][n_blocks]  start,end,mlen,motif byte offset
  uint64_t off_motifs;  // sanitised motif bytes
  uint64_t off_perm;    // u16 [n_lanes] lane -> state (0xFFFF: idle lane); only when n_lanes != 0
  uint32_t n_lanes;     // lanes of a workgroup when the states are not in lane order (models of more than one wave), else 0
  uint32_t ppl_lanes;   // motif positions + 1 (the skip block's) when they fit one wave: the set's Viterbi fill runs with one lane per position (hmm_ppl.hpp); else 0
};

struct HmmModels {
  std::vector<HmmSetDev> sets; std::vector<uint8_t> blob; int rc = 0; std::string err;
  const void* d_sets = nullptr; const void* d_blob = nullptr;  // device copies (hmm_models_on_device builds the blob there: `blob` stays empty)
  uint64_t blob_bytes = 0;
};

// built on the device; uploads and kernel on `up`, `done` (optional) recorded behind them
int hmm_models_on_device(trgt_hip_ctx* c, int32_t n_sets, const uint8_t* motif_blob, const uint32_t* motif_off, const uint32_t* set_motif_begin,
                         HmmModels& out, hipStream_t up, hipEvent_t done);
// host builder (what trgt_hmm_models_check compares the device builder with)
int hmm_build_models(int32_t n_sets, const uint8_t* motif_blob, const uint32_t* motif_off, const uint32_t* set_motif_begin, HmmModels& out);

int hmm_batch_impl(trgt_hip_ctx* c, const HmmModels* premade, int32_t n_sets, const uint8_t* motif_blob, const uint32_t* motif_off,
                   const uint32_t* set_motif_begin, int64_t n_jobs, const uint32_t* job_set, const uint8_t* seq_blob,
                   const uint64_t* seq_off, const uint32_t* seq_len, uint16_t* path, const uint64_t* path_off, uint32_t* path_len,
                   int32_t* spans3, const uint64_t* span_off, uint32_t* n_spans, uint32_t* motif_counts, const uint64_t* count_off,
                   double* purity, int32_t* edit_dist, int32_t* max_dist);

// Asynchronous form: hmm_enqueue uploads and launches on the ctx stream without waiting for the kernels; hmm_collect waits (on the
// stream the batch was enqueued on) and copies the results out (and frees the pending state).  *pending stays null for an empty
// batch.  buffer_set 0 / 1: which set of device / pinned scratch buffers to use -- two batches may be in flight on two streams.
struct HmmPending;
int hmm_enqueue(trgt_hip_ctx* c, const HmmModels* premade, int32_t n_sets, const uint8_t* motif_blob, const uint32_t* motif_off,
                const uint32_t* set_motif_begin, int64_t n_jobs, const uint32_t* job_set, const uint8_t* seq_blob,
                const uint64_t* seq_off, const uint32_t* seq_len, uint16_t* path, const uint64_t* path_off, uint32_t* path_len,
                int32_t* spans3, const uint64_t* span_off, uint32_t* n_spans, uint32_t* motif_counts, const uint64_t* count_off,
                double* purity, int32_t* edit_dist, int32_t* max_dist, HmmPending** pending, int buffer_set = 0);
int hmm_collect(trgt_hip_ctx* c, HmmPending* pending);
// Stage C enqueued BEHIND a device-side genotyper, before the host has its results: two candidate jobs per locus (allele slots 2 l,
// 2 l + 1), resolved into the job list on the device from the genotyper's outputs.  Results land at the slot's index of the caller's
// arrays (n_spans, purity: 2 n_loci entries; spans3 / motif_counts by span_off / count_off of the slot).
struct HmmSlots {
  int64_t n_loci = 0;
  const uint8_t* host_skip = nullptr;      // host [n_loci], optional: 1 = the locus never gets a job here (it takes the host path)
  const uint32_t* cap = nullptr;           // host [n_loci]   longest allele the locus can have (workspace is set aside for it)
  const uint64_t* seq_off = nullptr;       // host [2 n_loci] where the allele of a slot will be in seq_blob_dev
  const uint8_t* seq_blob_dev = nullptr;   // device
  const uint8_t* d_skip = nullptr; const int32_t* d_n_alleles = nullptr; const uint32_t* d_allele_len = nullptr;  // device: genotyper results
};
int hmm_enqueue_slots(trgt_hip_ctx* c, const HmmModels* models, const HmmSlots& slots, int32_t* spans3, const uint64_t* span_off,
                      uint32_t* n_spans, uint32_t* motif_counts, const uint64_t* count_off, double* purity, HmmPending** pending,
                      int buffer_set = 0);
// ... and once the host has the genotyper's results (the same arrays, host copies): which slots were jobs.  Returns their number.
int64_t hmm_slots_resolved(trgt_hip_ctx* c, HmmPending* pending, const HmmModels* models, const uint8_t* skip, const int32_t* n_alleles,
                           const uint32_t* allele_len);
void hmm_pending_free(HmmPending* pending);

}  // namespace trgt
