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
  uint32_t n_lanes, pad_;  // lanes of a workgroup when the states are not in lane order (models of more than one wave), else 0
};

struct HmmModels {
  std::vector<HmmSetDev> sets; std::vector<uint8_t> blob; int rc = 0; std::string err;
  const void* d_sets = nullptr; const void* d_blob = nullptr;  // optional device copies made ahead of time by the caller
};

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
void hmm_pending_free(HmmPending* pending);

}  // namespace trgt
