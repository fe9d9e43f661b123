// trgt_amd/csrc/wfa_host.hpp -- job record and host-side launch descriptor of the WFA kernel (no device code).
#pragma once
#include <functional>
#include <vector>
#include "common.hpp"

namespace trgt {

struct JobDev {  // one alignment: pattern / text inside pat_base / txt_base, output slots
  uint64_t pat_off, txt_off, cigar_off, ops_off;
  uint32_t pat_len, txt_len, out_index, pad;
};


// A read shorter than this cannot contain the repeat of its locus with both flanks around it (the longest read of the locus is
// the yardstick): at least one of its flank alignments ends with a long gap, i.e. runs through many score levels.
inline uint32_t heavy_read_len(uint32_t locus_max_read_len, int flank_len) {
  const uint32_t margin = (uint32_t)flank_len + (uint32_t)flank_len / 5;
  return locus_max_read_len > margin ? locus_max_read_len - margin : 0;
}

struct WfaLaunch {  // everything device-resident
  const JobDev* jobs_dev = nullptr; int64_t n_jobs_host = 0; const uint32_t* n_jobs_dev = nullptr;
  int64_t jobs_bound = 0;  // with n_jobs_dev: the most jobs the list can hold (0: n_jobs_host is that bound, not just a bound on the workgroups)
  // two-ended list: jobs expected to be expensive are appended from the front (count *n_jobs_dev) and are drained first,
  // the others from the back (jobs_dev[jobs_cap - 1 - k], count *n_jobs2_dev)
  const uint32_t* n_jobs2_dev = nullptr; uint32_t jobs_cap = 0;
  const uint8_t* pat_base = nullptr; const uint8_t* txt_base = nullptr;
  int64_t max_plen = 0, max_tlen = 0, max_sum = 0;
  int threads = 0;
  int timer_slot = TRGT_K_WFA;
  int max_score = 0;    // > 0: alignments whose penalty exceeds it are given up (status OOM, score INT32_MIN): with pattern_begin_free = 0 and a
                        // fixed text_begin_free the wavefronts then stay within [-max_score, text_begin_free + max_score] and the LDS ring of the
                        // dedicated kernel is sized (and indexed) for that range instead of the whole text
  int kernel_tag = -1;  // instantiation of the dedicated kernel to launch (-1: 1 for timer_slot == TRGT_K_WFA_FLANK_REST, else 0); names the launch in a trace
  bool keep_cells = false;  // do not reset the wavefront-offset counter (a second launch of the same logical batch)
  uint64_t ws_budget = 0;  // > 0: the generic kernel's workgroups (one workspace each) are limited to this many bytes of workspace -- launches whose jobs
                           // the register-resident kernels take almost entirely, planned for lengths far above what the jobs have
  int buffer_set = 0;  // 0 / 1 / 2: which workspace / counter buffers of the ctx to use (launches may be in flight on two streams; 2 = the device-side consensus repair, whose buffers must not be re-sized while the flank location is still running on sets 0 / 1)
  int32_t* status = nullptr; int32_t* score = nullptr; int32_t* n_match = nullptr; uint32_t* span4 = nullptr;
  uint32_t* cigar = nullptr; uint32_t* cigar_len = nullptr; uint8_t* ops = nullptr; uint32_t* ops_len = nullptr;
  unsigned int* refused = nullptr;  // device word that counts the jobs the kernel refuses (longer than the planned workspace): device-built lists carry one
                                    // that comes back with the call's count block, so that such a job fails the call instead of leaving an INT32_MIN behind
};

// Enqueue the WFA kernel on the ctx stream (asynchronous).  The number of wavefront offsets computed is
// accumulated in device memory at ctx->last_wfa_cells_dev.
int wfa_launch(trgt_hip_ctx* c, const trgt_wfa_params& p, const WfaLaunch& L);

// The register-resident BiWFA kernel for small end-to-end alignments (wfa_lean.hip), launched by wfa_launch in front of the generic
// kernel: alignments it does not take are appended to retry_jobs / *retry_count (device memory) and redone there from scratch.
int wfa_lean_launch(trgt_hip_ctx* c, const trgt_wfa_params& p, const WfaLaunch& L, JobDev* mid_jobs, JobDev* retry_jobs, unsigned int* retry_count, uint32_t retry_cap,
                    unsigned int* retry_lost, unsigned int* counters, unsigned long long* cells_out, unsigned int* why_hist);

// ---- register-resident pre-filter of the flank fallback alignments (wfa_reg.hip) ----
struct FilterArgs {  // by-value kernel argument
  const JobDev* jobs; const uint32_t* n_jobs_dev; uint32_t n_jobs;
  const uint8_t* pat_base; const uint8_t* txt_base;
  unsigned int* counter;
  int32_t min_matches;                      // an alignment is kept iff its match bound >= min_matches (or it could not be judged)
  int32_t early_reject;                     // stop an alignment as soon as no cell of its wavefronts can reach min_matches any more
  JobDev* keep_jobs; uint32_t* keep_count;  // kept alignments, appended (NULL: none wanted)
  int32_t* score; int32_t* bound; uint8_t* keep;  // optional, indexed by JobDev::out_index
  uint32_t* band;                           // optional, by out_index: bit 31 | penalty << 16 | biased end diagonal of a run that completed; bit 30 | level of one rejected early (its penalty is larger); 0: not judged
  int32_t diag_lo, diag_hi;                 // this launch takes the jobs with diag_lo < plen + tlen + 1 <= diag_hi (the others belong to another launch over the same list)
  unsigned long long* cells_out;
};
struct FilterLaunch {
  const JobDev* jobs_dev = nullptr; int64_t n_jobs_host = 0; const uint32_t* n_jobs_dev = nullptr;
  const uint8_t* pat_base = nullptr; const uint8_t* txt_base = nullptr;
  int64_t max_plen = 0, max_tlen = 0;
  int32_t min_matches = 0;
  int mism = 2, gapo = 5, gape = 1;  // --aln-scoring: 2,5,1 (wgs) or 1,0,1 (targeted) have an instantiation
  bool early_reject = false;  // see FilterArgs (then: score INT32_MIN + 1, bound min_matches - 1 for the alignments stopped early)
  JobDev* keep_jobs = nullptr; uint32_t* keep_count = nullptr;
  int32_t* score = nullptr; int32_t* bound = nullptr; uint8_t* keep = nullptr; uint32_t* band = nullptr;
  bool count_offsets = false;  // also count the wavefront offsets (a little slower: one more scalar walk per level)
  int timer_slot = TRGT_K_WFA_FILTER;
  int set = 0;  // 1: a second launch that may run next to the first (own job counter and offset counter; ctx->last_filter_cells_dev stays the first's)
};
// Longest text the filter can judge for this pattern length (0: the filter does not apply); longer texts are kept unseen.
int flank_filter_max_tlen(int flank_len);
// Enqueue the filter on the ctx stream (asynchronous); offsets computed are accumulated at ctx->last_filter_cells_dev.
int flank_filter_launch(trgt_hip_ctx* c, const FilterLaunch& L);

// Run-length CIGARs of a whole batch in one dense array: job j = data[off[j] .. off[j + 1]) (len << 4 | code, as cigar_get_CIGAR).
struct PackedCigars { std::vector<uint32_t> data; std::vector<uint64_t> off; };
// ... or left in HBM for a kernel of the caller (the consensus column voting): job list, CIGAR slots (job j at cigar[jobs[j].cigar_off ..],
// cigar_len[jobs[j].out_index] entries) and the sequence blob, valid until the next alignment batch of the context
struct WfaOnDevice { const JobDev* jobs = nullptr; const uint32_t* cigar = nullptr; const uint32_t* cigar_len = nullptr; const uint8_t* seqs = nullptr; };
// trgt_wfa_batch with an optional dense CIGAR result (packed != nullptr replaces cigar / cigar_off / cigar_len).
int wfa_batch_impl(trgt_hip_ctx* c, const trgt_wfa_params* p, int64_t n_jobs, const uint8_t* seqs, const uint64_t* pat_off,
                   const uint32_t* pat_len, const uint64_t* txt_off, const uint32_t* txt_len, int32_t* status, int32_t* score,
                   int32_t* n_match, uint32_t* span4, uint32_t* cigar, const uint64_t* cigar_off, uint32_t* cigar_len, uint8_t* ops,
                   const uint64_t* ops_off, uint32_t* ops_len, PackedCigars* packed,
                   const std::function<int()>* while_running = nullptr,  // host work to do between the launch and the wait
                   WfaOnDevice* on_device = nullptr);  // != nullptr: run-length CIGARs are produced but stay on the device

}  // namespace trgt
