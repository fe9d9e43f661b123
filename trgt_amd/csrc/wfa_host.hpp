// trgt_amd/csrc/wfa_host.hpp -- kernel-argument block and host-side launch descriptor of the WFA kernel.
#pragma once
#include "wfa_engine.hpp"

namespace trgt {

struct JobDev {  // one alignment: pattern / text inside pat_base / txt_base, output slots
  uint64_t pat_off, txt_off, cigar_off, ops_off;
  uint32_t pat_len, txt_len, out_index, pad;
};

namespace wfa {
struct KArgs {
  KParams kp;
  const JobDev* jobs; const uint32_t* n_jobs_dev; uint32_t n_jobs;
  const uint8_t* pat_base; const uint8_t* txt_base;
  unsigned int* counter;
  uint8_t* ws; uint64_t ws_per_block;
  uint64_t off_gdesc, off_arena_u, off_arena_f, off_arena_r, off_rle_tmp, off_rle_out, off_run_start;
  uint32_t uni_slots, arena_uni_cap, ring_stride, rle_cap, lds_seq_cap;
  int32_t* status; int32_t* score; int32_t* n_match; uint32_t* span4; uint32_t* cigar; uint32_t* cigar_len; uint8_t* ops; uint32_t* ops_len;
  unsigned long long* cells_out;
};
}  // namespace wfa

struct WfaLaunch {  // everything device-resident
  const JobDev* jobs_dev = nullptr; int64_t n_jobs_host = 0; const uint32_t* n_jobs_dev = nullptr;
  const uint8_t* pat_base = nullptr; const uint8_t* txt_base = nullptr;
  int64_t max_plen = 0, max_tlen = 0, max_sum = 0;
  int threads = 0;
  int32_t* status = nullptr; int32_t* score = nullptr; int32_t* n_match = nullptr; uint32_t* span4 = nullptr;
  uint32_t* cigar = nullptr; uint32_t* cigar_len = nullptr; uint8_t* ops = nullptr; uint32_t* ops_len = nullptr;
};

// Enqueue the WFA kernel on the ctx stream (asynchronous).  The number of wavefront offsets computed is
// accumulated in device memory at ctx->last_wfa_cells_dev.
int wfa_launch(trgt_hip_ctx* c, const trgt_wfa_params& p, const WfaLaunch& L);

}  // namespace trgt
