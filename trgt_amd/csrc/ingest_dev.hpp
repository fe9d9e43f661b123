// trgt_amd/csrc/ingest_dev.hpp -- host interface of the device-side read ingestion (ingest_dev.hip): everything of extract_reads /
// HiFiRead::from_hts_rec / clip_reads (src/trgt/workflows/tr.rs:268-361, 186-196; reads/read.rs:55-141; reads/snp.rs:51-79;
// reads/clip_region.rs:19-184) that touches inflated BAM bytes, as kernels behind the device inflate, so that those bytes never leave HBM.
#pragma once
#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "inflate_dev.hpp"

namespace trgt {
namespace ingd {

// One fetch of extract_reads: the window of a locus and its chunks of the .bai as positions in the inflated bytes of the call
struct LocusDesc {
  int32_t tid, chunk_begin, chunk_end, pad;
  int64_t beg, end;                     // fetch window: region -+ flank_len
  int64_t region_start, region_end;     // locus.region
  int64_t clip_start, clip_end;         // clip_reads: region -+ 2 * flank_len
};
struct ChunkDesc { uint64_t lin0, lin1, lin_limit; };  // first record, end of the chunk, end of the inflated range the chunk lies in

// A device slab + its pinned mirror: the per-read arrays of one batch.  They move into the batch (the read bytes stay in HBM for
// trgt_locus_batch) and come back to the pool when the batch is freed.
struct Slab { void* dev = nullptr; void* pin = nullptr; size_t cap = 0; int device = -1; };
struct SlabPool {
  std::mutex mu;
  std::vector<Slab> idle;
  ~SlabPool();
  bool take(int device, size_t bytes, Slab& out);
  void give(Slab& s);
};

struct HostOut {   // pointers into slab.pin (64-byte aligned pieces); dev_* into slab.dev
  int64_t n_reads = 0;
  uint64_t read_bytes = 0, name_bytes = 0, snp_n = 0, meth_n = 0, cig_n = 0, bam4_bytes = 0;
  const uint64_t* lrb = nullptr;            // [n_loci + 1]
  const int32_t* n_filt = nullptr; const int64_t* n_seen = nullptr;   // [n_loci]
  const uint64_t* read_off = nullptr; const uint32_t* read_len = nullptr; const uint8_t* reads = nullptr; const uint8_t* quals = nullptr;
  const char* names = nullptr; const uint64_t* name_off = nullptr;
  const double* rq = nullptr; const uint8_t* is_reverse = nullptr; const uint8_t* mapq = nullptr; const int16_t* hp = nullptr;
  const int32_t* start_offset = nullptr; const int32_t* end_offset = nullptr;
  const int32_t* snp = nullptr; const uint64_t* snp_off = nullptr;
  const uint8_t* meth = nullptr; const uint64_t* meth_off = nullptr; const uint8_t* has_meth = nullptr;
  const uint32_t* cig = nullptr; const uint64_t* cig_off = nullptr; const int64_t* cig_ref_pos = nullptr;
  const uint8_t* bam4 = nullptr; const uint64_t* bam4_off = nullptr;
  const uint8_t* dev_reads = nullptr;       // the ASCII read blob in HBM (same offsets as `reads`)
};

struct RunIn {
  uint64_t src_bytes = 0;                   // compressed bytes staged in slot_src()
  int64_t n_blocks = 0; const infl::BlockDesc* blocks = nullptr; const uint32_t* crc = nullptr;  // BGZF blocks: payload -> inflated position; CRC-32 of the footer
  uint64_t infl_bytes = 0;
  int64_t n_loci = 0; const LocusDesc* loci = nullptr;
  int64_t n_chunks = 0; const ChunkDesc* chunks = nullptr;
  uint32_t reservoir = 750; double min_rq = 0.98; bool keep_bam4 = false; int waves_per_cu = 0;
};
// Why a call went back to the host path (RunOut::fallback)
enum : int { FB_NONE = 0, FB_BLOCK = 1 /* a block the device could not inflate or whose CRC-32 / ISIZE does not match */, FB_WALK = 2 /* a record the walk refuses */,
             FB_RESERVOIR = 3 /* (unused since the reservoir's random stream runs in walk_kernel) */, FB_METH = 4 /* MM / ML beyond the kernel's LDS caps */ };
struct RunOut { int fallback = FB_NONE; Slab slab; HostOut out; double ms_upload = 0, ms_inflate = 0, ms_walk = 0, ms_reads = 0, ms_download = 0; uint64_t blocks_host_inflated = 0; };

class Slot;
Slot* slot_create(int device, std::string& err);
void slot_destroy(Slot* s);
uint8_t* slot_src(Slot* s, size_t bytes, std::string& err);   // pinned staging for the compressed bytes of a call (valid until the next call)
// 0 ok (out.fallback says whether the results are usable), < 0 TRGT_ERR_*
int slot_run(Slot* s, const RunIn& in, SlabPool& pool, RunOut& out, std::string& err);

}  // namespace ingd
}  // namespace trgt
