// trgt_amd/csrc/inflate_dev.hpp -- host interface of the device-side DEFLATE decoder (inflate_dev.hip)
#pragma once
#include <cstdint>

struct trgt_hip_ctx;

namespace trgt {
namespace infl {
struct BlockDesc { uint64_t src_off, dst_off; uint32_t src_len, dst_len; };  // one raw DEFLATE stream and where its bytes go
}
// n streams src[src_off .. + src_len) -> dst[dst_off .. + dst_len) (host or device memory; pinned host memory keeps the copies at link
// speed), status[b] = 1 inflated, 0 declined.  Synchronous.
int inflate_blocks_device(trgt_hip_ctx* c, int64_t n, const uint8_t* src, uint64_t src_bytes, const infl::BlockDesc* descs, uint8_t* dst, uint64_t dst_bytes,
                          uint8_t* status, bool preserve_dst = false);  // preserve_dst: host bytes of dst outside the blocks survive (the whole range is copied back)
// Asynchronous launch with everything in HBM already (ingest_dev.hip): `waves` streams in flight, `d_counter` zeroed by the caller on the
// same stream.  dst_off need not be aligned: the blocks of a BGZF range are laid end to end so that records run across them.
void inflate_launch(void* hip_stream, const uint8_t* d_src, const infl::BlockDesc* d_blocks, uint32_t n, uint8_t* d_dst, uint8_t* d_status, unsigned* d_counter,
                    unsigned waves);
}  // namespace trgt
