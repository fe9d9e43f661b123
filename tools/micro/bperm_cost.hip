// bperm_cost.hip -- what a ds_bpermute_b32 costs a single wave on gfx950: a dependent chain (latency) and batches of independent ones
// (issue / crossbar throughput), against dependent f64 adds, v_readlane and DPP moves.  hipcc --offload-arch=gfx950 -O3 -o bperm_cost bperm_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int mode, int iters, int* out, long long* cyc) {
  int lane = threadIdx.x, x = lane * 7 + 1, addr = ((lane * 13 + 5) & 63) << 2;
  double d = lane * 0.5, e = 1.25;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    if (mode == 0) {  // 16 dependent bpermutes
#pragma unroll
      for (int j = 0; j < 16; ++j) x = __builtin_amdgcn_ds_bpermute(addr, x);
    } else if (mode == 1) {  // 16 independent bpermutes, one wait
      int y[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) y[j] = __builtin_amdgcn_ds_bpermute(addr + 4 * j, x + j);
      int s = 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) s ^= y[j];
      x = s;
    } else if (mode == 2) {  // 4 independent bpermutes, one wait
      int y[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] = __builtin_amdgcn_ds_bpermute(addr + 4 * j, x + j);
      x = y[0] ^ y[1] ^ y[2] ^ y[3];
    } else if (mode == 3) {  // 16 dependent f64 adds
#pragma unroll
      for (int j = 0; j < 16; ++j) d = d + e;
    } else if (mode == 4) {  // 16 dependent (cmp, 2 cndmask) on f64
#pragma unroll
      for (int j = 0; j < 16; ++j) { double v = d + e; d = v > d ? v : d; e = -e; }
    } else if (mode == 5) {  // 16 dependent readlanes
#pragma unroll
      for (int j = 0; j < 16; ++j) x = __builtin_amdgcn_readlane(x, 5) + lane;
    } else if (mode == 6) {  // 16 dependent DPP wave_shr:1
#pragma unroll
      for (int j = 0; j < 16; ++j) x = __builtin_amdgcn_update_dpp(x, x, 0x138, 0xF, 0xF, false) + 1;
    } else if (mode == 8) {  // 16 independent f64 adds
      double z[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) z[j] = d + (e + j);
      double acc = z[0];
#pragma unroll
      for (int j = 1; j < 16; ++j) acc = z[j] > acc ? z[j] : acc;
      d = acc;
    } else if (mode == 9) {  // 32 independent 32-bit selects
      int y[32];
#pragma unroll
      for (int j = 0; j < 32; ++j) y[j] = (x & (1 << (j & 7))) ? x + j : lane;
      int s2 = 0;
#pragma unroll
      for (int j = 0; j < 32; ++j) s2 += y[j];
      x = s2;
    } else if (mode == 10) {  // 16 dependent 32-bit adds
#pragma unroll
      for (int j = 0; j < 16; ++j) x = x * 3 + lane;
    } else if (mode == 11) {  // 16 independent f64 adds only (summed as integers afterwards)
      double z[16]; long long acc = 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) z[j] = d + (double)(j + 1);
#pragma unroll
      for (int j = 0; j < 16; ++j) acc ^= __double_as_longlong(z[j]);
      d = __longlong_as_double(acc | 0x3FF0000000000000ll);
    } else if (mode == 7) {  // 16 independent bpermutes, all lanes the same source (broadcast)
      int y[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) y[j] = __builtin_amdgcn_ds_bpermute(20, x + j);
      int s = 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) s ^= y[j];
      x = s;
    }
  }
  long long t1 = clock64();
  out[lane] = x + (int)d;
  if (lane == 0) *cyc = t1 - t0;
}
int main() {
  int* out; long long* cyc;
  hipMalloc(&out, 256); hipMallocManaged(&cyc, 8);
  const char* names[] = {"16 dependent ds_bpermute", "16 independent ds_bpermute + wait", "4 independent ds_bpermute + wait", "16 dependent v_add_f64",
                         "16 dependent (add, cmp, 2 cndmask) f64", "16 dependent v_readlane(+add)", "16 dependent DPP wave_shr:1(+add)", "16 independent ds_bpermute, broadcast source", "16 indep f64 adds + 15 dependent (cmp, 2 cndmask)", "32 indep (and, cmp, add, cndmask) + 32 adds", "16 dependent (mul, add) u32", "16 indep f64 adds + 16 xor64"};
  for (int m = 0; m < 12; ++m) {
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, m, 10000, out, cyc); hipDeviceSynchronize(); }
    printf("%-46s %8.1f cycles per iteration (clock64 ticks)\n", names[m], (double)*cyc / 10000);
  }
  long long t;
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, 3, 1000000, out, cyc); hipDeviceSynchronize();
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, 3, 1000000, out, cyc); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); t = *cyc;
  printf("clock64 ticks per microsecond: %.1f\n", (double)t / (ms * 1000));
  return 0;
}
