// Dependent-load latency on gfx950: global (L2-resident), flat pointer into LDS, ds_read.  hipcc --offload-arch=gfx950 lat.hip -o lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_global(const uint32_t* p, int n, uint32_t* out, long long* cyc) {
  uint32_t i = threadIdx.x;
  long long t0 = clock64();
  for (int s = 0; s < n; ++s) i = p[i];
  long long t1 = clock64();
  if (threadIdx.x == 0) { *out = i; *cyc = t1 - t0; }
}
__global__ void k_flat_lds(const uint32_t* src, int n, uint32_t* out, long long* cyc) {
  __shared__ uint32_t l[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) l[i] = src[i];
  __syncthreads();
  const uint32_t* volatile gp = l;  // generic pointer the compiler cannot trace back to LDS
  const uint32_t* q = gp;
  uint32_t i = threadIdx.x;
  long long t0 = clock64();
  for (int s = 0; s < n; ++s) i = q[i];
  long long t1 = clock64();
  if (threadIdx.x == 0) { *out = i; *cyc = t1 - t0; }
}
__global__ void k_ds(const uint32_t* src, int n, uint32_t* out, long long* cyc) {
  __shared__ uint32_t l[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) l[i] = src[i];
  __syncthreads();
  uint32_t i = threadIdx.x;
  long long t0 = clock64();
  for (int s = 0; s < n; ++s) i = l[i];
  long long t1 = clock64();
  if (threadIdx.x == 0) { *out = i; *cyc = t1 - t0; }
}
int main() {
  const int N = 4096, steps = 2000;
  std::vector<uint32_t> h(N);
  for (int i = 0; i < N; ++i) h[i] = (uint32_t)((i * 1237 + 64) % N);
  uint32_t *d, *out; long long* cyc;
  hipMalloc(&d, N * 4); hipMalloc(&out, 4); hipMalloc(&cyc, 8);
  hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
  long long c; 
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_global, dim3(1), dim3(64), 0, 0, d, steps, out, cyc); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    if (rep) printf("global (L1/L2 resident) : %.0f cycles per dependent load (s_memtime ticks, 100 MHz: x24 for shader cycles?) raw %lld\n", (double)c / steps, c);
    hipLaunchKernelGGL(k_flat_lds, dim3(1), dim3(64), 0, 0, d, steps, out, cyc); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    if (rep) printf("flat pointer into LDS   : %.0f per load, raw %lld\n", (double)c / steps, c);
    hipLaunchKernelGGL(k_ds, dim3(1), dim3(64), 0, 0, d, steps, out, cyc); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    if (rep) printf("ds_read                 : %.0f per load, raw %lld\n", (double)c / steps, c);
  }
  return 0;
}
