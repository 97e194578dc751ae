// gather_calib: what does FETCH_SIZE (TCC -> memory read requests, KB) report for the raster's access pattern?
// MI355X_MICROARCH.md calibrates the "gfx950 reports half the bytes" rule on wide coalesced 16 B / lane streams only; the
// raster's dominant fetch is a 16-byte record GATHER.  Each kernel below reads a known number of bytes / cache lines
// exactly once per launch; run under  rocprofv3 --pmc FETCH_SIZE TCC_MISS_sum TCC_HIT_sum  and compare (tools/prof_calib.sh).
//   k_stream : lane i reads the 16 B at i * 16                       (wide coalesced: the guide's case)
//   k_gather : lane i reads the 16-B record perm(i), every record of the pool once, in a scrambled order
//   k_sparse : lane i reads 16 B at perm(i) * 128: ONE record per 128-byte line, every line once
// pools: 1 GiB (streams from HBM) and 8 MiB (the raster's record pool size: L2 / MALL resident after the warm-up launch)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ inline uint32_t scramble(uint32_t i, uint32_t mask) {   // a bijection on [0, mask]: odd multiply + xorshift, masked
  uint32_t x = (i * 2654435761u) & mask;
  x ^= x >> 7; x &= mask;
  x = (x * 40503u + 12345u) & mask;
  return x;
}
__global__ void k_stream(const uint4* __restrict__ p, uint4* __restrict__ sink, uint32_t n) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { const uint4 v = p[i]; acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w; }
  if (acc.x == 0x12345678u) sink[0] = acc;
}
__global__ void k_gather(const uint4* __restrict__ p, uint4* __restrict__ sink, uint32_t n) {   // n = 2^k records
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { const uint4 v = p[scramble(i, n - 1)]; acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w; }
  if (acc.x == 0x12345678u) sink[0] = acc;
}
__global__ void k_sparse(const uint4* __restrict__ p, uint4* __restrict__ sink, uint32_t n_lines) {   // one record per 128-B line
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_lines; i += gridDim.x * blockDim.x) { const uint4 v = p[(size_t)scramble(i, n_lines - 1) * 8]; acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w; }
  if (acc.x == 0x12345678u) sink[0] = acc;
}
int main() {
  const size_t big = (size_t)1 << 30, small = (size_t)8 << 20;
  uint4 *pb, *ps, *sink;
  CK(hipMalloc(&pb, big)); CK(hipMalloc(&ps, small)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(pb, 1, big)); CK(hipMemset(ps, 1, small));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto run = [&](const char* name, auto kern, uint4* p, uint32_t n, double bytes) {
    for (int r = 0; r < 3; ++r) {
      CK(hipEventRecord(a));
      hipLaunchKernelGGL(kern, dim3(4096), dim3(256), 0, 0, p, sink, n);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      if (r == 2) printf("%-28s %10.0f KB requested per launch  %.3f ms  %.2f TB/s\n", name, bytes / 1024.0, ms, bytes / ms / 1e9);
    }
  };
  run("k_stream 1GiB", k_stream, pb, (uint32_t)(big / 16), (double)big);
  run("k_gather 1GiB", k_gather, pb, (uint32_t)(big / 16), (double)big);
  run("k_sparse 1GiB (lines once)", k_sparse, pb, (uint32_t)(big / 128), (double)big / 8);
  run("k_stream 8MiB", k_stream, ps, (uint32_t)(small / 16), (double)small);
  run("k_gather 8MiB", k_gather, ps, (uint32_t)(small / 16), (double)small);
  run("k_sparse 8MiB (lines once)", k_sparse, ps, (uint32_t)(small / 128), (double)small / 8);
  return 0;
}
