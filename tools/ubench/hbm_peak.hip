// What the box sustains on the frame buffer of C3 (4096 x 640 x 480 x 3 B = 3.77 GB): the store-only
// ceiling the raster is priced against.  ("confirm on the box ... and report both".)
//   fill16      every thread streams 16-byte stores, grid-stride
//   fill_tiles  the store pattern of k_raster: workgroup = 64 x 16 pixel tile x 32 envs, lane = 12 B (4 px),
//               192 contiguous bytes per tile row, env-major loop
//   copy16      read half / write half (2 x 1.89 GB)
//   memset      hipMemsetAsync
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int N = 4096, W = 640, H = 480;
constexpr size_t FRAME = (size_t)W * H * 3, BYTES = FRAME * N;

__global__ __launch_bounds__(256) void fill16(uint4* p, size_t n16, uint32_t v) {
  const uint4 q = {v, v + 1, v + 2, v + 3};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = q;
}
__global__ __launch_bounds__(256) void copy16(const uint4* __restrict__ s, uint4* __restrict__ d, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) d[i] = s[i];
}
__global__ __launch_bounds__(256) void fill_tiles(uint8_t* frames, uint32_t v) {
  const int tiles_x = W / 64, n_tiles = tiles_x * (H / 16);
  const int tile = blockIdx.x % n_tiles, chunk = blockIdx.x / n_tiles;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int x = (tile % tiles_x) * 64 + (lane * 4) % 64, y = (tile / tiles_x) * 16 + wave * 4 + (lane * 4) / 64;
  const size_t off = ((size_t)y * W + x) * 3;
  for (int e = chunk * 32; e < chunk * 32 + 32; ++e) {
    uint32_t* d = reinterpret_cast<uint32_t*>(frames + (size_t)e * FRAME + off);
    d[0] = v + e; d[1] = v ^ lane; d[2] = v + wave;
  }
}
// same tile, 16 contiguous bytes per lane: 12 lanes per 192-B tile row, 48 active lanes, one dwordx4 store
__global__ __launch_bounds__(256) void fill_tiles16(uint8_t* frames, uint32_t v) {
  const int tiles_x = W / 64, n_tiles = tiles_x * (H / 16);
  const int tile = blockIdx.x % n_tiles, chunk = blockIdx.x / n_tiles;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = lane / 12, seg = lane % 12;
  const size_t off = ((size_t)((tile / tiles_x) * 16 + wave * 4 + row) * W + (tile % tiles_x) * 64) * 3 + seg * 16;
  if (lane >= 48) return;
  for (int e = chunk * 32; e < chunk * 32 + 32; ++e) {
    uint4* d = reinterpret_cast<uint4*>(frames + (size_t)e * FRAME + off);
    *d = uint4{v + e, v ^ lane, v + wave, v};
  }
}
// wavefront = 256 x 1 pixels (768 contiguous bytes), workgroup = 256 x 4
__global__ __launch_bounds__(256) void fill_rows16(uint8_t* frames, uint32_t v) {
  const int tiles_x = (W + 255) / 256, n_tiles = tiles_x * (H / 4);
  const int tile = blockIdx.x % n_tiles, chunk = blockIdx.x / n_tiles;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int x0 = (tile % tiles_x) * 256, y = (tile / tiles_x) * 4 + wave;
  const size_t off = ((size_t)y * W + x0) * 3 + lane * 16;
  if (lane >= 48 || x0 * 3 + lane * 16 >= W * 3) return;
  for (int e = chunk * 32; e < chunk * 32 + 32; ++e) {
    uint4* d = reinterpret_cast<uint4*>(frames + (size_t)e * FRAME + off);
    *d = uint4{v + e, v ^ lane, v + wave, v};
  }
}
template <typename F> float timeit(F f, int reps) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
  uint8_t* buf; CK(hipMalloc(&buf, BYTES));
  const size_t n16 = BYTES / 16;
  for (int blocks : {256 * 8, 256 * 32}) {
    float ms = timeit([&] { fill16<<<blocks, 256>>>((uint4*)buf, n16, 7u); }, 5);
    printf("fill16     grid %5d: %.3f ms  %.2f TB/s written\n", blocks, ms, BYTES / (ms * 1e-3) / 1e12);
  }
  {
    const int grid = (W / 64) * (H / 16) * (N / 32);
    float ms = timeit([&] { fill_tiles<<<grid, 256>>>(buf, 7u); }, 5);
    printf("fill_tiles grid %5d: %.3f ms  %.2f TB/s written (k_raster's store pattern, no compute)\n", grid, ms, BYTES / (ms * 1e-3) / 1e12);
  }
  {
    const int grid = (W / 64) * (H / 16) * (N / 32);
    float ms = timeit([&] { fill_tiles16<<<grid, 256>>>(buf, 7u); }, 5);
    printf("fill_tiles16 grid %5d: %.3f ms  %.2f TB/s written (same tile, one 16-B store per lane, 48 lanes)\n", grid, ms, BYTES / (ms * 1e-3) / 1e12);
  }
  {
    const int grid = ((W + 255) / 256) * (H / 4) * (N / 32);
    float ms = timeit([&] { fill_rows16<<<grid, 256>>>(buf, 7u); }, 5);
    printf("fill_rows16  grid %5d: %.3f ms  %.2f TB/s written (256 x 1 px per wavefront, 16-B stores)\n", grid, ms, BYTES / (ms * 1e-3) / 1e12);
  }
  {
    float ms = timeit([&] { copy16<<<256 * 32, 256>>>((const uint4*)buf, (uint4*)(buf + BYTES / 2), n16 / 2); }, 5);
    printf("copy16     : %.3f ms  %.2f TB/s read+written\n", ms, BYTES / (ms * 1e-3) / 1e12);
  }
  {
    float ms = timeit([&] { (void)hipMemsetAsync(buf, 1, BYTES, 0); }, 5);
    printf("hipMemset  : %.3f ms  %.2f TB/s written\n", ms, BYTES / (ms * 1e-3) / 1e12);
  }
  (void)hipFree(buf);
  return 0;
}
