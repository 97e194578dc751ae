// Store-only models of k_raster's frame writes (4096 x 640 x 480 x 3 B): which loop orders / chunk sizes the memory
// system likes.  Every kernel writes each byte of the 3.77 GB batch exactly once with 12 B per lane (4 pixels).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int N = 4096, W = 640, H = 480;
constexpr size_t FRAME = (size_t)W * H * 3, BYTES = FRAME * N;

// workgroup = TW x (1024 / TW) pixel tile, wavefront = TW x (256 / TW) block, EPB envs per workgroup.
// ORDER 0: blockIdx = chunk * n_tiles + tile (tile fastest)   1: blockIdx = tile * n_chunks + chunk (chunk fastest)
// ORDER 2: tile fastest, but each workgroup starts its env loop at a different env ((tile * 7) % EPB): de-correlated
template <int TW, int EPB, int ORDER>
__global__ __launch_bounds__(256) void fill(uint8_t* frames, uint32_t v) {
  constexpr int TH = 1024 / TW, WH = 256 / TW;
  constexpr int tiles_x = (W + TW - 1) / TW, n_tiles = tiles_x * (H / TH), n_chunks = N / EPB;
  const int tile = ORDER == 1 ? blockIdx.x / n_chunks : blockIdx.x % n_tiles;
  const int chunk = ORDER == 1 ? blockIdx.x % n_chunks : blockIdx.x / n_tiles;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int x = (tile % tiles_x) * TW + (lane * 4) % TW, y = (tile / tiles_x) * TH + wave * WH + (lane * 4) / TW;
  if (x >= W) return;
  const size_t off = ((size_t)y * W + x) * 3;
  const int rot = ORDER == 2 ? (tile * 7) % EPB : 0;
  for (int i = 0; i < EPB; ++i) {
    const int e = chunk * EPB + (i + rot) % EPB;
    uint32_t* d = reinterpret_cast<uint32_t*>(frames + (size_t)e * FRAME + off);
    d[0] = v + e; d[1] = v ^ lane; d[2] = v + wave;
  }
}
template <int TW, int EPB, int ORDER> void run(uint8_t* buf, const char* name) {
  constexpr int TH = 1024 / TW;
  const int grid = ((W + TW - 1) / TW) * (H / TH) * (N / EPB);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  fill<TW, EPB, ORDER><<<grid, 256>>>(buf, 3u); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  for (int r = 0; r < 4; ++r) fill<TW, EPB, ORDER><<<grid, 256>>>(buf, 7u + r);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= 4;
  printf("%-44s grid %6d: %.3f ms  %.2f TB/s\n", name, grid, ms, BYTES / (ms * 1e-3) / 1e12);
}
int main() {
  uint8_t* buf; if (hipMalloc(&buf, BYTES) != hipSuccess) { printf("alloc failed\n"); return 1; }
  run<64, 32, 0>(buf, "64x16 tile, 32 envs, tile-fastest (current)");
  run<64, 16, 0>(buf, "64x16 tile, 16 envs, tile-fastest");
  run<64, 8, 0>(buf, "64x16 tile,  8 envs, tile-fastest");
  run<64, 4, 0>(buf, "64x16 tile,  4 envs, tile-fastest");
  run<64, 64, 0>(buf, "64x16 tile, 64 envs, tile-fastest");
  run<64, 32, 1>(buf, "64x16 tile, 32 envs, chunk-fastest");
  run<64, 32, 2>(buf, "64x16 tile, 32 envs, rotated env start");
  run<128, 32, 0>(buf, "128x8 tile, 32 envs, tile-fastest");
  run<128, 16, 0>(buf, "128x8 tile, 16 envs, tile-fastest");
  run<256, 32, 0>(buf, "256x4 tile, 32 envs, tile-fastest");
  run<256, 8, 0>(buf, "256x4 tile,  8 envs, tile-fastest");
  run<32, 32, 0>(buf, "32x32 tile, 32 envs, tile-fastest");
  (void)hipFree(buf);
  return 0;
}
