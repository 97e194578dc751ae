// fmt_load: can the raster take its quad records through FORMAT-CONVERTING buffer loads?
//
// Round-4 question: the integer filter wants its taps as u16 pairs for v_dot2_u32_u16, the record pool
// wants them as bytes (16-byte records: the texture unit is 70 % busy on them already).  MTBUF loads convert on the way in:
//     tbuffer_load_format_d16_xyzw  format:[BUF_DATA_FORMAT_8_8_8_8, BUF_NUM_FORMAT_UINT]
// returns the four bytes of one dword as four u16 in two registers.  This checks (1) that gfx950 executes it with the values
// expected, and (2) what it costs next to the 16-byte load it replaces, alone and under the raster's own VALU load
// (FILL vector instructions per wavefront and four records, 0 = loads only).
//
//   hipcc --offload-arch=gfx950 -O3 -o fmt_load fmt_load.hip && ./fmt_load
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ h16x4 tbuf_load_d16x4(i32x4 rsrc, int voffset, int soffset, int format, int aux) __asm("llvm.amdgcn.raw.tbuffer.load.v4f16");
__device__ int buf_load_i32(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.i32");
__device__ i32x4 buf_load_v4i32(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4i32");
#define FMT_8888_UINT 74   // dfmt 10 (8_8_8_8) | nfmt 4 (UINT) << 4

__device__ inline i32x4 make_rsrc(const void* p, uint32_t bytes, uint32_t w3) {
  i32x4 r;
  r.x = (int)(uintptr_t)p; r.y = (int)(((uintptr_t)p >> 32) & 0xFFFFu); r.z = (int)bytes; r.w = (int)w3;
  return r;
}

// ---- (1) values: every lane loads one record both ways
__global__ void k_check(const uint8_t* pool, uint32_t bytes, uint32_t w3, const int* idx, uint32_t* out) {
  const i32x4 r = make_rsrc(pool, bytes, w3);
  const int off = idx[threadIdx.x] * 16;
  uint2 v[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { const h16x4 h = tbuf_load_d16x4(r, off + 4 * c, 0, FMT_8888_UINT, 0); __builtin_memcpy(&v[c], &h, 8); }
  const int m = buf_load_i32(r, off + 12, 0, 0);
  uint32_t* o = out + threadIdx.x * 8;
  o[0] = v[0].x; o[1] = v[0].y; o[2] = v[1].x; o[3] = v[1].y; o[4] = v[2].x; o[5] = v[2].y; o[6] = (uint32_t)m; o[7] = 0;
}

// ---- (2) cost.  Work shape of the raster's env loop: per wavefront and iteration four records per lane, neighbouring lanes =
// neighbouring records (a texture row under magnification) from a pseudo-random place of a 7 MB pool per (wavefront, iteration).
#ifndef ITERS
#define ITERS 256
#endif
template <int MODE, int FILL>
__global__ __launch_bounds__(256) void k_rate(const uint8_t* pool, const uint8_t* pool32, uint32_t n_rec, uint32_t w3, uint32_t* out, uint32_t lane_stride) {
  const i32x4 r = make_rsrc(pool, n_rec * 16u, w3), r32 = make_rsrc(pool32, n_rec * 32u, w3);
  const uint32_t wave = (blockIdx.x * 4u + (threadIdx.x >> 6)), lane = threadIdx.x & 63u;
  uint32_t acc = 0, h = wave * 2654435761u + 12345u;
  float f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = (float)(lane + i);
  for (int it = 0; it < ITERS; ++it) {
    uint32_t rec[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      h = h * 1664525u + 1013904223u;
      // lane_stride 0: mostly unit stride across lanes, sometimes 3 (a texture row under magnification); else that many records
      rec[k] = ((h >> 8) + lane * (lane_stride ? lane_stride : ((h >> 5) & 3u ? 1u : 3u))) & (n_rec - 1u);
    }
    uint32_t w0 = 0x40004000u + it, w1 = 0x3fff4000u - it;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (MODE == 0) {          // one 16-byte load (global), u8 taps: 2 x v_dot4 per channel stand-in
        const uint4 q = *reinterpret_cast<const uint4*>(pool + (size_t)rec[k] * 16u);
        acc += __builtin_amdgcn_udot4(q.x, w0, q.w, false) + __builtin_amdgcn_udot4(q.y, w1, 0u, false) + __builtin_amdgcn_udot4(q.z, w0, 0u, false);
      } else if (MODE == 1) {   // three format loads + the meta dword
        uint2 v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { const h16x4 hh = tbuf_load_d16x4(r, (int)(rec[k] * 16u) + 4 * c, 0, FMT_8888_UINT, 0); __builtin_memcpy(&v[c], &hh, 8); }
        const uint32_t m = (uint32_t)buf_load_i32(r, (int)(rec[k] * 16u) + 12, 0, 0);
#pragma unroll
        for (int c = 0; c < 3; ++c)
          acc += __builtin_amdgcn_udot2(__builtin_bit_cast(us2, v[c].x), __builtin_bit_cast(us2, w0), __builtin_amdgcn_udot2(__builtin_bit_cast(us2, v[c].y), __builtin_bit_cast(us2, w1), c ? 0u : m, false), false);
      } else if (MODE == 2) {   // two 16-byte loads from 32-byte records (u16 taps in memory)
        const uint4 qa = *reinterpret_cast<const uint4*>(pool32 + (size_t)rec[k] * 32u), qb = *reinterpret_cast<const uint4*>(pool32 + (size_t)rec[k] * 32u + 16u);
        const uint32_t t[6] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y};
#pragma unroll
        for (int c = 0; c < 3; ++c)
          acc += __builtin_amdgcn_udot2(__builtin_bit_cast(us2, t[2 * c]), __builtin_bit_cast(us2, w0), __builtin_amdgcn_udot2(__builtin_bit_cast(us2, t[2 * c + 1]), __builtin_bit_cast(us2, w1), c ? 0u : qb.w, false), false);
      } else if (MODE == 3) {   // one 16-byte BUFFER load (offen): buffer against global addressing
        const i32x4 q = buf_load_v4i32(r, (int)(rec[k] * 16u), 0, 0);
        acc += __builtin_amdgcn_udot4((uint32_t)q.x, w0, (uint32_t)q.w, false) + __builtin_amdgcn_udot4((uint32_t)q.y, w1, 0u, false) + __builtin_amdgcn_udot4((uint32_t)q.z, w0, 0u, false);
      } else {                  // no loads
        acc += rec[k];
      }
    }
    // the raster's own vector work, as independent FMAs (FILL instructions per iteration)
#pragma unroll
    for (int i = 0; i < FILL; ++i) f[i & 7] = __builtin_fmaf(f[i & 7], 1.0001f, 0.5f);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * 256 + threadIdx.x] = acc + (uint32_t)s;
}

int main() {
  const uint32_t n_rec = 7u * 65536u / 1u;                  // 7 blocks of 256 x 256 records ...
  uint32_t n_pow = 1; while (n_pow * 2 <= n_rec) n_pow *= 2; // ... rounded down to a power of two for the mask: 4 MB pool
  const uint32_t N = n_pow * 2;                              // 8 MB: the raster's 7 MB class
  std::vector<uint8_t> h(N * 16u), h32(N * 32u);
  srand(7);
  for (auto& b : h) b = (uint8_t)(rand() >> 5);
  for (uint32_t i = 0; i < N; ++i) {
    uint16_t* d = reinterpret_cast<uint16_t*>(&h32[i * 32u]);
    for (int j = 0; j < 12; ++j) d[j] = h[i * 16u + j];
    memcpy(&h32[i * 32u + 28], &h[i * 16u + 12], 4);
  }
  uint8_t *pool, *pool32; uint32_t* out; int* idx;
  CK(hipMalloc(&pool, h.size())); CK(hipMalloc(&pool32, h32.size())); CK(hipMalloc(&out, 1024 * 1024 * 16)); CK(hipMalloc(&idx, 256 * 4));
  CK(hipMemcpy(pool, h.data(), h.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(pool32, h32.data(), h32.size(), hipMemcpyHostToDevice));
  std::vector<int> hi(64);
  for (int i = 0; i < 64; ++i) hi[i] = (i * 7919 + 13) % (int)N;
  CK(hipMemcpy(idx, hi.data(), 64 * 4, hipMemcpyHostToDevice));
  const uint32_t w3s[2] = {0x00027FACu, 0x00027000u};       // dst_sel xyzw = RGBA | the raw-buffer default (dst_sel 0)
  uint32_t w3_ok = 0;
  for (int v = 0; v < 2; ++v) {
    CK(hipMemset(out, 0xff, 64 * 8 * 4));
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, pool, (uint32_t)h.size(), w3s[v], idx, out);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> o(64 * 8);
    CK(hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
      const uint8_t* rec = &h[(size_t)hi[l] * 16u];
      for (int c = 0; c < 3; ++c) {
        const uint32_t e0 = rec[4 * c] | ((uint32_t)rec[4 * c + 1] << 16), e1 = rec[4 * c + 2] | ((uint32_t)rec[4 * c + 3] << 16);
        bad += o[l * 8 + 2 * c] != e0; bad += o[l * 8 + 2 * c + 1] != e1;
      }
      uint32_t m; memcpy(&m, rec + 12, 4);
      bad += o[l * 8 + 6] != m;
    }
    printf("check word3 = 0x%08x: %s (%d mismatches; lane 0 got %08x %08x, record bytes %02x %02x %02x %02x)\n", w3s[v], bad ? "MISMATCH" : "values as expected", bad,
           o[0], o[1], h[(size_t)hi[0] * 16], h[(size_t)hi[0] * 16 + 1], h[(size_t)hi[0] * 16 + 2], h[(size_t)hi[0] * 16 + 3]);
    if (!bad && !w3_ok) w3_ok = w3s[v];
  }
  if (!w3_ok) { printf("format loads do not return the expected values on this part\n"); w3_ok = w3s[0]; }
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int blocks = 256 * 5 * 4;                           // 5 workgroups of 4 wavefronts per CU, four rounds
  uint32_t pool_recs = N, lane_stride = 0;
  auto run = [&](const char* name, auto kern) {
    float best = 1e30f;
    for (int rr = 0; rr < 4; ++rr) {
      CK(hipEventRecord(a));
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, pool, pool32, pool_recs, w3_ok, out, lane_stride);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      if (rr && ms < best) best = ms;
    }
    const double recs = (double)blocks * 4 * 64 * ITERS * 4;
    printf("%-44s %8.3f ms  %7.1f G records/s\n", name, best, recs / best / 1e6);
  };
#define RUNS(FILL) \
  run("no loads                     FILL " #FILL, k_rate<9, FILL>); \
  run("global_load_dwordx4 (16 B)   FILL " #FILL, k_rate<0, FILL>); \
  run("buffer_load_dwordx4 (16 B)   FILL " #FILL, k_rate<3, FILL>); \
  run("3 x tbuffer d16 + dword      FILL " #FILL, k_rate<1, FILL>); \
  run("2 x global dwordx4 (32 B)    FILL " #FILL, k_rate<2, FILL>);
  printf("-- 8 MB pool (the raster's class), lanes mostly unit stride\n");
  RUNS(0) RUNS(64) RUNS(96)
  // what a gather costs the texture unit by the number of cache lines its 64 lanes touch: a HOT pool (16 K records = 256 KB:
  // L2 hits, no fabric traffic), lanes 1 / 9 / 257 records apart (9: every lane its own 128-byte line, 257: the diagonal of a block)
  pool_recs = 16384;
  const uint32_t strides[3] = {1, 9, 257};
  for (int si = 0; si < 3; ++si) {
    lane_stride = strides[si];
    printf("-- 256 KB pool, lanes %u records apart\n", lane_stride);
    RUNS(0) RUNS(96)
  }
  return 0;
}
