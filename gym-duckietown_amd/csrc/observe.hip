// observe.hip -- learner-side observation post-processing on the device.
//
// Replaces, for the whole frame batch at once, what the reference's learners do per env on the host:
//   learning/utils/wrappers.py:38-54  ResizeWrapper   (scipy imresize == PIL Image.resize BILINEAR)
//   learning/utils/wrappers.py:57-69  NormalizeWrapper (/ 255 -> float32)
//   learning/utils/wrappers.py:72-86  ImgWrapper       (HWC -> CHW)
// The resize is Pillow's antialiased two-pass resampler in 8-bit fixed point (22-bit coefficients,
// uint8 intermediate after the horizontal pass); the coefficient tables come from the host
// (dtsim/resample.py, pinned bit-exact against PIL) so the device result is bit-identical to
// `Image.resize`.  HBM-read bound: every frame byte is read once (plus the row overlap between
// neighbouring row blocks), the output is 16x smaller at 640x480 -> 160x120.
#include "dtsim_dev.h"
#include <cstdlib>

namespace {

#define OB 256                 // threads per workgroup
#ifndef OBS_STAGE_ROWS
#define OBS_STAGE_ROWS DT_OBS_STAGE_ROWS   // input rows staged in LDS per horizontal step
#endif
#define PREC 22                // Pillow PRECISION_BITS (32 - 8 - 2)

__device__ inline uint32_t clip8(int32_t acc) {
  const int32_t v = acc >> PREC;
  return (uint32_t)min(max(v, 0), 255);
}

// ---- vertical pass + layout / normalisation.  PER = 4 consecutive intermediate bytes per thread when the
// rows are dword-sized (compile-time, so the accumulators stay in registers), else 1.
template <int PER>
__device__ inline void observe_vertical_t(const ObserveParams& P, const uint8_t* s_tmp, int e, int oy0, int oy1, int y_first, int tid) {
  const int tmp_row_bytes = P.ow * 3, per_row = tmp_row_bytes / PER;
  const int n_out = (oy1 - oy0) * per_row;
  for (int i = tid; i < n_out; i += OB) {
    const int oy = oy0 + i / per_row, j0 = (i % per_row) * PER;
    uint32_t v[PER];
    if (PER == 4 && P.vfast && oy > 0 && oy < P.oh - 1) {
      // uniform small-integer taps: four bytes per step in two 16-bit lanes (sum of the weights = 2^vsh: a lane never carries)
      const int S = P.vfast;
      const uint8_t* p = s_tmp + (size_t)(S * oy - S / 2 - y_first) * tmp_row_bytes + j0;
      uint32_t lo = 0x00010001u << (P.vsh - 1), hi = lo;                  // the rounding half
      for (int t = 0; t < 2 * S; ++t) {
        const uint32_t w = *reinterpret_cast<const uint32_t*>(p + (size_t)t * tmp_row_bytes);
        lo = __umul24(w & 0x00FF00FFu, P.vw[t]) + lo;
        hi = __umul24((w >> 8) & 0x00FF00FFu, P.vw[t]) + hi;
      }
      lo = (lo >> P.vsh) & 0x00FF00FFu; hi = (hi >> P.vsh) & 0x00FF00FFu;
      const uint32_t packed = lo | (hi << 8);
      if (!P.chw && !P.f32) {
        *reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(P.out) + ((size_t)e * P.oh + oy) * tmp_row_bytes + j0) = packed;
        continue;
      }
#pragma unroll
      for (int q = 0; q < PER; ++q) v[q] = (packed >> (8 * q)) & 255u;
    } else if (P.oh == P.H) {
#pragma unroll
      for (int q = 0; q < PER; ++q) v[q] = s_tmp[(size_t)(oy - y_first) * tmp_row_bytes + j0 + q];
    } else {
      const int y0 = P.by[2 * oy], n = P.by[2 * oy + 1];
      const int32_t* k = P.kky + oy * P.ky;
      const uint8_t* p = s_tmp + (size_t)(y0 - y_first) * tmp_row_bytes + j0;
      int32_t a[PER];
#pragma unroll
      for (int q = 0; q < PER; ++q) a[q] = 1 << (PREC - 1);
      for (int t = 0; t < n; ++t) {
        const uint32_t kt = (uint32_t)k[t];
        if (PER == 4) {
          const uint32_t w = *reinterpret_cast<const uint32_t*>(p + (size_t)t * tmp_row_bytes);
#pragma unroll
          for (int q = 0; q < PER; ++q) a[q] += (int32_t)__umul24((w >> (8 * q)) & 255u, kt);
        } else {
          a[0] += (int32_t)__umul24((uint32_t)p[(size_t)t * tmp_row_bytes], kt);
        }
      }
#pragma unroll
      for (int q = 0; q < PER; ++q) v[q] = clip8(a[q]);
    }
    if (PER == 4 && !P.chw && !P.f32) {                // HWC uint8: one dword store
      *reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(P.out) + ((size_t)e * P.oh + oy) * tmp_row_bytes + j0) =
          v[0] | (v[1 % PER] << 8) | (v[2 % PER] << 16) | (v[3 % PER] << 24);
      continue;
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int j = j0 + q, ox = j / 3, c = j % 3;
      const size_t o = P.chw ? (((size_t)e * 3 + c) * P.oh + oy) * P.ow + ox : ((size_t)e * P.oh + oy) * tmp_row_bytes + j;
      if (P.f32) reinterpret_cast<float*>(P.out)[o] = (float)v[q] / 255.0f;     // NormalizeWrapper: (obs - 0) / (255 - 0)
      else reinterpret_cast<uint8_t*>(P.out)[o] = (uint8_t)v[q];
    }
  }
}
__device__ inline void observe_vertical(const ObserveParams& P, const uint8_t* s_tmp, int e, int oy0, int oy1, int y_first, int tid) {
  if (((P.ow * 3) & 3) == 0) observe_vertical_t<4>(P, s_tmp, e, oy0, oy1, y_first, tid);
  else observe_vertical_t<1>(P, s_tmp, e, oy0, oy1, y_first, tid);
}

__global__ __launch_bounds__(OB) void k_observe(ObserveParams P) {
  extern __shared__ uint32_t s_mem[];
  const int tid = threadIdx.x;
  const int n_blocks = (P.oh + P.rows_per_block - 1) / P.rows_per_block;
  const int e = blockIdx.x / n_blocks, blk = blockIdx.x % n_blocks;
  const int oy0 = blk * P.rows_per_block, oy1 = min(oy0 + P.rows_per_block, P.oh);
  const int y_first = P.by[2 * oy0], y_last = P.by[2 * (oy1 - 1)] + P.by[2 * (oy1 - 1) + 1];
  const int rows_in = y_last - y_first;
  const int in_row_bytes = P.W * 3, in_row_words = (in_row_bytes + 3) >> 2;
  const int tmp_row_bytes = P.ow * 3;
  uint8_t* s_row = reinterpret_cast<uint8_t*>(s_mem);                               // [OBS_STAGE_ROWS][in_row_words * 4] (+32 B slack)
  uint8_t* s_tmp = s_row + (size_t)OBS_STAGE_ROWS * in_row_words * 4 + 32;          // [max_rows_in][ow * 3]
  int32_t* s_bx = reinterpret_cast<int32_t*>(s_tmp + (((size_t)P.max_rows_in * tmp_row_bytes + 3) & ~(size_t)3));   // [ow][2]
  int32_t* s_kx = s_bx + 2 * P.ow;                                                  // [ow][9] (fast path only)
  if (P.ow != P.W && P.kx <= 9 && !P.hfast) {
    for (int i = tid; i < 2 * P.ow; i += OB) s_bx[i] = P.bx[i];
    for (int i = tid; i < 9 * P.ow; i += OB) s_kx[i] = (i % 9) < P.kx ? P.kkx[(i / 9) * P.kx + (i % 9)] : 0;
  }
  const uint8_t* frame = P.frames + (size_t)e * P.H * in_row_bytes;
  const bool aligned = (in_row_bytes & 3) == 0;

  // ---- horizontal pass (skipped when the width is unchanged: Pillow then resamples rows only).
  // The rows of stage s+1 are prefetched into registers while stage s is filtered from LDS, so the
  // global-load latency is off the critical path (one stage = OBS_STAGE_ROWS rows).
  constexpr int PF = 16;                               // prefetch registers per thread
  const bool pipelined = aligned && OBS_STAGE_ROWS * in_row_words <= PF * OB;
  uint32_t pf[PF];
  auto prefetch = [&](int r0) {
    const int nr = min(OBS_STAGE_ROWS, rows_in - r0);
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int i = tid + q * OB;
      if (i < nr * in_row_words) {
        const int rr = i / in_row_words, wd = i % in_row_words;
        pf[q] = reinterpret_cast<const uint32_t*>(frame + (size_t)(y_first + r0 + rr) * in_row_bytes)[wd];
      }
    }
  };
  if (pipelined && rows_in > 0) prefetch(0);
  for (int r0 = 0; r0 < rows_in; r0 += OBS_STAGE_ROWS) {
    const int nr = min(OBS_STAGE_ROWS, rows_in - r0);
    if (pipelined) {
#pragma unroll
      for (int q = 0; q < PF; ++q) {
        const int i = tid + q * OB;
        if (i < nr * in_row_words) reinterpret_cast<uint32_t*>(s_row)[i] = pf[q];    // rows are contiguous: i == rr * in_row_words + wd
      }
    } else if (aligned) {                            // coalesced dword loads of whole rows
      for (int i = tid; i < nr * in_row_words; i += OB) {
        const int rr = i / in_row_words, wd = i % in_row_words;
        reinterpret_cast<uint32_t*>(s_row)[rr * in_row_words + wd] =
            reinterpret_cast<const uint32_t*>(frame + (size_t)(y_first + r0 + rr) * in_row_bytes)[wd];
      }
    } else {
      for (int i = tid; i < nr * in_row_bytes; i += OB) {
        const int rr = i / in_row_bytes, bb = i % in_row_bytes;
        s_row[rr * in_row_words * 4 + bb] = frame[(size_t)(y_first + r0 + rr) * in_row_bytes + bb];
      }
    }
    __syncthreads();
    if (pipelined && r0 + OBS_STAGE_ROWS < rows_in) prefetch(r0 + OBS_STAGE_ROWS);
    if (P.hfast) {
      // power-of-two scale: every interior column has the same taps; three chains of v_dot4_u32_u8 over the column's aligned
      // dwords with the channel's weights at their byte positions (kernel arguments: scalar registers)
      const int S = P.hfast, sh = P.hsh;
      for (int i = tid; i < nr * P.ow; i += OB) {
        const int rr = i / P.ow, ox = i % P.ow;
        uint8_t* dst = s_tmp + (size_t)(r0 + rr) * tmp_row_bytes + ox * 3;
        if (ox == 0 || ox == P.ow - 1) {             // border columns: clipped, renormalised taps from the tables
          const uint8_t* src = s_row + rr * in_row_words * 4;
          const int x0 = P.bx[2 * ox], n = P.bx[2 * ox + 1];
          const int32_t* k = P.kkx + ox * P.kx;
          int32_t a0 = 1 << (PREC - 1), a1 = a0, a2 = a0;
          const uint8_t* p = src + x0 * 3;
          for (int t = 0; t < n; ++t) {
            const int32_t kt = k[t];
            a0 += (int32_t)p[3 * t] * kt; a1 += (int32_t)p[3 * t + 1] * kt; a2 += (int32_t)p[3 * t + 2] * kt;
          }
          dst[0] = (uint8_t)clip8(a0); dst[1] = (uint8_t)clip8(a1); dst[2] = (uint8_t)clip8(a2);
          continue;
        }
        const uint32_t* wsrc = reinterpret_cast<const uint32_t*>(s_row + rr * in_row_words * 4) + ((3 * S * ox + P.hoff) >> 2);
        uint32_t a0 = 1u << (sh - 1), a1 = a0, a2 = a0;
        if (P.hn == 7) {
#pragma unroll
          for (int d = 0; d < 7; ++d) {
            const uint32_t w = wsrc[d];
            a0 = __builtin_amdgcn_udot4(w, P.hw[0][d], a0, false); a1 = __builtin_amdgcn_udot4(w, P.hw[1][d], a1, false); a2 = __builtin_amdgcn_udot4(w, P.hw[2][d], a2, false);
          }
        } else {
#pragma unroll
          for (int d = 0; d < 12; ++d) {
            const uint32_t w = wsrc[d];
            a0 = __builtin_amdgcn_udot4(w, P.hw[0][d], a0, false); a1 = __builtin_amdgcn_udot4(w, P.hw[1][d], a1, false); a2 = __builtin_amdgcn_udot4(w, P.hw[2][d], a2, false);
          }
        }
        dst[0] = (uint8_t)(a0 >> sh); dst[1] = (uint8_t)(a1 >> sh); dst[2] = (uint8_t)(a2 >> sh);
      }
    } else if (P.ow != P.W && P.kx <= 9) {
      // fast path (e.g. 640 -> 160: 8 taps): the 27 bytes of the 9-tap window come from 8 dword LDS reads,
      // are byte-aligned with v_alignbyte and multiplied out with 24-bit integer MADs
      for (int i = tid; i < nr * P.ow; i += OB) {
        const int rr = i / P.ow, ox = i % P.ow;
        const int x0 = s_bx[2 * ox];
        const int32_t* k = s_kx + ox * 9;
        const uint32_t* wsrc = reinterpret_cast<const uint32_t*>(s_row + rr * in_row_words * 4);
        const int b0 = x0 * 3, w0 = b0 >> 2;
        const uint32_t sh = (uint32_t)(b0 & 3);
        uint32_t w[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) w[q] = wsrc[w0 + q];               // may run past the row: those taps are 0
        uint32_t a[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) a[q] = __builtin_amdgcn_alignbyte(w[q + 1], w[q], sh);   // bytes b0+4q .. b0+4q+3
        int32_t acc[3] = {1 << (PREC - 1), 1 << (PREC - 1), 1 << (PREC - 1)};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int32_t kt = k[t];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const int b = 3 * t + c;
            acc[c] += (int32_t)__umul24((a[b >> 2] >> (8 * (b & 3))) & 255u, (uint32_t)kt);
          }
        }
        uint8_t* dst = s_tmp + (size_t)(r0 + rr) * tmp_row_bytes + ox * 3;
        dst[0] = (uint8_t)clip8(acc[0]); dst[1] = (uint8_t)clip8(acc[1]); dst[2] = (uint8_t)clip8(acc[2]);
      }
    } else {
      for (int i = tid; i < nr * P.ow; i += OB) {
        const int rr = i / P.ow, ox = i % P.ow;
        const uint8_t* src = s_row + rr * in_row_words * 4;
        uint8_t* dst = s_tmp + (size_t)(r0 + rr) * tmp_row_bytes + ox * 3;
        if (P.ow == P.W) { dst[0] = src[ox * 3]; dst[1] = src[ox * 3 + 1]; dst[2] = src[ox * 3 + 2]; continue; }
        const int x0 = P.bx[2 * ox], n = P.bx[2 * ox + 1];
        const int32_t* k = P.kkx + ox * P.kx;
        int32_t a0 = 1 << (PREC - 1), a1 = a0, a2 = a0;
        const uint8_t* p = src + x0 * 3;
        for (int t = 0; t < n; ++t) {
          const int32_t kt = k[t];
          a0 += (int32_t)p[3 * t] * kt; a1 += (int32_t)p[3 * t + 1] * kt; a2 += (int32_t)p[3 * t + 2] * kt;
        }
        dst[0] = (uint8_t)clip8(a0); dst[1] = (uint8_t)clip8(a1); dst[2] = (uint8_t)clip8(a2);
      }
    }
    __syncthreads();
  }

  observe_vertical(P, s_tmp, e, oy0, oy1, y_first, tid);
}

// ---- power-of-two down-scaling on both axes (640 x 480 -> 160 x 120, 80 x 60, 160 x 240 ...): no staging at all ----------
// Away from the borders every output pixel has the same small-integer taps (ObserveParams::hw / vw).  A thread owns R = 4
// vertically adjacent output pixels of one column: it walks the (R + 1) SY input rows they need once, filters each row's
// window with three v_dot4_u32_u8 chains straight from global memory (HN aligned dwords; neighbouring threads read
// neighbouring windows, the rows are shared through L1 / L2), rounds to the uint8 intermediate Pillow keeps, and adds it to
// the one or two outputs the row belongs to.  The frame is read from HBM once; nothing is synchronised.  Border rows /
// columns (clipped, renormalised taps) are k_observe_border's.
template <int HN, int SY>
__global__ __launch_bounds__(OB) void k_observe_pow2(ObserveParams P) {
  constexpr int R = 4;
  const int wi = P.ow - 2, G = (P.oh - 2 + R - 1) / R;
  const int per_env = G * wi;
  const int idx = blockIdx.x * OB + threadIdx.x;
  const int e = idx / per_env, rem = idx - e * per_env;
  if (e >= P.N) return;
  const int g = rem / wi, ox = 1 + (rem - g * wi);
  const int oy0 = 1 + g * R;
  const int SX = P.hfast;
  const int in_row_bytes = P.W * 3;
  const uint8_t* col = P.frames + (size_t)e * P.H * in_row_bytes + (3 * SX * ox + P.hoff);
  const int ybase = SY * oy0 - SY / 2;
  uint32_t acc[R][3];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[r][c] = 1u << (P.vsh - 1);
  const uint32_t hhalf = 1u << (P.hsh - 1);
#pragma unroll
  for (int t = 0; t < (R + 1) * SY; ++t) {
    const int y = min(ybase + t, P.H - 1);             // rows past the frame only feed outputs that are not stored
    const uint32_t* wsrc = reinterpret_cast<const uint32_t*>(col + (size_t)y * in_row_bytes);
    uint32_t a0 = hhalf, a1 = hhalf, a2 = hhalf;
#pragma unroll
    for (int d = 0; d < HN; ++d) {
      const uint32_t w = wsrc[d];
      a0 = __builtin_amdgcn_udot4(w, P.hw[0][d], a0, false); a1 = __builtin_amdgcn_udot4(w, P.hw[1][d], a1, false); a2 = __builtin_amdgcn_udot4(w, P.hw[2][d], a2, false);
    }
    a0 >>= P.hsh; a1 >>= P.hsh; a2 >>= P.hsh;         // Pillow's uint8 intermediate
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int tt = t - SY * r;                       // compile-time: which tap of output r this row is
      if (tt >= 0 && tt < 2 * SY) {
        acc[r][0] = __umul24(a0, P.vw[tt]) + acc[r][0]; acc[r][1] = __umul24(a1, P.vw[tt]) + acc[r][1]; acc[r][2] = __umul24(a2, P.vw[tt]) + acc[r][2];
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int oy = oy0 + r;
    if (oy > P.oh - 2) break;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const uint32_t v = acc[r][c] >> P.vsh;
      const size_t o = P.chw ? (((size_t)e * 3 + c) * P.oh + oy) * P.ow + ox : (((size_t)e * P.oh + oy) * P.ow + ox) * 3 + c;
      if (P.f32) reinterpret_cast<float*>(P.out)[o] = (float)v / 255.0f;
      else reinterpret_cast<uint8_t*>(P.out)[o] = (uint8_t)v;
    }
  }
}

// the border pixels of the same output (first / last row and column): Pillow's two passes per pixel from the tables
__global__ __launch_bounds__(OB) void k_observe_border(ObserveParams P) {
  const int nb = 2 * P.ow + 2 * (P.oh - 2);
  const int idx = blockIdx.x * OB + threadIdx.x;
  const int e = idx / nb, b = idx - e * nb;
  if (e >= P.N) return;
  int oy, ox;
  if (b < P.ow) { oy = 0; ox = b; }
  else if (b < 2 * P.ow) { oy = P.oh - 1; ox = b - P.ow; }
  else { const int q = b - 2 * P.ow; oy = 1 + (q >> 1); ox = (q & 1) ? P.ow - 1 : 0; }
  const int in_row_bytes = P.W * 3;
  const uint8_t* frame = P.frames + (size_t)e * P.H * in_row_bytes;
  const int x0 = P.bx[2 * ox], nx = P.bx[2 * ox + 1], y0 = P.by[2 * oy], ny = P.by[2 * oy + 1];
  const int32_t* kx = P.kkx + ox * P.kx;
  const int32_t* ky = P.kky + oy * P.ky;
  int32_t acc[3] = {1 << (PREC - 1), 1 << (PREC - 1), 1 << (PREC - 1)};
  for (int t = 0; t < ny; ++t) {
    const uint8_t* p = frame + (size_t)(y0 + t) * in_row_bytes + x0 * 3;
    int32_t a[3] = {1 << (PREC - 1), 1 << (PREC - 1), 1 << (PREC - 1)};
    for (int j = 0; j < nx; ++j) {
      const int32_t kj = kx[j];
      a[0] += (int32_t)p[3 * j] * kj; a[1] += (int32_t)p[3 * j + 1] * kj; a[2] += (int32_t)p[3 * j + 2] * kj;
    }
    const int32_t kt = ky[t];
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[c] += (int32_t)clip8(a[c]) * kt;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const uint32_t v = clip8(acc[c]);
    const size_t o = P.chw ? (((size_t)e * 3 + c) * P.oh + oy) * P.ow + ox : (((size_t)e * P.oh + oy) * P.ow + ox) * 3 + c;
    if (P.f32) reinterpret_cast<float*>(P.out)[o] = (float)v / 255.0f;
    else reinterpret_cast<uint8_t*>(P.out)[o] = (uint8_t)v;
  }
}

// ---- OpenCV INTER_CUBIC (the reference's own ResizeWrapper, wrappers.py:129-138; dtsim/resample.py cubic_coeffs) ----
// One workgroup per (env, output row).  The four source rows of the output row (replicated at the borders) are combined
// FIRST, vertically: thread t sums the same dword of the four rows with the row's four 11-bit taps, byte by byte, into
// int32 -- coalesced dword loads, every needed frame byte read once -- and leaves the 3 W sums in LDS; then thread ox sums
// four of them per channel with the column's taps.  OpenCV filters rows first and columns second, keeping int32 rows
// without rounding: both orders are the same exact integer sum (|sum| < 2^31 for 8-bit pixels), and the single rounding
// is saturate_cast<uchar>((v + 2^21) >> 22).
__global__ __launch_bounds__(OB) void k_observe_cubic(ObserveParams P) {
  extern __shared__ uint32_t s_mem[];
  int32_t* s_v = reinterpret_cast<int32_t*>(s_mem);          // [W * 3] vertical sums
  const int tid = threadIdx.x;
  const int e = blockIdx.x / P.oh, oy = blockIdx.x % P.oh;
  const int in_row_bytes = P.W * 3;
  const uint8_t* frame = P.frames + (size_t)e * P.H * in_row_bytes;
  const int y0 = P.by[oy];                               // first of the four rows (may be < 0)
  const int32_t* ky = P.kky + 4 * oy;
  const int32_t k0 = ky[0], k1 = ky[1], k2 = ky[2], k3 = ky[3];
  const uint8_t* r0 = frame + (size_t)min(max(y0, 0), P.H - 1) * in_row_bytes;
  const uint8_t* r1 = frame + (size_t)min(max(y0 + 1, 0), P.H - 1) * in_row_bytes;
  const uint8_t* r2 = frame + (size_t)min(max(y0 + 2, 0), P.H - 1) * in_row_bytes;
  const uint8_t* r3 = frame + (size_t)min(max(y0 + 3, 0), P.H - 1) * in_row_bytes;
  if ((in_row_bytes & 3) == 0) {
    for (int i = tid; i < in_row_bytes / 4; i += OB) {
      const uint32_t a = reinterpret_cast<const uint32_t*>(r0)[i], b = reinterpret_cast<const uint32_t*>(r1)[i];
      const uint32_t c = reinterpret_cast<const uint32_t*>(r2)[i], d = reinterpret_cast<const uint32_t*>(r3)[i];
      int4 v;
      v.x = (int32_t)(a & 255u) * k0 + (int32_t)(b & 255u) * k1 + (int32_t)(c & 255u) * k2 + (int32_t)(d & 255u) * k3;
      v.y = (int32_t)((a >> 8) & 255u) * k0 + (int32_t)((b >> 8) & 255u) * k1 + (int32_t)((c >> 8) & 255u) * k2 + (int32_t)((d >> 8) & 255u) * k3;
      v.z = (int32_t)((a >> 16) & 255u) * k0 + (int32_t)((b >> 16) & 255u) * k1 + (int32_t)((c >> 16) & 255u) * k2 + (int32_t)((d >> 16) & 255u) * k3;
      v.w = (int32_t)(a >> 24) * k0 + (int32_t)(b >> 24) * k1 + (int32_t)(c >> 24) * k2 + (int32_t)(d >> 24) * k3;
      reinterpret_cast<int4*>(s_v)[i] = v;
    }
  } else {
    for (int i = tid; i < in_row_bytes; i += OB) s_v[i] = (int32_t)r0[i] * k0 + (int32_t)r1[i] * k1 + (int32_t)r2[i] * k2 + (int32_t)r3[i] * k3;
  }
  __syncthreads();
  for (int ox = tid; ox < P.ow; ox += OB) {
    const int x0 = P.bx[ox];
    const int32_t* kx = P.kkx + 4 * ox;
    int32_t acc[3] = {0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int x = min(max(x0 + j, 0), P.W - 1);
      const int32_t kj = kx[j];
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[c] += s_v[x * 3 + c] * kj;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const uint32_t v = (uint32_t)min(max((acc[c] + (1 << 21)) >> 22, 0), 255);
      const size_t o = P.chw ? (((size_t)e * 3 + c) * P.oh + oy) * P.ow + ox : (((size_t)e * P.oh + oy) * P.ow + ox) * 3 + c;
      if (P.f32) reinterpret_cast<float*>(P.out)[o] = (float)v / 255.0f;
      else reinterpret_cast<uint8_t*>(P.out)[o] = (uint8_t)v;
    }
  }
}

}  // namespace

void dt_launch_observe_cubic(hipStream_t s, const ObserveParams& P) {
  hipLaunchKernelGGL(k_observe_cubic, dim3((unsigned)((size_t)P.N * P.oh)), dim3(OB), (size_t)P.W * 3 * sizeof(int32_t) + 16, s, P);
}

size_t dt_observe_lds_bytes(const ObserveParams& P) {
  const size_t in_row_words = ((size_t)P.W * 3 + 3) >> 2;
  const size_t tabs = (P.ow != P.W && P.kx <= 9) ? (size_t)P.ow * (2 + 9) * 4 : 0;
  return OBS_STAGE_ROWS * in_row_words * 4 + 32 + (((size_t)P.max_rows_in * P.ow * 3 + 3) & ~(size_t)3) + tabs + 16;
}

void dt_launch_observe(hipStream_t s, const ObserveParams& P) {
  if (P.hfast && P.vfast && P.ow >= 3 && P.oh >= 3 && !getenv("DTSIM_OBSERVE_STAGED")) {   // power-of-two scale on both axes
    const int G = (P.oh - 2 + 3) / 4;
    const size_t n_in = (size_t)P.N * G * (P.ow - 2), n_b = (size_t)P.N * (2 * P.ow + 2 * (P.oh - 2));
    const dim3 grid((unsigned)((n_in + OB - 1) / OB)), gridb((unsigned)((n_b + OB - 1) / OB));
    bool done = true;
#define DT_POW2(HN_, SY_) hipLaunchKernelGGL((k_observe_pow2<HN_, SY_>), grid, dim3(OB), 0, s, P)
    if (P.hn == 7 && P.vfast == 2) DT_POW2(7, 2); else if (P.hn == 7 && P.vfast == 4) DT_POW2(7, 4); else if (P.hn == 7 && P.vfast == 8) DT_POW2(7, 8);
    else if (P.hn == 12 && P.vfast == 2) DT_POW2(12, 2); else if (P.hn == 12 && P.vfast == 4) DT_POW2(12, 4); else if (P.hn == 12 && P.vfast == 8) DT_POW2(12, 8);
    else done = false;
#undef DT_POW2
    if (done) { hipLaunchKernelGGL(k_observe_border, gridb, dim3(OB), 0, s, P); return; }
  }
  const int n_blocks = (P.oh + P.rows_per_block - 1) / P.rows_per_block;
  hipLaunchKernelGGL(k_observe, dim3((unsigned)((size_t)P.N * n_blocks)), dim3(OB), dt_observe_lds_bytes(P), s, P);
}
