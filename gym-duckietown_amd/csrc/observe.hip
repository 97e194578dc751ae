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

namespace {

#define OB 256                 // threads per workgroup
#define OBS_STAGE_ROWS 4       // input rows staged in LDS per horizontal step
#define PREC 22                // Pillow PRECISION_BITS (32 - 8 - 2)

__device__ inline uint32_t clip8(int32_t acc) {
  const int32_t v = acc >> PREC;
  return (uint32_t)min(max(v, 0), 255);
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
  uint8_t* s_row = reinterpret_cast<uint8_t*>(s_mem);                               // [OBS_STAGE_ROWS][in_row_words * 4]
  uint8_t* s_tmp = s_row + (size_t)OBS_STAGE_ROWS * in_row_words * 4;               // [rows_in][ow * 3]
  const uint8_t* frame = P.frames + (size_t)e * P.H * in_row_bytes;
  const bool aligned = (in_row_bytes & 3) == 0;

  // ---- horizontal pass (skipped when the width is unchanged: Pillow then resamples rows only)
  for (int r0 = 0; r0 < rows_in; r0 += OBS_STAGE_ROWS) {
    const int nr = min(OBS_STAGE_ROWS, rows_in - r0);
    if (aligned) {                                   // coalesced dword loads of whole rows
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
    __syncthreads();
  }

  // ---- vertical pass + layout / normalisation
  const int n_out = (oy1 - oy0) * tmp_row_bytes;
  for (int i = tid; i < n_out; i += OB) {
    const int oy = oy0 + i / tmp_row_bytes, j = i % tmp_row_bytes;
    uint32_t v;
    if (P.oh == P.H) v = s_tmp[(size_t)(oy - y_first) * tmp_row_bytes + j];
    else {
      const int y0 = P.by[2 * oy], n = P.by[2 * oy + 1];
      const int32_t* k = P.kky + oy * P.ky;
      const uint8_t* p = s_tmp + (size_t)(y0 - y_first) * tmp_row_bytes + j;
      int32_t a = 1 << (PREC - 1);
      for (int t = 0; t < n; ++t) a += (int32_t)p[(size_t)t * tmp_row_bytes] * k[t];
      v = clip8(a);
    }
    const int ox = j / 3, c = j % 3;
    const size_t o = P.chw ? (((size_t)e * 3 + c) * P.oh + oy) * P.ow + ox : ((size_t)e * P.oh + oy) * tmp_row_bytes + j;
    if (P.f32) reinterpret_cast<float*>(P.out)[o] = (float)v / 255.0f;     // NormalizeWrapper: (obs - 0) / (255 - 0)
    else reinterpret_cast<uint8_t*>(P.out)[o] = (uint8_t)v;
  }
}

}  // namespace

size_t dt_observe_lds_bytes(const ObserveParams& P) {
  const size_t in_row_words = ((size_t)P.W * 3 + 3) >> 2;
  return OBS_STAGE_ROWS * in_row_words * 4 + (size_t)P.max_rows_in * P.ow * 3 + 16;
}

void dt_launch_observe(hipStream_t s, const ObserveParams& P) {
  const int n_blocks = (P.oh + P.rows_per_block - 1) / P.rows_per_block;
  hipLaunchKernelGGL(k_observe, dim3((unsigned)((size_t)P.N * n_blocks)), dim3(OB), dt_observe_lds_bytes(P), s, P);
}
