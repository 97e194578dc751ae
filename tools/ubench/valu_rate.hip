// micro-benchmark: issue rate of scalar vs packed fp32 VALU and a few other ops on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITER 4096
template <int MODE>
__global__ void k(float* out, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
  f2 pa = {a, a}, pb = {b, b};
  unsigned u0 = threadIdx.x, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7;
  for (int i = 0; i < ITER; ++i) {
    if (MODE == 0) {  // 8 independent scalar fma
      x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
      x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
    } else if (MODE == 1) {  // 4 packed fma (= 8 fma)
      p0 = __builtin_elementwise_fma(p0, pa, pb); p1 = __builtin_elementwise_fma(p1, pa, pb);
      p2 = __builtin_elementwise_fma(p2, pa, pb); p3 = __builtin_elementwise_fma(p3, pa, pb);
    } else if (MODE == 2) {  // 8 cvt ubyte
      asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(x0) : "v"(u0)); asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(x1) : "v"(u1));
      asm volatile("v_cvt_f32_ubyte2 %0, %1" : "=v"(x2) : "v"(u2)); asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(x3) : "v"(u3));
      asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(x4) : "v"(u0)); asm volatile("v_cvt_f32_ubyte2 %0, %1" : "=v"(x5) : "v"(u1));
      asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(x6) : "v"(u2)); asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(x7) : "v"(u3));
      u0 += 1; u1 += 1;
    } else if (MODE == 3) {  // 8 fract
      x0 = __builtin_amdgcn_fractf(x0 + a); x1 = __builtin_amdgcn_fractf(x1 + a); x2 = __builtin_amdgcn_fractf(x2 + a); x3 = __builtin_amdgcn_fractf(x3 + a);
      x4 = __builtin_amdgcn_fractf(x4 + a); x5 = __builtin_amdgcn_fractf(x5 + a); x6 = __builtin_amdgcn_fractf(x6 + a); x7 = __builtin_amdgcn_fractf(x7 + a);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + u0 + u1;
}
template <int MODE> void run(const char* name, int ops_per_iter, int waves_per_simd) {
  float* d; hipMalloc(&d, 256 * 4 * 2048 * 4);
  const int blocks = 256 * waves_per_simd;  // 256 CUs x (4 SIMDs: block of 256 threads = 4 waves = 1 per SIMD)
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<blocks, 256>>>(d, 1.0001f, 0.5f); hipDeviceSynchronize();
  hipEventRecord(a); k<MODE><<<blocks, 256>>>(d, 1.0001f, 0.5f); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double winstr = (double)blocks * 4 * ITER * ops_per_iter;   // wave-instructions
  printf("%-28s waves/SIMD=%d: %.3f ms  %.1f G wave-instr/s  => %.2f cycles/instr/SIMD @2.4GHz\n", name, waves_per_simd, ms,
         winstr / ms / 1e6, 1024.0 * 2.4e9 / (winstr / (ms * 1e-3)));
  hipFree(d);
}
int main() {
  for (int w : {1, 2, 4}) {
    run<0>("v_fma_f32 x8", 8, w);
    run<1>("v_pk_fma_f32 x4 (=8 fma)", 4, w);
    run<2>("v_cvt_f32_ubyteN x8 (+2 add)", 10, w);
    run<3>("v_add+v_fract x8", 16, w);
  }
  return 0;
}
