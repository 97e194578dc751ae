// micro-benchmark 5: the integer dot / pack ops of the fixed-point filter experiment (4 waves/SIMD, 8 chains)
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
template <int MODE>
__global__ void k(float* out, float a, float b, unsigned m) {
  float x[8]; unsigned u[8];
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x + i; u[i] = threadIdx.x * (2 * i + 1); }
  for (int it = 0; it < ITER; ++it) {
#define OP(i) \
    if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b)); \
    else if (MODE == 1) asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(u[i]) : "v"(m), "v"(u[(i+1)&7])); \
    else if (MODE == 2) asm volatile("v_cvt_pknorm_u16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(x[i]), "v"(a)); \
    else if (MODE == 3) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(u[i]) : "v"(m), "v"(u[(i+1)&7])); \
    else if (MODE == 4) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(m), "v"(u[(i+1)&7])); \
    else if (MODE == 5) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(u[i]) : "v"(m), "v"(u[(i+1)&7])); \
    else if (MODE == 6) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(u[i]) : "v"(m), "v"(u[(i+1)&7])); \
    else if (MODE == 7) asm volatile("v_cvt_f32_ubyte2 %0, %1" : "=v"(x[i]) : "v"(u[i]));
    REP8(OP)
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += x[i] + u[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name) {
  float* d; (void)hipMalloc(&d, 256 * 4 * 1024 * 4);
  const int blocks = 256 * 4;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int w = 0; w < 3; ++w) k<MODE><<<blocks, 256>>>(d, 1.0001f, 0.5f, 0x3c003c00u);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a); k<MODE><<<blocks, 256>>>(d, 1.0001f, 0.5f, 0x3c003c00u); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  printf("%-26s %.3f ms  %7.1f G wave-instr/s\n", name, ms, (double)blocks * 4 * ITER * 8 / ms / 1e6);
  (void)hipFree(d);
}
int main() {
  run<0>("v_fma_f32"); run<1>("v_dot2_u32_u16"); run<5>("v_dot2_i32_i16"); run<3>("v_dot4_u32_u8"); run<2>("v_cvt_pknorm_u16_f32");
  run<4>("v_perm_b32"); run<6>("v_mad_u32_u24"); run<7>("v_cvt_f32_ubyte2");
  return 0;
}
