// micro-benchmark 4: cost of lane-mask selects on gfx950 (4 waves/SIMD, 8 independent chains)
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
template <int MODE>
__global__ void k(float* out, float a, float b, unsigned m) {
  float x[8]; unsigned u[8];
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x + i; u[i] = threadIdx.x * (2 * i + 1); }
  unsigned long long sm = __ballot(threadIdx.x & 1);
  for (int it = 0; it < ITER; ++it) {
#define OP(i) \
    if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b)); \
    else if (MODE == 1) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(m)); \
    else if (MODE == 2) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[i]) : "v"(m), "s"(sm)); \
    else if (MODE == 3) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(u[i]) : "v"(x[i]), "v"(a), "v"(m) : "vcc"); \
    else if (MODE == 4) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(u[i]) : "v"(m), "v"(u[(i+1)&7])); \
    else if (MODE == 5) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(x[i]), "v"(a) : "vcc"); \
    else if (MODE == 6) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(sm) : "v"(x[i]), "v"(a)); \
    else if (MODE == 7) asm volatile("v_cmp_lt_f32_e64 %0, %2, %3\n v_cndmask_b32_e64 %1, %1, %4, %0" : "=&s"(sm), "+v"(u[i]) : "v"(x[i]), "v"(a), "v"(m)); \
    else if (MODE == 9) asm volatile("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(u[i]) : "v"(m), "s"(sm));
    REP8(OP)
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += x[i] + u[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)sm;
}
template <int MODE> void run(const char* name, int n) {
  float* d; (void)hipMalloc(&d, 256 * 4 * 1024 * 4);
  const int blocks = 256 * 4;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int w = 0; w < 3; ++w) k<MODE><<<blocks, 256>>>(d, 1.0001f, 0.5f, 0x3c003c00u);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a); k<MODE><<<blocks, 256>>>(d, 1.0001f, 0.5f, 0x3c003c00u); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  double winstr = (double)blocks * 4 * ITER * 8 * n;
  printf("%-40s %.3f ms  %7.1f G wave-instr/s\n", name, ms, winstr / ms / 1e6);
  (void)hipFree(d);
}
int main() {
  run<0>("v_fma_f32", 1); run<0>("v_fma_f32 (again)", 1);
  run<1>("v_cndmask_b32 e32 vcc", 1); run<2>("v_cndmask_b32_e64 sgpr mask", 1);
  run<9>("v_cndmask_b32_e64 0, v, mask (no dep)", 1);
  run<5>("v_cmp_lt_f32 vcc", 1); run<6>("v_cmp_lt_f32_e64 sgpr", 1);
  run<3>("v_cmp vcc + v_cndmask vcc (2 instr)", 2); run<7>("v_cmp_e64 + v_cndmask_e64 (2 instr)", 2);
  run<4>("v_bfi_b32", 1);
  return 0;
}
