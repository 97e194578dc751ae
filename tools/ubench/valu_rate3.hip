// micro-benchmark 3: packed-fp32 VALU ops on gfx950 (4 waves/SIMD, 8 independent chains)
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, float a, float b) {
  float x[8]; f2 p[8];
  const f2 pa = {a, a * 1.0001f}, pb = {b, b * 0.5f};
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x + i; p[i] = f2{(float)threadIdx.x + i, (float)i}; }
  for (int it = 0; it < ITER; ++it) {
#define OP(i) \
    if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b)); \
    else if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pa), "v"(pb)); \
    else if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pa)); \
    else if (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pa)); \
    else if (MODE == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a)); \
    else if (MODE == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(p[i]) : "v"(pa), "v"(pb));
    REP8(OP)
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += x[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int per) {
  float* d; hipMalloc(&d, 256 * 4 * 1024 * 4);
  const int blocks = 256 * 4;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<blocks, 256>>>(d, 1.0001f, 0.5f); hipDeviceSynchronize();
  hipEventRecord(a); k<MODE><<<blocks, 256>>>(d, 1.0001f, 0.5f); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double winstr = (double)blocks * 4 * ITER * 8;
  printf("%-26s %.3f ms  %7.1f G wave-instr/s  %7.1f G lane-op-waves/s\n", name, ms, winstr / ms / 1e6, per * winstr / ms / 1e6);
  hipFree(d);
}
int main() {
  run<0>("v_fma_f32", 1); run<4>("v_mul_f32", 1); run<1>("v_pk_fma_f32", 2); run<2>("v_pk_mul_f32", 2); run<3>("v_pk_add_f32", 2);
  run<5>("v_pk_fma_f32 (bcast src1/2)", 2);
  return 0;
}
