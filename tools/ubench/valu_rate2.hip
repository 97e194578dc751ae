// micro-benchmark 2: issue cost of assorted VALU ops on gfx950 (4 waves/SIMD, 8 independent chains)
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
template <int MODE>
__global__ void k(float* out, float a, float b, unsigned m) {
  float x[8]; unsigned u[8];
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x + i; u[i] = threadIdx.x * (2 * i + 1); }
  unsigned long long w[8]; for (int i = 0; i < 8; ++i) w[i] = (unsigned long long)out + i;
  for (int it = 0; it < ITER; ++it) {
#define OP(i) \
    if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b)); \
    else if (MODE == 1) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(u[i]) : "v"(m), "v"(m)); \
    else if (MODE == 2) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(u[i]) : "v"(x[i])); \
    else if (MODE == 3) asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(u[i]) : "v"(x[i])); \
    else if (MODE == 4) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[i]) : "v"(m)); \
    else if (MODE == 5) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(w[i]) : "v"(w[(i+1)&7])); \
    else if (MODE == 6) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(m), "v"(u[(i+1)&7])); \
    else if (MODE == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(m)); \
    else if (MODE == 8) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b)); \
    else if (MODE == 9) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(u[i]) : "v"(m), "v"(u[(i+1)&7])); \
    else if (MODE == 10) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u[i]) : "v"(m), "v"(m)); \
    else if (MODE == 11) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(x[i]) : "v"(u[i])); \
    else if (MODE == 12) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(m)); \
    else if (MODE == 13) asm volatile("v_fract_f32 %0, %0" : "+v"(x[i])); \
    else if (MODE == 14) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(x[i]) : "v"(u[i])); \
    else if (MODE == 15) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a)); \
    else if (MODE == 16) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(x[i]), "v"(a) : "vcc"); \
    else if (MODE == 17) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(u[i]) : "v"(m)); \
    else if (MODE == 18) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %1" : "=v"(u[i]) : "v"(x[i])); \
    else if (MODE == 19) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(x[i]) : "v"(m), "v"(u[i])); \
    else if (MODE == 20) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(x[i]) : "v"(m), "v"(u[i])); \
    else if (MODE == 21) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b)); \
    else if (MODE == 22) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(x[i]) : "v"(a)); \
    else if (MODE == 23) asm volatile("v_add_u32 %0, %1, %0" : "+v"(u[i]) : "v"(m));
    REP8(OP)
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += x[i] + u[i] + (float)w[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name) {
  float* d; hipMalloc(&d, 256 * 4 * 1024 * 4);
  const int blocks = 256 * 4;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<blocks, 256>>>(d, 1.0001f, 0.5f, 0x3c003c00u); hipDeviceSynchronize();
  hipEventRecord(a); k<MODE><<<blocks, 256>>>(d, 1.0001f, 0.5f, 0x3c003c00u); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double winstr = (double)blocks * 4 * ITER * 8;
  printf("%-22s %.3f ms  %7.1f G wave-instr/s  rel. to v_fma: ", name, ms, winstr / ms / 1e6);
  static double base = 0; if (MODE == 0) base = ms; printf("%.2fx\n", ms / base);
  hipFree(d);
}
int main() {
  run<0>("v_fma_f32"); run<15>("v_mul_f32"); run<12>("v_and_b32"); run<13>("v_fract_f32"); run<8>("v_min3_f32"); run<7>("v_cndmask_b32");
  run<16>("v_cmp_lt_f32"); run<14>("v_cvt_f32_ubyte1"); run<3>("v_cvt_flr_i32_f32"); run<2>("v_cvt_pk_u8_f32"); run<11>("v_cvt_f32_f16");
  run<18>("v_cvt_pkrtz_f16_f32"); run<1>("v_pk_fma_f16"); run<17>("v_pk_mul_f16"); run<4>("v_mul_u32_u24"); run<10>("v_mad_u32_u24");
  run<5>("v_lshl_add_u64"); run<6>("v_perm_b32"); run<9>("v_dot4_u32_u8");
  run<19>("v_dot2_f32_f16"); run<20>("v_dot2c_f32_f16"); run<21>("v_fmac_f32"); run<22>("v_sub_f32"); run<23>("v_add_u32");
  return 0;
}
