// micro-benchmark 6: issue cost of the VALU opcodes considered for the round-2 raster (quad-layout texels,
// integer dot filter, packed fp32 geometry, SDWA / perm address arithmetic), one opcode per kernel, 8 independent
// chains per lane, WAVES waves per SIMD (argv[1], default 4 and 8 both run).  Reports cycles per wave-instruction per
// SIMD at the clock derived from a v_mov reference run is NOT attempted: the figure printed is G wave-instr/s and the
// ratio to v_fma_f32; DVFS moves both.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define ITER 2048
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

#define OPS(X) \
  X(0, "v_fma_f32", "v_fma_f32 %0, %0, %3, %4", F) \
  X(1, "v_mul_f32", "v_mul_f32 %0, %0, %3", F) \
  X(2, "v_add_f32", "v_add_f32 %0, %0, %3", F) \
  X(3, "v_fmac_f32", "v_fmac_f32 %0, %3, %4", F) \
  X(4, "v_pk_fma_f32", "v_pk_fma_f32 %2, %2, %5, %5", P) \
  X(5, "v_pk_mul_f32", "v_pk_mul_f32 %2, %2, %5", P) \
  X(6, "v_cvt_flr_i32_f32", "v_cvt_flr_i32_f32 %1, %0", FU) \
  X(7, "v_fract_f32", "v_fract_f32 %0, %0", F) \
  X(8, "v_floor_f32", "v_floor_f32 %0, %0", F) \
  X(9, "v_cvt_i32_f32", "v_cvt_i32_f32 %1, %0", FU) \
  X(10, "v_cvt_f32_u32", "v_cvt_f32_u32 %0, %1", UF) \
  X(11, "v_med3_f32", "v_med3_f32 %0, %0, %3, %4", F) \
  X(12, "v_med3_i32", "v_med3_i32 %1, %1, %6, %7", U) \
  X(13, "v_max_f32", "v_max_f32 %0, %0, %3", F) \
  X(14, "v_and_b32", "v_and_b32 %1, %1, %6", U) \
  X(15, "v_or_b32", "v_or_b32 %1, %1, %6", U) \
  X(16, "v_xor_b32", "v_xor_b32 %1, %1, %6", U) \
  X(17, "v_lshlrev_b32", "v_lshlrev_b32 %1, 1, %1", U) \
  X(18, "v_ashrrev_i32", "v_ashrrev_i32 %1, 1, %1", U) \
  X(19, "v_bfe_u32", "v_bfe_u32 %1, %1, 1, 30", U) \
  X(20, "v_bfi_b32", "v_bfi_b32 %1, %6, %1, %7", U) \
  X(21, "v_lshl_or_b32", "v_lshl_or_b32 %1, %1, 1, %6", U) \
  X(22, "v_lshl_add_u32", "v_lshl_add_u32 %1, %1, 1, %6", U) \
  X(23, "v_and_or_b32", "v_and_or_b32 %1, %1, %6, %7", U) \
  X(24, "v_add3_u32", "v_add3_u32 %1, %1, %6, %7", U) \
  X(25, "v_add_u32", "v_add_u32 %1, %1, %6", U) \
  X(26, "v_sub_u32", "v_sub_u32 %1, %1, %6", U) \
  X(27, "v_mad_u32_u24", "v_mad_u32_u24 %1, %1, %6, %7", U) \
  X(28, "v_mul_u32_u24", "v_mul_u32_u24 %1, %1, %6", U) \
  X(29, "v_mad_i32_i24", "v_mad_i32_i24 %1, %1, %6, %7", U) \
  X(30, "v_mul_lo_u32", "v_mul_lo_u32 %1, %1, %6", U) \
  X(31, "v_perm_b32", "v_perm_b32 %1, %1, %6, %7", U) \
  X(32, "v_dot4_u32_u8", "v_dot4_u32_u8 %1, %6, %7, %1", U) \
  X(33, "v_dot4_i32_i8", "v_dot4_i32_i8 %1, %6, %7, %1", U) \
  X(34, "v_dot2_u32_u16", "v_dot2_u32_u16 %1, %6, %7, %1", U) \
  X(35, "v_cvt_pknorm_u16_f32", "v_cvt_pknorm_u16_f32 %1, %0, %3", FU) \
  X(36, "v_cvt_pk_u8_f32", "v_cvt_pk_u8_f32 %1, %0, 1, %1", FU) \
  X(37, "v_cvt_f32_ubyte0", "v_cvt_f32_ubyte0 %0, %1", UF) \
  X(38, "v_cndmask_b32 (sgpr mask)", "v_cndmask_b32_e64 %1, %1, %6, s[10:11]", U) \
  X(39, "v_cmp_lt_f32 (vcc)", "v_cmp_lt_f32 vcc, %0, %3", FC) \
  X(40, "v_cmp_gt_u32 (vcc)", "v_cmp_gt_u32 vcc, %1, %6", UC) \
  X(41, "v_cmp_gt_u32_e64 (sgpr)", "v_cmp_gt_u32_e64 s[10:11], %1, %6", UC) \
  X(42, "v_mul_u32_u24_sdwa b1", "v_mul_u32_u24_sdwa %1, %1, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD", U) \
  X(43, "v_add_u32_sdwa b1", "v_add_u32_sdwa %1, %1, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1", U) \
  X(44, "v_cmp_gt_u32_sdwa w0", "v_cmp_gt_u32_sdwa vcc, %1, %6 src0_sel:WORD_0 src1_sel:DWORD", UC) \
  X(45, "v_mov_b32", "v_mov_b32 %1, %6", U) \
  X(46, "v_mov_b32 dpp row_shr:1", "v_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf", U) \
  X(47, "v_cvt_f32_ubyte1 sdwa-free", "v_cvt_f32_ubyte1 %0, %1", UF) \
  X(48, "v_rcp_f32", "v_rcp_f32 %0, %0", F) \
  X(49, "v_sub_f32", "v_sub_f32 %0, %3, %0", F) \
  X(50, "v_pk_add_f32", "v_pk_add_f32 %2, %2, %5", P) \
  X(51, "v_lshrrev_b32", "v_lshrrev_b32 %1, 1, %1", U) \
  X(52, "v_alignbit_b32", "v_alignbit_b32 %1, %1, %6, 8", U) \
  X(53, "v_pk_fma_f16", "v_pk_fma_f16 %1, %1, %6, %7", U) \
  X(54, "v_cvt_pkrtz_f16_f32", "v_cvt_pkrtz_f16_f32 %1, %0, %3", FU) \
  X(55, "v_mad_u64_u32", "v_mad_u64_u32 %2, vcc, %1, %6, %2", P) \
  X(56, "v_cvt_u32_f32", "v_cvt_u32_f32 %1, %0", FU) \
  X(57, "v_max3_f32", "v_max3_f32 %0, %0, %3, %4", F) \
  X(58, "v_dot2_f32_f16", "v_dot2_f32_f16 %0, %6, %7, %0", F) \
  X(59, "v_sad_u8", "v_sad_u8 %1, %1, %6, %7", U) \
  X(60, "v_lerp_u8", "v_lerp_u8 %1, %1, %6, %7", U)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, unsigned m, unsigned n) {
  float x[8]; unsigned u[8]; double p[8];
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x + i; u[i] = threadIdx.x * (2 * i + 1); p[i] = threadIdx.x + 0.5 * i; }
  double pc = (double)a;
  asm volatile("s_mov_b64 s[10:11], exec" ::: "s10", "s11");
  for (int it = 0; it < ITER; ++it) {
#define X(id, name, text, kind) \
    if (MODE == id) { _Pragma("unroll") for (int i = 0; i < 8; ++i) \
      asm volatile(text : "+v"(x[i]), "+v"(u[i]), "+v"(p[i]) : "v"(a), "v"(b), "v"(pc), "v"(m), "v"(n) : "vcc", "s10", "s11"); }
    OPS(X)
#undef X
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += x[i] + (float)u[i] + (float)p[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static double g_base[3];
template <int MODE> void run(const char* name, int wps, int slot) {
  float* d; hipMalloc(&d, 256 * 8 * 1024 * 4);
  const int blocks = 256 * wps;   // 256 CUs x wps workgroups of 4 waves = wps waves per SIMD
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<blocks, 256>>>(d, 1.0001f, 0.5f, 0x01020304u, 0x00070003u); hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(a); k<MODE><<<blocks, 256>>>(d, 1.0001f, 0.5f, 0x01020304u, 0x00070003u); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
  }
  const double winstr = (double)blocks * 4 * ITER * 8;
  if (MODE == 0) g_base[slot] = best;
  printf("%-30s wps=%d %.3f ms %7.1f G wave-instr/s  x%.2f of v_fma_f32\n", name, wps, best, winstr / best / 1e6, best / g_base[slot]);
  hipFree(d);
}

int main(int argc, char** argv) {
  const int wlist[2] = {4, 8};
  for (int w = 0; w < 2; ++w) {
#define X(id, name, text, kind) run<id>(name, wlist[w], w);
    OPS(X)
#undef X
  }
  return 0;
}
