// Micro-benchmark 2: issue cost of many gfx950 VALU opcodes at 4 waves per SIMD (8 independent chains per lane).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define OPS(X) \
  X(0, "v_mov_b32 %0, %1", "v_mov_b32") \
  X(1, "v_and_b32 %0, %0, %1", "v_and_b32") \
  X(2, "v_or_b32 %0, %0, %1", "v_or_b32") \
  X(3, "v_xor_b32 %0, %0, %1", "v_xor_b32") \
  X(4, "v_add_u32 %0, %0, %1", "v_add_u32") \
  X(5, "v_sub_u32 %0, %0, %1", "v_sub_u32") \
  X(6, "v_lshrrev_b32 %0, 3, %0", "v_lshrrev_b32") \
  X(7, "v_lshlrev_b32 %0, 3, %0", "v_lshlrev_b32") \
  X(8, "v_min_u32 %0, %0, %1", "v_min_u32") \
  X(9, "v_max_u32 %0, %0, %1", "v_max_u32") \
  X(10, "v_bfe_u32 %0, %0, 1, 31", "v_bfe_u32") \
  X(11, "v_bfi_b32 %0, %1, %0, %1", "v_bfi_b32") \
  X(12, "v_alignbit_b32 %0, %0, %1, 7", "v_alignbit_b32") \
  X(13, "v_and_or_b32 %0, %0, %1, %1", "v_and_or_b32") \
  X(14, "v_lshl_add_u32 %0, %0, 2, %1", "v_lshl_add_u32") \
  X(15, "v_lshl_or_b32 %0, %0, 2, %1", "v_lshl_or_b32") \
  X(16, "v_add3_u32 %0, %0, %1, %1", "v_add3_u32") \
  X(17, "v_or3_b32 %0, %0, %1, %1", "v_or3_b32") \
  X(18, "v_min3_u32 %0, %0, %1, %1", "v_min3_u32") \
  X(19, "v_xad_u32 %0, %0, %1, %1", "v_xad_u32") \
  X(20, "v_cndmask_b32 %0, %0, %1, vcc", "v_cndmask_b32") \
  X(21, "v_cmp_ne_u32 vcc, %0, %1", "v_cmp_ne_u32(vcc)") \
  X(22, "v_cmp_ne_u32 s[20:21], %0, %1", "v_cmp_ne_u32(sgpr)") \
  X(23, "v_cmpx_ne_u32 %0, %1", "v_cmpx_ne_u32") \
  X(24, "v_perm_b32 %0, %0, %1, %1", "v_perm_b32") \
  X(25, "v_mad_u32_u24 %0, %0, %1, %1", "v_mad_u32_u24") \
  X(26, "v_mul_u32_u24 %0, %0, %1", "v_mul_u32_u24") \
  X(27, "v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD", "v_and_b32_sdwa") \
  X(28, "v_xor_b32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf", "v_xor_b32_dpp") \
  X(29, "v_not_b32 %0, %0", "v_not_b32") \
  X(30, "v_bfrev_b32 %0, %0", "v_bfrev_b32") \
  X(31, "v_xor_b32 %0, 0x12345, %0", "v_xor_b32(lit)") \
  X(32, "v_xor_b32 %0, s22, %0", "v_xor_b32(sgpr)") \
  X(33, "v_sad_u32 %0, %0, %1, %1", "v_sad_u32") \
  X(34, "v_pk_add_u16 %0, %0, %1", "v_pk_add_u16") \
  X(35, "v_pk_min_u16 %0, %0, %1", "v_pk_min_u16") \
  X(36, "v_cmp_eq_u32_sdwa s[20:21], %0, %1 src0_sel:BYTE_0 src1_sel:DWORD", "v_cmp_eq_u32_sdwa") \
  X(37, "v_ashrrev_i32 %0, 3, %0", "v_ashrrev_i32") \
  X(38, "v_subrev_u32 %0, %0, %1", "v_subrev_u32") \
  X(39, "v_mbcnt_lo_u32_b32 %0, %0, %1", "v_mbcnt_lo") \
  X(40, "v_readlane_b32 s22, %0, 5", "v_readlane_b32")

template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters) {
  unsigned a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x + 1;
  unsigned c = out[0] + 0x9e3779b9u;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#define X(N, S, NAME) if (KIND == N) asm volatile(S : "+v"(a[i]) : "v"(c) : "vcc", "s20", "s21", "s22");
        OPS(X)
#undef X
      }
    }
  }
  unsigned s = 0;
  for (int i = 0; i < 8; ++i) s ^= a[i];
  if (s == 0x12345678u) out[1] = s;
}

template <int KIND>
int run(const char* name, unsigned* d) {
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const int iters = 2000, blocks = 256 * 4;
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 10);
  CHK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters);
  CHK(hipEventRecord(e1));
  CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  const double winstr = (double)blocks * 4 * iters * 64;
  const double per_simd_per_s = winstr / (ms * 1e-3) / 1024;
  printf("%-22s %.3f ms  %.2f cycles per wave-instr (at 2.4 GHz, 4 waves/SIMD)\n", name, ms, 2.4e9 / per_simd_per_s);
  return 0;
}

int main() {
  unsigned* d; CHK(hipMalloc(&d, 64)); CHK(hipMemset(d, 0, 64));
#define X(N, S, NAME) run<N>(NAME, d);
  OPS(X)
#undef X
  return 0;
}
