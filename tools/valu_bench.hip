// Calibration microbenchmark (not product): VALU issue cost per instruction type on gfx950, measured
// as wave-instructions per SIMD-cycle with 1..8 waves per SIMD (256 blocks x k of 256 threads).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32;
typedef unsigned long long u64;

#define BODY8(stmt) stmt stmt stmt stmt stmt stmt stmt stmt
template <int OP>
__global__ __launch_bounds__(256) void k(u32* out, int iters) {
  u32 a = threadIdx.x, b = a * 7 + 1, c = a ^ 0x55, d = a + 9;
  u64 x = ((u64) a << 32) | b, y = ((u64) c << 32) | d;
  u32 acc = 0;
  for (int i = 0; i < iters; i++) {
    if (OP == 0) { BODY8(asm volatile ("v_add_u32 %0, %1, %0\n v_add_u32 %2, %3, %2\n v_xor_b32 %1, %0, %1\n v_xor_b32 %3, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
    if (OP == 1) { BODY8(asm volatile ("v_cmp_lt_u64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_u64 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc" : "+v"(x), "+v"(y), "+v"(a), "+v"(b) :: "vcc");) }
    if (OP == 2) { BODY8(asm volatile ("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_u32 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "vcc");) }
    if (OP == 3) { BODY8(asm volatile ("v_mul_lo_u32 %0, %1, %0\n v_mul_lo_u32 %2, %3, %2\n v_mul_lo_u32 %1, %0, %1\n v_mul_lo_u32 %3, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
    if (OP == 4) { BODY8(asm volatile ("v_mad_u32_u24 %0, %1, %0, %2\n v_mad_u32_u24 %2, %3, %2, %0\n v_mad_u32_u24 %1, %0, %1, %3\n v_mad_u32_u24 %3, %2, %3, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
    if (OP == 5) { BODY8(asm volatile ("v_lshl_add_u32 %0, %1, 2, %0\n v_lshl_add_u32 %2, %3, 2, %2\n v_lshl_add_u32 %1, %0, 1, %1\n v_lshl_add_u32 %3, %2, 1, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
    if (OP == 6) { BODY8(asm volatile ("v_lshlrev_b64 %0, 3, %0\n v_lshrrev_b64 %1, 1, %1\n v_lshlrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 3, %1" : "+v"(x), "+v"(y));) }
    if (OP == 7) { BODY8(asm volatile ("v_mbcnt_lo_u32_b32 %0, %1, %0\n v_mbcnt_hi_u32_b32 %2, %3, %2\n v_bcnt_u32_b32 %1, %0, %1\n v_bcnt_u32_b32 %3, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
    if (OP == 8) { BODY8(asm volatile ("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:2 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %2 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
    if (OP == 9) { BODY8(asm volatile ("v_mul_hi_u32 %0, %1, %0\n v_mul_hi_u32 %2, %3, %2\n v_mul_hi_u32 %1, %0, %1\n v_mul_hi_u32 %3, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
    if (OP == 10) { BODY8(asm volatile ("v_add_co_u32 %0, vcc, %1, %0\n v_addc_co_u32 %2, vcc, %3, %2, vcc\n v_add_co_u32 %1, vcc, %0, %1\n v_addc_co_u32 %3, vcc, %2, %3, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "vcc");) }
    if (OP == 11) { BODY8(asm volatile ("v_readfirstlane_b32 s20, %0\n v_add_u32 %1, s20, %1\n v_readfirstlane_b32 s21, %2\n v_add_u32 %3, s21, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "s20", "s21");) }
  }
  acc = a ^ b ^ c ^ d ^ (u32) x ^ (u32) (x >> 32) ^ (u32) y ^ (u32) (y >> 32);
  if (acc == 0x12345) out[0] = acc;
}

template <int OP> void run (const char *name, u32 *out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("%-28s", name);
  for (int bpc = 1; bpc <= 8; bpc *= 2) {
    const int iters = 4000;
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<OP>, dim3(256 * bpc), dim3(256), 0, 0, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    const double wave_instr = 256.0 * bpc * 4 * (double) iters * 32;
    printf("  w/SIMD %d: %5.2f cyc/instr", bpc, 1.0 / (wave_instr / ms / 1e6 / 1024 / 2.4));
  }
  printf("\n");
}
int main() {
  u32* out; hipMalloc(&out, 4);
  run<0>("add/xor u32", out);
  run<1>("cmp_lt_u64 + cndmask", out);
  run<2>("cmp_lt_u32 + cndmask", out);
  run<3>("mul_lo_u32", out);
  run<4>("mad_u32_u24", out);
  run<5>("lshl_add_u32", out);
  run<6>("lsh{l,r}rev_b64", out);
  run<7>("mbcnt/bcnt", out);
  run<8>("mov_dpp", out);
  run<9>("mul_hi_u32", out);
  run<10>("add_co/addc_co", out);
  run<11>("readfirstlane + add(sgpr)", out);
  return 0;
}
