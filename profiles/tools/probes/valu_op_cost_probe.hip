// What does one wave64 vector instruction of each kind cost a gfx950 SIMD?  Every SIMD of the chip holds W waves (4, then 1) that
// each run a dependent chain of ONE opcode (inline asm, 64 per loop trip); HIP events around 5 launches give SIMD-cycles per
// instruction at the clock hipDeviceProp reports.  The walk kernels' instruction budgets are priced with this table.
//   hipcc --offload-arch=gfx950 -O2 valu_op_cost_probe.hip -o /tmp/valu_op_cost_probe && /tmp/valu_op_cost_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHAIN32(NAME, ASM)                                                                     \
  __global__ void __launch_bounds__(1024) NAME(unsigned *out, int iters) {                     \
    uint32_t a = threadIdx.x, b = blockIdx.x + 1, c = 0x01020304u;                             \
    for (int i = 0; i < iters; ++i) {                                                          \
      _Pragma("unroll") for (int k = 0; k < 64; ++k) asm volatile(ASM : "+v"(a) : "v"(b), "v"(c)); \
    }                                                                                          \
    if (a == 0x12345u) out[0] = a;                                                             \
  }
#define CHAIN64(NAME, ASM)                                                                     \
  __global__ void __launch_bounds__(1024) NAME(unsigned *out, int iters) {                     \
    uint64_t a = threadIdx.x; uint32_t b = (blockIdx.x & 7) + 1, c = 3;                         \
    for (int i = 0; i < iters; ++i) {                                                          \
      _Pragma("unroll") for (int k = 0; k < 64; ++k) asm volatile(ASM : "+v"(a) : "v"(b), "v"(c)); \
    }                                                                                          \
    if (a == 0x12345u) out[0] = (unsigned)a;                                                   \
  }
CHAIN32(k_add, "v_add_u32 %0, %0, %1")
CHAIN32(k_xor, "v_xor_b32 %0, %0, %1")
CHAIN32(k_and, "v_and_b32 %0, %0, %1")
CHAIN32(k_or, "v_or_b32 %0, %0, %1")
CHAIN32(k_sub, "v_sub_u32 %0, %0, %1")
CHAIN32(k_mov, "v_mov_b32 %0, %0")
CHAIN32(k_not, "v_not_b32 %0, %0")
CHAIN32(k_cnd, "v_cndmask_b32 %0, %0, %1, vcc")
CHAIN32(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
CHAIN32(k_lshr, "v_lshrrev_b32 %0, %1, %0")
CHAIN32(k_addxor, "v_add_u32 %0, %0, %1\n v_lshlrev_b32 %0, %2, %0")
CHAIN32(k_lshl, "v_lshlrev_b32 %0, %1, %0")
CHAIN32(k_andor, "v_and_or_b32 %0, %0, %1, %2")
CHAIN32(k_or3, "v_or3_b32 %0, %0, %1, %2")
CHAIN32(k_add3, "v_add3_u32 %0, %0, %1, %2")
CHAIN32(k_lshlor, "v_lshl_or_b32 %0, %0, %1, %2")
CHAIN32(k_bfe, "v_bfe_u32 %0, %0, %1, %2")
CHAIN32(k_bcnt, "v_bcnt_u32_b32 %0, %0, %1")
CHAIN32(k_ffbl, "v_ffbl_b32 %0, %0")
CHAIN32(k_perm, "v_perm_b32 %0, %0, %1, %2")
CHAIN32(k_alignbyte, "v_alignbyte_b32 %0, %0, %1, %2")
CHAIN32(k_mullo, "v_mul_lo_u32 %0, %0, %1")
CHAIN32(k_mulhi, "v_mul_hi_u32 %0, %0, %1")
CHAIN32(k_mul24, "v_mul_u32_u24 %0, %0, %1")
CHAIN32(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2")
CHAIN32(k_min, "v_min_u32 %0, %0, %1")
CHAIN32(k_cmp_cnd, "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
CHAIN32(k_mov_dpp, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
CHAIN32(k_readlane, "v_readfirstlane_b32 s20, %0\n v_add_u32 %0, s20, %1")
CHAIN32(k_sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD")
CHAIN64(k_lshl64, "v_lshlrev_b64 %0, %1, %0")
CHAIN64(k_lshr64, "v_lshrrev_b64 %0, %1, %0")
CHAIN64(k_mad64, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
typedef void (*K)(unsigned *, int);
int main() {
  unsigned *d; hipMalloc(&d, 4);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  struct { const char *name; K k; int per; } ops[] = {
    {"v_add_u32", k_add, 1}, {"v_xor_b32", k_xor, 1}, {"v_and_b32", k_and, 1}, {"v_or_b32", k_or, 1}, {"v_sub_u32", k_sub, 1}, {"v_mov_b32", k_mov, 1}, {"v_not_b32", k_not, 1},
    {"v_cndmask_b32 (vcc)", k_cnd, 1}, {"v_bitop3_b32", k_bitop3, 1}, {"v_lshrrev_b32", k_lshr, 1}, {"v_add + v_lshlrev (pair)", k_addxor, 1}, {"v_lshlrev_b32", k_lshl, 1}, {"v_and_or_b32", k_andor, 1}, {"v_or3_b32", k_or3, 1},
    {"v_add3_u32", k_add3, 1}, {"v_lshl_or_b32", k_lshlor, 1}, {"v_bfe_u32", k_bfe, 1}, {"v_bcnt_u32_b32", k_bcnt, 1}, {"v_ffbl_b32", k_ffbl, 1},
    {"v_perm_b32", k_perm, 1}, {"v_alignbyte_b32", k_alignbyte, 1}, {"v_mul_lo_u32", k_mullo, 1}, {"v_mul_hi_u32", k_mulhi, 1},
    {"v_mul_u32_u24", k_mul24, 1}, {"v_mad_u32_u24", k_mad24, 1}, {"v_min_u32", k_min, 1}, {"v_cmp + v_cndmask (pair)", k_cmp_cnd, 1},
    {"v_mov_b32 dpp", k_mov_dpp, 1}, {"v_readfirstlane + v_add (pair)", k_readlane, 1}, {"v_add_u32 sdwa", k_sdwa, 1},
    {"v_lshlrev_b64", k_lshl64, 1}, {"v_lshrrev_b64", k_lshr64, 1}, {"v_mad_u64_u32", k_mad64, 1}};
  printf("%d CUs, clockRate %d MHz; SIMD-cycles per wave64 instruction (or pair) in a dependent chain\n", cus, p.clockRate / 1000);
  printf("%-34s %12s %12s\n", "opcode", "4 waves/SIMD", "1 wave/SIMD");
  for (auto &op : ops) {
    double cyc[2];
    int wi = 0;
    for (int waves_per_simd : {4, 1}) {
      const int threads = 64 * 4 * waves_per_simd, iters = 4000;
      hipLaunchKernelGGL(op.k, dim3(cus), dim3(threads), 0, 0, d, 10);
      hipEventRecord(s);
      for (int l = 0; l < 5; ++l) hipLaunchKernelGGL(op.k, dim3(cus), dim3(threads), 0, 0, d, iters);
      hipEventRecord(e); hipEventSynchronize(e);
      float ms; hipEventElapsedTime(&ms, s, e);
      const double per_simd = 5.0 * waves_per_simd * (double)iters * 64;        // instructions (or pairs) issued on one SIMD
      cyc[wi++] = (ms * 1e-3) * (p.clockRate * 1e3) / per_simd;
    }
    printf("%-34s %12.2f %12.2f\n", op.name, cyc[0], cyc[1]);
  }
  return 0;
}
