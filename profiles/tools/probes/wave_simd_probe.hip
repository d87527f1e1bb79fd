// Which SIMD does wave w of a 1024-thread workgroup land on?  (observed placement, for speed only: HIP promises nothing)
// hipcc --offload-arch=gfx950 -O2 wave_simd_probe.hip -o /tmp/wave_simd_probe && /tmp/wave_simd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(1024) probe(unsigned *out) {
  extern __shared__ unsigned char smem[];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[(blockIdx.x * 16 + w) * 2] = hw;
    out[(blockIdx.x * 16 + w) * 2 + 1] = xcc;
    smem[w * 10240] = (unsigned char)w;   // touch the wave's slice
  }
  // keep the workgroup alive a little so that all 256 are resident together
  for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);
}
int main() {
  const int nb = 256;
  unsigned *d; hipMalloc(&d, nb * 16 * 2 * 4);
  hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(probe, dim3(nb), dim3(1024), 160 * 1024, 0, d);
  hipError_t e = hipDeviceSynchronize();
  printf("launch: %s\n", hipGetErrorString(e));
  std::vector<unsigned> h(nb * 16 * 2);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
  int hist[16][4] = {};
  for (int b = 0; b < nb; ++b)
    for (int w = 0; w < 16; ++w) hist[w][(h[(b * 16 + w) * 2] >> 4) & 3]++;
  for (int w = 0; w < 16; ++w) printf("wave %2d: simd0 %3d simd1 %3d simd2 %3d simd3 %3d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
  for (int b = 0; b < 4; ++b) {
    printf("block %d:", b);
    for (int w = 0; w < 16; ++w) { unsigned v = h[(b * 16 + w) * 2]; printf(" w%d->simd%u/slot%u cu%u se%u xcc%u |", w, (v >> 4) & 3, v & 15, (v >> 8) & 15, (v >> 13) & 7, h[(b * 16 + w) * 2 + 1] & 15); }
    printf("\n");
  }
  return 0;
}
