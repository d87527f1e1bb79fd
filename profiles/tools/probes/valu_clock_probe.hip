// What clock do the SIMDs run at under sustained vector-issue load?  1024 waves per... every SIMD of the chip holds 4 waves
// that each issue N vector instructions (a dependent chain per wave: 4 waves cover the 4-cycle issue interval), timed with HIP
// events: instructions x 4 cycles / (SIMDs x seconds) = effective clock.  Also times the chain with ONE wave per SIMD.
//   hipcc --offload-arch=gfx950 -O2 valu_clock_probe.hip -o /tmp/valu_clock_probe && /tmp/valu_clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(1024) spin(unsigned *out, int iters) {
  unsigned a = threadIdx.x, b = blockIdx.x + 1;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 64; ++k) { a = a * 1u + b; b = b ^ a; }   // v_add / v_xor chain: 128 vector instructions per trip
  }
  if (a == 0x12345u) out[0] = b;
}
int main() {
  unsigned *d; hipMalloc(&d, 4);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int waves_per_simd : {4, 1}) {
    const int threads = 64 * 4 * waves_per_simd, iters = 20000;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(s);
      for (int l = 0; l < 10; ++l) hipLaunchKernelGGL(spin, dim3(cus), dim3(threads), 0, 0, d, iters);
      hipEventRecord(e); hipEventSynchronize(e);
      float ms; hipEventElapsedTime(&ms, s, e);
      const double insts = 10.0 * cus * 4 * waves_per_simd * (double)iters * 128;       // wave-instructions
      printf("%d waves/SIMD: %.3f ms for %.3g vector wave-instructions -> %.0f MHz if each costs 4 cycles of its SIMD (clockRate says %d MHz)\n",
             waves_per_simd, ms, insts, insts * 4 / (cus * 4.0) / (ms * 1e-3) / 1e6, p.clockRate / 1000);
    }
  }
  return 0;
}
