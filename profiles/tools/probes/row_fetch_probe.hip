// How many bytes does FETCH_SIZE report for SCATTERED row reads of R bytes (every lane its own random row of a table far
// larger than L2 + Infinity Cache)?  The guide calibrates the counter for wide streaming reads only (FETCH_SIZE = RDREQ x 64 B,
// a 128-byte request counted as 64): the adjacency-row reads of sent_blane_kernel (8 W bytes per lane and step) and the
// 16-byte token pieces of the lane kernels are this other shape.  Build + run under rocprofv3 --pmc FETCH_SIZE:
//   hipcc --offload-arch=gfx950 -O2 row_fetch_probe.hip -o /tmp/row_fetch_probe
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- /tmp/row_fetch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int R>   // R = 8, 16, 32, 64, 128 bytes per lane, R-byte aligned
__global__ void __launch_bounds__(256) rows(const uint8_t *__restrict__ table, uint64_t nrows, uint32_t *__restrict__ sink, int reps) {
  uint64_t x = (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
  uint32_t acc = 0;
  for (int i = 0; i < reps; ++i) {
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    const uint8_t *p = table + (x % nrows) * R;
    if (R == 8) { const uint2 v = *reinterpret_cast<const uint2 *>(p); acc += v.x ^ v.y; }
    else {
#pragma unroll
      for (int k = 0; k < R / 16; ++k) { const uint4 v = *reinterpret_cast<const uint4 *>(p + 16 * k); acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int R>
void run(const uint8_t *t, size_t bytes, uint32_t *sink, const char *name) {
  const int blocks = 256 * 8, threads = 256, reps = 64;
  hipLaunchKernelGGL(rows<R>, dim3(blocks), dim3(threads), 0, 0, t, (uint64_t)(bytes / R), sink, reps);
  hipDeviceSynchronize();
  printf("%s: %d lanes x %d reads x %d B = %.1f MB requested\n", name, blocks * threads, reps, R, (double)blocks * threads * reps * R / 1e6);
}

int main() {
  const size_t bytes = (size_t)4 << 30;   // 4 GiB: every read misses L2 and the 256 MiB Infinity Cache
  uint8_t *t; uint32_t *sink;
  if (hipMalloc(&t, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(t, 1, bytes);
  hipDeviceSynchronize();
  run<8>(t, bytes, sink, "rows<8>");
  run<16>(t, bytes, sink, "rows<16>");
  run<32>(t, bytes, sink, "rows<32>");
  run<64>(t, bytes, sink, "rows<64>");
  run<128>(t, bytes, sink, "rows<128>");
  return 0;
}
