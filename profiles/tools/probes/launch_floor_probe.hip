// What does a kernel boundary cost?  An empty kernel of G workgroups x 256 threads launched N times back to back on one stream
// (HIP events around the N launches), then the same N launches captured once into a hipGraph and replayed.  The small-batch
// configurations (12 k molecules: 1,500 workgroups) are priced against this floor in DESIGN §4.
//   hipcc --offload-arch=gfx950 -O2 launch_floor_probe.hip -o /tmp/launch_floor_probe && /tmp/launch_floor_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) empty_kernel(int *p) { if (p && threadIdx.x == 9999) p[0] = 1; }
__global__ void __launch_bounds__(256) touch_kernel(int *p, int n) {   // one 4-byte store per thread: a kernel that dirties lines
  const int i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = i;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  int *d; CK(hipMalloc(&d, 1500 * 256 * 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t s, e; CK(hipEventCreate(&s)); CK(hipEventCreate(&e));
  const int N = 2000;
  for (int grid : {1, 256, 1500, 6000}) {
    for (int which = 0; which < 2; ++which) {
      auto launch = [&]() { if (which == 0) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, st, d); else hipLaunchKernelGGL(touch_kernel, dim3(grid), dim3(256), 0, st, d, 1500 * 256); };
      for (int i = 0; i < 50; ++i) launch();
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(s, st));
      for (int i = 0; i < N; ++i) launch();
      CK(hipEventRecord(e, st)); CK(hipEventSynchronize(e));
      float ms; CK(hipEventElapsedTime(&ms, s, e));
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
      for (int i = 0; i < 100; ++i) launch();
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
      CK(hipEventRecord(s, st));
      for (int i = 0; i < N / 100; ++i) CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e, st)); CK(hipEventSynchronize(e));
      float gms; CK(hipEventElapsedTime(&gms, s, e));
      printf("%-6s kernel, %4d workgroups: %.2f us per launch on the stream, %.2f us inside a 100-node hipGraph\n", which ? "store" : "empty", grid, ms * 1e3 / N, gms * 1e3 / N);
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
  }
  return 0;
}
