// Cost of a grid-wide barrier (cooperative launch) on MI355X: is a single persistent kernel for the small-n fit
// (16 leaves + ~80 dependent block steps) cheaper than the chain of ~95 launches it would replace?
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ void sync_loop(int iters, double* out) {
  cg::grid_group g = cg::this_grid();
  double acc = threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    acc = acc * 1.0000001 + 1.0;
    g.sync();
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = acc;
}

int main() {
  double* d;
  hipMalloc(&d, 8);
  for (int blocks : {16, 64, 256, 512}) {
    for (int threads : {64, 256}) {
      int iters = 2000;
      void* args[] = {&iters, &d};
      hipEvent_t a, b;
      hipEventCreate(&a);
      hipEventCreate(&b);
      hipLaunchCooperativeKernel((void*)sync_loop, dim3(blocks), dim3(threads), args, 0, 0);
      hipDeviceSynchronize();
      hipEventRecord(a);
      hipError_t e = hipLaunchCooperativeKernel((void*)sync_loop, dim3(blocks), dim3(threads), args, 0, 0);
      hipEventRecord(b);
      hipDeviceSynchronize();
      float ms = 0;
      hipEventElapsedTime(&ms, a, b);
      printf("blocks %4d threads %3d: %s  %.2f us per grid sync\n", blocks, threads, hipGetErrorString(e), ms * 1e3 / iters);
    }
  }
  return 0;
}
