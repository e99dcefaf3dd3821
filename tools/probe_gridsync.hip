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

// hand-written barrier: one agent-scope atomic add per workgroup, one lane polls; spins are bounded so that a
// launch whose workgroups are not all resident ends (with a wrong count) instead of hanging
__global__ void atomic_barrier_loop(int iters, unsigned* cnt, double* out, unsigned* failed) {
  double acc = threadIdx.x;
  const unsigned nb = gridDim.x;
  for (int i = 0; i < iters; ++i) {
    acc = acc * 1.0000001 + 1.0;
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      const unsigned v = atomicAdd(cnt, 1u);
      const unsigned target = (v / nb + 1u) * nb;
      unsigned spins = 0;
      while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++spins > 2000000u) {
          atomicAdd(failed, 1u);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      __threadfence();
    }
    __syncthreads();
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
  unsigned* c;
  hipMalloc(&c, 8);
  for (int blocks : {16, 64, 256}) {
    int iters = 2000;
    hipMemset(c, 0, 8);
    void* args[] = {&iters, &c, &d, (void*)nullptr};
    unsigned* f = c + 1;
    args[3] = &f;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchCooperativeKernel((void*)atomic_barrier_loop, dim3(blocks), dim3(256), args, 0, 0);
    hipDeviceSynchronize();
    hipMemset(c, 0, 8);
    hipEventRecord(a);
    hipLaunchCooperativeKernel((void*)atomic_barrier_loop, dim3(blocks), dim3(256), args, 0, 0);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    unsigned h[2];
    hipMemcpy(h, c, 8, hipMemcpyDeviceToHost);
    printf("atomic barrier, blocks %4d: %.2f us per barrier (count %u, timeouts %u)\n", blocks, ms * 1e3 / iters, h[0], h[1]);
  }
  return 0;
}
