// Barrier + data exchange between 16 workgroups, (a) spread over the XCDs with agent-scope operations, (b) confined to
// ONE XCD (launch 16 * 8 workgroups, only blockIdx.x % 8 == 0 take part) with workgroup-scope operations, which on
// gfx942 / gfx950 carry sc0: coherent at the XCD's L2, not beyond.  Prints the XCC ids seen, the time per exchange and
// whether every exchange delivered the right data.   hipcc --offload-arch=gfx950 -O3 -o probe_xcd_sync probe_xcd_sync.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xfu; }

template <int SCOPE>
__global__ __launch_bounds__(256) void exchange_loop(int iters, int stride, unsigned* bar, double* slots, unsigned* xcc,
                                                     unsigned* bad) {
  if (blockIdx.x % stride != 0) return;
  const int wg = blockIdx.x / stride, G = gridDim.x / stride;
  if (threadIdx.x == 0) xcc[wg] = xcc_id();
  unsigned epoch = 0;
  __shared__ double got[64];
  for (int it = 0; it < iters; ++it) {
    double* slot = slots + (size_t)(it & 1) * 64;
    if (threadIdx.x == 0) __hip_atomic_store(&slot[wg], (double)(it * 100 + wg), __ATOMIC_RELAXED, SCOPE);
    __syncthreads();
    ++epoch;
    if (threadIdx.x == 0) {
      if (SCOPE == __HIP_MEMORY_SCOPE_AGENT) __threadfence();
      else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, SCOPE);
      unsigned spins = 0;
      while (__hip_atomic_load(bar, __ATOMIC_RELAXED, SCOPE) < epoch * (unsigned)G) {
        if (++spins > 2000000u) {
          atomicAdd(bad + 1, 1u);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      if (SCOPE == __HIP_MEMORY_SCOPE_AGENT) __threadfence();
      else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    __syncthreads();
    if (threadIdx.x < G) got[threadIdx.x] = __hip_atomic_load(&slot[threadIdx.x], __ATOMIC_RELAXED, SCOPE);
    __syncthreads();
    if (threadIdx.x < G && got[threadIdx.x] != (double)(it * 100 + (int)threadIdx.x)) atomicAdd(bad, 1u);
  }
}

int main() {
  unsigned *bar, *xcc, *bad;
  double* slots;
  hipMalloc(&bar, 64); hipMalloc(&xcc, 256); hipMalloc(&bad, 8); hipMalloc(&slots, 2 * 64 * 8);
  const int iters = 20000;
  for (int mode = 0; mode < 3; ++mode) {   // 0: all XCDs / agent, 1: one XCD / workgroup scope, 2: one XCD / agent scope
    for (int G : {16, 32}) {
      const int stride = mode ? 8 : 1;
      for (int rep = 0; rep < 2; ++rep) {
        hipMemset(bar, 0, 64); hipMemset(bad, 0, 8); hipMemset(xcc, 0xff, 256);
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        if (mode == 1) hipLaunchKernelGGL(exchange_loop<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(G * stride), dim3(256), 0, 0, iters, stride, bar, slots, xcc, bad);
        else hipLaunchKernelGGL(exchange_loop<__HIP_MEMORY_SCOPE_AGENT>, dim3(G * stride), dim3(256), 0, 0, iters, stride, bar, slots, xcc, bad);
        hipEventRecord(b);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        unsigned hx[64], hb[2];
        hipMemcpy(hx, xcc, 256, hipMemcpyDeviceToHost);
        hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost);
        if (rep) {
          printf("%s G=%2d: %.2f us per exchange, wrong values %u, barrier timeouts %u, xcc ids:", mode == 1 ? "one XCD, workgroup scope" : (mode == 2 ? "one XCD, agent scope    " : "all XCDs, agent scope   "), G, ms * 1e3 / iters, hb[0], hb[1]);
          for (int i = 0; i < G; ++i) printf(" %u", hx[i]);
          printf("\n");
        }
      }
    }
  }
  return 0;
}
