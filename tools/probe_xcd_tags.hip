// Follow-up of probe_xcd_sync.hip: the barrier-free exchange the LU panel wanted.  Every workgroup publishes a 16-double
// line whose element 0 is a tag written AFTER the payload (payload by lanes 1..15 of wave 1, s_waitcnt, workgroup
// barrier, tag by lane 0 of wave 0); every workgroup spins on all tags and then reads all payloads.  READ: 0 = plain
// workgroup-scope atomic load, 1 = fetch-or 0 (a read-modify-write executes at the L2).  One XCD (stride 8) or all.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int SCOPE, int READ>
__device__ __forceinline__ double ld(const double* p) {
  if (READ) {
    unsigned long long* q = reinterpret_cast<unsigned long long*>(const_cast<double*>(p));
    return __builtin_bit_cast(double, __hip_atomic_fetch_or(q, 0ull, __ATOMIC_RELAXED, SCOPE));
  }
  return __hip_atomic_load(p, __ATOMIC_RELAXED, SCOPE);
}

template <int SCOPE, int READ>
__global__ __launch_bounds__(1024) void tag_loop(int iters, int stride, double* rec, unsigned* bad, unsigned long long seq) {
  if (blockIdx.x % stride != 0) return;
  const int wg = blockIdx.x / stride, G = gridDim.x / stride;
  const int tid = threadIdx.x;
  __shared__ int s_dead;
  if (tid == 0) s_dead = 0;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    double* slot = rec + (size_t)(it & 1) * 64 * 16;
    const double expect = __builtin_bit_cast(double, (seq << 20) | (unsigned long long)(it + 1));
    if (tid >= 65 && tid < 80) {
      __hip_atomic_store(&slot[wg * 16 + (tid - 64)], (double)(it * 1000 + wg * 16 + (tid - 64)), __ATOMIC_RELAXED, SCOPE);
      if (SCOPE == __HIP_MEMORY_SCOPE_AGENT) __threadfence();
      else __builtin_amdgcn_s_waitcnt(0);
    }
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&slot[wg * 16], expect, __ATOMIC_RELAXED, SCOPE);
    if (tid < G) {
      unsigned spins = 0;
      while (ld<SCOPE, READ>(&slot[tid * 16]) != expect) {
        if (++spins > 1000000u) {
          atomicAdd(bad + 1, 1u);
          s_dead = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (s_dead) return;
    if (tid < G * 16 && (tid & 15) != 0) {
      const double v = ld<SCOPE, READ>(&slot[tid]);
      if (v != (double)(it * 1000 + tid)) atomicAdd(bad, 1u);
    }
    __syncthreads();
  }
}

int main() {
  unsigned* bad;
  double* rec;
  hipMalloc(&bad, 8);
  hipMalloc(&rec, 2 * 64 * 16 * 8);
  const int iters = 20000;
  unsigned long long seq = 0;
  for (int mode = 0; mode < 3; ++mode) {   // 0: all XCDs agent; 1: one XCD workgroup-scope plain loads; 2: one XCD, RMW reads
    for (int G : {16, 32}) {
      const int stride = mode ? 8 : 1;
      for (int rep = 0; rep < 2; ++rep) {
        hipMemset(bad, 0, 8);
        hipMemset(rec, 0, 2 * 64 * 16 * 8);
        ++seq;
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        if (mode == 0) hipLaunchKernelGGL((tag_loop<__HIP_MEMORY_SCOPE_AGENT, 0>), dim3(G * stride), dim3(1024), 0, 0, iters, stride, rec, bad, seq);
        else if (mode == 1) hipLaunchKernelGGL((tag_loop<__HIP_MEMORY_SCOPE_WORKGROUP, 0>), dim3(G * stride), dim3(1024), 0, 0, iters, stride, rec, bad, seq);
        else hipLaunchKernelGGL((tag_loop<__HIP_MEMORY_SCOPE_WORKGROUP, 1>), dim3(G * stride), dim3(1024), 0, 0, iters, stride, rec, bad, seq);
        hipEventRecord(b);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        unsigned hb[2];
        hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost);
        if (rep) printf("mode %d G=%2d: %.2f us per exchange, wrong payloads %u, spin timeouts %u\n", mode, G, ms * 1e3 / iters, hb[0], hb[1]);
      }
    }
  }
  return 0;
}
