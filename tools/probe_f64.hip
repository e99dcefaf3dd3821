// Micro-probe: sustained v_mfma_f64_16x16x4_f64 and v_fma_f64 rates + streaming copy bandwidth on
// the device at hand.  Used once to ground the roofline peaks quoted in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void mfma_probe(double* out, int iters, double a0, double b0) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void fma_probe(double* out, int iters, double a0, double b0) {
  double acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = i;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = fma(acc[i], a, b);
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void copy_probe(const double2* __restrict__ in, double2* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  size_t stride = (size_t)gridDim.x * 256;
  for (; i < n; i += stride) out[i] = in[i];
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs %d clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  double* out;
  CK(hipMalloc(&out, sizeof(double) * 256 * 4096));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 20000;
  for (int wg_per_cu = 1; wg_per_cu <= 2; ++wg_per_cu) {
    int blocks = p.multiProcessorCount * wg_per_cu;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL((mfma_probe<8>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-3);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      double flops = (double)blocks * 4 * iters * 8 * 2048.0;
      if (rep) printf("mfma_f64_16x16x4 x8acc, %d waves/SIMD: %.2f TFLOP/s (%.3f ms)\n", wg_per_cu, flops / ms / 1e9, ms);
    }
  }
  for (int wg_per_cu = 1; wg_per_cu <= 4; wg_per_cu *= 2) {
    int blocks = p.multiProcessorCount * wg_per_cu;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(fma_probe, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 1e-3);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      double flops = (double)blocks * 256 * iters * 16 * 2.0;
      if (rep) printf("v_fma_f64, %d waves/SIMD: %.2f TFLOP/s (%.3f ms)\n", wg_per_cu, flops / ms / 1e9, ms);
    }
  }
  size_t n = (size_t)1 << 27;  // 2 GiB per buffer as double2
  double2 *a, *b;
  CK(hipMalloc(&a, n * sizeof(double2)));
  CK(hipMalloc(&b, n * sizeof(double2)));
  CK(hipMemset(a, 1, n * sizeof(double2)));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(copy_probe, dim3(p.multiProcessorCount * 8), dim3(256), 0, 0, a, b, n);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) printf("copy 2 GiB -> 2 GiB: %.1f GB/s (read+write)\n", 2.0 * n * 16 / ms / 1e6);
  }
  return 0;
}
