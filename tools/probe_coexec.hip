// Do FP64 MFMA and FP64 VALU FMA share execution resources on gfx950?  Each wave interleaves one
// v_mfma_f64_16x16x4_f64 with NF independent v_fma_f64; if the pipes were independent the loop would cost
// max(64, 4 NF) cycles per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int NF, int NM>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0, double b0) {
  d4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
  double f[16];
  for (int i = 0; i < 16; ++i) f[i] = i + threadIdx.x;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (NM) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[m], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NF; ++i) f[(m * NF + i) & 15] = fma(f[(m * NF + i) & 15], a, b);
    }
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += f[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(c1 - c0) / (iters * 4.0);
}
template <int NF, int NM>
int run(double* out, int CUs) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 20000;
  for (int wpc = 1; wpc <= 2; ++wpc) {
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL((k<NF, NM>), dim3(CUs * wpc), dim3(256), 0, 0, out, iters, 1.0000001, 1e-3);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
    }
    double cyc; CK(hipMemcpy(&cyc, out, 8, hipMemcpyDeviceToHost));
    double fl = (double)CUs * wpc * 4 * iters * 4.0 * (NM * 2048.0 + NF * 128.0);
    printf("MFMA %d + %2d FMA per slot, %d waves/SIMD: %.1f cycles/slot/wave, %.2f TFLOP/s total\n", NM, NF, wpc, cyc, fl / ms / 1e9);
  }
  return 0;
}
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  double* out; CK(hipMalloc(&out, sizeof(double) * 256 * 2048));
  if (run<0, 1>(out, p.multiProcessorCount)) return 1;
  if (run<4, 1>(out, p.multiProcessorCount)) return 1;
  if (run<8, 1>(out, p.multiProcessorCount)) return 1;
  if (run<12, 1>(out, p.multiProcessorCount)) return 1;
  if (run<16, 1>(out, p.multiProcessorCount)) return 1;
  if (run<16, 0>(out, p.multiProcessorCount)) return 1;
  return 0;
}
