// Micro-probe: sustained v_mfma_f64_16x16x4_f64 / v_mfma_f64_4x4x4_4b_f64 / v_fma_f64 rates with the
// shader clock measured in-kernel (s_memtime vs the 100 MHz s_memrealtime), plus copy bandwidth.
// Grounds the roofline peaks quoted in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Stamp { unsigned long long c0, c1, r0, r1; };

template <int NACC, int KIND>
__global__ __launch_bounds__(256) void mfma_probe(double* out, Stamp* st, int iters, double a0, double b0) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double accs[NACC];
  for (int i = 0; i < NACC; ++i) accs[i] = 0;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      else accs[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, accs[i], 0, 0, 0);
    }
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + accs[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *st = Stamp{c0, c1, r0, r1};
}

__global__ __launch_bounds__(256) void fma_probe(double* out, Stamp* st, int iters, double a0, double b0) {
  double acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = i;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = fma(acc[i], a, b);
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *st = Stamp{c0, c1, r0, r1};
}

__global__ __launch_bounds__(256) void copy_probe(const double2* __restrict__ in, double2* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  size_t stride = (size_t)gridDim.x * 256;
  for (; i < n; i += stride) out[i] = in[i];
}

static double ghz(const Stamp& s) { return (double)(s.c1 - s.c0) / ((double)(s.r1 - s.r0) * 10.0); }

template <int NACC, int KIND>
int run_mfma(const char* name, double flop_per_inst, int CUs, double* out, Stamp* dst, hipEvent_t e0, hipEvent_t e1, int iters) {
  for (int wpc = 1; wpc <= 4; wpc *= 2) {
    int blocks = CUs * wpc;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL((mfma_probe<NACC, KIND>), dim3(blocks), dim3(256), 0, 0, out, dst, iters, 1.0, 1e-3);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
    }
    Stamp s;
    CK(hipMemcpy(&s, dst, sizeof(s), hipMemcpyDeviceToHost));
    double insts = (double)iters * NACC;
    double flops = (double)blocks * 4 * insts * flop_per_inst;
    printf("%s x%dacc, %d waves/SIMD: %.2f TFLOP/s (%.2f ms) clock %.2f GHz, %.1f shader cycles per MFMA per wave\n", name, NACC,
           wpc, flops / ms / 1e9, ms, ghz(s), (double)(s.c1 - s.c0) / insts);
  }
  return 0;
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs %d clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  double* out;
  Stamp* dst;
  CK(hipMalloc(&out, sizeof(double) * 256 * 4096));
  CK(hipMalloc(&dst, sizeof(Stamp)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 40000;
  if (run_mfma<4, 0>("mfma_f64_16x16x4", 2048.0, p.multiProcessorCount, out, dst, e0, e1, iters)) return 1;
  if (run_mfma<16, 0>("mfma_f64_16x16x4", 2048.0, p.multiProcessorCount, out, dst, e0, e1, iters / 4)) return 1;
  if (run_mfma<8, 1>("mfma_f64_4x4x4_4b", 512.0, p.multiProcessorCount, out, dst, e0, e1, iters)) return 1;
  for (int wpc = 1; wpc <= 8; wpc *= 2) {
    int blocks = p.multiProcessorCount * wpc;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(fma_probe, dim3(blocks), dim3(256), 0, 0, out, dst, iters, 1.0000001, 1e-3);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
    }
    Stamp s;
    CK(hipMemcpy(&s, dst, sizeof(s), hipMemcpyDeviceToHost));
    double flops = (double)blocks * 256 * iters * 16 * 2.0;
    printf("v_fma_f64, %d waves/SIMD: %.2f TFLOP/s (%.2f ms) clock %.2f GHz\n", wpc, flops / ms / 1e9, ms, ghz(s));
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
