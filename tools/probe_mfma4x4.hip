// Determines the lane mapping of v_mfma_f64_4x4x4_4b_f64 empirically: for every (la, lb) pair a one-hot A (lane la)
// and one-hot B (lane lb) is multiplied; the lane(s) where D != 0 are recorded.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int* out) {
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
      double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      if (d != 0.0) out[la * 64 + lb] = lane;
    }
}
int main() {
  int* d;
  hipMalloc(&d, 4096 * sizeof(int));
  hipMemset(d, 0xff, 4096 * sizeof(int));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  int h[4096];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int la = 0; la < 64; ++la) {
    printf("A lane %2d:", la);
    for (int lb = 0; lb < 64; ++lb)
      if (h[la * 64 + lb] >= 0) printf(" (B%d->D%d)", lb, h[la * 64 + lb]);
    printf("\n");
  }
  return 0;
}
