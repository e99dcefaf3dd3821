#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__constant__ double EXP2_TAB[64];
__device__ __forceinline__ double fast_exp(double x) {
  // exp(x) for x <= ~0 ... general finite x in [-745, 709]
  const double C = 92.332482616893656;            // 64 / ln 2
  const double L_HI = 0.010830424493178725;     // ln2/64, 27 trailing bits cleared (n * L_HI exact)
  const double L_LO = 2.030704202170295e-10;    // ln2/64 - L_HI
  const double n = __builtin_rint(x * C);
  double r = fma(-n, L_HI, x);
  r = fma(-n, L_LO, r);
  double p = fma(r, 1.0 / 120.0, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = p * r;                                       // exp(r) - 1
  const int ni = (int)n;
  const double t = EXP2_TAB[ni & 63];
  const double y = fma(t, p, t);
  return __builtin_amdgcn_ldexp(y, ni >> 6);
}
__device__ __forceinline__ double fast_exp_poly(double x) {
  x = x < -800.0 ? -800.0 : x;
  const double n = __builtin_rint(x * 1.4426950408889634);
  double r = fma(-n, 6.93147180369123816490e-01, x);
  r = fma(-n, 1.90821492927058770002e-10, r);
  double p = 1.0 / 6227020800.0;
  p = fma(p, r, 1.0 / 479001600.0);
  p = fma(p, r, 1.0 / 39916800.0);
  p = fma(p, r, 1.0 / 3628800.0);
  p = fma(p, r, 1.0 / 362880.0);
  p = fma(p, r, 1.0 / 40320.0);
  p = fma(p, r, 1.0 / 5040.0);
  p = fma(p, r, 1.0 / 720.0);
  p = fma(p, r, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return __builtin_amdgcn_ldexp(p, (int)n);
}
__device__ __forceinline__ double fast_sqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double s = x * y;
  double h = 0.5 * y;
  double e = fma(-s, s, x);
  s = fma(e, h, s);
  e = fma(-s, s, x);
  s = fma(e, h, s);
  return x > 0.0 ? s : 0.0;
}
template <bool POLY>
__global__ void k(const double* x, double* ye, double* ys, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) { ye[i] = POLY ? fast_exp_poly(x[i]) : fast_exp(x[i]); ys[i] = fast_sqrt(-x[i]); }
}
int main() {
  double tab[64];
  for (int j = 0; j < 64; ++j) tab[j] = exp2((double)j / 64.0);
  hipMemcpyToSymbol(HIP_SYMBOL(EXP2_TAB), tab, sizeof(tab));
  const int n = 1 << 20;
  std::vector<double> x(n), ye(n), ys(n);
  for (int i = 0; i < n; ++i) { double u = (double)rand() / RAND_MAX; x[i] = -pow(10.0, -8.0 + 10.9 * u); }
  x[0] = -0.0; x[1] = -1e-300; x[2] = -745.0; x[3] = -700.0; x[4] = -1e-17;
  double *dx, *de, *ds;
  hipMalloc(&dx, n * 8); hipMalloc(&de, n * 8); hipMalloc(&ds, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  for (int variant = 0; variant < 2; ++variant) {
  if (variant) hipLaunchKernelGGL(k<true>, dim3(n / 256), dim3(256), 0, 0, dx, de, ds, n);
  else hipLaunchKernelGGL(k<false>, dim3(n / 256), dim3(256), 0, 0, dx, de, ds, n);
  printf("variant %s\n", variant ? "polynomial" : "table");
  hipMemcpy(ye.data(), de, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(ys.data(), ds, n * 8, hipMemcpyDeviceToHost);
  double me = 0, ms = 0; int ie = 0, is = 0;
  for (int i = 0; i < n; ++i) {
    double re = exp(x[i]), rs = sqrt(-x[i]);
    double ee = re > 1e-300 ? fabs(ye[i] - re) / re : fabs(ye[i] - re);
    double es = rs > 0 ? fabs(ys[i] - rs) / rs : fabs(ys[i] - rs);
    if (ee > me) { me = ee; ie = i; }
    if (es > ms) { ms = es; is = i; }
  }
  printf("exp max rel err %.3e at x=%.6g (%.17g vs %.17g)\n", me, x[ie], ye[ie], exp(x[ie]));
  printf("sqrt max rel err %.3e at x=%.6g\n", ms, -x[is]);
  for (int i = 0; i < 5; ++i) printf("x=%g exp=%.17g ref=%.17g sqrt=%.17g\n", x[i], ye[i], exp(x[i]), ys[i]);
  }
  return 0;
}
