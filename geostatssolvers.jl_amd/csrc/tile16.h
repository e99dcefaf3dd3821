// 16 x 16 tiles in the accumulator layout of v_mfma_f64_16x16x4_f64 (lane l, register r <-> element
// (row (l >> 4) + 4 r, column l & 15)), shared by the moving-neighbourhood kernel (krig_local.hip) and the 64 x 64
// Cholesky leaf (dense_la.hip).  Register s of a tile X is the A operand of k-slice s of X' and register s of a tile
// Y is the B operand of k-slice s of Y, so acc + X'Y is four MFMAs on the tiles as they are.
#pragma once

#include <hip/hip_runtime.h>

namespace gss {

typedef double d4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ d4_t xty(const d4_t& x, const d4_t& y, d4_t acc) {  // acc + X'Y
#pragma unroll
  for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[s], y[s], acc, 0, 0, 0);
  return acc;
}

__device__ __forceinline__ double rl64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// t: symmetric positive definite 16 x 16 tile (tile layout).  Returns V = U^-1 (tile layout, upper triangular) for
// t = U'U; *bad is set when a pivot is not positive.  S: 16 x 17 doubles of LDS owned by this wave.
__device__ __forceinline__ d4_t potrf16_inverse(const d4_t& t, double* S, int lane, bool* bad) {
  const int g = lane >> 4, c = lane & 15;
#pragma unroll
  for (int r = 0; r < 4; ++r) S[(g + 4 * r) * 17 + c] = t[r];
  __syncthreads();
  const int i = c;  // lanes 16..63 shadow lanes 0..15
  double row[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) row[q] = S[i * 17 + q];
  __syncthreads();
  // lower Cholesky t = L L', right-looking so that the updates of one step are independent of each other; lane i
  // owns row i (its upper part holds don't-care values); the diagonal keeps 1 / L_jj
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    double d = rl64(row[j], j);
    if (!(d > 0.0)) {
      *bad = true;
      d = 1.0;
    }
    double y = __builtin_amdgcn_rsq(d);  // refined to full precision by two Newton steps
    const double h = 0.5 * d;
    y = fma(y, fma(-h * y, y, 0.5), y);
    y = fma(y, fma(-h * y, y, 0.5), y);
    row[j] = (i == j) ? y : row[j] * y;
#pragma unroll
    for (int q = j + 1; q < 16; ++q) row[q] = fma(-row[j], rl64(row[j], q), row[q]);
    __builtin_amdgcn_sched_barrier(0);  // keep the broadcasts of later columns from being hoisted (SGPR pressure)
  }
  // W = L^-1 in place (unblocked trtri, last column first): lane i ends up with row i of W.  Column j of W is
  // -W22 * L[j+1.., j] / L_jj with W22 the already inverted trailing block, whose row i is in lane i's registers.
#pragma unroll
  for (int q = 1; q < 16; ++q)
    if (q > i) row[q] = 0.0;  // clear the don't-care upper part: W is lower triangular
#pragma unroll
  for (int j = 15; j >= 0; --j) {
    const double dinv = rl64(row[j], j);  // 1 / L_jj (kept on the diagonal by the factorisation)
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int q = j + 1; q < 16; ++q) {
      const double lqj = rl64(row[j], q);  // L[q][j]
      if ((q - j) & 1) a0 = fma(row[q], lqj, a0);
      else a1 = fma(row[q], lqj, a1);
    }
    row[j] = (i == j) ? dinv : (i > j ? -(a0 + a1) * dinv : 0.0);
    __builtin_amdgcn_sched_barrier(0);
  }
  // V = W' back to tile layout: V[a][b] = W[b][a], lane b writes column b
#pragma unroll
  for (int r = 0; r < 16; ++r) S[r * 17 + i] = row[r];
  __syncthreads();
  d4_t v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = S[(g + 4 * r) * 17 + c];
  __syncthreads();
  return v;
}

constexpr int tile_id(int i, int j) { return i * 4 - (i * (i - 1)) / 2 + (j - i); }  // upper block triangle, i <= j

// Same factorisation, with everything the 64 x 64 leaf needs from a diagonal tile: u = U (tile layout, upper,
// zero below the diagonal), v = U^-1, vt = (U^-1)' (lower), and the first column whose pivot was not positive
// (-1 if none).  S and S2: 16 x 17 doubles of LDS each.
__device__ __forceinline__ void potrf16_full(const d4_t& t, double* S, double* S2, int lane, d4_t* u, d4_t* v,
                                             d4_t* vt, int* bad_col) {
  const int g = lane >> 4, c = lane & 15;
#pragma unroll
  for (int r = 0; r < 4; ++r) S[(g + 4 * r) * 17 + c] = t[r];
  __syncthreads();
  const int i = c;
  double row[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) row[q] = S[i * 17 + q];
  __syncthreads();
  double mydiag = 1.0;
  int badc = -1;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    double d = rl64(row[j], j);
    if (!(d > 0.0)) {
      if (badc < 0) badc = j;
      d = 1.0;
    }
    double y = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
    y = fma(y, fma(-h * y, y, 0.5), y);
    y = fma(y, fma(-h * y, y, 0.5), y);
    if (i == j) mydiag = d * y;
    row[j] = (i == j) ? y : row[j] * y;
#pragma unroll
    for (int q = j + 1; q < 16; ++q) row[q] = fma(-row[j], rl64(row[j], q), row[q]);
    __builtin_amdgcn_sched_barrier(0);
  }
  // U[a][b] = L[b][a]: lane b writes column b of U
#pragma unroll
  for (int q = 0; q < 16; ++q) S2[q * 17 + i] = (q < i) ? row[q] : (q == i ? mydiag : 0.0);
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) (*u)[r] = S2[(g + 4 * r) * 17 + c];
  __syncthreads();
#pragma unroll
  for (int q = 1; q < 16; ++q)
    if (q > i) row[q] = 0.0;
#pragma unroll
  for (int j = 15; j >= 0; --j) {
    const double dinv = rl64(row[j], j);
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int q = j + 1; q < 16; ++q) {
      const double lqj = rl64(row[j], q);
      if ((q - j) & 1) a0 = fma(row[q], lqj, a0);
      else a1 = fma(row[q], lqj, a1);
    }
    row[j] = (i == j) ? dinv : (i > j ? -(a0 + a1) * dinv : 0.0);
    __builtin_amdgcn_sched_barrier(0);
  }
  // lane i holds row i of W = L^-1: V = W' (lane b writes column b), VT = W (lane a writes row a)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    S[r * 17 + i] = row[r];
    S2[i * 17 + r] = row[r];
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    (*v)[r] = S[(g + 4 * r) * 17 + c];
    (*vt)[r] = S2[(g + 4 * r) * 17 + c];
  }
  __syncthreads();
  *bad_col = badc;
}

}  // namespace gss
