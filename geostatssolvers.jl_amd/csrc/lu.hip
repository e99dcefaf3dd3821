// Partial-pivot LU for LUGS' `factorization = lu` (/root/reference/src/simulation/lu.jl:70,107,128,134,139).
//
// The reference calls `lu(Symmetric(C)).L`: LAPACK getrf on the dense matrix (P C = L U, pivot = the first row of
// maximal |a| in the column) and keeps only the unit lower-triangular factor -- the row permutation and U are
// dropped, so L is NOT a square root of C; the solver reproduces what the reference computes, not what Alabert's
// method intends (LUGS).  Right-looking blocked factorisation, column-major, in place:
//   panel (LU_NB columns): one workgroup, per column an arg-max reduction, the row interchange inside the panel,
//                          the scaling by the reciprocal pivot (as LAPACK's getrf2) and the rank-1 update of the
//                          rest of the panel;
//   interchanges applied to the columns left and right of the panel (laswp);
//   U12 = inv(L11) A12: one thread per column, L11 in LDS;
//   A22 -= L21 U12 on the FP64-MFMA GEMM (dense_la.hip).
// The option exists for parity with the reference's parameter surface; its own test runs it on 100 cells
// (test/simulation/lu.jl:72).  The single-workgroup panel makes large systems slow (about a second at n = 12 288).
#include "gss_internal.h"

namespace gss {

constexpr int LU_NB = 32;
constexpr int LU_NT = 1024;

// factorises the m x jb panel at P (leading dimension lda); ipiv[c] = j0 + (row of the pivot of column c)
__global__ __launch_bounds__(LU_NT) void getrf_panel_kernel(double* __restrict__ P, int64_t m, int jb, int64_t lda,
                                                            int64_t j0, int* __restrict__ ipiv, int* __restrict__ info) {
  __shared__ double s_val[LU_NT / 64];
  __shared__ long long s_row[LU_NT / 64];
  __shared__ double s_urow[LU_NB];
  __shared__ long long s_piv;
  __shared__ double s_inv;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  for (int c = 0; c < jb && c < m; ++c) {
    double* col = P + (int64_t)c * lda;
    // ---- pivot: first row of maximal |a| among rows c .. m-1 (idamax)
    double best = -1.0;
    long long brow = m;
    for (int64_t r = c + tid; r < m; r += LU_NT) {
      const double v = fabs(col[r]);
      if (v > best) {  // rows are visited in increasing order per thread: strict > keeps the first
        best = v;
        brow = r;
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const double ov = __shfl_xor(best, off);
      const long long orow = __shfl_xor(brow, off);
      if (ov > best || (ov == best && orow < brow)) {
        best = ov;
        brow = orow;
      }
    }
    if (lane == 0) {
      s_val[wave] = best;
      s_row[wave] = brow;
    }
    __syncthreads();
    if (tid == 0) {
      double b = s_val[0];
      long long br = s_row[0];
      for (int w = 1; w < LU_NT / 64; ++w)
        if (s_val[w] > b || (s_val[w] == b && s_row[w] < br)) {
          b = s_val[w];
          br = s_row[w];
        }
      if (!(b > 0.0)) {  // exactly singular (or NaN): report like LAPACK's info > 0, keep going with no interchange
        if (*info == 0) *info = (int)(j0 + c + 1);
        br = c;
      }
      s_piv = br;
      ipiv[c] = (int)(j0 + br);
    }
    __syncthreads();
    const long long p = s_piv;
    // ---- interchange rows c and p inside the panel; keep row c (columns c .. jb-1) for the update
    if (tid < jb) {
      double* a = P + (int64_t)tid * lda;
      const double vc = a[c], vp = a[p];
      if (p != c) {
        a[c] = vp;
        a[p] = vc;
      }
      s_urow[tid] = vp;
      if (tid == c) s_inv = vp != 0.0 ? 1.0 / vp : 0.0;
    }
    __syncthreads();
    const double inv = s_inv;
    // ---- scale the column below the pivot and update the rest of the panel (row-wise: coalesced over rows)
    for (int64_t r = c + 1 + tid; r < m; r += LU_NT) {
      const double l = col[r] * inv;
      col[r] = l;
      for (int c2 = c + 1; c2 < jb; ++c2) {
        double* a = P + (int64_t)c2 * lda + r;
        *a = fma(-l, s_urow[c2], *a);
      }
    }
    __syncthreads();
  }
}

// apply the jb interchanges of a panel to `ncols` other columns starting at A0 (row indices are global)
__global__ __launch_bounds__(256) void laswp_kernel(double* __restrict__ A0, int64_t ncols, int64_t lda, int64_t j0,
                                                    int jb, const int* __restrict__ ipiv) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= ncols) return;
  double* a = A0 + j * lda;
  for (int c = 0; c < jb; ++c) {
    const int64_t r = j0 + c, p = ipiv[c];
    if (p != r) {
      const double t = a[r];
      a[r] = a[p];
      a[p] = t;
    }
  }
}

// B <- inv(L11) B, L11 unit lower triangular jb x jb (strict lower of the panel's top block), B is jb x ncols
__global__ __launch_bounds__(256) void trsm_unit_lower_kernel(const double* __restrict__ L11, int jb, int64_t lda,
                                                              double* __restrict__ B, int64_t ncols) {
  __shared__ double sL[LU_NB * LU_NB];
  for (int e = threadIdx.x; e < jb * jb; e += 256) {
    const int i = e % jb, k = e / jb;
    sL[i * LU_NB + k] = L11[i + (int64_t)k * lda];
  }
  __syncthreads();
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= ncols) return;
  double* b = B + j * lda;
  double x[LU_NB];
#pragma unroll
  for (int i = 0; i < LU_NB; ++i) x[i] = i < jb ? b[i] : 0.0;
#pragma unroll
  for (int i = 1; i < LU_NB; ++i) {
    if (i < jb) {
      double acc = x[i];
#pragma unroll
      for (int k = 0; k < LU_NB; ++k)
        if (k < i) acc = fma(-sL[i * LU_NB + k], x[k], acc);
      x[i] = acc;
    }
  }
#pragma unroll
  for (int i = 0; i < LU_NB; ++i)
    if (i < jb) b[i] = x[i];
}

// A <- unit lower triangular factor: zero above the diagonal, ones on it (`.L` of the LU object)
__global__ __launch_bounds__(256) void unit_lower_kernel(double* __restrict__ A, int64_t n, int64_t lda) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t j = blockIdx.y;
  if (i < n && i <= j) A[i + j * lda] = (i == j) ? 1.0 : 0.0;
}

// upper triangle <- transpose of the lower triangle (the SYRK-shaped update writes lower tiles only)
__global__ __launch_bounds__(256) void mirror_lower_kernel(double* __restrict__ A, int64_t n, int64_t lda) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // row, i > j
  const int64_t j = blockIdx.y;
  if (i < n && i > j) A[j + i * lda] = A[i + j * lda];
}

int32_t mirror_lower_f64(double* A, int64_t n, int64_t lda, hipStream_t s) {
  if (n <= 1) return GSS_OK;
  hipLaunchKernelGGL(mirror_lower_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)n), dim3(256), 0, s, A, n, lda);
  GSS_HIP(hipGetLastError());
  return GSS_OK;
}

// A (n x n, column-major, full) <- L of P A = L U as a dense unit lower-triangular matrix.  *d_info (device int,
// zeroed by the caller) receives 1 + column of the first exactly-zero pivot.  ipiv: device scratch of n ints.
int32_t getrf_unit_lower_f64(double* A, int64_t n, int64_t lda, int* ipiv, int* d_info, hipStream_t s) {
  for (int64_t j0 = 0; j0 < n; j0 += LU_NB) {
    const int jb = (int)((n - j0) < LU_NB ? (n - j0) : LU_NB);
    const int64_t m = n - j0;
    double* P = A + j0 + j0 * lda;
    hipLaunchKernelGGL(getrf_panel_kernel, dim3(1), dim3(LU_NT), 0, s, P, m, jb, lda, j0, ipiv + j0, d_info);
    if (j0 > 0)
      hipLaunchKernelGGL(laswp_kernel, dim3((unsigned)((j0 + 255) / 256)), dim3(256), 0, s, A, j0, lda, j0, jb, ipiv + j0);
    const int64_t nr = n - j0 - jb;
    if (nr > 0) {
      double* A12 = A + (j0 + jb) * lda;   // column j0 + jb, row 0 (interchanges use global rows)
      hipLaunchKernelGGL(laswp_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, s, A12, nr, lda, j0, jb, ipiv + j0);
      hipLaunchKernelGGL(trsm_unit_lower_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, s, P, jb, lda,
                         A12 + j0, nr);
      GSS_HIP(hipGetLastError());
      // A22 -= L21 U12
      GSS_TRY(gemm_f64(nr, nr, jb, -1.0, P + jb, 1, lda, A12 + j0, 1, lda, 1.0, A12 + j0 + jb, 1, lda, false, s));
    }
  }
  hipLaunchKernelGGL(unit_lower_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)n), dim3(256), 0, s, A, n, lda);
  GSS_HIP(hipGetLastError());
  return GSS_OK;
}

}  // namespace gss

using namespace gss;

extern "C" {

int32_t gss_dev_getrf_l(double* a, int64_t n, int64_t lda, void* stream) {
  GSS_ENTRY();
  GSS_REQUIRE(a != nullptr && n >= 1 && lda >= n, "gss_dev_getrf_l: bad arguments");
  hipStream_t s = to_stream(stream);
  DevBuf info, ipiv;
  GSS_TRY(info.alloc(sizeof(int)));
  GSS_TRY(ipiv.alloc(sizeof(int) * (size_t)n));
  GSS_TRY(dev_zero_bytes(info.p, sizeof(int), s));
  GSS_TRY(getrf_unit_lower_f64(a, n, lda, ipiv.as<int>(), info.as<int>(), s));
  int h = 0;
  GSS_HIP(hipMemcpyAsync(&h, info.p, sizeof(int), hipMemcpyDeviceToHost, s));
  GSS_HIP(hipStreamSynchronize(s));
  if (h != 0) {
    set_error("LU factorisation: exactly singular matrix (zero pivot in column %d)", h - 1);
    return GSS_ERR_NOT_POSDEF;
  }
  return GSS_OK;
}

}  // extern "C"
