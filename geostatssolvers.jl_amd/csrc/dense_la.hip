// Dense FP64 toolkit on the gfx950 matrix cores: generic strided GEMM, recursive Cholesky (potrf),
// triangular inverse (trtri) and right triangular solve.  These serve
//   - the one-off factorisation of the (n+nc)^2 kriging system  (GeoStatsModels.fit, call site
//     /root/reference/src/estimation/krig.jl:176), and
//   - LUGS' preprocess: cholesky / `\` / `*` at /root/reference/src/simulation/lu.jl:128,134-139.
// All matrices are column-major with the lower triangle significant.  The recursions turn the
// factorisations into large-K GEMMs, which is where the MFMA tile kernel is efficient; the leaves
// (<= 64 x 64) run in one workgroup out of LDS.
#include "gss_internal.h"
#include "mfma_f64.h"

namespace gss {

// ---------------------------------------------------------------------------------------------
// generic GEMM
// ---------------------------------------------------------------------------------------------
struct GemmArgs {
  int64_t M, N, K;
  double alpha, beta;
  const double* A;
  int64_t sa_i, sa_k;
  const double* B;
  int64_t sb_k, sb_j;
  double* D;
  int64_t sd_i, sd_j;
  int lower_only;
};

template <bool A_ICONTIG, bool B_JCONTIG>
__global__ __launch_bounds__(256) void gemm_f64_generic_kernel(GemmArgs g) {
  __shared__ double As[TILE_LDS];
  __shared__ double Bs[TILE_LDS];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t i0 = (int64_t)blockIdx.y * BM;
  const int64_t j0 = (int64_t)blockIdx.x * BN;
  if (g.lower_only && j0 > i0 + BM - 1) return;

  d4 acc[4][4];
  zero_acc(acc);

  for (int64_t k0 = 0; k0 < g.K; k0 += BK) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      int i, k;
      if (A_ICONTIG) {
        i = tid & 127;
        k = (tid >> 7) + 2 * r;
      } else {
        k = tid & 15;
        i = (tid >> 4) + 16 * r;
      }
      const int64_t gi = i0 + i, gk = k0 + k;
      double v = 0.0;
      if (gi < g.M && gk < g.K) v = g.A[gi * g.sa_i + gk * g.sa_k];
      As[k * LDS_LD + i] = v;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      int j, k;
      if (B_JCONTIG) {
        j = tid & 127;
        k = (tid >> 7) + 2 * r;
      } else {
        k = tid & 15;
        j = (tid >> 4) + 16 * r;
      }
      const int64_t gj = j0 + j, gk = k0 + k;
      double v = 0.0;
      if (gj < g.N && gk < g.K) v = g.B[gk * g.sb_k + gj * g.sb_j];
      Bs[k * LDS_LD + j] = v;
    }
    __syncthreads();
    mma_stage(As, Bs, acc, wm, wn, lane);
    __syncthreads();
  }

  const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int tm = 0; tm < 4; ++tm)
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t gi = i0 + wm * 64 + tm * 16 + lk + 4 * r;
        const int64_t gj = j0 + wn * 64 + tn * 16 + lr;
        if (gi < g.M && gj < g.N) {
          double* d = g.D + gi * g.sd_i + gj * g.sd_j;
          double v = g.alpha * acc[tm][tn][r];
          if (g.beta != 0.0) v += g.beta * (*d);
          *d = v;
        }
      }
}

int32_t gemm_f64(int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t sa_i, int64_t sa_k,
                 const double* B, int64_t sb_k, int64_t sb_j, double beta, double* D, int64_t sd_i,
                 int64_t sd_j, bool lower_only, hipStream_t s) {
  if (M <= 0 || N <= 0) return GSS_OK;
  GemmArgs g{M, N, K, alpha, beta, A, sa_i, sa_k, B, sb_k, sb_j, D, sd_i, sd_j, lower_only ? 1 : 0};
  dim3 grid((unsigned)((N + BN - 1) / BN), (unsigned)((M + BM - 1) / BM));
  const bool ai = (sa_i == 1) || (sa_k != 1);
  const bool bj = (sb_j == 1) || (sb_k != 1);
  if (ai && bj)
    hipLaunchKernelGGL((gemm_f64_generic_kernel<true, true>), grid, dim3(256), 0, s, g);
  else if (ai && !bj)
    hipLaunchKernelGGL((gemm_f64_generic_kernel<true, false>), grid, dim3(256), 0, s, g);
  else if (!ai && bj)
    hipLaunchKernelGGL((gemm_f64_generic_kernel<false, true>), grid, dim3(256), 0, s, g);
  else
    hipLaunchKernelGGL((gemm_f64_generic_kernel<false, false>), grid, dim3(256), 0, s, g);
  GSS_HIP(hipGetLastError());
  return GSS_OK;
}

// ---------------------------------------------------------------------------------------------
// GEMV (column-major A): one wave per output element group; deterministic reduction order
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemv_n_kernel(int64_t m, int64_t n, const double* __restrict__ A,
                                                     int64_t lda, const double* __restrict__ x,
                                                     double* __restrict__ y) {
  // y[i] = sum_j A[i + j*lda] x[j]; thread per row (coalesced over i)
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  double acc = 0.0;
  for (int64_t j = 0; j < n; ++j) acc = fma(A[i + j * lda], x[j], acc);
  y[i] = acc;
}

__global__ __launch_bounds__(64) void gemv_t_kernel(int64_t m, int64_t n, const double* __restrict__ A,
                                                    int64_t lda, const double* __restrict__ x,
                                                    double* __restrict__ y) {
  // y[j] = sum_i A[i + j*lda] x[i]; one wave per column, lanes stride over i, butterfly reduce
  const int64_t j = blockIdx.x;
  const int lane = threadIdx.x;
  double acc = 0.0;
  for (int64_t i = lane; i < m; i += 64) acc = fma(A[i + j * lda], x[i], acc);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) y[j] = acc;
}

int32_t gemv_f64(bool trans, int64_t m, int64_t n, const double* A, int64_t lda, const double* x, double* y,
                 hipStream_t s) {
  if (!trans) {
    hipLaunchKernelGGL(gemv_n_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, m, n, A, lda, x, y);
  } else {
    hipLaunchKernelGGL(gemv_t_kernel, dim3((unsigned)n), dim3(64), 0, s, m, n, A, lda, x, y);
  }
  GSS_HIP(hipGetLastError());
  return GSS_OK;
}

// ---------------------------------------------------------------------------------------------
// leaves: Cholesky and triangular inverse of one block (n <= 64) in LDS
// ---------------------------------------------------------------------------------------------
constexpr int LEAF = 64;
constexpr int LEAF_LD = LEAF + 1;

__global__ __launch_bounds__(256) void potrf_leaf_kernel(double* __restrict__ A, int n, int64_t lda,
                                                         int row_offset, int* __restrict__ info) {
  __shared__ double S[LEAF * LEAF_LD];  // S[i * LEAF_LD + j] = A(i, j)
  __shared__ int bad;
  const int tid = threadIdx.x;
  if (tid == 0) bad = 0;
  for (int idx = tid; idx < n * n; idx += 256) {
    const int i = idx % n, j = idx / n;
    S[i * LEAF_LD + j] = (j <= i) ? A[i + (int64_t)j * lda] : 0.0;
  }
  __syncthreads();
  for (int j = 0; j < n; ++j) {
    if (tid == 0) {
      const double d = S[j * LEAF_LD + j];
      if (!(d > 0.0)) {
        bad = 1;
        if (*info == 0) *info = row_offset + j + 1;
        S[j * LEAF_LD + j] = 1.0;
      } else {
        S[j * LEAF_LD + j] = sqrt(d);
      }
    }
    __syncthreads();
    const double inv = 1.0 / S[j * LEAF_LD + j];
    for (int i = j + 1 + tid; i < n; i += 256) S[i * LEAF_LD + j] *= inv;
    __syncthreads();
    const int rem = n - j - 1;
    for (int idx = tid; idx < rem * rem; idx += 256) {
      const int ii = j + 1 + idx / rem;
      const int kk = j + 1 + idx % rem;
      if (kk <= ii) S[ii * LEAF_LD + kk] -= S[ii * LEAF_LD + j] * S[kk * LEAF_LD + j];
    }
    __syncthreads();
  }
  for (int idx = tid; idx < n * n; idx += 256) {
    const int i = idx % n, j = idx / n;
    if (j <= i) A[i + (int64_t)j * lda] = S[i * LEAF_LD + j];
  }
}

// W = inv(L) for one lower-triangular block; each of the first n threads owns one column of W.
__global__ __launch_bounds__(64) void trtri_leaf_kernel(const double* __restrict__ L, int n, int64_t ldl,
                                                        double* __restrict__ W, int64_t ldw) {
  __shared__ double S[LEAF * LEAF_LD];
  __shared__ double X[LEAF * LEAF_LD];  // X[c * LEAF_LD + i] = W(i, c)
  const int tid = threadIdx.x;
  for (int idx = tid; idx < n * n; idx += 64) {
    const int i = idx % n, j = idx / n;
    S[i * LEAF_LD + j] = (j <= i) ? L[i + (int64_t)j * ldl] : 0.0;
  }
  __syncthreads();
  if (tid < n) {
    const int c = tid;
    double* x = X + c * LEAF_LD;
    x[c] = 1.0 / S[c * LEAF_LD + c];
    for (int i = c + 1; i < n; ++i) {
      double acc = 0.0;
      for (int k = c; k < i; ++k) acc = fma(S[i * LEAF_LD + k], x[k], acc);
      x[i] = -acc / S[i * LEAF_LD + i];
    }
  }
  __syncthreads();
  for (int idx = tid; idx < n * n; idx += 64) {
    const int i = idx % n, j = idx / n;
    W[i + (int64_t)j * ldw] = (j <= i) ? X[j * LEAF_LD + i] : 0.0;
  }
}

static int64_t split_point(int64_t n) {
  // first half size: multiple of LEAF, at least LEAF, roughly n/2
  int64_t h = (n / 2 + LEAF - 1) / LEAF * LEAF;
  if (h >= n) h = n - 1 > LEAF ? (n - 1) / LEAF * LEAF : LEAF;
  if (h <= 0) h = LEAF;
  return h;
}

int32_t trtri_f64(const double* L, int64_t n, int64_t ldl, double* W, int64_t ldw, double* T, hipStream_t s) {
  if (n <= 0) return GSS_OK;
  if (n <= LEAF) {
    hipLaunchKernelGGL(trtri_leaf_kernel, dim3(1), dim3(64), 0, s, L, (int)n, ldl, W, ldw);
    GSS_HIP(hipGetLastError());
    return GSS_OK;
  }
  const int64_t n1 = split_point(n), n2 = n - n1;
  const double* L21 = L + n1;
  const double* L22 = L + n1 + n1 * ldl;
  double* W21 = W + n1;
  double* W22 = W + n1 + n1 * ldw;
  GSS_TRY(trtri_f64(L, n1, ldl, W, ldw, T, s));
  GSS_TRY(trtri_f64(L22, n2, ldl, W22, ldw, T, s));
  // T (n2 x n1, row-major) = L21 * W11
  GSS_TRY(gemm_f64(n2, n1, n1, 1.0, L21, 1, ldl, W, 1, ldw, 0.0, T, n1, 1, false, s));
  // W21 = -W22 * T
  GSS_TRY(gemm_f64(n2, n1, n2, -1.0, W22, 1, ldw, T, n1, 1, 0.0, W21, 1, ldw, false, s));
  return GSS_OK;
}

int32_t trsm_right_lt_f64(double* X, int64_t m, int64_t n, int64_t ldx, const double* L, int64_t ldl,
                          double* scratch, hipStream_t s) {
  if (m <= 0 || n <= 0) return GSS_OK;
  if (n <= LEAF) {
    // X <- X * inv(L)'  with inv(L) formed in scratch (n x n, ld = LEAF)
    hipLaunchKernelGGL(trtri_leaf_kernel, dim3(1), dim3(64), 0, s, L, (int)n, ldl, scratch, (int64_t)LEAF);
    GSS_HIP(hipGetLastError());
    // D(i,j) = sum_k X(i,k) Winv(j,k): one column block (n <= BN), so each workgroup reads all of its
    // rows of X before writing them back
    return gemm_f64(m, n, n, 1.0, X, 1, ldx, scratch, LEAF, 1, 0.0, X, 1, ldx, false, s);
  }
  const int64_t n1 = split_point(n), n2 = n - n1;
  double* X1 = X;
  double* X2 = X + n1 * ldx;
  const double* L21 = L + n1;
  const double* L22 = L + n1 + n1 * ldl;
  GSS_TRY(trsm_right_lt_f64(X1, m, n1, ldx, L, ldl, scratch, s));
  // X2 -= X1 * L21'
  GSS_TRY(gemm_f64(m, n2, n1, -1.0, X1, 1, ldx, L21, ldl, 1, 1.0, X2, 1, ldx, false, s));
  GSS_TRY(trsm_right_lt_f64(X2, m, n2, ldx, L22, ldl, scratch, s));
  return GSS_OK;
}

static int32_t potrf_rec(double* A, int64_t n, int64_t lda, int64_t row_offset, int* d_info, double* scratch,
                         hipStream_t s) {
  if (n <= 0) return GSS_OK;
  if (n <= LEAF) {
    hipLaunchKernelGGL(potrf_leaf_kernel, dim3(1), dim3(256), 0, s, A, (int)n, lda, (int)row_offset, d_info);
    GSS_HIP(hipGetLastError());
    return GSS_OK;
  }
  const int64_t n1 = split_point(n), n2 = n - n1;
  double* A21 = A + n1;
  double* A22 = A + n1 + n1 * lda;
  GSS_TRY(potrf_rec(A, n1, lda, row_offset, d_info, scratch, s));
  GSS_TRY(trsm_right_lt_f64(A21, n2, n1, lda, A, lda, scratch, s));
  // A22 -= A21 * A21'   (lower tiles)
  GSS_TRY(gemm_f64(n2, n2, n1, -1.0, A21, 1, lda, A21, lda, 1, 1.0, A22, 1, lda, true, s));
  GSS_TRY(potrf_rec(A22, n2, lda, row_offset + n1, d_info, scratch, s));
  return GSS_OK;
}

int32_t potrf_f64(double* A, int64_t n, int64_t lda, int* d_info, hipStream_t s) {
  DevBuf scratch;
  GSS_TRY(scratch.alloc(sizeof(double) * LEAF * LEAF));
  GSS_HIP(hipMemsetAsync(d_info, 0, sizeof(int), s));
  GSS_TRY(potrf_rec(A, n, lda, 0, d_info, scratch.as<double>(), s));
  GSS_HIP(hipStreamSynchronize(s));  // scratch is freed on return
  return GSS_OK;
}

}  // namespace gss

using namespace gss;

extern "C" {

int32_t gss_dev_potrf(double* a, int64_t n, int64_t lda, void* stream) {
  GSS_REQUIRE(a != nullptr && n >= 0 && lda >= n, "gss_dev_potrf: bad arguments");
  hipStream_t s = to_stream(stream);
  DevBuf info;
  GSS_TRY(info.alloc(sizeof(int)));
  GSS_TRY(potrf_f64(a, n, lda, info.as<int>(), s));
  int h = 0;
  GSS_HIP(hipMemcpy(&h, info.p, sizeof(int), hipMemcpyDeviceToHost));
  if (h != 0) {
    set_error("Cholesky failed: non-positive pivot at row %d", h - 1);
    return GSS_ERR_NOT_POSDEF;
  }
  return GSS_OK;
}

int32_t gss_dev_trtri(const double* l, int64_t n, int64_t ldl, double* w, int64_t ldw, void* stream) {
  GSS_REQUIRE(l != nullptr && w != nullptr && n >= 0 && ldl >= n && ldw >= n, "gss_dev_trtri: bad arguments");
  hipStream_t s = to_stream(stream);
  DevBuf T;
  const int64_t h = n / 2 + LEAF;
  GSS_TRY(T.alloc(sizeof(double) * (size_t)(h * h)));
  GSS_HIP(hipMemset2DAsync(w, sizeof(double) * ldw, 0, sizeof(double) * n, n, s));
  GSS_TRY(trtri_f64(l, n, ldl, w, ldw, T.as<double>(), s));
  GSS_HIP(hipStreamSynchronize(s));
  return GSS_OK;
}

int32_t gss_dev_gemm(int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t sa_i, int64_t sa_k,
                     const double* B, int64_t sb_k, int64_t sb_j, double beta, double* D, int64_t sd_i,
                     int64_t sd_j, int32_t lower_only, void* stream) {
  GSS_REQUIRE(A && B && D && M >= 0 && N >= 0 && K >= 0, "gss_dev_gemm: bad arguments");
  return gemm_f64(M, N, K, alpha, A, sa_i, sa_k, B, sb_k, sb_j, beta, D, sd_i, sd_j, lower_only != 0,
                  to_stream(stream));
}

}  // extern "C"
