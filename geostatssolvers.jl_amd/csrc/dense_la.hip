// Dense FP64 toolkit on the gfx950 matrix cores: generic strided GEMM, recursive Cholesky (potrf),
// triangular inverse (trtri) and right triangular solve.  These serve
//   - the one-off factorisation of the (n+nc)^2 kriging system  (GeoStatsModels.fit, call site
//     /root/reference/src/estimation/krig.jl:176), and
//   - LUGS' preprocess: cholesky / `\` / `*` at /root/reference/src/simulation/lu.jl:128,134-139.
// All matrices are column-major with the lower triangle significant.  The recursions turn the
// factorisations into large-K GEMMs, which is where the MFMA tile kernel is efficient; the leaves
// (<= 64 x 64) run in one workgroup out of LDS.
#include "gss_internal.h"

#include <atomic>
#include <mutex>
#include <unordered_map>
#include "mfma_f64.h"
#include "tile16.h"

#include <cstdlib>

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
  int tri;  // zero structure of an operand, so that k-tiles that only meet zeros are skipped (GEMM_TRI_*)
  int vec;  // both operands contiguous along their tile edge, 16-byte aligned, K a multiple of BK: interior tiles may
            // stage with 16-byte loads
};

// Register-staged, LDS double-buffered pipeline (one barrier per BK stage), bounds-checked scalar loads that
// are coalesced along whichever index is contiguous.  Two tile shapes: 128 x 128 (4 waves x 4x4 MFMA tiles) for
// large problems and 64 x 64 (4 waves x 2x2 MFMA tiles) for the many small products of the recursions, where a
// 128-tile grid would leave most CUs idle and waste 3/4 of the matrix work on padding.
// B(k, j) = 0 for k > j (the transpose of a lower-triangular factor): columns j0.. need k < j0 + T only
constexpr int GEMM_TRI_B_UPPER = 1;
// B(k, j) = 0 for k < j (a lower-triangular factor): columns j0.. need k >= j0 only
constexpr int GEMM_TRI_B_LOWER = 2;
// A(i, k) = 0 for k > i (a lower-triangular factor on the left): rows i0.. need k < i0 + T only
constexpr int GEMM_TRI_A_LOWER = 4;

template <int T>
struct GemmTile {
  static constexpr int TILE = T;             // workgroup tile edge
  static constexpr int WT = T / 32;          // MFMA tiles per wave edge (wave tile = T/2)
  static constexpr int LD = T + 16;          // LDS row stride: 16 (mod 32) doubles
  static constexpr int STAGE = BK * LD;      // doubles per operand stage
  static constexpr int NLOAD = T * BK / 256; // elements per thread per operand per stage
  static constexpr size_t LDS_BYTES = sizeof(double) * 4 * STAGE;
};

template <int T, bool A_ICONTIG, bool B_JCONTIG>
__global__ __launch_bounds__(256, 2) void gemm_f64_generic_kernel(GemmArgs g) {
  using G = GemmTile<T>;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* As = smem;                  // [2][STAGE]
  double* Bs = smem + 2 * G::STAGE;   // [2][STAGE]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // Workgroups reach the CUs in launch order, a CU's next workgroup waits for that CU (measured: a launch whose cost
  // varies with the FAST block index takes as long as if every workgroup had the maximum cost).  When the k-range
  // depends on the column tile the row tile is therefore the fast index, and the expensive columns go first.
  // For the same reason a lower-tiles-only product is launched as a 1-D grid over the tiles that exist (block
  // triangle, then the full rows below it), not as a 2-D grid whose upper workgroups exit at once.
  const bool by_col = (g.tri & (GEMM_TRI_B_UPPER | GEMM_TRI_B_LOWER)) != 0;
  int64_t bx = by_col ? blockIdx.y : blockIdx.x, by = by_col ? blockIdx.x : blockIdx.y;
  const int64_t ncol = by_col ? gridDim.y : gridDim.x;
  if (g.lower_only) {
    const int64_t tn = (g.N + T - 1) / T, tm = (g.M + T - 1) / T;
    const int64_t d = tm < tn ? tm : tn, ntri = d * (d + 1) / 2;
    const int64_t L = blockIdx.x;
    if (L < ntri) {
      by = (int64_t)((__builtin_sqrt(8.0 * (double)L + 1.0) - 1.0) * 0.5);
      while (by * (by + 1) / 2 > L) --by;            // the square root may be one off either way
      while ((by + 1) * (by + 2) / 2 <= L) ++by;
      bx = L - by * (by + 1) / 2;
    } else {
      by = tn + (L - ntri) / tn;
      bx = (L - ntri) % tn;
    }
  }
  const int64_t i0 = by * T;
  const int64_t j0 = ((g.tri & GEMM_TRI_B_UPPER) ? ncol - 1 - bx : bx) * T;
  if (g.lower_only && j0 > i0 + T - 1) return;

  // staging map of this thread: NLOAD elements of each operand tile per stage
  int ai[G::NLOAD], ak[G::NLOAD], bj[G::NLOAD], bk[G::NLOAD];
#pragma unroll
  for (int r = 0; r < G::NLOAD; ++r) {
    const int e = tid + r * 256;
    if (A_ICONTIG) {
      ai[r] = e % T;
      ak[r] = e / T;
    } else {
      ak[r] = e % BK;
      ai[r] = e / BK;
    }
    if (B_JCONTIG) {
      bj[r] = e % T;
      bk[r] = e / T;
    } else {
      bk[r] = e % BK;
      bj[r] = e / BK;
    }
  }
  double ra[G::NLOAD], rb[G::NLOAD];
  // interior tiles of aligned, edge-contiguous operands: 16-byte loads without bounds checks (K3's staging map:
  // thread = 2 consecutive elements of k-row kq + RPP * pass); everything else: bounds-checked scalar loads
  constexpr int RPP = 512 / T;  // k-rows covered per pass of the 256 threads
  const bool fastp = A_ICONTIG && B_JCONTIG && g.vec && (i0 + T <= g.M) && (j0 + T <= g.N);
  const int e2 = (tid % (T / 2)) * 2, kq = tid / (T / 2);
  auto load_stage = [&](int64_t k0) {
    if (fastp) {
#pragma unroll
      for (int r = 0; r < G::NLOAD / 2; ++r) {
        const int64_t kr = k0 + kq + RPP * r;
        const d2v va = *reinterpret_cast<const d2v*>(g.A + (i0 + e2) + kr * g.sa_k);
        const d2v vb = *reinterpret_cast<const d2v*>(g.B + kr * g.sb_k + (j0 + e2));
        ra[2 * r] = va.x;
        ra[2 * r + 1] = va.y;
        rb[2 * r] = vb.x;
        rb[2 * r + 1] = vb.y;
      }
      return;
    }
#pragma unroll
    for (int r = 0; r < G::NLOAD; ++r) {
      const int64_t gi = i0 + ai[r], gka = k0 + ak[r];
      ra[r] = (gi < g.M && gka < g.K) ? g.A[gi * g.sa_i + gka * g.sa_k] : 0.0;
      const int64_t gj = j0 + bj[r], gkb = k0 + bk[r];
      rb[r] = (gj < g.N && gkb < g.K) ? g.B[gkb * g.sb_k + gj * g.sb_j] : 0.0;
    }
  };
  auto store_stage = [&](int buf) {
    double* a = As + buf * G::STAGE;
    double* b = Bs + buf * G::STAGE;
    if (fastp) {
#pragma unroll
      for (int r = 0; r < G::NLOAD / 2; ++r) {
        *reinterpret_cast<d2v*>(a + (kq + RPP * r) * G::LD + e2) = d2v{ra[2 * r], ra[2 * r + 1]};
        *reinterpret_cast<d2v*>(b + (kq + RPP * r) * G::LD + e2) = d2v{rb[2 * r], rb[2 * r + 1]};
      }
      return;
    }
#pragma unroll
    for (int r = 0; r < G::NLOAD; ++r) {
      a[ak[r] * G::LD + ai[r]] = ra[r];
      b[bk[r] * G::LD + bj[r]] = rb[r];
    }
  };

  d4 acc[G::WT][G::WT];
#pragma unroll
  for (int tm = 0; tm < G::WT; ++tm)
#pragma unroll
    for (int tn = 0; tn < G::WT; ++tn) acc[tm][tn] = d4{0.0, 0.0, 0.0, 0.0};
  const int lr = lane & 15, lk = lane >> 4;
  int64_t kend = g.K, kbeg = 0;
  if ((g.tri & GEMM_TRI_B_UPPER) && j0 + T < kend) kend = j0 + T;
  if ((g.tri & GEMM_TRI_A_LOWER) && i0 + T < kend) kend = i0 + T;
  if (g.tri & GEMM_TRI_B_LOWER) kbeg = j0 < g.K ? j0 : g.K;   // j0 is a multiple of T, hence of BK
  const int64_t t0 = kbeg / BK;
  const int64_t ntile = (kend + BK - 1) / BK;
  if (ntile > t0) {
    load_stage(t0 * BK);
    store_stage((int)(t0 & 1));
  }
  __syncthreads();
  for (int64_t t = t0; t < ntile; ++t) {
    const int cur = (int)(t & 1);
    const bool more = (t + 1) < ntile;
    if (more) load_stage((t + 1) * BK);
    const double* as = As + cur * G::STAGE;
    const double* bs = Bs + cur * G::STAGE;
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      const double* ap = as + (kk * 4 + lk) * G::LD + wm * (T / 2) + lr;
      const double* bp = bs + (kk * 4 + lk) * G::LD + wn * (T / 2) + lr;
      double a[G::WT], b[G::WT];
#pragma unroll
      for (int q = 0; q < G::WT; ++q) {
        a[q] = ap[q * 16];
        b[q] = bp[q * 16];
      }
#pragma unroll
      for (int tm = 0; tm < G::WT; ++tm)
#pragma unroll
        for (int tn = 0; tn < G::WT; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
    }
    if (more) store_stage(cur ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int tm = 0; tm < G::WT; ++tm)
#pragma unroll
    for (int tn = 0; tn < G::WT; ++tn)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t gi = i0 + wm * (T / 2) + tm * 16 + lk + 4 * r;
        const int64_t gj = j0 + wn * (T / 2) + tn * 16 + lr;
        if (gi < g.M && gj < g.N) {
          double* d = g.D + gi * g.sd_i + gj * g.sd_j;
          double v = g.alpha * acc[tm][tn][r];
          if (g.beta != 0.0) v += g.beta * (*d);
          *d = v;
        }
      }
}

template <int T, bool AI, bool BJ>
static int32_t gemm_launch(const GemmArgs& g, hipStream_t s) {
  static uint64_t attr = 0;
  if (first_on_this_device(attr)) {
    GSS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_generic_kernel<T, AI, BJ>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)GemmTile<T>::LDS_BYTES));
  }
  const unsigned tn = (unsigned)((g.N + T - 1) / T), tm = (unsigned)((g.M + T - 1) / T);
  const bool by_col = (g.tri & (GEMM_TRI_B_UPPER | GEMM_TRI_B_LOWER)) != 0;   // see the kernel: row tile = fast index
  dim3 grid(by_col ? tm : tn, by_col ? tn : tm);
  if (g.lower_only) {
    const unsigned d = tm < tn ? tm : tn;
    grid = dim3(d * (d + 1) / 2 + (tm > tn ? (tm - tn) * tn : 0));
  }
  hipLaunchKernelGGL((gemm_f64_generic_kernel<T, AI, BJ>), grid, dim3(256), GemmTile<T>::LDS_BYTES, s, g);
  GSS_HIP(hipGetLastError());
  return GSS_OK;
}

template <int T>
static int32_t gemm_dispatch(const GemmArgs& g, bool ai, bool bj, hipStream_t s) {
  if (ai && bj) return gemm_launch<T, true, true>(g, s);
  if (ai && !bj) return gemm_launch<T, true, false>(g, s);
  if (!ai && bj) return gemm_launch<T, false, true>(g, s);
  return gemm_launch<T, false, false>(g, s);
}

int32_t gemm_f64(int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t sa_i, int64_t sa_k,
                 const double* B, int64_t sb_k, int64_t sb_j, double beta, double* D, int64_t sd_i,
                 int64_t sd_j, bool lower_only, hipStream_t s, int tri) {
  if (M <= 0 || N <= 0) return GSS_OK;
  GemmArgs g{M, N, K, alpha, beta, A, sa_i, sa_k, B, sb_k, sb_j, D, sd_i, sd_j, lower_only ? 1 : 0, tri, 0};
  g.vec = (sa_i == 1 && sb_j == 1 && K % BK == 0 && sa_k % 2 == 0 && sb_k % 2 == 0 &&
           reinterpret_cast<uintptr_t>(A) % 16 == 0 && reinterpret_cast<uintptr_t>(B) % 16 == 0)
              ? 1
              : 0;
  const bool ai = (sa_i == 1) || (sa_k != 1);
  const bool bj = (sb_j == 1) || (sb_k != 1);
  // Tile shape by the time the launch takes (round 3, tools/gemm_late_steps.py, K = 1 024; both scale with K): 128-tiles
  // behave as if one workgroup ran per CU, in rounds of 256 (two are resident, but one already keeps the CU's matrix
  // pipes busy): 0.03 + 0.14 ms per round, whether the round is full or not (256 tiles 0.18 ms, 320 ... 512 tiles 0.30); 64-tiles stream through two per CU without visible rounds, 0.035 + 0.038 ms per
  // 256 of them (4 096 x 1 024 x 1 024: 0.19 ms either way; 5 120 x 1 024: 0.24 against 0.31 ms; the 4 560 lower tiles of
  // a 12 288^2 trailing update: 128-tiles by 7 %).  The products of the late steps of a blocked factorisation are
  // partial rounds, which is where the choice matters.
  auto tiles = [&](int64_t T) {
    const int64_t tm = (M + T - 1) / T, tn = (N + T - 1) / T;
    if (!lower_only) return tm * tn;
    const int64_t d = tm < tn ? tm : tn;          // tiles on or below the block diagonal
    return d * (d + 1) / 2 + (tm > tn ? (tm - tn) * tn : 0);
  };
  const int64_t t128 = tiles(128), t64 = tiles(64);
  const double cost128 = 0.03 + 0.14 * (double)((t128 + 255) / 256);
  const double cost64 = 0.035 + 0.038 * (double)t64 / 256.0;
  if (cost128 <= cost64) return gemm_dispatch<128>(g, ai, bj, s);
  return gemm_dispatch<64>(g, ai, bj, s);
}

// ---------------------------------------------------------------------------------------------
// GEMV (column-major A): 2-D grid of partial sums, then a fixed-order reduction (deterministic)
// ---------------------------------------------------------------------------------------------
constexpr int GEMV_CHUNK = 64;    // columns per partial (non-transposed)
constexpr int GEMV_TCHUNK = 256;  // rows per partial (transposed)

__global__ __launch_bounds__(64) void gemv_n_partial_kernel(int64_t m, int64_t n, const double* __restrict__ A,
                                                            int64_t lda, const double* __restrict__ x,
                                                            double* __restrict__ partial) {
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const int64_t j0 = (int64_t)blockIdx.y * GEMV_CHUNK;
  const int64_t j1 = j0 + GEMV_CHUNK < n ? j0 + GEMV_CHUNK : n;
  if (i >= m) return;
  double acc = 0.0;
  for (int64_t j = j0; j < j1; ++j) acc = fma(A[i + j * lda], x[j], acc);
  partial[(int64_t)blockIdx.y * m + i] = acc;
}

__global__ __launch_bounds__(64) void gemv_t_partial_kernel(int64_t m, int64_t n, const double* __restrict__ A,
                                                            int64_t lda, const double* __restrict__ x,
                                                            double* __restrict__ partial) {
  const int64_t j = blockIdx.x;
  const int64_t i0 = (int64_t)blockIdx.y * GEMV_TCHUNK;
  const int64_t i1 = i0 + GEMV_TCHUNK < m ? i0 + GEMV_TCHUNK : m;
  const int lane = threadIdx.x;
  double acc = 0.0;
  for (int64_t i = i0 + lane; i < i1; i += 64) acc = fma(A[i + j * lda], x[i], acc);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) partial[(int64_t)blockIdx.y * n + j] = acc;
}

__global__ __launch_bounds__(256) void gemv_reduce_kernel(const double* __restrict__ partial, int nparts, int64_t len,
                                                          double* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= len) return;
  double acc = 0.0;
  for (int p = 0; p < nparts; ++p) acc += partial[(int64_t)p * len + i];
  y[i] = acc;
}

int64_t gemv_work_doubles(bool trans, int64_t m, int64_t n) {
  return trans ? ((m + GEMV_TCHUNK - 1) / GEMV_TCHUNK) * n : ((n + GEMV_CHUNK - 1) / GEMV_CHUNK) * m;
}

int32_t gemv_f64(bool trans, int64_t m, int64_t n, const double* A, int64_t lda, const double* x, double* y,
                 double* work, hipStream_t s) {
  if (m <= 0 || n <= 0) return GSS_OK;
  if (!trans) {
    const int np = (int)((n + GEMV_CHUNK - 1) / GEMV_CHUNK);
    hipLaunchKernelGGL(gemv_n_partial_kernel, dim3((unsigned)((m + 63) / 64), (unsigned)np), dim3(64), 0, s, m, n, A,
                       lda, x, work);
    hipLaunchKernelGGL(gemv_reduce_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, work, np, m, y);
  } else {
    const int np = (int)((m + GEMV_TCHUNK - 1) / GEMV_TCHUNK);
    hipLaunchKernelGGL(gemv_t_partial_kernel, dim3((unsigned)n, (unsigned)np), dim3(64), 0, s, m, n, A, lda, x, work);
    hipLaunchKernelGGL(gemv_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, work, np, n, y);
  }
  GSS_HIP(hipGetLastError());
  return GSS_OK;
}

// ---------------------------------------------------------------------------------------------
// leaves: Cholesky and triangular inverse of one diagonal block (n <= 64), one wave, out of LDS
// ---------------------------------------------------------------------------------------------
constexpr int LEAF = 64;
constexpr int LEAF_LD = LEAF + 1;

// Lower-triangular inverse in LDS: X[c * LEAF_LD + i] = W(i, c); lane c owns column c.  S holds L row-major.
__device__ __forceinline__ void leaf_invert(const double* S, double* X, int n, int lane) {
  for (int i = lane; i < LEAF * LEAF_LD; i += 64) X[i] = 0.0;
  __syncthreads();
  double* x = X + lane * LEAF_LD;
  for (int i = 0; i < n; ++i) {
    // columns k < c hold zeros, so the sum may start at 0 for every lane (uniform trip count)
    double acc = 0.0;
    const double* row = S + i * LEAF_LD;
    for (int k = 0; k < i; ++k) acc = fma(row[k], x[k], acc);
    const double d = row[i];
    if (lane < n) {
      if (i == lane) x[i] = 1.0 / d;
      else if (i > lane) x[i] = -acc / d;
    }
  }
  __syncthreads();
}

__device__ __forceinline__ double bcast_lane(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}


// The same leaf on MFMA tiles (default).  The block, embedded in diag(A, I) when n < 64, is held by one wave as the
// upper block triangle of 16 x 16 tiles and factorised as A = U'U exactly like the moving-neighbourhood systems
// (tile16.h): diagonal tiles through potrf16_full, U_kj = V_k' A_kj, A_ij -= U_ki' U_kj.  L = U' goes back to A;
// W = inv(L) follows from  W_II = V_I',  W_JI = -V_J' sum_{K=I..J-1} U_KJ' W_KI  (J > I) -- again only X'Y products.
// 22 us instead of 51 us for the register/readlane leaf of round 1.
__global__ __launch_bounds__(64) void potrf_inv_leaf_tile_kernel(double* __restrict__ A, int n, int64_t lda,
                                                                 int row_offset, int* __restrict__ info,
                                                                 double* __restrict__ dinv, int64_t ldd, int full64) {
  __shared__ double S[16 * 17];
  __shared__ double S2[16 * 17];
  const int lane = threadIdx.x;
  const int g = lane >> 4, c = lane & 15;
  const int nt = (n + 15) >> 4;
  const d4_t zero4 = {0.0, 0.0, 0.0, 0.0};
  d4_t T[10], Vs[4], VTs[4];
#pragma unroll
  for (int I = 0; I < 4; ++I)
#pragma unroll
    for (int J = I; J < 4; ++J) {
      d4_t t = zero4;
      if (J < nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * I + g + 4 * r, col = 16 * J + c;
          double v = (row == col) ? 1.0 : 0.0;
          if (row < n && col < n) {
            const int hi = row > col ? row : col, lo = row > col ? col : row;
            v = A[hi + (int64_t)lo * lda];
          }
          t[r] = v;
        }
      }
      T[tile_id(I, J)] = t;
    }
  int bad = 0;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    if (kk < nt) {
      d4_t U, V, VT;
      int bc;
      potrf16_full(T[tile_id(kk, kk)], S, S2, lane, &U, &V, &VT, &bc);
      if (bc >= 0 && bad == 0) bad = row_offset + 16 * kk + bc + 1;
      T[tile_id(kk, kk)] = U;
      Vs[kk] = V;
      VTs[kk] = VT;
#pragma unroll
      for (int j = kk + 1; j < 4; ++j)
        if (j < nt) T[tile_id(kk, j)] = xty(V, T[tile_id(kk, j)], zero4);
#pragma unroll
      for (int i = kk + 1; i < 4; ++i) {
        if (i < nt) {
          const d4_t N = -T[tile_id(kk, i)];
#pragma unroll
          for (int j = i; j < 4; ++j)
            if (j < nt) T[tile_id(i, j)] = xty(N, T[tile_id(kk, j)], T[tile_id(i, j)]);
        }
      }
    }
  }
  if (bad != 0 && lane == 0 && *info == 0) *info = bad;
  // L = U': element (r, c) of U_IJ is L[16 J + c][16 I + r]
#pragma unroll
  for (int I = 0; I < 4; ++I)
#pragma unroll
    for (int J = I; J < 4; ++J) {
      if (J < nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int lrow = 16 * J + c, lcol = 16 * I + g + 4 * r;
          if (lrow < n && lcol < n && lrow >= lcol) A[lrow + (int64_t)lcol * lda] = T[tile_id(I, J)][r];
        }
      }
    }
  if (!dinv) return;
  // W = inv(L), lower block triangle; Wt[tile_id(I, J)] holds the block W_JI (rows of block J, columns of block I)
  d4_t Wt[10];
#pragma unroll
  for (int I = 0; I < 4; ++I) {
    Wt[tile_id(I, I)] = (I < nt) ? VTs[I] : zero4;
#pragma unroll
    for (int J = I + 1; J < 4; ++J) {
      d4_t w = zero4;
      if (J < nt) {
        d4_t acc = zero4;
#pragma unroll
        for (int K = I; K < J; ++K) acc = xty(T[tile_id(K, J)], Wt[tile_id(I, K)], acc);
        w = xty(-Vs[J], acc, zero4);
      }
      Wt[tile_id(I, J)] = w;
    }
  }
  // dinv (column-major, leading dimension ldd): zero above the diagonal; full64: the whole 64 x 64 block is written
  // (zero outside n x n), otherwise only the n x n block
#pragma unroll
  for (int I = 0; I < 4; ++I)
#pragma unroll
    for (int J = 0; J < 4; ++J)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int wrow = 16 * J + g + 4 * r, wcol = 16 * I + c;
        const bool inside = wrow < n && wcol < n;
        double v = 0.0;
        if (J >= I && inside) v = Wt[tile_id(I, J >= I ? J : I)][r];
        if (full64 || inside) dinv[wrow + (int64_t)wcol * ldd] = v;
      }
}

// The same leaf for blocks of up to 128 x 128, one workgroup of NW waves: the 8 x 8 grid of 16 x 16 tiles lives in
// LDS in tile layout (element of lane l, register r at [r * 64 + l]: conflict-free).  The leaf is the latency chain of
// every fit (eight of them per 1 024 rows), so it is arranged around its one sequential part, the sixteen-column
// factorisation of the diagonal tiles (potrf16_full, 2.5 us each, wave 0):
//   R(k)  all waves: row solve U_kj = V_k' A_kj (j > k); each solved tile goes to global memory at once;
//   T(k)  wave 0: A_k+1,k+1 -= U_k,k+1' U_k,k+1, then its factorisation (U, V = U^-1, V');
//         the other waves meanwhile: the rest of the trailing update of step k, and row k of W = inv(L):
//         W_kI = -V_k' sum_{K = I .. k-1} U_Kk' W_KI, kept in registers until everybody has read column k of U,
//         whose slots it then takes (W_KI sits in slot (I, K)).
// Two LDS-only barriers per step (fence on the local address space: the global stores issued along the way keep
// draining behind them); the loads of the block are all in flight before the first tile is stored.
constexpr int L128_TILES = 36;  // upper block triangle of 8 x 8
constexpr size_t L128_LDS_BYTES = sizeof(double) * ((L128_TILES + 16) * 256 + 2 * 16 * 17);
__host__ __device__ constexpr int tile_id8(int i, int j) { return i * 8 - (i * (i - 1)) / 2 + (j - i); }

__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// NW waves (8 = two per SIMD); dbg: time stamps of wave 0 (GSS_PANEL_TIMES)
template <int NW>
__device__ __forceinline__ void leaf128_body(double* __restrict__ A, int n, int64_t lda, int row_offset,
                                             int* __restrict__ info, double* __restrict__ dinv, int64_t ldd,
                                             double* __restrict__ sm128, long long* dbg = nullptr) {
  static_assert(NW >= 2, "one wave factors, the others update");
  // waves that share the trailing update and the rows of W: not wave 0, and not wave 4 either, which sits on wave 0's
  // SIMD -- FP64 MFMAs and FP64 vector instructions use the same pipe, its MFMAs would stretch the factorisation chain
  constexpr int NWK = NW > 4 ? NW - 2 : NW - 1;
  constexpr int LMAX = (L128_TILES + NW - 1) / NW;  // tiles per wave when the block is loaded
  constexpr int WMAX = (7 + NWK - 1) / NWK;        // tiles of a row of W per wave
  // wave 0's chain carries no global stores when there is an idle wave: wave SW writes out the diagonal tiles (from
  // LDS, one step later) and the one row tile wave 0 solves itself
  constexpr int SW = NW > 4 ? 4 : 0;
  int ndbg = 0;
  auto stamp = [&] {
    if (dbg && threadIdx.x == 0) dbg[ndbg++] = (long long)wall_clock64();
  };
  stamp();
  double* Tl = sm128;                  // 36 tiles: U (upper block triangle), later W below the diagonal
  double* Vl = Tl + L128_TILES * 256;  // V_k = U_kk^-1
  double* VTl = Vl + 8 * 256;          // V_k'
  double* S = VTl + 8 * 256;
  double* S2 = S + 16 * 17;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int nt = (n + 15) >> 4;
  const d4_t zero4 = {0.0, 0.0, 0.0, 0.0};
  auto ld = [&](const double* base, int id) {
    d4_t t;
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = base[id * 256 + r * 64 + lane];
    return t;
  };
  auto st = [&](double* base, int id, const d4_t& t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) base[id * 256 + r * 64 + lane] = t[r];
  };
  // L = U': element (rho, gamma) of U_IJ is L[16 J + gamma][16 I + rho]
  auto store_L = [&](int I, int J, const d4_t& u) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int lrow = 16 * J + c, lcol = 16 * I + g + 4 * r;
      if (lrow < n && lcol < n && lrow >= lcol) A[lrow + (int64_t)lcol * lda] = u[r];
    }
  };
  // W_JI from its LDS tile (element (rho, gamma) at [(rho & 3) * 16 + gamma + (rho >> 2) * 64]), read transposed so
  // that the lanes run along the rows of W: 128-byte column pieces.  The caller has just written the tile itself.
  auto store_W = [&](int J, int I, const double* src) {
    tile_sync<true>();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int wrow = 16 * J + c, wcol = 16 * I + g + 4 * r;
      const double v = src[(c >> 2) * 64 + (c & 3) * 16 + g + 4 * r];
      if (wrow < n && wcol < n) dinv[wrow + (int64_t)wcol * ldd] = v;
    }
  };
  // ---- block -> tiles (lower triangle of A is the storage; rows / columns beyond n are the identity)
  {
    d4_t t[LMAX];
    int tid[LMAX];
#pragma unroll
    for (int u = 0; u < LMAX; ++u) {
      int idx = wave + u * NW, I = 0, cnt = nt;
      const bool have = idx < nt * (nt + 1) / 2;
      if (!have) idx = 0;
      while (idx >= cnt) {
        idx -= cnt;
        ++I;
        --cnt;
      }
      const int J = I + idx;
      tid[u] = have ? tile_id8(I, J) : -1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * I + g + 4 * r, col = 16 * J + c;
        const bool ok = row < n && col < n;
        const int hi = row > col ? row : col, lo = row > col ? col : row;
        const double v = A[ok ? hi + (int64_t)lo * lda : (int64_t)0];
        t[u][r] = ok ? v : (row == col ? 1.0 : 0.0);
      }
    }
#pragma unroll
    for (int u = 0; u < LMAX; ++u)
      if (tid[u] >= 0) st(Tl, tid[u], t[u]);
  }
  lds_barrier();
  stamp();
  int bad = 0;
  d4_t Vmine = zero4, umine = zero4;  // wave 0: V of the tile it factored last, the row tile it solved last
  auto factor_tile = [&](int kk, const d4_t& t) {  // wave 0
    d4_t U, V, VT;
    int bc;
    potrf16_full<true>(t, S, S2, lane, &U, &V, &VT, &bc);
    Vmine = V;
    if (bc >= 0 && bad == 0) bad = row_offset + 16 * kk + bc + 1;
    st(Tl, tile_id8(kk, kk), U);
    st(Vl, kk, V);
    st(VTl, kk, VT);
    if (SW == 0) {  // (with more than four waves the idle wave SW takes the tile to global memory, see R(kk))
      store_L(kk, kk, U);
      if (dinv) store_W(kk, kk, VTl + kk * 256);
    }
  };
  if (wave == 0) factor_tile(0, ld(Tl, tile_id8(0, 0)));
  lds_barrier();
  stamp();
  d4_t wreg[WMAX];
  for (int kk = 0; kk < nt; ++kk) {
    // ---- R(kk): row solve (wave 0 keeps V_kk and the tile it solves in registers: no LDS round trips on its chain)
    {
      const d4_t V = wave == 0 ? Vmine : ld(Vl, kk);
      for (int j = kk + 1 + wave; j < nt; j += NW) {
        const d4_t u = xty(V, ld(Tl, tile_id8(kk, j)), zero4);
        st(Tl, tile_id8(kk, j), u);
        if (SW == 0 || wave != 0) store_L(kk, j, u);
        if (wave == 0) umine = u;  // (kk, kk + 1): wave 0 has one tile per row at most
      }
      if (SW != 0 && wave == SW) {  // the diagonal tile factored one phase ago
        store_L(kk, kk, ld(Tl, tile_id8(kk, kk)));
        if (dinv) store_W(kk, kk, VTl + kk * 256);
      }
    }
    lds_barrier();
    // ---- T(kk)
    const int m = nt - 1 - kk;                                  // rows below
    const int ntr = m > 0 ? m * (m + 1) / 2 - 1 : 0;            // trailing tiles without (kk+1, kk+1)
    if (wave == 0) {
      if (m > 0) {
        factor_tile(kk + 1, xty(-umine, umine, ld(Tl, tile_id8(kk + 1, kk + 1))));
      }
      stamp();
    } else if (wave == 4) {
      if (SW != 0 && m > 0) store_L(kk, kk + 1, ld(Tl, tile_id8(kk, kk + 1)));  // the row tile wave 0 solved
    } else {
      const int w = wave < 4 ? wave - 1 : wave - 2;
      int q = 0;
      for (int i = kk + 1; i < nt; ++i)
        for (int j = i; j < nt; ++j) {
          if (i == kk + 1 && j == kk + 1) continue;
          if ((q++ % NWK) != w) continue;
          const d4_t N = -ld(Tl, tile_id8(kk, i));
          st(Tl, tile_id8(i, j), xty(N, ld(Tl, tile_id8(kk, j)), ld(Tl, tile_id8(i, j))));
        }
      if (dinv && kk >= 1) {
        const d4_t NV = -ld(Vl, kk);
        const int I0 = ((w - ntr) % NWK + NWK) % NWK;  // chain I belongs to worker (ntr + I) mod NWK
#pragma unroll
        for (int u = 0; u < WMAX; ++u) {
          const int I = I0 + u * NWK;
          if (I < kk) {
            d4_t acc = zero4;
            for (int K = I; K < kk; ++K) {
              const d4_t Wki = (K == I) ? ld(VTl, I) : ld(Tl, tile_id8(I, K));
              acc = xty(ld(Tl, tile_id8(K, kk)), Wki, acc);
            }
            wreg[u] = xty(NV, acc, zero4);
          }
        }
      }
    }
    lds_barrier();  // every wave has read what it needs of column kk of U
    if (wave != 0 && wave != 4 && dinv && kk >= 1) {
      const int w = wave < 4 ? wave - 1 : wave - 2;
      const int I0 = ((w - ntr) % NWK + NWK) % NWK;
#pragma unroll
      for (int u = 0; u < WMAX; ++u) {
        const int I = I0 + u * NWK;
        if (I < kk) {
          st(Tl, tile_id8(I, kk), wreg[u]);
          store_W(kk, I, Tl + tile_id8(I, kk) * 256);
        }
      }
    }
    // (the W tiles written above are first read in T(kk+1), behind the barrier of R(kk+1))
  }
  if (wave == 0 && bad != 0 && lane == 0 && *info == 0) *info = bad;
  stamp();
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void potrf_inv_leaf128_kernel(double* __restrict__ A, int n, int64_t lda, int row_offset,
                                                                int* __restrict__ info, double* __restrict__ dinv,
                                                                int64_t ldd) {
  extern __shared__ __attribute__((aligned(16))) double sm128[];
  leaf128_body<NW>(A, n, lda, row_offset, info, dinv, ldd, sm128);
}

// ---------------------------------------------------------------------------------------------------------------
// Factor and inverse of a block of up to PANEL_MAX rows in ONE launch (the fit at n = 1 000 and every 1 024-column
// panel of the LUGS factorisations used to be a chain of ~95 dependent launches: 8 leaves of 58 us and ~30 GEMMs of
// 10-30 us on a few of the 256 CUs each).  A grid of a few dozen workgroups walks the 128 x 128 block columns; what
// lies between two leaves is kept as short as the data dependence allows, everything else runs beside the next leaf:
//   leaf k        workgroup 0: L_kk and W_kk = inv(L_kk) (the leaf above as a device function).
//                 The other workgroups meanwhile finish step k-1 in two rounds with a barrier of their own between:
//                   B1  L_i,k-1 = A_i,k-1 W_k-1,k-1' for the rows i > k (into the scratch P) and row k-1 of the
//                       inverse, W_k-1,j = -W_k-1,k-1 T_k-1,j;
//                   B2  every trailing update of step k-1 except the diagonal block (k, k), P -> A, and the
//                       contributions of row k-1 of W to the running sums T_ij += L_i,k-1 W_k-1,j (i >= k, j < k);
//   grid barrier, L_k+1,k = A_k+1,k W_kk' (64 tiles), grid barrier, A_k+1,k+1 -= L_k+1,k L_k+1,k' (36 tiles),
//   grid barrier, leaf k+1.
// Barriers are one agent-scope atomic per workgroup (tools/probe_gridsync.hip).  Work items are single MFMA tiles
// read straight from global memory (L2): the kernel is latency-bound, its 0.7 GFLOP would take 9 us at the rate of
// K3.  Tiles are loaded "transposed" (lane c runs along a column: 128-byte pieces), which is the orientation
// acc + X'Y wants for D = A B' with all three operands coalesced; the one product that is not of that shape
// (T += L W) reads its W tiles with 32-byte pieces.
constexpr int PANEL_NB = 128;
constexpr int PANEL_WAVES = 8;

// Tile loads.  Transposed: element (rho, gamma) of the tile <- M(r0 + gamma, c0 + rho) (lane c runs along a column
// of M); NATURAL: element (rho, gamma) <- M(r0 + rho, c0 + gamma).  r0, c0 and ld are wave-uniform (the wave index
// goes through readfirstlane), so a tile is four loads from scalar bases with one lane offset per leading dimension,
// and the sixteen tiles of a product are all in flight before its first MFMA.  No guards: the kernel works on whole
// tiles (n a multiple of 16, or rows and columns up to the next multiple present in memory and zero).
template <bool NATURAL>
__device__ __forceinline__ d4_t tile_ld(const double* __restrict__ M, int ld, int r0, int c0, int g, int c) {
  d4_t t;
  const double* __restrict__ Mu = M + (unsigned)(r0 + c0 * ld);
  const unsigned lo = NATURAL ? (unsigned)(g + c * ld) : (unsigned)(c + g * ld);
#pragma unroll
  for (int r = 0; r < 4; ++r) t[r] = (Mu + (unsigned)(NATURAL ? 4 * r : 4 * r * ld))[lo];
  return t;
}
__device__ __forceinline__ void tile_st(double* __restrict__ M, int ld, int r0, int c0, int g, int c, const d4_t& t) {
  double* __restrict__ Mu = M + (unsigned)(r0 + c0 * ld);
  const unsigned lo = (unsigned)(c + g * ld);
#pragma unroll
  for (int r = 0; r < 4; ++r)  // write-through (sc1): nothing dirty is left in this XCD's L2 for the next barrier to flush
    __hip_atomic_store(&(Mu + (unsigned)(4 * r * ld))[lo], t[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// acc (+/-)= sum over the k-tiles tp_lo .. tp_hi of  X_tp' Y_tp  with X = tile (xr0, xc0 + 16 tp) of MX and Y = tile
// (yr0, yc0 + 16 tp) of MY (k runs along the columns of both: D = A B' in the transposed tile orientation);
// YNAT: Y = natural tile (yr0 + 16 tp, yc0) of MY instead (k along its rows: the product T = L W).  All eight
// k-tiles are loaded (the skipped ones below tp_lo are structural zeros in memory, those above tp_hi repeat tile tp_hi).
template <bool NEG, bool YNAT>
__device__ __forceinline__ d4_t tile_dot(d4_t acc, const double* __restrict__ MX, int ldx, int xr0, int xc0,
                                         const double* __restrict__ MY, int ldy, int yr0, int yc0, int tp_lo, int tp_hi,
                                         int g, int c) {
  d4_t X[8], Y[8];
#pragma unroll
  for (int tp = 0; tp < 8; ++tp) {
    const int tq = tp < tp_hi ? tp : tp_hi;  // never beyond the last k-tile that is used (the block may end there)
    X[tp] = tile_ld<false>(MX, ldx, xr0, xc0 + 16 * tq, g, c);
    Y[tp] = YNAT ? tile_ld<true>(MY, ldy, yr0 + 16 * tq, yc0, g, c) : tile_ld<false>(MY, ldy, yr0, yc0 + 16 * tq, g, c);
  }
#pragma unroll
  for (int tp = 0; tp < 8; ++tp)
    if (tp >= tp_lo && tp <= tp_hi) acc = xty(NEG ? -X[tp] : X[tp], Y[tp], acc);
  return acc;
}

// cnt: arrivals (monotone) of the `members` workgroups taking part; dead: set when a wait gave up (every later
// barrier falls through and the grid drains)
__device__ __forceinline__ void panel_sync(unsigned* cnt, unsigned* dead, unsigned& epoch, unsigned members, int* info) {
  __syncthreads();
  ++epoch;
  if (threadIdx.x == 0) {
    __threadfence();
    atomicAdd(cnt, 1u);
    const unsigned target = epoch * members;
    unsigned spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (__hip_atomic_load(dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
      if (++spins > 4000000u) {
        __hip_atomic_store(dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicCAS(info, 0, -1);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    __threadfence();
  }
  __syncthreads();
}

// lower tiles of a diagonal block, 36 of them: idx -> (ta >= tb)
__device__ __forceinline__ void lower_tile(int idx, int& ta, int& tb) {
  tb = 0;
  int cnt = 8;
  while (idx >= cnt) {
    idx -= cnt;
    ++tb;
    --cnt;
  }
  ta = tb + idx;
}

// The leading dimensions are re-read through an opaque copy at the top of every phase: otherwise the address
// arithmetic of all phases is hoisted out of the step loop and stays live across the leaf.
#define PANEL_OPAQUE_LDS()                                   \
  int lda = (int)lda_, ldw = (int)ldw_, ldp = ldp_;          \
  asm volatile("" : "+s"(lda), "+s"(ldw), "+s"(ldp))

// n: rows of the block as the tiles see it (a multiple of 16); nreal <= n: rows of the matrix (the leaf treats the
// rest of its block as the identity; in memory rows and columns nreal .. n-1 are zeros and stay zeros).
// scr: P (ldp x 128: column k of L while step k is being finished) and T (ldp x ldp: slab i holds T_i.' as
// TT_i(b, a) = T_ib-block(a, b)), ldp = 128 ceil(n / 128).  bar: [0] grid arrivals, [1] gave-up flag, [2] time stamps
// wanted, [4] arrivals of the barrier among the workgroups 1 .. G-1, [8] arrivals of the team's barrier, [12] workgroups
// that have left, [16 ...] stamps.
template <int NW>
__global__ __launch_bounds__(64 * NW) void potrf_inv_panel_kernel(double* __restrict__ A, int n, int nreal, int64_t lda_,
                                                                  double* __restrict__ W, int64_t ldw_,
                                                                  double* __restrict__ scr, int row_offset,
                                                                  int* __restrict__ info, unsigned* __restrict__ bar) {
  extern __shared__ __attribute__((aligned(16))) double sm128[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int G = gridDim.x;
  const int nbk = (n + PANEL_NB - 1) / PANEL_NB;
  const int ldp_ = nbk * PANEL_NB;
  double* P = scr;
  double* TT = scr + (int64_t)ldp_ * PANEL_NB;
  const d4_t zero4 = {0.0, 0.0, 0.0, 0.0};
  unsigned epoch = 0, epoch_bg = 0, epoch_team = 0;
  const int team = G < 16 ? G : 16;
  long long* stamps = (bar[2] != 0u && blockIdx.x == 0 && threadIdx.x == 0) ? reinterpret_cast<long long*>(bar + 16) : nullptr;
  int nstamp = 0;
  auto stamp = [&] {
    if (stamps) stamps[nstamp++] = (long long)wall_clock64();
  };
  stamp();

  for (int k = 0; k < nbk; ++k) {
    const int k0 = k * PANEL_NB;
    // ---------------- leaf k, and the rest of step k-1 beside it
    if (blockIdx.x == 0) {
      const int nk = nreal - k0 < PANEL_NB ? nreal - k0 : PANEL_NB;
      leaf128_body<NW>(A + k0 + (int64_t)k0 * lda_, nk, lda_, row_offset + k0, info, W + k0 + (int64_t)k0 * ldw_, ldw_,
                       sm128, (stamps && k == 1) ? stamps + 128 : nullptr);
    } else if (k >= 1) {
      const int nw = (G - 1) * NW, wid = (blockIdx.x - 1) + (G - 1) * wave;  // consecutive tasks on different CUs
      const int kp = k - 1, kp0 = k0 - PANEL_NB;
      {  // B1
        PANEL_OPAQUE_LDS();
        int t = wid, base = 0;
        for (int i = k + 1; i < nbk; ++i) {   // L_i,kp = A_i,kp W_kp,kp' -> P
          for (; t < base + 64; t += nw) {
            const int ta = (t - base) & 7, tb = (t - base) >> 3;
            const int ar = i * PANEL_NB + 16 * ta;
            if (ar < n) {
              const d4_t acc = tile_dot<false, false>(zero4, W, ldw, kp0 + 16 * tb, kp0, A, lda, ar, kp0, 0, tb, g, c);
              tile_st(P, ldp, ar, 16 * tb, g, c, acc);
            }
          }
          base += 64;
        }
        const double* TTk = TT + (int64_t)kp * ldp * PANEL_NB;
        for (int j = 0; j < kp; ++j) {        // W_kp,j = -W_kp,kp T_kp,j (block kp is a full block: kp < nbk - 1)
          for (; t < base + 64; t += nw) {
            const int ta = (t - base) & 7, tb = (t - base) >> 3;
            const int ar = kp0 + 16 * ta, bc = j * PANEL_NB + 16 * tb;
            const d4_t acc = tile_dot<true, false>(zero4, TTk, ldp, bc, 0, W, ldw, ar, kp0, 0, ta, g, c);
            tile_st(W, ldw, ar, bc, g, c, acc);
          }
          base += 64;
        }
      }
      panel_sync(bar + 4, bar + 1, epoch_bg, G - 1, info);
      {  // B2
        PANEL_OPAQUE_LDS();
        int t = wid, base = 0;
        // trailing updates of step kp, all but the diagonal block (k, k): A_ij -= L_i,kp L_j,kp' from P
        for (int j = k; j < nbk; ++j)
          for (int i = (j == k ? k + 1 : j); i < nbk; ++i) {
            const int ntile = (i == j) ? 36 : 64;
            for (; t < base + ntile; t += nw) {
              int ta, tb;
              if (i == j) lower_tile(t - base, ta, tb);
              else {
                ta = (t - base) & 7;
                tb = (t - base) >> 3;
              }
              const int ar = i * PANEL_NB + 16 * ta, br = j * PANEL_NB + 16 * tb;
              if (ar < n && br < n) {
                d4_t acc = tile_ld<false>(A, lda, ar, br, g, c);
                acc = tile_dot<true, false>(acc, P, ldp, br, 0, P, ldp, ar, 0, 0, 7, g, c);
                tile_st(A, lda, ar, br, g, c, acc);
              }
            }
            base += ntile;
          }
        // T_ij (+)= L_i,kp W_kp,j for the rows i >= k and j <= kp (natural orientation (a, b) -> TT_i(j0 + b, a));
        // j == kp starts the sum with the lower-triangular W_kp,kp (k-tiles from tb on)
        for (int i = k; i < nbk; ++i) {
          double* TTi = TT + (int64_t)i * ldp * PANEL_NB;
          for (int j = 0; j <= kp; ++j) {
            for (; t < base + 64; t += nw) {
              const int ta = (t - base) & 7, tb = (t - base) >> 3;
              const int ar = i * PANEL_NB + 16 * ta, bc = j * PANEL_NB + 16 * tb;
              if (ar < n) {
                d4_t acc = (j == kp) ? zero4 : tile_ld<false>(TTi, ldp, bc, 16 * ta, g, c);
                acc = tile_dot<false, true>(acc, P, ldp, ar, 0, W, ldw, kp0, bc, j == kp ? tb : 0, 7, g, c);
                tile_st(TTi, ldp, bc, 16 * ta, g, c, acc);
              }
            }
            base += 64;
          }
        }
        // P -> A_i,kp (nobody reads column kp of A in this round)
        for (int i = k; i < nbk; ++i) {
          for (; t < base + 64; t += nw) {
            const int ta = (t - base) & 7, tb = (t - base) >> 3;
            const int ar = i * PANEL_NB + 16 * ta;
            if (ar < n) tile_st(A, lda, ar, kp0 + 16 * tb, g, c, tile_ld<false>(P, ldp, ar, 16 * tb, g, c));
          }
          base += 64;
        }
      }
    }
    stamp();
    panel_sync(bar, bar + 1, epoch, G, info);
    stamp();
    if (k == nbk - 1) {
      // ---------------- the last row of the inverse: W_kj = -W_kk T_kj (the k-tiles stop at the row tile, inside n)
      PANEL_OPAQUE_LDS();
      (void)lda;
      const int nw = G * NW;
      int t = blockIdx.x + G * wave, base = 0;
      const double* TTk = TT + (int64_t)k * ldp * PANEL_NB;
      for (int j = 0; j < k; ++j) {
        for (; t < base + 64; t += nw) {
          const int ta = (t - base) & 7, tb = (t - base) >> 3;
          const int ar = k0 + 16 * ta, bc = j * PANEL_NB + 16 * tb;
          if (ar < n) {
            const d4_t acc = tile_dot<true, false>(zero4, TTk, ldp, bc, 0, W, ldw, ar, k0, 0, ta, g, c);
            tile_st(W, ldw, ar, bc, g, c, acc);
          }
        }
        base += 64;
      }
      stamp();
      break;
    }
    // ---------------- the chain to the next leaf, on a team of a few workgroups (sixteen arrivals at a barrier cost
    // 1.5 us, sixty-four 5 us: the atomics on one word serialise); the rest go straight on to B1, which touches
    // nothing these two steps touch, and meet the team again at the barrier between B1 and B2
    if ((int)blockIdx.x < team) {
      {  // L_k+1,k = A_k+1,k W_kk' -> P
        PANEL_OPAQUE_LDS();
        const int nw = team * NW;
        for (int t = blockIdx.x + team * wave; t < 64; t += nw) {
          const int ta = t & 7, tb = t >> 3;
          const int ar = k0 + PANEL_NB + 16 * ta;
          if (ar < n) {
            const d4_t acc = tile_dot<false, false>(zero4, W, ldw, k0 + 16 * tb, k0, A, lda, ar, k0, 0, tb, g, c);
            tile_st(P, ldp, ar, 16 * tb, g, c, acc);
          }
        }
      }
      stamp();
      panel_sync(bar + 8, bar + 1, epoch_team, team, info);
      stamp();
      {  // A_k+1,k+1 -= L_k+1,k L_k+1,k'
        PANEL_OPAQUE_LDS();
        (void)ldw;
        const int nw = team * NW;
        for (int t = blockIdx.x + team * wave; t < 36; t += nw) {
          int ta, tb;
          lower_tile(t, ta, tb);
          const int ar = k0 + PANEL_NB + 16 * ta, br = k0 + PANEL_NB + 16 * tb;
          if (ar < n && br < n) {
            d4_t acc = tile_ld<false>(A, lda, ar, br, g, c);
            acc = tile_dot<true, false>(acc, P, ldp, br, 0, P, ldp, ar, 0, 0, 7, g, c);
            tile_st(A, lda, ar, br, g, c, acc);
          }
        }
      }
      stamp();
      panel_sync(bar + 8, bar + 1, epoch_team, team, info);
      stamp();
    }
  }
  // the last workgroup to leave puts the barrier words back to zero for the next launch on this stream (everybody
  // is out of every barrier by then): no memset in front of each launch
  __syncthreads();
  if (threadIdx.x == 0) {
    if (atomicAdd(&bar[12], 1u) == (unsigned)G - 1u) {
      __hip_atomic_store(&bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&bar[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&bar[4], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&bar[8], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&bar[12], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
#undef PANEL_OPAQUE_LDS

// W = inv(L) for one lower-triangular block (used when no cached inverse exists)
__global__ __launch_bounds__(64) void trtri_leaf_kernel(const double* __restrict__ L, int n, int64_t ldl,
                                                        double* __restrict__ W, int64_t ldw) {
  __shared__ double S[LEAF * LEAF_LD];
  __shared__ double X[LEAF * LEAF_LD];
  const int lane = threadIdx.x;
  for (int idx = lane; idx < n * n; idx += 64) {
    const int i = idx % n, j = idx / n;
    S[i * LEAF_LD + j] = (j <= i) ? L[i + (int64_t)j * ldl] : 0.0;
  }
  __syncthreads();
  leaf_invert(S, X, n, lane);
  for (int idx = lane; idx < n * n; idx += 64) {
    const int i = idx % n, j = idx / n;
    W[i + (int64_t)j * ldw] = (j <= i) ? X[j * LEAF_LD + i] : 0.0;
  }
}

// W block <- cached inverse block (ld = 64)
__global__ __launch_bounds__(256) void copy_leaf_kernel(const double* __restrict__ dinv, int n, double* __restrict__ W,
                                                        int64_t ldw) {
  for (int idx = threadIdx.x; idx < n * n; idx += 256) {
    const int i = idx % n, j = idx / n;
    W[i + (int64_t)j * ldw] = dinv[i + j * LEAF];
  }
}

static int64_t split_point(int64_t n) {
  // first half size: multiple of LEAF, at least LEAF, roughly n/2
  int64_t h = (n / 2 + LEAF - 1) / LEAF * LEAF;
  if (h >= n) h = n - 1 > LEAF ? (n - 1) / LEAF * LEAF : LEAF;
  if (h <= 0) h = LEAF;
  return h;
}

int32_t trtri_f64(const double* L, int64_t n, int64_t ldl, double* W, int64_t ldw, double* T, const double* dinv,
                  hipStream_t s) {
  if (n <= 0) return GSS_OK;
  if (n <= LEAF) {
    if (dinv) hipLaunchKernelGGL(copy_leaf_kernel, dim3(1), dim3(256), 0, s, dinv, (int)n, W, ldw);
    else hipLaunchKernelGGL(trtri_leaf_kernel, dim3(1), dim3(64), 0, s, L, (int)n, ldl, W, ldw);
    GSS_HIP(hipGetLastError());
    return GSS_OK;
  }
  const int64_t n1 = split_point(n), n2 = n - n1;
  const double* L21 = L + n1;
  const double* L22 = L + n1 + n1 * ldl;
  double* W21 = W + n1;
  double* W22 = W + n1 + n1 * ldw;
  const double* dinv2 = dinv ? dinv + (n1 / LEAF) * LEAF * LEAF : nullptr;
  GSS_TRY(trtri_f64(L, n1, ldl, W, ldw, T, dinv, s));
  GSS_TRY(trtri_f64(L22, n2, ldl, W22, ldw, T, dinv2, s));
  // T (n2 x n1, row-major) = L21 * W11
  GSS_TRY(gemm_f64(n2, n1, n1, 1.0, L21, 1, ldl, W, 1, ldw, 0.0, T, n1, 1, false, s));
  // W21 = -W22 * T
  GSS_TRY(gemm_f64(n2, n1, n2, -1.0, W22, 1, ldw, T, n1, 1, 0.0, W21, 1, ldw, false, s));
  return GSS_OK;
}

// dinv != nullptr: cached inverses of L's diagonal leaf blocks (block b = rows 64 b ..), else computed in scratch
int32_t trsm_right_lt_f64(double* X, int64_t m, int64_t n, int64_t ldx, const double* L, int64_t ldl,
                          double* scratch, const double* dinv, hipStream_t s) {
  if (m <= 0 || n <= 0) return GSS_OK;
  if (n <= LEAF) {
    const double* inv = dinv;
    if (!inv) {
      hipLaunchKernelGGL(trtri_leaf_kernel, dim3(1), dim3(64), 0, s, L, (int)n, ldl, scratch, (int64_t)LEAF);
      GSS_HIP(hipGetLastError());
      inv = scratch;
    }
    // D(i,j) = sum_k X(i,k) Winv(j,k): one column block (n <= BN), so each workgroup reads all of its
    // rows of X before writing them back
    return gemm_f64(m, n, n, 1.0, X, 1, ldx, inv, LEAF, 1, 0.0, X, 1, ldx, false, s);
  }
  const int64_t n1 = split_point(n), n2 = n - n1;
  double* X1 = X;
  double* X2 = X + n1 * ldx;
  const double* L21 = L + n1;
  const double* L22 = L + n1 + n1 * ldl;
  const double* dinv2 = dinv ? dinv + (n1 / LEAF) * LEAF * LEAF : nullptr;
  GSS_TRY(trsm_right_lt_f64(X1, m, n1, ldx, L, ldl, scratch, dinv, s));
  // X2 -= X1 * L21'
  GSS_TRY(gemm_f64(m, n2, n1, -1.0, X1, 1, ldx, L21, ldl, 1, 1.0, X2, 1, ldx, false, s));
  GSS_TRY(trsm_right_lt_f64(X2, m, n2, ldx, L22, ldl, scratch, dinv2, s));
  return GSS_OK;
}

static int32_t potrf_rec(double* A, int64_t n, int64_t lda, int64_t row_offset, int* d_info, double* dinv,
                         hipStream_t s) {
  if (n <= 0) return GSS_OK;
  if (n <= LEAF) {
    hipLaunchKernelGGL(potrf_inv_leaf_tile_kernel, dim3(1), dim3(64), 0, s, A, (int)n, lda, (int)row_offset, d_info,
                       dinv, (int64_t)LEAF, 1);
    GSS_HIP(hipGetLastError());
    return GSS_OK;
  }
  const int64_t n1 = split_point(n), n2 = n - n1;
  double* A21 = A + n1;
  double* A22 = A + n1 + n1 * lda;
  GSS_TRY(potrf_rec(A, n1, lda, row_offset, d_info, dinv, s));
  GSS_TRY(trsm_right_lt_f64(A21, n2, n1, lda, A, lda, nullptr, dinv, s));
  // A22 -= A21 * A21'   (lower tiles)
  GSS_TRY(gemm_f64(n2, n2, n1, -1.0, A21, 1, lda, A21, lda, 1, 1.0, A22, 1, lda, true, s));
  GSS_TRY(potrf_rec(A22, n2, lda, row_offset + n1, d_info, dinv + (n1 / LEAF) * LEAF * LEAF, s));
  return GSS_OK;
}

int64_t potrf_dinv_doubles(int64_t n) { return ((n + LEAF - 1) / LEAF) * LEAF * LEAF; }

// dinv: caller workspace of potrf_dinv_doubles(n) doubles receiving the inverses of the diagonal leaf blocks
// (they are needed by the panel solves anyway); asynchronous on s, *d_info is zeroed first.
int32_t potrf_f64(double* A, int64_t n, int64_t lda, int* d_info, double* dinv, hipStream_t s) {
  GSS_TRY(dev_zero_bytes(d_info, sizeof(int), s));
  return potrf_rec(A, n, lda, 0, d_info, dinv, s);
}

// Cholesky factor and its inverse in one recursion: the inverse of the leading block turns the panel solve into one
// GEMM, so an internal node costs four GEMMs (L21 = A21 W11', A22 -= L21 L21', T = L21 W11, W21 = -W22 T) instead of the
// ~9 launches of separate potrf / trsm / trtri recursions.  W (n x n, ldw, strict upper triangle zero on entry)
// receives inv(L); A is destroyed (only its diagonal leaf blocks end up holding L).  scr: n * n doubles.
// dst (m x n, ldd) <- src (m x n, lds), column-major
__global__ __launch_bounds__(256) void copy_block_kernel(const double* __restrict__ src, int64_t lds, int64_t m,
                                                         int64_t n, double* __restrict__ dst, int64_t ldd) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t j = blockIdx.y;
  if (i < m && j < n) dst[i + j * ldd] = src[i + j * lds];
}

static int32_t copy_block(const double* src, int64_t lds, int64_t m, int64_t n, double* dst, int64_t ldd,
                          hipStream_t s) {
  if (m <= 0 || n <= 0) return GSS_OK;
  hipLaunchKernelGGL(copy_block_kernel, dim3((unsigned)((m + 255) / 256), (unsigned)n), dim3(256), 0, s, src, lds, m, n,
                     dst, ldd);
  GSS_HIP(hipGetLastError());
  return GSS_OK;
}

// barrier words of the panel kernel, one pair per stream (launches on one stream are ordered)
static unsigned* panel_barrier_words(hipStream_t s) {
  static std::mutex mu;
  static std::unordered_map<hipStream_t, unsigned*> words;
  std::lock_guard<std::mutex> lock(mu);
  auto it = words.find(s);
  if (it != words.end()) return it->second;
  unsigned* p = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&p), 4096) != hipSuccess) return nullptr;
  if (hipMemset(p, 0, 4096) != hipSuccess) {
    (void)hipFree(p);
    return nullptr;
  }
  words.emplace(s, p);
  return p;
}

constexpr int64_t PANEL_MIN = 257;
static int64_t panel_max_rows() {
  static const int64_t v = [] {
    const char* e = std::getenv("GSS_PANEL_MAX");  // 0 switches the single-launch panel off
    return e ? (int64_t)std::atoll(e) : (int64_t)1024;
  }();
  return v;
}
static int panel_workgroups() {
  static const int v = [] {
    const char* e = std::getenv("GSS_PANEL_WGS");
    int w = e ? std::atoi(e) : 64;
    return w < 2 ? 2 : (w > 256 ? 256 : w);
  }();
  return v;
}
// set when a launch reported -1 (its workgroups were not all resident within the bounded spin, e.g. many such
// kernels at once): the process goes back to the launch-per-block recursion from then on
static std::atomic<bool> g_panel_off{false};
static std::atomic<int> g_panel_giveups{0};   // gss_stat "panel_giveups"
void potrf_panel_disable() {
  g_panel_off.store(true);
  g_panel_giveups.fetch_add(1);
}
int panel_giveups() { return g_panel_giveups.load(); }
static bool use_panel(int64_t n) { return !g_panel_off.load() && n >= PANEL_MIN && n <= panel_max_rows(); }
constexpr int64_t PANEL_MAX_LD = (int64_t)1 << 20;  // 32-bit element offsets inside the kernel

// scratch of potrf_inverse_f64 (doubles): enough for either way the recursion may go (the single-launch panel can be
// switched off for good between the sizing and the use -- potrf_panel_disable)
static int64_t inverse_work_mode(int64_t n, bool panel_on) {
  if (n <= 2 * LEAF) return 0;
  if (panel_on && n >= PANEL_MIN && n <= panel_max_rows()) {
    const int64_t ldp = ((n + PANEL_NB - 1) / PANEL_NB) * PANEL_NB, n16 = (n + 15) / 16 * 16;
    return ldp * (ldp + PANEL_NB) + (n16 != n ? 2 * n16 * n16 : 0);  // + the zero-padded copies of a ragged size
  }
  const int64_t n1 = split_point(n), n2 = n - n1;
  const int64_t a = inverse_work_mode(n1, panel_on), b = 2 * n1 * n2 + inverse_work_mode(n2, panel_on);
  return a > b ? a : b;
}
int64_t potrf_inverse_work_doubles(int64_t n) {
  const int64_t a = inverse_work_mode(n, true), b = inverse_work_mode(n, false);
  return a > b ? a : b;
}
// scratch region of the panel-wise factorisations: every diagonal block they hand to potrf_inverse_rec has B rows
// except the last, which has n mod B -- and a ragged last block (not a multiple of 16) needs MORE than a full one (its
// zero-padded copies: 2 n16^2 on top of the panel kernel's own ldp (ldp + 128))
static int64_t panel_scratch_doubles(int64_t n, int64_t B) {
  int64_t w = B * B;
  if (potrf_inverse_work_doubles(B) > w) w = potrf_inverse_work_doubles(B);
  if (n % B != 0 && potrf_inverse_work_doubles(n % B) > w) w = potrf_inverse_work_doubles(n % B);
  return w;
}

// padded16: rows and columns n .. 16 ceil(n / 16) - 1 of A and W exist in memory and are zero (they stay zero)
static int32_t potrf_inverse_rec(double* A, int64_t lda, double* W, int64_t ldw, int64_t n, int64_t row_offset,
                                 int* d_info, double* scr, bool keep_L, hipStream_t s, bool padded16) {
  if (n <= 0) return GSS_OK;
  if (use_panel(n) && n % 16 != 0 && !padded16) {
    // whole tiles through zero-padded copies at the head of the scratch (three copies and a fill: ~25 us against the
    // 0.6 ms the launch chain would add)
    const int64_t n16 = (n + 15) / 16 * 16;
    double* Ap = scr;
    double* Wp = scr + n16 * n16;
    GSS_TRY(dev_zero_bytes(Ap, sizeof(double) * (size_t)(2 * n16 * n16), s));
    GSS_TRY(copy_block(A, lda, n, n, Ap, n16, s));
    GSS_TRY(potrf_inverse_rec(Ap, n16, Wp, n16, n, row_offset, d_info, scr + 2 * n16 * n16, keep_L, s, true));
    GSS_TRY(copy_block(Ap, n16, n, n, A, lda, s));
    GSS_TRY(copy_block(Wp, n16, n, n, W, ldw, s));
    return GSS_OK;
  }
  if (use_panel(n) && lda < PANEL_MAX_LD && ldw < PANEL_MAX_LD && (n % 16 == 0 || padded16)) {
    static uint64_t attr = 0;
    if (first_on_this_device(attr)) {
      GSS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(potrf_inv_panel_kernel<PANEL_WAVES>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)L128_LDS_BYTES));
      }
    // test hook: GSS_PANEL_FAIL=1 makes the first such launch of the process report what a launch whose workgroups did
    // not all arrive reports (*d_info = -1), so that the callers' recovery (potrf_panel_disable, retry) can be exercised
    static std::atomic<bool> fail_once{std::getenv("GSS_PANEL_FAIL") != nullptr};
    if (fail_once.exchange(false)) {
      GSS_HIP(hipMemsetAsync(d_info, 0xFF, sizeof(int), s));
      return GSS_OK;
    }
    unsigned* bar = panel_barrier_words(s);
    GSS_REQUIRE(bar != nullptr, "potrf_inverse: no memory for the barrier words");
    static const bool times = std::getenv("GSS_PANEL_TIMES") != nullptr;
    if (times) GSS_HIP(hipMemsetAsync(bar + 2, 1, 1, s));  // (the kernel leaves the barrier words at zero)
    int n16 = (int)((n + 15) / 16 * 16), n32 = (int)n, ro = (int)row_offset;
    // An ordinary launch: 64 workgroups of a 256-CU device are resident together whenever fewer than four such
    // kernels run at once (one workgroup per CU: 110 KB of LDS); a cooperative launch would guarantee it but goes
    // through a queue of its own (12 us of cross-queue hand-over on either side of every launch).  Should the
    // workgroups ever not all arrive, the barrier gives up after a bounded spin and *d_info becomes -1.
    hipLaunchKernelGGL(potrf_inv_panel_kernel<PANEL_WAVES>, dim3((unsigned)panel_workgroups()), dim3(64 * PANEL_WAVES),
                       L128_LDS_BYTES, s, A, n16, n32, lda, W, ldw, scr, ro, d_info, bar);
    GSS_HIP(hipGetLastError());
    if (times) {  // debugging aid: phase boundaries as seen by workgroup 0, in microseconds from the start
      long long st[64] = {0};
      GSS_HIP(hipStreamSynchronize(s));
      GSS_HIP(hipMemcpy(st, bar + 16, sizeof(st), hipMemcpyDeviceToHost));
      const int nb = (int)((n + PANEL_NB - 1) / PANEL_NB), ns = 6 * nb - 2;
      fprintf(stderr, "panel n=%d:", n32);
      for (int i = 1; i < ns && i < 64; ++i) fprintf(stderr, " %.1f", (double)(st[i] - st[0]) * 0.01);
      fprintf(stderr, "\n");
      long long lf[24] = {0};
      GSS_HIP(hipMemcpy(lf, reinterpret_cast<long long*>(bar + 16) + 128, sizeof(lf), hipMemcpyDeviceToHost));
      fprintf(stderr, "leaf of step 1 (block loaded | first tile factored | next tile factored, per step ... | end):");
      for (int i = 1; i < 12; ++i) fprintf(stderr, " %.1f", (double)(lf[i] - lf[0]) * 0.01);
      fprintf(stderr, "\n");
    }
    return GSS_OK;
  }
  if (n <= LEAF) {
    hipLaunchKernelGGL(potrf_inv_leaf_tile_kernel, dim3(1), dim3(64), 0, s, A, (int)n, lda, (int)row_offset, d_info, W,
                       ldw, 0);
    GSS_HIP(hipGetLastError());
    return GSS_OK;
  }
  if (n <= 2 * LEAF) {
    static uint64_t attr = 0;
    if (first_on_this_device(attr)) {
      GSS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(potrf_inv_leaf128_kernel<8>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)L128_LDS_BYTES));
      }
    hipLaunchKernelGGL(potrf_inv_leaf128_kernel<8>, dim3(1), dim3(512), L128_LDS_BYTES, s, A, (int)n, lda, (int)row_offset,
                       d_info, W, ldw);
    GSS_HIP(hipGetLastError());
    return GSS_OK;
  }
  const int64_t n1 = split_point(n), n2 = n - n1;
  double* A21 = A + n1;
  double* A22 = A + n1 + n1 * lda;
  double* W21 = W + n1;
  double* W22 = W + n1 + n1 * ldw;
  GSS_TRY(potrf_inverse_rec(A, lda, W, ldw, n1, row_offset, d_info, scr, keep_L, s, false));
  double* L21 = scr;            // n2 x n1, column-major, ld n2
  double* T2 = scr + n1 * n2;   // n2 x n1
  double* rest = T2 + n1 * n2;
  GSS_TRY(gemm_f64(n2, n1, n1, 1.0, A21, 1, lda, W, ldw, 1, 0.0, L21, 1, n2, false, s, GEMM_TRI_B_UPPER));
  if (keep_L) GSS_TRY(copy_block(L21, n2, n2, n1, A21, lda, s));
  GSS_TRY(gemm_f64(n2, n2, n1, -1.0, L21, 1, n2, L21, n2, 1, 1.0, A22, 1, lda, true, s));
  GSS_TRY(potrf_inverse_rec(A22, lda, W22, ldw, n2, row_offset + n1, d_info, rest, keep_L, s, padded16));
  GSS_TRY(gemm_f64(n2, n1, n1, 1.0, L21, 1, n2, W, 1, ldw, 0.0, T2, 1, n2, false, s, GEMM_TRI_B_LOWER));
  GSS_TRY(gemm_f64(n2, n1, n2, -1.0, W22, 1, ldw, T2, 1, n2, 0.0, W21, 1, ldw, false, s, GEMM_TRI_A_LOWER));
  return GSS_OK;
}

int32_t potrf_inverse_f64(double* A, int64_t n, int64_t lda, double* W, int64_t ldw, double* scr, int* d_info,
                          bool keep_L, hipStream_t s, bool padded16) {
  GSS_TRY(dev_zero_bytes(d_info, sizeof(int), s));
  const int64_t n16 = (n + 15) / 16 * 16;
  return potrf_inverse_rec(A, lda, W, ldw, n, 0, d_info, scr, keep_L, s, padded16 && lda >= n16 && ldw >= n16);
}

// Right-looking Cholesky by panels of POTRF_PANEL columns for large blocks whose inverse is not wanted (LUGS): the
// diagonal block of a panel is factorised together with its inverse (potrf_inverse_rec), which makes the rows below
// one GEMM (L_ik = A_ik W_kk'), followed by one trailing update per panel.  About 95 launches per panel instead of
// ~370 with the fully recursive potrf_f64.  A holds L in its lower triangle on return.
constexpr int64_t POTRF_PANEL = 1024;

int64_t potrf_blocked_work_doubles(int64_t n) {
  const int64_t B = n < POTRF_PANEL ? n : POTRF_PANEL;
  return B * B + panel_scratch_doubles(n, B) + 2 * n * B;
}


// Look-ahead: the trailing update of panel k is split into (a) the next 1 024 columns, which the next panel needs,
// and (b) everything to the right of them.  (b) runs on a helper stream beside the next panel's single-launch
// factorisation (64 workgroups: most of the device would idle) and its rows-below product; the panel column P is
// double buffered for that.  Order of the updates of one column block: (b) of the earlier panels, then (a) of the
// panel just before it (the caller's stream waits for the helper's event in front of every (a)).
int32_t potrf_blocked_f64(double* A, int64_t n, int64_t lda, int* d_info, double* work, hipStream_t s) {
  GSS_TRY(dev_zero_bytes(d_info, sizeof(int), s));
  const int64_t B = n < POTRF_PANEL ? n : POTRF_PANEL;
  double* Wk = work;           // B x B inverse of the diagonal block
  double* scr = Wk + B * B;    // scratch of the factor-and-inverse step (at least B x B)
  double* Pbuf[2];
  Pbuf[0] = scr + panel_scratch_doubles(n, B);     // (n - k0 - nb) x nb panel column, two of them
  Pbuf[1] = Pbuf[0] + n * B;
  static const bool la_on = [] {
    const char* e = std::getenv("GSS_POTRF_LOOKAHEAD");
    return !(e && e[0] == '0');
  }();
  hipStream_t side = (la_on && n > 2 * B) ? lookahead_stream() : nullptr;
  ScopedEvent ev_p, ev_b;
  if (side) {
    GSS_HIP(ev_p.create());
    GSS_HIP(ev_b.create());
  }
  bool b_pending = false;
  int32_t rc = GSS_OK;
  int ip = 0;
  for (int64_t k0 = 0; k0 < n && rc == GSS_OK; k0 += B, ip ^= 1) {
    const int64_t nb = (n - k0) < B ? (n - k0) : B;
    const int64_t m2 = n - k0 - nb;
    double* Akk = A + k0 + k0 * lda;
    double* P = Pbuf[ip];
    // (Wk's strict upper triangle is zero on entry and nothing ever writes there: zeroed once, and again only for
    // the last panel, whose leading dimension differs)
    if (k0 == 0 || nb != B) rc = dev_zero_bytes(Wk, sizeof(double) * (size_t)(nb * nb), s);
    // a ragged last panel works through padded copies in the scratch: it starts behind the helper's work of the
    // panel before it (which reads the other panel column), whatever the sizes of the regions
    if (rc == GSS_OK && nb != B && b_pending) {
      if (hipStreamWaitEvent(s, ev_b, 0) != hipSuccess) rc = GSS_ERR_HIP;
      b_pending = false;
    }
    if (rc == GSS_OK) rc = potrf_inverse_rec(Akk, lda, Wk, nb, nb, k0, d_info, scr, true, s, false);
    if (rc != GSS_OK || m2 <= 0) continue;
    double* Ap = Akk + nb;                 // rows below the diagonal block
    double* A22 = Akk + nb + nb * lda;
    rc = gemm_f64(m2, nb, nb, 1.0, Ap, 1, lda, Wk, nb, 1, 0.0, P, 1, m2, false, s, GEMM_TRI_B_UPPER);
    if (rc != GSS_OK) continue;
    if (!side) {
      rc = copy_block(P, m2, m2, nb, Ap, lda, s);
      if (rc == GSS_OK) rc = gemm_f64(m2, m2, nb, -1.0, P, 1, m2, P, m2, 1, 1.0, A22, 1, lda, true, s);
      continue;
    }
    const int64_t nbn = m2 < B ? m2 : B, m3 = m2 - nbn;
    if (b_pending && hipStreamWaitEvent(s, ev_b, 0) != hipSuccess) rc = GSS_ERR_HIP;   // (b) of the panel before
    b_pending = false;
    // (a) the next column block, lower tiles; (b) starts behind it, so that (a) -- which the next panel waits for --
    // has the device to itself
    if (rc == GSS_OK) rc = gemm_f64(m2, nbn, nb, -1.0, P, 1, m2, P, m2, 1, 1.0, A22, 1, lda, true, s);
    if (rc == GSS_OK && hipEventRecord(ev_p, s) != hipSuccess) rc = GSS_ERR_HIP;
    if (rc == GSS_OK) {
      // the helper: the factor's columns P -> A (nothing on the caller's stream reads them again), then (b)
      if (hipStreamWaitEvent(side, ev_p, 0) != hipSuccess) rc = GSS_ERR_HIP;
      if (rc == GSS_OK) rc = copy_block(P, m2, m2, nb, Ap, lda, side);
      if (rc == GSS_OK && m3 > 0)
        rc = gemm_f64(m3, m3, nb, -1.0, P + nbn, 1, m2, P + nbn, m2, 1, 1.0, A22 + nbn + nbn * lda, 1, lda, true, side);
      if (rc == GSS_OK && hipEventRecord(ev_b, side) != hipSuccess) rc = GSS_ERR_HIP;
      b_pending = rc == GSS_OK;
    }
  }
  if (side) {
    // whatever happened, the caller's stream comes back behind the helper (the buffers belong to the caller)
    if (hipEventRecord(ev_b, side) == hipSuccess) (void)hipStreamWaitEvent(s, ev_b, 0);
  }
  return rc;
}

// LUGS preprocess, data columns (lu.jl:134-139 as one right-looking factorisation of [C11 . ; C21 C22] over the data
// columns): on return C11 holds L11 (lower triangle), C21 (mb x nd, leading dimension ld21) holds C21 L11^-T -- rows
// past the ns simulation rows ride along, the caller puts z1' there and reads L11^-1 z1 back --, and the lower tiles
// of C22 hold C22 - A21 A21'.  Per panel of POTRF_PANEL data columns: single-launch factor and inverse of the diagonal
// block, the rows below it in C11 and C21, the update of the rest of C11 and of the NEXT column block of C21 on the
// caller's stream; the update of the remaining columns of C21 and the panel's share of the SYRK on C22 (ns^2 x 1 024:
// the bulk) go to the helper stream and run beside the next panel.  No inverse of L11 is formed.
int64_t potrf_joint_work_doubles(int64_t nd, int64_t mb) {
  const int64_t B = nd < POTRF_PANEL ? nd : POTRF_PANEL;
  return B * B + panel_scratch_doubles(nd, B) + 2 * (nd + mb) * B;
}

int32_t potrf_joint_f64(double* C11, int64_t nd, double* C21, int64_t mb, int64_t ld21, double* C22, int64_t ns,
                        int* d_info, double* work, hipStream_t s) {
  GSS_TRY(dev_zero_bytes(d_info, sizeof(int), s));
  const int64_t B = nd < POTRF_PANEL ? nd : POTRF_PANEL;
  double* Wk = work;
  double* scr = Wk + B * B;
  double* Pt[2];
  double* Pb[2];
  Pt[0] = scr + panel_scratch_doubles(nd, B);
  Pt[1] = Pt[0] + nd * B;
  Pb[0] = Pt[1] + nd * B;
  Pb[1] = Pb[0] + mb * B;
  hipStream_t side = lookahead_stream();
  if (!side) side = s;   // no helper stream: the same work in the same order on the caller's stream
  ScopedEvent ev_s, ev_b;
  GSS_HIP(ev_s.create());
  GSS_HIP(ev_b.create());
  bool b_pending = false;
  int32_t rc = GSS_OK;
  int ip = 0;
  for (int64_t k0 = 0; k0 < nd && rc == GSS_OK; k0 += B, ip ^= 1) {
    const int64_t nb = (nd - k0) < B ? (nd - k0) : B;
    const int64_t mt = nd - k0 - nb;              // rows of C11 below the diagonal block
    double* Akk = C11 + k0 + k0 * nd;
    double* Atop = Akk + nb;                      // mt x nb
    double* Abot = C21 + k0 * ld21;               // mb x nb
    double *Ptop = Pt[ip], *Pbot = Pb[ip];
    if (k0 == 0 || nb != B) rc = dev_zero_bytes(Wk, sizeof(double) * (size_t)(nb * nb), s);
    if (rc == GSS_OK) rc = potrf_inverse_rec(Akk, nd, Wk, nb, nb, k0, d_info, scr, true, s, false);
    // rows below: [Ptop; Pbot] = [Atop; Abot] Wk'
    if (rc == GSS_OK && mt > 0) rc = gemm_f64(mt, nb, nb, 1.0, Atop, 1, nd, Wk, nb, 1, 0.0, Ptop, 1, mt, false, s, GEMM_TRI_B_UPPER);
    if (rc == GSS_OK && mt > 0) rc = copy_block(Ptop, mt, mt, nb, Atop, nd, s);
    if (rc == GSS_OK) rc = gemm_f64(mb, nb, nb, 1.0, Abot, 1, ld21, Wk, nb, 1, 0.0, Pbot, 1, mb, false, s, GEMM_TRI_B_UPPER);
    if (rc == GSS_OK) rc = copy_block(Pbot, mb, mb, nb, Abot, ld21, s);
    if (rc != GSS_OK) break;
    // the helper's work of the panel before touched the columns of C21 that are updated next
    if (b_pending && hipStreamWaitEvent(s, ev_b, 0) != hipSuccess) rc = GSS_ERR_HIP;
    b_pending = false;
    const int64_t nbn = mt < B ? mt : B;          // the next panel's columns
    if (rc == GSS_OK && mt > 0) {
      // rest of C11 (lower tiles) and the next column block of C21: what the next panel reads
      rc = gemm_f64(mt, mt, nb, -1.0, Ptop, 1, mt, Ptop, mt, 1, 1.0, Akk + nb + nb * nd, 1, nd, true, s);
      if (rc == GSS_OK)
        rc = gemm_f64(mb, nbn, nb, -1.0, Pbot, 1, mb, Ptop, mt, 1, 1.0, Abot + nb * ld21, 1, ld21, false, s);
    }
    if (rc == GSS_OK && hipEventRecord(ev_s, s) != hipSuccess) rc = GSS_ERR_HIP;
    if (rc == GSS_OK && hipStreamWaitEvent(side, ev_s, 0) != hipSuccess) rc = GSS_ERR_HIP;
    if (rc == GSS_OK && mt > nbn)                 // the columns of C21 behind the next block
      rc = gemm_f64(mb, mt - nbn, nb, -1.0, Pbot, 1, mb, Ptop + nbn, mt, 1, 1.0, Abot + (nb + nbn) * ld21, 1, ld21, false, side);
    if (rc == GSS_OK && ns > 0)                   // this panel's share of C22 -= A21 A21' (lower tiles)
      rc = gemm_f64(ns, ns, nb, -1.0, Pbot, 1, mb, Pbot, mb, 1, 1.0, C22, 1, ns, true, side);
    if (rc == GSS_OK && hipEventRecord(ev_b, side) != hipSuccess) rc = GSS_ERR_HIP;
    b_pending = rc == GSS_OK;
  }
  if (hipEventRecord(ev_b, side) == hipSuccess) (void)hipStreamWaitEvent(s, ev_b, 0);
  return rc;
}

}  // namespace gss

using namespace gss;

extern "C" {

int32_t gss_dev_potrf(double* a, int64_t n, int64_t lda, void* stream) {
  GSS_ENTRY();
  GSS_REQUIRE(a != nullptr && n >= 0 && lda >= n, "gss_dev_potrf: bad arguments");
  hipStream_t s = to_stream(stream);
  DevBuf info, dinv;
  GSS_TRY(info.alloc(sizeof(int)));
  GSS_TRY(dinv.alloc(sizeof(double) * (size_t)potrf_dinv_doubles(n)));
  GSS_TRY(potrf_f64(a, n, lda, info.as<int>(), dinv.as<double>(), s));
  int h = 0;
  GSS_HIP(hipMemcpyAsync(&h, info.p, sizeof(int), hipMemcpyDeviceToHost, s));
  GSS_HIP(hipStreamSynchronize(s));
  if (h != 0) {
    set_error("Cholesky failed: non-positive pivot at row %d", h - 1);
    return GSS_ERR_NOT_POSDEF;
  }
  return GSS_OK;
}

int32_t gss_dev_potrf_inverse(double* a, int64_t n, int64_t lda, double* w, int64_t ldw, void* stream) {
  GSS_ENTRY();
  GSS_REQUIRE(a != nullptr && w != nullptr && n >= 0 && lda >= n && ldw >= n, "gss_dev_potrf_inverse: bad arguments");
  if (n == 0) return GSS_OK;
  hipStream_t s = to_stream(stream);
  DevBuf info, scr;
  GSS_TRY(info.alloc(sizeof(int)));
  const int64_t ws = potrf_inverse_work_doubles(n) > n * n ? potrf_inverse_work_doubles(n) : n * n;
  GSS_TRY(scr.alloc(sizeof(double) * (size_t)ws));
  GSS_HIP(hipMemset2DAsync(w, sizeof(double) * ldw, 0, sizeof(double) * n, n, s));
  GSS_TRY(potrf_inverse_f64(a, n, lda, w, ldw, scr.as<double>(), info.as<int>(), true, s));
  int h = 0;
  GSS_HIP(hipMemcpyAsync(&h, info.p, sizeof(int), hipMemcpyDeviceToHost, s));
  GSS_HIP(hipStreamSynchronize(s));
  if (h < 0) {
    set_error("factor and inverse: the kernel gave up waiting at a grid barrier");
    return GSS_ERR_HIP;
  }
  if (h != 0) {
    set_error("Cholesky failed: non-positive pivot at row %d", h - 1);
    return GSS_ERR_NOT_POSDEF;
  }
  return GSS_OK;
}

int32_t gss_dev_trtri(const double* l, int64_t n, int64_t ldl, double* w, int64_t ldw, void* stream) {
  GSS_ENTRY();
  GSS_REQUIRE(l != nullptr && w != nullptr && n >= 0 && ldl >= n && ldw >= n, "gss_dev_trtri: bad arguments");
  hipStream_t s = to_stream(stream);
  DevBuf T;
  const int64_t h = n / 2 + LEAF;
  GSS_TRY(T.alloc(sizeof(double) * (size_t)(h * h)));
  GSS_HIP(hipMemset2DAsync(w, sizeof(double) * ldw, 0, sizeof(double) * n, n, s));
  GSS_TRY(trtri_f64(l, n, ldl, w, ldw, T.as<double>(), nullptr, s));
  GSS_HIP(hipStreamSynchronize(s));
  return GSS_OK;
}

int32_t gss_dev_gemm(int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t sa_i, int64_t sa_k,
                     const double* B, int64_t sb_k, int64_t sb_j, double beta, double* D, int64_t sd_i,
                     int64_t sd_j, int32_t lower_only, void* stream) {
  GSS_ENTRY();
  GSS_REQUIRE(A && B && D && M >= 0 && N >= 0 && K >= 0, "gss_dev_gemm: bad arguments");
  // bit 0: lower tiles only; bits 1..3: zero structure of an operand (GEMM_TRI_* << 1), test support
  return gemm_f64(M, N, K, alpha, A, sa_i, sa_k, B, sb_k, sb_j, beta, D, sd_i, sd_j, (lower_only & 1) != 0,
                  to_stream(stream), (lower_only >> 1) & 7);
}

}  // extern "C"
