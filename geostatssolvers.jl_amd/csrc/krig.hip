// KrigingSolver, global neighbourhood: fit once, predict every domain point.
// Replaces GeoStatsModels.fit + the predictprob loop of exactsolve
// (/root/reference/src/estimation/krig.jl:166-186) and preprocess (krig.jl:76-128).
//
// Mathematics.  The kriging system K = [C F; F' 0] (covariance form, size
// N1 = n + nc) has the block factorisation K = M D M' with M = [L 0; B L_S] lower triangular,
// L L' = C, B = (L^-1 F)', L_S L_S' = B B', D = diag(I_n, -I_nc).  With W' = M^-1 precomputed,
//     y       = W' [c0; f0]                    (one triangular GEMM over all points of a chunk)
//     sigma^2 = max(0, sill - sum_{i<n} y_i^2 + sum_{i>=n} y_i^2)
//     mu      = wd . [c0; f0] ,  wd = K^-1 [z; 0] = W'' D W' [z; 0]     (dual form)
// so that per point the work is exactly the (n+nc)^2 flop of one triangular solve plus n covariance
// evaluations, and nothing of size n x m ever returns to the host.
//
// Kernels:
//   krig_rhs_kernel       K1+K2: assembles R = [c0; f0] for a chunk of points (coalesced HBM stores)
//                         and accumulates the dual-form mean on the fly
//   krig_quadform_kernel  K3: FP64 MFMA triangular GEMM W' * R fused with the signed column norms
#include "gss_internal.h"

#include <mutex>
#include "mfma_f64.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

namespace gss {

constexpr int MAX_NC = 64;
constexpr int NSEG = 4;  // row segments of the RHS assembly (mean partials are summed in fixed order)

struct DriftSpec {
  int variant;
  int nc;
  int dim;
  signed char e[MAX_NC][3];
  double center[3];
  double inv_scale[3];
};

// out[c * ld + p] = f_c(x_p) for c < nc, zero rows up to nrows
template <int DIM>
__global__ __launch_bounds__(256) void drift_rows_kernel(DriftSpec ds, const double* __restrict__ x,
                                                         const double* __restrict__ drift_vals, int64_t npts,
                                                         int64_t ncols_pad, double* __restrict__ out, int64_t ld,
                                                         int nrows) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= ncols_pad) return;
  const int64_t pc = p < npts ? p : npts - 1;
  double xs[DIM];
#pragma unroll
  for (int k = 0; k < DIM; ++k) xs[k] = (x[pc * DIM + k] - ds.center[k]) * ds.inv_scale[k];
  for (int c = 0; c < nrows; ++c) {
    double f = 0.0;
    if (c < ds.nc) {
      if (ds.variant == GSS_KRIG_ORDINARY) {
        f = 1.0;
      } else if (ds.variant == GSS_KRIG_UNIVERSAL) {
        f = 1.0;
#pragma unroll
        for (int k = 0; k < DIM; ++k)
          for (int q = 0; q < ds.e[c][k]; ++q) f *= xs[k];
      } else {
        f = drift_vals[pc * ds.nc + c];
      }
    }
    out[(int64_t)c * ld + p] = f;
  }
}

// C(d2) for a single-structure model fixed at compile time, without control flow: the shape is evaluated on
// max(d2, 1e-300) for every lane and the zero-lag value (the sill, nugget included) is selected afterwards.  For
// d2 > 0 the operations are those of cov_pair, so values agree with the fit's data-data matrix bit for bit.
template <int KIND>
__device__ __forceinline__ double cov_d2_select(const VgDev& vg, double d2) {
  const double val = vg.cs * vg_shape(KIND, fmax(d2, 1e-300), vg.inv_range, vg.mscale, vg.pw);
  return d2 <= 0.0 ? vg.sill : val;
}

// R[j * ldr + p] = C(x_j, x0_p) for the j rows of segment blockIdx.y
// KIND >= 0: single-structure model fixed at compile time -- the same arithmetic as cov_pair (which the fit uses, so
// a domain point on a datum reproduces that datum's column of C bit for bit) with the model switch folded away; the
// common models then carry no out-of-line call (the Bessel routines of the rarer Matern orders would otherwise set the
// register budget of every launch).  KIND < 0: any model, nested or not.
template <int DIM, int KIND>
__global__ __launch_bounds__(256) void krig_rhs_kernel(VgDev vg, const double* __restrict__ xd, int n,
                                                       const double* __restrict__ x0, int64_t m_valid,
                                                       double* __restrict__ R, int64_t ldr, int seg_len,
                                                       int nblk) {
  // units = (point block, row segment), walked with a grid stride
  for (int unit = blockIdx.x; unit < nblk * NSEG; unit += gridDim.x) {
    const int seg = unit % NSEG;
    const int64_t p = (int64_t)(unit / NSEG) * 256 + threadIdx.x;
    const int64_t pc = p < m_valid ? p : m_valid - 1;
    double c[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) c[k] = x0[pc * DIM + k];
    const int j0 = seg * seg_len;
    const int j1 = j0 + seg_len < n ? j0 + seg_len : n;
    double* rp = R + (int64_t)j0 * ldr + p;
#pragma unroll 4
    for (int j = j0; j < j1; ++j) {
      double x[DIM];
#pragma unroll
      for (int k = 0; k < DIM; ++k) x[k] = xd[j * DIM + k];
      if (KIND < 0) {
        *rp = cov_pair<DIM>(vg, x, c);
      } else {
        const double d2 = sqdist_nofma<DIM>(x, c, vg.ir, vg.aniso != 0);
        *rp = d2 <= 0.0 ? vg.sill : vg.cs * vg_shape(KIND < 0 ? 0 : KIND, d2, vg.inv_range, vg.mscale, vg.pw);
      }
      rp += ldr;
    }
  }
}

// The same assembly with two adjacent points per thread: every store instruction writes 16 B per lane (1 KiB per
// wave and row) instead of 8 -- the store path of this device takes 16-B-per-lane streams at 6.0-6.2 TB/s
// (tools/probe_hbm.hip) against 4.8 TB/s for the 8-B form of this kernel.  ncols = padded point count (even).
template <int DIM, int KIND>
__global__ __launch_bounds__(256) void krig_rhs2_kernel(VgDev vg, const double* __restrict__ xd, int n,
                                                        const double* __restrict__ x0, int64_t m_valid,
                                                        double* __restrict__ R, int64_t ldr, int seg_len, int nblk,
                                                        int64_t ncols) {
  for (int unit = blockIdx.x; unit < nblk * NSEG; unit += gridDim.x) {
    const int seg = unit % NSEG;
    const int64_t p = (int64_t)(unit / NSEG) * 512 + 2 * threadIdx.x;
    if (p >= ncols) continue;
    const int64_t pa = p < m_valid ? p : m_valid - 1, pb = p + 1 < m_valid ? p + 1 : m_valid - 1;
    double ca[DIM], cb[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
      ca[k] = x0[pa * DIM + k];
      cb[k] = x0[pb * DIM + k];
    }
    const int j0 = seg * seg_len;
    const int j1 = j0 + seg_len < n ? j0 + seg_len : n;
    double2* rp = reinterpret_cast<double2*>(R + (int64_t)j0 * ldr + p);
    const int64_t ld2 = ldr >> 1;
#pragma unroll 4
    for (int j = j0; j < j1; ++j) {
      double x[DIM];
#pragma unroll
      for (int k = 0; k < DIM; ++k) x[k] = xd[j * DIM + k];
      double2 v;
      if (KIND < 0) {
        v.x = cov_pair<DIM>(vg, x, ca);
        v.y = cov_pair<DIM>(vg, x, cb);
      } else {
        // isotropic models carry ir = 1 (make_vgdev): multiplying by it is exact, and cheaper than selecting per
        // component on the run-time flag
        const double da = sqdist_nofma<DIM>(x, ca, vg.ir, true);
        const double db = sqdist_nofma<DIM>(x, cb, vg.ir, true);
        v.x = cov_d2_select < KIND < 0 ? 0 : KIND > (vg, da);
        v.y = cov_d2_select < KIND < 0 ? 0 : KIND > (vg, db);
      }
      *rp = v;
      rp += ld2;
    }
  }
}

template <int DIM>
static void launch_krig_rhs(dim3 grid, hipStream_t s, const VgDev& vg, const double* xd, int n, const double* x0,
                            int64_t m_valid, double* R, int64_t ldr, int seg_len, int nblk) {
  if ((ldr & 1) == 0) {   // two adjacent points per thread, 16-B stores (every workspace the library makes is even)
    const int64_t ncols = (int64_t)nblk * 256;
    const int nblk2 = (int)((ncols + 511) / 512);
    const dim3 g2((unsigned)(nblk2 * NSEG));
#define GSS_K1W_LAUNCH(KIND)                                                                                         \
  hipLaunchKernelGGL((krig_rhs2_kernel<DIM, KIND>), g2, dim3(256), 0, s, vg, xd, n, x0, m_valid, R, ldr, seg_len,    \
                     nblk2, ncols)
    switch (vg.nextra == 0 ? vg.kind : -1) {
      case GSS_VG_GAUSSIAN: GSS_K1W_LAUNCH(GSS_VG_GAUSSIAN); break;
      case GSS_VG_EXPONENTIAL: GSS_K1W_LAUNCH(GSS_VG_EXPONENTIAL); break;
      case GSS_VG_SPHERICAL: GSS_K1W_LAUNCH(GSS_VG_SPHERICAL); break;
      case VG_MATERN12: GSS_K1W_LAUNCH(VG_MATERN12); break;
      case VG_MATERN32: GSS_K1W_LAUNCH(VG_MATERN32); break;
      case VG_MATERN52: GSS_K1W_LAUNCH(VG_MATERN52); break;
      default: GSS_K1W_LAUNCH(-1); break;
    }
#undef GSS_K1W_LAUNCH
    return;
  }
#define GSS_K1_LAUNCH(KIND)                                                                                       \
  hipLaunchKernelGGL((krig_rhs_kernel<DIM, KIND>), grid, dim3(256), 0, s, vg, xd, n, x0, m_valid, R, ldr, seg_len, \
                     nblk)
  switch (vg.nextra == 0 ? vg.kind : -1) {
    case GSS_VG_GAUSSIAN: GSS_K1_LAUNCH(GSS_VG_GAUSSIAN); break;
    case GSS_VG_EXPONENTIAL: GSS_K1_LAUNCH(GSS_VG_EXPONENTIAL); break;
    case GSS_VG_SPHERICAL: GSS_K1_LAUNCH(GSS_VG_SPHERICAL); break;
    case VG_MATERN12: GSS_K1_LAUNCH(VG_MATERN12); break;
    case VG_MATERN32: GSS_K1_LAUNCH(VG_MATERN32); break;
    case VG_MATERN52: GSS_K1_LAUNCH(VG_MATERN52); break;
    default: GSS_K1_LAUNCH(-1); break;
  }
#undef GSS_K1_LAUNCH
}

// Means only, for up to BATCH_NB data vectors at once (conditional simulation: fft.jl:125,187 krige the data and every
// unconditional realisation at the same locations and use the means only):  out(b, p) = sum_k WD(k, b) R(k, p)  without
// ever writing R.  Thread = domain point; the data index is wave-uniform, so x_j and the BATCH_NB dual weights of row j
// come through the scalar cache and enter the FMAs as scalar operands; drift rows are added the same way.
constexpr int BATCH_NB = 16;

template <int DIM, int KIND>
__global__ __launch_bounds__(256) void krig_batch_mean_kernel(VgDev vg, DriftSpec ds, const double* __restrict__ xd, int n,
                                                              const double* __restrict__ x0, int64_t m,
                                                              const double* __restrict__ WDt, int nb, double add,
                                                              double* __restrict__ out, int64_t ldo) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = p < m;
  const int64_t pc = live ? p : m - 1;
  double c[DIM];
#pragma unroll
  for (int k = 0; k < DIM; ++k) c[k] = x0[pc * DIM + k];
  double acc[BATCH_NB];
#pragma unroll
  for (int b = 0; b < BATCH_NB; ++b) acc[b] = 0.0;
#pragma unroll 2
  for (int j = 0; j < n; ++j) {
    double x[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) x[k] = xd[j * DIM + k];
    double cv;
    if (KIND < 0) {
      cv = cov_pair<DIM>(vg, x, c);
    } else {
      const double d2 = sqdist_nofma<DIM>(x, c, vg.ir, true);   // ir = 1 for isotropic models: exact
      cv = cov_d2_select < KIND < 0 ? 0 : KIND > (vg, d2);
    }
#pragma unroll
    for (int b = 0; b < BATCH_NB; ++b) acc[b] = fma(cv, WDt[(int64_t)j * BATCH_NB + b], acc[b]);
  }
  if (ds.nc > 0) {
    double xs[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) xs[k] = (c[k] - ds.center[k]) * ds.inv_scale[k];
    for (int r = 0; r < ds.nc; ++r) {
      double f = 1.0;
      if (ds.variant == GSS_KRIG_UNIVERSAL) {
#pragma unroll
        for (int k = 0; k < DIM; ++k)
          for (int q = 0; q < ds.e[r][k]; ++q) f *= xs[k];
      }
#pragma unroll
      for (int b = 0; b < BATCH_NB; ++b) acc[b] = fma(f, WDt[(int64_t)(n + r) * BATCH_NB + b], acc[b]);
    }
  }
  if (!live) return;
#pragma unroll
  for (int b = 0; b < BATCH_NB; ++b)
    if (b < nb) out[(int64_t)b * ldo + p] = acc[b] + add;
}

// WDt[k][b] = WD(k, b) for b < nb, 0 up to BATCH_NB: the weights of one row side by side for the scalar loads
__global__ __launch_bounds__(256) void wd_rows_kernel(const double* __restrict__ WD, int64_t ldw, int N1, int nb,
                                                      double* __restrict__ WDt) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= N1) return;
#pragma unroll
  for (int b = 0; b < BATCH_NB; ++b) WDt[(int64_t)k * BATCH_NB + b] = b < nb ? WD[k + (int64_t)b * ldw] : 0.0;
}

template <int DIM>
static void launch_krig_batch_mean(hipStream_t s, const VgDev& vg, const DriftSpec& ds, const double* xd, int n,
                                   const double* x0, int64_t m, const double* WDt, int nb, double add, double* out,
                                   int64_t ldo) {
  const dim3 grid((unsigned)((m + 255) / 256));
#define GSS_BM_LAUNCH(KIND)                                                                                          \
  hipLaunchKernelGGL((krig_batch_mean_kernel<DIM, KIND>), grid, dim3(256), 0, s, vg, ds, xd, n, x0, m, WDt, nb, add, \
                     out, ldo)
  switch (vg.nextra == 0 ? vg.kind : -1) {
    case GSS_VG_GAUSSIAN: GSS_BM_LAUNCH(GSS_VG_GAUSSIAN); break;
    case GSS_VG_EXPONENTIAL: GSS_BM_LAUNCH(GSS_VG_EXPONENTIAL); break;
    case GSS_VG_SPHERICAL: GSS_BM_LAUNCH(GSS_VG_SPHERICAL); break;
    case VG_MATERN32: GSS_BM_LAUNCH(VG_MATERN32); break;
    case VG_MATERN52: GSS_BM_LAUNCH(VG_MATERN52); break;
    default: GSS_BM_LAUNCH(-1); break;
  }
#undef GSS_BM_LAUNCH
}

// K3.  One workgroup owns a strip of BN = 128 points and walks the row blocks I of W' (lower
// triangular, column-major, zero padded to ldw x N1pad).  For row block I only k < (I+1)*BM
// contributes.  Global -> register -> LDS staging is double buffered: one barrier per BK stage.
constexpr size_t QUADFORM_LDS_BYTES = sizeof(double) * 4 * TILE_LDS;

// SPLIT: one workgroup per (strip, row block I) instead of one per strip.  The row block is the slow block index,
// heaviest first, so that the workgroups in flight at any time cost the same (see the kernel); a strip's units keep
// one XCD.  Partial column sums go to qpart[I][p]; krig_finish_kernel adds them in fixed order.
template <bool SPLIT>
__global__ __launch_bounds__(256, 2) void krig_quadform_kernel(
    const double* __restrict__ W, int64_t ldw, int N1pad, int n, int N1, const double* __restrict__ R,
    int64_t ldr, double sill, double mean0, int64_t m_valid,
    double* __restrict__ mean_out, double* __restrict__ var_out, uint8_t* __restrict__ status_out,
    double* __restrict__ qpart, int strip0, int nstrips) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* As = smem;                 // [2][TILE_LDS]
  double* Bs = smem + 2 * TILE_LDS;  // [2][TILE_LDS]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  const int nI = (N1 + BM) / BM;  // rows 0..N1: the system plus the dual-weight row N1 (the mean)
  int64_t strip = blockIdx.x;  // strips strip0 .. strip0 + nstrips - 1 belong to this launch
  int Ibeg = 0, Iend = nI;
  if (SPLIT) {
    // The row block is the SLOW index (heaviest first) and the strip the fast one: workgroups reach the CUs in launch
    // order, so costs that alternate along the fast index pile the expensive units onto the same CUs (dense_la.hip).
    // The launch pads the strips to a multiple of 8, hence blockIdx.x % 8 == strip % 8: the units of one strip still
    // share an XCD.
    const int per = (int)(gridDim.x / nI);
    strip = blockIdx.x % per;
    if (strip >= nstrips) return;
    Ibeg = nI - 1 - (int)(blockIdx.x / per);
    Iend = Ibeg + 1;
  }
  const int64_t p0 = (strip0 + strip) * BN;

  // staging map: thread moves 2 consecutive doubles of rows kq, kq+4, kq+8, kq+12 of each operand
  const int i2 = (tid & 63) * 2;
  const int kq = tid >> 6;

  double qacc[2] = {0.0, 0.0};
  double macc[2] = {0.0, 0.0};  // row N1 of the product: wd . rhs

  int buf = 0;  // LDS stage holding the operands of the stage about to be multiplied; runs on across row blocks
  for (int I = Ibeg; I < Iend; ++I) {
    const int i0 = I * BM;
    const int kend = (i0 + BM < N1pad) ? i0 + BM : N1pad;
    const int ntile = kend / BK;
    // rows i0 .. N1 of this block carry data (row N1 = dual weights); 16-row tiles beyond them are zero padding
    const int tm_max = (N1 + 1 - i0 + 15) / 16 < 8 ? (N1 + 1 - i0 + 15) / 16 : 8;

    d4 accw[8][2];   // 1 x 4 wave layout: every wave sees all 128 rows (8 tiles) of 32 columns (2 tiles)
#pragma unroll
    for (int tm = 0; tm < 8; ++tm) accw[tm][0] = accw[tm][1] = d4{0.0, 0.0, 0.0, 0.0};

    d2v ra[4], rb[4];
    const double* wp = W + (int64_t)kq * ldw + i0 + i2;
    const double* rp = R + (int64_t)kq * ldr + p0 + i2;
    if (I == Ibeg) {  // later row blocks find their first stage in LDS: the last stage of the block before fetched it
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ra[r] = *reinterpret_cast<const d2v*>(wp + (int64_t)(4 * r) * ldw);
        rb[r] = *reinterpret_cast<const d2v*>(rp + (int64_t)(4 * r) * ldr);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        *reinterpret_cast<d2v*>(As + (kq + 4 * r) * LDS_LD + i2) = ra[r];
        *reinterpret_cast<d2v*>(Bs + (kq + 4 * r) * LDS_LD + i2) = rb[r];
      }
      __syncthreads();
    }

    auto stage = [&](int t, auto guard) {
      constexpr bool GUARD = decltype(guard)::value;
      const int cur = buf;
      const bool last = (t + 1) >= ntile;
      const bool more = !last || (I + 1 < Iend);
      if (more) {
        // next stage of this row block, or stage 0 of the next one (same strip of R from k = 0, rows i0 + BM.. of W)
        const double* wq = last ? wp + BM : wp + (int64_t)(t + 1) * BK * ldw;
        const double* rq = last ? rp : rp + (int64_t)(t + 1) * BK * ldr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          ra[r] = *reinterpret_cast<const d2v*>(wq + (int64_t)(4 * r) * ldw);
          rb[r] = *reinterpret_cast<const d2v*>(rq + (int64_t)(4 * r) * ldr);
        }
      }
      {
        // stages inside the diagonal block (k > i0) only touch row tiles tm >= (k - i0) / 16
        const int tmn = (t * BK - i0) >> 4;
        mma_stage_w14<GUARD>(As + cur * TILE_LDS, Bs + cur * TILE_LDS, accw, wave, lane, tmn > 0 ? tmn : 0, tm_max);
      }
      if (more) {
        double* an = As + (cur ^ 1) * TILE_LDS;
        double* bn = Bs + (cur ^ 1) * TILE_LDS;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          *reinterpret_cast<d2v*>(an + (kq + 4 * r) * LDS_LD + i2) = ra[r];
          *reinterpret_cast<d2v*>(bn + (kq + 4 * r) * LDS_LD + i2) = rb[r];
        }
      }
      __syncthreads();
      buf ^= 1;
    };
    {
      const int tdiag = (i0 / BK + 1) < ntile ? (i0 / BK + 1) : ntile;
      if (tm_max < 8) {  // last row block: its bottom row tiles are padding, every stage is guarded
        for (int t = 0; t < ntile; ++t) stage(t, std::true_type{});
      } else {
        for (int t = 0; t < tdiag; ++t) stage(t, std::false_type{});
        for (int t = tdiag; t < ntile; ++t) stage(t, std::true_type{});
      }
    }

    // signed squares: rows < n count +, constraint rows n..N1-1 count -, padding rows are zero
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      double s = 0.0;
#pragma unroll
      for (int tm = 0; tm < 8; ++tm)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = i0 + tm * 16 + lk + 4 * r;
          const double v = accw[tm][tn][r];
          const double vv = v * v;
          s += (row < n) ? vv : (row < N1 ? -vv : 0.0);
          if (row == N1) macc[tn] += v;
        }
      qacc[tn] += s;
    }
  }

  // reduce over the 4 lane groups (rows) of the wave
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    qacc[tn] += __shfl_xor(qacc[tn], 16);
    qacc[tn] += __shfl_xor(qacc[tn], 32);
    macc[tn] += __shfl_xor(macc[tn], 16);
    macc[tn] += __shfl_xor(macc[tn], 32);
  }
  __syncthreads();
  double* red = smem;            // [2][BN] signed squares
  double* redm = smem + 2 * BN;  // [2][BN] mean row
  if (lk == 0) {
    red[wave * 32 + lr] = qacc[0];
    red[wave * 32 + 16 + lr] = qacc[1];
    red[BN + wave * 32 + lr] = 0.0;
    red[BN + wave * 32 + 16 + lr] = 0.0;
    redm[wave * 32 + lr] = macc[0];
    redm[wave * 32 + 16 + lr] = macc[1];
    redm[BN + wave * 32 + lr] = 0.0;
    redm[BN + wave * 32 + 16 + lr] = 0.0;
  }
  __syncthreads();
  if (SPLIT) {
    if (tid < BN) {
      qpart[(int64_t)Ibeg * ldr + p0 + tid] = red[tid] + red[BN + tid];
      // the unit that owns the row block of row N1 delivers the mean
      if (Ibeg == N1 / BM && p0 + tid < m_valid) mean_out[p0 + tid] = mean0 + (redm[tid] + redm[BN + tid]);
    }
    return;
  }
  if (tid < BN) {
    const int64_t p = p0 + tid;
    if (p < m_valid) {
      const double q = red[tid] + red[BN + tid];
      const double mu = mean0 + (redm[tid] + redm[BN + tid]);
      const double v = sill - q;
      mean_out[p] = mu;
      var_out[p] = v > 0.0 ? v : 0.0;
      if (status_out) status_out[p] = GSS_PT_OK;
    }
  }
}

// variance from the per-row-block partial sums of the SPLIT quadratic form (fixed summation order); the mean is
// written by the unit that owns row N1
__global__ __launch_bounds__(256) void krig_finish_kernel(const double* __restrict__ qpart, int nI, int64_t ldr,
                                                          double sill, int64_t m_valid, double* __restrict__ var_out,
                                                          uint8_t* __restrict__ status_out) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= m_valid) return;
  double q = 0.0;
  for (int I = 0; I < nI; ++I) q += qpart[(int64_t)I * ldr + p];
  const double v = sill - q;
  var_out[p] = v > 0.0 ? v : 0.0;
  if (status_out) status_out[p] = GSS_PT_OK;
}

__global__ void flip_tail_kernel(double* u, int64_t from, int64_t to) {
  const int64_t i = from + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < to) u[i] = -u[i];
}

// rows [from, to) of a column-major matrix with `ncols` columns change sign
__global__ void flip_rows_kernel(double* u, int64_t ld, int64_t from, int64_t to, int64_t ncols) {
  const int64_t i = from + (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t c = blockIdx.y;
  if (i < to && c < ncols) u[i + c * ld] = -u[i + c * ld];
}

__global__ void add_scalar_kernel(double* x, int64_t n, double v) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] += v;
}

__global__ void sub_scalar_kernel(double* z, int64_t n, double mu) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) z[i] -= mu;
}

}  // namespace gss

using namespace gss;

struct gss_krig {
  VgDev vg;
  int variant = GSS_KRIG_ORDINARY;
  double sk_mean = 0.0;
  int degree = 0;
  int ndrift = 0;
  int dim = 0;
  int64_t n = 0;
  int nc = 0;
  int64_t N1 = 0, N1pad = 0, ldw = 0;
  DriftSpec ds;
  DevBuf xdata, z, drift_data;
  DevBuf factor;  // W' (ldw x N1pad, column-major) followed by wd (N1pad)
  // fit in flight: workspace, completion event and status words (joined by krig_fit_wait)
  DevBuf fit_ws;
  hipEvent_t fit_done = nullptr;
  bool fit_pending = false;
  int* fit_info = nullptr;
  hipStream_t fit_stream = nullptr;  // stream of the fit in flight (a retry goes back on it)
  // GSS_KRIG_ASYNC_FIT: the fit runs on the library's fit stream beside whatever the caller queues next (K1 of the
  // first prediction); its two status words travel to pinned host memory on that stream, in front of fit_done
  bool fit_async = false;
  int* fit_info_host = nullptr;
  ~gss_krig() {
    if (fit_pending && fit_done) (void)hipEventSynchronize(fit_done);
    if (fit_done) (void)hipEventDestroy(fit_done);
    if (fit_info_host) (void)hipHostFree(fit_info_host);
  }
  bool factored = false;
  // block support (gss_krig_set_block_support): right-hand sides regularised over a cell of size `cell` sampled at
  // the centres of nsub^dim sub-cells; c_vv = mean covariance between two samples of the cell (replaces the sill in
  // the variance).  nsub = 0: point support.
  int block_nsub = 0;
  double block_cell[3] = {0.0, 0.0, 0.0};
  double block_cvv = 0.0;
  double* Wp() const { return factor.as<double>(); }
  double* wd() const { return factor.as<double>() + ldw * N1pad; }
};

namespace gss {
int32_t krig_local_dev(const VgDev& vg, int variant, int nc, int dim, const signed char* exps, double inv_scale,
                       double sk_mean, const double* xdata, const double* z, const double* drift_data, int64_t n,
                       const double* x0, const double* drift_dom, int64_t m, int k, int minneighbors, double radius,
                       const double* inv_radii_host, double* mean, double* var, uint8_t* status, int* idx_out,
                       int* count_out, hipStream_t s, int metric, HostPipe* pipe = nullptr, int block_nsub = 0,
                       const double* block_cell = nullptr, double block_cvv = 0.0);
}


// Block support (krig.jl:180 passes the cell geometry to predictprob; [RECALL] its covariances are averages over sample
// points inside the cell): row j of R, column p = mean over the nsub^DIM sub-cell centres s of C(x_j, c_p + s).  Thread
// = domain point, the datum index is wave-uniform.  An option, not the tuned path: nsub^DIM covariances per entry.
template <int DIM>
__global__ __launch_bounds__(256) void krig_rhs_block_kernel(VgDev vg, const double* __restrict__ xd, int n,
                                                             const double* __restrict__ x0, int64_t m_valid,
                                                             double* __restrict__ R, int64_t ldr, int nsub,
                                                             double c1, double c2, double c3) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= ldr) return;
  const bool live = p < m_valid;
  double c[DIM];
#pragma unroll
  for (int a = 0; a < DIM; ++a) c[a] = live ? x0[p * DIM + a] : 0.0;
  const double cell[3] = {c1, c2, c3};
  const int ns = DIM == 1 ? nsub : (DIM == 2 ? nsub * nsub : nsub * nsub * nsub);
  const double inv = 1.0 / (double)ns;
  for (int j = 0; j < n; ++j) {
    double xj[DIM];
#pragma unroll
    for (int a = 0; a < DIM; ++a) xj[a] = xd[(int64_t)j * DIM + a];
    double acc = 0.0;
    for (int s = 0; s < ns; ++s) {
      int q = s;
      double sp[DIM];
#pragma unroll
      for (int a = DIM - 1; a >= 0; --a) {      // the same order of samples as the oracle (first axis slowest)
        const int ia = q % nsub;
        q /= nsub;
        sp[a] = c[a] + (((double)ia + 0.5) / (double)nsub - 0.5) * cell[a];
      }
      acc += cov_pair<DIM>(vg, xj, sp);
    }
    R[(int64_t)j * ldr + p] = live ? acc * inv : 0.0;
  }
}

// mean covariance between two samples of a cell: one workgroup, pairs dealt to the threads, fixed-order reduction
template <int DIM>
__global__ __launch_bounds__(256) void block_cvv_kernel(VgDev vg, int nsub, double c1, double c2, double c3,
                                                        double* __restrict__ out) {
  __shared__ double part[256];
  const double cell[3] = {c1, c2, c3};
  const int ns = DIM == 1 ? nsub : (DIM == 2 ? nsub * nsub : nsub * nsub * nsub);
  double acc = 0.0;
  for (int e = threadIdx.x; e < ns * ns; e += 256) {
    double a[DIM], b[DIM];
    int qa = e / ns, qb = e % ns;
#pragma unroll
    for (int d = DIM - 1; d >= 0; --d) {
      a[d] = (((double)(qa % nsub) + 0.5) / (double)nsub - 0.5) * cell[d];
      b[d] = (((double)(qb % nsub) + 0.5) / (double)nsub - 0.5) * cell[d];
      qa /= nsub;
      qb /= nsub;
    }
    acc += cov_pair<DIM>(vg, a, b);
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 256; ++i) t += part[i];
    *out = t / ((double)ns * (double)ns);
  }
}

static void uk_exponents(int dim, int degree, std::vector<signed char>& e) {
  // graded order: total degree 0, 1, ..., `degree`; column order does not change the weights
  e.clear();
  for (int total = 0; total <= degree; ++total) {
    int idx[3] = {0, 0, 0};
    // enumerate tuples in lexicographic order with the first coordinate slowest
    const int lim = total + 1;
    const int cnt = (dim == 1) ? lim : (dim == 2 ? lim * lim : lim * lim * lim);
    for (int t = 0; t < cnt; ++t) {
      int r = t;
      for (int k = dim - 1; k >= 0; --k) {
        idx[k] = r % lim;
        r /= lim;
      }
      int sum = 0;
      for (int k = 0; k < dim; ++k) sum += idx[k];
      if (sum == total) {
        for (int k = 0; k < 3; ++k) e.push_back((signed char)(k < dim ? idx[k] : 0));
      }
    }
  }
}

// Points per chunk of the RHS workspace R (N1pad x mc doubles).  A K3 launch runs mc / 128 workgroups on 512
// resident slots (2 per CU), so small chunks lose up to one slot-round to the tail: the budget defaults to
// min(16 GiB, half of the free HBM) -- 10^6 points at n = 1000 are one launch -- and, when the domain must be
// split, chunks are multiples of 512 * 128 points.  GSS_KRIG_WS_MB overrides the budget.
static int64_t krig_chunk_points(int64_t N1pad, int64_t m) {
  size_t ws = (size_t)16 << 30;
  size_t freeb = 0, totalb = 0;
  if (hipMemGetInfo(&freeb, &totalb) == hipSuccess && freeb / 2 < ws) ws = freeb / 2;
  if (const char* e = std::getenv("GSS_KRIG_WS_MB")) {
    const long v = std::atol(e);
    if (v > 0) ws = (size_t)v << 20;
  }
  int64_t cap = (int64_t)(ws / (sizeof(double) * (size_t)N1pad));
  const int64_t round = 512 * 128;
  cap = cap >= round ? cap / round * round : cap / 256 * 256;
  if (cap < 256) cap = 256;
  return round_up(m, 256) < cap ? round_up(m, 256) : cap;
}

// Prediction workspace (RHS tiles + mean partials), shared by all handles of the process and grown on
// demand: handles are short-lived (one per solve) while the workspace is GiB-sized, and one process
// drives one GPU with non-thread-safe handles, so a process-wide cache is safe.
struct KrigWorkspace {
  DevBuf R, mean_part;
  int64_t doubles = 0, mc = 0, nI = 0;
};
static KrigWorkspace g_ws;

static int32_t krig_workspace(int64_t N1pad, int64_t mc, hipStream_t s, double** R, double** mean_part) {
  const int64_t nI = (N1pad + BM - 1) / BM + 1;
  if (g_ws.doubles < N1pad * mc || g_ws.mc < mc || g_ws.nI < nI) {
    GSS_HIP(hipStreamSynchronize(s));
    g_ws.R.release();
    g_ws.mean_part.release();
    GSS_TRY(g_ws.R.alloc(sizeof(double) * (size_t)(N1pad * mc)));
    GSS_TRY(g_ws.mean_part.alloc(sizeof(double) * (size_t)((NSEG + 1 + nI) * mc)));  // + qpart[nI][mc]
    g_ws.doubles = N1pad * mc;
    g_ws.mc = mc;
    g_ws.nI = nI;
  }
  *R = g_ws.R.as<double>();
  *mean_part = g_ws.mean_part.as<double>();
  return GSS_OK;
}

static int32_t launch_drift_rows(const gss_krig* h, const double* x, const double* drift_vals, int64_t npts,
                                 int64_t ncols_pad, double* out, int64_t ld, int nrows, hipStream_t s) {
  dim3 grid((unsigned)((ncols_pad + 255) / 256));
  switch (h->dim) {
    case 1:
      hipLaunchKernelGGL((drift_rows_kernel<1>), grid, dim3(256), 0, s, h->ds, x, drift_vals, npts, ncols_pad, out,
                         ld, nrows);
      break;
    case 2:
      hipLaunchKernelGGL((drift_rows_kernel<2>), grid, dim3(256), 0, s, h->ds, x, drift_vals, npts, ncols_pad, out,
                         ld, nrows);
      break;
    default:
      hipLaunchKernelGGL((drift_rows_kernel<3>), grid, dim3(256), 0, s, h->ds, x, drift_vals, npts, ncols_pad, out,
                         ld, nrows);
      break;
  }
  GSS_HIP(hipGetLastError());
  return GSS_OK;
}

// small dense helpers for the constraint block (nc <= 64): one thread per output element
__global__ void small_gram_kernel(const double* __restrict__ Bm, int nc, int64_t n, double* __restrict__ S, int lds) {
  // S(c, c2) = Bm[c] . Bm[c2], one wave per entry of the lower triangle
  const int c = blockIdx.x, c2 = blockIdx.y;
  if (c2 > c) return;
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 64) acc = fma(Bm[(int64_t)c * n + i], Bm[(int64_t)c2 * n + i], acc);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  if (threadIdx.x == 0) {
    S[c + c2 * lds] = acc;
    S[c2 + c * lds] = acc;
  }
}

// Wp[(n + i) + j * ldw] = -sum_c WS(i, c) T[c * n + j]   (WS = inv(L_S), lower, ld = ldws)
__global__ void constraint_rows_kernel(const double* __restrict__ WS, int64_t ldws, const double* __restrict__ T,
                                       int nc, int64_t n, double* __restrict__ Wp_rows, int64_t ldw) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  for (int i = 0; i < nc; ++i) {
    double acc = 0.0;
    for (int c = 0; c <= i; ++c) acc = fma(WS[i + c * ldws], T[(int64_t)c * n + j], acc);
    Wp_rows[i + j * ldw] = -acc;
  }
}

__global__ __launch_bounds__(256) void wd_row_kernel(const double* __restrict__ wd, int64_t N1,
                                                     double* __restrict__ row, int64_t ldw) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k < N1) row[k * ldw] = wd[k];
}

// Factorisation of the kriging system on the device (GeoStatsModels.fit, krig.jl:176).  One workspace
// allocation; everything is enqueued on `s` without host synchronisation (krig_fit_wait joins).
struct FitPlan {
  double *M, *T, *Dinv, *Fd, *Bm, *S, *SDinv, *gwork, *zz, *u;
  int *info, *info2;
  int64_t szM, szS;
};

static int32_t krig_fit_plan(gss_krig* h, FitPlan* fp) {
  const int64_t n = h->n, N1pad = h->N1pad, ldw = h->ldw;
  const int64_t szM = ldw * N1pad;
  int64_t szT = n * n > (int64_t)MAX_NC * n ? n * n : (int64_t)MAX_NC * n;  // potrf_inverse_f64 scratch
  if (potrf_inverse_work_doubles(n) > szT) szT = potrf_inverse_work_doubles(n);
  const int64_t szDinv = 64 * 64;  // scratch of the constraint-block leaf
  const int64_t szFd = (int64_t)MAX_NC * n, szBm = (int64_t)MAX_NC * n;
  const int64_t szS = 2 * MAX_NC * MAX_NC + 64 * 64;
  int64_t szGemv = gemv_work_doubles(false, ldw, ldw);
  if (gemv_work_doubles(true, ldw, ldw) > szGemv) szGemv = gemv_work_doubles(true, ldw, ldw);
  const int64_t szVec = 2 * ldw;
  DevBuf& ws = h->fit_ws;
  GSS_TRY(ws.alloc(sizeof(double) * (size_t)(szM + szT + szDinv + szFd + szBm + szS + szGemv + szVec) + 64));
  fp->M = ws.as<double>();
  fp->T = fp->M + szM;
  fp->Dinv = fp->T + szT;
  fp->Fd = fp->Dinv + szDinv;
  fp->Bm = fp->Fd + szFd;
  fp->S = fp->Bm + szBm;            // S (MAX_NC^2), then its leaf inverse scratch (64 x 64)
  fp->SDinv = fp->S + MAX_NC * MAX_NC;
  fp->gwork = fp->S + szS;
  fp->zz = fp->gwork + szGemv;
  fp->u = fp->zz + ldw;
  fp->info = reinterpret_cast<int*>(fp->u + ldw);
  fp->info2 = fp->info + 1;
  fp->szM = szM;
  fp->szS = szS;
  return GSS_OK;
}

// pure enqueue (fills, copies, kernels), about 140 launches: roughly 2 ms of host time
static int32_t krig_fit_enqueue(gss_krig* h, const FitPlan& fp, hipStream_t s) {
  const int64_t n = h->n, N1 = h->N1, ldw = h->ldw;
  const int nc = h->nc;
  double *M = fp.M, *T = fp.T, *Fd = fp.Fd, *Bm = fp.Bm, *S = fp.S, *SDinv = fp.SDinv;
  double *gwork = fp.gwork, *zz = fp.zz, *u = fp.u;
  int *info = fp.info, *info2 = fp.info2;
  const int64_t szM = fp.szM, szS = fp.szS;
  GSS_TRY(dev_zero_bytes(M, sizeof(double) * (size_t)szM, s));
  GSS_TRY(dev_zero_bytes(h->factor.p, h->factor.bytes, s));
  GSS_TRY(dev_zero_bytes(info2, sizeof(int), s));
  GSS_TRY(cov_pairwise_dev(h->vg, h->xdata.as<double>(), n, h->xdata.as<double>(), n, M, ldw, s));
  double* Wp = h->Wp();
  // M and W' are zero outside the n x n blocks and reach past the next multiple of 16 (N1pad, ldw): the padded contract
  GSS_TRY(potrf_inverse_f64(M, n, ldw, Wp, ldw, T, info, false, s, true));

  if (nc > 0) {
    GSS_TRY(dev_zero_bytes(S, sizeof(double) * (size_t)szS, s));
    // Fd (n x nc, column-major): drift functions at the data locations
    GSS_TRY(launch_drift_rows(h, h->xdata.as<double>(), h->drift_data.as<double>(), n, n, Fd, n, nc, s));
    // Bm[c] = W F[:, c]   (row c of B = (L^-1 F)')
    for (int c = 0; c < nc; ++c) GSS_TRY(gemv_f64(false, n, n, Wp, ldw, Fd + (int64_t)c * n, Bm + (int64_t)c * n, gwork, s));
    hipLaunchKernelGGL(small_gram_kernel, dim3(nc, nc), dim3(64), 0, s, Bm, nc, n, S, MAX_NC);
    GSS_HIP(hipGetLastError());
    // W'[n:, n:] = inv(L_S)  (nc <= 64: one leaf)
    GSS_TRY(potrf_inverse_f64(S, nc, MAX_NC, Wp + n + n * ldw, ldw, SDinv, info2, false, s));
    // T[c] = W' Bm[c]  (row c of B W)
    for (int c = 0; c < nc; ++c) GSS_TRY(gemv_f64(true, n, n, Wp, ldw, Bm + (int64_t)c * n, T + (int64_t)c * n, gwork, s));
    // W'[n:, 0:n] = -inv(L_S) * T
    hipLaunchKernelGGL(constraint_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, Wp + n + n * ldw, ldw, T,
                       nc, n, Wp + n, ldw);
    GSS_HIP(hipGetLastError());
  }

  // dual weights wd = W'' D W' [z - mean; 0]
  GSS_TRY(dev_zero_bytes(zz, sizeof(double) * (size_t)(2 * ldw), s));
  GSS_TRY(dev_copy_f64(zz, h->z.as<double>(), n, s));
  if (h->variant == GSS_KRIG_SIMPLE && h->sk_mean != 0.0) {
    hipLaunchKernelGGL(sub_scalar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, zz, n, h->sk_mean);
  }
  GSS_TRY(gemv_f64(false, N1, N1, Wp, ldw, zz, u, gwork, s));
  if (nc > 0) hipLaunchKernelGGL(flip_tail_kernel, dim3(1), dim3(256), 0, s, u, n, N1);
  GSS_TRY(gemv_f64(true, N1, N1, Wp, ldw, u, h->wd(), gwork, s));
  // row N1 of W' <- wd: the mean then comes out of K3's product as row N1 of W' R
  hipLaunchKernelGGL(wd_row_kernel, dim3((unsigned)((N1 + 255) / 256)), dim3(256), 0, s, h->wd(), N1, Wp + N1, ldw);
  GSS_HIP(hipGetLastError());
  return GSS_OK;
}

// high-priority stream of the process for asynchronous fits (fenced by events against the caller's stream)
static hipStream_t krig_fit_stream() { return helper_stream(HELPER_FIT); }

// async: on the fit stream, behind everything `s` holds now; the caller's stream is joined by krig_join_device
static int32_t krig_factorize(gss_krig* h, hipStream_t s, bool async = false) {
  hipStream_t fs = s;
  if (async) {
    fs = krig_fit_stream();
    if (!fs) {
      async = false;
      fs = s;
    }
  }
  if (async) {
    ScopedEvent e0;
    GSS_HIP(e0.create());
    GSS_HIP(hipEventRecord(e0, s));
    GSS_HIP(hipStreamWaitEvent(fs, e0, 0));
    if (!h->fit_info_host) GSS_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->fit_info_host), 2 * sizeof(int), hipHostMallocDefault));
  }
  FitPlan fp;
  GSS_TRY(krig_fit_plan(h, &fp));
  GSS_TRY(krig_fit_enqueue(h, fp, fs));
  h->fit_info = fp.info;
  h->fit_stream = fs;
  h->fit_async = async;
  if (async) GSS_HIP(hipMemcpyAsync(h->fit_info_host, fp.info, 2 * sizeof(int), hipMemcpyDeviceToHost, fs));
  if (!h->fit_done) GSS_HIP(hipEventCreateWithFlags(&h->fit_done, hipEventDisableTiming));
  GSS_HIP(hipEventRecord(h->fit_done, fs));
  h->fit_pending = true;
  h->factored = true;
  return GSS_OK;
}

// device side of the join: what `s` receives next runs behind the fit
static int32_t krig_join_device(gss_krig* h, hipStream_t s) {
  if (h->fit_pending && h->fit_done && h->fit_stream != s) GSS_HIP(hipStreamWaitEvent(s, h->fit_done, 0));
  return GSS_OK;
}

// Joins a pending fit: waits for it, reads the two status words, releases the workspace.
static int32_t krig_fit_wait(gss_krig* h) {
  if (!h->fit_pending) return GSS_OK;
  h->fit_pending = false;
  GSS_HIP(hipEventSynchronize(h->fit_done));
  int hinfo[2] = {0, 0};
  if (h->fit_async) {   // the words were copied on the fit stream in front of the event: no blocking copy (it would
    hinfo[0] = h->fit_info_host[0];   // wait for the caller's stream as well)
    hinfo[1] = h->fit_info_host[1];
  } else {
    GSS_HIP(hipMemcpy(hinfo, h->fit_info, 2 * sizeof(int), hipMemcpyDeviceToHost));
  }
  h->fit_ws.release();
  h->fit_info = nullptr;
  if (hinfo[0] < 0) {
    // the single-launch factorisation gave up at a grid barrier (its workgroups were not resident together):
    // once, on the launch-per-block path, which has no such requirement
    h->factored = false;
    static thread_local bool retrying = false;
    if (!retrying) {
      potrf_panel_disable();
      retrying = true;
      int32_t rc = krig_factorize(h, h->fit_stream, false);
      if (rc == GSS_OK) rc = krig_fit_wait(h);
      retrying = false;
      return rc;
    }
    set_error("kriging fit: the factorisation kernel gave up waiting at a grid barrier");
    return GSS_ERR_HIP;
  }
  if (hinfo[0] != 0) {
    h->factored = false;
    set_error("kriging covariance matrix is not positive definite (pivot %d of %lld); add a nugget or remove "
              "duplicate samples", hinfo[0] - 1, (long long)h->n);
    return GSS_ERR_NOT_POSDEF;
  }
  if (hinfo[1] != 0) {
    h->factored = false;
    set_error("drift functions are linearly dependent on the sample locations (constraint %d)", hinfo[1] - 1);
    return GSS_ERR_NOT_POSDEF;
  }
  return GSS_OK;
}

extern "C" {

int32_t gss_krig_create(gss_krig_t** out, const gss_variogram_t* vg, int32_t variant, double sk_mean,
                        int32_t degree, int32_t ndrift, const double* xdata, const double* z,
                        const double* drift_data, int64_t n, int32_t flags, void* stream) {
  GSS_ENTRY();
  GSS_REQUIRE(out != nullptr, "gss_krig_create: out is NULL");
  *out = nullptr;
  GSS_REQUIRE(xdata != nullptr && z != nullptr, "gss_krig_create: NULL data");
  // krig.jl:100-102: at least one non-missing sample
  GSS_REQUIRE(n >= 1, "all samples are missing, aborting...");
  GSS_REQUIRE(n < (1 << 30), "too many samples");
  gss_krig* h = new (std::nothrow) gss_krig();
  if (!h) return GSS_ERR_ALLOC;
  struct Guard {
    gss_krig* h;
    ~Guard() { delete h; }
  } guard{h};
  GSS_TRY(make_vgdev(vg, &h->vg));
  GSS_REQUIRE(variant != GSS_KRIG_SIMPLE || vg_is_stationary(vg),
              "simple kriging needs a stationary variogram (a power variogram has no sill)");
  h->variant = variant;
  h->sk_mean = sk_mean;
  h->degree = degree;
  h->ndrift = ndrift;
  h->dim = h->vg.dim;
  h->n = n;
  std::memset(&h->ds, 0, sizeof(h->ds));
  h->ds.variant = variant;
  h->ds.dim = h->dim;
  for (int k = 0; k < 3; ++k) {
    h->ds.center[k] = 0.0;
    h->ds.inv_scale[k] = 1.0;
  }
  switch (variant) {
    case GSS_KRIG_SIMPLE: h->nc = 0; break;
    case GSS_KRIG_ORDINARY: h->nc = 1; break;
    case GSS_KRIG_UNIVERSAL: {
      GSS_REQUIRE(degree >= 0 && degree <= 3, "universal kriging degree %d outside 0..3", degree);
      std::vector<signed char> e;
      uk_exponents(h->dim, degree, e);
      h->nc = (int)(e.size() / 3);
      GSS_REQUIRE(h->nc <= MAX_NC, "too many drift terms");
      for (int c = 0; c < h->nc; ++c)
        for (int k = 0; k < 3; ++k) h->ds.e[c][k] = e[3 * c + k];
      // centre / scale the monomials' coordinates (same polynomial space, better conditioning)
      for (int k = 0; k < h->dim; ++k) {
        double lo = xdata[k], hi = xdata[k];
        for (int64_t i = 1; i < n; ++i) {
          const double v = xdata[i * h->dim + k];
          lo = v < lo ? v : lo;
          hi = v > hi ? v : hi;
        }
        h->ds.center[k] = 0.5 * (lo + hi);
        const double half = 0.5 * (hi - lo);
        h->ds.inv_scale[k] = half > 0.0 ? 1.0 / half : 1.0;
      }
      break;
    }
    case GSS_KRIG_EXTDRIFT:
      GSS_REQUIRE(ndrift >= 1 && ndrift <= MAX_NC && drift_data != nullptr, "external drift needs 1..%d drifts",
                  MAX_NC);
      h->nc = ndrift;
      break;
    default:
      GSS_REQUIRE(false, "unknown kriging variant %d", variant);
  }
  h->ds.nc = h->nc;
  if ((flags & GSS_KRIG_NO_FACTOR) == 0)
    GSS_REQUIRE(n + h->nc >= 1 && n >= h->nc, "fewer samples (%lld) than drift constraints (%d)", (long long)n,
                h->nc);
  h->N1 = n + h->nc;
  // one spare row below the system: row N1 of W' holds the dual weights wd, so that the mean wd . rhs falls out of
  // the same MFMA product as the quadratic form (and K1 no longer depends on the fit)
  h->N1pad = round_up(h->N1 + 1, BK);
  h->ldw = round_up(h->N1 + 1, BM);

  hipStream_t s = to_stream(stream);
  GSS_TRY(h->xdata.alloc(sizeof(double) * (size_t)(n * h->dim)));
  GSS_TRY(h->z.alloc(sizeof(double) * (size_t)n));
  GSS_HIP(hipMemcpyAsync(h->xdata.p, xdata, sizeof(double) * n * h->dim, hipMemcpyHostToDevice, s));
  GSS_HIP(hipMemcpyAsync(h->z.p, z, sizeof(double) * n, hipMemcpyHostToDevice, s));
  if (variant == GSS_KRIG_EXTDRIFT) {
    GSS_TRY(h->drift_data.alloc(sizeof(double) * (size_t)(n * ndrift)));
    GSS_HIP(hipMemcpyAsync(h->drift_data.p, drift_data, sizeof(double) * n * ndrift, hipMemcpyHostToDevice, s));
  }
  GSS_HIP(hipStreamSynchronize(s));
  // the factor buffer always exists so that a broadcast can land in it
  GSS_TRY(h->factor.alloc(sizeof(double) * (size_t)(h->ldw * h->N1pad + h->N1pad)));
  if ((flags & GSS_KRIG_NO_FACTOR) == 0) {
    const bool async = (flags & GSS_KRIG_ASYNC_FIT) != 0;
    GSS_TRY(krig_factorize(h, s, async));
    if (!async) GSS_TRY(krig_fit_wait(h));   // otherwise joined by the first call that needs the factor
  }
  guard.h = nullptr;
  *out = h;
  return GSS_OK;
}

int32_t gss_krig_destroy(gss_krig_t* h) {
  GSS_ENTRY();
  delete h;
  return GSS_OK;
}

int32_t gss_krig_info(const gss_krig_t* h, int64_t* n, int32_t* nc) {
  GSS_ENTRY();
  GSS_REQUIRE(h != nullptr, "NULL handle");
  if (n) *n = h->n;
  if (nc) *nc = h->nc;
  return GSS_OK;
}

int32_t gss_krig_factor_buffer(gss_krig_t* h, void** dev_ptr, int64_t* bytes) {
  GSS_ENTRY();
  GSS_REQUIRE(h != nullptr && dev_ptr != nullptr && bytes != nullptr, "NULL argument");
  GSS_TRY(krig_fit_wait(h));   // (an asynchronous fit: the buffer is handed out complete)
  *dev_ptr = h->factor.p;
  *bytes = (int64_t)(sizeof(double) * (size_t)(h->ldw * h->N1pad + h->N1pad));
  return GSS_OK;
}

int32_t gss_krig_adopt_factor(gss_krig_t* h) {
  GSS_ENTRY();
  GSS_REQUIRE(h != nullptr, "NULL handle");
  h->factored = true;
  return GSS_OK;
}

int32_t gss_krig_predict_global(gss_krig_t* h, const double* xdom, const double* drift_dom, int64_t m,
                                double* mean, double* var, uint8_t* status, int32_t mem, void* stream) {
  GSS_ENTRY();
  GSS_REQUIRE(h != nullptr, "NULL handle");
  GSS_REQUIRE(h->factored, "handle has no factor (created with GSS_KRIG_NO_FACTOR and never adopted one)");
  GSS_REQUIRE(m >= 0 && (m == 0 || (xdom && mean && var)), "gss_krig_predict_global: NULL array");
  GSS_REQUIRE(h->variant != GSS_KRIG_EXTDRIFT || drift_dom != nullptr, "external drift values missing");
  if (m == 0) return GSS_OK;
  hipStream_t s = to_stream(stream);
  const int dim = h->dim;

  static uint64_t attr_set = 0;
  if (first_on_this_device(attr_set)) {
    GSS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(krig_quadform_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)QUADFORM_LDS_BYTES));
    GSS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(krig_quadform_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)QUADFORM_LDS_BYTES));
  }

  int64_t mc = krig_chunk_points(h->N1pad, m);
  // host arrays: pieces of the call (two rounds of resident workgroups of the quadratic form each) overlap their
  // transfers with the computation of their neighbours
  HostPipe pipe;
  GSS_TRY(pipe.begin(mem, m, s));
  const bool piped = pipe.on;
  const int64_t piece = HostPipe::PIECE;
  if (piped && piece >= 256 && mc > piece) mc = piece;
  double *Rws = nullptr, *mpart = nullptr;
  GSS_TRY(krig_workspace(h->N1pad, mc, s, &Rws, &mpart));
  const int64_t ldr = mc;
  const int seg_len = (int)((h->n + NSEG - 1) / NSEG);

  Staged sx, sd, smean, svar, sstat;
  if (piped) {
    GSS_TRY(sx.out(const_cast<double*>(xdom), sizeof(double) * m * dim, mem));   // device scratch only
    if (h->variant == GSS_KRIG_EXTDRIFT)
      GSS_TRY(sd.out(const_cast<double*>(drift_dom), sizeof(double) * m * h->ndrift, mem));
  } else {
    GSS_TRY(sx.in(xdom, sizeof(double) * m * dim, mem, s));
    if (h->variant == GSS_KRIG_EXTDRIFT) GSS_TRY(sd.in(drift_dom, sizeof(double) * m * h->ndrift, mem, s));
  }
  GSS_TRY(smean.out(mean, sizeof(double) * m, mem));
  GSS_TRY(svar.out(var, sizeof(double) * m, mem));
  GSS_TRY(sstat.out(status, (size_t)m, mem));
  if (piped) {
    pipe.add_in(xdom, sx.p, sizeof(double) * dim);
    if (h->variant == GSS_KRIG_EXTDRIFT) pipe.add_in(drift_dom, sd.p, sizeof(double) * h->ndrift);
    pipe.add_out(mean, smean.p, sizeof(double));
    pipe.add_out(var, svar.p, sizeof(double));
    pipe.add_out(status, sstat.p, 1);
  }

  for (int64_t off = 0; off < m; off += mc) {
    const int64_t mv = (m - off) < mc ? (m - off) : mc;
    const int64_t cols = round_up(mv, 256);  // multiple of BN as well
    const double* x0 = sx.as<double>() + off * dim;
    GSS_TRY(pipe.fetch(off, mv, s));
    const int nblk = (int)(cols / 256);
    dim3 g1((unsigned)(nblk * NSEG));
    const int nrows = (int)(h->N1pad - h->n);
    {
      ProfScope ps("krig_rhs", s);
      if (h->block_nsub > 0) {
        const dim3 gb((unsigned)((ldr + 255) / 256));
        const double* bc = h->block_cell;
        switch (dim) {
          case 1: hipLaunchKernelGGL(krig_rhs_block_kernel<1>, gb, dim3(256), 0, s, h->vg, h->xdata.as<double>(), (int)h->n, x0, mv, Rws, ldr, h->block_nsub, bc[0], bc[1], bc[2]); break;
          case 2: hipLaunchKernelGGL(krig_rhs_block_kernel<2>, gb, dim3(256), 0, s, h->vg, h->xdata.as<double>(), (int)h->n, x0, mv, Rws, ldr, h->block_nsub, bc[0], bc[1], bc[2]); break;
          default: hipLaunchKernelGGL(krig_rhs_block_kernel<3>, gb, dim3(256), 0, s, h->vg, h->xdata.as<double>(), (int)h->n, x0, mv, Rws, ldr, h->block_nsub, bc[0], bc[1], bc[2]); break;
        }
      } else
      switch (dim) {
        case 1: launch_krig_rhs<1>(g1, s, h->vg, h->xdata.as<double>(), (int)h->n, x0, mv, Rws, ldr, seg_len, nblk); break;
        case 2: launch_krig_rhs<2>(g1, s, h->vg, h->xdata.as<double>(), (int)h->n, x0, mv, Rws, ldr, seg_len, nblk); break;
        default: launch_krig_rhs<3>(g1, s, h->vg, h->xdata.as<double>(), (int)h->n, x0, mv, Rws, ldr, seg_len, nblk); break;
      }
    GSS_HIP(hipGetLastError());
    // drift rows n..N1-1 and zero rows up to N1pad, plus their share of the mean
    if (nrows > 0) {
      const double* dv = h->variant == GSS_KRIG_EXTDRIFT ? sd.as<double>() + off * h->ndrift : nullptr;
      GSS_TRY(launch_drift_rows(h, x0, dv, mv, cols, Rws + h->n * ldr, ldr, nrows, s));
    }
    }
    GSS_TRY(krig_join_device(h, s));   // an asynchronous fit ran beside the assembly; the quadratic form needs it
    ProfScope pq("krig_quadform", s);
    {
      const int nstrips = (int)(cols / BN);
      const int nI = (int)((h->N1 + BM) / BM);  // row blocks of rows 0..N1 (row N1 = dual weights)
      double* qpart = mpart;
      const double mean0 = h->variant == GSS_KRIG_SIMPLE ? h->sk_mean : 0.0;
      uint8_t* stp = status ? sstat.as<uint8_t>() + off : nullptr;
      const double c00 = h->block_nsub > 0 ? h->block_cvv : h->vg.sill;   // variance = C(V, V) - rhs . weights
#define GSS_K3_ARGS(S0, NS) h->Wp(), h->ldw, (int)h->N1pad, (int)h->n, (int)h->N1, Rws, ldr, c00, mean0, mv, \
                            smean.as<double>() + off, svar.as<double>() + off, stp, qpart, (S0), (NS)
      // Whole rounds of 512 resident workgroups (2 per CU) run one workgroup per strip; the remainder strips
      // would occupy a full extra round, so they run as (strip, row block) units, which pack ~3x tighter.
      const int nmain = (nstrips / 512) * 512;
      const int nrem = nstrips - nmain;
      if (nmain > 0)
        hipLaunchKernelGGL((krig_quadform_kernel<false>), dim3((unsigned)nmain), dim3(256), QUADFORM_LDS_BYTES, s,
                           GSS_K3_ARGS(0, nmain));
      if (nrem > 0) {
        const unsigned grid = (unsigned)(8 * ((nrem + 7) / 8) * nI);
        hipLaunchKernelGGL((krig_quadform_kernel<true>), dim3(grid), dim3(256), QUADFORM_LDS_BYTES, s,
                           GSS_K3_ARGS(nmain, nrem));
        const int64_t pbeg = (int64_t)nmain * BN;
        if (mv > pbeg)
          hipLaunchKernelGGL(krig_finish_kernel, dim3((unsigned)((mv - pbeg + 255) / 256)), dim3(256), 0, s, qpart + pbeg,
                             nI, ldr, c00, mv - pbeg, svar.as<double>() + off + pbeg, stp ? stp + pbeg : nullptr);
      }
#undef GSS_K3_ARGS
    }
    GSS_HIP(hipGetLastError());
    GSS_TRY(pipe.deliver(off, mv, s));
  }
  if (piped) {
    GSS_TRY(pipe.finish(s));
  } else {
    GSS_TRY(smean.back(mean, sizeof(double) * m, mem, s));
    GSS_TRY(svar.back(var, sizeof(double) * m, mem, s));
    GSS_TRY(sstat.back(status, (size_t)m, mem, s));
  }
  return krig_fit_wait(h);   // status of an asynchronous fit (it finished while the assembly ran)
}


int32_t gss_krig_set_block_support(gss_krig_t* h, const double* cell, int32_t nsub, void* stream) {
  GSS_ENTRY();
  GSS_REQUIRE(h != nullptr, "NULL handle");
  if (nsub <= 0 || cell == nullptr) {   // back to point support
    h->block_nsub = 0;
    return GSS_OK;
  }
  GSS_REQUIRE(nsub <= 8, "block support: at most 8 samples per axis (got %d)", nsub);
  GSS_REQUIRE(h->vg.kind != GSS_VG_POWER, "block support needs a stationary variogram");
  GSS_REQUIRE(h->variant != GSS_KRIG_EXTDRIFT && !(h->variant == GSS_KRIG_UNIVERSAL && h->degree > 1),
              "block support: simple / ordinary kriging or a drift of degree <= 1 (the cell average of a linear drift is "
              "its value at the centroid; higher degrees and external drifts would need their own cell averages)");
  for (int a = 0; a < h->dim; ++a) GSS_REQUIRE(cell[a] > 0.0, "block support: cell sizes must be positive");
  hipStream_t s = to_stream(stream);
  DevBuf out;
  GSS_TRY(out.alloc(sizeof(double)));
  double c[3] = {0.0, 0.0, 0.0};
  for (int a = 0; a < h->dim; ++a) c[a] = cell[a];
  switch (h->dim) {
    case 1: hipLaunchKernelGGL(block_cvv_kernel<1>, dim3(1), dim3(256), 0, s, h->vg, nsub, c[0], c[1], c[2], out.as<double>()); break;
    case 2: hipLaunchKernelGGL(block_cvv_kernel<2>, dim3(1), dim3(256), 0, s, h->vg, nsub, c[0], c[1], c[2], out.as<double>()); break;
    default: hipLaunchKernelGGL(block_cvv_kernel<3>, dim3(1), dim3(256), 0, s, h->vg, nsub, c[0], c[1], c[2], out.as<double>()); break;
  }
  GSS_HIP(hipGetLastError());
  double cvv = 0.0;
  GSS_HIP(hipMemcpyAsync(&cvv, out.p, sizeof(double), hipMemcpyDeviceToHost, s));
  GSS_HIP(hipStreamSynchronize(s));
  h->block_nsub = nsub;
  for (int a = 0; a < 3; ++a) h->block_cell[a] = c[a];
  h->block_cvv = cvv;
  return GSS_OK;
}

int32_t gss_krig_predict_knn(gss_krig_t* h, const double* xdom, const double* drift_dom, int64_t m, int32_t k,
                             int32_t minneighbors, double radius, const double* inv_radii, int32_t metric,
                             double metric_param, double* mean, double* var, uint8_t* status, int32_t* idx_out,
                             int32_t* count_out, int32_t mem, void* stream) {
  GSS_ENTRY();
  GSS_REQUIRE(h != nullptr, "NULL handle");

  GSS_TRY(check_metric(metric, metric_param, h->dim, radius, inv_radii));
  GSS_REQUIRE(m >= 0 && (m == 0 || (xdom && mean && var)), "gss_krig_predict_knn: NULL array");
  GSS_REQUIRE(k >= 1 && k <= h->n, "maxneighbors %d outside 1..%lld (searcher_ui clamps it, ui.jl:18-20)", k,
              (long long)h->n);
  GSS_REQUIRE(h->variant != GSS_KRIG_EXTDRIFT || drift_dom != nullptr, "external drift values missing");
  if (m == 0) return GSS_OK;
  hipStream_t s = to_stream(stream);
  const int dim = h->dim;
  Staged sx, sd, smean, svar, sstat, sidx, scnt;
  HostPipe pipe;   // host arrays: in and out piece by piece beside the computation (gss_internal.h)
  GSS_TRY(pipe.begin(mem, m, s));
  if (pipe.on) {
    GSS_TRY(sx.out(const_cast<double*>(xdom), sizeof(double) * m * dim, mem));   // device scratch only
    if (h->variant == GSS_KRIG_EXTDRIFT)
      GSS_TRY(sd.out(const_cast<double*>(drift_dom), sizeof(double) * m * h->ndrift, mem));
  } else {
    GSS_TRY(sx.in(xdom, sizeof(double) * m * dim, mem, s));
    if (h->variant == GSS_KRIG_EXTDRIFT) GSS_TRY(sd.in(drift_dom, sizeof(double) * m * h->ndrift, mem, s));
  }
  GSS_TRY(smean.out(mean, sizeof(double) * m, mem));
  GSS_TRY(svar.out(var, sizeof(double) * m, mem));
  GSS_TRY(sstat.out(status, (size_t)m, mem));
  GSS_TRY(sidx.out(idx_out, sizeof(int32_t) * (size_t)(m * k), mem));
  GSS_TRY(scnt.out(count_out, sizeof(int32_t) * (size_t)m, mem));
  if (pipe.on) {
    pipe.add_in(xdom, sx.p, sizeof(double) * dim);
    if (h->variant == GSS_KRIG_EXTDRIFT) pipe.add_in(drift_dom, sd.p, sizeof(double) * h->ndrift);
    pipe.add_out(mean, smean.p, sizeof(double));
    pipe.add_out(var, svar.p, sizeof(double));
    pipe.add_out(status, sstat.p, 1);
    pipe.add_out(idx_out, sidx.p, sizeof(int32_t) * (size_t)k);
    pipe.add_out(count_out, scnt.p, sizeof(int32_t));
  }
  GSS_TRY(krig_local_dev(h->vg, h->variant, h->nc, dim, &h->ds.e[0][0], h->ds.inv_scale[0], h->sk_mean,
                         h->xdata.as<double>(), h->z.as<double>(), h->drift_data.as<double>(), h->n,
                         sx.as<double>(), sd.as<double>(), m, k, minneighbors, radius, inv_radii, smean.as<double>(),
                         svar.as<double>(), sstat.as<uint8_t>(), sidx.as<int>(), scnt.as<int>(), s, metric, &pipe,
                         h->block_nsub, h->block_cell, h->block_cvv));
  if (pipe.on) return GSS_OK;   // everything is home (krig_local_dev ends with pipe.finish)
  GSS_TRY(smean.back(mean, sizeof(double) * m, mem, s));
  GSS_TRY(svar.back(var, sizeof(double) * m, mem, s));
  GSS_TRY(sstat.back(status, (size_t)m, mem, s));
  GSS_TRY(sidx.back(idx_out, sizeof(int32_t) * (size_t)(m * k), mem, s));
  GSS_TRY(scnt.back(count_out, sizeof(int32_t) * (size_t)m, mem, s));
  return GSS_OK;
}


// One factorised system, many data vectors: the conditional-FFTGS pattern (fft.jl:176-188) where every
// realisation is kriged from the same locations.  mean_b(p) = wd_b . rhs(p) with WD = K^-1 [Z; 0]:
// one GEMM for the dual weights and one GEMM per chunk of domain points.
int32_t gss_krig_predict_global_batch(gss_krig_t* h, const double* xdom, int64_t m, const double* zbatch,
                                      int64_t nbatch, double* mean_out, int32_t mem, void* stream) {
  GSS_ENTRY();
  GSS_REQUIRE(h != nullptr, "NULL handle");
  GSS_REQUIRE(h->factored, "handle has no factor");
  GSS_REQUIRE(m >= 0 && nbatch >= 0, "negative sizes");
  GSS_REQUIRE(h->variant != GSS_KRIG_EXTDRIFT, "batched prediction is not available with external drifts");
  if (m == 0 || nbatch == 0) return GSS_OK;
  GSS_REQUIRE(xdom && zbatch && mean_out, "gss_krig_predict_global_batch: NULL array");
  hipStream_t s = to_stream(stream);
  GSS_TRY(krig_join_device(h, s));
  GSS_TRY(krig_fit_wait(h));
  const int dim = h->dim;
  const int64_t n = h->n, N1 = h->N1, ldw = h->ldw;
  Staged sx, sz, so;
  GSS_TRY(sx.in(xdom, sizeof(double) * m * dim, mem, s));
  GSS_TRY(sz.in(zbatch, sizeof(double) * nbatch * n, mem, s));
  GSS_TRY(so.out(mean_out, sizeof(double) * (size_t)(nbatch * m), mem));

  DevBuf Zm, U, WD;
  GSS_TRY(Zm.alloc(sizeof(double) * (size_t)(ldw * nbatch)));
  GSS_TRY(U.alloc(sizeof(double) * (size_t)(ldw * nbatch)));
  GSS_TRY(WD.alloc(sizeof(double) * (size_t)(ldw * nbatch)));
  GSS_HIP(hipMemsetAsync(Zm.p, 0, Zm.bytes, s));
  GSS_HIP(hipMemsetAsync(WD.p, 0, WD.bytes, s));
  GSS_HIP(hipMemcpy2DAsync(Zm.p, sizeof(double) * ldw, sz.p, sizeof(double) * n, sizeof(double) * n, nbatch,
                           hipMemcpyDeviceToDevice, s));
  if (h->variant == GSS_KRIG_SIMPLE && h->sk_mean != 0.0) {
    // only the first n rows of each column hold data; the padding rows are multiplied by zero columns of W'
    hipLaunchKernelGGL(add_scalar_kernel, dim3((unsigned)((ldw * nbatch + 255) / 256)), dim3(256), 0, s,
                       Zm.as<double>(), ldw * nbatch, -h->sk_mean);
  }
  // U = W' Zm ; flip constraint rows ; WD = W'' U
  GSS_TRY(gemm_f64(N1, nbatch, n, 1.0, h->Wp(), 1, ldw, Zm.as<double>(), 1, ldw, 0.0, U.as<double>(), 1, ldw,
                   false, s));
  if (h->nc > 0) {
    hipLaunchKernelGGL(flip_rows_kernel, dim3(1, (unsigned)nbatch), dim3(256), 0, s, U.as<double>(), ldw, n, N1,
                       nbatch);
  }
  GSS_TRY(gemm_f64(N1, nbatch, N1, 1.0, h->Wp(), ldw, 1, U.as<double>(), 1, ldw, 0.0, WD.as<double>(), 1, ldw,
                   false, s));
  GSS_HIP(hipGetLastError());

  // out(b, p) = sum_k WD(k, b) R(k, p), sixteen data vectors per pass, R never materialised
  const double add = (h->variant == GSS_KRIG_SIMPLE) ? h->sk_mean : 0.0;
  DevBuf WDt;
  GSS_TRY(WDt.alloc(sizeof(double) * (size_t)(N1 * BATCH_NB)));
  for (int64_t b0 = 0; b0 < nbatch; b0 += BATCH_NB) {
    const int nb = (int)((nbatch - b0) < BATCH_NB ? (nbatch - b0) : BATCH_NB);
    hipLaunchKernelGGL(wd_rows_kernel, dim3((unsigned)((N1 + 255) / 256)), dim3(256), 0, s, WD.as<double>() + b0 * ldw, ldw,
                       (int)N1, nb, WDt.as<double>());
    switch (dim) {
      case 1: launch_krig_batch_mean<1>(s, h->vg, h->ds, h->xdata.as<double>(), (int)n, sx.as<double>(), m, WDt.as<double>(), nb, add, so.as<double>() + b0 * m, m); break;
      case 2: launch_krig_batch_mean<2>(s, h->vg, h->ds, h->xdata.as<double>(), (int)n, sx.as<double>(), m, WDt.as<double>(), nb, add, so.as<double>() + b0 * m, m); break;
      default: launch_krig_batch_mean<3>(s, h->vg, h->ds, h->xdata.as<double>(), (int)n, sx.as<double>(), m, WDt.as<double>(), nb, add, so.as<double>() + b0 * m, m); break;
    }
    GSS_HIP(hipGetLastError());
  }
  GSS_HIP(hipGetLastError());
  GSS_TRY(so.back(mean_out, sizeof(double) * (size_t)(nbatch * m), mem, s));
  GSS_HIP(hipStreamSynchronize(s));  // Zm / U / WD are freed on return
  return GSS_OK;
}

}  // extern "C"
