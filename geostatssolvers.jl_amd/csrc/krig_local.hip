// K5: moving-neighbourhood kriging -- one small kriging system per domain point.
// Replaces the body of approxsolve's loop (/root/reference/src/estimation/krig.jl:205-228):
//   :210  search!(neighbors, center, searcher)              -> K4 (knn.hip)
//   :213  nneigh < minneighbors -> missing                  -> status GSS_PT_MISSING, NaN outputs
//   :223  fit(estimator, view(pdata, neighbors))            -> k' x k' covariance, Cholesky in LDS
//   :226  predictprob(krig, var, pdomain[ind])              -> mean / variance by block elimination
//
// One wave owns one domain point; lane j owns neighbour j.  With
// C = L L', Y = L^-1 [c0 | F | z] (forward substitution, all right-hand sides in one sweep):
//   q = |y_c|^2, a = y_z . y_c, S = Y_F' Y_F, r = Y_F' y_c - f0, t = Y_F' y_z
//   sigma^2 = max(0, sill - q + r' S^-1 r),   mu = a - t' S^-1 r        (SK: mu = mean + a, nc = 0)
// which equals solving [C F; F' 0][lambda; nu] = [c0; f0] without forming weights.
// Universal-kriging monomials are taken about the estimation point (the polynomial space is
// translation invariant), so f0 = (1, 0, ..., 0).
#include "gss_internal.h"
#include "tile16.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace gss {

constexpr int LMAX_K = 64;
constexpr int LMAX_NC = 10;
constexpr int LMAX_RHS = LMAX_NC + 2;

struct LocalSpec {
  int variant;
  int nc;
  int dim;
  signed char e[LMAX_NC][3];
  double inv_scale;  // monomial scaling (1 / data extent), conditioning only
  double sk_mean;
};

__device__ __forceinline__ int tri(int i) { return (i * (i + 1)) >> 1; }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// ---------------------------------------------------------------------------------------------
// More than 64 neighbours (maxneighbors in 65 .. n - 1; krig.jl:201-210 and ui.jl:16-23 accept any count).
// One workgroup of 256 threads per estimation point, same mathematics as the MFMA-tile kernel below with the
// right-hand sides [c0 | z | F] appended to the covariance block as extra ROWS: a root-free Cholesky sweep over the
// (k' + nrhs) x k' array leaves A(i, c) = L(i, c) sqrt(d_c) in the triangle and, in the extra rows, the forward-
// substituted right-hand sides in the same scaling, so that every dot product of the block elimination is
// sum_j A(r, j) A(s, j) / d_j.  Inner products skip the square roots (one barrier per column instead of two).
// The array lives in LDS while it fits (k' + nrhs up to ~180 rows), otherwise in a per-workgroup slab of HBM
// (L2-resident for a few hundred neighbours): the same code runs on either through a generic pointer.
// This is the functional path for large neighbourhoods, not a tuned one.
// ---------------------------------------------------------------------------------------------
constexpr int BIG_NT = 256;
constexpr int BIG_LDS_DOUBLES = 17408;   // 136 KiB of matrix in LDS
constexpr int BIG_MAX_K = 4096;

template <int DIM>
__global__ __launch_bounds__(BIG_NT) void krig_local_big_kernel(VgDev vg, LocalSpec sp, const double* __restrict__ xdata,
                                                                const double* __restrict__ z,
                                                                const double* __restrict__ drift_data,
                                                                const double* __restrict__ x0,
                                                                const double* __restrict__ drift_dom, int64_t m, int k,
                                                                int minneighbors, const int* __restrict__ idx,
                                                                const int* __restrict__ count,
                                                                double* __restrict__ mean_out,
                                                                double* __restrict__ var_out,
                                                                uint8_t* __restrict__ status_out,
                                                                double* __restrict__ scratch, int64_t slab,
                                                                int use_lds) {
  extern __shared__ double big_sm[];
  __shared__ double G[LMAX_RHS][LMAX_RHS];
  __shared__ double Ssm[LMAX_NC][LMAX_NC + 1];
  __shared__ double rv[LMAX_NC], tv[LMAX_NC];
  const int tid = threadIdx.x;
  const int nc = sp.nc;
  const int nrhs = 2 + nc;
  const double NaN = __longlong_as_double(0x7ff8000000000000LL);
  for (int64_t p = blockIdx.x; p < m; p += gridDim.x) {
    const int cnt = count[p];
    __syncthreads();  // the previous point's results have been read
    if (cnt < minneighbors || cnt <= 0) {  // krig.jl:213-214
      if (tid == 0) {
        mean_out[p] = NaN;
        var_out[p] = NaN;
        status_out[p] = GSS_PT_MISSING;
      }
      continue;
    }
    const int K1 = cnt;
    const int RT = K1 + nrhs;
    const int64_t ntri = (int64_t)K1 * (K1 + 1) / 2;
    double* M = use_lds ? big_sm : scratch + (int64_t)blockIdx.x * slab;
    double* invd = M + ntri + (int64_t)nrhs * K1;
    const int* nb = idx + p * k;
    double c0[DIM];
#pragma unroll
    for (int a = 0; a < DIM; ++a) c0[a] = x0[p * DIM + a];
    bool failed = false;
    // ---- assembly: covariance triangle (entries dealt round-robin), then the extra rows
    for (int64_t e = tid; e < ntri; e += BIG_NT) {
      int i = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
      while ((int64_t)(i + 1) * (i + 2) / 2 <= e) ++i;
      while ((int64_t)i * (i + 1) / 2 > e) --i;
      const int c = (int)(e - (int64_t)i * (i + 1) / 2);
      const int ni = nb[i], ncl = nb[c];
      double xi[DIM], xc[DIM];
#pragma unroll
      for (int a = 0; a < DIM; ++a) {
        xi[a] = xdata[(int64_t)ni * DIM + a];
        xc[a] = xdata[(int64_t)ncl * DIM + a];
      }
      M[e] = cov_pair<DIM>(vg, xi, xc);
    }
    for (int j = tid; j < K1; j += BIG_NT) {
      const int nj = nb[j];
      double xj[DIM];
#pragma unroll
      for (int a = 0; a < DIM; ++a) xj[a] = xdata[(int64_t)nj * DIM + a];
      M[ntri + j] = cov_pair<DIM>(vg, xj, c0);
      double zz = z[nj];
      if (sp.variant == GSS_KRIG_SIMPLE) zz -= sp.sk_mean;
      M[ntri + K1 + j] = zz;
      for (int c = 0; c < nc; ++c) {
        double f = 1.0;
        if (sp.variant == GSS_KRIG_UNIVERSAL) {
#pragma unroll
          for (int a = 0; a < DIM; ++a) {
            const double u = (xj[a] - c0[a]) * sp.inv_scale;
            for (int q = 0; q < sp.e[c][a]; ++q) f *= u;
          }
        } else if (sp.variant == GSS_KRIG_EXTDRIFT) {
          f = drift_data[(int64_t)nj * nc + c];
        }
        M[ntri + (int64_t)(2 + c) * K1 + j] = f;
      }
    }
    __syncthreads();
    // ---- root-free Cholesky sweep by columns: row i of the array is owned by thread i mod 256
    for (int j = 0; j < K1; ++j) {
      const double* rowj = M + (int64_t)j * (j + 1) / 2;
      int pivot_bad = 0;  // set by the one thread that owns the diagonal entry of this column
      for (int i = j + tid; i < RT; i += BIG_NT) {
        double* rowi = i < K1 ? M + (int64_t)i * (i + 1) / 2 : M + ntri + (int64_t)(i - K1) * K1;
        double a0 = rowi[j], a1 = 0.0;
        int c = 0;
        for (; c + 2 <= j; c += 2) {
          a0 = fma(-rowi[c] * invd[c], rowj[c], a0);
          a1 = fma(-rowi[c + 1] * invd[c + 1], rowj[c + 1], a1);
        }
        for (; c < j; ++c) a0 = fma(-rowi[c] * invd[c], rowj[c], a0);
        const double acc = a0 + a1;
        rowi[j] = acc;
        if (i == j) {
          if (!(acc > 0.0)) pivot_bad = 1;
          invd[j] = 1.0 / acc;
        }
      }
      // the column's barrier carries the verdict on its pivot, so every wave decides on the SAME column (a flag in
      // LDS read behind the barrier could already be the next column's)
      if (__syncthreads_or(pivot_bad)) {
        failed = true;
        break;
      }
    }
    if (failed) {
      if (tid == 0) {
        mean_out[p] = NaN;
        var_out[p] = NaN;
        status_out[p] = GSS_PT_SINGULAR;
      }
      continue;
    }
    // ---- Gram matrix of the forward-substituted right-hand sides: one (r, s) pair per wave slot
    {
      const int lane = tid & 63, wave = tid >> 6;
      const int npairs = nrhs * (nrhs + 1) / 2;
      for (int pr = wave; pr < npairs; pr += BIG_NT / 64) {
        int r = 0;
        while ((r + 1) * (r + 2) / 2 <= pr) ++r;
        const int sidx = pr - r * (r + 1) / 2;
        const double* yr = M + ntri + (int64_t)r * K1;
        const double* ys = M + ntri + (int64_t)sidx * K1;
        double acc = 0.0;
        for (int j = lane; j < K1; j += 64) acc = fma(yr[j] * invd[j], ys[j], acc);
        acc = wave_sum(acc);
        if (lane == 0) {
          G[r][sidx] = acc;
          G[sidx][r] = acc;
        }
      }
    }
    __syncthreads();
    if (tid == 0) {
      const double qf = G[0][0], af = G[1][0];
      double rsr = 0.0, tsr = 0.0;
      int okS = 1;
      if (nc > 0) {
        for (int c = 0; c < nc; ++c) {
          double f0 = 1.0;
          if (sp.variant == GSS_KRIG_UNIVERSAL) f0 = (sp.e[c][0] + sp.e[c][1] + sp.e[c][2]) == 0 ? 1.0 : 0.0;
          else if (sp.variant == GSS_KRIG_EXTDRIFT) f0 = drift_dom[p * nc + c];
          rv[c] = G[2 + c][0] - f0;
          tv[c] = G[2 + c][1];
          for (int c2 = 0; c2 <= c; ++c2) Ssm[c][c2] = G[2 + c][2 + c2];
        }
        for (int j = 0; j < nc && okS; ++j) {
          double d = Ssm[j][j];
          for (int c = 0; c < j; ++c) d -= Ssm[j][c] * Ssm[j][c];
          if (!(d > 0.0)) {
            okS = 0;
            break;
          }
          const double sq = sqrt(d);
          Ssm[j][j] = sq;
          for (int i = j + 1; i < nc; ++i) {
            double v = Ssm[i][j];
            for (int c = 0; c < j; ++c) v -= Ssm[i][c] * Ssm[j][c];
            Ssm[i][j] = v / sq;
          }
        }
        if (okS) {
          for (int i = 0; i < nc; ++i) {
            double u = rv[i], v = tv[i];
            for (int c = 0; c < i; ++c) {
              u -= Ssm[i][c] * rv[c];
              v -= Ssm[i][c] * tv[c];
            }
            u /= Ssm[i][i];
            v /= Ssm[i][i];
            rv[i] = u;
            tv[i] = v;
            rsr += u * u;
            tsr += u * v;
          }
        }
      }
      if (!okS) {
        mean_out[p] = NaN;
        var_out[p] = NaN;
        status_out[p] = GSS_PT_SINGULAR;
      } else {
        const double mu = (sp.variant == GSS_KRIG_SIMPLE ? sp.sk_mean : 0.0) + af - tsr;
        const double v = vg.sill - qf + rsr;
        mean_out[p] = mu;
        var_out[p] = v > 0.0 ? v : 0.0;
        status_out[p] = GSS_PT_OK;
      }
    }
  }
}

// Block elimination on the (2 + nc) x (2 + nc) Gram matrix G = Y'Y of the forward-substituted right-hand sides
// [c0 | z | F] (one wave; lane i = drift term i, everything in registers; lanes 16..63 shadow lanes 0..15):
// S = Y_F'Y_F = L L', u = L^-1 (Y_F'y_c - f0), v = L^-1 Y_F'y_z, r'S^-1 r = |u|^2, t'S^-1 r = u.v, then
// sigma^2 = max(0, sill - q + r'S^-1 r), mu = a - t'S^-1 r (module header).  Writes the point's outputs when `live`.
__device__ __forceinline__ void gram_finish(const double (*G)[17], const signed char (*se)[4], const VgDev& vg,
                                            const LocalSpec& sp, const double* __restrict__ drift_dom, int64_t p,
                                            int lane, bool live, double* __restrict__ mean_out,
                                            double* __restrict__ var_out, uint8_t* __restrict__ status_out) {
  const int nc = sp.nc;
  const double NaN = __longlong_as_double(0x7ff8000000000000LL);
  const double qf = G[0][0], af = G[1][0];
  double rsr = 0.0, tsr = 0.0;
  bool okS = true;
  if (nc > 0) {
    const int i = (lane & 15) < LMAX_NC ? (lane & 15) : LMAX_NC - 1;  // lanes 16..63 shadow lanes 0..15 (bc16)
    double srow[LMAX_NC];
#pragma unroll
    for (int cc = 0; cc < LMAX_NC; ++cc) srow[cc] = G[2 + i][2 + cc];
    double f0 = 1.0;
    if (sp.variant == GSS_KRIG_UNIVERSAL) f0 = (se[i][0] + se[i][1] + se[i][2]) == 0 ? 1.0 : 0.0;
    else if (sp.variant == GSS_KRIG_EXTDRIFT) f0 = drift_dom[p * nc + (i < nc ? i : 0)];
    double u = G[2 + i][0] - f0, v = G[2 + i][1];
    static_for<0, LMAX_NC>([&](auto J) {
      constexpr int j = decltype(J)::value;
      if (j < nc) {
        double d = bc16<j>(srow[j]);
        if (!(d > 0.0)) {
          okS = false;
          d = 1.0;
        }
        double y = __builtin_amdgcn_rsq(d);
        const double h = 0.5 * d;
        y = fma(y, fma(-h * y, y, 0.5), y);
        y = fma(y, fma(-h * y, y, 0.5), y);
        double lij = srow[j] * y;  // L[i][j] for i > j; lane cc holds L[cc][j]
        if (j + 1 < LMAX_NC) dpp_fence(lij);
        static_for<j + 1, LMAX_NC>([&](auto CC) {
          constexpr int cc = decltype(CC)::value;
          fmac_bc16<cc, true>(srow[cc], lij, lij);  // srow[cc] -= L[cc][j] * L[i][j]
        });
        const double uj = bc16<j>(u) * y, vj = bc16<j>(v) * y;
        rsr = fma(uj, uj, rsr);
        tsr = fma(uj, vj, tsr);
        u = fma(-lij, uj, u);
        v = fma(-lij, vj, v);
      }
    });
  }
  if (live && lane == 0) {
    if (!okS) {
      mean_out[p] = NaN;
      var_out[p] = NaN;
      status_out[p] = GSS_PT_SINGULAR;
    } else {
      const double mu = (sp.variant == GSS_KRIG_SIMPLE ? sp.sk_mean : 0.0) + af - tsr;
      const double vv = vg.sill - qf + rsr;
      mean_out[p] = mu;
      var_out[p] = vv > 0.0 ? vv : 0.0;
      status_out[p] = GSS_PT_OK;
    }
  }
}


// ---------------------------------------------------------------------------------------------
// K5, MFMA-tiled variant (default).  The k x k system is held in
// registers as 16 x 16 tiles in the accumulator layout of v_mfma_f64_16x16x4_f64 (lane l, register r <-> element
// (row (l >> 4) + 4 r, column l & 15)) and factorised as A = U'U by tiles:
//   diagonal tile   : through LDS into "lane = row" form, 16 x 16 Cholesky with the inverse V = U_kk^-1 built in the
//                     same sweep, row-local DPP broadcasts; one wave does this for the four points of its workgroup
//                     (tile16.h potrf16_inverse_x4); V back in tile layout (U_kk itself is never needed again)
//   U_kj  = V' A_kj,  A_ij -= U_ki' U_kj,  right-hand sides  Y_k = V' B_k,  B_i -= U_ki' Y_k
// Every product has the form X'Y with X and Y in tile layout, which is exactly what the MFMA consumes: register s
// of X is the A operand of k-slice s (A[i][k] on lane i + 16 k) and register s of Y is its B operand, so no data
// ever changes layout between products.  The right-hand sides [c0 | z | F] ride along as 16 columns, and one
// last product G = Y'Y yields every dot product the block elimination needs (|y_c|^2, y_z.y_c, Y_F'Y_F, ...).
// Rows beyond the neighbour count are padded with the identity.
// ---------------------------------------------------------------------------------------------

// step KK of the tile factorisation, after V = U_KK^-1 is known: U_KK,j = V' A_KK,j; Y_KK = V' B_KK; trailing updates
template <int KK>
__device__ __forceinline__ void k5_block_step(d4_t (&T)[10], d4_t (&B)[4], const d4_t& V, int nt) {
  const d4_t zero4 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int j = KK + 1; j < 4; ++j)
    if (j < nt) T[tile_id(KK, j)] = xty(V, T[tile_id(KK, j)], zero4);
  B[KK] = xty(V, B[KK], zero4);
#pragma unroll
  for (int i = KK + 1; i < 4; ++i) {
    if (i < nt) {
      const d4_t N = -T[tile_id(KK, i)];
#pragma unroll
      for (int j = i; j < 4; ++j)
        if (j < nt) T[tile_id(i, j)] = xty(N, T[tile_id(KK, j)], T[tile_id(i, j)]);
      B[i] = xty(N, B[KK], B[i]);
    }
  }
}

// Four domain points per workgroup, one per wave.  The waves only meet for the diagonal tiles: the row broadcasts of
// the 16 x 16 factorisation are local to a 16-lane row (tile16.h), so ONE wave factors the four waves' diagonal tiles
// in its four lane rows for the issue cost of one, and the duty rotates with the block step (wave kk does step kk)
// so that the four SIMDs share it.  Every wave passes every barrier: points without enough neighbours, and tile
// steps beyond a point's neighbour count, take part with an identity tile.
constexpr int K5_WAVES = 4;

template <int DIM, int KIND>
__global__ __launch_bounds__(64 * K5_WAVES) __attribute__((amdgpu_waves_per_eu(3, 3))) void krig_local_mfma_kernel(VgDev vg, LocalSpec sp, const double* __restrict__ xdata,
                                                             const double* __restrict__ z,
                                                             const double* __restrict__ drift_data,
                                                             const double* __restrict__ x0,
                                                             const double* __restrict__ drift_dom, int64_t m, int k,
                                                             int minneighbors, const int* __restrict__ idx,
                                                             const int* __restrict__ count,
                                                             double* __restrict__ mean_out,
                                                             double* __restrict__ var_out,
                                                             uint8_t* __restrict__ status_out) {
  __shared__ double nxs_[K5_WAVES][LMAX_K][3];   // neighbour coordinates, divided by the radii of the model's ball (KIND >= 0)
  __shared__ signed char se[LMAX_NC][4];
  __shared__ double S4[2][K5_WAVES][16 * 17];    // diagonal tiles in / inverse factors out, double buffered by step parity
  __shared__ int badflag[2][K5_WAVES];
  __shared__ double G_[K5_WAVES][16][17];
  // The right-hand-side columns [c0 | z | F] are evaluated with lane = neighbour (one value per lane and column, the
  // exponents of a drift term being uniform) and pass through LDS into tile layout.  Column q of wave w lives in the
  // wave's OWN pieces of the buffers above -- S4[0][w], S4[1][w], G_[w], four columns of 64 each -- which nobody
  // touches before the wave has read them back, so the hand-over needs no barrier.
  static_assert(4 * (LMAX_K + 4) <= 16 * 17 && LMAX_RHS <= 12, "right-hand-side columns do not fit the wave's own tiles");

  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  double (*nxs)[3] = nxs_[wave];
  double (*G)[17] = G_[wave];
  auto rhs_col = [&](int q) -> double* {   // column q (0 = c0, 1 = data, 2 + t = drift term t) of this wave
    double* base = q < 4 ? &S4[0][wave][0] : (q < 8 ? &S4[1][wave][0] : &G_[wave][0][0]);
    return base + (q & 3) * (LMAX_K + 4);   // 68 doubles apart: the sixteen columns a row of lanes reads spread over the banks
  };
  const int64_t pw = (int64_t)blockIdx.x * K5_WAVES + wave;
  const bool inrange = pw < m;
  const int64_t p = inrange ? pw : m - 1;
  const double NaN = __longlong_as_double(0x7ff8000000000000LL);
  // wave-uniform by construction (one point per wave): telling the compiler turns the tile-count tests below into
  // scalar branches instead of EXEC-masked regions
  int cnt = __builtin_amdgcn_readfirstlane(inrange ? count[p] : 0);
  bool live = inrange;
  if (cnt < minneighbors || cnt <= 0) {  // krig.jl:213-214
    if (inrange && lane == 0) {
      mean_out[p] = NaN;
      var_out[p] = NaN;
      status_out[p] = GSS_PT_MISSING;
    }
    live = false;
    cnt = 0;
  }
  const int nc = sp.nc;
  const int g = lane >> 4, c = lane & 15;
  double c0[DIM], c0s[DIM];
#pragma unroll
  for (int a = 0; a < DIM; ++a) {
    c0[a] = x0[p * DIM + a];
    c0s[a] = (KIND >= 0 && vg.aniso) ? c0[a] * vg.ir[a] : c0[a];
  }
  {
    const bool act = lane < cnt;
    const int nj = act ? idx[p * k + lane] : 0;
    double xj[DIM], xjs[DIM];
#pragma unroll
    for (int a = 0; a < DIM; ++a) {
      xj[a] = act ? xdata[(int64_t)nj * DIM + a] : 0.0;
      xjs[a] = (KIND >= 0 && vg.aniso) ? xj[a] * vg.ir[a] : xj[a];
      nxs[lane][a] = xjs[a];
    }
    double zz = act ? z[nj] : 0.0;
    if (sp.variant == GSS_KRIG_SIMPLE) zz -= sp.sk_mean;
    rhs_col(0)[lane] = act ? cov_pair_k<DIM, KIND>(vg, xjs, c0s) : 0.0;
    rhs_col(1)[lane] = act ? zz : 0.0;
    // drift columns: monomials about the estimation point (uniform exponents), external drifts, or the constant
    double um[DIM];
#pragma unroll
    for (int a = 0; a < DIM; ++a) um[a] = (xj[a] - c0[a]) * sp.inv_scale;
#pragma unroll
    for (int t = 0; t < LMAX_NC; ++t) {
      if (t < nc) {
        double f = 1.0;
        if (sp.variant == GSS_KRIG_UNIVERSAL) {
#pragma unroll
          for (int a = 0; a < DIM; ++a)
            for (int w = 0; w < sp.e[t][a]; ++w) f *= um[a];
        } else if (sp.variant == GSS_KRIG_EXTDRIFT) {
          f = act ? drift_data[(int64_t)nj * nc + t] : 0.0;
        }
        rhs_col(2 + t)[lane] = act ? f : 0.0;
      }
    }
    if (threadIdx.x == 0) {  // static indices only: a lane-indexed read would force the argument struct into scratch
#pragma unroll
      for (int cc = 0; cc < LMAX_NC; ++cc)
#pragma unroll
        for (int a = 0; a < 3; ++a) se[cc][a] = sp.e[cc][a];
    }
  }
  __syncthreads();
  const int nt = (cnt + 15) >> 4;

  // system matrix, upper block triangle, tile layout; the four rows a lane holds of a tile are evaluated together
  d4_t T[10];
#pragma unroll
  for (int I = 0; I < 4; ++I) {
    if (I < nt) {
      double xr[4][DIM];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int a = 0; a < DIM; ++a) xr[r][a] = nxs[16 * I + g + 4 * r][a];
#pragma unroll
      for (int J = I; J < 4; ++J) {
        if (J < nt) {
          const int col = 16 * J + c;
          double xcol[DIM], v[4];
#pragma unroll
          for (int a = 0; a < DIM; ++a) xcol[a] = nxs[col][a];
          cov_pair4_k<DIM, KIND>(vg, xr, xcol, v);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * I + g + 4 * r;
            T[tile_id(I, J)][r] = (row < cnt && col < cnt) ? v[r] : (row == col ? 1.0 : 0.0);
          }
        }
      }
    }
  }
  // right-hand sides: column 0 = c0, 1 = data, 2 + q = drift q, zero beyond; rows beyond the neighbour count are zero
  d4_t B[4];
  {
    const double* colc = rhs_col(c < LMAX_RHS ? c : 0);
    const bool used = c < 2 + nc;
#pragma unroll
    for (int K = 0; K < 4; ++K) {
      if (K < nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double v = colc[16 * K + g + 4 * r];
          B[K][r] = used ? v : 0.0;
        }
      }
    }
  }
  bool bad = false;
  const d4_t zero4 = {0.0, 0.0, 0.0, 0.0};
  // fully unrolled block steps (a rolled loop around one copy of the diagonal factorisation, with switch-selected
  // per-step code, shrinks the kernel from 84 KB to 53 KB but measured 5 % slower)
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    double* mine = S4[kk & 1][wave];
#pragma unroll
    for (int r = 0; r < 4; ++r)
      mine[(g + 4 * r) * 17 + c] = (kk < nt) ? T[tile_id(kk, kk)][r] : ((g + 4 * r) == c ? 1.0 : 0.0);
    __syncthreads();
    if (wave == kk) potrf16_inverse_x4(&S4[kk & 1][0][0], lane, badflag[kk & 1]);
    __syncthreads();
    if (kk < nt) {
      d4_t V;
#pragma unroll
      for (int r = 0; r < 4; ++r) V[r] = mine[(g + 4 * r) * 17 + c];
      bad = bad || (badflag[kk & 1][wave] != 0);
      switch (kk) {
        case 0: k5_block_step<0>(T, B, V, nt); break;
        case 1: k5_block_step<1>(T, B, V, nt); break;
        case 2: k5_block_step<2>(T, B, V, nt); break;
        default: k5_block_step<3>(T, B, V, nt); break;
      }
    }
  }
  if (bad && live) {
    if (lane == 0) {
      mean_out[p] = NaN;
      var_out[p] = NaN;
      status_out[p] = GSS_PT_SINGULAR;
    }
    live = false;
  }
  d4_t Gt = zero4;
#pragma unroll
  for (int K = 0; K < 4; ++K)
    if (K < nt) Gt = xty(B[K], B[K], Gt);
#pragma unroll
  for (int r = 0; r < 4; ++r) G[g + 4 * r][c] = Gt[r];
  __syncthreads();
  gram_finish(G, se, vg, sp, drift_dom, p, lane, live, mean_out, var_out, status_out);
}

// ---------------------------------------------------------------------------------------------
// 65 .. 256 neighbours on the MFMA tile core: ONE point per workgroup, W = NTMAX / 2 waves, the upper block triangle of
// the k x k system distributed over the waves' REGISTERS by tile columns -- wave w owns columns w and nt - 1 - w
// (j + 1 tiles and nt - j tiles: nt + 1 per wave, balanced) plus the right-hand-side tiles [c0 | z | F] of the same
// two tile rows.  Right-looking block steps, A = U'U as in the 64-neighbour kernel above:
//   1. the owner of column K factors the diagonal tile (potrf16_inv, 16 dependent column steps) and posts V = U_KK^-1;
//   2. every wave solves its own tiles of block row K, U_Kj = V'A_Kj and Y_K = V'B_K, and posts -U_Kj / Y_K in LDS
//      (the pivot row: at most NTMAX + 1 tiles, register images, conflict-free);
//   3. every wave updates its own columns, A_ij -= U_Ki'U_Kj (operand -U_Ki from the pivot row), B_i -= U_Ki'Y_K.
// Two workgroup barriers per step; nothing but the pivot row ever leaves the registers.  The Gram matrix Y'Y is summed
// over the waves and wave 0 finishes as gram_finish does for the small kernel.  k <= 128: 4 waves (3 workgroups per
// CU), k <= 192: 6 waves, k <= 256: 8 waves (one workgroup per CU).  Beyond 256 neighbours: krig_local_big_kernel.
// ---------------------------------------------------------------------------------------------
template <int NTMAX>
struct TilesLds {
  static constexpr int W = NTMAX / 2;
  static constexpr int KMAX = 16 * NTMAX;
  static constexpr int NX = 0;                            // KMAX x 3 neighbour coordinates
  static constexpr int PV = NX + KMAX * 3;                // pivot row: NTMAX tiles (-U_Kj at j) + Y_K; before step 0 the
  static constexpr int VS = PV + (NTMAX + 1) * 256;       //   staged right-hand-side columns (LMAX_RHS x KMAX <= 17 x 256 / ...)
  static constexpr int SS = VS + 256;                     // 16 x 17 scratch of the diagonal factorisation
  static constexpr int GG = SS + 272;                     // 16 x 17 Gram matrix
  static constexpr int GP = GG + 272;                     // W partial Gram tiles
  static constexpr int DOUBLES = GP + W * 256;
  static_assert(LMAX_RHS * KMAX <= (NTMAX + 1) * 256, "right-hand-side staging does not fit the pivot row");
};

template <int DIM, int NTMAX>
__global__ __launch_bounds__(32 * NTMAX, NTMAX == 8 ? 3 : 2) void krig_local_tiles_kernel(
    VgDev vg, LocalSpec sp, const double* __restrict__ xdata, const double* __restrict__ z,
    const double* __restrict__ drift_data, const double* __restrict__ x0, const double* __restrict__ drift_dom, int64_t m,
    int k, int minneighbors, const int* __restrict__ idx, const int* __restrict__ count, double* __restrict__ mean_out,
    double* __restrict__ var_out, uint8_t* __restrict__ status_out) {
  using L = TilesLds<NTMAX>;
  constexpr int W = L::W, KMAX = L::KMAX;
  extern __shared__ double tl_sm[];
  double* nx = tl_sm + L::NX;
  double* P = tl_sm + L::PV;
  double* Vs = tl_sm + L::VS;
  double* S = tl_sm + L::SS;
  double (*G)[17] = reinterpret_cast<double (*)[17]>(tl_sm + L::GG);
  double* Gp = tl_sm + L::GP;
  __shared__ signed char se[LMAX_NC][4];
  __shared__ int s_bad;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int g = lane >> 4, c = lane & 15;
  const int nc = sp.nc;
  const double NaN = __longlong_as_double(0x7ff8000000000000LL);
  const d4_t zero4 = {0.0, 0.0, 0.0, 0.0};
  if (tid == 0) {
#pragma unroll
    for (int cc = 0; cc < LMAX_NC; ++cc)
#pragma unroll
      for (int a = 0; a < 3; ++a) se[cc][a] = sp.e[cc][a];
  }
  for (int64_t p = blockIdx.x; p < m; p += gridDim.x) {
    const int cnt = __builtin_amdgcn_readfirstlane(count[p]);
    __syncthreads();   // the point before has been finished by every wave (LDS is re-used)
    if (cnt < minneighbors || cnt <= 0) {  // krig.jl:213-214
      if (tid == 0) {
        mean_out[p] = NaN;
        var_out[p] = NaN;
        status_out[p] = GSS_PT_MISSING;
      }
      continue;
    }
    const int nt = (cnt + 15) >> 4;
    double c0[DIM];
#pragma unroll
    for (int a = 0; a < DIM; ++a) c0[a] = x0[p * DIM + a];
    if (tid == 0) s_bad = 0;
    // neighbour coordinates and the right-hand-side columns (lane = neighbour), staged where the pivot row will live
    // (the exponents come from their LDS copy: a runtime-indexed read of the argument struct would go through scratch;
    // the barrier at the top of the loop has published them)
    double* rhs = P;
    for (int j = tid; j < 16 * nt; j += 64 * W) {
      const bool act = j < cnt;
      const int nj = act ? idx[p * k + j] : 0;
      double xj[DIM];
#pragma unroll
      for (int a = 0; a < DIM; ++a) {
        xj[a] = act ? xdata[(int64_t)nj * DIM + a] : 0.0;
        nx[j * 3 + a] = xj[a];
      }
      double zz = act ? z[nj] : 0.0;
      if (sp.variant == GSS_KRIG_SIMPLE) zz -= sp.sk_mean;
      rhs[0 * KMAX + j] = act ? cov_pair<DIM>(vg, xj, c0) : 0.0;
      rhs[1 * KMAX + j] = act ? zz : 0.0;
      for (int t = 0; t < nc; ++t) {
        double f = 1.0;
        if (sp.variant == GSS_KRIG_UNIVERSAL) {
#pragma unroll
          for (int a = 0; a < DIM; ++a) {
            const double u = (xj[a] - c0[a]) * sp.inv_scale;
            for (int q = 0; q < se[t][a]; ++q) f *= u;
          }
        } else if (sp.variant == GSS_KRIG_EXTDRIFT) {
          f = act ? drift_data[(int64_t)nj * nc + t] : 0.0;
        }
        rhs[(2 + t) * KMAX + j] = act ? f : 0.0;
      }
    }
    __syncthreads();
    // ownership: the columns are paired from the two ends, (0, nt - 1), (1, nt - 2), ...: wave w takes the pair
    // (cA, cB) = (w, nt - 1 - w) while cA <= cB (the middle column of an odd count stands alone as cA; waves beyond
    // the pairs own nothing and only keep the barriers company)
    const int cA = wave;
    const int cB = nt - 1 - wave;
    const bool hasA = cA <= cB;
    const bool hasB = cB > cA;
    auto build = [&](int i, int col) -> d4_t {   // tile (i, col) of the covariance matrix, identity beyond cnt
      double xr[4][DIM], xc[DIM], v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int a = 0; a < DIM; ++a) xr[r][a] = nx[(16 * i + g + 4 * r) * 3 + a];
      const int cj = 16 * col + c;
#pragma unroll
      for (int a = 0; a < DIM; ++a) xc[a] = nx[cj * 3 + a];
      cov_pair4<DIM>(vg, xr, xc, v);
      d4_t t;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * i + g + 4 * r;
        t[r] = (row < cnt && cj < cnt) ? v[r] : (row == cj ? 1.0 : 0.0);
      }
      return t;
    };
    auto build_rhs = [&](int i) -> d4_t {        // right-hand-side tile of block row i: 16 columns, zero beyond 2 + nc
      d4_t t;
      const bool used = c < 2 + nc;
      const double* col = rhs + (used ? c : 0) * KMAX;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double v = col[16 * i + g + 4 * r];
        t[r] = used ? v : 0.0;
      }
      return t;
    };
    d4_t A[W], Bc[NTMAX], RA = zero4, RB = zero4;
#pragma unroll
    for (int i = 0; i < W; ++i) A[i] = (hasA && i <= cA) ? build(i, cA) : zero4;
#pragma unroll
    for (int i = 0; i < NTMAX; ++i) Bc[i] = (hasB && i <= cB) ? build(i, cB) : zero4;
    if (hasA) RA = build_rhs(cA);
    if (hasB) RB = build_rhs(cB);
    __syncthreads();   // the staged columns have been read: the pivot row may take their place

    static_for<0, NTMAX>([&](auto KK) {
      constexpr int K = decltype(KK)::value;
      if (K < nt) {
        const bool ownA = hasA && cA == K, ownB = hasB && cB == K;
        if (ownA || ownB) {
          d4_t t = Bc[K];
          if constexpr (K < W) {
            if (ownA) t = A[K];
          }
          d4_t v;
          int badc;
          potrf16_inv<true>(t, S, lane, &v, &badc);
          tile_store(Vs, v, lane);
          if (badc >= 0 && lane == 0) s_bad = 1;
        }
        __syncthreads();
        const d4_t V = tile_load(Vs, lane);
        if constexpr (K < W) {
          if (hasA && cA > K) {
            A[K] = xty(V, A[K], zero4);
            tile_store(P + cA * 256, -A[K], lane);
          }
          if (ownA) {
            RA = xty(V, RA, zero4);
            tile_store(P + NTMAX * 256, RA, lane);
          }
        }
        if (hasB && cB > K) {
          Bc[K] = xty(V, Bc[K], zero4);
          tile_store(P + cB * 256, -Bc[K], lane);
        }
        if (ownB) {
          RB = xty(V, RB, zero4);
          tile_store(P + NTMAX * 256, RB, lane);
        }
        __syncthreads();
        if constexpr (K < W) {
          if (hasA && cA > K) {
            const d4_t U = A[K];
            static_for<K + 1, W>([&](auto II) {
              constexpr int i = decltype(II)::value;
              if (i <= cA) A[i] = xty(tile_load(P + i * 256, lane), U, A[i]);
            });
            RA = xty(tile_load(P + cA * 256, lane), tile_load(P + NTMAX * 256, lane), RA);
          }
        }
        if (hasB && cB > K) {
          const d4_t U = Bc[K];
          static_for<K + 1, NTMAX>([&](auto II) {
            constexpr int i = decltype(II)::value;
            if (i <= cB) Bc[i] = xty(tile_load(P + i * 256, lane), U, Bc[i]);
          });
          RB = xty(tile_load(P + cB * 256, lane), tile_load(P + NTMAX * 256, lane), RB);
        }
      }
    });
    // Gram matrix of the forward-substituted right-hand sides, summed over the waves
    d4_t Gt = zero4;
    if (hasA) Gt = xty(RA, RA, Gt);
    if (hasB) Gt = xty(RB, RB, Gt);
    tile_store(Gp + wave * 256, Gt, lane);
    __syncthreads();
    if (wave == 0) {
      d4_t sum = zero4;
#pragma unroll
      for (int w = 0; w < W; ++w) sum += tile_load(Gp + w * 256, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) G[g + 4 * r][c] = sum[r];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (s_bad) {
        if (lane == 0) {
          mean_out[p] = NaN;
          var_out[p] = NaN;
          status_out[p] = GSS_PT_SINGULAR;
        }
      } else {
        gram_finish(G, se, vg, sp, drift_dom, p, lane, true, mean_out, var_out, status_out);
      }
    }
  }
}

// Host driver: chunks the domain so that the neighbour-index scratch stays small, runs K4 then K5.
int32_t krig_local_dev(const VgDev& vg, int variant, int nc, int dim, const signed char* exps, double inv_scale,
                       double sk_mean, const double* xdata, const double* z, const double* drift_data, int64_t n,
                       const double* x0, const double* drift_dom, int64_t m, int k, int minneighbors, double radius,
                       const double* inv_radii_host, double* mean, double* var, uint8_t* status, int* idx_out,
                       int* count_out, hipStream_t s, int metric, HostPipe* pipe) {
  GSS_REQUIRE(nc <= LMAX_NC, "moving-neighbourhood kriging supports at most %d drift terms (got %d)", LMAX_NC, nc);
  GSS_REQUIRE(k >= 1 && k <= BIG_MAX_K, "maxneighbors = %d: moving neighbourhoods hold at most %d neighbours "
                                        "(use the global neighbourhood beyond that)", k, BIG_MAX_K);
  const bool big = k > LMAX_K;   // 65 .. 4096 neighbours: passes of the search, one workgroup per point
  LocalSpec sp;
  std::memset(&sp, 0, sizeof(sp));
  sp.variant = variant;
  sp.nc = nc;
  sp.dim = dim;
  for (int c = 0; c < nc; ++c)
    for (int a = 0; a < 3; ++a) sp.e[c][a] = exps ? exps[3 * c + a] : 0;
  sp.inv_scale = inv_scale;
  sp.sk_mean = sk_mean;

  const bool piped = pipe && pipe->on;   // host arrays arrive and leave piece by piece (gss_internal.h)
  const int64_t chunk = big ? (k > 512 ? (1 << 14) : (1 << 17)) : (piped ? HostPipe::PIECE : (1 << 20));
  KnnIndex ix;  // k-d ordered batches + boxes, built once per call
  const char* brute = std::getenv("GSS_KNN_BRUTE");
  // (the haversine distance always searches exhaustively, in passes of 64 beyond 64 neighbours)
  const bool use_index = metric != GSS_METRIC_HAVERSINE && (big || !(brute && brute[0] == '1'));
  if (use_index) GSS_TRY(knn_index_build_from_device(xdata, n, dim, &ix, s));
  DevBuf idx_s, cnt_s, st_s;
  if (!idx_out) GSS_TRY(idx_s.alloc(sizeof(int) * (size_t)((m < chunk ? m : chunk) * k)));
  if (!count_out) GSS_TRY(cnt_s.alloc(sizeof(int) * (size_t)(m < chunk ? m : chunk)));
  if (!status) GSS_TRY(st_s.alloc((size_t)(m < chunk ? m : chunk)));
  for (int64_t off = 0; off < m; off += chunk) {
    const int64_t mv = (m - off) < chunk ? (m - off) : chunk;
    int* idx = idx_out ? idx_out + off * k : idx_s.as<int>();
    int* cnt = count_out ? count_out + off : cnt_s.as<int>();
    uint8_t* st = status ? status + off : st_s.as<uint8_t>();
    if (piped) GSS_TRY(pipe->fetch(off, mv, s));
    {
      ProfScope ps("knn", s);
      if (use_index)
        GSS_TRY(knn_search_indexed_any(ix, xdata, x0 + off * dim, mv, k, radius, inv_radii_host, idx, cnt, s, metric));
      else GSS_TRY(knn_search_dev(xdata, n, dim, x0 + off * dim, mv, k, radius, inv_radii_host, idx, cnt, s, metric));
    }
    const double* dd = drift_dom ? drift_dom + off * nc : nullptr;
    ProfScope pl("krig_local", s);
    if (big && k <= 256) {
      // 65 .. 256 neighbours: the register-distributed tile kernel, one point per workgroup
      const int ntmax = k <= 128 ? 8 : (k <= 192 ? 12 : 16);
      const int per_cu = ntmax == 8 ? 3 : 1;
      int64_t blocks = (int64_t)256 * per_cu * 2;   // two rounds of resident workgroups, points handed out by stride
      if (blocks > mv) blocks = mv;
#define GSS_TILES_LAUNCH(D, NT)                                                                                       \
  do {                                                                                                                \
    const size_t lds = sizeof(double) * (size_t)TilesLds<NT>::DOUBLES;                                                \
    GSS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(krig_local_tiles_kernel<D, NT>),                        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                               \
    hipLaunchKernelGGL((krig_local_tiles_kernel<D, NT>), dim3((unsigned)blocks), dim3(32 * NT), lds, s, vg, sp,       \
                       xdata, z, drift_data, x0 + off * dim, dd, mv, k, minneighbors, idx, cnt, mean + off,          \
                       var + off, st);                                                                                \
  } while (0)
#define GSS_TILES_DIM(D)                                \
  do {                                                  \
    if (ntmax == 8) GSS_TILES_LAUNCH(D, 8);             \
    else if (ntmax == 12) GSS_TILES_LAUNCH(D, 12);      \
    else GSS_TILES_LAUNCH(D, 16);                       \
  } while (0)
      switch (dim) {
        case 1: GSS_TILES_DIM(1); break;
        case 2: GSS_TILES_DIM(2); break;
        default: GSS_TILES_DIM(3); break;
      }
#undef GSS_TILES_DIM
#undef GSS_TILES_LAUNCH
      GSS_HIP(hipGetLastError());
      if (piped) GSS_TRY(pipe->deliver(off, mv, s));
      continue;
    }
    if (big) {
      const int64_t rows = (int64_t)k + 2 + nc;
      const int64_t need = (int64_t)k * (k + 1) / 2 + (int64_t)(2 + nc) * k + k;
      const bool in_lds = need <= BIG_LDS_DOUBLES;
      const size_t lds = in_lds ? sizeof(double) * (size_t)BIG_LDS_DOUBLES : 0;
      // in LDS one workgroup per CU is resident; through HBM slabs as many as 8 GiB of slabs allow (>= 64)
      int64_t blocks = in_lds ? 256 * 4 : (int64_t)((size_t)8 << 30) / (int64_t)(sizeof(double) * (size_t)need);
      if (blocks > 2048) blocks = 2048;
      if (blocks < 64) blocks = 64;
      if (blocks > mv) blocks = mv;
      DevBuf slab;
      if (!in_lds) GSS_TRY(slab.alloc(sizeof(double) * (size_t)(need * blocks)));
      (void)rows;
#define GSS_BIG_LAUNCH(D)                                                                                             \
  do {                                                                                                                \
    if (in_lds)                                                                                                       \
      GSS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(krig_local_big_kernel<D>),                            \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                            \
    hipLaunchKernelGGL((krig_local_big_kernel<D>), dim3((unsigned)blocks), dim3(BIG_NT), lds, s, vg, sp, xdata, z,    \
                       drift_data, x0 + off * dim, dd, mv, k, minneighbors, idx, cnt, mean + off, var + off, st,     \
                       slab.as<double>(), need, in_lds ? 1 : 0);                                                      \
  } while (0)
      switch (dim) {
        case 1: GSS_BIG_LAUNCH(1); break;
        case 2: GSS_BIG_LAUNCH(2); break;
        default: GSS_BIG_LAUNCH(3); break;
      }
#undef GSS_BIG_LAUNCH
      GSS_HIP(hipGetLastError());
      if (piped) GSS_TRY(pipe->deliver(off, mv, s));
      GSS_HIP(hipStreamSynchronize(s));   // the slab is released at the end of this iteration
      continue;
    }
    {
#define GSS_K5_ARGS vg, sp, xdata, z, drift_data, x0 + off * dim, dd, mv, k, minneighbors, idx, cnt, mean + off, \
                    var + off, st
#define GSS_K5_LAUNCH(D, K) \
  hipLaunchKernelGGL((krig_local_mfma_kernel<D, K>), dim3((unsigned)((mv + K5_WAVES - 1) / K5_WAVES)), \
                     dim3(64 * K5_WAVES), 0, s, GSS_K5_ARGS)
      // one instantiation per common single-structure model in 2-D / 3-D, the general kernel otherwise
      const int kind = vg.nextra == 0 ? vg.kind : -1;
      if (dim == 3) {
        switch (kind) {
          case GSS_VG_GAUSSIAN: GSS_K5_LAUNCH(3, GSS_VG_GAUSSIAN); break;
          case GSS_VG_EXPONENTIAL: GSS_K5_LAUNCH(3, GSS_VG_EXPONENTIAL); break;
          case GSS_VG_SPHERICAL: GSS_K5_LAUNCH(3, GSS_VG_SPHERICAL); break;
          case VG_MATERN32: GSS_K5_LAUNCH(3, VG_MATERN32); break;
          case VG_MATERN52: GSS_K5_LAUNCH(3, VG_MATERN52); break;
          default: GSS_K5_LAUNCH(3, -1); break;
        }
      } else if (dim == 2) {
        switch (kind) {
          case GSS_VG_GAUSSIAN: GSS_K5_LAUNCH(2, GSS_VG_GAUSSIAN); break;
          case GSS_VG_EXPONENTIAL: GSS_K5_LAUNCH(2, GSS_VG_EXPONENTIAL); break;
          case GSS_VG_SPHERICAL: GSS_K5_LAUNCH(2, GSS_VG_SPHERICAL); break;
          case VG_MATERN32: GSS_K5_LAUNCH(2, VG_MATERN32); break;
          case VG_MATERN52: GSS_K5_LAUNCH(2, VG_MATERN52); break;
          default: GSS_K5_LAUNCH(2, -1); break;
        }
      } else {
        GSS_K5_LAUNCH(1, -1);
      }
#undef GSS_K5_LAUNCH
#undef GSS_K5_ARGS
      GSS_HIP(hipGetLastError());
      if (piped) GSS_TRY(pipe->deliver(off, mv, s));
      continue;
    }
  }
  if (piped) GSS_TRY(pipe->finish(s));
  GSS_HIP(hipStreamSynchronize(s));  // scratch and the search index are released on return
  return GSS_OK;
}

}  // namespace gss
