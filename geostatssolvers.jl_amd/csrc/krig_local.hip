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
#include "krig_local.h"
#include "tile16.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace gss {

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
      M[ntri + j] = c0_entry<DIM, -1>(vg, sp, xj, c0, c0);
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
        const double v = sp.c00 - qf + rsr;
        mean_out[p] = mu;
        var_out[p] = v > 0.0 ? v : 0.0;
        status_out[p] = GSS_PT_OK;
      }
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
template <int KK, int NT = 4>
__device__ __forceinline__ void k5_block_step(d4_t (&T)[10], d4_t (&B)[4], const d4_t& V, int nt) {
  const d4_t zero4 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int j = KK + 1; j < NT; ++j)
    if (j < nt) T[tile_id(KK, j)] = xty(V, T[tile_id(KK, j)], zero4);
  B[KK] = xty(V, B[KK], zero4);
#pragma unroll
  for (int i = KK + 1; i < NT; ++i) {
    if (i < nt) {
      const d4_t N = -T[tile_id(KK, i)];
#pragma unroll
      for (int j = i; j < NT; ++j)
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

// NT = tiles of 16 neighbours the instantiation holds (1, 2 or 4: maxneighbors <= 16 NT).  The common moving
// neighbourhoods are small -- 8 ... 32 neighbours --, and an instantiation that carries one or three tiles instead
// of ten needs a third of the registers: five or four workgroups per CU instead of three hide the latencies of the
// gathers and of the one factorisation such a point needs.
template <int DIM, int KIND, int NT>
__global__ __launch_bounds__(64 * K5_WAVES) __attribute__((amdgpu_waves_per_eu(NT == 1 ? 4 : 3, NT == 1 ? 5 : (NT == 2 ? 4 : 3))))
void krig_local_mfma_kernel(VgDev vg, LocalSpec sp, const double* __restrict__ xdata,
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
  // (single-structure instantiations: the coordinates carry the radii of the model's ball AND the model's own scale, so
  //  that the scaled distance is the argument of the shape -- kpos_scale, gss_internal.h)
  constexpr bool UNIT = KIND >= 0;
  double c0[DIM], c0s[DIM], sca[DIM];
#pragma unroll
  for (int a = 0; a < DIM; ++a) {
    sca[a] = UNIT ? (vg.aniso ? vg.ir[a] : 1.0) * kpos_scale<(KIND < 0 ? 0 : KIND)>(vg) : 1.0;
    c0[a] = x0[p * DIM + a];
    c0s[a] = UNIT ? mul_rounded(c0[a], sca[a]) : c0[a];   // (rounded products: a contraction with the difference below
                                                          //  would make a coincident sample's distance nonzero)
  }
  {
    const bool act = lane < cnt;
    const int nj = act ? idx[p * k + lane] : 0;
    double xj[DIM], xjs[DIM];
#pragma unroll
    for (int a = 0; a < DIM; ++a) {
      xj[a] = act ? xdata[(int64_t)nj * DIM + a] : 0.0;
      xjs[a] = UNIT ? mul_rounded(xj[a], sca[a]) : xj[a];
      nxs[lane][a] = xjs[a];
    }
    double zz = act ? z[nj] : 0.0;
    if (sp.variant == GSS_KRIG_SIMPLE) zz -= sp.sk_mean;
    rhs_col(0)[lane] = act ? c0_entry<DIM, KIND, UNIT>(vg, sp, xjs, c0, c0s, sca) : 0.0;
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

  // system matrix, upper block triangle, tile layout; the four rows a lane holds of a tile are evaluated together.
  // Rows and columns beyond the neighbour count are padded with the identity; only the last tile row / column of a
  // point can hold such entries, and none at all when the count is a multiple of 16 (the full neighbourhoods of a
  // dense data set): the selects are behind a scalar branch.
  const bool ragged = (cnt & 15) != 0;
  d4_t T[10];
  // Diagonal tiles in PAIRS (round 4).  A diagonal tile is symmetric, and all the factorisation ever reads of it is one
  // triangle: the hand-over below gives potrf16_inverse_x4 the lower triangle, which is all its lanes use.  So the
  // registers of one tile evaluation serve two diagonal tiles: positions (a, b) with a <= b hold tile 2q's entry (a, b),
  // positions a > b hold tile 2q + 1's entry (a, b) -- both tiles keep a copy, tile 2q is read through its upper
  // triangle (handed over transposed), tile 2q + 1 through its lower triangle with its diagonal (the sill) put back.
  // The trailing updates add symmetric matrices, so each copy's live triangle stays that tile's data.  Eight tile
  // evaluations per point instead of ten at 64 neighbours (2 144 of the 2 560 covariances were ever needed).
  if constexpr (NT >= 2) {
#pragma unroll
    for (int Q = 0; Q < NT / 2; ++Q) {
      if (2 * Q < nt) {
        double xr[4][DIM], xc[4][DIM], v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int a = g + 4 * r;
          const int blk = 16 * (2 * Q + (a > c ? 1 : 0));
#pragma unroll
          for (int d = 0; d < DIM; ++d) {
            xr[r][d] = nxs[blk + a][d];
            xc[r][d] = nxs[blk + c][d];
          }
        }
        cov_pairs4_k<DIM, KIND, UNIT>(vg, xr, xc, v);
        d4_t t0, t1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          t0[r] = v[r];
          t1[r] = (g + 4 * r == c) ? vg.sill : v[r];
        }
        if (ragged && 2 * Q + 2 >= nt) {   // one of the two is the point's last tile (or lies beyond it)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int a = g + 4 * r;
            const double pad = a == c ? 1.0 : 0.0;
            if (!(32 * Q + a < cnt && 32 * Q + c < cnt)) t0[r] = pad;
            if (!(32 * Q + 16 + a < cnt && 32 * Q + 16 + c < cnt)) t1[r] = pad;
          }
        }
        T[tile_id(2 * Q, 2 * Q)] = t0;
        if (2 * Q + 1 < nt) T[tile_id(2 * Q + 1, 2 * Q + 1)] = t1;
      }
    }
  }
#pragma unroll
  for (int I = 0; I < NT; ++I) {
    if (I < nt) {
      double xr[4][DIM];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int a = 0; a < DIM; ++a) xr[r][a] = nxs[16 * I + g + 4 * r][a];
#pragma unroll
      for (int J = (NT >= 2 ? I + 1 : I); J < NT; ++J) {
        if (J < nt) {
          const int col = 16 * J + c;
          double xcol[DIM], v[4];
#pragma unroll
          for (int a = 0; a < DIM; ++a) xcol[a] = nxs[col][a];
          cov_pair4_k<DIM, KIND, UNIT>(vg, xr, xcol, v);
#pragma unroll
          for (int r = 0; r < 4; ++r) T[tile_id(I, J)][r] = v[r];
          if (ragged && J == nt - 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = 16 * I + g + 4 * r;
              if (!(row < cnt && col < cnt)) T[tile_id(I, J)][r] = row == col ? 1.0 : 0.0;
            }
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
    for (int K = 0; K < NT; ++K) {
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
  // block steps beyond ceil(maxneighbors / 16) have nothing to do for any point of the launch: skipped by the whole
  // grid alike (a scalar branch; the barriers stay matched).  With 16 neighbours or fewer -- the common case of a
  // moving neighbourhood -- that leaves one diagonal factorisation instead of four: 5.6 -> 2.x ms per 1.25e6 points.
  const int ntk = (k + 15) >> 4;
#pragma unroll
  for (int kk = 0; kk < NT; ++kk) {
    if (kk >= ntk) break;
    double* mine = S4[kk & 1][wave];
    // (the factorisation reads the lower triangle: even tiles of a pair carry their data in the upper one and are
    //  handed over transposed)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double e = (kk < nt) ? T[tile_id(kk, kk)][r] : ((g + 4 * r) == c ? 1.0 : 0.0);
      if (NT >= 2 && (kk & 1) == 0) mine[c * 17 + (g + 4 * r)] = e;
      else mine[(g + 4 * r) * 17 + c] = e;
    }
    __syncthreads();
    if (wave == kk) potrf16_inverse_x4(&S4[kk & 1][0][0], lane, badflag[kk & 1]);
    __syncthreads();
    if (kk < nt) {
      d4_t V;
#pragma unroll
      for (int r = 0; r < 4; ++r) V[r] = mine[(g + 4 * r) * 17 + c];
      bad = bad || (badflag[kk & 1][wave] != 0);
      switch (kk) {
        case 0: k5_block_step<0, NT>(T, B, V, nt); break;
        case 1: k5_block_step<(1 < NT ? 1 : 0), NT>(T, B, V, nt); break;
        case 2: k5_block_step<(2 < NT ? 2 : 0), NT>(T, B, V, nt); break;
        default: k5_block_step<(3 < NT ? 3 : 0), NT>(T, B, V, nt); break;
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
  for (int K = 0; K < NT; ++K)
    if (K < nt) Gt = xty(B[K], B[K], Gt);
#pragma unroll
  for (int r = 0; r < 4; ++r) G[g + 4 * r][c] = Gt[r];
  __syncthreads();
  gram_finish(G, se, vg, sp, drift_dom, p, lane, live, mean_out, var_out, status_out);
}

// Host driver: chunks the domain so that the neighbour-index scratch stays small, runs K4 then K5.
int32_t krig_local_dev(const VgDev& vg, int variant, int nc, int dim, const signed char* exps, double inv_scale,
                       double sk_mean, const double* xdata, const double* z, const double* drift_data, int64_t n,
                       const double* x0, const double* drift_dom, int64_t m, int k, int minneighbors, double radius,
                       const double* inv_radii_host, double* mean, double* var, uint8_t* status, int* idx_out,
                       int* count_out, hipStream_t s, int metric, HostPipe* pipe, int block_nsub,
                       const double* block_cell, double block_cvv) {
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
  sp.bsub = block_nsub > 0 ? block_nsub : 0;
  for (int a = 0; a < 3; ++a) sp.bcell[a] = (block_nsub > 0 && block_cell) ? block_cell[a] : 0.0;
  sp.c00 = block_nsub > 0 ? block_cvv : vg.sill;

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
      // 65 .. 256 neighbours: the register-distributed tile kernel, one point per workgroup (krig_tiles.hip)
      GSS_TRY(krig_local_tiles_launch(vg, sp, dim, xdata, z, drift_data, x0 + off * dim, dd, mv, k, minneighbors, idx,
                                      cnt, mean + off, var + off, st, s));
      GSS_HIP(hipGetLastError());
      if (piped) GSS_TRY(pipe->deliver(off, mv, s));
      continue;
    }
    if (big && k <= 768 && !std::getenv("GSS_KRIG_SLAB_OFF")) {
      // 257 .. 768 neighbours: the tile algorithm with the triangle in a per-workgroup slab (krig_slab.hip)
      GSS_TRY(krig_local_slab_launch(vg, sp, dim, xdata, z, drift_data, x0 + off * dim, dd, mv, k, minneighbors, idx, cnt,
                                     mean + off, var + off, st, s));
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
#define GSS_K5_LAUNCH_NT(D, K, NTV) \
  hipLaunchKernelGGL((krig_local_mfma_kernel<D, K, NTV>), dim3((unsigned)((mv + K5_WAVES - 1) / K5_WAVES)), \
                     dim3(64 * K5_WAVES), 0, s, GSS_K5_ARGS)
#define GSS_K5_LAUNCH(D, K)                        \
  do {                                             \
    if (k <= 16) GSS_K5_LAUNCH_NT(D, K, 1);        \
    else if (k <= 32) GSS_K5_LAUNCH_NT(D, K, 2);   \
    else GSS_K5_LAUNCH_NT(D, K, 4);                \
  } while (0)
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
#undef GSS_K5_LAUNCH_NT
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
