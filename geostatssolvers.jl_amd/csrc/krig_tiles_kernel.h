// Kernel template of the 65 .. 256-neighbour path; instantiated per variogram model in krig_tiles_*.hip (one translation
// unit per model: the fully unrolled block steps compile slowly, the units build side by side).
#pragma once

#include "gss_internal.h"
#include "krig_local.h"
#include "tile16.h"

namespace gss {

// ---------------------------------------------------------------------------------------------
// 65 .. 256 neighbours on the MFMA tile core: ONE point per workgroup, W = NTMAX / 2 waves, the upper block triangle of
// the k x k system distributed over the waves' REGISTERS by tile columns -- wave w owns columns w and nt - 1 - w
// (j + 1 tiles and nt - j tiles: nt + 1 per wave, balanced, in one array of NTMAX + 1 register tiles) plus the
// right-hand-side tiles [c0 | z | F] of the same two tile rows.  Right-looking block steps, A = U'U as in the 64-neighbour kernel above:
//   1. the owner of column K factors the diagonal tile (potrf16_inv, 16 dependent column steps) and posts V = U_KK^-1;
//   2. every wave solves its own tiles of block row K, U_Kj = V'A_Kj and Y_K = V'B_K, and posts -U_Kj / Y_K in LDS
//      (the pivot row: at most NTMAX + 1 tiles, register images, conflict-free);
//   3. every wave updates its own columns, A_ij -= U_Ki'U_Kj (operand -U_Ki from the pivot row), B_i -= U_Ki'Y_K.
// Two workgroup barriers per step; nothing but the pivot row ever leaves the registers.  The Gram matrix Y'Y is summed
// over the waves and wave 0 finishes as gram_finish does for the small kernel.  k <= 96: 3 waves (round 4: with six tile
// columns the fourth wave of the 128-neighbour instantiation owns nothing; 192 threads make four workgroups per CU),
// k <= 128: 4 waves (3 workgroups per CU), k <= 256: 8 waves (one workgroup per CU).  Beyond 256: krig_local_big_kernel.
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

template <int DIM, int KIND, int NTMAX>
__global__ __launch_bounds__(32 * NTMAX, NTMAX <= 8 ? 3 : 2) void krig_local_tiles_kernel(
    VgDev vg, LocalSpec sp, const double* __restrict__ xdata, const double* __restrict__ z,
    const double* __restrict__ drift_data, const double* __restrict__ x0, const double* __restrict__ drift_dom, int64_t m,
    int k, int minneighbors, const int* __restrict__ idx, const int* __restrict__ count, double* __restrict__ mean_out,
    double* __restrict__ var_out, uint8_t* __restrict__ status_out) {
#ifdef GSS_HOST_SANITIZER_BUILD
  // (the host-sanitizer build only exercises the host side; the fully unrolled device code would dominate its build time)
#else
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
    // KIND >= 0 (one structure fixed at compile time): coordinates are divided by the radii of the model's ball once,
    // as in the 64-neighbour kernel, and every pair is three differences and a fused sum of squares
    // (and, round 4, multiplied by the model's own scale -- kpos_scale -- so that the scaled distance is the argument of
    //  the shape; products rounded once and never contracted with the difference: a coincident sample stays at distance 0)
    constexpr bool UNIT = KIND >= 0;
    double c0[DIM], c0s[DIM], sca[DIM];
#pragma unroll
    for (int a = 0; a < DIM; ++a) {
      sca[a] = UNIT ? (vg.aniso ? vg.ir[a] : 1.0) * kpos_scale<(KIND < 0 ? 0 : KIND)>(vg) : 1.0;
      c0[a] = x0[p * DIM + a];
      c0s[a] = UNIT ? mul_rounded(c0[a], sca[a]) : c0[a];
    }
    if (tid == 0) s_bad = 0;
    // neighbour coordinates and the right-hand-side columns (lane = neighbour), staged where the pivot row will live
    // (the exponents come from their LDS copy: a runtime-indexed read of the argument struct would go through scratch;
    // the barrier at the top of the loop has published them)
    double* rhs = P;
    for (int j = tid; j < 16 * nt; j += 64 * W) {
      const bool act = j < cnt;
      const int nj = act ? idx[p * k + j] : 0;
      double xj[DIM], xjs[DIM];
#pragma unroll
      for (int a = 0; a < DIM; ++a) {
        xj[a] = act ? xdata[(int64_t)nj * DIM + a] : 0.0;
        xjs[a] = UNIT ? mul_rounded(xj[a], sca[a]) : xj[a];
        nx[j * 3 + a] = xjs[a];
      }
      double zz = act ? z[nj] : 0.0;
      if (sp.variant == GSS_KRIG_SIMPLE) zz -= sp.sk_mean;
      rhs[0 * KMAX + j] = act ? c0_entry<DIM, KIND, UNIT>(vg, sp, xjs, c0, c0s, sca) : 0.0;
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
    // Ownership: the tile columns are paired from the two ends, (0, nt - 1), (1, nt - 2), ...: wave w takes the pair
    // (cA, cB) = (w, nt - 1 - w) while cA <= cB (the middle column of an odd count stands alone as cA; waves beyond
    // the pairs own nothing and only keep the barriers company).  A pair holds (cA + 1) + (cB + 1) = nt + 1 tiles in
    // ONE register array of NTMAX + 1 slots: slot s <= cA is tile (s, cA), slot s > cA is tile (s - cA - 1, cB).  The
    // slot index is a compile-time constant everywhere (registers cannot be indexed at run time); which tile a slot
    // holds is wave-uniform run-time data, so every test below is a scalar branch.
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int cA = wv;
    const int cB = nt - 1 - wv;
    const bool hasA = cA <= cB;
    const bool hasB = cB > cA;
    auto slot_row = [&](int sl) { return sl <= cA ? sl : sl - cA - 1; };
    auto slot_col = [&](int sl) { return sl <= cA ? cA : cB; };
    auto slot_used = [&](int sl) { return sl <= cA ? hasA : (hasB && sl <= nt); };
    auto build = [&](int i, int col) -> d4_t {   // tile (i, col) of the covariance matrix, identity beyond cnt
      double xr[4][DIM], xc[DIM], v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int a = 0; a < DIM; ++a) xr[r][a] = nx[(16 * i + g + 4 * r) * 3 + a];
      const int cj = 16 * col + c;
#pragma unroll
      for (int a = 0; a < DIM; ++a) xc[a] = nx[cj * 3 + a];
      cov_pair4_k<DIM, KIND, UNIT>(vg, xr, xc, v);
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
    d4_t T[NTMAX + 1], RA = zero4, RB = zero4;
    // (one tile at a time: interleaving the covariance chains of several tiles costs more registers than it hides)
#pragma unroll
    for (int sl = 0; sl <= NTMAX; ++sl) {
      T[sl] = zero4;
      if (slot_used(sl)) T[sl] = build(slot_row(sl), slot_col(sl));
      __builtin_amdgcn_sched_barrier(0);
    }
    if (hasA) RA = build_rhs(cA);
    if (hasB) RB = build_rhs(cB);
    __syncthreads();   // the staged columns have been read: the pivot row may take their place

    static_for<0, NTMAX>([&](auto KK) {
      constexpr int K = decltype(KK)::value;
      if (K < nt) {
        // 1. the diagonal tile: the wave that owns column K (as cA or as cB) factors it and posts V = U_KK^-1
        const bool ownA = hasA && cA == K, ownB = hasB && cB == K;
        if (ownA || ownB) {
          d4_t t = zero4;
          static_for<K, NTMAX + 1>([&](auto SS) {     // (a slot below K holds a row below K)
            constexpr int sl = decltype(SS)::value;
            if (slot_used(sl) && slot_row(sl) == K && slot_col(sl) == K) t = T[sl];
          });
          d4_t v;
          int badc;
          potrf16_inv<true>(t, S, lane, &v, &badc);
          tile_store(Vs, v, lane);
          if (badc >= 0 && lane == 0) s_bad = 1;
        }
        __syncthreads();
        // 2. block row K of the wave's own columns: U_Kj = V'A_Kj, kept (UA / UB) and posted negated; Y_K = V'B_K
        const d4_t V = tile_load(Vs, lane);
        d4_t UA = zero4, UB = zero4;
        static_for<K, NTMAX + 1>([&](auto SS) {
          constexpr int sl = decltype(SS)::value;
          if (slot_used(sl) && slot_row(sl) == K && slot_col(sl) > K) {
            const d4_t U = xty(V, T[sl], zero4);
            T[sl] = U;
            tile_store(P + slot_col(sl) * 256, -U, lane);
            if (sl <= cA) UA = U;
            else UB = U;
          }
        });
        if (ownA) {
          RA = xty(V, RA, zero4);
          tile_store(P + NTMAX * 256, RA, lane);
        }
        if (ownB) {
          RB = xty(V, RB, zero4);
          tile_store(P + NTMAX * 256, RB, lane);
        }
        __syncthreads();
        // 3. trailing update of the wave's own tiles below block row K and of its right-hand-side tiles
        static_for<K + 1, NTMAX + 1>([&](auto SS) {
          constexpr int sl = decltype(SS)::value;
          if (slot_used(sl) && slot_row(sl) > K) {
            const d4_t X = tile_load(P + slot_row(sl) * 256, lane);   // -U_Ki
            if (sl <= cA) T[sl] = xty(X, UA, T[sl]);
            else T[sl] = xty(X, UB, T[sl]);
          }
          __builtin_amdgcn_sched_barrier(0);   // (hoisting every operand load to the top costs 8 registers a tile)
        });
        if ((hasA && cA > K) || (hasB && cB > K)) {
          const d4_t Y = tile_load(P + NTMAX * 256, lane);
          if (hasA && cA > K) RA = xty(tile_load(P + cA * 256, lane), Y, RA);
          if (hasB && cB > K) RB = xty(tile_load(P + cB * 256, lane), Y, RB);
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
#endif
}

template <int DIM, int KIND>
static int32_t tiles_launch_nt(int ntmax, int64_t blocks, hipStream_t s, const VgDev& vg, const LocalSpec& sp,
                               const double* xdata, const double* z, const double* drift_data, const double* x0,
                               const double* drift_dom, int64_t m, int k, int minneighbors, const int* idx,
                               const int* count, double* mean, double* var, uint8_t* status) {
#define GSS_TILES_LAUNCH(NT)                                                                                          \
  do {                                                                                                                \
    const size_t lds = sizeof(double) * (size_t)TilesLds<NT>::DOUBLES;                                                \
    GSS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(krig_local_tiles_kernel<DIM, KIND, NT>),                \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                               \
    hipLaunchKernelGGL((krig_local_tiles_kernel<DIM, KIND, NT>), dim3((unsigned)blocks), dim3(32 * NT), lds, s, vg,   \
                       sp, xdata, z, drift_data, x0, drift_dom, m, k, minneighbors, idx, count, mean, var, status);   \
  } while (0)
  if (ntmax == 6) GSS_TILES_LAUNCH(6);
  else if (ntmax == 8) GSS_TILES_LAUNCH(8);
  else GSS_TILES_LAUNCH(16);
#undef GSS_TILES_LAUNCH
  GSS_HIP(hipGetLastError());
  return GSS_OK;
}


// arguments of one launch, passed through the per-model entry points
struct TilesArgs {
  int ntmax;
  int64_t blocks;
  hipStream_t s;
  const VgDev* vg;
  const LocalSpec* sp;
  const double *xdata, *z, *drift_data, *x0, *drift_dom;
  int64_t m;
  int k, minneighbors;
  const int *idx, *count;
  double *mean, *var;
  uint8_t* status;
};
template <int DIM, int KIND>
static int32_t tiles_launch(const TilesArgs& a) {
  return tiles_launch_nt<DIM, KIND>(a.ntmax, a.blocks, a.s, *a.vg, *a.sp, a.xdata, a.z, a.drift_data, a.x0, a.drift_dom,
                                    a.m, a.k, a.minneighbors, a.idx, a.count, a.mean, a.var, a.status);
}
// 2-D / 3-D instantiations of one single-structure model (krig_tiles_<model>.hip)
int32_t krig_tiles_exponential(int dim, const TilesArgs& a);
int32_t krig_tiles_spherical(int dim, const TilesArgs& a);
int32_t krig_tiles_matern32(int dim, const TilesArgs& a);

}  // namespace gss
