// Moving-neighbourhood kriging with 257 .. 768 neighbours (krig.jl:201-210 and ui.jl:16-23 accept any count): the
// per-point systems of approxsolve's loop (/root/reference/src/estimation/krig.jl:205-228) on the MFMA tile core with
// the tile triangle in MEMORY.
//
// Up to 256 neighbours the upper block triangle of a point's system lives in the registers of its workgroup
// (krig_tiles_kernel.h); at 512 neighbours it is 528 tiles of 2 KB -- 1 MB, more than the registers and the LDS of a CU
// together.  The round-3 path for this range (krig_local_big_kernel: an unblocked column sweep) streams its O(k^3 / 3)
// operands from a per-workgroup slab in HBM and is bound by exactly that: 92 us per point at k = 512, 0.006 of the FP64
// matrix peak (profiles/r04_bigk_hugek_rows.jsonl).  Here the same right-looking block algorithm as the tile kernels,
// A = U'U by 16 x 16 tiles, runs with
//   * the tiles in a per-workgroup slab of global memory, as register images (tile_store / tile_load layout: a tile is
//     four coalesced 512-byte rows), 560 tiles = 1.1 MB per workgroup at 512 neighbours (1 224 tiles at 768), one workgroup
//     of eight waves per CU: the slabs of the whole device are 290 MB, i.e. they live in the memory-side cache and the L2s; a tile is read and written once
//     per block step that touches it: nt^3 / 6 tile round trips of 4 KB = 22 MB per point at 512 neighbours instead of
//     716 MB;
//   * the pivot row of a block step (U_Kj for every j > K, and Y_K) in LDS, as in the tile kernel;
//   * run-time loops over the tiles -- no register array, hence no compile-time slot bookkeeping: the tiles of block row
//     K and of the trailing update are dealt to the eight waves round-robin, which balances better than owned columns.
// Per block step: the wave K mod 8 factors the diagonal tile (potrf16_inv) and posts V = U_KK^-1; barrier; block row K:
// U_Kj = V'A_Kj into the pivot row, Y_K = V'B_K (its Gram contribution is taken at once); barrier; trailing update
// A_ij -= U_Ki'U_Kj, B_i -= U_Ki'Y_K on the slab; barrier.  The finish is gram_finish, shared with the other kernels.
#include "gss_internal.h"
#include "krig_local.h"
#include "tile16.h"

namespace gss {

constexpr int SLAB_KMAX = 768;
constexpr int SLAB_NTMAX = SLAB_KMAX / 16;            // 48 tile columns: the pivot row is 98 KB of the CU's 160 KB of LDS
constexpr int SLAB_W = 8;                             // waves per workgroup
// LDS of a workgroup, sized by the launch's neighbour count (ntm = ceil(k / 16) tile columns)
struct SlabLds {
  int NX, PV, VS, SS, GG, GP, DOUBLES, KROW;
  __host__ __device__ explicit SlabLds(int ntm) {
    KROW = 16 * ntm;                                       // stride of the staged right-hand-side columns
    NX = 0;                                                // KROW x 3 neighbour coordinates
    PV = NX + KROW * 3;                                    // pivot row: ntm tiles + Y_K; before step 0 the staged
    int pv = (ntm + 1) * 256;                              //   right-hand-side columns (LMAX_RHS x KROW)
    if (pv < LMAX_RHS * KROW) pv = LMAX_RHS * KROW;
    VS = PV + pv;
    SS = VS + 256;                                         // 16 x 17 scratch of the diagonal factorisation
    GG = SS + 272;                                         // 16 x 17 Gram matrix
    GP = GG + 272;                                         // W partial Gram tiles
    DOUBLES = GP + SLAB_W * 256;
  }
};
// tiles of a slab: the upper block triangle row by row, then the right-hand-side tiles
__host__ __device__ inline int slab_tix(int i, int j, int nt) { return i * nt - (i * (i - 1)) / 2 + (j - i); }
__host__ __device__ inline int64_t slab_doubles(int nt) { return (int64_t)(nt * (nt + 1) / 2 + nt) * 256; }

template <int DIM, int KIND>
__global__ __launch_bounds__(64 * SLAB_W) void krig_local_slab_kernel(
    VgDev vg, LocalSpec sp, const double* __restrict__ xdata, const double* __restrict__ z,
    const double* __restrict__ drift_data, const double* __restrict__ x0, const double* __restrict__ drift_dom, int64_t m,
    int k, int minneighbors, const int* __restrict__ idx, const int* __restrict__ count, double* __restrict__ mean_out,
    double* __restrict__ var_out, uint8_t* __restrict__ status_out, double* __restrict__ slabs, int64_t slab_stride) {
#ifndef GSS_HOST_SANITIZER_BUILD
  constexpr int W = SLAB_W;
  const int ntm = (k + 15) >> 4;                  // tile columns the launch is sized for
  const SlabLds L(ntm);
  const int KMAX = L.KROW;
  extern __shared__ double sl_sm[];
  double* nx = sl_sm + L.NX;
  double* P = sl_sm + L.PV;
  double* Vs = sl_sm + L.VS;
  double* S = sl_sm + L.SS;
  double (*G)[17] = reinterpret_cast<double (*)[17]>(sl_sm + L.GG);
  double* Gp = sl_sm + L.GP;
  __shared__ signed char se[LMAX_NC][4];
  __shared__ int s_bad;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int nc = sp.nc;
  const double NaN = __longlong_as_double(0x7ff8000000000000LL);
  const d4_t zero4 = {0.0, 0.0, 0.0, 0.0};
  double* slab = slabs + (int64_t)blockIdx.x * slab_stride;
  if (tid == 0) {
#pragma unroll
    for (int cc = 0; cc < LMAX_NC; ++cc)
#pragma unroll
      for (int a = 0; a < 3; ++a) se[cc][a] = sp.e[cc][a];
  }
  for (int64_t p = blockIdx.x; p < m; p += gridDim.x) {
    const int cnt = __builtin_amdgcn_readfirstlane(count[p]);
    __syncthreads();   // the point before has been finished by every wave (LDS and the slab are re-used)
    if (cnt < minneighbors || cnt <= 0) {  // krig.jl:213-214
      if (tid == 0) {
        mean_out[p] = NaN;
        var_out[p] = NaN;
        status_out[p] = GSS_PT_MISSING;
      }
      continue;
    }
    const int nt = (cnt + 15) >> 4;
    const int ntri = nt * (nt + 1) / 2;
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
    // ---- assembly: tiles dealt to the waves round-robin, written to the slab as register images
    {
      int t = 0;
      for (int i = 0; i < nt; ++i) {
        for (int j = i; j < nt; ++j, ++t) {
          if (t % W != wave) continue;
          double xr[4][DIM], xc[DIM], v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int a = 0; a < DIM; ++a) xr[r][a] = nx[(16 * i + g + 4 * r) * 3 + a];
          const int cj = 16 * j + c;
#pragma unroll
          for (int a = 0; a < DIM; ++a) xc[a] = nx[cj * 3 + a];
          cov_pair4_k<DIM, KIND, UNIT>(vg, xr, xc, v);
          d4_t tl;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * i + g + 4 * r;
            tl[r] = (row < cnt && cj < cnt) ? v[r] : (row == cj ? 1.0 : 0.0);
          }
          tile_store(slab + (int64_t)t * 256, tl, lane);
        }
      }
      for (int i = wave; i < nt; i += W) {   // right-hand-side tile of block row i: 16 columns, zero beyond 2 + nc
        d4_t tl;
        const bool used = c < 2 + nc;
        const double* col = rhs + (used ? c : 0) * KMAX;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double v = col[16 * i + g + 4 * r];
          tl[r] = used ? v : 0.0;
        }
        tile_store(slab + (int64_t)(ntri + i) * 256, tl, lane);
      }
    }
    __syncthreads();   // the staged columns have been read (the pivot row may take their place); the slab is complete
    d4_t Gt = zero4;
    for (int K = 0; K < nt; ++K) {
      // 1. the diagonal tile: wave K mod W factors it and posts V = U_KK^-1
      if (K % W == wave) {
        const d4_t t = tile_load(slab + (int64_t)slab_tix(K, K, nt) * 256, lane);
        d4_t v;
        int badc;
        potrf16_inv<true>(t, S, lane, &v, &badc);
        tile_store(Vs, v, lane);
        if (badc >= 0 && lane == 0) s_bad = 1;
      }
      __syncthreads();
      // 2. block row K: U_Kj = V'A_Kj into the pivot row (up to four tiles of a wave in flight); Y_K = V'B_K (one wave;
      // its share of the Gram matrix at once)
      const d4_t V = tile_load(Vs, lane);
      for (int j0 = K + 1 + wave; j0 < nt; j0 += 4 * W) {
        d4_t A[4];
#pragma unroll
        for (int b = 0; b < 4; ++b)
          if (j0 + b * W < nt) A[b] = tile_load(slab + (int64_t)slab_tix(K, j0 + b * W, nt) * 256, lane);
#pragma unroll
        for (int b = 0; b < 4; ++b)
          if (j0 + b * W < nt) tile_store(P + (j0 + b * W) * 256, xty(V, A[b], zero4), lane);
      }
      if ((K + 1) % W == wave) {
        const d4_t B = tile_load(slab + (int64_t)(ntri + K) * 256, lane);
        const d4_t Y = xty(V, B, zero4);
        tile_store(P + ntm * 256, Y, lane);
        Gt = xty(Y, Y, Gt);
      }
      __syncthreads();
      // 3. trailing update on the slab: A_ij -= U_Ki'U_Kj (K < i <= j), B_i -= U_Ki'Y_K.  The tiles of the trailing
      // triangle are numbered row by row and dealt to the waves round-robin; a wave keeps four of its tiles in flight
      // (a tile comes from the memory-side cache or the L2: ~1-2 us, against 0.1 us of products)
      {
        int i = K + 1, off = wave;
        auto norm = [&]() {
          while (i < nt && off >= nt - i) {
            off -= nt - i;
            ++i;
          }
        };
        norm();
        while (i < nt) {
          int bi[4], bj[4];
          bool ok[4];
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            ok[b] = i < nt;
            bi[b] = i;
            bj[b] = i + off;
            off += W;
            norm();
          }
          d4_t T[4];
#pragma unroll
          for (int b = 0; b < 4; ++b)
            if (ok[b]) T[b] = tile_load(slab + (int64_t)slab_tix(bi[b], bj[b], nt) * 256, lane);
#pragma unroll
          for (int b = 0; b < 4; ++b)
            if (ok[b]) T[b] = xty(-tile_load(P + bi[b] * 256, lane), tile_load(P + bj[b] * 256, lane), T[b]);
#pragma unroll
          for (int b = 0; b < 4; ++b)
            if (ok[b]) tile_store(slab + (int64_t)slab_tix(bi[b], bj[b], nt) * 256, T[b], lane);
        }
        for (int ib = K + 1 + wave; ib < nt; ib += W) {   // right-hand-side tiles
          double* tp = slab + (int64_t)(ntri + ib) * 256;
          const d4_t B = tile_load(tp, lane);
          tile_store(tp, xty(-tile_load(P + ib * 256, lane), tile_load(P + ntm * 256, lane), B), lane);
        }
      }
      __syncthreads();
    }
    // Gram matrix of the forward-substituted right-hand sides, summed over the waves
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

// 257 .. 768 neighbours.  `scratch` holds blocks * slab_doubles(nt of k) doubles.
template <int DIM, int KIND>
static int32_t slab_launch(int64_t blocks, hipStream_t s, const VgDev& vg, const LocalSpec& sp, const double* xdata,
                           const double* z, const double* drift_data, const double* x0, const double* drift_dom, int64_t m,
                           int k, int minneighbors, const int* idx, const int* count, double* mean, double* var,
                           uint8_t* status, double* scratch) {
  const size_t lds = sizeof(double) * (size_t)SlabLds((k + 15) / 16).DOUBLES;
  GSS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(krig_local_slab_kernel<DIM, KIND>),
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(sizeof(double) * (size_t)SlabLds(SLAB_NTMAX).DOUBLES)));
  hipLaunchKernelGGL((krig_local_slab_kernel<DIM, KIND>), dim3((unsigned)blocks), dim3(64 * SLAB_W), lds, s, vg, sp, xdata, z,
                     drift_data, x0, drift_dom, m, k, minneighbors, idx, count, mean, var, status, scratch,
                     slab_doubles((k + 15) / 16));
  GSS_HIP(hipGetLastError());
  return GSS_OK;
}

int32_t krig_local_slab_launch(const VgDev& vg, const LocalSpec& sp, int dim, const double* xdata, const double* z,
                               const double* drift_data, const double* x0, const double* drift_dom, int64_t m, int k,
                               int minneighbors, const int* idx, const int* count, double* mean, double* var,
                               uint8_t* status, hipStream_t s) {
  GSS_REQUIRE(k > 256 && k <= SLAB_KMAX, "krig_local_slab_launch: %d neighbours outside 257..%d", k, SLAB_KMAX);
  // One workgroup per CU.  (LDS sized by k would let two share a CU up to ~370 neighbours, but the kernel needs ~230
  // registers per lane; capped at 128 it spills 94 and ran 13-17 % slower with two workgroups per CU: measured, round 4.)
  int64_t blocks = 256;
  if (blocks > m) blocks = m;
  DevBuf scratch;
  GSS_TRY(scratch.alloc(sizeof(double) * (size_t)(blocks * slab_doubles((k + 15) / 16))));
#define GSS_SLAB_ARGS blocks, s, vg, sp, xdata, z, drift_data, x0, drift_dom, m, k, minneighbors, idx, count, mean, var, \
                      status, scratch.as<double>()
  const int kind = vg.nextra == 0 ? vg.kind : -1;
  int32_t rc;
  if (dim == 3 && kind == VG_MATERN32) rc = slab_launch<3, VG_MATERN32>(GSS_SLAB_ARGS);
  else if (dim == 3 && kind == GSS_VG_EXPONENTIAL) rc = slab_launch<3, GSS_VG_EXPONENTIAL>(GSS_SLAB_ARGS);
  else if (dim == 3 && kind == GSS_VG_SPHERICAL) rc = slab_launch<3, GSS_VG_SPHERICAL>(GSS_SLAB_ARGS);
  else if (dim == 2 && kind == VG_MATERN32) rc = slab_launch<2, VG_MATERN32>(GSS_SLAB_ARGS);
  else if (dim == 2 && kind == GSS_VG_EXPONENTIAL) rc = slab_launch<2, GSS_VG_EXPONENTIAL>(GSS_SLAB_ARGS);
  else if (dim == 2 && kind == GSS_VG_SPHERICAL) rc = slab_launch<2, GSS_VG_SPHERICAL>(GSS_SLAB_ARGS);
  else if (dim == 3) rc = slab_launch<3, -1>(GSS_SLAB_ARGS);
  else if (dim == 2) rc = slab_launch<2, -1>(GSS_SLAB_ARGS);
  else rc = slab_launch<1, -1>(GSS_SLAB_ARGS);
#undef GSS_SLAB_ARGS
  GSS_TRY(rc);
  GSS_HIP(hipStreamSynchronize(s));   // the slabs are released on return
  return GSS_OK;
}

}  // namespace gss
