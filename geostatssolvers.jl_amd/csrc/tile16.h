// 16 x 16 tiles in the accumulator layout of v_mfma_f64_16x16x4_f64 (lane l, register r <-> element
// (row (l >> 4) + 4 r, column l & 15)), shared by the moving-neighbourhood kernel (krig_local.hip) and the 64 x 64
// Cholesky leaf (dense_la.hip).  Register s of a tile X is the A operand of k-slice s of X' and register s of a tile
// Y is the B operand of k-slice s of Y, so acc + X'Y is four MFMAs on the tiles as they are.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

namespace gss {

typedef double d4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ d4_t xty(const d4_t& x, const d4_t& y, d4_t acc) {  // acc + X'Y
#pragma unroll
  for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[s], y[s], acc, 0, 0, 0);
  return acc;
}

// Lane J of every 16-lane row, to all lanes of that row (v_mov_b64_dpp row_newbcast): one VALU instruction and no
// SGPR round trip where v_readlane needs two plus the SGPR pair.  In the factorisations below lanes 16..63 shadow
// lanes 0..15, so the row broadcast returns what a wave-wide readlane would.
template <int J>
__device__ __forceinline__ double bc16(double x) {
  const long v = __builtin_bit_cast(long, x);
  const long r = __builtin_amdgcn_update_dpp(0L, v, 0x150 + J, 0xf, 0xf, true);
  return __builtin_bit_cast(double, r);
}
// acc (+/-)= (lane J of src's row) * mul as one v_fmac_f64_dpp.  Inline assembly is opaque to the compiler's hazard
// recogniser, and a VGPR written by a VALU instruction may be read through DPP only two wait states later: the
// caller puts dpp_fence(src) between the instruction that produces src and the first fmac_bc16 that reads it, and
// tools/check_dpp_hazards.py (run by tests/test_abi.py) scans the built code objects for any such read that the
// register allocator might have placed too early after a copy.
template <int J, bool NEG>
__device__ __forceinline__ void fmac_bc16(double& acc, double src, double mul) {
  if (NEG)
    asm("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(J));
  else
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(J));
}
__device__ __forceinline__ void dpp_fence(double& x) { asm volatile("s_nop 1" : "+v"(x)); }
// The same with the wait states inside the block, for a source that an earlier fmac_bc16 may have written only just
// before (the compiler is free to schedule the independent updates of consecutive steps back to back).
template <int J>
__device__ __forceinline__ void fmac_bc16_after_write(double& acc, double src, double mul) {
  asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(J));
}
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// Four tiles at once, one per 16-lane row of the calling wave: S4 holds four symmetric positive definite tiles
// (tile t at S4 + t * 272, element (a, b) at [a * 17 + b]); on return each holds V = U^-1 of its own tile in the same
// storage and flags[t] says whether a pivot of tile t was not positive.  The caller puts a workgroup barrier before
// (tiles written by other waves) and after (inverse factors read by them).
//
// The other waves of the workgroup wait for this routine, so it is arranged for a short dependent chain rather than
// for few instructions: W = L^-1 is built during the factorisation instead of by a second 16-step sweep.  Lane i
// holds row i; once column j of L has been used (step j) its registers take column j of M = D W (D = diag(L), i.e.
// the rows of W before their final scaling by 1 / L_ii):  M_i. = e_i - sum_{j < i} L_ij W_j.  and W_j. = M_j. / L_jj
// is complete when step j starts, so step j applies  M_ic -= (L_ij / L_jj) M_jc  (c < j, rows i > j) next to the
// usual  A_iq -= L_ij L_qj  (q > j): sixteen independent multiply-adds per step on one chain of sixteen steps.
__device__ __forceinline__ void potrf16_inverse_x4(double* S4, int lane, int* flags) {
  const int t = lane >> 4, i = lane & 15;
  double* S = S4 + t * (16 * 17);
  double row[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) row[q] = S[i * 17 + q];
  bool bad = false;
  double myy = 1.0;  // 1 / L_ii
  static_for<0, 16>([&](auto J) {
    constexpr int j = decltype(J)::value;
    double d = bc16<j>(row[j]);
    if (!(d > 0.0)) {
      bad = true;
      d = 1.0;
    }
    double y = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
    y = fma(y, fma(-h * y, y, 0.5), y);
    y = fma(y, fma(-h * y, y, 0.5), y);
    double lij = row[j] * y;                      // L_ij (rows i >= j)
    double tm = (i > j) ? -(lij * y) : 0.0;       // -L_ij / L_jj for the rows still being eliminated, 0 for finished rows
    if (i == j) myy = y;
    if (j < 15) dpp_fence(lij);
    static_for<j + 1, 16>([&](auto Q) {
      constexpr int q = decltype(Q)::value;
      fmac_bc16<q, true>(row[q], lij, lij);       // A_iq -= L_qj L_ij
    });
    static_for<0, j>([&](auto C) {
      constexpr int c = decltype(C)::value;
      fmac_bc16<j, false>(row[c], row[c], tm);    // M_ic += tm M_jc  (row j's own entry is final: tm = 0 there)
    });
    row[j] = (i == j) ? 1.0 : tm;                 // column j of M: M_jj = 1, M_ij = -L_ij / L_jj below, 0 above
    // nothing moves across the step boundary: the updates of M_ic in consecutive steps read the register the step
    // before wrote, through DPP, and must stay separated by the next step's pivot chain (checked on the built code)
    __builtin_amdgcn_sched_barrier(0);
  });
  // V = W' with W_ic = M_ic / L_ii: lane b = i writes column b.  Every lane has read its row before any write (LDS
  // operations of one wave complete in order), so the tile is overwritten in place.
#pragma unroll
  for (int r = 0; r < 16; ++r) S[r * 17 + i] = row[r] * myy;
  if (i == 0) flags[t] = bad ? 1 : 0;
}

constexpr int tile_id(int i, int j) { return i * 4 - (i * (i - 1)) / 2 + (j - i); }  // upper block triangle, i <= j

// Barrier between the LDS phases of a tile routine: the whole workgroup (one wave per workgroup, or every wave taking
// part), or -- WAVE_LOCAL -- only the calling wave of a larger workgroup whose other waves are busy elsewhere (LDS
// operations of one wave complete in order; the fence keeps the compiler from moving them).
template <bool WAVE_LOCAL>
__device__ __forceinline__ void tile_sync() {
  if (WAVE_LOCAL) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  } else {
    __syncthreads();
  }
}

// One tile (lanes 16..63 shadow lanes 0..15), with everything the 64 x 64 leaf needs from a diagonal tile:
// u = U (tile layout, upper, zero below the diagonal) for t = U'U, v = U^-1, vt = (U^-1)' (lower), and the first column
// whose pivot was not positive (-1 if none).  Same single sweep as potrf16_inverse_x4: the leaf is one wave on the
// latency chain of the fit, so the dependent chain counts.  S and S2: 16 x 17 doubles of LDS each.
template <bool WAVE_LOCAL = false>
__device__ __forceinline__ void potrf16_full(const d4_t& t, double* S, double* S2, int lane, d4_t* u, d4_t* v,
                                             d4_t* vt, int* bad_col) {
  const int g = lane >> 4, c = lane & 15;
#pragma unroll
  for (int r = 0; r < 4; ++r) S[(g + 4 * r) * 17 + c] = t[r];
  tile_sync<WAVE_LOCAL>();
  const int i = c;
  double row[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) row[q] = S[i * 17 + q];
  tile_sync<WAVE_LOCAL>();
  double myy = 1.0;
  int badc = -1;
  static_for<0, 16>([&](auto J) {
    constexpr int j = decltype(J)::value;
    double d = bc16<j>(row[j]);
    if (!(d > 0.0)) {
      if (badc < 0) badc = j;
      d = 1.0;
    }
    double y = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
    y = fma(y, fma(-h * y, y, 0.5), y);
    y = fma(y, fma(-h * y, y, 0.5), y);
    double lij = row[j] * y;                      // L_ij (rows i >= j); lane j: sqrt(d)
    S2[j * 17 + i] = (i >= j) ? lij : 0.0;        // U[j][i] = L[i][j]
    double tm = (i > j) ? -(lij * y) : 0.0;
    if (i == j) myy = y;
    if (j < 15) dpp_fence(lij);
    static_for<j + 1, 16>([&](auto Q) {
      constexpr int q = decltype(Q)::value;
      fmac_bc16<q, true>(row[q], lij, lij);
    });
    static_for<0, j>([&](auto C) {
      constexpr int cc = decltype(C)::value;
      fmac_bc16_after_write<j>(row[cc], row[cc], tm);
    });
    row[j] = (i == j) ? 1.0 : tm;
  });
  tile_sync<WAVE_LOCAL>();
#pragma unroll
  for (int r = 0; r < 4; ++r) (*u)[r] = S2[(g + 4 * r) * 17 + c];
  tile_sync<WAVE_LOCAL>();
  // lane i holds row i of W = L^-1 (after the scaling by 1 / L_ii): V = W' (lane b writes column b), VT = W
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const double w = row[r] * myy;
    S[r * 17 + i] = w;
    S2[i * 17 + r] = w;
  }
  tile_sync<WAVE_LOCAL>();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    (*v)[r] = S[(g + 4 * r) * 17 + c];
    (*vt)[r] = S2[(g + 4 * r) * 17 + c];
  }
  tile_sync<WAVE_LOCAL>();
  *bad_col = badc;
}

// One tile, only what a right-looking block step needs from it: v = U^-1 in tile layout (t = U'U) and the first column
// whose pivot was not positive (-1 if none).  potrf16_full without the U / V' outputs: 16 x 17 doubles of LDS (S), the
// calling wave alone when WAVE_LOCAL (the other waves of the workgroup wait at a barrier of their own).
// TRANSPOSED: the tile is stored transposed, i.e. the factorisation sees t' -- for a symmetric tile that is only valid in
// its upper triangle (the paired diagonal tiles of the moving-neighbourhood kernels): the sweep reads the lower triangle.
template <bool WAVE_LOCAL = false, bool TRANSPOSED = false>
__device__ __forceinline__ void potrf16_inv(const d4_t& t, double* S, int lane, d4_t* v, int* bad_col) {
  const int g = lane >> 4, c = lane & 15;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (TRANSPOSED) S[c * 17 + (g + 4 * r)] = t[r];
    else S[(g + 4 * r) * 17 + c] = t[r];
  }
  tile_sync<WAVE_LOCAL>();
  const int i = c;
  double row[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) row[q] = S[i * 17 + q];
  tile_sync<WAVE_LOCAL>();
  double myy = 1.0;
  int badc = -1;
  static_for<0, 16>([&](auto J) {
    constexpr int j = decltype(J)::value;
    double d = bc16<j>(row[j]);
    if (!(d > 0.0)) {
      if (badc < 0) badc = j;
      d = 1.0;
    }
    double y = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
    y = fma(y, fma(-h * y, y, 0.5), y);
    y = fma(y, fma(-h * y, y, 0.5), y);
    double lij = row[j] * y;
    double tm = (i > j) ? -(lij * y) : 0.0;
    if (i == j) myy = y;
    if (j < 15) dpp_fence(lij);
    static_for<j + 1, 16>([&](auto Q) {
      constexpr int q = decltype(Q)::value;
      fmac_bc16<q, true>(row[q], lij, lij);
    });
    static_for<0, j>([&](auto C) {
      constexpr int cc = decltype(C)::value;
      fmac_bc16_after_write<j>(row[cc], row[cc], tm);
    });
    row[j] = (i == j) ? 1.0 : tm;
  });
#pragma unroll
  for (int r = 0; r < 16; ++r) S[r * 17 + i] = row[r] * myy;   // V = W': lane b = i writes column b
  tile_sync<WAVE_LOCAL>();
#pragma unroll
  for (int r = 0; r < 4; ++r) (*v)[r] = S[(g + 4 * r) * 17 + c];
  tile_sync<WAVE_LOCAL>();
  *bad_col = badc;
}

// A tile parked in LDS as the image of its registers: element (lane, r) at [r * 64 + lane] (conflict-free 8-byte
// accesses); this is how the waves of the many-neighbour kernel hand tiles to each other.
__device__ __forceinline__ void tile_store(double* p, const d4_t& t, int lane) {
#pragma unroll
  for (int r = 0; r < 4; ++r) p[r * 64 + lane] = t[r];
}
__device__ __forceinline__ d4_t tile_load(const double* p, int lane) {
  d4_t t;
#pragma unroll
  for (int r = 0; r < 4; ++r) t[r] = p[r * 64 + lane];
  return t;
}

}  // namespace gss
