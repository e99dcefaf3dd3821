// FP64 MFMA tile core shared by the generic GEMM and the fused kriging quadratic-form kernel.
//
// v_mfma_f64_16x16x4_f64 (gfx950): one wave computes D(16x16) += A(16x4) * B(4x16).
//   lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15] (one double each) and
//   holds D[row = (l >> 4) + 4 r][col = l & 15] in element r of its 4-double accumulator.
//
// Workgroup tile: BM x BN = 128 x 128 outputs, BK = 16 per LDS stage, 256 threads = 4 waves in a
// 2 x 2 arrangement, each wave owning 64 x 64 = 4 x 4 MFMA tiles (128 accumulator VGPRs).
// LDS images are k-major: As[k][i], Bs[k][j] with a row stride of 144 doubles; 144 = 16 (mod 32)
// puts the two k rows that one 32-lane ds_read_b64 group touches on disjoint bank halves.
#pragma once

#include <hip/hip_runtime.h>

namespace gss {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2v __attribute__((ext_vector_type(2)));

constexpr int BM = 128;
constexpr int BN = 128;
constexpr int BK = 16;
constexpr int LDS_LD = 144;
constexpr int TILE_LDS = BK * LDS_LD;  // doubles per operand stage

__device__ __forceinline__ void mma_stage(const double* __restrict__ As, const double* __restrict__ Bs,
                                          d4 (&acc)[4][4], int wm, int wn, int lane) {
  const int lr = lane & 15;
  const int lk = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < BK / 4; ++kk) {
    const double* ap = As + (kk * 4 + lk) * LDS_LD + wm * 64 + lr;
    const double* bp = Bs + (kk * 4 + lk) * LDS_LD + wn * 64 + lr;
    double a[4], b[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      a[t] = ap[t * 16];
      b[t] = bp[t * 16];
    }
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < 4; ++tn)
        acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
  }
}

// 1 x 4 wave arrangement for triangular operands: every wave owns all 128 rows of the workgroup tile and 32 of
// its columns (8 x 2 MFMA tiles, still 128 accumulator VGPRs).  All waves then see the same rows, so the zero
// 16-row tiles of a lower-triangular A can be skipped uniformly: in the stage that covers k = i0 + 16 s .. + 15
// of the diagonal block only row tiles tm >= s are non-zero (tm_min = s; 0 for ordinary stages).
// Measured alternative, kept out of the build: the 4-block form v_mfma_f64_4x4x4_4b_f64 (lane maps
// A = 16k + 4b + i, B = 16k + 4b + j, D = 16i + 4b + j, tools/probe_mfma4x4.hip) runs this kernel at the same
// rate as the 16x16x4 form although it is 1.5x faster in a register-only loop (tools/probe_f64.hip).
// tm_max < 8: row tiles tm >= tm_max lie entirely in the zero padding below the last row of the operand.
template <bool GUARD>
__device__ __forceinline__ void mma_stage_w14(const double* __restrict__ As, const double* __restrict__ Bs,
                                              d4 (&acc)[8][2], int wave, int lane, int tm_min, int tm_max = 8) {
  const int lr = lane & 15;
  const int lk = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < BK / 4; ++kk) {
    const double* ap = As + (kk * 4 + lk) * LDS_LD + lr;
    const double* bp = Bs + (kk * 4 + lk) * LDS_LD + wave * 32 + lr;
    const double b0 = bp[0], b1 = bp[16];
    double a[8];
#pragma unroll
    for (int tm = 0; tm < 8; ++tm) a[tm] = ap[tm * 16];
#pragma unroll
    for (int tm = 0; tm < 8; ++tm) {
      if (!GUARD || (tm >= tm_min && tm < tm_max)) {
        acc[tm][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tm], b0, acc[tm][0], 0, 0, 0);
        acc[tm][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tm], b1, acc[tm][1], 0, 0, 0);
      }
    }
  }
}

__device__ __forceinline__ void zero_acc(d4 (&acc)[4][4]) {
#pragma unroll
  for (int tm = 0; tm < 4; ++tm)
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = d4{0.0, 0.0, 0.0, 0.0};
}

}  // namespace gss
