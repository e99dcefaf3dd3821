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

// Same 64 x 64 wave tile on v_mfma_f64_4x4x4_4b_f64 (4 independent 4x4x4 blocks per instruction), which
// sustains ~1.5x the FLOP rate of the 16x16x4 form on gfx950 (tools/probe_f64.hip).  Lane maps, measured
// with tools/probe_mfma4x4.hip: A lane = 16k + 4b + i, B lane = 16k + 4b + j, D lane = 16i + 4b + j for
// block b.  Instruction (u, v) gives every block the row group u and block b the column group 4v + b:
//   A operand u : lane holds A[4u + (lane & 3)][k = lane >> 4]      (replicated over b: LDS broadcast)
//   B operand v : lane holds B[k = lane >> 4][16v + (lane & 15)]    (the 16x16x4 B layout)
//   acc[u][v]   : lane holds D[4u + (lane >> 4)][16v + (lane & 15)]
__device__ __forceinline__ void mma_stage_b4(const double* __restrict__ As, const double* __restrict__ Bs,
                                             double (&acc)[16][4], int wm, int wn, int lane) {
  const int lk = lane >> 4;
#pragma unroll 1
  for (int kk = 0; kk < BK / 4; ++kk) {
    const double* ap = As + (kk * 4 + lk) * LDS_LD + wm * 64 + (lane & 3);
    const double* bp = Bs + (kk * 4 + lk) * LDS_LD + wn * 64 + (lane & 15);
    double a[16], b[4];
#pragma unroll
    for (int u = 0; u < 16; ++u) a[u] = ap[4 * u];
#pragma unroll
    for (int v = 0; v < 4; ++v) b[v] = bp[16 * v];
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[u][v] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[u], b[v], acc[u][v], 0, 0, 0);
  }
}

__device__ __forceinline__ void zero_acc(d4 (&acc)[4][4]) {
#pragma unroll
  for (int tm = 0; tm < 4; ++tm)
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = d4{0.0, 0.0, 0.0, 0.0};
}

}  // namespace gss
