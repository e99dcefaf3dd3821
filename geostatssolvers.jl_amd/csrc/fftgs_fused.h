// Fused FFTGS pipeline for power-of-two 3-D grids: five hand-written passes instead of
// noise kernel + rocFFT R2C (3 passes) + phase kernel + rocFFT C2R (3 passes).
//
//   P1  x lines : Philox noise generated in registers -> packed half-length complex FFT -> real-FFT
//                 post-processing -> half spectrum written once                       (8N B written)
//   P2  y lines : in-place forward FFT, tiles of TX = 8 consecutive kx                (8N r + 8N w)
//   P3  z lines : forward FFT -> phase X <- Fh X/|X| (fft.jl:163) -> inverse FFT,
//                 one trip through LDS                                                 (8N + 4N r, 8N w)
//   P4  y lines : in-place inverse FFT                                                 (8N r + 8N w)
//   P5  x lines : real-inverse pre-processing -> half-length inverse FFT -> realisation  (8N r + 8N w)
//
// 76 N bytes of HBM traffic per realisation against ~128 N for the rocFFT pipeline.  The transforms are
// radix-2 decimation-in-frequency forward / decimation-in-time inverse in LDS, so transformed y and z axes
// stay in bit-reversed order between passes and no reordering pass exists; the amplitude Fh is permuted
// once at create time to match (and tiled so that P3 reads it contiguously).
// Verified against a numpy model of the same dataflow (3.6e-15 from the oracle) before being written.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gss_internal.h"
#include "philox.h"

namespace gss {

// source of the real input of P1
enum { FF_SRC_PHILOX = 0,   // uniform noise generated in registers (fft.jl:163 `rand(rng, V, dims)`)
       FF_SRC_ARRAY = 1,    // caller-supplied noise
       FF_SRC_COV = 2 };    // covariance to the centre cell (fft.jl:96-99): the spectrum build runs on the same passes
struct CovSrc {
  VgDev vg;
  int c1, c2, c3;     // 0-based centre cell
  double s1, s2, s3;  // spacing
};

constexpr int FF_TX_LOG = 1;
constexpr int FF_TX = 1 << FF_TX_LOG;  // lines per tile in the strided passes: 4 complex = 64 B per row; tiles 2u and
                                       // 2u + 1 (the two halves of a 128-B line) are given to workgroups b and b + 8,
                                       // which share an XCD and are dispatched together, so the L2 merges the halves
constexpr int FF_PITCH_ALIGN = 8;      // rows of the half-spectrum buffer start on 128-B boundaries
constexpr int FF_ROWS = 4;      // x lines per workgroup in P1 / P5 (4 keeps 4+ workgroups per CU resident)
constexpr int FF_THREADS = 256;   // strided passes
constexpr int FF_XTHREADS = 256;  // x-line passes (128 threads measured slower: fewer waves to hide latency)

struct FusedGrid {
  int n1, n2, n3;      // n1 fastest; all powers of two
  int l1, l2, l3;      // log2
  int nh;              // n1 / 2 + 1
  int nhp;             // row pitch of the half-spectrum buffer: nh rounded up to FF_PITCH_ALIGN
  int ntx;             // nhp / FF_TX
  int slab_t0, slab_nt;  // strided passes restricted to the x tiles slab_t0 .. slab_t0 + slab_nt - 1 (slab_nt = 0: all)
};

__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 cconj(double2 a) { return make_double2(a.x, -a.y); }
__device__ __forceinline__ int brev_bits(int k, int bits) { return (int)(__brev((unsigned)k) >> (32 - bits)); }
// =====================================================================================================
// x passes, second generation (ff_x_fwd2_kernel / ff_x_inv2_kernel): Stockham autosort passes.
// Every pass reads its R inputs at stride M / R and writes its outputs at stride Ns (the product of the radices
// done so far), so natural order goes in and natural order comes out: no bit reversal, the first pass takes its
// inputs straight from their source into registers (Philox pairs / the supplied noise / covariance rows for P1,
// the half spectrum with the real-inverse pre-processing applied on the fly for P5) and the last pass of P5
// stores the realisation straight from registers.  LDS round trips per 256-point line: P1 3 (5 before),
// P5 2 (5 before).  Radices: 8, 8, ..., then 2^(log2 M mod 3); first-pass twiddles are all one; the other passes
// read theirs from per-pass tables T_p[(r - 1) Ns + k] = exp(-2 pi i r k / (Ns R)) (contiguous in k, so the LDS
// reads of a pass are unit stride) built once per handle.  In-register DFTs of 2, 4 and 8 points use the
// constant twiddles +-i and (+-1 +- i) / sqrt 2.
// LDS image of a line: element i at i ^ ((i >> 3) & 7): the stride-8 stores of the first pass and the
// unit-stride accesses of the others both spread over the banks.
__device__ __forceinline__ int xs_phys(int i) { return i ^ ((i >> 3) & 7); }

template <bool INV>
__device__ __forceinline__ double2 mul_mi(double2 a) {   // forward: a * (-i); inverse: a * (+i)
  return INV ? make_double2(-a.y, a.x) : make_double2(a.y, -a.x);
}

template <bool INV>
__device__ __forceinline__ void x_dft4(double2& a0, double2& a1, double2& a2, double2& a3) {
  const double2 s0 = make_double2(a0.x + a2.x, a0.y + a2.y), s1 = make_double2(a0.x - a2.x, a0.y - a2.y);
  const double2 s2 = make_double2(a1.x + a3.x, a1.y + a3.y);
  const double2 s3 = mul_mi<INV>(make_double2(a1.x - a3.x, a1.y - a3.y));
  a0 = make_double2(s0.x + s2.x, s0.y + s2.y);
  a2 = make_double2(s0.x - s2.x, s0.y - s2.y);
  a1 = make_double2(s1.x + s3.x, s1.y + s3.y);
  a3 = make_double2(s1.x - s3.x, s1.y - s3.y);
}

// v <- DFT_R(v), R = 2^NS, outputs in natural order
template <int NS, bool INV>
__device__ __forceinline__ void x_dft(double2 (&v)[1 << NS]) {
  if (NS == 1) {
    const double2 a = v[0], b = v[1];
    v[0] = make_double2(a.x + b.x, a.y + b.y);
    v[1] = make_double2(a.x - b.x, a.y - b.y);
  } else if (NS == 2) {
    x_dft4<INV>(v[0], v[1], v[2], v[3]);
  } else {
    const double h = 0.70710678118654752440;
    double2 t0 = make_double2(v[0].x + v[4].x, v[0].y + v[4].y), u0 = make_double2(v[0].x - v[4].x, v[0].y - v[4].y);
    double2 t1 = make_double2(v[1].x + v[5].x, v[1].y + v[5].y), u1 = make_double2(v[1].x - v[5].x, v[1].y - v[5].y);
    double2 t2 = make_double2(v[2].x + v[6].x, v[2].y + v[6].y), u2 = make_double2(v[2].x - v[6].x, v[2].y - v[6].y);
    double2 t3 = make_double2(v[3].x + v[7].x, v[3].y + v[7].y), u3 = make_double2(v[3].x - v[7].x, v[3].y - v[7].y);
    // u_q *= W8^q (forward: exp(-i pi q / 4); inverse: conjugate)
    if (INV) {
      u1 = make_double2((u1.x - u1.y) * h, (u1.x + u1.y) * h);
      u3 = make_double2((-u3.x - u3.y) * h, (u3.x - u3.y) * h);
    } else {
      u1 = make_double2((u1.x + u1.y) * h, (u1.y - u1.x) * h);
      u3 = make_double2((u3.y - u3.x) * h, (-u3.x - u3.y) * h);
    }
    u2 = mul_mi<INV>(u2);
    x_dft4<INV>(t0, t1, t2, t3);   // -> V0, V2, V4, V6
    x_dft4<INV>(u0, u1, u2, u3);   // -> V1, V3, V5, V7
    v[0] = t0; v[2] = t1; v[4] = t2; v[6] = t3;
    v[1] = u0; v[3] = u1; v[5] = u2; v[7] = u3;
  }
}

// twiddle multiply of inputs 1 .. R-1 by T[(r-1) Ns + k] (conjugated for the inverse)
template <int NS, bool INV>
__device__ __forceinline__ void x_twiddle(double2 (&v)[1 << NS], const double2* T, int Ns, int k) {
#pragma unroll
  for (int r = 1; r < (1 << NS); ++r) {
    double2 w = T[(r - 1) * Ns + k];
    if (INV) w.y = -w.y;
    v[r] = cmul(v[r], w);
  }
}

// middle pass (LDS -> LDS, in place) of `rows` lines of length M: radix 8.  One item per thread (the host picks
// the rows per workgroup so that rows * M / 8 <= NT): every read of the tile is done before the first write.
template <bool INV, int NT>
__device__ __forceinline__ void x_middle_pass(double2* buf, int M, int rows, const double2* T, int Ns, int logNs,
                                              int tid) {
  const int ipr = M >> 3;                 // items per row
  const bool on = tid < rows * ipr;
  double2 v[8];
  int row = 0, j = 0;
  if (on) {
    row = tid / ipr;
    j = tid - row * ipr;
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = buf[row * M + xs_phys(j + r * ipr)];
  }
  __syncthreads();
  if (on) {
    const int k = j & (Ns - 1);
    x_twiddle<3, INV>(v, T, Ns, k);
    x_dft<3, INV>(v);
    const int j0 = ((j >> logNs) << (logNs + 3)) + k;
#pragma unroll
    for (int r = 0; r < 8; ++r) buf[row * M + xs_phys(j0 + (r << logNs))] = v[r];
  }
  __syncthreads();
}

// last pass of P1 (LDS -> LDS, in place): radix R = 2^NS, at most 8 / R items per thread, outputs in natural order
template <int NS, int NT>
__device__ __forceinline__ void x_last_pass_lds(double2* buf, int M, int rows, const double2* T, int tid) {
  constexpr int R = 1 << NS;
  constexpr int IPT = 8 / R;
  const int ipr = M >> NS;
  const int nitems = rows * ipr;
  double2 v[IPT][R];
#pragma unroll
  for (int i = 0; i < IPT; ++i) {
    const int it = tid + i * NT;
    if (it < nitems) {
      const int row = it / ipr, j = it - row * ipr;
#pragma unroll
      for (int r = 0; r < R; ++r) v[i][r] = buf[row * M + xs_phys(j + r * ipr)];
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < IPT; ++i) {
    const int it = tid + i * NT;
    if (it < nitems) {
      const int row = it / ipr, j = it - row * ipr;
      x_twiddle<NS, false>(v[i], T, ipr, j);
      x_dft<NS, false>(v[i]);
#pragma unroll
      for (int r = 0; r < R; ++r) buf[row * M + xs_phys(j + r * ipr)] = v[i][r];
    }
  }
  __syncthreads();
}

struct XPlan {       // radix plan of a length-M line: passes = 1 (radix 8) + nmid (radix 8) + 1 (radix 2^ns_last)
  int nmid;
  int ns_last;
  int off_last;      // offset of the last pass's table in the per-pass twiddle table (middle passes precede it)
};
__host__ __device__ inline XPlan x_plan(int logM) {
  XPlan p;
  const int rem = logM % 3;
  p.ns_last = rem ? rem : 3;
  p.nmid = (logM - 3 - p.ns_last) / 3;
  int off = 0, Ns = 8;
  for (int m = 0; m < p.nmid; ++m) {
    off += 7 * Ns;
    Ns <<= 3;
  }
  p.off_last = off;
  return p;
}
__host__ __device__ inline int x_table_len(int logM) {
  const XPlan p = x_plan(logM);
  return p.off_last + ((1 << p.ns_last) - 1) * ((1 << logM) >> p.ns_last);
}

// ---- P1, second generation.  LDS: tw[M] | T[tlen] | buf[ROWS * M]; needs ROWS * M / 8 <= NT --------------------
// LOGM >= 0: line length fixed at compile time (index arithmetic folds into immediates); -1: any length
template <int SRC, int ROWS, int NT, int LOGM>
__global__ __launch_bounds__(NT) void ff_x_fwd2_kernel(FusedGrid g, const double2* __restrict__ tw1,
                                                       const double2* __restrict__ xtw, uint64_t seed, uint32_t real,
                                                       const double* __restrict__ noise, double2* __restrict__ X,
                                                       const CovSrc* __restrict__ cs) {
  extern __shared__ __attribute__((aligned(16))) double2 sm[];
  const int logM = LOGM >= 0 ? LOGM : g.l1 - 1;
  const int M = 1 << logM;
  const XPlan plan = x_plan(logM);
  const int tlen = x_table_len(logM);
  double2* tw = sm;
  double2* T = sm + M;
  double2* buf = T + tlen;
  const int tid = threadIdx.x;
  const int64_t nrows = (int64_t)g.n2 * g.n3;
  const int64_t row0 = (int64_t)blockIdx.x * ROWS;
  for (int k = tid; k < M; k += NT) tw[k] = tw1[k];
  for (int k = tid; k < tlen; k += NT) T[k] = xtw[k];
  // pass 1: radix 8 straight from the source, no twiddles (Ns = 1), outputs at 8 j + r
  {
    const int ipr = M >> 3;
    for (int it = tid; it < ROWS * ipr; it += NT) {
      const int row = it / ipr, j = it - row * ipr;
      const int64_t grow = row0 + row;
      double2 v[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int n = j + r * ipr;
        double2 x = make_double2(0.0, 0.0);
        if (grow < nrows) {
          const int64_t blk = grow * M + n;  // elements 2 blk, 2 blk + 1 of the realisation
          if (SRC == FF_SRC_ARRAY) {
            x = reinterpret_cast<const double2*>(noise)[blk];
          } else if (SRC == FF_SRC_COV) {
            const int i2 = (int)(grow % g.n2), i3 = (int)(grow / g.n2);
            const double zero[3] = {0.0, 0.0, 0.0};
            double a[3] = {(double)(2 * n - cs->c1) * cs->s1, (double)(i2 - cs->c2) * cs->s2, (double)(i3 - cs->c3) * cs->s3};
            x.x = cov_pair<3>(cs->vg, a, zero);
            a[0] = (double)(2 * n + 1 - cs->c1) * cs->s1;
            x.y = cov_pair<3>(cs->vg, a, zero);
          } else {
            philox_pair(seed, real, STREAM_UNIFORM, (uint64_t)blk, x.x, x.y);
          }
        }
        v[r] = x;
      }
      x_dft<3, false>(v);
#pragma unroll
      for (int r = 0; r < 8; ++r) buf[row * M + xs_phys(8 * j + r)] = v[r];
    }
  }
  __syncthreads();
  int Ns = 8, logNs = 3;
  const double2* Tp = T;
  for (int m = 0; m < plan.nmid; ++m) {
    x_middle_pass<false, NT>(buf, M, ROWS, Tp, Ns, logNs, tid);
    Tp += 7 * Ns;
    Ns <<= 3;
    logNs += 3;
  }
  // last pass: radix 2^ns_last, Ns = M / R, outputs Z[k] at k = j + r Ns: natural order, in place
  if (plan.ns_last == 3) x_last_pass_lds<3, NT>(buf, M, ROWS, Tp, tid);
  else if (plan.ns_last == 2) x_last_pass_lds<2, NT>(buf, M, ROWS, Tp, tid);
  else x_last_pass_lds<1, NT>(buf, M, ROWS, Tp, tid);
  const double2* Z = buf;
  // X[k] = ((Zk + conj(Z_{M-k})) - i w^k (Zk - conj(Z_{M-k}))) / 2, k = 0 .. M.  Outputs k and M - k share
  // s = Zk + conj(Z_{M-k}) and p = w^k (Zk - conj(Z_{M-k})) (w^{M-k} = -conj(w^k)):
  //   X[k] = (s.x + p.y, s.y - p.x) / 2,   X[M-k] = (s.x - p.y, -s.y - p.x) / 2
  // (k = 0 gives X[0] and X[M]; k = M / 2 gives X[M/2] twice)
  const int half = M >> 1;
  for (int t = tid; t < ROWS * half + ROWS; t += NT) {
    int row, k;
    if (t < ROWS * half) {
      row = t >> (logM - 1);
      k = t & (half - 1);
    } else {
      row = t - ROWS * half;
      k = half;
    }
    const int64_t grow = row0 + row;
    if (grow >= nrows) continue;
    const double2* z = Z + row * M;
    const double2 a = z[xs_phys(k)];
    const double2 b = cconj(z[xs_phys((M - k) & (M - 1))]);
    const double2 w = tw[k];
    const double2 sm2 = make_double2(a.x + b.x, a.y + b.y);
    const double2 pp = cmul(make_double2(a.x - b.x, a.y - b.y), w);
    double2* xr = X + grow * g.nhp;
    xr[k] = make_double2(0.5 * (sm2.x + pp.y), 0.5 * (sm2.y - pp.x));
    if (k != M - k) xr[M - k] = make_double2(0.5 * (sm2.x - pp.y), 0.5 * (-sm2.y - pp.x));
  }
}

// ---- P5, second generation.  LDS: T[tlen] | buf[ROWS * M]; needs ROWS * M / 8 <= NT -----------------------------
template <int ROWS, int NT, int LOGM>
__global__ __launch_bounds__(NT) void ff_x_inv2_kernel(FusedGrid g, const double2* __restrict__ tw1,
                                                       const double2* __restrict__ xtw, const double2* __restrict__ X,
                                                       double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) double2 sm[];
  const int logM = LOGM >= 0 ? LOGM : g.l1 - 1;
  const int M = 1 << logM;
  const XPlan plan = x_plan(logM);
  const int tlen = x_table_len(logM);
  double2* T = sm;
  double2* buf = sm + tlen;
  const int tid = threadIdx.x;
  const int64_t nrows = (int64_t)g.n2 * g.n3;
  const int64_t row0 = (int64_t)blockIdx.x * ROWS;
  for (int k = tid; k < tlen; k += NT) T[k] = xtw[k];
  // pass 1: Z'[k] = (Xk + conj(X_{M-k})) + i conj(w)^k (Xk - conj(X_{M-k})) formed in registers from the half
  // spectrum (k and its mirror: two unit-stride runs per load instruction), radix 8 without twiddles
  {
    const int ipr = M >> 3;
    for (int it = tid; it < ROWS * ipr; it += NT) {
      const int row = it / ipr, j = it - row * ipr;
      const int64_t grow = row0 + row;
      double2 v[8];
      if (grow < nrows) {
        const double2* xr = X + grow * g.nhp;
        double2 a[8], b[8], w[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int k = j + r * ipr;
          a[r] = xr[k];
          b[r] = xr[M - k];
          w[r] = tw1[k];
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const double2 bb = cconj(b[r]);
          const double2 d = cmul(make_double2(a[r].x - bb.x, a[r].y - bb.y), cconj(w[r]));  // i * d = (-d.y, d.x)
          v[r] = make_double2(a[r].x + bb.x - d.y, a[r].y + bb.y + d.x);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = make_double2(0.0, 0.0);
      }
      x_dft<3, true>(v);
#pragma unroll
      for (int r = 0; r < 8; ++r) buf[row * M + xs_phys(8 * j + r)] = v[r];
    }
  }
  __syncthreads();
  int Ns = 8, logNs = 3;
  const double2* Tp = T;
  for (int m = 0; m < plan.nmid; ++m) {
    x_middle_pass<true, NT>(buf, M, ROWS, Tp, Ns, logNs, tid);
    Tp += 7 * Ns;
    Ns <<= 3;
    logNs += 3;
  }
  // last pass: outputs z[n], n = j + r M / R, are (u[2n], u[2n+1]) of the realisation: stored from registers
  {
    const int ipr = M >> plan.ns_last;
    double2* o2 = reinterpret_cast<double2*>(out);
    for (int it = tid; it < ROWS * ipr; it += NT) {
      const int row = it / ipr, j = it - row * ipr;
      const int64_t grow = row0 + row;
      if (plan.ns_last == 3) {
        double2 v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = buf[row * M + xs_phys(j + r * ipr)];
        x_twiddle<3, true>(v, Tp, ipr, j);
        x_dft<3, true>(v);
        if (grow < nrows) {
#pragma unroll
          for (int r = 0; r < 8; ++r) o2[grow * M + j + r * ipr] = v[r];
        }
      } else if (plan.ns_last == 2) {
        double2 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = buf[row * M + xs_phys(j + r * ipr)];
        x_twiddle<2, true>(v, Tp, ipr, j);
        x_dft<2, true>(v);
        if (grow < nrows) {
#pragma unroll
          for (int r = 0; r < 4; ++r) o2[grow * M + j + r * ipr] = v[r];
        }
      } else {
        double2 v[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) v[r] = buf[row * M + xs_phys(j + r * ipr)];
        x_twiddle<1, true>(v, Tp, ipr, j);
        x_dft<1, true>(v);
        if (grow < nrows) {
#pragma unroll
          for (int r = 0; r < 2; ++r) o2[grow * M + j + r * ipr] = v[r];
        }
      }
    }
  }
}

// spectrum build on the fused passes: X holds the forward transform in the padded, (z, y) bit-reversed layout;
// Fh[kz][ky][kx] = sqrt(|X[brev kz][brev ky][kx]|), DC = 0 (fft.jl:102-103); partial[b] = this block's share of
// sum F^2 over the FULL spectrum (w = 2 for entries that stand for a conjugate pair)
__global__ __launch_bounds__(256) void ff_amp_kernel(FusedGrid g, const double2* __restrict__ X, double* __restrict__ Fh,
                                                     double* __restrict__ partial) {
  __shared__ double red[256];
  const int64_t NH = (int64_t)g.nh * g.n2 * g.n3;
  double acc = 0.0;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < NH; idx += (int64_t)gridDim.x * 256) {
    const int kx = (int)(idx % g.nh);
    const int64_t r = idx / g.nh;
    const int ky = (int)(r % g.n2), kz = (int)(r / g.n2);
    const double2 x = X[((int64_t)brev_bits(kz, g.l3) * g.n2 + brev_bits(ky, g.l2)) * g.nhp + kx];
    double f = sqrt(sqrt(x.x * x.x + x.y * x.y));
    if (idx == 0) f = 0.0;
    Fh[idx] = f;
    const bool self = (kx == 0) || (2 * kx == g.n1);
    acc += (self ? 1.0 : 2.0) * f * f;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// =====================================================================================================
// Strided passes, second generation (ff_axis2_kernel).  Same mathematics and the same bit-reversed data order as
// ff_axis_kernel above; what changes is where the data sit between the radix passes:
//   * the FIRST pass of a transform takes its inputs straight from global memory into registers and the LAST pass
//     stores its outputs straight from registers (a radix-R item holds R rows of one column; lanes run over the
//     columns of a tile first, so every load / store instruction moves whole 16 * TX byte row pieces);
//   * in P3 the last forward pass, the phase step (fft.jl:163) and the first inverse pass act on the same R
//     consecutive line elements, so they run back to back in registers;
//   * LDS only carries the exchanges between passes: 2 round trips and 2 barriers for a 512-point line where the
//     first generation needs 4 and 5 (P3: 4 and 4 instead of 9 and 10).
// Tiles are TX = 8 columns wide (128-B row pieces: whole cache lines, no reliance on two workgroups meeting in L2)
// or 4; the LDS image is row-major [j][c] with the rows of each 16-slot window permuted (a2_phys) so that the
// d = 1 pass (rows 8 r + q for consecutive r) and the unit-stride passes are both bank-conflict free for the lane
// groups of ds_read/write_b128.
// The inverse transform runs its remainder radix (log2 L mod 3 stages) FIRST, so that its first pass covers the
// same elements as the last forward pass.
constexpr int FF2_PAD = 0;

// LDS slot (16-B units) of tile element (row j, column c): rows are permuted inside every aligned window of
// 16 / TX rows (one window = 16 slots = all 64 banks) by XOR with bits of j >> 3, so that rows 8 r + q of
// consecutive r -- the d = 1 pass -- fall on different bank groups while unit-stride row runs stay conflict free
template <int TXLOG>
__device__ __forceinline__ int a2_phys(int j, int c) {
  if (TXLOG >= 4) return (j << TXLOG) + c;
  constexpr int WL = 4 - TXLOG;            // log2 rows per window
  constexpr int RW = 1 << WL;
  return ((j >> WL) << 4) + ((((j ^ (j >> 3)) & (RW - 1)) << TXLOG) | c);
}

// forward DIF butterflies of stages s0 .. s0+NS-1 on the R = 2^NS elements e0 + q d of one column (d = 2^logd)
template <int NS>
__device__ __forceinline__ void a2_dif(double2 (&x)[1 << NS], int p, int logd, int s0, const double2* tw) {
  constexpr int R = 1 << NS;
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    const int span = R >> (u + 1);
    const int s = s0 + u;
#pragma unroll
    for (int q = 0; q < R; ++q) {
      if ((q & span) == 0) {
        const int pos = p + ((q & (span - 1)) << logd);
        const double2 w = tw[pos << s];
        const double2 a = x[q], c = x[q + span];
        x[q] = make_double2(a.x + c.x, a.y + c.y);
        x[q + span] = cmul(make_double2(a.x - c.x, a.y - c.y), w);
      }
    }
  }
}

// inverse DIT butterflies of stages s0 .. s0+NS-1 (element spacing d = 2^s0)
template <int NS>
__device__ __forceinline__ void a2_dit(double2 (&x)[1 << NS], int p, int s0, int logL, const double2* tw) {
  constexpr int R = 1 << NS;
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    const int span = 1 << u;
    const int s = s0 + u;
#pragma unroll
    for (int q = 0; q < R; ++q) {
      if ((q & span) == 0) {
        const int pos = p + ((q & (span - 1)) << s0);
        const double2 w = cconj(tw[pos << (logL - 1 - s)]);
        const double2 a = x[q], c = cmul(x[q + span], w);
        x[q] = make_double2(a.x + c.x, a.y + c.y);
        x[q + span] = make_double2(a.x - c.x, a.y - c.y);
      }
    }
  }
}

enum { A2_GLOBAL = 0, A2_LDS = 1, A2_PHASE = 2 };

// One pass over the tile.  FWD: DIF stages s0..s0+NS-1, else DIT.  SRC: A2_GLOBAL | A2_LDS.
// DST: A2_GLOBAL | A2_LDS | A2_PHASE (forward only: phase step, then the DIT stages 0..NS-1, result to LDS).
template <int NS, bool FWD, int SRC, int DST, int TXLOG, int NT>
__device__ __forceinline__ void a2_pass(double2* __restrict__ gbase, int64_t lstride, double2* buf, const double2* tw,
                                        int logL, int s0, int tid, const double* __restrict__ fh, bool dc_tile,
                                        double mean) {
  constexpr int R = 1 << NS;
  constexpr int TX = 1 << TXLOG;
  const int logd = FWD ? (logL - s0 - NS) : s0;
  const int d = 1 << logd;
  const int nitems = (TX << logL) >> NS;
  for (int it = tid; it < nitems; it += NT) {
    const int c = it & (TX - 1);
    const int r = it >> TXLOG;
    const int p = r & (d - 1);
    const int e0 = ((r >> logd) << (logd + NS)) + p;
    double2 x[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int j = e0 + (q << logd);
      x[q] = (SRC == A2_GLOBAL) ? gbase[(int64_t)j * lstride + c] : buf[a2_phys<TXLOG>(j, c)];
    }
    if (FWD) a2_dif<NS>(x, p, logd, s0, tw);
    else a2_dit<NS>(x, p, s0, logL, tw);
    if (DST == A2_PHASE) {
      // last forward pass: logd = 0, the item holds line elements R r .. R r + R - 1 (frequency = their bit reversal);
      // Fh_tiled is stored in exactly the order the items read it: fh[q * nitems + it]
#pragma unroll
      for (int q = 0; q < R; ++q) {
        const double f = fh[(int64_t)q * nitems + it];
        const double mag2 = x[q].x * x[q].x + x[q].y * x[q].y;
        double2 pz;
        if (mag2 > 0.0) {
          // f / |X| as f * rsqrt(|X|^2): hardware seed + two Newton steps (relative error a few 1e-16) instead of a
          // square root followed by a division (12 instead of ~45 instructions per element)
          double y = __builtin_amdgcn_rsq(mag2);
          double e = fma(-(mag2 * y), y, 1.0);
          y = fma(0.5 * y, e, y);
          e = fma(-(mag2 * y), y, 1.0);
          y = fma(0.5 * y, e, y);
          const double inv = f * y;
          pz = make_double2(x[q].x * inv, x[q].y * inv);
        } else {
          pz = make_double2(f, 0.0);  // angle(0) = 0
        }
        if (dc_tile && it == 0 && q == 0) pz = make_double2(mean, 0.0);  // DC <- mean
        x[q] = pz;
      }
      a2_dit<NS>(x, 0, 0, logL, tw);  // first inverse pass: same elements, spacing 1 (p = 0)
    }
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int j = e0 + (q << logd);
      if (DST == A2_GLOBAL) gbase[(int64_t)j * lstride + c] = x[q];
      else buf[a2_phys<TXLOG>(j, c)] = x[q];
    }
  }
}

template <bool FWD, int SRC, int DST, int TXLOG, int NT>
__device__ __forceinline__ void a2_pass_ns(int ns, double2* gbase, int64_t lstride, double2* buf, const double2* tw,
                                           int logL, int s0, int tid, const double* fh, bool dc_tile, double mean) {
  if (ns == 3) a2_pass<3, FWD, SRC, DST, TXLOG, NT>(gbase, lstride, buf, tw, logL, s0, tid, fh, dc_tile, mean);
  else if (ns == 2) a2_pass<2, FWD, SRC, DST, TXLOG, NT>(gbase, lstride, buf, tw, logL, s0, tid, fh, dc_tile, mean);
  else a2_pass<1, FWD, SRC, DST, TXLOG, NT>(gbase, lstride, buf, tw, logL, s0, tid, fh, dc_tile, mean);
}

// MODE 0: forward DIF in place.  MODE 1: inverse DIT in place.  MODE 2: forward, phase with Fh, inverse.
// Needs log2 L >= 4 (at least two passes per transform).  LDS: tw[L/2] | buf[L * TX + FF2_PAD]
// LOGL >= 0: line length fixed at compile time; -1: taken from the argument
template <int MODE, int TXLOG, int NT, int LOGL>
__global__ __launch_bounds__(NT) void ff_axis2_kernel(FusedGrid g, int logL_arg, const double2* __restrict__ twL,
                                                      int64_t ostride, int64_t lstride, double2* __restrict__ X,
                                                      const double* __restrict__ Fh_tiled, double mean) {
  extern __shared__ __attribute__((aligned(16))) double2 sm[];
  constexpr int TX = 1 << TXLOG;
  const int logL = LOGL >= 0 ? LOGL : logL_arg;
  const int L = 1 << logL;
  double2* tw = sm;
  double2* buf = sm + (L >> 1);
  const int tid = threadIdx.x;
  const int ntx = g.nhp >> TXLOG;
  int tile = blockIdx.x;
  if (TXLOG < 3) {  // tiles that share a 128-B line go to blocks b, b + 8, ... (same XCD, dispatched together)
    constexpr int GL = 3 - (TXLOG < 3 ? TXLOG : 3);
    const int bb = blockIdx.x;
    tile = (bb & ~((8 << GL) - 1)) + ((bb & 7) << GL) + ((bb >> 3) & ((1 << GL) - 1));
  }
  const int t = tile % ntx;
  const int o = tile / ntx;
  double2* gbase = X + (int64_t)o * ostride + (int64_t)t * TX;
  const int rem = logL % 3;
  const int last_ns = rem ? rem : 3;   // radix of the last forward pass = radix of the first inverse pass
  const int nitems_last = (TX << logL) >> last_ns;
  const double* fh = (MODE == 2) ? Fh_tiled + (int64_t)tile * ((int64_t)L * TX) : nullptr;
  (void)nitems_last;
  for (int k = tid; k < (L >> 1); k += NT) tw[k] = twL[k];
  __syncthreads();
  if (MODE == 0 || MODE == 2) {
    // first pass: radix-8 from global
    a2_pass<3, true, A2_GLOBAL, A2_LDS, TXLOG, NT>(gbase, lstride, buf, tw, logL, 0, tid, nullptr, false, 0.0);
    __syncthreads();
    int s = 3;
    for (; s + 3 <= logL - last_ns; s += 3) {
      a2_pass<3, true, A2_LDS, A2_LDS, TXLOG, NT>(gbase, lstride, buf, tw, logL, s, tid, nullptr, false, 0.0);
      __syncthreads();
    }
    // last forward pass (s == logL - last_ns)
    if (MODE == 0) {
      a2_pass_ns<true, A2_LDS, A2_GLOBAL, TXLOG, NT>(last_ns, gbase, lstride, buf, tw, logL, s, tid, nullptr, false, 0.0);
    } else {
      // the pass reads and writes the same LDS elements of its own items only: no barrier in between
      a2_pass_ns<true, A2_LDS, A2_PHASE, TXLOG, NT>(last_ns, gbase, lstride, buf, tw, logL, s, tid, fh,
                                                     o == 0 && t == 0, mean);
      __syncthreads();
    }
  }
  if (MODE == 1) {
    a2_pass_ns<false, A2_GLOBAL, A2_LDS, TXLOG, NT>(last_ns, gbase, lstride, buf, tw, logL, 0, tid, nullptr, false, 0.0);
    __syncthreads();
  }
  if (MODE == 1 || MODE == 2) {
    int s = last_ns;
    for (; s + 3 < logL; s += 3) {
      a2_pass<3, false, A2_LDS, A2_LDS, TXLOG, NT>(gbase, lstride, buf, tw, logL, s, tid, nullptr, false, 0.0);
      __syncthreads();
    }
    // last inverse pass: radix-8 (s + 3 == logL) to global
    a2_pass<3, false, A2_LDS, A2_GLOBAL, TXLOG, NT>(gbase, lstride, buf, tw, logL, s, tid, nullptr, false, 0.0);
  }
}

// The same passes for lines whose length is a power of 8 with exactly one radix-8 item per thread and pass
// (512-point lines: 8 columns x 64 items = 512 threads), written out pass by pass so that every global load of the
// workgroup -- the twiddle table, the tile rows of the first pass and, in P3, the amplitudes of the phase step --
// is issued before the first wait: the table's barrier no longer stands between the kernel start and the tile
// loads, and the amplitudes arrive while the first two passes run.
template <int MODE, int TXLOG, int NT, int LOGL>
__global__ __launch_bounds__(NT) void ff_axis2_fast_kernel(FusedGrid g, const double2* __restrict__ twL,
                                                           int64_t ostride, int64_t lstride, double2* __restrict__ X,
                                                           const double* __restrict__ Fh_tiled, double mean) {
  static_assert(LOGL % 3 == 0 && LOGL >= 6, "all passes radix 8, at least two");
  constexpr int TX = 1 << TXLOG;
  constexpr int L = 1 << LOGL;
  constexpr int NITEMS = (TX << LOGL) >> 3;
  static_assert(NITEMS == NT && (L >> 1) <= NT, "one item per thread");
  extern __shared__ __attribute__((aligned(16))) double2 sm[];
  double2* tw = sm;
  double2* buf = sm + (L >> 1);
  const int tid = threadIdx.x;
  const int ntx = g.nhp >> TXLOG;
  int tile = blockIdx.x;
  if (g.slab_nt > 0) tile = (int)(blockIdx.x / (unsigned)g.slab_nt) * ntx + g.slab_t0 + (int)(blockIdx.x % (unsigned)g.slab_nt);
  if (TXLOG < 3) {
    constexpr int GL = 3 - (TXLOG < 3 ? TXLOG : 3);
    const int bb = blockIdx.x;
    tile = (bb & ~((8 << GL) - 1)) + ((bb & 7) << GL) + ((bb >> 3) & ((1 << GL) - 1));
  }
  const int t = tile % ntx;
  const int o = tile / ntx;
  double2* gbase = X + (int64_t)o * ostride + (int64_t)t * TX;
  const int c = tid & (TX - 1);
  const int r = tid >> TXLOG;          // 0 .. L/8 - 1
  // ---- every global load of the workgroup, issued up front
  double2 twv = make_double2(0.0, 0.0);
  if (tid < (L >> 1)) twv = twL[tid];
  double2 x[8];
  if (MODE == 1) {  // first inverse pass: rows 8 r + q
#pragma unroll
    for (int q = 0; q < 8; ++q) x[q] = gbase[(int64_t)(8 * r + q) * lstride + c];
  } else {          // first forward pass: rows r + q L / 8
#pragma unroll
    for (int q = 0; q < 8; ++q) x[q] = gbase[(int64_t)(r + q * (L >> 3)) * lstride + c];
  }
  double fhv[8];
  if (MODE == 2) {
    const double* fh = Fh_tiled + (int64_t)tile * ((int64_t)L * TX);
#pragma unroll
    for (int q = 0; q < 8; ++q) fhv[q] = fh[q * NITEMS + tid];
  }
  if (tid < (L >> 1)) tw[tid] = twv;
  __syncthreads();
  if (MODE != 1) {
    // forward passes s0 = 0, 3, ..., LOGL - 3 (spacing 2^(LOGL - s0 - 3))
    a2_dif<3>(x, r, LOGL - 3, 0, tw);
#pragma unroll
    for (int q = 0; q < 8; ++q) buf[a2_phys<TXLOG>(r + q * (L >> 3), c)] = x[q];
    __syncthreads();
#pragma unroll
    for (int s0 = 3; s0 < LOGL; s0 += 3) {
      const int logd = LOGL - s0 - 3;
      const int p = r & ((1 << logd) - 1);
      const int e0 = ((r >> logd) << (logd + 3)) + p;
#pragma unroll
      for (int q = 0; q < 8; ++q) x[q] = buf[a2_phys<TXLOG>(e0 + (q << logd), c)];
      a2_dif<3>(x, p, logd, s0, tw);
      if (s0 + 3 < LOGL) {
#pragma unroll
        for (int q = 0; q < 8; ++q) buf[a2_phys<TXLOG>(e0 + (q << logd), c)] = x[q];
        __syncthreads();
      }
    }
    // x holds line elements 8 r .. 8 r + 7 of column c (bit-reversed frequency order)
    if (MODE == 0) {
#pragma unroll
      for (int q = 0; q < 8; ++q) gbase[(int64_t)(8 * r + q) * lstride + c] = x[q];
      return;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const double f = fhv[q];
      const double mag2 = x[q].x * x[q].x + x[q].y * x[q].y;
      double2 pz;
      if (mag2 > 0.0) {
        double y = __builtin_amdgcn_rsq(mag2);
        double e = fma(-(mag2 * y), y, 1.0);
        y = fma(0.5 * y, e, y);
        e = fma(-(mag2 * y), y, 1.0);
        y = fma(0.5 * y, e, y);
        const double inv = f * y;
        pz = make_double2(x[q].x * inv, x[q].y * inv);
      } else {
        pz = make_double2(f, 0.0);  // angle(0) = 0
      }
      if (o == 0 && t == 0 && tid == 0 && q == 0) pz = make_double2(mean, 0.0);  // DC <- mean
      x[q] = pz;
    }
  }
  // ---- inverse passes s0 = 0, 3, ..., LOGL - 3 (spacing 2^s0); x holds the inputs of the first one
  a2_dit<3>(x, 0, 0, LOGL, tw);
#pragma unroll
  for (int q = 0; q < 8; ++q) buf[a2_phys<TXLOG>(8 * r + q, c)] = x[q];
  __syncthreads();
#pragma unroll
  for (int s0 = 3; s0 < LOGL; s0 += 3) {
    const int p = r & ((1 << s0) - 1);
    const int e0 = ((r >> s0) << (s0 + 3)) + p;
#pragma unroll
    for (int q = 0; q < 8; ++q) x[q] = buf[a2_phys<TXLOG>(e0 + (q << s0), c)];
    a2_dit<3>(x, p, s0, LOGL, tw);
    if (s0 + 3 < LOGL) {
#pragma unroll
      for (int q = 0; q < 8; ++q) buf[a2_phys<TXLOG>(e0 + (q << s0), c)] = x[q];
      __syncthreads();
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) gbase[(int64_t)(e0 + (q << s0)) * lstride + c] = x[q];
    }
  }
}

// Fh -> the order in which the items of ff_axis2_kernel<2> read it.  Tile (y', t) of TX columns; the last forward
// pass has radix R = 2^last_ns and nitems = L TX / R items it = r TX + c holding line elements z' = R r + q:
//   dst[tile * L * TX + q * nitems + it] = Fh[brev(z')][brev(y')][t * TX + c]   (0 beyond nh)
template <int TXLOG>
__global__ __launch_bounds__(256) void ff_tile_fh2_kernel(FusedGrid g, const double* __restrict__ Fh,
                                                          double* __restrict__ dst) {
  constexpr int TX = 1 << TXLOG;
  const int ntx = g.nhp >> TXLOG;
  const int L = g.n3;
  const int rem = g.l3 % 3;
  const int last_ns = rem ? rem : 3;
  const int nitems = (TX << g.l3) >> last_ns;
  const int64_t per_tile = (int64_t)L * TX;
  const int64_t total = (int64_t)g.n2 * ntx * per_tile;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t tile = e / per_tile;
    const int w = (int)(e - tile * per_tile);
    const int q = w / nitems, it = w - q * nitems;
    const int c = it & (TX - 1), r = it >> TXLOG;
    const int zp = (r << last_ns) + q;
    const int t = (int)(tile % ntx), yp = (int)(tile / ntx);
    const int kx = t * TX + c;
    const int kz = brev_bits(zp, g.l3), ky = brev_bits(yp, g.l2);
    dst[e] = kx < g.nh ? Fh[((int64_t)kz * g.n2 + ky) * g.nh + kx] : 0.0;
  }
}

}  // namespace gss
