// FFTGS on the library's own passes for grids the power-of-two pipeline (fftgs_fused.h) does not take: 2-D grids and
// 3-D grids whose sizes are products of 2, 3, 5 and 7 -- the reference's own test grids are 100 x 100
// (/root/reference/test/simulation/fft.jl:4,11,26), and 200^3 / 300^3 / 500^3 are the round numbers users pick.
//
// Same mathematics and the same pass structure as the fused pipeline (fft.jl:96-103 for the spectrum, :163-170 for a
// realisation):
//   GP1  x lines : noise (Philox in registers / supplied) or covariance rows -> packed half-length complex FFT ->
//                  real-FFT post-processing -> half spectrum                               (8N B written)
//   GP2  y lines : forward, tiles of TX columns (3-D only)                                 (8N r + 8N w)
//   GP3  last axis (z, or y of a 2-D grid): forward -> phase X <- Fh X / |X| -> inverse    (8N + 4N r, 8N w)
//   GP4  y lines : inverse (3-D only)
//   GP5  x lines : real-inverse pre-processing -> half-length inverse FFT -> realisation   (8N r + 8N w)
// i.e. five passes (three on 2-D grids) against the eight full-volume passes of noise kernel + rocFFT R2C + phase kernel
// + rocFFT C2R.  What differs from the power-of-two passes is the transform inside a pass: Stockham autosort passes of
// radix 2, 3, 4, 5, 7 or 8 chosen at run time from the factorisation of the line length, natural order in and out on every
// axis (so the amplitudes Fh are read in their natural layout -- the state buffer of the handle -- and no permuted copy
// exists), every pass in place in LDS with the items of a thread held in registers between the read and the write of
// the tile.  A line is at most 2 048 complex elements (x: n1 / 2) or 1 024 (y, z); everything else stays on rocFFT.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fftgs_fused.h"
#include "gss_internal.h"
#include "philox.h"

namespace gss {

constexpr int GEN_MAX_PASSES = 12;
struct GenPlan {          // one axis: radices of the Stockham passes and where each pass's twiddles start in its table
  int L;                  // line length (x: n1 / 2)
  int npass;
  int radix[GEN_MAX_PASSES];
  int toff[GEN_MAX_PASSES];
  int tlen;               // complex entries of the table: sum over the passes of (R - 1) * Ns
  int has7;               // a radix-7 pass: tiles of this line hold at most 7 elements per thread (g_pass)
};
struct GenGrid {
  int n1, n2, n3;         // n1 fastest; n3 = 1 on 2-D grids
  int nh, nhp;            // n1 / 2 + 1 and the row pitch of the half-spectrum buffer (a multiple of 8)
  int ndim;
  int c1, c2, c3;         // centre cell (covariance source)
  double s1, s2, s3;
};

// ---- in-register DFTs ------------------------------------------------------------------------------------------------
template <bool INV>
__device__ __forceinline__ void g_dft3(double2 (&v)[3]) {
  const double s = 0.86602540378443864676;   // sin(2 pi / 3)
  const double2 t1 = make_double2(v[1].x + v[2].x, v[1].y + v[2].y);
  const double2 t2 = make_double2(v[0].x - 0.5 * t1.x, v[0].y - 0.5 * t1.y);
  const double2 d = make_double2((v[1].x - v[2].x) * s, (v[1].y - v[2].y) * s);
  const double2 r = mul_mi<INV>(d);          // forward: -i d, inverse: +i d
  v[0] = make_double2(v[0].x + t1.x, v[0].y + t1.y);
  v[1] = make_double2(t2.x + r.x, t2.y + r.y);
  v[2] = make_double2(t2.x - r.x, t2.y - r.y);
}
template <bool INV>
__device__ __forceinline__ void g_dft5(double2 (&v)[5]) {
  const double c1 = 0.30901699437494742410, c2 = -0.80901699437494742410;   // cos(2 pi / 5), cos(4 pi / 5)
  const double s1 = 0.95105651629515357212, s2 = 0.58778525229247312917;    // sin(2 pi / 5), sin(4 pi / 5)
  const double2 t1 = make_double2(v[1].x + v[4].x, v[1].y + v[4].y), t2 = make_double2(v[2].x + v[3].x, v[2].y + v[3].y);
  const double2 t3 = make_double2(v[1].x - v[4].x, v[1].y - v[4].y), t4 = make_double2(v[2].x - v[3].x, v[2].y - v[3].y);
  const double2 a1 = make_double2(v[0].x + c1 * t1.x + c2 * t2.x, v[0].y + c1 * t1.y + c2 * t2.y);
  const double2 a2 = make_double2(v[0].x + c2 * t1.x + c1 * t2.x, v[0].y + c2 * t1.y + c1 * t2.y);
  const double2 b1 = mul_mi<INV>(make_double2(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y));
  const double2 b2 = mul_mi<INV>(make_double2(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y));
  v[0] = make_double2(v[0].x + t1.x + t2.x, v[0].y + t1.y + t2.y);
  v[1] = make_double2(a1.x + b1.x, a1.y + b1.y);
  v[4] = make_double2(a1.x - b1.x, a1.y - b1.y);
  v[2] = make_double2(a2.x + b2.x, a2.y + b2.y);
  v[3] = make_double2(a2.x - b2.x, a2.y - b2.y);
}
template <bool INV>
__device__ __forceinline__ void g_dft7(double2 (&v)[7]) {
  const double c1 = 0.62348980185873353053, c2 = -0.22252093395631440429, c3 = -0.90096886790241912624;   // cos(2 pi k / 7)
  const double s1 = 0.78183148246802980871, s2 = 0.97492791218182360702, s3 = 0.43388373911755812048;    // sin(2 pi k / 7)
  const double2 t1 = make_double2(v[1].x + v[6].x, v[1].y + v[6].y), d1 = make_double2(v[1].x - v[6].x, v[1].y - v[6].y);
  const double2 t2 = make_double2(v[2].x + v[5].x, v[2].y + v[5].y), d2 = make_double2(v[2].x - v[5].x, v[2].y - v[5].y);
  const double2 t3 = make_double2(v[3].x + v[4].x, v[3].y + v[4].y), d3 = make_double2(v[3].x - v[4].x, v[3].y - v[4].y);
  const double2 a1 = make_double2(v[0].x + c1 * t1.x + c2 * t2.x + c3 * t3.x, v[0].y + c1 * t1.y + c2 * t2.y + c3 * t3.y);
  const double2 a2 = make_double2(v[0].x + c2 * t1.x + c3 * t2.x + c1 * t3.x, v[0].y + c2 * t1.y + c3 * t2.y + c1 * t3.y);
  const double2 a3 = make_double2(v[0].x + c3 * t1.x + c1 * t2.x + c2 * t3.x, v[0].y + c3 * t1.y + c1 * t2.y + c2 * t3.y);
  const double2 b1 = mul_mi<INV>(make_double2(s1 * d1.x + s2 * d2.x + s3 * d3.x, s1 * d1.y + s2 * d2.y + s3 * d3.y));
  const double2 b2 = mul_mi<INV>(make_double2(s2 * d1.x - s3 * d2.x - s1 * d3.x, s2 * d1.y - s3 * d2.y - s1 * d3.y));
  const double2 b3 = mul_mi<INV>(make_double2(s3 * d1.x - s1 * d2.x + s2 * d3.x, s3 * d1.y - s1 * d2.y + s2 * d3.y));
  v[0] = make_double2(v[0].x + t1.x + t2.x + t3.x, v[0].y + t1.y + t2.y + t3.y);
  v[1] = make_double2(a1.x + b1.x, a1.y + b1.y);
  v[6] = make_double2(a1.x - b1.x, a1.y - b1.y);
  v[2] = make_double2(a2.x + b2.x, a2.y + b2.y);
  v[5] = make_double2(a2.x - b2.x, a2.y - b2.y);
  v[3] = make_double2(a3.x + b3.x, a3.y + b3.y);
  v[4] = make_double2(a3.x - b3.x, a3.y - b3.y);
}
template <int R, bool INV>
__device__ __forceinline__ void g_dft(double2 (&v)[R]) {
  if constexpr (R == 2) x_dft<1, INV>(v);
  else if constexpr (R == 4) x_dft<2, INV>(v);
  else if constexpr (R == 8) x_dft<3, INV>(v);
  else if constexpr (R == 3) g_dft3<INV>(v);
  else if constexpr (R == 5) g_dft5<INV>(v);
  else g_dft7<INV>(v);
}

// One Stockham pass of radix R over `nlines` lines of length L, in place: element n of line l at
// buf[l * lstr + n * nstr].  Item (l, jj), jj < L / R: inputs n = jj + r L / R, twiddles T[(r - 1) Ns + jj mod Ns]
// (conjugated for the inverse), outputs n = (jj div Ns) Ns R + jj mod Ns + r Ns.  A thread keeps its items (at most
// MAXI: the tile holds at most 4 096 elements for 512 threads, 2 048 for 256) in registers across the barrier that
// separates the reads of the tile from its writes.
// a / b for 0 <= a < 4 096, 1 <= b <= 2 048 without an integer division (~25 instructions, three per item, next to a
// butterfly of ~40): (a + 1/2) / b stays at least 1 / (2 b) away from every integer, far more than single precision can
// be off, so the truncation is exact -- checked exhaustively over that whole range on the host
// (tests/test_oracle_kat.py::test_float_division_trick_of_the_generic_fft_passes).
__device__ __forceinline__ int g_div(int a, float inv_b) { return (int)(((float)a + 0.5f) * inv_b); }

// Line l starts at buf[(l mod lmod) + (l div lmod) * lstr_hi]: lmod = 1, lstr_hi = M for the rows of the x passes; lmod =
// nlines for a tile whose lines are its columns and column groups (the strided passes); lmod = TX, lstr_hi = L2 TX for the
// inner transforms of a long line (gen_long_inner_kernel).
// CL fixes the common layouts at compile time (the index arithmetic of an item is a third of its vector instructions):
// CL >= 0: a column tile, nlines = lmod = nstr = 2^CL, lstr_hi = 0 (shifts and masks, no division); CL = -2: rows, lmod = 1,
// nstr = 1, lstr_hi = the line length; CL = -1: general.
template <int R, bool INV, int NT, int CL>
__device__ __forceinline__ void g_pass(double2* buf, int nlines, int lmod, int lstr_hi, int nstr, int L, const double2* T,
                                       int Ns, int tid) {
  // ceil(4096 / (R * 512)) = ceil(2048 / (R * 256)) items; radix 7 holds one (two would be 112 registers and halve the
  // occupancy of every launch, whatever its radices: measured 300^3 0.57 -> 0.83 ms): the host sizes the tiles of lines
  // with a factor 7 to at most 7 items per thread-count (3 584 / 1 792 elements, GenPlan::has7)
  constexpr int MAXI = R == 7 ? 1 : (8 + R - 1) / R;
  const int LR = L / R;
  const int nitems = (CL >= 0 ? LR << (CL >= 0 ? CL : 0) : nlines * LR);
  const float inv_nl = 1.0f / (float)nlines, inv_ns = 1.0f / (float)Ns, inv_lm = 1.0f / (float)lmod;
  auto split = [&](int it, int& jj, int& lb) {   // item -> butterfly index within the line, offset of the line
    if constexpr (CL >= 0) {
      jj = it >> CL;
      lb = it & ((1 << CL) - 1);
    } else if constexpr (CL == -2) {
      jj = g_div(it, inv_nl);
      lb = (it - jj * nlines) * lstr_hi;
    } else {
      jj = g_div(it, inv_nl);
      const int l = it - jj * nlines;
      const int lh = g_div(l, inv_lm);
      lb = (l - lh * lmod) + lh * lstr_hi;
    }
  };
  auto at = [&](int lb, int n) { return CL >= 0 ? lb + (n << (CL >= 0 ? CL : 0)) : (CL == -2 ? lb + n : lb + n * nstr); };
  double2 v[MAXI][R];
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    const int it = tid + i * NT;
    if (it < nitems) {
      int jj, lb;
      split(it, jj, lb);
#pragma unroll
      for (int r = 0; r < R; ++r) v[i][r] = buf[at(lb, jj + r * LR)];
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    const int it = tid + i * NT;
    if (it < nitems) {
      int jj, lb;
      split(it, jj, lb);
      int k = 0;
      if (Ns > 1) {   // (the first pass of a transform has unit twiddles: no table reads, no products)
        k = jj - g_div(jj, inv_ns) * Ns;
#pragma unroll
        for (int r = 1; r < R; ++r) {
          double2 w = T[(r - 1) * Ns + k];
          if (INV) w.y = -w.y;
          v[i][r] = cmul(v[i][r], w);
        }
      }
      g_dft<R, INV>(v[i]);
      const int j0 = (jj - k) * R + k;
#pragma unroll
      for (int r = 0; r < R; ++r) buf[at(lb, j0 + r * Ns)] = v[i][r];
    }
  }
  __syncthreads();
}

template <bool INV, int NT, int CL = -1>
__device__ __forceinline__ void g_transform(double2* buf, int nlines, int lmod, int lstr_hi, int nstr, const GenPlan& pl,
                                            const double2* T, int tid) {
  int Ns = 1;
  for (int p = 0; p < pl.npass; ++p) {
    const int R = pl.radix[p];
    const double2* Tp = T + pl.toff[p];
    switch (R) {
      case 2: g_pass<2, INV, NT, CL>(buf, nlines, lmod, lstr_hi, nstr, pl.L, Tp, Ns, tid); break;
      case 3: g_pass<3, INV, NT, CL>(buf, nlines, lmod, lstr_hi, nstr, pl.L, Tp, Ns, tid); break;
      case 4: g_pass<4, INV, NT, CL>(buf, nlines, lmod, lstr_hi, nstr, pl.L, Tp, Ns, tid); break;
      case 5: g_pass<5, INV, NT, CL>(buf, nlines, lmod, lstr_hi, nstr, pl.L, Tp, Ns, tid); break;
      case 7: g_pass<7, INV, NT, CL>(buf, nlines, lmod, lstr_hi, nstr, pl.L, Tp, Ns, tid); break;
      default: g_pass<8, INV, NT, CL>(buf, nlines, lmod, lstr_hi, nstr, pl.L, Tp, Ns, tid); break;
    }
    Ns *= R;
  }
}

constexpr int GEN_XNT = 256;     // x passes: rows * M <= 2 048
constexpr int GEN_ANT = 512;     // strided passes: TX * L <= 4 096

// ---- GP1.  LDS: tw[M] | T[tlen] | buf[rows * M]; TG: buf only, the two tables are read where they lie in global memory
// (long lines: with the tables a 2 048-point line takes 98 KB, one workgroup = one wave per SIMD on a CU; without them
// 32 KB, five workgroups -- the tables are 64 KB shared by every workgroup and stay in the L2) -------------------------------
template <int SRC, bool TG>
__global__ __launch_bounds__(GEN_XNT) void gen_x_fwd_kernel(GenGrid g, GenPlan pl, int rows, const double2* __restrict__ tw1,
                                                            const double2* __restrict__ xtw, uint64_t seed, uint32_t real,
                                                            const double* __restrict__ noise, double2* __restrict__ X,
                                                            VgDev vg, int64_t xbs) {
  extern __shared__ __attribute__((aligned(16))) double2 gsm[];
  const int M = pl.L;
  const float inv_m = 1.0f / (float)M;   // (rows * M <= 2 048: g_div is exact)
  const double2* tw = TG ? tw1 : gsm;
  const double2* T = TG ? xtw : gsm + M;
  double2* buf = TG ? gsm : gsm + M + pl.tlen;
  const int tid = threadIdx.x;
  const int64_t nrows = (int64_t)g.n2 * g.n3;
  const int64_t row0 = (int64_t)blockIdx.x * rows;
  // blockIdx.y: member of a batch of realisations (consecutive realisation numbers, noise arrays and X buffers xbs apart)
  real += blockIdx.y;
  X += (int64_t)blockIdx.y * xbs;
  if (SRC == FF_SRC_ARRAY) noise += (int64_t)blockIdx.y * (2 * nrows * M);
  if (!TG) {
    for (int k = tid; k < M; k += GEN_XNT) gsm[k] = tw1[k];
    for (int k = tid; k < pl.tlen; k += GEN_XNT) gsm[M + k] = xtw[k];
  }
  for (int e = tid; e < rows * M; e += GEN_XNT) {
    const int row = g_div(e, inv_m), n = e - row * M;
    const int64_t grow = row0 + row;
    double2 x = make_double2(0.0, 0.0);
    if (grow < nrows) {
      const int64_t blk = grow * M + n;   // elements 2 blk, 2 blk + 1 of the realisation
      if (SRC == FF_SRC_ARRAY) {
        x = reinterpret_cast<const double2*>(noise)[blk];
      } else if (SRC == FF_SRC_COV) {
        const int i2 = (int)(grow % g.n2), i3 = (int)(grow / g.n2);
        const double zero[3] = {0.0, 0.0, 0.0};
        double a[3] = {(double)(2 * n - g.c1) * g.s1, (double)(i2 - g.c2) * g.s2, (double)(i3 - g.c3) * g.s3};
        x.x = cov_pair<3>(vg, a, zero);
        a[0] = (double)(2 * n + 1 - g.c1) * g.s1;
        x.y = cov_pair<3>(vg, a, zero);
      } else {
        philox_pair(seed, real, STREAM_UNIFORM, (uint64_t)blk, x.x, x.y);
      }
    }
    buf[e] = x;
  }
  __syncthreads();
  g_transform<false, GEN_XNT, -2>(buf, rows, 1, M, 1, pl, T, tid);
  // X[k] = ((Zk + conj(Z_{M-k})) - i w^k (Zk - conj(Z_{M-k}))) / 2 for k = 0 .. M, the pair (k, M - k) together
  const int half = M / 2;
  const float inv_h1 = 1.0f / (float)(half + 1);
  for (int t = tid; t < rows * (half + 1); t += GEN_XNT) {
    const int row = g_div(t, inv_h1), k = t - row * (half + 1);
    const int64_t grow = row0 + row;
    if (grow >= nrows) continue;
    const double2* z = buf + row * M;
    const double2 a = z[k];
    const double2 b = cconj(z[k == 0 ? 0 : M - k]);
    const double2 w = tw[k];
    const double2 sm2 = make_double2(a.x + b.x, a.y + b.y);
    const double2 pp = cmul(make_double2(a.x - b.x, a.y - b.y), w);
    double2* xr = X + grow * g.nhp;
    xr[k] = make_double2(0.5 * (sm2.x + pp.y), 0.5 * (sm2.y - pp.x));
    if (k != M - k) xr[M - k] = make_double2(0.5 * (sm2.x - pp.y), 0.5 * (-sm2.y - pp.x));
  }
}

// ---- GP5.  LDS: tw[M] | T[tlen] | buf[rows * M] (TG: as in GP1) ---------------------------------------------------------
template <bool TG>
__global__ __launch_bounds__(GEN_XNT) void gen_x_inv_kernel(GenGrid g, GenPlan pl, int rows, const double2* __restrict__ tw1,
                                                            const double2* __restrict__ xtw, const double2* __restrict__ X,
                                                            double* __restrict__ out, int64_t xbs, int64_t obs) {
  extern __shared__ __attribute__((aligned(16))) double2 gsm[];
  const int M = pl.L;
  const float inv_m = 1.0f / (float)M;   // (rows * M <= 2 048: g_div is exact)
  const double2* tw = TG ? tw1 : gsm;
  const double2* T = TG ? xtw : gsm + M;
  double2* buf = TG ? gsm : gsm + M + pl.tlen;
  const int tid = threadIdx.x;
  const int64_t nrows = (int64_t)g.n2 * g.n3;
  const int64_t row0 = (int64_t)blockIdx.x * rows;
  X += (int64_t)blockIdx.y * xbs;      // batch member: its X buffer, its output (obs doubles apart)
  out += (int64_t)blockIdx.y * obs;
  if (!TG) {
    for (int k = tid; k < M; k += GEN_XNT) gsm[k] = tw1[k];
    for (int k = tid; k < pl.tlen; k += GEN_XNT) gsm[M + k] = xtw[k];
    __syncthreads();
  }
  // Z'[k] = (Xk + conj(X_{M-k})) + i conj(w)^k (Xk - conj(X_{M-k})), k = 0 .. M - 1
  for (int e = tid; e < rows * M; e += GEN_XNT) {
    const int row = g_div(e, inv_m), k = e - row * M;
    const int64_t grow = row0 + row;
    double2 v = make_double2(0.0, 0.0);
    if (grow < nrows) {
      const double2* xr = X + grow * g.nhp;
      const double2 a = xr[k];
      const double2 bb = cconj(xr[M - k]);
      const double2 d = cmul(make_double2(a.x - bb.x, a.y - bb.y), cconj(tw[k]));   // i * d = (-d.y, d.x)
      v = make_double2(a.x + bb.x - d.y, a.y + bb.y + d.x);
    }
    buf[e] = v;
  }
  __syncthreads();
  g_transform<true, GEN_XNT, -2>(buf, rows, 1, M, 1, pl, T, tid);
  double2* o2 = reinterpret_cast<double2*>(out);
  for (int e = tid; e < rows * M; e += GEN_XNT) {
    const int row = g_div(e, inv_m), n = e - row * M;
    const int64_t grow = row0 + row;
    if (grow < nrows) o2[grow * M + n] = buf[e];
  }
}

// ---- strided passes.  MODE 0: forward, 1: inverse, 2: forward, phase with Fh, inverse.  Tile (o, t): TX columns
// t TX .. of line o of the axis; element j of the line at X + o * ostride + j * lstride.  LDS: T[tlen] | buf[L * TX],
// element (j, c) at j * TX + c.
// slab_nt > 0: only the x tiles slab_t0 .. slab_t0 + slab_nt - 1 (the slab order of the three strided passes, as in the
// power-of-two pipeline: what a pass has written is read back by the next from the memory-side cache)
template <int MODE, int TXLOG>
__global__ __launch_bounds__(GEN_ANT) void gen_axis_kernel(GenGrid g, GenPlan pl, int axis, const double2* __restrict__ atw,
                                                           int64_t ostride, int64_t lstride, double2* __restrict__ X,
                                                           const double* __restrict__ Fh, double mean, int slab_t0,
                                                           int slab_nt, int64_t xbs) {
  extern __shared__ __attribute__((aligned(16))) double2 gsm[];
  constexpr int TX = 1 << TXLOG;
  const int L = pl.L;
  double2* T = gsm;
  double2* buf = gsm + pl.tlen;
  const int tid = threadIdx.x;
  const int ntx = g.nhp >> TXLOG;
  const int tpo = slab_nt > 0 ? slab_nt : ntx;            // tiles per line of the axis in this launch
  const int t = slab_t0 + (int)(blockIdx.x % (unsigned)tpo), o = (int)(blockIdx.x / (unsigned)tpo);
  const int tile = o * ntx + t;                            // (the amplitudes are tiled over the whole buffer)
  double2* gbase = X + (int64_t)blockIdx.y * xbs + (int64_t)o * ostride + (int64_t)t * TX;
  for (int k = tid; k < pl.tlen; k += GEN_ANT) T[k] = atw[k];
  for (int e = tid; e < L * TX; e += GEN_ANT) {
    const int c = e & (TX - 1), j = e >> TXLOG;
    buf[e] = gbase[(int64_t)j * lstride + c];
  }
  __syncthreads();
  if (MODE == 1) {
    g_transform<true, GEN_ANT, TXLOG>(buf, TX, TX, 0, TX, pl, T, tid);
  } else {
    g_transform<false, GEN_ANT, TXLOG>(buf, TX, TX, 0, TX, pl, T, tid);
    if (MODE == 2) {
      // fft.jl:163: P = F exp(i angle(X)); the amplitudes in the order of the tile's elements (gen_tile_fh_kernel):
      // one contiguous, aligned run of L * TX doubles per workgroup
      const double* fh = Fh + (int64_t)tile * ((int64_t)L << TXLOG);
      for (int e = tid; e < L * TX; e += GEN_ANT) {
        const int c = e & (TX - 1), j = e >> TXLOG;
        const int kx = t * TX + c;
        if (kx < g.nh) {
          const int ky = axis == 1 ? j : o, kz = axis == 1 ? 0 : j;
          const double f = fh[e];
          const double2 x = buf[e];
          const double mag2 = x.x * x.x + x.y * x.y;
          double2 pz;
          if (mag2 > 0.0) {
            double y = __builtin_amdgcn_rsq(mag2);
            double er = fma(-(mag2 * y), y, 1.0);
            y = fma(0.5 * y, er, y);
            er = fma(-(mag2 * y), y, 1.0);
            y = fma(0.5 * y, er, y);
            const double inv = f * y;
            pz = make_double2(x.x * inv, x.y * inv);
          } else {
            pz = make_double2(f, 0.0);   // angle(0) = 0
          }
          if (kx == 0 && ky == 0 && kz == 0) pz = make_double2(mean, 0.0);   // DC <- mean
          buf[e] = pz;
        }
      }
      __syncthreads();
      g_transform<true, GEN_ANT, TXLOG>(buf, TX, TX, 0, TX, pl, T, tid);
    }
  }
  for (int e = tid; e < L * TX; e += GEN_ANT) {
    const int c = e & (TX - 1), j = e >> TXLOG;
    gbase[(int64_t)j * lstride + c] = buf[e];
  }
}

// ---- long lines: the y axis of a 2-D grid with 1 024 < n2 <= 4 096 (2 048^2, 4 096^2, 3 000 x 2 000 ...) -------------------
// A column tile of such a line does not fit the LDS, so the line is split n2 = L1 L2 (both <= 64; y = L2 a + b, frequency
// k = c + L1 d) and its transform X^[c + L1 d] = sum_b W_L2^(b d) [ W_n2^(b c) sum_a x[L2 a + b] W_L1^(a c) ] runs as
//   outer (forward)  : for NB values of b, L1-point transforms over a (rows L2 apart), times W_n2^(b c), stored in place
//                      (frequency index c where a was)
//   inner            : for NC values of c, L2-point transforms over b (L2 consecutive rows) -> frequency c + L1 d at row
//                      L2 c + d; MODE 2 continues in the same trip with the phase step and the inverse inner transforms
//   outer (inverse)  : times conj W_n2^(b c), inverse L1-point transforms over c -> natural order
// i.e. the pass GP3 of a 2-D grid becomes three trips over the buffer (five passes per realisation in all); the rows of
// every tile are 128-byte pieces exactly as in the other strided passes.  twl = exp(-2 pi i k / n2), k < n2 (global; b c <
// L2 L1 = n2, so the product indexes it directly).
struct GenLong {
  int L1, L2;      // n2 = L1 * L2
  int NB;          // values of b per outer tile (divides L2; TX * L1 * NB <= 4 096)
  int NC;          // values of c per inner tile (divides L1; TX * L2 * NC <= 4 096)
};

template <bool INV>
__global__ __launch_bounds__(GEN_ANT) void gen_long_outer_kernel(GenGrid g, GenPlan pl, GenLong lg,
                                                                 const double2* __restrict__ atw, const double2* __restrict__ twl,
                                                                 double2* __restrict__ X, int64_t xbs) {
  extern __shared__ __attribute__((aligned(16))) double2 gsm[];
  constexpr int TX = 8;
  const int L1 = lg.L1, L2 = lg.L2, NB = lg.NB;
  double2* T = gsm;
  double2* buf = gsm + pl.tlen;
  const int tid = threadIdx.x;
  const int ntx = g.nhp >> 3;
  const int t = blockIdx.x % ntx, b0 = (blockIdx.x / ntx) * NB;
  double2* gbase = X + (int64_t)blockIdx.y * xbs + (int64_t)t * TX;
  const int nel = L1 * NB * TX;
  for (int k = tid; k < pl.tlen; k += GEN_ANT) T[k] = atw[k];
  // element e = (a * NB + bb) * TX + c  <->  row L2 a + b0 + bb, column t TX + c
  for (int e = tid; e < nel; e += GEN_ANT) {
    const int c = e & 7, q = e >> 3;
    const int a = q / NB, bb = q - a * NB;
    double2 x = gbase[(int64_t)(L2 * a + b0 + bb) * g.nhp + c];
    if (INV) {   // undo the forward twiddle before the inverse transform over the frequency index a = c'
      double2 w = twl[(b0 + bb) * a];
      w.y = -w.y;
      x = cmul(x, w);
    }
    buf[e] = x;
  }
  __syncthreads();
  g_transform<INV, GEN_ANT>(buf, NB * TX, NB * TX, 0, NB * TX, pl, T, tid);
  for (int e = tid; e < nel; e += GEN_ANT) {
    const int c = e & 7, q = e >> 3;
    const int a = q / NB, bb = q - a * NB;
    double2 x = buf[e];
    if (!INV) x = cmul(x, twl[(b0 + bb) * a]);
    gbase[(int64_t)(L2 * a + b0 + bb) * g.nhp + c] = x;
  }
}

// MODE 0: forward inner transforms only (spectrum build); MODE 2: forward, phase with Fh (tiled: gen_tile_fh_long_kernel),
// inverse.  Tile: columns t TX .., rows L2 c0 .. L2 (c0 + NC) - 1 (consecutive); element e = (cc * L2 + b) * TX + col.
template <int MODE>
__global__ __launch_bounds__(GEN_ANT) void gen_long_inner_kernel(GenGrid g, GenPlan pl, GenLong lg,
                                                                 const double2* __restrict__ atw, double2* __restrict__ X,
                                                                 const double* __restrict__ Fh, double mean, int64_t xbs) {
  extern __shared__ __attribute__((aligned(16))) double2 gsm[];
  constexpr int TX = 8;
  const int L2 = lg.L2, NC = lg.NC;
  double2* T = gsm;
  double2* buf = gsm + pl.tlen;
  const int tid = threadIdx.x;
  const int ntx = g.nhp >> 3;
  const int t = blockIdx.x % ntx, c0 = (blockIdx.x / ntx) * NC;
  double2* gbase = X + (int64_t)blockIdx.y * xbs + (int64_t)L2 * c0 * g.nhp + (int64_t)t * TX;
  const int nel = NC * L2 * TX;
  for (int k = tid; k < pl.tlen; k += GEN_ANT) T[k] = atw[k];
  for (int e = tid; e < nel; e += GEN_ANT) buf[e] = gbase[(int64_t)(e >> 3) * g.nhp + (e & 7)];
  __syncthreads();
  g_transform<false, GEN_ANT>(buf, NC * TX, TX, L2 * TX, TX, pl, T, tid);
  if (MODE == 2) {
    const double* fh = Fh + (int64_t)blockIdx.x * nel;
    for (int e = tid; e < nel; e += GEN_ANT) {
      const int col = e & 7;
      if (t * TX + col < g.nh) {
        const double f = fh[e];
        const double2 x = buf[e];
        const double mag2 = x.x * x.x + x.y * x.y;
        double2 pz;
        if (mag2 > 0.0) {
          double y = __builtin_amdgcn_rsq(mag2);
          double er = fma(-(mag2 * y), y, 1.0);
          y = fma(0.5 * y, er, y);
          er = fma(-(mag2 * y), y, 1.0);
          y = fma(0.5 * y, er, y);
          const double inv = f * y;
          pz = make_double2(x.x * inv, x.y * inv);
        } else {
          pz = make_double2(f, 0.0);
        }
        if (blockIdx.x == 0 && e == 0) pz = make_double2(mean, 0.0);   // (kx, ky) = (0, 0): c = 0, d = 0, column 0
        buf[e] = pz;
      }
    }
    __syncthreads();
    g_transform<true, GEN_ANT>(buf, NC * TX, TX, L2 * TX, TX, pl, T, tid);
  }
  for (int e = tid; e < nel; e += GEN_ANT) gbase[(int64_t)(e >> 3) * g.nhp + (e & 7)] = buf[e];
}

// amplitudes in the order of gen_long_inner_kernel's tiles: dst[(cg * ntx + t) * NC L2 TX + (cc L2 + d) TX + col] =
// Fh[(c0 + cc + L1 d) nh + t TX + col]
__global__ __launch_bounds__(256) void gen_tile_fh_long_kernel(GenGrid g, GenLong lg, const double* __restrict__ Fh,
                                                               double* __restrict__ dst) {
  const int ntx = g.nhp >> 3;
  const int64_t per_tile = (int64_t)lg.NC * lg.L2 * 8;
  const int64_t total = (int64_t)(lg.L1 / lg.NC) * ntx * per_tile;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t tile = e / per_tile;
    const int w = (int)(e - tile * per_tile);
    const int col = w & 7, q = w >> 3;
    const int cc = q / lg.L2, d = q - cc * lg.L2;
    const int t = (int)(tile % ntx), cg = (int)(tile / ntx);
    const int kx = t * 8 + col, ky = cg * lg.NC + cc + lg.L1 * d;
    dst[e] = kx < g.nh ? Fh[(int64_t)ky * g.nh + kx] : 0.0;
  }
}

// 1-D grids: the phase step between GP1 and GP5 (fft.jl:163), elementwise on the nh coefficients of every member of a
// batch (blockIdx.y); Fh in its natural layout
__global__ __launch_bounds__(256) void gen_phase1d_kernel(GenGrid g, double2* __restrict__ X, const double* __restrict__ Fh,
                                                          double mean, int64_t xbs) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= g.nh) return;
  double2* x = X + (int64_t)blockIdx.y * xbs + k;
  const double2 v = *x;
  const double f = Fh[k];
  const double mag2 = v.x * v.x + v.y * v.y;
  double2 pz;
  if (mag2 > 0.0) {
    double y = __builtin_amdgcn_rsq(mag2);
    double er = fma(-(mag2 * y), y, 1.0);
    y = fma(0.5 * y, er, y);
    er = fma(-(mag2 * y), y, 1.0);
    y = fma(0.5 * y, er, y);
    const double inv = f * y;
    pz = make_double2(v.x * inv, v.y * inv);
  } else {
    pz = make_double2(f, 0.0);
  }
  if (k == 0) pz = make_double2(mean, 0.0);
  *x = pz;
}

// Fh (natural layout, the handle's state) -> the order in which the workgroups of the last-axis pass read it:
// dst[(o * ntx + t) * L * TX + j * TX + c] = Fh[(kz n2 + ky) nh + kx], kx = t TX + c (0 beyond nh), (ky, kz) = (j, 0) for
// the y axis of a 2-D grid and (o, j) for the z axis
template <int TXLOG>
__global__ __launch_bounds__(256) void gen_tile_fh_kernel(GenGrid g, int axis, int L, const double* __restrict__ Fh,
                                                          double* __restrict__ dst) {
  constexpr int TX = 1 << TXLOG;
  const int ntx = g.nhp >> TXLOG;
  const int nouter = axis == 1 ? g.n3 : g.n2;
  const int64_t per_tile = (int64_t)L << TXLOG;
  const int64_t total = (int64_t)nouter * ntx * per_tile;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t tile = e / per_tile;
    const int w = (int)(e - tile * per_tile);
    const int c = w & (TX - 1), j = w >> TXLOG;
    const int t = (int)(tile % ntx), o = (int)(tile / ntx);
    const int kx = t * TX + c;
    const int ky = axis == 1 ? j : o, kz = axis == 1 ? 0 : j;
    dst[e] = kx < g.nh ? Fh[((int64_t)kz * g.n2 + ky) * g.nh + kx] : 0.0;
  }
}

// spectrum build: Fh[idx] = sqrt(|X|) from the padded buffer (natural order), DC = 0 (fft.jl:102-103); partial sums of
// F^2 over the FULL spectrum as in fftgs_amp_kernel
// (L1 > 0: the y axis was transformed as a long line, frequency c + L1 d sits at row L2 c + d)
__global__ __launch_bounds__(256) void gen_amp_kernel(GenGrid g, const double2* __restrict__ X, double* __restrict__ Fh,
                                                      double* __restrict__ partial, int L1, int L2) {
  __shared__ double red[256];
  const int64_t NH = (int64_t)g.nh * g.n2 * g.n3;
  double acc = 0.0;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < NH; idx += (int64_t)gridDim.x * 256) {
    const int kx = (int)(idx % g.nh);
    int64_t r = idx / g.nh;
    if (L1 > 0) r = (int64_t)L2 * (r % L1) + r / L1;
    const double2 x = X[r * g.nhp + kx];
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

}  // namespace gss
