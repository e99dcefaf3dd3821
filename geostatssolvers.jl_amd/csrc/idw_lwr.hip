// IDW and LWR estimation on the exact k-NN kernel.
//
//   gss_idw_predict  <- /root/reference/src/estimation/idw.jl:111-142
//   gss_lwr_predict  <- /root/reference/src/estimation/lwr.jl:114-147
//
// Two device paths, chosen by the neighbour count the searcher would return (ui.jl:16-23):
//   * k <= 64     : K4 (k-d ordered exact k-NN) then one wave per estimation point, lane = neighbour;
//   * k == n > 64 : "all samples" (maxneighbors = nothing): no search at all, one thread per estimation
//                   point sweeping the samples staged through LDS (broadcast reads).
// LWR solves its (d+1) x (d+1) normal equations about the estimation point (same predictor, better conditioned
// than the raw coordinates the reference uses) with an unpivoted Cholesky; a non-positive pivot is reported as
// GSS_PT_SINGULAR where the reference's `\` would throw.
#include "gss_internal.h"

#include <climits>
#include <cstring>

namespace gss {

struct EstSpec {
  int method;       // 0 = IDW, 1 = LWR
  const double* wsup;   // LWR: weights supplied per (point, neighbour) instead of wkind (gss_lwr_predict_weights)
  int wkind;        // GSS_WEIGHT_*
  double exponent;  // IDW
  double wa, wp;    // LWR weight parameters
  int metric;       // GSS_METRIC_* of the search (distances entering the weights use it too)
  double mparam;
  // value columns that share the search and the weights (idw.jl:128-141 is generic over the value type: a composition's
  // log-parts, several variables on one sample set): column c of the data at z + c * ldz, of the estimates at
  // mean + c * ldm; distance / variance / status are per point
  int nz;
  int64_t ldz, ldm;
};

// ranking key of the search for a runtime metric
template <int DIM>
__device__ __forceinline__ double est_key(int metric, const double* a, const double* b, const double* ir, bool aniso) {
  switch (metric) {
    case GSS_METRIC_CITYBLOCK: return metric_key<DIM, GSS_METRIC_CITYBLOCK>(a, b, ir, aniso);
    case GSS_METRIC_CHEBYSHEV: return metric_key<DIM, GSS_METRIC_CHEBYSHEV>(a, b, ir, aniso);
    case GSS_METRIC_HAVERSINE: return metric_key<DIM, GSS_METRIC_HAVERSINE>(a, b, ir, aniso);
    default: return sqdist_nofma<DIM>(a, b, ir, aniso);
  }
}

__device__ __forceinline__ double idw_weight(double d, double d2, double e) {
  if (e == 1.0) return 1.0 / d;
  if (e == 2.0) return 1.0 / d2;
  return 1.0 / pow(d, e);
}

__device__ __forceinline__ double lwr_weight(int kind, double a, double p, double h) {
  if (kind == GSS_WEIGHT_TRICUBE) {
    const double t = 1.0 - h * h * h;
    return t * t * t;
  }
  const double hp = (p == 2.0) ? h * h : (p == 1.0 ? h : pow(h, p));
  return gss_exp(-a * hp);
}

// Weighted least squares about the estimation point.  S1 = X'WX, S2 = X'W^2 X (packed lower triangles, NP = DIM+1,
// u_0 = 1, u_a = x_a - x0_a), b = X'Wz.  mean = theta_0 with S1 theta = b; "variance" = |W X S1^-1 e_1| =
// sqrt(a' S2 a) with S1 a = e_1 (lwr.jl:139-145).  Returns false when S1 is not positive definite.
template <int NP>
__device__ __forceinline__ bool lwr_solve(const double* S1, const double* S2, const double* b, double* mean,
                                          double* var) {
  double L[NP][NP];
#pragma unroll
  for (int i = 0; i < NP; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) L[i][j] = S1[i * (i + 1) / 2 + j];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const double ajj = L[j][j];
    double d = ajj;
#pragma unroll
    for (int c = 0; c < j; ++c) d -= L[j][c] * L[j][c];
    if (!(d > 1e-13 * ajj) || !(ajj > 0.0)) ok = false;
    const double l = sqrt(d > 0.0 ? d : 1.0);
    L[j][j] = l;
#pragma unroll
    for (int i = j + 1; i < NP; ++i) {
      double s = L[i][j];
#pragma unroll
      for (int c = 0; c < j; ++c) s -= L[i][c] * L[j][c];
      L[i][j] = s / l;
    }
  }
  if (!ok) return false;
  double t[NP], a[NP];
  // theta = S1^-1 b
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    double s = b[i];
#pragma unroll
    for (int c = 0; c < i; ++c) s -= L[i][c] * t[c];
    t[i] = s / L[i][i];
  }
#pragma unroll
  for (int i = NP - 1; i >= 0; --i) {
    double s = t[i];
#pragma unroll
    for (int c = i + 1; c < NP; ++c) s -= L[c][i] * t[c];
    t[i] = s / L[i][i];
  }
  *mean = t[0];
  // a = S1^-1 e_1
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    double s = (i == 0) ? 1.0 : 0.0;
#pragma unroll
    for (int c = 0; c < i; ++c) s -= L[i][c] * a[c];
    a[i] = s / L[i][i];
  }
#pragma unroll
  for (int i = NP - 1; i >= 0; --i) {
    double s = a[i];
#pragma unroll
    for (int c = i + 1; c < NP; ++c) s -= L[c][i] * a[c];
    a[i] = s / L[i][i];
  }
  double q = 0.0;
#pragma unroll
  for (int i = 0; i < NP; ++i)
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int hi = i > j ? i : j, lo = i > j ? j : i;
      q += a[i] * S2[hi * (hi + 1) / 2 + lo] * a[j];
    }
  *var = sqrt(q > 0.0 ? q : 0.0);
  return true;
}

// ---------------------------------------------------------------------------------------------
// k <= 64: one wave per estimation point on the neighbour lists written by K4
// ---------------------------------------------------------------------------------------------
// GW = lanes per estimation point: 64 (one wave per point) or 16 when k <= 16 (four points per wave: the reductions
// are four shuffle steps instead of six and are shared by four points)
template <int GW>
__device__ __forceinline__ double grp_sum(double v) {
#pragma unroll
  for (int off = GW / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
template <int GW>
__device__ __forceinline__ double grp_max(double v) {
#pragma unroll
  for (int off = GW / 2; off >= 1; off >>= 1) {
    const double o = __shfl_xor(v, off);
    v = o > v ? o : v;
  }
  return v;
}
template <int GW>
__device__ __forceinline__ double grp_min(double v) {
#pragma unroll
  for (int off = GW / 2; off >= 1; off >>= 1) {
    const double o = __shfl_xor(v, off);
    v = o < v ? o : v;
  }
  return v;
}

template <int DIM, int GW>
__global__ __launch_bounds__(256) void est_knn_kernel(EstSpec sp, const double* __restrict__ xdata,
                                                      const double* __restrict__ z, const double* __restrict__ x0,
                                                      int64_t m, int k, int minneighbors,
                                                      const int* __restrict__ idx, const int* __restrict__ count,
                                                      int aniso, double ir0, double ir1, double ir2,
                                                      double* __restrict__ mean_out, double* __restrict__ aux_out,
                                                      uint8_t* __restrict__ status_out) {
  constexpr int PPW = 64 / GW;  // points per wave
  const int lane = threadIdx.x & 63;
  const int gl = lane % GW, grp = lane / GW;
  const int64_t p = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * PPW + grp;
  const bool valid = p < m;
  const int64_t pc = valid ? p : m - 1;
  const double NaN = __builtin_nan("");
  const int cnt = count[pc];
  const bool missing = cnt < minneighbors || cnt < 1;  // idw.jl:123-124, lwr.jl:126-127
  const double ir[3] = {ir0, ir1, ir2};
  double qc[DIM], c[DIM];
#pragma unroll
  for (int a = 0; a < DIM; ++a) qc[a] = x0[pc * DIM + a];
  const bool act = !missing && gl < cnt;
  const int i = act ? idx[pc * k + gl] : 0;
#pragma unroll
  for (int a = 0; a < DIM; ++a) c[a] = xdata[(int64_t)i * DIM + a];
  const double d2 = est_key<DIM>(sp.metric, c, qc, ir, aniso != 0);  // the key the search ranked by
  const double d = metric_dist(sp.metric, d2, sp.mparam);
  double res_aux = NaN;
  int res_status = GSS_PT_MISSING;

  if (sp.method == 0) {
    const unsigned long long zb = __ballot(act && d2 == 0.0);
    unsigned long long gz = zb;
    if (GW < 64) gz = (zb >> (grp * (GW & 63))) & ((1ull << (GW & 63)) - 1ull);
    const bool haszero = gz != 0ull;  // idw.jl:131-134: some distance is zero -> copy the first such sample
    const int jz = haszero ? __builtin_ctzll(gz) + grp * GW : lane;
    const double w = (act && !haszero) ? idw_weight(d, sp.metric == GSS_METRIC_EUCLIDEAN ? d2 : d * d, sp.exponent) : 0.0;
    const double sw = grp_sum<GW>(w);
    const double dmin = grp_min<GW>(act ? d : __builtin_huge_val());
    for (int cz = 0; cz < sp.nz; ++cz) {   // one weight vector, every value column (idw.jl:138)
      const double zi = z[(int64_t)cz * sp.ldz + i];
      const double zj = __shfl(zi, jz);
      const double swz = grp_sum<GW>(w * zi);
      if (valid && gl == 0) mean_out[(int64_t)cz * sp.ldm + p] = missing ? NaN : (haszero ? zj : swz / sw);
    }
    if (!missing) {
      res_aux = haszero ? 0.0 : dmin;  // idw.jl:139
      res_status = GSS_PT_OK;
    }
  } else {
    constexpr int NP = DIM + 1, NT = NP * (NP + 1) / 2;
    const double dmax = grp_max<GW>(act ? d : 0.0);
    const double w = act ? lwr_weight(sp.wkind, sp.wa, sp.wp, d / dmax) : 0.0;  // lwr.jl:132,136
    double u[NP];
    u[0] = 1.0;
#pragma unroll
    for (int a = 0; a < DIM; ++a) u[a + 1] = c[a] - qc[a];
    double S1[NT], S2[NT];
#pragma unroll
    for (int r = 0; r < NP; ++r) {
#pragma unroll
      for (int q = 0; q <= r; ++q) {
        const double t = w * u[r] * u[q];
        S1[r * (r + 1) / 2 + q] = grp_sum<GW>(t);
        S2[r * (r + 1) / 2 + q] = grp_sum<GW>(w * t);
      }
    }
    bool ok = !missing && (dmax > 0.0);
    double var = 0.0;
    for (int cz = 0; cz < sp.nz; ++cz) {   // the normal equations are the same for every value column
      const double zi = z[(int64_t)cz * sp.ldz + i];
      double b[NP];
#pragma unroll
      for (int r = 0; r < NP; ++r) b[r] = grp_sum<GW>(w * u[r] * zi);
      double mu = 0.0;
      ok = ok && lwr_solve<NP>(S1, S2, b, &mu, &var);
      if (valid && gl == 0) mean_out[(int64_t)cz * sp.ldm + p] = ok ? mu : NaN;
    }
    if (!missing) {
      res_aux = ok ? var : NaN;
      res_status = ok ? GSS_PT_OK : GSS_PT_SINGULAR;
    }
  }
  if (valid && gl == 0) {
    aux_out[p] = res_aux;
    status_out[p] = (uint8_t)res_status;
  }
}

// ---------------------------------------------------------------------------------------------
// k == n: every sample (inside the ball, if any) is a neighbour; one thread per estimation point
// ---------------------------------------------------------------------------------------------
constexpr int EST_TILE = 1024;

// The reference's default IDW -- every sample a neighbour (idw.jl:93), Euclidean distance, no ball, exponent 1 or 2 --
// as a dedicated kernel: thread = estimation point, and the sample index is wave-uniform, so coordinates and values
// arrive through the scalar cache into SGPRs (no LDS staging, no vector loads) and enter the VALU instructions as
// scalar operands.  Seventeen vector instructions per sample and point: three differences, the squared distance,
// its running minimum (idw.jl:137 returns the nearest distance), 1 / d from v_rsq_f64 (1 / d^2 from v_rcp_f64) with two
// Newton steps, and the two sums.  A coincident sample turns the weight into NaN (0 * inf inside the Newton step): that
// is detected once at the end, and only such points rescan the samples for the first zero distance (idw.jl:131-134).
template <int DIM, bool E1>
__global__ __launch_bounds__(256) void idw_all_fast_kernel(const double* __restrict__ xdata,
                                                           const double* __restrict__ z, int n,
                                                           const double* __restrict__ x0, int64_t m,
                                                           double* __restrict__ mean_out,
                                                           double* __restrict__ aux_out,
                                                           uint8_t* __restrict__ status_out) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = p < m;
  double qc[DIM];
#pragma unroll
  for (int a = 0; a < DIM; ++a) qc[a] = live ? x0[p * DIM + a] : 0.0;
  double sw = 0.0, swz = 0.0, dmin2 = __builtin_huge_val();
#pragma unroll 8
  for (int j = 0; j < n; ++j) {
    double d2 = 0.0;
#pragma unroll
    for (int a = 0; a < DIM; ++a) {
      const double t = xdata[(int64_t)j * DIM + a] - qc[a];
      d2 = fma(t, t, d2);
    }
    dmin2 = fmin(dmin2, d2);
    double y;
    if (E1) {
      y = __builtin_amdgcn_rsq(d2);
      const double hh = 0.5 * d2;
      y = fma(y, fma(-hh * y, y, 0.5), y);
      y = fma(y, fma(-hh * y, y, 0.5), y);
    } else {
      y = __builtin_amdgcn_rcp(d2);
      y = fma(y, fma(-d2, y, 1.0), y);
      y = fma(y, fma(-d2, y, 1.0), y);
    }
    sw += y;
    swz = fma(y, z[j], swz);
  }
  if (!live) return;
  double mu = swz / sw, dist = gss_sqrt(dmin2);
  if (!(sw < __builtin_huge_val())) {  // NaN (coincident sample) or overflow: the estimate is the first coincident value
    for (int j = 0; j < n; ++j) {
      double d2 = 0.0;
#pragma unroll
      for (int a = 0; a < DIM; ++a) {
        const double t = xdata[(int64_t)j * DIM + a] - qc[a];
        d2 = fma(t, t, d2);
      }
      if (d2 == 0.0) {
        mu = z[j];
        dist = 0.0;
        break;
      }
    }
  }
  mean_out[p] = mu;
  aux_out[p] = dist;
  status_out[p] = GSS_PT_OK;
}

// ZC = value columns carried through one sweep (1, or up to 4 of a multi-column call: sp.nz columns are served in
// chunks of ZC, each chunk sweeping the samples again -- the weights of a sweep are shared by its columns)
template <int DIM, int ZC>
__global__ __launch_bounds__(256) void est_all_kernel(EstSpec sp, const double* __restrict__ xdata,
                                                      const double* __restrict__ z, int n,
                                                      const double* __restrict__ x0, int64_t m, int minneighbors,
                                                      double r2, int use_ball, int aniso, double ir0, double ir1,
                                                      double ir2, double* __restrict__ mean_out,
                                                      double* __restrict__ aux_out, uint8_t* __restrict__ status_out) {
  __shared__ double sx[EST_TILE * DIM];
  __shared__ double sz[ZC][EST_TILE];
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = p < m;
  const double ir[3] = {ir0, ir1, ir2};
  const double NaN = __builtin_nan("");
  double qc[DIM];
#pragma unroll
  for (int a = 0; a < DIM; ++a) qc[a] = live ? x0[p * DIM + a] : 0.0;
  constexpr int NP = DIM + 1, NT = NP * (NP + 1) / 2;

  for (int c0 = 0; c0 < sp.nz; c0 += ZC) {
  const int ncz = (sp.nz - c0) < ZC ? (sp.nz - c0) : ZC;
  const double* zc = z + (int64_t)c0 * sp.ldz;
  double* mo = mean_out + (int64_t)c0 * sp.ldm;
  int cnt = 0;
  double dmax2 = 0.0, dmin2 = __builtin_huge_val();
  double sw = 0.0, swz[ZC], zzero[ZC];
#pragma unroll
  for (int c = 0; c < ZC; ++c) swz[c] = zzero[c] = 0.0;
  bool haszero = false;
  // sweep 1: IDW sums (complete) / LWR farthest neighbour
  for (int t0 = 0; t0 < n; t0 += EST_TILE) {
    const int tn = (n - t0) < EST_TILE ? (n - t0) : EST_TILE;
    __syncthreads();
    for (int e = threadIdx.x; e < tn * DIM; e += 256) sx[e] = xdata[(int64_t)t0 * DIM + e];
#pragma unroll
    for (int c = 0; c < ZC; ++c)
      if (c < ncz)
        for (int e = threadIdx.x; e < tn; e += 256) sz[c][e] = zc[(int64_t)c * sp.ldz + t0 + e];
    __syncthreads();
    if (sp.method == 0 && sp.metric == GSS_METRIC_EUCLIDEAN && (sp.exponent == 1.0 || sp.exponent == 2.0)) {
      // the reference's default (all samples, exponent 1) and its square: branch-free, 1 / d from v_rsq_f64 and
      // 1 / d^2 from v_rcp_f64 (two Newton steps each), four samples in flight
      const bool e1 = sp.exponent == 1.0;
#pragma unroll 4
      for (int j = 0; j < tn; ++j) {
        const double d2 = sqdist_nofma<DIM>(&sx[j * DIM], qc, ir, aniso != 0);
        const bool in = !use_ball || d2 <= r2;
        const bool zero = in && d2 == 0.0;
        cnt += in ? 1 : 0;
        dmin2 = (in && d2 < dmin2) ? d2 : dmin2;
#pragma unroll
        for (int c = 0; c < ZC; ++c) zzero[c] = (zero && !haszero && c < ncz) ? sz[c][j] : zzero[c];
        haszero = haszero || zero;
        double y;
        if (e1) {
          y = __builtin_amdgcn_rsq(d2);
          const double hh = 0.5 * d2;
          y = fma(y, fma(-hh * y, y, 0.5), y);
          y = fma(y, fma(-hh * y, y, 0.5), y);
        } else {
          y = __builtin_amdgcn_rcp(d2);
          y = fma(y, fma(-d2, y, 1.0), y);
          y = fma(y, fma(-d2, y, 1.0), y);
        }
        const double w = (in && !zero) ? y : 0.0;
        sw += w;
#pragma unroll
        for (int c = 0; c < ZC; ++c)
          if (c < ncz) swz[c] = fma(w, sz[c][j], swz[c]);
      }
      continue;
    }
    if (sp.method == 1 && sp.metric == GSS_METRIC_EUCLIDEAN) {  // LWR, first sweep: farthest neighbour and count
#pragma unroll 4
      for (int j = 0; j < tn; ++j) {
        const double d2 = sqdist_nofma<DIM>(&sx[j * DIM], qc, ir, aniso != 0);
        const bool in = !use_ball || d2 <= r2;
        cnt += in ? 1 : 0;
        dmax2 = (in && d2 > dmax2) ? d2 : dmax2;
      }
      continue;
    }
    for (int j = 0; j < tn; ++j) {
      const double d2 = est_key<DIM>(sp.metric, &sx[j * DIM], qc, ir, aniso != 0);
      if (use_ball && !(d2 <= r2)) continue;
      ++cnt;
      dmax2 = d2 > dmax2 ? d2 : dmax2;
      dmin2 = d2 < dmin2 ? d2 : dmin2;
      if (sp.method == 0) {
        if (d2 == 0.0) {
          if (!haszero) {
#pragma unroll
            for (int c = 0; c < ZC; ++c)
              if (c < ncz) zzero[c] = sz[c][j];
          }
          haszero = true;
        } else {
          const double dd = metric_dist(sp.metric, d2, sp.mparam);
          const double w = idw_weight(dd, sp.metric == GSS_METRIC_EUCLIDEAN ? d2 : dd * dd, sp.exponent);
          sw += w;
#pragma unroll
          for (int c = 0; c < ZC; ++c)
            if (c < ncz) swz[c] += w * sz[c][j];
        }
      }
    }
  }
  const bool enough = cnt >= minneighbors && cnt >= 1;
  if (sp.method == 0) {
    if (live) {
#pragma unroll
      for (int c = 0; c < ZC; ++c)
        if (c < ncz) mo[(int64_t)c * sp.ldm + p] = !enough ? NaN : (haszero ? zzero[c] : swz[c] / sw);
      if (c0 == 0) {
        aux_out[p] = !enough ? NaN : (haszero ? 0.0 : metric_dist(sp.metric, dmin2, sp.mparam));
        status_out[p] = enough ? GSS_PT_OK : GSS_PT_MISSING;
      }
    }
    continue;
  }
  // sweep 2 (LWR): moments with delta = d / dmax
  const double dmax = metric_dist(sp.metric, dmax2, sp.mparam);
  double S1[NT], S2[NT], b[ZC][NP];
#pragma unroll
  for (int e = 0; e < NT; ++e) S1[e] = S2[e] = 0.0;
#pragma unroll
  for (int c = 0; c < ZC; ++c)
#pragma unroll
    for (int e = 0; e < NP; ++e) b[c][e] = 0.0;
  for (int t0 = 0; t0 < n; t0 += EST_TILE) {
    const int tn = (n - t0) < EST_TILE ? (n - t0) : EST_TILE;
    __syncthreads();
    for (int e = threadIdx.x; e < tn * DIM; e += 256) sx[e] = xdata[(int64_t)t0 * DIM + e];
#pragma unroll
    for (int c = 0; c < ZC; ++c)
      if (c < ncz)
        for (int e = threadIdx.x; e < tn; e += 256) sz[c][e] = zc[(int64_t)c * sp.ldz + t0 + e];
    __syncthreads();
    const bool euclid = sp.metric == GSS_METRIC_EUCLIDEAN;
    const bool gauss_w = sp.wkind == GSS_WEIGHT_EXP && sp.wp == 2.0;  // exp(-a delta^2): no square root needed
    const double inv_dmax2 = 1.0 / dmax2;
    for (int j = 0; j < tn; ++j) {
      const double d2 = euclid ? sqdist_nofma<DIM>(&sx[j * DIM], qc, ir, aniso != 0)
                               : est_key<DIM>(sp.metric, &sx[j * DIM], qc, ir, aniso != 0);
      if (use_ball && !(d2 <= r2)) continue;
      double w;
      if (euclid && gauss_w) w = gss_exp(-sp.wa * (d2 * inv_dmax2));
      else if (euclid) w = lwr_weight(sp.wkind, sp.wa, sp.wp, gss_sqrt(d2 * inv_dmax2));
      else w = lwr_weight(sp.wkind, sp.wa, sp.wp, metric_dist(sp.metric, d2, sp.mparam) / dmax);
      double u[NP];
      u[0] = 1.0;
#pragma unroll
      for (int a = 0; a < DIM; ++a) u[a + 1] = sx[j * DIM + a] - qc[a];
#pragma unroll
      for (int r = 0; r < NP; ++r) {
        const double wu = w * u[r];
#pragma unroll
        for (int c = 0; c < ZC; ++c)
          if (c < ncz) b[c][r] += wu * sz[c][j];
#pragma unroll
        for (int q = 0; q <= r; ++q) {
          const double t = wu * u[q];
          S1[r * (r + 1) / 2 + q] += t;
          S2[r * (r + 1) / 2 + q] += w * t;
        }
      }
    }
  }
  if (live) {
    bool ok = enough && (sp.wsup != nullptr || dmax > 0.0);
    double var = 0.0;
#pragma unroll
    for (int c = 0; c < ZC; ++c) {
      if (c < ncz) {
        double mu = 0.0;
        ok = ok && lwr_solve<NP>(S1, S2, b[c], &mu, &var);
        mo[(int64_t)c * sp.ldm + p] = ok ? mu : NaN;
      }
    }
    if (c0 == 0) {
      aux_out[p] = ok ? var : NaN;
      status_out[p] = !enough ? GSS_PT_MISSING : (ok ? GSS_PT_OK : GSS_PT_SINGULAR);
    }
  }
  }
}

// The reference's default LWR -- every sample a neighbour (lwr.jl:96), Euclidean distance, no ball, weight
// exp(-a delta^2) (lwr.jl:58 has a = 3) -- with the samples as scalar operands like idw_all_fast_kernel.  Sweep 1: the
// farthest sample (lwr.jl:132 normalises by it); sweep 2: the weighted moments of lwr.jl:136-145; the weight needs no
// square root because delta^2 = d^2 / dmax^2.
template <int DIM>
__global__ __launch_bounds__(256) void lwr_all_fast_kernel(double wa, const double* __restrict__ xdata,
                                                           const double* __restrict__ z, int n,
                                                           const double* __restrict__ x0, int64_t m,
                                                           double* __restrict__ mean_out,
                                                           double* __restrict__ aux_out,
                                                           uint8_t* __restrict__ status_out) {
  constexpr int NP = DIM + 1, NT = NP * (NP + 1) / 2;
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = p < m;
  double qc[DIM];
#pragma unroll
  for (int a = 0; a < DIM; ++a) qc[a] = live ? x0[p * DIM + a] : 0.0;
  double dmax2 = 0.0;
#pragma unroll 8
  for (int j = 0; j < n; ++j) {
    double d2 = 0.0;
#pragma unroll
    for (int a = 0; a < DIM; ++a) {
      const double t = xdata[(int64_t)j * DIM + a] - qc[a];
      d2 = fma(t, t, d2);
    }
    dmax2 = fmax(dmax2, d2);
  }
  const double scale = -wa / dmax2;
  double S1[NT], S2[NT], b[NP];
#pragma unroll
  for (int e = 0; e < NT; ++e) S1[e] = S2[e] = 0.0;
#pragma unroll
  for (int e = 0; e < NP; ++e) b[e] = 0.0;
#pragma unroll 2
  for (int j = 0; j < n; ++j) {
    double u[NP];
    u[0] = 1.0;
    double d2 = 0.0;
#pragma unroll
    for (int a = 0; a < DIM; ++a) {
      u[a + 1] = xdata[(int64_t)j * DIM + a] - qc[a];
      d2 = fma(u[a + 1], u[a + 1], d2);
    }
    const double w = gss_exp_poly(scale * d2);
    const double w2 = w * w, wz = w * z[j];
#pragma unroll
    for (int r = 0; r < NP; ++r) {
      b[r] = fma(wz, u[r], b[r]);
#pragma unroll
      for (int q = 0; q <= r; ++q) {
        const double t = u[r] * u[q];
        S1[r * (r + 1) / 2 + q] = fma(w, t, S1[r * (r + 1) / 2 + q]);
        S2[r * (r + 1) / 2 + q] = fma(w2, t, S2[r * (r + 1) / 2 + q]);
      }
    }
  }
  if (!live) return;
  const double NaN = __builtin_nan("");
  double mu, var;
  const bool ok = (dmax2 > 0.0) && lwr_solve<NP>(S1, S2, b, &mu, &var);
  mean_out[p] = ok ? mu : NaN;
  aux_out[p] = ok ? var : NaN;
  status_out[p] = ok ? GSS_PT_OK : GSS_PT_SINGULAR;
}

// ---------------------------------------------------------------------------------------------
// 64 < k < n: one thread per estimation point walks its neighbour list (m x k, ascending key, written by the
// passes of the search).  Same sums, in the same (ascending-distance) order, as est_knn_kernel.
// ---------------------------------------------------------------------------------------------
template <int DIM, int ZC>
__global__ __launch_bounds__(256) void est_list_kernel(EstSpec sp, const double* __restrict__ xdata,
                                                       const double* __restrict__ z, const double* __restrict__ x0,
                                                       int64_t m, int k, int minneighbors, const int* __restrict__ idx,
                                                       const int* __restrict__ count, int aniso, double ir0, double ir1,
                                                       double ir2, double* __restrict__ mean_out,
                                                       double* __restrict__ aux_out, uint8_t* __restrict__ status_out,
                                                       int ncheck) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= m) return;
  const double ir[3] = {ir0, ir1, ir2};
  const double NaN = __builtin_nan("");
  constexpr int NP = DIM + 1, NT = NP * (NP + 1) / 2;
  double qc[DIM];
#pragma unroll
  for (int a = 0; a < DIM; ++a) qc[a] = x0[p * DIM + a];
  int cnt = count[p];
  const int* nb = idx + p * k;
  bool bad = false;
  if (ncheck > 0) {   // lists that come from the caller (gss_lwr_predict_weights): never gather through a bad entry
    cnt = cnt > k ? k : cnt;
    for (int j = 0; j < cnt; ++j) bad |= (unsigned)nb[j] >= (unsigned)ncheck;
  }
  if (bad || cnt < minneighbors || cnt < 1) {
    for (int c = 0; c < sp.nz; ++c) mean_out[(int64_t)c * sp.ldm + p] = NaN;
    aux_out[p] = NaN;
    status_out[p] = bad ? GSS_PT_SINGULAR : GSS_PT_MISSING;
    return;
  }
  for (int c0 = 0; c0 < sp.nz; c0 += ZC) {   // the columns of a chunk share one walk along the list and its weights
    const int ncz = (sp.nz - c0) < ZC ? (sp.nz - c0) : ZC;
    const double* zc = z + (int64_t)c0 * sp.ldz;
    double* mo = mean_out + (int64_t)c0 * sp.ldm;
    if (sp.method == 0) {
      double sw = 0.0, swz[ZC], dmin2 = __builtin_huge_val();
#pragma unroll
      for (int c = 0; c < ZC; ++c) swz[c] = 0.0;
      int jz = -1;
      for (int j = 0; j < cnt; ++j) {
        const int nj = nb[j];
        double xj[DIM];
#pragma unroll
        for (int a = 0; a < DIM; ++a) xj[a] = xdata[(int64_t)nj * DIM + a];
        const double d2 = est_key<DIM>(sp.metric, xj, qc, ir, aniso != 0);
        if (d2 == 0.0) {  // neighbours are sorted: the first zero distance is the first neighbour (idw.jl:131-134)
          jz = nj;
          break;
        }
        dmin2 = d2 < dmin2 ? d2 : dmin2;
        const double dd = metric_dist(sp.metric, d2, sp.mparam);
        const double w = idw_weight(dd, sp.metric == GSS_METRIC_EUCLIDEAN ? d2 : dd * dd, sp.exponent);
        sw += w;
#pragma unroll
        for (int c = 0; c < ZC; ++c)
          if (c < ncz) swz[c] += w * zc[(int64_t)c * sp.ldz + nj];
      }
#pragma unroll
      for (int c = 0; c < ZC; ++c)
        if (c < ncz) mo[(int64_t)c * sp.ldm + p] = jz >= 0 ? zc[(int64_t)c * sp.ldz + jz] : swz[c] / sw;
      if (c0 == 0) {
        aux_out[p] = jz >= 0 ? 0.0 : metric_dist(sp.metric, dmin2, sp.mparam);
        status_out[p] = GSS_PT_OK;
      }
      continue;
    }
    // LWR: delta = d / d_max with d_max the distance of the last (farthest) neighbour (lwr.jl:132)
    double dmax2 = 0.0;
    {
      const int nl = nb[cnt - 1];
      double xl[DIM];
#pragma unroll
      for (int a = 0; a < DIM; ++a) xl[a] = xdata[(int64_t)nl * DIM + a];
      dmax2 = est_key<DIM>(sp.metric, xl, qc, ir, aniso != 0);
    }
    const double dmax = metric_dist(sp.metric, dmax2, sp.mparam);
    double S1[NT], S2[NT], b[ZC][NP];
#pragma unroll
    for (int e = 0; e < NT; ++e) S1[e] = S2[e] = 0.0;
#pragma unroll
    for (int c = 0; c < ZC; ++c)
#pragma unroll
      for (int e = 0; e < NP; ++e) b[c][e] = 0.0;
    for (int j = 0; j < cnt; ++j) {
      const int nj = nb[j];
      double xj[DIM];
#pragma unroll
      for (int a = 0; a < DIM; ++a) xj[a] = xdata[(int64_t)nj * DIM + a];
      const double d2 = est_key<DIM>(sp.metric, xj, qc, ir, aniso != 0);
      const double w = sp.wsup ? sp.wsup[p * k + j]
                               : lwr_weight(sp.wkind, sp.wa, sp.wp, metric_dist(sp.metric, d2, sp.mparam) / dmax);
      double u[NP];
      u[0] = 1.0;
#pragma unroll
      for (int a = 0; a < DIM; ++a) u[a + 1] = xj[a] - qc[a];
#pragma unroll
      for (int r = 0; r < NP; ++r) {
        const double wu = w * u[r];
#pragma unroll
        for (int c = 0; c < ZC; ++c)
          if (c < ncz) b[c][r] += wu * zc[(int64_t)c * sp.ldz + nj];
#pragma unroll
        for (int q = 0; q <= r; ++q) {
          const double t = wu * u[q];
          S1[r * (r + 1) / 2 + q] += t;
          S2[r * (r + 1) / 2 + q] += w * t;
        }
      }
    }
    bool ok = sp.wsup != nullptr || dmax > 0.0;
    double var = 0.0;
#pragma unroll
    for (int c = 0; c < ZC; ++c) {
      if (c < ncz) {
        double mu = 0.0;
        ok = ok && lwr_solve<NP>(S1, S2, b[c], &mu, &var);
        mo[(int64_t)c * sp.ldm + p] = ok ? mu : NaN;
      }
    }
    if (c0 == 0) {
      aux_out[p] = ok ? var : NaN;
      status_out[p] = ok ? GSS_PT_OK : GSS_PT_SINGULAR;
    }
  }
}

static int32_t est_local_dev(const EstSpec& sp, const double* xdata, const double* z, int64_t n, int dim,
                             const double* x0, int64_t m, int k, int minneighbors, double radius,
                             const double* inv_radii_host, double* mean, double* aux, uint8_t* status,
                             hipStream_t s, HostPipe* pipe = nullptr /* k <= 64 only */) {
  const int use_ball = (radius >= 0.0 || inv_radii_host != nullptr) ? 1 : 0;
  const int aniso = inv_radii_host != nullptr ? 1 : 0;
  const double r2 = aniso ? 1.0 : radius * radius;
  double ir[3] = {1.0, 1.0, 1.0};
  if (aniso)
    for (int a = 0; a < dim; ++a) ir[a] = inv_radii_host[a];
  const char* pname = sp.method == 0 ? "idw" : "lwr";

  if (k > 64 && (int64_t)k < n) {
    // 65 .. n - 1 neighbours (ui.jl:16-23 accepts any count): the search runs in passes of 64, the estimator walks
    // the lists with one thread per point
    const int64_t chunk = k > 512 ? (1 << 16) : (1 << 19);
    const bool hav = sp.metric == GSS_METRIC_HAVERSINE;   // no box bounds for the indexed search: exhaustive passes
    KnnIndex ix;
    if (!hav) GSS_TRY(knn_index_build_from_device(xdata, n, dim, &ix, s));
    DevBuf idx_s, cnt_s;
    GSS_TRY(idx_s.alloc(sizeof(int) * (size_t)((m < chunk ? m : chunk) * k)));
    GSS_TRY(cnt_s.alloc(sizeof(int) * (size_t)(m < chunk ? m : chunk)));
    for (int64_t off = 0; off < m; off += chunk) {
      const int64_t mv = (m - off) < chunk ? (m - off) : chunk;
      {
        ProfScope ps("knn", s);
        if (hav)
          GSS_TRY(knn_search_dev(xdata, n, dim, x0 + off * dim, mv, k, radius, inv_radii_host, idx_s.as<int>(),
                                 cnt_s.as<int>(), s, sp.metric));
        else
          GSS_TRY(knn_search_indexed_any(ix, xdata, x0 + off * dim, mv, k, radius, inv_radii_host, idx_s.as<int>(),
                                         cnt_s.as<int>(), s, sp.metric));
      }
      ProfScope pl(pname, s);
      const dim3 grid((unsigned)((mv + 255) / 256));
#define GSS_EST_LIST_ARGS sp, xdata, z, x0 + off * dim, mv, k, minneighbors, idx_s.as<int>(), cnt_s.as<int>(), aniso, \
                          ir[0], ir[1], ir[2], mean + off, aux + off, status + off, 0
#define GSS_EST_LIST(D)                                                                                      \
  if (sp.nz > 1) hipLaunchKernelGGL((est_list_kernel<D, 4>), grid, dim3(256), 0, s, GSS_EST_LIST_ARGS);      \
  else hipLaunchKernelGGL((est_list_kernel<D, 1>), grid, dim3(256), 0, s, GSS_EST_LIST_ARGS)
      switch (dim) {
        case 1: GSS_EST_LIST(1); break;
        case 2: GSS_EST_LIST(2); break;
        default: GSS_EST_LIST(3); break;
      }
#undef GSS_EST_LIST
#undef GSS_EST_LIST_ARGS
      GSS_HIP(hipGetLastError());
    }
    GSS_HIP(hipStreamSynchronize(s));  // scratch and the search index are released on return
    return GSS_OK;
  }
  if (k > 64) {
    ProfScope ps(pname, s);
    dim3 grid((unsigned)((m + 255) / 256));
    // (the two dedicated kernels carry one value column; several columns share the sweeps of the general kernel)
    if (sp.nz == 1 && sp.method == 0 && sp.metric == GSS_METRIC_EUCLIDEAN && !use_ball &&
        (sp.exponent == 1.0 || sp.exponent == 2.0) && minneighbors <= n) {
#define GSS_IDW_FAST(D)                                                                                               \
  if (sp.exponent == 1.0)                                                                                            \
    hipLaunchKernelGGL((idw_all_fast_kernel<D, true>), grid, dim3(256), 0, s, xdata, z, (int)n, x0, m, mean, aux,     \
                       status);                                                                                      \
  else                                                                                                               \
    hipLaunchKernelGGL((idw_all_fast_kernel<D, false>), grid, dim3(256), 0, s, xdata, z, (int)n, x0, m, mean, aux,    \
                       status)
      switch (dim) {
        case 1: GSS_IDW_FAST(1); break;
        case 2: GSS_IDW_FAST(2); break;
        default: GSS_IDW_FAST(3); break;
      }
#undef GSS_IDW_FAST
      GSS_HIP(hipGetLastError());
      return GSS_OK;
    }
    if (sp.nz == 1 && sp.method == 1 && sp.metric == GSS_METRIC_EUCLIDEAN && !use_ball && sp.wkind == GSS_WEIGHT_EXP &&
        sp.wp == 2.0 && minneighbors <= n) {
      switch (dim) {
        case 1: hipLaunchKernelGGL((lwr_all_fast_kernel<1>), grid, dim3(256), 0, s, sp.wa, xdata, z, (int)n, x0, m, mean, aux, status); break;
        case 2: hipLaunchKernelGGL((lwr_all_fast_kernel<2>), grid, dim3(256), 0, s, sp.wa, xdata, z, (int)n, x0, m, mean, aux, status); break;
        default: hipLaunchKernelGGL((lwr_all_fast_kernel<3>), grid, dim3(256), 0, s, sp.wa, xdata, z, (int)n, x0, m, mean, aux, status); break;
      }
      GSS_HIP(hipGetLastError());
      return GSS_OK;
    }
#define GSS_EST_ALL_ARGS sp, xdata, z, (int)n, x0, m, minneighbors, r2, use_ball, aniso, ir[0], ir[1], ir[2], mean, \
                         aux, status
#define GSS_EST_ALL(D)                                                                                     \
  if (sp.nz > 1) hipLaunchKernelGGL((est_all_kernel<D, 4>), grid, dim3(256), 0, s, GSS_EST_ALL_ARGS);      \
  else hipLaunchKernelGGL((est_all_kernel<D, 1>), grid, dim3(256), 0, s, GSS_EST_ALL_ARGS)
    switch (dim) {
      case 1: GSS_EST_ALL(1); break;
      case 2: GSS_EST_ALL(2); break;
      default: GSS_EST_ALL(3); break;
    }
#undef GSS_EST_ALL
#undef GSS_EST_ALL_ARGS
    GSS_HIP(hipGetLastError());
    return GSS_OK;
  }

  const bool piped = pipe && pipe->on;   // host arrays of the domain arrive and leave piece by piece (gss_internal.h)
  const int64_t chunk = piped ? HostPipe::PIECE : (1 << 20);
  KnnIndex ix;  // k-d ordered batches + boxes, built once per call (Euclidean / Mahalanobis search only)
  const bool use_index = sp.metric != GSS_METRIC_HAVERSINE;
  if (use_index) GSS_TRY(knn_index_build_from_device(xdata, n, dim, &ix, s));
  DevBuf idx_s, cnt_s;
  GSS_TRY(idx_s.alloc(sizeof(int) * (size_t)((m < chunk ? m : chunk) * k)));
  GSS_TRY(cnt_s.alloc(sizeof(int) * (size_t)(m < chunk ? m : chunk)));
  for (int64_t off = 0; off < m; off += chunk) {
    const int64_t mv = (m - off) < chunk ? (m - off) : chunk;
    if (piped) GSS_TRY(pipe->fetch(off, mv, s));
    {
      ProfScope ps("knn", s);
      if (use_index)
        GSS_TRY(knn_search_indexed(ix, x0 + off * dim, mv, k, radius, inv_radii_host, idx_s.as<int>(),
                                   cnt_s.as<int>(), s, nullptr, nullptr, nullptr, sp.metric));
      else
        GSS_TRY(knn_search_dev(xdata, n, dim, x0 + off * dim, mv, k, radius, inv_radii_host, idx_s.as<int>(),
                               cnt_s.as<int>(), s, sp.metric));
    }
    ProfScope pl(pname, s);
#define GSS_EST_KNN_ARGS sp, xdata, z, x0 + off * dim, mv, k, minneighbors, idx_s.as<int>(), cnt_s.as<int>(), aniso, \
                         ir[0], ir[1], ir[2], mean + off, aux + off, status + off
    if (k <= 16) {  // sixteen lanes per point, sixteen points per workgroup
      dim3 grid((unsigned)((mv + 15) / 16));
      switch (dim) {
        case 1: hipLaunchKernelGGL((est_knn_kernel<1, 16>), grid, dim3(256), 0, s, GSS_EST_KNN_ARGS); break;
        case 2: hipLaunchKernelGGL((est_knn_kernel<2, 16>), grid, dim3(256), 0, s, GSS_EST_KNN_ARGS); break;
        default: hipLaunchKernelGGL((est_knn_kernel<3, 16>), grid, dim3(256), 0, s, GSS_EST_KNN_ARGS); break;
      }
    } else {
      dim3 grid((unsigned)((mv + 3) / 4));
      switch (dim) {
        case 1: hipLaunchKernelGGL((est_knn_kernel<1, 64>), grid, dim3(256), 0, s, GSS_EST_KNN_ARGS); break;
        case 2: hipLaunchKernelGGL((est_knn_kernel<2, 64>), grid, dim3(256), 0, s, GSS_EST_KNN_ARGS); break;
        default: hipLaunchKernelGGL((est_knn_kernel<3, 64>), grid, dim3(256), 0, s, GSS_EST_KNN_ARGS); break;
      }
    }
#undef GSS_EST_KNN_ARGS
    GSS_HIP(hipGetLastError());
    if (piped) GSS_TRY(pipe->deliver(off, mv, s));
  }
  if (piped) GSS_TRY(pipe->finish(s));
  GSS_HIP(hipStreamSynchronize(s));  // scratch and the search index are released on return
  return GSS_OK;
}

static int32_t est_predict(EstSpec sp, const double* xdata, const double* z, int64_t n, int32_t dim, int32_t nz,
                           const double* xdom, int64_t m, int32_t k, int32_t minneighbors, double radius,
                           const double* inv_radii, double* mean, double* aux, uint8_t* status, int32_t mem,
                           void* stream) {
  GSS_REQUIRE(nz >= 1 && nz <= 4096, "%d value columns: 1 .. 4096", nz);
  sp.nz = nz;
  sp.ldz = n;
  sp.ldm = m;
  GSS_REQUIRE(n >= 1 && n < INT_MAX, "estimation requires data");  // idw.jl:95
  GSS_REQUIRE(dim >= 1 && dim <= 3, "dim = %d outside 1..3", dim);
  GSS_REQUIRE(k >= 1 && k <= n, "maxneighbors %d outside 1..%lld (searcher_ui clamps it, ui.jl:18-20)", k,
              (long long)n);
  GSS_REQUIRE(minneighbors <= k, "invalid min/max number of neighbors");  // idw.jl:97, lwr.jl:99
  GSS_TRY(check_metric(sp.metric, sp.mparam, dim, radius, inv_radii));
  GSS_REQUIRE(m >= 0 && (m == 0 || (xdata && z && xdom && mean && aux)), "NULL array");
  if (m == 0) return GSS_OK;
  hipStream_t s = to_stream(stream);
  Staged sxd, sz, sx, smean, saux, sstat;
  GSS_TRY(sxd.in(xdata, sizeof(double) * n * dim, mem, s));
  GSS_TRY(sz.in(z, sizeof(double) * n * nz, mem, s));
  HostPipe pipe;   // host arrays of the domain: in and out piece by piece beside the computation (k <= 64, one column)
  GSS_TRY(pipe.begin(k <= 64 && nz == 1 ? mem : GSS_MEM_DEVICE, m, s));
  if (pipe.on) GSS_TRY(sx.out(const_cast<double*>(xdom), sizeof(double) * m * dim, mem));   // device scratch only
  else GSS_TRY(sx.in(xdom, sizeof(double) * m * dim, mem, s));
  GSS_TRY(smean.out(mean, sizeof(double) * m * nz, mem));
  GSS_TRY(saux.out(aux, sizeof(double) * m, mem));
  DevBuf st_own;
  uint8_t* st = nullptr;
  if (status) {
    GSS_TRY(sstat.out(status, (size_t)m, mem));
    st = sstat.as<uint8_t>();
  } else {
    GSS_TRY(st_own.alloc((size_t)m));
    st = st_own.as<uint8_t>();
  }
  if (pipe.on) {
    pipe.add_in(xdom, sx.p, sizeof(double) * dim);
    pipe.add_out(mean, smean.p, sizeof(double));
    pipe.add_out(aux, saux.p, sizeof(double));
    pipe.add_out(status, status ? sstat.p : nullptr, 1);
  }
  GSS_TRY(est_local_dev(sp, sxd.as<double>(), sz.as<double>(), n, dim, sx.as<double>(), m, k, minneighbors, radius,
                        inv_radii, smean.as<double>(), saux.as<double>(), st, s, &pipe));
  if (pipe.on) return GSS_OK;   // everything is home (est_local_dev ends with pipe.finish and a synchronisation)
  GSS_TRY(smean.back(mean, sizeof(double) * m * nz, mem, s));
  GSS_TRY(saux.back(aux, sizeof(double) * m, mem, s));
  if (status) GSS_TRY(sstat.back(status, (size_t)m, mem, s));
  if (!status) GSS_HIP(hipStreamSynchronize(s));  // st_own is released on return
  return GSS_OK;
}

}  // namespace gss

using namespace gss;

extern "C" {

int32_t gss_idw_predict_cols(const double* xdata, const double* z, int64_t n, int32_t dim, int32_t nz,
                             const double* xdom, int64_t m, int32_t k, int32_t minneighbors, double radius,
                             const double* inv_radii, int32_t metric, double metric_param, double exponent, double* mean,
                             double* dist, uint8_t* status, int32_t mem, void* stream) {
  GSS_ENTRY();
  GSS_REQUIRE(exponent > 0.0, "exponent must be positive");  // idw.jl:96
  EstSpec sp;
  std::memset(&sp, 0, sizeof(sp));
  sp.method = 0;
  sp.exponent = exponent;
  sp.metric = metric;
  sp.mparam = metric_param;
  return est_predict(sp, xdata, z, n, dim, nz, xdom, m, k, minneighbors, radius, inv_radii, mean, dist, status, mem,
                     stream);
}

int32_t gss_idw_predict(const double* xdata, const double* z, int64_t n, int32_t dim, const double* xdom, int64_t m,
                        int32_t k, int32_t minneighbors, double radius, const double* inv_radii, int32_t metric,
                        double metric_param, double exponent, double* mean, double* dist, uint8_t* status, int32_t mem,
                        void* stream) {
  return gss_idw_predict_cols(xdata, z, n, dim, 1, xdom, m, k, minneighbors, radius, inv_radii, metric, metric_param,
                              exponent, mean, dist, status, mem, stream);
}

int32_t gss_lwr_predict_weights(const double* xdata, const double* z, int64_t n, int32_t dim, const double* xdom, int64_t m,
                                int32_t k, int32_t minneighbors, const int32_t* idx, const int32_t* count,
                                const double* weights, double* mean, double* var, uint8_t* status, int32_t mem,
                                void* stream) {
  GSS_ENTRY();
  GSS_REQUIRE(n >= 1 && n < INT_MAX && dim >= 1 && dim <= 3 && k >= 1 && k <= n, "gss_lwr_predict_weights: bad sizes");
  GSS_REQUIRE(m >= 0 && (m == 0 || (xdata && z && xdom && idx && count && weights && mean && var)), "NULL array");
  if (m == 0) return GSS_OK;
  hipStream_t s = to_stream(stream);
  Staged sxd, sz, sx, si, sc, sw, smean, svar, sstat;
  GSS_TRY(sxd.in(xdata, sizeof(double) * n * dim, mem, s));
  GSS_TRY(sz.in(z, sizeof(double) * n, mem, s));
  GSS_TRY(sx.in(xdom, sizeof(double) * m * dim, mem, s));
  GSS_TRY(si.in(idx, sizeof(int32_t) * (size_t)(m * k), mem, s));
  GSS_TRY(sc.in(count, sizeof(int32_t) * (size_t)m, mem, s));
  GSS_TRY(sw.in(weights, sizeof(double) * (size_t)(m * k), mem, s));
  GSS_TRY(smean.out(mean, sizeof(double) * m, mem));
  GSS_TRY(svar.out(var, sizeof(double) * m, mem));
  DevBuf st_own;
  uint8_t* st = nullptr;
  if (status) {
    GSS_TRY(sstat.out(status, (size_t)m, mem));
    st = sstat.as<uint8_t>();
  } else {
    GSS_TRY(st_own.alloc((size_t)m));
    st = st_own.as<uint8_t>();
  }
  EstSpec sp;
  std::memset(&sp, 0, sizeof(sp));
  sp.method = 1;
  sp.wsup = sw.as<double>();
  sp.nz = 1;
  sp.ldz = n;
  sp.ldm = m;
  const dim3 grid((unsigned)((m + 255) / 256));
#define GSS_LWRW_ARGS sp, sxd.as<double>(), sz.as<double>(), sx.as<double>(), m, (int)k, (int)minneighbors, si.as<int>(), \
                      sc.as<int>(), 0, 1.0, 1.0, 1.0, smean.as<double>(), svar.as<double>(), st, (int)n
  switch (dim) {
    case 1: hipLaunchKernelGGL((est_list_kernel<1, 1>), grid, dim3(256), 0, s, GSS_LWRW_ARGS); break;
    case 2: hipLaunchKernelGGL((est_list_kernel<2, 1>), grid, dim3(256), 0, s, GSS_LWRW_ARGS); break;
    default: hipLaunchKernelGGL((est_list_kernel<3, 1>), grid, dim3(256), 0, s, GSS_LWRW_ARGS); break;
  }
#undef GSS_LWRW_ARGS
  GSS_HIP(hipGetLastError());
  GSS_TRY(smean.back(mean, sizeof(double) * m, mem, s));
  GSS_TRY(svar.back(var, sizeof(double) * m, mem, s));
  if (status) GSS_TRY(sstat.back(status, (size_t)m, mem, s));
  GSS_HIP(hipStreamSynchronize(s));   // the staged copies are released on return
  return GSS_OK;
}

int32_t gss_lwr_predict(const double* xdata, const double* z, int64_t n, int32_t dim, const double* xdom, int64_t m,
                        int32_t k, int32_t minneighbors, double radius, const double* inv_radii, int32_t metric,
                        double metric_param, int32_t weight_kind, double weight_a, double weight_p, double* mean,
                        double* var, uint8_t* status, int32_t mem, void* stream) {
  return gss_lwr_predict_cols(xdata, z, n, dim, 1, xdom, m, k, minneighbors, radius, inv_radii, metric, metric_param,
                              weight_kind, weight_a, weight_p, mean, var, status, mem, stream);
}

int32_t gss_lwr_predict_cols(const double* xdata, const double* z, int64_t n, int32_t dim, int32_t nz,
                             const double* xdom, int64_t m, int32_t k, int32_t minneighbors, double radius,
                             const double* inv_radii, int32_t metric, double metric_param, int32_t weight_kind,
                             double weight_a, double weight_p, double* mean, double* var, uint8_t* status, int32_t mem,
                             void* stream) {
  GSS_ENTRY();
  GSS_REQUIRE(weight_kind == GSS_WEIGHT_EXP || weight_kind == GSS_WEIGHT_TRICUBE, "unknown weight function %d",
              weight_kind);
  GSS_REQUIRE(weight_kind != GSS_WEIGHT_EXP || weight_p > 0.0, "weight exponent must be positive");
  EstSpec sp;
  std::memset(&sp, 0, sizeof(sp));
  sp.method = 1;
  sp.wkind = weight_kind;
  sp.wa = weight_a;
  sp.wp = weight_p;
  sp.metric = metric;
  sp.mparam = metric_param;
  return est_predict(sp, xdata, z, n, dim, nz, xdom, m, k, minneighbors, radius, inv_radii, mean, var, status, mem,
                     stream);
}

}  // extern "C"
