// K4: exact k-nearest-neighbour search by brute force, deterministic ordering.
// Replaces `search!(neighbors, center, searcher)` (/root/reference/src/estimation/krig.jl:210) with
// the KNearestSearch / KBallSearch built at /root/reference/src/ui.jl:27,30 ([DEP] Meshes ->
// NearestNeighbors KD-tree), and the 1-NN lookups of fft.jl:129-132 and initbuff (lu.jl:86).
//
// Contract: neighbours are ranked by ascending (d2, index) where d2 is the FP64
// squared distance accumulated in dimension order with one rounding per operation (no FMA), so
// the oracle reproduces the indices bit for bit.  With a ball only d2 <= r^2 qualifies.
//
// Mapping: one wave owns Q = 4 query points and keeps each query's current best-k list sorted
// across its lanes (lane l holds the l-th nearest so far; k <= 64).  Data points are staged in LDS
// as structure-of-arrays tiles; every 64-candidate batch is tested against the four thresholds with
// one ballot each, and only qualifying candidates pay for an insertion (ballot + lane shift).
#include "gss_internal.h"

#include <algorithm>
#include <future>
#include <climits>
#include <cstdlib>
#include <utility>
#include <vector>

namespace gss {

constexpr int KNN_Q = 4;
constexpr int KNN_TILE = 2048;
constexpr int KNN_SORT_MIN = 12;   // candidates per batch from which sort + merge beats serial insertion ...
constexpr int KNN_SORT_MIN_K = 8;  // ... for lists of at least this length (measured, tools/knn_sweep.py)

__device__ __forceinline__ bool key_less(double ad, int ai, double bd, int bi) {
  return ad < bd || (ad == bd && ai < bi);
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// v of lane (l ^ stride).  Strides 1, 2, 4, 8 stay inside a 16-lane row and are DPP moves (quad_perm, row_shl / row_shr
// by bank, row_ror: one or two VALU instructions per dword, no LDS round trip); 16 and 32 cross rows: ds_bpermute.
// The kernels below call this with every lane of the wave active.
__device__ __forceinline__ int xor_shfl_b32(int v, int stride) {
  switch (stride) {
    case 1: return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false);   // quad_perm:[1,0,3,2]
    case 2: return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false);   // quad_perm:[2,3,0,1]
    case 4: {
      const int r = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xf, 0x5, false);  // row_shl:4 into banks 0, 2
      return __builtin_amdgcn_update_dpp(r, v, 0x114, 0xf, 0xa, false);         // row_shr:4 into banks 1, 3
    }
    case 8: return __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false);  // row_ror:8
    default: return __shfl_xor(v, stride);
  }
}
__device__ __forceinline__ double xor_shfl_f64(double v, int stride) {
  return __hiloint2double(xor_shfl_b32(__double2hiint(v), stride), xor_shfl_b32(__double2loint(v), stride));
}
// v of lane l - 1 (lane 0 keeps its own): wave_shr:1
__device__ __forceinline__ int shfl_up1_b32(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ double shfl_up1_f64(double v) {
  return __hiloint2double(shfl_up1_b32(__double2hiint(v)), shfl_up1_b32(__double2loint(v)));
}

__device__ __forceinline__ double wave_min_f64(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const double o = xor_shfl_f64(v, off);
    v = o < v ? o : v;
  }
  return v;
}

// one compare-exchange stage of a bitonic network on (d, i) keys: lanes `stride` apart, ascending when `up`
__device__ __forceinline__ void bitonic_cx(double& d, int& i, int lane, int stride, bool up) {
  const double od = xor_shfl_f64(d, stride);
  const int oi = xor_shfl_b32(i, stride);
  const bool lower = (lane & stride) == 0;
  // keys are distinct (unique indices) except for two empty slots, where the choice does not matter: "mine is less"
  // is the negation of "the other is less", so one comparison serves both directions
  const bool take = key_less(od, oi, d, i) == (lower == up);
  if (take) {
    d = od;
    i = oi;
  }
}

// full ascending sort of 64 keys held one per lane
__device__ __forceinline__ void bitonic_sort64(double& d, int& i, int lane) {
#pragma unroll
  for (int size = 2; size <= 64; size <<= 1) {
    const bool up = (lane & size) == 0;
#pragma unroll
    for (int stride = size >> 1; stride >= 1; stride >>= 1) bitonic_cx(d, i, lane, stride, up);
  }
}

// list (ascending, one key per lane) <- the 64 smallest keys of list U batch, ascending; batch is consumed
__device__ __forceinline__ void bitonic_merge64(double& ld, int& li, double bd, int bi, int lane) {
  const double rd = __shfl(bd, 63 - lane);
  const int ri = __shfl(bi, 63 - lane);
  if (key_less(rd, ri, ld, li)) {
    ld = rd;
    li = ri;
  }
#pragma unroll
  for (int stride = 32; stride >= 1; stride >>= 1) bitonic_cx(ld, li, lane, stride, true);
}

template <int DIM, int METRIC>
__global__ __launch_bounds__(256) void knn_kernel(const double* __restrict__ xdata, int n,
                                                  const double* __restrict__ centers, int64_t m, int k, double r2,
                                                  int use_ball, int aniso, double ir0, double ir1, double ir2,
                                                  int* __restrict__ idx_out, int* __restrict__ count_out,
                                                  const double* __restrict__ lowd, const int* __restrict__ lowi) {
  // lowd / lowi (may be NULL): per query, only candidates whose key (distance, index) lies strictly above this one
  // take part -- the passes of 64 of a search for more than 64 neighbours
  __shared__ double tile[3][KNN_TILE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int64_t qbase = ((int64_t)blockIdx.x * 4 + wave) * KNN_Q;
  const double ir[3] = {ir0, ir1, ir2};

  double qc[KNN_Q][DIM];
  bool qvalid[KNN_Q];
#pragma unroll
  for (int q = 0; q < KNN_Q; ++q) {
    const int64_t p = qbase + q;
    qvalid[q] = p < m;
    const int64_t pc = qvalid[q] ? p : m - 1;
#pragma unroll
    for (int a = 0; a < DIM; ++a) qc[q][a] = centers[pc * DIM + a];
  }
  const double INF = __longlong_as_double(0x7ff0000000000000LL);
  double ld[KNN_Q], tau_d[KNN_Q], low_d[KNN_Q];
  int li[KNN_Q], tau_i[KNN_Q], low_i[KNN_Q];
#pragma unroll
  for (int q = 0; q < KNN_Q; ++q) {
    ld[q] = INF;
    li[q] = INT_MAX;
    tau_d[q] = INF;
    tau_i[q] = INT_MAX;
    const int64_t pc = qvalid[q] ? qbase + q : m - 1;
    low_d[q] = lowd ? lowd[pc] : -1.0;
    low_i[q] = lowd ? lowi[pc] : -1;
  }

  for (int t0 = 0; t0 < n; t0 += KNN_TILE) {
    const int tn = (n - t0) < KNN_TILE ? (n - t0) : KNN_TILE;
    __syncthreads();
    for (int e = tid; e < tn * DIM; e += 256) tile[e % DIM][e / DIM] = xdata[(int64_t)t0 * DIM + e];
    __syncthreads();
    for (int b = 0; b < tn; b += 64) {
      const int j = b + lane;
      const bool valid = j < tn;
      double c[DIM];
#pragma unroll
      for (int a = 0; a < DIM; ++a) c[a] = valid ? tile[a][j] : 0.0;
      const int gidx = t0 + j;
#pragma unroll
      for (int q = 0; q < KNN_Q; ++q) {
        const double d2 = metric_key<DIM, METRIC>(c, qc[q], ir, aniso != 0);
        const bool qual = valid && (!use_ball || d2 <= r2) && key_less(d2, gidx, tau_d[q], tau_i[q]) &&
                          key_less(low_d[q], low_i[q], d2, gidx);
        unsigned long long mask = __ballot(qual);
        while (mask) {
          const int src = __builtin_ctzll(mask);
          mask &= mask - 1;
          const double cd = readlane_f64(d2, src);
          const int ci = __builtin_amdgcn_readlane(gidx, src);
          if (!key_less(cd, ci, tau_d[q], tau_i[q])) continue;  // an earlier insertion tightened tau
          const int pos = __popcll(__ballot(key_less(ld[q], li[q], cd, ci)));
          const double up_d = shfl_up1_f64(ld[q]);
          const int up_i = shfl_up1_b32(li[q]);
          if (lane > pos) {
            ld[q] = up_d;
            li[q] = up_i;
          } else if (lane == pos) {
            ld[q] = cd;
            li[q] = ci;
          }
          tau_d[q] = readlane_f64(ld[q], k - 1);
          tau_i[q] = __builtin_amdgcn_readlane(li[q], k - 1);
        }
      }
    }
  }

#pragma unroll
  for (int q = 0; q < KNN_Q; ++q) {
    const int64_t p = qbase + q;
    const bool has = lane < k && li[q] != INT_MAX;
    const int cnt = __popcll(__ballot(has));
    if (p < m) {
      if (lane < k) idx_out[p * k + lane] = has ? li[q] : -1;
      if (lane == 0 && count_out) count_out[p] = cnt;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Pruned search.  The data are put in a balanced k-d order once (host, O(n log n), like the KD-tree build the
// reference pays in Meshes/NearestNeighbors) and cut into batches of 64 consecutive points with bounding boxes.  One wave
// owns one query: lane b first bounds the distance to batch b (min squared distance to its box, accumulated with
// the same rounded operations as the point distance, hence never larger than the distance to any point inside);
// a second level of boxes covers groups of 64 batches.  Groups, and batches inside an opened group, are visited
// nearest box first, and a box whose bound exceeds the current k-th distance is never opened.  A batch in which
// many candidates beat the current k-th key is sorted and merged into the list with bitonic networks, otherwise
// candidates are inserted one at a time.  Results are identical to the brute-force kernel: the ranking key is
// still (d2, original index).
// ---------------------------------------------------------------------------------------------
// Balanced k-d ordering: the index range [lo, hi) of `perm` is split at a multiple of `unit` points (4096 while a
// range holds more than one group of 64 batches, 64 below) by the median along the widest axis of its bounding
// box, recursively, so that every batch of 64 consecutive points -- and every group of 64 consecutive batches -- is
// a subtree: compact, nearly cubic, non-overlapping boxes.  (A Morton sort leaves batches that straddle the
// curve's jumps and therefore have large boxes: 16 batches were opened per query at k = 16 instead of 6-7.)
struct KdPoint {
  double c[3];
  int32_t idx;
};

static void kd_order(KdPoint* pts, int dim, int64_t lo, int64_t hi, int depth) {
  const int64_t count = hi - lo;
  if (count <= 64) return;
  const int64_t unit = count > 4096 ? 4096 : 64;
  const int64_t units = (count + unit - 1) / unit;
  const int64_t left = unit * ((units + 1) / 2);
  double bl[3], bh[3];
  for (int a = 0; a < dim; ++a) bl[a] = bh[a] = pts[lo].c[a];
  for (int64_t i = lo + 1; i < hi; ++i)
    for (int a = 0; a < dim; ++a) {
      const double v = pts[i].c[a];
      bl[a] = v < bl[a] ? v : bl[a];
      bh[a] = v > bh[a] ? v : bh[a];
    }
  int axis = 0;
  for (int a = 1; a < dim; ++a)
    if (bh[a] - bl[a] > bh[axis] - bl[axis]) axis = a;
  std::nth_element(pts + lo, pts + lo + left, pts + hi, [axis](const KdPoint& p, const KdPoint& q) {
    return p.c[axis] < q.c[axis] || (p.c[axis] == q.c[axis] && p.idx < q.idx);
  });
  if (depth < 4 && count > 4096) {  // the two halves are independent: up to 16 host threads on large inputs
    auto fut = std::async(std::launch::async, kd_order, pts, dim, lo, lo + left, depth + 1);
    kd_order(pts, dim, lo + left, hi, depth + 1);
    fut.get();
  } else {
    kd_order(pts, dim, lo, lo + left, depth + 1);
    kd_order(pts, dim, lo + left, hi, depth + 1);
  }
}

int32_t knn_index_build(const double* xhost, int64_t n, int dim, KnnIndex* ix, hipStream_t s) {
  std::vector<KdPoint> pts((size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    for (int a = 0; a < 3; ++a) pts[(size_t)i].c[a] = a < dim ? xhost[i * dim + a] : 0.0;
    pts[(size_t)i].idx = (int32_t)i;
  }
  kd_order(pts.data(), dim, 0, n, 0);
  const int nb = (int)((n + 63) / 64);
  std::vector<double> xs((size_t)(n * dim)), blo((size_t)nb * dim), bhi((size_t)nb * dim);
  std::vector<int32_t> perm((size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    perm[(size_t)i] = pts[(size_t)i].idx;
    for (int a = 0; a < dim; ++a) xs[(size_t)(i * dim + a)] = pts[(size_t)i].c[a];
  }
  for (int b = 0; b < nb; ++b) {
    const int64_t j0 = (int64_t)b * 64, j1 = j0 + 64 < n ? j0 + 64 : n;
    for (int a = 0; a < dim; ++a) {
      double l = xs[(size_t)(j0 * dim + a)], h = l;
      for (int64_t j = j0 + 1; j < j1; ++j) {
        const double v = xs[(size_t)(j * dim + a)];
        l = v < l ? v : l;
        h = v > h ? v : h;
      }
      blo[(size_t)b * dim + a] = l;
      bhi[(size_t)b * dim + a] = h;
    }
  }
  const int nb1 = (nb + 63) / 64;
  std::vector<double> blo1((size_t)nb1 * dim), bhi1((size_t)nb1 * dim);
  for (int g = 0; g < nb1; ++g) {
    const int b0 = g * 64, b1 = b0 + 64 < nb ? b0 + 64 : nb;
    for (int a = 0; a < dim; ++a) {
      double l = blo[(size_t)b0 * dim + a], h = bhi[(size_t)b0 * dim + a];
      for (int b = b0 + 1; b < b1; ++b) {
        l = blo[(size_t)b * dim + a] < l ? blo[(size_t)b * dim + a] : l;
        h = bhi[(size_t)b * dim + a] > h ? bhi[(size_t)b * dim + a] : h;
      }
      blo1[(size_t)g * dim + a] = l;
      bhi1[(size_t)g * dim + a] = h;
    }
  }
  GSS_TRY(ix->lo1.alloc(sizeof(double) * blo1.size()));
  GSS_TRY(ix->hi1.alloc(sizeof(double) * bhi1.size()));
  GSS_HIP(hipMemcpyAsync(ix->lo1.p, blo1.data(), sizeof(double) * blo1.size(), hipMemcpyHostToDevice, s));
  GSS_HIP(hipMemcpyAsync(ix->hi1.p, bhi1.data(), sizeof(double) * bhi1.size(), hipMemcpyHostToDevice, s));
  ix->nb1 = nb1;
  GSS_TRY(ix->xs.alloc(sizeof(double) * xs.size()));
  GSS_TRY(ix->perm.alloc(sizeof(int32_t) * perm.size()));
  GSS_TRY(ix->lo.alloc(sizeof(double) * blo.size()));
  GSS_TRY(ix->hi.alloc(sizeof(double) * bhi.size()));
  GSS_HIP(hipMemcpyAsync(ix->xs.p, xs.data(), sizeof(double) * xs.size(), hipMemcpyHostToDevice, s));
  GSS_HIP(hipMemcpyAsync(ix->perm.p, perm.data(), sizeof(int32_t) * perm.size(), hipMemcpyHostToDevice, s));
  GSS_HIP(hipMemcpyAsync(ix->lo.p, blo.data(), sizeof(double) * blo.size(), hipMemcpyHostToDevice, s));
  GSS_HIP(hipMemcpyAsync(ix->hi.p, bhi.data(), sizeof(double) * bhi.size(), hipMemcpyHostToDevice, s));
  GSS_HIP(hipStreamSynchronize(s));  // host vectors go out of scope
  ix->n = n;
  ix->nb = nb;
  ix->dim = dim;
  return GSS_OK;
}

int32_t knn_index_build_from_device(const double* xdev, int64_t n, int dim, KnnIndex* ix, hipStream_t s) {
  // large sets are ordered on the device (no copy back, no host sort); GSS_KNN_BUILD=host / device forces one path
  const char* e = std::getenv("GSS_KNN_BUILD");
  const bool dev = e ? (e[0] == 'd') : (n >= KNN_DEVICE_BUILD_MIN);
  if (dev) return knn_index_build_device(xdev, n, dim, ix, s);
  std::vector<double> xh((size_t)(n * dim));
  GSS_HIP(hipMemcpyAsync(xh.data(), xdev, sizeof(double) * xh.size(), hipMemcpyDeviceToHost, s));
  GSS_HIP(hipStreamSynchronize(s));
  return knn_index_build(xh.data(), n, dim, ix, s);
}

// lower bound of sqdist_nofma(point in box, q): same operation order, every step monotone in |t|
template <int DIM>
__device__ __forceinline__ double box_sqdist_nofma(const double* lo, const double* hi, const double* q,
                                                   const double* ir, bool aniso) {
#pragma clang fp contract(off)
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < DIM; ++k) {
    const double a = lo[k] - q[k], b = q[k] - hi[k];
    double t = a > b ? a : b;
    t = t > 0.0 ? t : 0.0;
    if (aniso) t = t * ir[k];
    const double tt = t * t;
    acc = acc + tt;
  }
  return acc;
}

// Lower bound of the search key between a query and any point of a box, per metric: the per-axis gap
// max(lo - q, q - hi, 0) never exceeds |x - q| for lo <= x <= hi (floating-point subtraction is monotone and
// antisymmetric), and it is accumulated with the same operation, in the same order, as the key itself.
template <int DIM, int METRIC>
__device__ __forceinline__ double box_key(const double* lo, const double* hi, const double* q, const double* ir, bool aniso) {
#pragma clang fp contract(off)
  if (METRIC == GSS_METRIC_EUCLIDEAN) return box_sqdist_nofma<DIM>(lo, hi, q, ir, aniso);
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < DIM; ++k) {
    const double a = lo[k] - q[k], b = q[k] - hi[k];
    double t = a > b ? a : b;
    t = t > 0.0 ? t : 0.0;
    if (METRIC == GSS_METRIC_CITYBLOCK) acc = acc + t;
    else acc = t > acc ? t : acc;
  }
  return acc;
}

// MASKED (sequential simulation, seq.jl:105 `search!(..., mask=simulated)`): a sample qualifies only if its
// visiting rank is lower than the query's (rank[] per original index, qrank[] per query, bminrank[] = lowest
// rank inside each batch so that batches with nothing simulated yet are skipped).
template <int DIM, bool MASKED, int METRIC = GSS_METRIC_EUCLIDEAN>
__global__ __launch_bounds__(256) void knn_pruned_kernel(const double* __restrict__ xs, const int* __restrict__ perm,
                                                         const double* __restrict__ blo, const double* __restrict__ bhi,
                                                         const double* __restrict__ blo1,
                                                         const double* __restrict__ bhi1, int n, int nb, int nb1,
                                                         const double* __restrict__ centers, int64_t m, int k,
                                                         double r2, int use_ball, int aniso, double ir0, double ir1,
                                                         double ir2, const int* __restrict__ rank,
                                                         const int* __restrict__ qrank,
                                                         const int* __restrict__ bminrank, int* __restrict__ idx_out,
                                                         int* __restrict__ count_out,
                                                         const double* __restrict__ lowd,
                                                         const int* __restrict__ lowi) {
  const int lane = threadIdx.x & 63;
  const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= m) return;  // whole wave
  const int myrank = MASKED ? qrank[p] : 0;
  // more than 64 neighbours are found 64 at a time: a later pass only accepts keys above the last key of the pass
  // before it (keys are >= 0, so (-1, -1) accepts everything)
  const double my_lowd = lowd ? lowd[p] : -1.0;
  const int my_lowi = lowd ? lowi[p] : -1;
  const double ir[3] = {ir0, ir1, ir2};
  double qc[DIM];
#pragma unroll
  for (int a = 0; a < DIM; ++a) qc[a] = centers[p * DIM + a];
  const double INF = __builtin_huge_val();
  double ld = INF;   // lane l: l-th nearest so far (INF / INT_MAX = empty)
  int li = INT_MAX;

  // two box levels: groups of 64 batches, then the batches of a group; both visited nearest box first, and a
  // box whose bound exceeds the current k-th distance is never opened
  for (int c1 = 0; c1 < nb1; c1 += 64) {
    const int g = c1 + lane;
    double d1 = INF;
    if (g < nb1) {
      double lo[DIM], hi[DIM];
#pragma unroll
      for (int a = 0; a < DIM; ++a) {
        lo[a] = blo1[g * DIM + a];
        hi[a] = bhi1[g * DIM + a];
      }
      d1 = box_key<DIM, METRIC>(lo, hi, qc, ir, aniso != 0);
    }
    bool done1 = !(g < nb1);
    while (true) {
      const double tau1 = readlane_f64(ld, k - 1);
      const bool cand1 = !done1 && d1 <= tau1 && (!use_ball || d1 <= r2);
      if (!__ballot(cand1)) break;
      const double mn1 = wave_min_f64(cand1 ? d1 : INF);
      const int pick1 = __builtin_ctzll(__ballot(cand1 && d1 == mn1));
      if (lane == pick1) done1 = true;
      const int b = (c1 + pick1) * 64 + lane;
      double dmin = INF;
      bool done = !(b < nb);
      if (!done) {
        double lo[DIM], hi[DIM];
#pragma unroll
        for (int a = 0; a < DIM; ++a) {
          lo[a] = blo[b * DIM + a];
          hi[a] = bhi[b * DIM + a];
        }
        dmin = box_key<DIM, METRIC>(lo, hi, qc, ir, aniso != 0);
        if (MASKED && !(bminrank[b] < myrank)) done = true;
      }
      while (true) {
        // current k-th best key, re-read from the list (lane k - 1) whenever it may have changed
        double tau_d = readlane_f64(ld, k - 1);
        int tau_i = __builtin_amdgcn_readlane(li, k - 1);
        const bool cand = !done && dmin <= tau_d && (!use_ball || dmin <= r2);
        if (!__ballot(cand)) break;
        const double mn = wave_min_f64(cand ? dmin : INF);
        const int pick = __builtin_ctzll(__ballot(cand && dmin == mn));
        if (lane == pick) done = true;
        const int j = ((c1 + pick1) * 64 + pick) * 64 + lane;
        const bool valid = j < n;
        double c[DIM];
#pragma unroll
        for (int a = 0; a < DIM; ++a) c[a] = valid ? xs[(int64_t)j * DIM + a] : 0.0;
        const int oidx = valid ? perm[j] : INT_MAX;
        const double d2 = metric_key<DIM, METRIC>(c, qc, ir, aniso != 0);
        bool qual = valid && (!use_ball || d2 <= r2) && key_less(d2, oidx, tau_d, tau_i) &&
                    key_less(my_lowd, my_lowi, d2, oidx);
        if (MASKED) qual = qual && rank[valid ? oidx : 0] < myrank;
        unsigned long long qm = __ballot(qual);
        // (short lists take the sort too while they are not full: the first batch would otherwise be 64 insertions)
        if ((k >= KNN_SORT_MIN_K || tau_i == INT_MAX) && __popcll(qm) >= KNN_SORT_MIN) {
          // many candidates beat the current k-th key (always true for the first batches): sort the batch and merge
          // it into the list with bitonic networks instead of inserting one candidate at a time
          double bd = qual ? d2 : INF;
          int bi = qual ? oidx : INT_MAX;
          bitonic_sort64(bd, bi, lane);
          bitonic_merge64(ld, li, bd, bi, lane);
          qm = 0;
        }
        while (qm) {
          const int src = __builtin_ctzll(qm);
          qm &= qm - 1;
          const double cd = readlane_f64(d2, src);
          const int ci = __builtin_amdgcn_readlane(oidx, src);
          if (!key_less(cd, ci, tau_d, tau_i)) continue;
          const int pos = __popcll(__ballot(key_less(ld, li, cd, ci)));
          const double up_d = shfl_up1_f64(ld);
          const int up_i = shfl_up1_b32(li);
          if (lane > pos) {
            ld = up_d;
            li = up_i;
          } else if (lane == pos) {
            ld = cd;
            li = ci;
          }
          tau_d = readlane_f64(ld, k - 1);
          tau_i = __builtin_amdgcn_readlane(li, k - 1);
        }
      }
    }
  }
  const bool has = lane < k && li != INT_MAX;
  const int cnt = __popcll(__ballot(has));
  if (lane < k) idx_out[p * k + lane] = has ? li : -1;
  if (lane == 0 && count_out) count_out[p] = cnt;
}

int32_t knn_search_indexed(const KnnIndex& ix, const double* centers, int64_t m, int k, double radius,
                           const double* inv_radii_host, int* idx, int* count, hipStream_t s, const int* rank,
                           const int* qrank, const int* bminrank, int metric, const double* lowd, const int* lowi) {
  GSS_REQUIRE(k >= 1 && k <= 64, "knn_search_indexed: one pass finds at most 64 neighbours (got k = %d); "
                                 "knn_search_indexed_any runs the passes for more", k);
  if (m <= 0) return GSS_OK;
  const int use_ball = (radius >= 0.0 || inv_radii_host != nullptr) ? 1 : 0;
  const int aniso = inv_radii_host != nullptr ? 1 : 0;
  const double r2 = aniso ? 1.0 : radius * radius;
  double ir[3] = {1.0, 1.0, 1.0};
  if (aniso)
    for (int a = 0; a < ix.dim; ++a) ir[a] = inv_radii_host[a];
  dim3 grid((unsigned)((m + 3) / 4));
#define GSS_KNN_ARGS ix.xs.as<double>(), ix.perm.as<int>(), ix.lo.as<double>(), ix.hi.as<double>(), \
                     ix.lo1.as<double>(), ix.hi1.as<double>(), (int)ix.n, ix.nb, ix.nb1, \
                     centers, m, k, r2, use_ball, aniso, ir[0], ir[1], ir[2], rank, qrank, bminrank, idx, count, lowd, lowi
  GSS_REQUIRE(metric == GSS_METRIC_EUCLIDEAN || metric == GSS_METRIC_CITYBLOCK || metric == GSS_METRIC_CHEBYSHEV,
              "the indexed search has box bounds for the Euclidean, Cityblock and Chebyshev keys only");
  if (rank) GSS_REQUIRE(qrank && bminrank, "masked search needs query ranks and per-batch minimum ranks");
#define GSS_KNN_LAUNCH(MASKED, METRIC)                                                                               \
  switch (ix.dim) {                                                                                                   \
    case 1: hipLaunchKernelGGL((knn_pruned_kernel<1, MASKED, METRIC>), grid, dim3(256), 0, s, GSS_KNN_ARGS); break;   \
    case 2: hipLaunchKernelGGL((knn_pruned_kernel<2, MASKED, METRIC>), grid, dim3(256), 0, s, GSS_KNN_ARGS); break;   \
    default: hipLaunchKernelGGL((knn_pruned_kernel<3, MASKED, METRIC>), grid, dim3(256), 0, s, GSS_KNN_ARGS); break;  \
  }
  if (rank) {   // SGS: candidates whose rank lies below the query's
    if (metric == GSS_METRIC_CITYBLOCK) { GSS_KNN_LAUNCH(true, GSS_METRIC_CITYBLOCK) }
    else if (metric == GSS_METRIC_CHEBYSHEV) { GSS_KNN_LAUNCH(true, GSS_METRIC_CHEBYSHEV) }
    else { GSS_KNN_LAUNCH(true, GSS_METRIC_EUCLIDEAN) }
  } else {
    if (metric == GSS_METRIC_CITYBLOCK) { GSS_KNN_LAUNCH(false, GSS_METRIC_CITYBLOCK) }
    else if (metric == GSS_METRIC_CHEBYSHEV) { GSS_KNN_LAUNCH(false, GSS_METRIC_CHEBYSHEV) }
    else { GSS_KNN_LAUNCH(false, GSS_METRIC_EUCLIDEAN) }
  }
#undef GSS_KNN_LAUNCH
#undef GSS_KNN_ARGS
  GSS_HIP(hipGetLastError());
  return GSS_OK;
}

// ---- more than 64 neighbours (ui.jl:16-23 accepts any maxneighbors <= n): passes of 64 -----------------------------
__global__ __launch_bounds__(256) void knn_any_init_kernel(int64_t m, int* __restrict__ count, double* __restrict__ lowd,
                                                           int* __restrict__ lowi) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p < m) {
    count[p] = 0;
    lowd[p] = -1.0;
    lowi[p] = -1;
  }
}

// appends pass results (m x kk) to the full lists (m x k) at column `base`; the key of the last neighbour found
// becomes the lower bound of the next pass (recomputed with the rounding of the search key itself); a pass that
// came back short has exhausted the candidates: the bound goes to +inf
template <int DIM>
__global__ __launch_bounds__(256) void knn_any_append_kernel(const double* __restrict__ xdata,
                                                             const double* __restrict__ centers, int64_t m, int k,
                                                             int base, int kk, const int* __restrict__ tidx,
                                                             const int* __restrict__ tcnt, int metric, int aniso,
                                                             double ir0, double ir1, double ir2, int* __restrict__ idx,
                                                             int* __restrict__ count, double* __restrict__ lowd,
                                                             int* __restrict__ lowi) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= m) return;
  const int c = tcnt[p];
  for (int j = 0; j < kk; ++j) idx[p * k + base + j] = j < c ? tidx[p * kk + j] : -1;
  count[p] += c;
  if (c < kk) {
    lowd[p] = __builtin_huge_val();
    lowi[p] = INT_MAX;
    return;
  }
  const int last = tidx[p * kk + c - 1];
  const double ir[3] = {ir0, ir1, ir2};
  double q[DIM], x[DIM];
#pragma unroll
  for (int a = 0; a < DIM; ++a) {
    q[a] = centers[p * DIM + a];
    x[a] = xdata[(int64_t)last * DIM + a];
  }
  double key;
  if (metric == GSS_METRIC_CITYBLOCK) key = metric_key<DIM, GSS_METRIC_CITYBLOCK>(x, q, ir, false);
  else if (metric == GSS_METRIC_CHEBYSHEV) key = metric_key<DIM, GSS_METRIC_CHEBYSHEV>(x, q, ir, false);
  else if (metric == GSS_METRIC_HAVERSINE) key = metric_key<DIM, GSS_METRIC_HAVERSINE>(x, q, ir, false);
  else key = metric_key<DIM, GSS_METRIC_EUCLIDEAN>(x, q, ir, aniso != 0);
  lowd[p] = key;
  lowi[p] = last;
}

int32_t knn_search_indexed_any(const KnnIndex& ix, const double* xdata, const double* centers, int64_t m, int k,
                               double radius, const double* inv_radii_host, int* idx, int* count, hipStream_t s,
                               int metric, const int* rank, const int* qrank, const int* bminrank) {
  if (k <= 64) return knn_search_indexed(ix, centers, m, k, radius, inv_radii_host, idx, count, s, rank, qrank,
                                         bminrank, metric);
  if (m <= 0) return GSS_OK;
  DevBuf tidx, tcnt, lowd, lowi, cnt_own;
  GSS_TRY(tidx.alloc(sizeof(int) * (size_t)(m * 64)));
  GSS_TRY(tcnt.alloc(sizeof(int) * (size_t)m));
  GSS_TRY(lowd.alloc(sizeof(double) * (size_t)m));
  GSS_TRY(lowi.alloc(sizeof(int) * (size_t)m));
  if (!count) {
    GSS_TRY(cnt_own.alloc(sizeof(int) * (size_t)m));
    count = cnt_own.as<int>();
  }
  const int aniso = inv_radii_host != nullptr ? 1 : 0;
  double ir[3] = {1.0, 1.0, 1.0};
  if (aniso)
    for (int a = 0; a < ix.dim; ++a) ir[a] = inv_radii_host[a];
  const dim3 grid((unsigned)((m + 255) / 256));
  hipLaunchKernelGGL(knn_any_init_kernel, grid, dim3(256), 0, s, m, count, lowd.as<double>(), lowi.as<int>());
  for (int base = 0; base < k; base += 64) {
    const int kk = (k - base) < 64 ? (k - base) : 64;
    GSS_TRY(knn_search_indexed(ix, centers, m, kk, radius, inv_radii_host, tidx.as<int>(), tcnt.as<int>(), s, rank,
                               qrank, bminrank, metric, lowd.as<double>(), lowi.as<int>()));
#define GSS_APPEND(D)                                                                                                  \
  hipLaunchKernelGGL(knn_any_append_kernel<D>, grid, dim3(256), 0, s, xdata, centers, m, k, base, kk, tidx.as<int>(),  \
                     tcnt.as<int>(), metric, aniso, ir[0], ir[1], ir[2], idx, count, lowd.as<double>(), lowi.as<int>())
    switch (ix.dim) {
      case 1: GSS_APPEND(1); break;
      case 2: GSS_APPEND(2); break;
      default: GSS_APPEND(3); break;
    }
#undef GSS_APPEND
    GSS_HIP(hipGetLastError());
  }
  GSS_HIP(hipStreamSynchronize(s));  // the pass buffers are released on return
  return GSS_OK;
}

int32_t check_metric(int metric, double metric_param, int dim, double radius, const double* inv_radii) {
  GSS_REQUIRE(metric >= GSS_METRIC_EUCLIDEAN && metric <= GSS_METRIC_HAVERSINE, "unknown search metric %d", metric);
  if (metric == GSS_METRIC_EUCLIDEAN) return GSS_OK;
  // searcher_ui (ui.jl:25-31): a neighbourhood replaces the metric search, the two are never combined
  GSS_REQUIRE(radius < 0.0 && inv_radii == nullptr, "a search ball cannot be combined with a non-Euclidean distance");
  if (metric == GSS_METRIC_HAVERSINE) {
    GSS_REQUIRE(dim == 2, "the haversine distance needs (longitude, latitude) points, got %d-D", dim);
    GSS_REQUIRE(metric_param > 0.0, "the haversine distance needs a positive sphere radius");
  }
  return GSS_OK;
}

template <int DIM, int METRIC>
static void launch_brute(dim3 grid, hipStream_t s, const double* xdata, int n, const double* centers, int64_t m, int k,
                         double r2, int use_ball, int aniso, const double* ir, int* idx, int* count,
                         const double* lowd = nullptr, const int* lowi = nullptr) {
  hipLaunchKernelGGL((knn_kernel<DIM, METRIC>), grid, dim3(256), 0, s, xdata, n, centers, m, k, r2, use_ball, aniso,
                     ir[0], ir[1], ir[2], idx, count, lowd, lowi);
}

template <int DIM>
static void launch_brute_metric(int metric, dim3 grid, hipStream_t s, const double* xdata, int n,
                                const double* centers, int64_t m, int k, double r2, int use_ball, int aniso,
                                const double* ir, int* idx, int* count, const double* lowd = nullptr,
                                const int* lowi = nullptr) {
  switch (metric) {
    case GSS_METRIC_CITYBLOCK:
      launch_brute<DIM, GSS_METRIC_CITYBLOCK>(grid, s, xdata, n, centers, m, k, r2, use_ball, aniso, ir, idx, count, lowd, lowi);
      break;
    case GSS_METRIC_CHEBYSHEV:
      launch_brute<DIM, GSS_METRIC_CHEBYSHEV>(grid, s, xdata, n, centers, m, k, r2, use_ball, aniso, ir, idx, count, lowd, lowi);
      break;
    case GSS_METRIC_HAVERSINE:
      launch_brute<DIM, GSS_METRIC_HAVERSINE>(grid, s, xdata, n, centers, m, k, r2, use_ball, aniso, ir, idx, count, lowd, lowi);
      break;
    default:
      launch_brute<DIM, GSS_METRIC_EUCLIDEAN>(grid, s, xdata, n, centers, m, k, r2, use_ball, aniso, ir, idx, count, lowd, lowi);
      break;
  }
}

// more than 64 neighbours on the exhaustive kernel (the haversine distance has no box bounds for the indexed search):
// passes of 64, each restricted to the keys above the last neighbour of the pass before
static int32_t knn_search_brute_any(const double* xdata, int64_t n, int dim, const double* centers, int64_t m, int k,
                                    double r2, int use_ball, int aniso, const double* ir, int* idx, int* count,
                                    hipStream_t s, int metric) {
  DevBuf tidx, tcnt, lowd, lowi, cnt_own;
  GSS_TRY(tidx.alloc(sizeof(int) * (size_t)(m * 64)));
  GSS_TRY(tcnt.alloc(sizeof(int) * (size_t)m));
  GSS_TRY(lowd.alloc(sizeof(double) * (size_t)m));
  GSS_TRY(lowi.alloc(sizeof(int) * (size_t)m));
  if (!count) {
    GSS_TRY(cnt_own.alloc(sizeof(int) * (size_t)m));
    count = cnt_own.as<int>();
  }
  const dim3 g1((unsigned)((m + 255) / 256));
  const dim3 grid((unsigned)((m + 4 * KNN_Q - 1) / (4 * KNN_Q)));
  hipLaunchKernelGGL(knn_any_init_kernel, g1, dim3(256), 0, s, m, count, lowd.as<double>(), lowi.as<int>());
  for (int base = 0; base < k; base += 64) {
    const int kk = (k - base) < 64 ? (k - base) : 64;
#define GSS_BRUTE_PASS(D)                                                                                               \
  do {                                                                                                                   \
    launch_brute_metric<D>(metric, grid, s, xdata, (int)n, centers, m, kk, r2, use_ball, aniso, ir, tidx.as<int>(),     \
                           tcnt.as<int>(), lowd.as<double>(), lowi.as<int>());                                          \
    hipLaunchKernelGGL(knn_any_append_kernel<D>, g1, dim3(256), 0, s, xdata, centers, m, k, base, kk, tidx.as<int>(),   \
                       tcnt.as<int>(), metric, aniso, ir[0], ir[1], ir[2], idx, count, lowd.as<double>(),               \
                       lowi.as<int>());                                                                                  \
  } while (0)
    switch (dim) {
      case 1: GSS_BRUTE_PASS(1); break;
      case 2: GSS_BRUTE_PASS(2); break;
      default: GSS_BRUTE_PASS(3); break;
    }
#undef GSS_BRUTE_PASS
    GSS_HIP(hipGetLastError());
  }
  GSS_HIP(hipStreamSynchronize(s));  // the pass buffers are released on return
  return GSS_OK;
}

// Euclidean: pruned search (exhaustive kernel with GSS_KNN_BRUTE=1, kept for A/B checks); other metrics: exhaustive
int32_t knn_search_dev(const double* xdata, int64_t n, int dim, const double* centers, int64_t m, int k,
                       double radius, const double* inv_radii_host, int* idx, int* count, hipStream_t s, int metric) {
  GSS_REQUIRE(k >= 1, "maxneighbors = %d", k);
  GSS_REQUIRE(n >= 1 && n < INT_MAX && dim >= 1 && dim <= 3, "knn: bad sizes");
  if (m <= 0) return GSS_OK;
  const char* e = std::getenv("GSS_KNN_BRUTE");
  // few queries into a large set (e.g. the data -> grid-cell lookup of conditional simulation, fft.jl:129-132):
  // one brute-force sweep of the set costs less than ordering it on the host for the index
  const bool few_queries = m <= 4096 && n >= 32768 && k <= 64;
  if (metric != GSS_METRIC_HAVERSINE && (k > 64 || (!(e && e[0] == '1') && !few_queries))) {
    KnnIndex ix;
    GSS_TRY(knn_index_build_from_device(xdata, n, dim, &ix, s));
    GSS_TRY(knn_search_indexed_any(ix, xdata, centers, m, k, radius, inv_radii_host, idx, count, s, metric));
    GSS_HIP(hipStreamSynchronize(s));  // the index is released on return
    return GSS_OK;
  }
  const int use_ball = (radius >= 0.0 || inv_radii_host != nullptr) ? 1 : 0;
  const int aniso = inv_radii_host != nullptr ? 1 : 0;
  const double r2 = aniso ? 1.0 : radius * radius;
  double ir[3] = {1.0, 1.0, 1.0};
  if (aniso)
    for (int a = 0; a < dim; ++a) ir[a] = inv_radii_host[a];
  if (k > 64) return knn_search_brute_any(xdata, n, dim, centers, m, k, r2, use_ball, aniso, ir, idx, count, s, metric);
  dim3 grid((unsigned)((m + 4 * KNN_Q - 1) / (4 * KNN_Q)));
  switch (dim) {
    case 1: launch_brute_metric<1>(metric, grid, s, xdata, (int)n, centers, m, k, r2, use_ball, aniso, ir, idx, count); break;
    case 2: launch_brute_metric<2>(metric, grid, s, xdata, (int)n, centers, m, k, r2, use_ball, aniso, ir, idx, count); break;
    default: launch_brute_metric<3>(metric, grid, s, xdata, (int)n, centers, m, k, r2, use_ball, aniso, ir, idx, count); break;
  }
  GSS_HIP(hipGetLastError());
  return GSS_OK;
}

}  // namespace gss

using namespace gss;

extern "C" int32_t gss_knn_search(const double* xdata, int64_t n, int32_t dim, const double* centers, int64_t m,
                                  int32_t k, double radius, const double* inv_radii, int32_t metric,
                                  double metric_param, int32_t* idx, int32_t* count, int32_t mem, void* stream) {
  GSS_ENTRY();
  GSS_REQUIRE(xdata && centers && idx, "gss_knn_search: NULL array");
  GSS_REQUIRE(k >= 1 && k <= n, "gss_knn_search: k = %d outside 1..n = %lld", k, (long long)n);
  GSS_TRY(check_metric(metric, metric_param, dim, radius, inv_radii));
  hipStream_t s = to_stream(stream);
  Staged sx, sc, si, sn;
  GSS_TRY(sx.in(xdata, sizeof(double) * (size_t)(n * dim), mem, s));
  GSS_TRY(sc.in(centers, sizeof(double) * (size_t)(m * dim), mem, s));
  GSS_TRY(si.out(idx, sizeof(int32_t) * (size_t)(m * k), mem));
  GSS_TRY(sn.out(count, sizeof(int32_t) * (size_t)m, mem));
  GSS_TRY(knn_search_dev(sx.as<double>(), n, dim, sc.as<double>(), m, k, radius, inv_radii, si.as<int>(),
                         sn.as<int>(), s, metric));
  GSS_TRY(si.back(idx, sizeof(int32_t) * (size_t)(m * k), mem, s));
  GSS_TRY(sn.back(count, sizeof(int32_t) * (size_t)m, mem, s));
  return GSS_OK;
}
