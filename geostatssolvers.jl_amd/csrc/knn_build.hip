// Device-side construction of the k-d ordered neighbour-search index (KnnIndex, knn.hip) for large sample sets.
//
// The host builder (knn.hip: nth_element recursion) costs about 40 ns per sample plus the copies across PCIe -- more
// than the search itself once n reaches tens of thousands.  The tree is balanced by construction: where a range is
// split depends on its length only (multiples of 64 / 4096 points), never on the data, so the host lays out every
// level's ranges from n alone and the device only has to order the points inside them.  Level by level:
//   1. bounding box of every range (wave-reduced, then 64-bit integer atomics on an order-preserving image of the
//      doubles),
//   2. key = (range number, position along the range's widest axis quantised to 32 bits), value = original index,
//   3. one radix sort (hipCUB) over 32 + log2(#ranges) key bits.
// After the sort the lower `left` points of each range are its left child.  Quantisation can put two points that are
// closer than extent / 2^32 along the axis on the "wrong" side of a split; the batch boxes are computed from the points
// actually in each batch afterwards, so the search stays exact -- only the boxes may overlap by that sliver.
#include "gss_internal.h"

#include <hipcub/hipcub.hpp>

#include <vector>

namespace gss {

namespace {

__device__ __forceinline__ unsigned long long f64_to_key(double v) {  // monotone: a < b  <=>  key(a) < key(b)
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_to_f64(unsigned long long k) {
  const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)b);
}

// largest s with off[s] <= i  (off[0] = 0 < off[1] < ... < off[nseg] = n)
__device__ __forceinline__ int find_range(const int* __restrict__ off, int nseg, int i) {
  int lo = 0, hi = nseg;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (off[mid] <= i) lo = mid;
    else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void kd_init_kernel(int* __restrict__ perm, int n, unsigned long long* __restrict__ bmin,
                                                      unsigned long long* __restrict__ bmax, int nbox, int first) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (first && i < n) perm[i] = i;
  if (i < nbox) {
    bmin[i] = ~0ull;
    bmax[i] = 0ull;
  }
}

template <int DIM>
__global__ __launch_bounds__(256) void kd_bbox_kernel(const double* __restrict__ x, const int* __restrict__ perm, int n,
                                                      const int* __restrict__ off, int nseg,
                                                      unsigned long long* __restrict__ bmin,
                                                      unsigned long long* __restrict__ bmax) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool live = i < n;
  const int ic = live ? i : n - 1;
  const int seg = find_range(off, nseg, ic);
  unsigned long long lo[DIM], hi[DIM];
  const int p = perm[ic];
#pragma unroll
  for (int a = 0; a < DIM; ++a) lo[a] = hi[a] = f64_to_key(x[(int64_t)p * DIM + a]);
  // a wave that lies inside one range (all but the few that straddle a boundary) reduces before it touches memory
  const int seg0 = __builtin_amdgcn_readfirstlane(seg);
  if (__all(seg == seg0)) {
#pragma unroll
    for (int a = 0; a < DIM; ++a) {
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) {
        const unsigned long long l2 = __shfl_xor(lo[a], o), h2 = __shfl_xor(hi[a], o);
        lo[a] = l2 < lo[a] ? l2 : lo[a];
        hi[a] = h2 > hi[a] ? h2 : hi[a];
      }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
      for (int a = 0; a < DIM; ++a) {
        atomicMin(&bmin[seg0 * 3 + a], lo[a]);
        atomicMax(&bmax[seg0 * 3 + a], hi[a]);
      }
    }
  } else {
#pragma unroll
    for (int a = 0; a < DIM; ++a) {
      atomicMin(&bmin[seg * 3 + a], lo[a]);
      atomicMax(&bmax[seg * 3 + a], hi[a]);
    }
  }
}

template <int DIM>
__global__ __launch_bounds__(256) void kd_keys_kernel(const double* __restrict__ x, const int* __restrict__ perm, int n,
                                                      const int* __restrict__ off, int nseg,
                                                      const unsigned long long* __restrict__ bmin,
                                                      const unsigned long long* __restrict__ bmax,
                                                      unsigned long long* __restrict__ key, int* __restrict__ val) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int seg = find_range(off, nseg, i);
  const int p = perm[i];
  unsigned int q = 0u;
  if (off[seg + 1] - off[seg] > 64) {  // ranges of one batch are final: their points keep their order
    int axis = 0;
    double best = -1.0, lo_ax = 0.0;
#pragma unroll
    for (int a = 0; a < DIM; ++a) {
      const double l = key_to_f64(bmin[seg * 3 + a]), h = key_to_f64(bmax[seg * 3 + a]);
      if (h - l > best) {
        best = h - l;
        axis = a;
        lo_ax = l;
      }
    }
    double c = 0.0;
#pragma unroll
    for (int a = 0; a < DIM; ++a)
      if (a == axis) c = x[(int64_t)p * DIM + a];
    const double t = best > 0.0 ? (c - lo_ax) / best * 4294967296.0 : 0.0;
    q = t >= 4294967295.0 ? 4294967295u : (unsigned int)t;
  }
  key[i] = ((unsigned long long)(unsigned int)seg << 32) | q;
  val[i] = p;
}

template <int DIM>
__global__ __launch_bounds__(256) void kd_gather_kernel(const double* __restrict__ x, const int* __restrict__ perm, int n,
                                                        double* __restrict__ xs) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int p = perm[i];
#pragma unroll
  for (int a = 0; a < DIM; ++a) xs[(int64_t)i * DIM + a] = x[(int64_t)p * DIM + a];
}

// boxes of `count` items of `stride`-spaced groups: level 0 (points -> batches of 64) reads xs for both corners, level 1
// (batches -> groups of 64 batches) reads the batch corners.  One wave per output box.
template <int DIM>
__global__ __launch_bounds__(256) void kd_boxes_kernel(const double* __restrict__ lo_in, const double* __restrict__ hi_in,
                                                       int nin, double* __restrict__ lo_out, double* __restrict__ hi_out,
                                                       int nout) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (b >= nout) return;  // whole wave
  const int j = b * 64 + lane;
  const bool live = j < nin;
  const int jc = live ? j : nin - 1;  // a clamped duplicate does not change the box
  double lo[DIM], hi[DIM];
#pragma unroll
  for (int a = 0; a < DIM; ++a) {
    lo[a] = lo_in[(int64_t)jc * DIM + a];
    hi[a] = hi_in[(int64_t)jc * DIM + a];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const double l2 = __shfl_xor(lo[a], o), h2 = __shfl_xor(hi[a], o);
      lo[a] = l2 < lo[a] ? l2 : lo[a];
      hi[a] = h2 > hi[a] ? h2 : hi[a];
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < DIM; ++a) {
      lo_out[(int64_t)b * DIM + a] = lo[a];
      hi_out[(int64_t)b * DIM + a] = hi[a];
    }
  }
}

template <int DIM>
int32_t build_dim(const double* xdev, int n, KnnIndex* ix, hipStream_t s) {
  // ranges of every level, from n alone (same rule as kd_order in knn.hip)
  std::vector<std::vector<int>> levels;
  std::vector<int> cur{0, n};
  while (true) {
    std::vector<int> next;
    bool any = false;
    for (size_t r = 0; r + 1 < cur.size(); ++r) {
      const int lo = cur[r], count = cur[r + 1] - cur[r];
      next.push_back(lo);
      if (count > 64) {
        const int unit = count > 4096 ? 4096 : 64;
        const int units = (count + unit - 1) / unit;
        next.push_back(lo + unit * ((units + 1) / 2));
        any = true;
      }
    }
    next.push_back(n);
    if (!any) break;
    levels.push_back(cur);
    cur.swap(next);
  }
  size_t total = 0, maxseg = 1;
  for (auto& l : levels) {
    total += l.size();
    maxseg = l.size() - 1 > maxseg ? l.size() - 1 : maxseg;
  }
  std::vector<int> flat;
  flat.reserve(total);
  for (auto& l : levels) flat.insert(flat.end(), l.begin(), l.end());

  DevBuf d_off, d_bmin, d_bmax, d_key[2], d_val[2], d_tmp;
  GSS_TRY(d_off.alloc(sizeof(int) * (flat.empty() ? 1 : flat.size())));
  if (!flat.empty()) GSS_HIP(hipMemcpyAsync(d_off.p, flat.data(), sizeof(int) * flat.size(), hipMemcpyHostToDevice, s));
  GSS_TRY(d_bmin.alloc(sizeof(unsigned long long) * maxseg * 3));
  GSS_TRY(d_bmax.alloc(sizeof(unsigned long long) * maxseg * 3));
  for (int b = 0; b < 2; ++b) {
    GSS_TRY(d_key[b].alloc(sizeof(unsigned long long) * (size_t)n));
    GSS_TRY(d_val[b].alloc(sizeof(int) * (size_t)n));
  }
  hipcub::DoubleBuffer<unsigned long long> keys(d_key[0].as<unsigned long long>(), d_key[1].as<unsigned long long>());
  hipcub::DoubleBuffer<int> vals(d_val[0].as<int>(), d_val[1].as<int>());
  size_t tmp_bytes = 0;
  GSS_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys, vals, n, 0, 64, s));
  GSS_TRY(d_tmp.alloc(tmp_bytes));

  GSS_TRY(ix->perm.alloc(sizeof(int32_t) * (size_t)n));
  int* perm = ix->perm.as<int>();
  const dim3 gn((unsigned)((n + 255) / 256));
  size_t pos = 0;
  bool first = true;
  for (auto& l : levels) {
    const int nseg = (int)l.size() - 1;
    const int* off = d_off.as<int>() + pos;
    pos += l.size();
    const int ninit = n > nseg * 3 ? n : nseg * 3;
    hipLaunchKernelGGL(kd_init_kernel, dim3((unsigned)((ninit + 255) / 256)), dim3(256), 0, s, perm, n,
                       d_bmin.as<unsigned long long>(), d_bmax.as<unsigned long long>(), nseg * 3, first ? 1 : 0);
    first = false;
    hipLaunchKernelGGL(kd_bbox_kernel<DIM>, gn, dim3(256), 0, s, xdev, perm, n, off, nseg, d_bmin.as<unsigned long long>(),
                       d_bmax.as<unsigned long long>());
    hipLaunchKernelGGL(kd_keys_kernel<DIM>, gn, dim3(256), 0, s, xdev, perm, n, off, nseg, d_bmin.as<unsigned long long>(),
                       d_bmax.as<unsigned long long>(), keys.Current(), vals.Current());
    GSS_HIP(hipGetLastError());
    int bits = 1;
    while ((1 << bits) < nseg) ++bits;
    size_t tb = tmp_bytes;
    GSS_HIP(hipcub::DeviceRadixSort::SortPairs(d_tmp.p, tb, keys, vals, n, 0, 32 + bits, s));
    GSS_HIP(hipMemcpyAsync(perm, vals.Current(), sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, s));
  }
  if (first) {  // n <= 64: a single batch
    hipLaunchKernelGGL(kd_init_kernel, gn, dim3(256), 0, s, perm, n, d_bmin.as<unsigned long long>(),
                       d_bmax.as<unsigned long long>(), 0, 1);
  }
  const int nb = (n + 63) / 64, nb1 = (nb + 63) / 64;
  GSS_TRY(ix->xs.alloc(sizeof(double) * (size_t)n * DIM));
  GSS_TRY(ix->lo.alloc(sizeof(double) * (size_t)nb * DIM));
  GSS_TRY(ix->hi.alloc(sizeof(double) * (size_t)nb * DIM));
  GSS_TRY(ix->lo1.alloc(sizeof(double) * (size_t)nb1 * DIM));
  GSS_TRY(ix->hi1.alloc(sizeof(double) * (size_t)nb1 * DIM));
  hipLaunchKernelGGL(kd_gather_kernel<DIM>, gn, dim3(256), 0, s, xdev, perm, n, ix->xs.as<double>());
  hipLaunchKernelGGL(kd_boxes_kernel<DIM>, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0, s, ix->xs.as<double>(),
                     ix->xs.as<double>(), n, ix->lo.as<double>(), ix->hi.as<double>(), nb);
  hipLaunchKernelGGL(kd_boxes_kernel<DIM>, dim3((unsigned)((nb1 + 3) / 4)), dim3(256), 0, s, ix->lo.as<double>(),
                     ix->hi.as<double>(), nb, ix->lo1.as<double>(), ix->hi1.as<double>(), nb1);
  GSS_HIP(hipGetLastError());
  GSS_HIP(hipStreamSynchronize(s));  // the sort buffers are released on return
  ix->n = n;
  ix->nb = nb;
  ix->nb1 = nb1;
  ix->dim = DIM;
  return GSS_OK;
}

}  // namespace

int32_t knn_index_build_device(const double* xdev, int64_t n, int dim, KnnIndex* ix, hipStream_t s) {
  GSS_REQUIRE(n >= 1 && n < INT_MAX && dim >= 1 && dim <= 3, "knn index: bad sizes");
  switch (dim) {
    case 1: return build_dim<1>(xdev, (int)n, ix, s);
    case 2: return build_dim<2>(xdev, (int)n, ix, s);
    default: return build_dim<3>(xdev, (int)n, ix, s);
  }
}

}  // namespace gss
