// LUGS: LU (Cholesky) Gaussian simulation (Alabert 1987).
// Replaces preprocess (/root/reference/src/simulation/lu.jl:105-147) and lusim (lu.jl:198-224) for one
// variable; the host front-end keeps the reference's loop over co-variables and the rho mixing
// (lu.jl:171-196) and passes w1 back in for the second variable.
//
//   lu.jl:124      C22 = sill - pairwise(g, Ds)                   cov_pairwise (K1)
//   lu.jl:128      L22 = cholesky(C22).L          (unconditional)  potrf_blocked_f64 (panels of 1024, FP64 MFMA GEMMs)
//   lu.jl:131-132  C11, C12
//   lu.jl:134      L11 = cholesky(C11).L
//   lu.jl:135      B12 = L11 \ C12     -> stored transposed: A21 = C21 * inv(L11)'   (one GEMM with W11 = inv(L11))
//   lu.jl:136-138  d2  = A21 * (L11 \ z1)
//   lu.jl:139      L22 = cholesky(C22 - A21 * B12).L               (lower-tile SYRK + potrf)
//   lu.jl:209-213  y2  = d2 + L22 * w          for all realisations at once: one GEMM L22 * W
//   lu.jl:217-221  scatter to dlocs / slocs, add the mean when unconditional
// State kept in HBM: L22 (ns x ns, column-major, strict upper triangle zeroed) followed by d2 (ns).
#include "gss_internal.h"

#include <mutex>
#include "philox.h"

#include <cmath>
#include <vector>

namespace gss {

__global__ __launch_bounds__(256) void zero_upper_kernel(double* __restrict__ A, int64_t n, int64_t ld) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // row
  const int64_t j = blockIdx.y;                               // column
  if (i < n && i < j) A[i + j * ld] = 0.0;
}

// W(k, r) = rho * w1(k, r) + sqrt(1 - rho^2) * w2(k, r)
__global__ __launch_bounds__(256) void mix_kernel(const double* __restrict__ w1, const double* __restrict__ w2,
                                                  double rho, double c, int64_t n, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = rho * w1[i] + c * w2[i];
}

__global__ __launch_bounds__(256) void philox_normal_batch_kernel(uint64_t seed, int64_t first_real, int64_t ns,
                                                                  double* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t r = blockIdx.y;
  if (e < ns) out[r * ns + e] = philox_normal(seed, (uint32_t)(first_real + r), (uint64_t)e);
}

// out[r * N + slocs[i]] = d2[i] + Y2[i + r * ns] (+ mean);  out[r * N + dlocs[j]] = z1[j]
__global__ __launch_bounds__(256) void lugs_scatter_kernel(const double* __restrict__ Y2,
                                                           const double* __restrict__ d2,
                                                           const int64_t* __restrict__ slocs, int64_t ns,
                                                           const double* __restrict__ z1,
                                                           const int64_t* __restrict__ dlocs, int64_t nd, double add,
                                                           int64_t N, double* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t r = blockIdx.y;
  if (e < ns) out[r * N + slocs[e]] = d2[e] + Y2[e + r * ns] + add;
  else if (e < ns + nd) out[r * N + dlocs[e - ns]] = z1[e - ns];
}

}  // namespace gss

using namespace gss;

struct gss_lugs {
  VgDev vg;
  int dim = 0;
  int64_t N = 0, nd = 0, ns = 0;
  double mean = 0.0;
  DevBuf state;  // L22 (ns*ns) + d2 (ns)
  DevBuf slocs, dlocs, z1;
  bool ready = false;
  double* L22() const { return state.as<double>(); }
  double* d2() const { return state.as<double>() + ns * ns; }
};

constexpr int LUGS_KC = 6;   // column blocks of L22 in gss_lugs_realize
constexpr int32_t LUGS_RETRY = 1000;   // internal: the single-launch factorisation gave up, run the preprocess again

// Y(i, r) = sum over the column blocks c whose first column is <= i of part_c(i, r), in block order
__global__ __launch_bounds__(256) void sum_partials_kernel(const double* __restrict__ part, int nchunk, int64_t kc,
                                                           int64_t ns, int64_t R, double* __restrict__ Y) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t r = blockIdx.y;
  if (i >= ns) return;
  double acc = 0.0;
  for (int c = 0; c < nchunk && (int64_t)c * kc <= i; ++c) acc += part[((int64_t)c * R + r) * ns + i];
  Y[r * ns + i] = acc;
}

// helper streams of the process for gss_lugs_realize (fenced by events on both sides of every use)
static hipStream_t realize_stream(int i) { return helper_stream(HELPER_GEN0 + (i - 1)); }   // i = 1 .. 3

static int32_t check_info(DevBuf& info, const char* what, hipStream_t s) {
  int h = 0;
  GSS_HIP(hipMemcpyAsync(&h, info.p, sizeof(int), hipMemcpyDeviceToHost, s));
  GSS_HIP(hipStreamSynchronize(s));
  if (h < 0) {
    potrf_panel_disable();  // from now on the launch-per-block path, which has no residency requirement
    lu_grid_disable();      // (and single-workgroup LU panels)
    set_error("%s: the single-launch factorisation gave up waiting at a grid barrier (its workgroups were not resident "
              "together); switched off for this process", what);
    return LUGS_RETRY;
  }
  if (h != 0) {
    set_error("%s is not positive definite (pivot %d); add a nugget or remove duplicate locations", what, h - 1);
    return GSS_ERR_NOT_POSDEF;
  }
  return GSS_OK;
}

static int32_t lugs_create_impl(gss_lugs_t** out, const gss_variogram_t* vg, const double* centroids, int64_t N,
                                const int64_t* dlocs, const double* z1, int64_t nd, double mean, int32_t flags,
                                void* stream) {
  GSS_REQUIRE(out != nullptr, "gss_lugs_create: out is NULL");
  *out = nullptr;
  GSS_REQUIRE(centroids != nullptr && N >= 1 && nd >= 0 && nd <= N, "gss_lugs_create: bad sizes");
  GSS_REQUIRE(nd == 0 || (dlocs != nullptr && z1 != nullptr), "gss_lugs_create: NULL conditioning data");
  gss_lugs* h = new (std::nothrow) gss_lugs();
  if (!h) return GSS_ERR_ALLOC;
  struct Guard {   // an early exit: the look-ahead stream may still be writing the state (C22 is assembled there, the
    gss_lugs* h;   // trailing updates of the factorisations run there) -- join it before the handle's blocks return to the pool
    ~Guard() {
      if (h) {
        if (hipStream_t side = lookahead_stream()) (void)hipStreamSynchronize(side);
        delete h;
      }
    }
  } guard{h};
  GSS_REQUIRE(vg_is_stationary(vg), "variogram model must be stationary");  // fft.jl:91, lu.jl:110
  GSS_TRY(make_vgdev(vg, &h->vg));
  const int dim = h->dim = h->vg.dim;
  h->N = N;
  h->nd = nd;
  h->ns = N - nd;
  h->mean = mean;
  const int64_t ns = h->ns;
  hipStream_t s = to_stream(stream);

  // slocs = complement of dlocs (lu.jl:117), coordinates gathered on the host (O(N) bookkeeping)
  std::vector<char> isdata((size_t)N, 0);
  for (int64_t j = 0; j < nd; ++j) {
    GSS_REQUIRE(dlocs[j] >= 0 && dlocs[j] < N && !isdata[(size_t)dlocs[j]], "dlocs must be distinct and inside 0..N-1");
    GSS_REQUIRE(j == 0 || dlocs[j] > dlocs[j - 1], "dlocs must be sorted (findall(mask), lu.jl:113)");
    isdata[(size_t)dlocs[j]] = 1;
  }
  std::vector<int64_t> sl;
  sl.reserve((size_t)ns);
  for (int64_t l = 0; l < N; ++l)
    if (!isdata[(size_t)l]) sl.push_back(l);
  std::vector<double> xs((size_t)(ns * dim)), xd((size_t)(nd * dim));
  for (int64_t i = 0; i < ns; ++i)
    for (int a = 0; a < dim; ++a) xs[(size_t)(i * dim + a)] = centroids[sl[(size_t)i] * dim + a];
  for (int64_t j = 0; j < nd; ++j)
    for (int a = 0; a < dim; ++a) xd[(size_t)(j * dim + a)] = centroids[dlocs[j] * dim + a];

  DevBuf dxs, dxd, info;
  GSS_TRY(dxs.alloc(sizeof(double) * xs.size()));
  GSS_TRY(dxd.alloc(sizeof(double) * xd.size()));
  GSS_TRY(info.alloc(sizeof(int)));
  GSS_TRY(h->slocs.alloc(sizeof(int64_t) * (size_t)ns));
  GSS_TRY(h->dlocs.alloc(sizeof(int64_t) * (size_t)nd));
  GSS_TRY(h->z1.alloc(sizeof(double) * (size_t)nd));
  GSS_TRY(h->state.alloc(sizeof(double) * (size_t)(ns * ns + ns)));
  if (ns) GSS_HIP(hipMemcpyAsync(dxs.p, xs.data(), sizeof(double) * xs.size(), hipMemcpyHostToDevice, s));
  if (ns) GSS_HIP(hipMemcpyAsync(h->slocs.p, sl.data(), sizeof(int64_t) * (size_t)ns, hipMemcpyHostToDevice, s));
  if (nd) {
    GSS_HIP(hipMemcpyAsync(dxd.p, xd.data(), sizeof(double) * xd.size(), hipMemcpyHostToDevice, s));
    GSS_HIP(hipMemcpyAsync(h->dlocs.p, dlocs, sizeof(int64_t) * (size_t)nd, hipMemcpyHostToDevice, s));
    GSS_HIP(hipMemcpyAsync(h->z1.p, z1, sizeof(double) * (size_t)nd, hipMemcpyHostToDevice, s));
  }
  GSS_HIP(hipMemsetAsync(h->d2(), 0, sizeof(double) * (size_t)ns, s));

  if (flags & GSS_LUGS_NO_FACTOR) {  // the state arrives by broadcast (gss_lugs_adopt_state)
    GSS_HIP(hipStreamSynchronize(s));
    guard.h = nullptr;
    *out = h;
    return GSS_OK;
  }
  const bool use_lu = (flags & GSS_LUGS_FACT_LU) != 0;                                           // lu.jl:107
  GSS_TRY(dev_zero_bytes(info.p, sizeof(int), s));
  if (ns > 0) {
    double* C22 = h->L22();
    // C22 is not needed before the first trailing update: with conditioning data on the Cholesky path it is assembled
    // on the helper stream, in front of the products potrf_joint_f64 puts there, beside the first data panel
    hipStream_t side = (nd > 0 && !use_lu) ? lookahead_stream() : nullptr;
    if (side) {
      ScopedEvent ev;
      GSS_HIP(ev.create());
      GSS_HIP(hipEventRecord(ev, s));              // the coordinates are on the device
      GSS_HIP(hipStreamWaitEvent(side, ev, 0));
    }
    GSS_TRY(cov_pairwise_dev(h->vg, dxs.as<double>(), ns, dxs.as<double>(), ns, C22, ns, side ? side : s));   // lu.jl:124
    if (nd > 0 && !use_lu) {
      // Cholesky path: the data columns of [C11 . ; C21 C22] as one blocked factorisation (potrf_joint_f64): L11,
      // A21 = C21 L11^-T (= B12', lu.jl:135) and C22 - A21 A21' (:139) without forming inv(L11); z1' rides along as
      // one more row of C21 and comes back as (L11^-1 z1)', so that d2 = A21 (L11 \ z1) (:138) is one GEMV.
      const int64_t mb = ns + 2;   // row ns: z1', row ns + 1: zero (an even leading dimension keeps the 16-byte loads)
      DevBuf C11, C21, w, work, gwork;
      GSS_TRY(C11.alloc(sizeof(double) * (size_t)(nd * nd)));
      GSS_TRY(C21.alloc(sizeof(double) * (size_t)(mb * nd)));
      GSS_TRY(w.alloc(sizeof(double) * (size_t)nd));
      GSS_TRY(work.alloc(sizeof(double) * (size_t)potrf_joint_work_doubles(nd, mb)));
      GSS_TRY(gwork.alloc(sizeof(double) * (size_t)gemv_work_doubles(false, ns, nd)));
      GSS_TRY(cov_pairwise_dev(h->vg, dxd.as<double>(), nd, dxd.as<double>(), nd, C11.as<double>(), nd, s));  // :131
      GSS_TRY(dev_zero_bytes(C21.p, C21.bytes, s));
      GSS_TRY(cov_pairwise_dev(h->vg, dxd.as<double>(), nd, dxs.as<double>(), ns, C21.as<double>(), mb, s));  // :132
      GSS_HIP(hipMemcpy2DAsync(C21.as<double>() + ns, sizeof(double) * mb, h->z1.p, sizeof(double), sizeof(double),
                               (size_t)nd, hipMemcpyDeviceToDevice, s));
      GSS_TRY(potrf_joint_f64(C11.as<double>(), nd, C21.as<double>(), mb, mb, C22, ns, info.as<int>(),
                              work.as<double>(), s));
      GSS_TRY(check_info(info, "data covariance C11", s));
      GSS_HIP(hipMemcpy2DAsync(w.p, sizeof(double), C21.as<double>() + ns, sizeof(double) * mb, sizeof(double),
                               (size_t)nd, hipMemcpyDeviceToDevice, s));
      GSS_TRY(gemv_f64(false, ns, nd, C21.as<double>(), mb, w.as<double>(), h->d2(), gwork.as<double>(), s));
      GSS_HIP(hipStreamSynchronize(s));
    }
    if (nd > 0 && use_lu) {
      DevBuf C11, W11, C21, A21, w, scr, gwork, ipiv;
      GSS_TRY(ipiv.alloc(sizeof(int) * (size_t)nd));
      GSS_TRY(C11.alloc(sizeof(double) * (size_t)(nd * nd)));
      GSS_TRY(W11.alloc(sizeof(double) * (size_t)(nd * nd)));
      const int64_t tq = nd / 2 + 64;   // trtri_f64's scratch requirement
      int64_t wscr = nd * nd > tq * tq ? nd * nd : tq * tq;
      if (potrf_inverse_work_doubles(nd) > wscr) wscr = potrf_inverse_work_doubles(nd);
      GSS_TRY(scr.alloc(sizeof(double) * (size_t)wscr));
      GSS_TRY(C21.alloc(sizeof(double) * (size_t)(ns * nd)));
      GSS_TRY(A21.alloc(sizeof(double) * (size_t)(ns * nd)));
      GSS_TRY(w.alloc(sizeof(double) * (size_t)nd));
      int64_t gw = gemv_work_doubles(false, ns, nd);
      if (gemv_work_doubles(false, nd, nd) > gw) gw = gemv_work_doubles(false, nd, nd);
      GSS_TRY(gwork.alloc(sizeof(double) * (size_t)gw));
      GSS_TRY(cov_pairwise_dev(h->vg, dxd.as<double>(), nd, dxd.as<double>(), nd, C11.as<double>(), nd, s));  // :131
      // row-major nd x ns == column-major ns x nd: C21                                                    // :132
      GSS_TRY(cov_pairwise_dev(h->vg, dxd.as<double>(), nd, dxs.as<double>(), ns, C21.as<double>(), ns, s));
      // L11 and W11 = inv(L11) (lu.jl:134): the two solves below become products with W11
      GSS_TRY(dev_zero_bytes(W11.p, W11.bytes, s));
      if (use_lu) {
        GSS_TRY(getrf_unit_lower_f64(C11.as<double>(), nd, nd, ipiv.as<int>(), info.as<int>(), s));
        GSS_TRY(check_info(info, "data covariance C11", s));
        GSS_TRY(trtri_f64(C11.as<double>(), nd, nd, W11.as<double>(), nd, scr.as<double>(), nullptr, s));
      } else {
        GSS_TRY(potrf_inverse_f64(C11.as<double>(), nd, nd, W11.as<double>(), nd, scr.as<double>(), info.as<int>(),
                                  false, s));
        GSS_TRY(check_info(info, "data covariance C11", s));
      }
      // A21 = C21 * inv(L11)'  (= B12', lu.jl:135)
      // (with lu, W11 = inv(L11) of the pivoted factorisation: A21 * B12 is then no Schur complement, but it is what
      //  lu.jl:135-139 computes)
      GSS_TRY(gemm_f64(ns, nd, nd, 1.0, C21.as<double>(), 1, ns, W11.as<double>(), nd, 1, 0.0, A21.as<double>(), 1, ns,
                       false, s, 1 /* W11' is upper triangular: half of the k-tiles are skipped */));
      // w = L11 \ z1 = W11 z1,  d2 = A21 w                                                               // :138
      GSS_TRY(gemv_f64(false, nd, nd, W11.as<double>(), nd, h->z1.as<double>(), w.as<double>(), gwork.as<double>(), s));
      GSS_TRY(gemv_f64(false, ns, nd, A21.as<double>(), ns, w.as<double>(), h->d2(), gwork.as<double>(), s));
      // C22 -= A21 * A21'  (lower tiles)                                                                  // :139
      GSS_TRY(gemm_f64(ns, ns, nd, -1.0, A21.as<double>(), 1, ns, A21.as<double>(), ns, 1, 1.0, C22, 1, ns, true, s));
      GSS_HIP(hipStreamSynchronize(s));
    }
    if (use_lu) {                                                                                           // :128/:139
      if (nd > 0) GSS_TRY(mirror_lower_f64(C22, ns, ns, s));   // the update above wrote the lower tiles only
      DevBuf ipiv2;
      GSS_TRY(ipiv2.alloc(sizeof(int) * (size_t)ns));
      GSS_TRY(getrf_unit_lower_f64(C22, ns, ns, ipiv2.as<int>(), info.as<int>(), s));
      GSS_TRY(check_info(info, nd > 0 ? "conditional covariance C22 - A21 B12" : "covariance C22", s));
    } else {
      DevBuf work22;
      GSS_TRY(work22.alloc(sizeof(double) * (size_t)potrf_blocked_work_doubles(ns)));
      GSS_TRY(potrf_blocked_f64(C22, ns, ns, info.as<int>(), work22.as<double>(), s));
      GSS_TRY(check_info(info, nd > 0 ? "conditional covariance C22 - A21 B12" : "covariance C22", s));
      hipLaunchKernelGGL(zero_upper_kernel, dim3((unsigned)((ns + 255) / 256), (unsigned)ns), dim3(256), 0, s, C22, ns,
                         ns);
      GSS_HIP(hipGetLastError());
    }
  }
  GSS_HIP(hipStreamSynchronize(s));
  h->ready = true;
  guard.h = nullptr;
  *out = h;
  return GSS_OK;
}

extern "C" {

int32_t gss_lugs_create(gss_lugs_t** out, const gss_variogram_t* vg, const double* centroids, int64_t N,
                        const int64_t* dlocs, const double* z1, int64_t nd, double mean, int32_t flags,
                        void* stream) {
  GSS_ENTRY();
  int32_t rc = lugs_create_impl(out, vg, centroids, N, dlocs, z1, nd, mean, flags, stream);
  // once more on the launch-per-block path (check_info has switched the single-launch kernel off for the process)
  if (rc == LUGS_RETRY) rc = lugs_create_impl(out, vg, centroids, N, dlocs, z1, nd, mean, flags, stream);
  if (rc == LUGS_RETRY) {
    set_error("gss_lugs_create: the factorisation kernel keeps giving up at a grid barrier");
    rc = GSS_ERR_HIP;
  }
  return rc;
}

int32_t gss_lugs_destroy(gss_lugs_t* h) {
  GSS_ENTRY();
  delete h;
  return GSS_OK;
}

int32_t gss_lugs_info(const gss_lugs_t* h, int64_t* ns, int64_t* nd) {
  GSS_ENTRY();
  GSS_REQUIRE(h != nullptr, "NULL handle");
  if (ns) *ns = h->ns;
  if (nd) *nd = h->nd;
  return GSS_OK;
}

int32_t gss_lugs_factor(gss_lugs_t* h, double* l22, double* d2, int32_t mem, void* stream) {
  GSS_ENTRY();
  GSS_REQUIRE(h != nullptr, "NULL handle");
  hipStream_t s = to_stream(stream);
  const hipMemcpyKind kind = mem == GSS_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  if (l22) GSS_HIP(hipMemcpyAsync(l22, h->L22(), sizeof(double) * (size_t)(h->ns * h->ns), kind, s));
  if (d2) GSS_HIP(hipMemcpyAsync(d2, h->d2(), sizeof(double) * (size_t)h->ns, kind, s));
  GSS_HIP(hipStreamSynchronize(s));
  return GSS_OK;
}

int32_t gss_lugs_state_buffer(gss_lugs_t* h, void** dev_ptr, int64_t* bytes) {
  GSS_ENTRY();
  GSS_REQUIRE(h != nullptr && dev_ptr != nullptr && bytes != nullptr, "NULL argument");
  *dev_ptr = h->state.p;
  *bytes = (int64_t)(sizeof(double) * (size_t)(h->ns * h->ns + h->ns));
  return GSS_OK;
}

int32_t gss_lugs_adopt_state(gss_lugs_t* h) {
  GSS_ENTRY();
  GSS_REQUIRE(h != nullptr, "NULL handle");
  h->ready = true;
  return GSS_OK;
}

// Realisations first_real .. first_real + R - 1 with every array in HBM (lu.jl:198-224); asynchronous on s.
static int32_t lugs_realize_block(gss_lugs_t* h, uint64_t seed, int64_t first_real, int64_t R, const double* noise,
                                  double rho, const double* w1, double* out, double* w_out, hipStream_t s) {
  const int64_t ns = h->ns, nd = h->nd, N = h->N;
  DevBuf w2buf, wmix, Y2;
  const double* w2 = noise;
  if (!noise) {                                                                                 // lu.jl:209
    double* dst = w_out;
    if (!dst) {
      GSS_TRY(w2buf.alloc(sizeof(double) * (size_t)(R * ns)));
      dst = w2buf.as<double>();
    }
    if (ns) {
      hipLaunchKernelGGL(philox_normal_batch_kernel, dim3((unsigned)((ns + 255) / 256), (unsigned)R), dim3(256), 0, s,
                         seed, first_real, ns, dst);
      GSS_HIP(hipGetLastError());
    }
    w2 = dst;
  } else if (w_out) {
    GSS_HIP(hipMemcpyAsync(w_out, w2, sizeof(double) * (size_t)(R * ns), hipMemcpyDeviceToDevice, s));
  }
  const double* weff = w2;
  if (w1 && ns) {                                                                               // lu.jl:213
    GSS_TRY(wmix.alloc(sizeof(double) * (size_t)(R * ns)));
    hipLaunchKernelGGL(mix_kernel, dim3((unsigned)((R * ns + 255) / 256)), dim3(256), 0, s, w1, w2, rho,
                       std::sqrt(1.0 - rho * rho), R * ns, wmix.as<double>());
    GSS_HIP(hipGetLastError());
    weff = wmix.as<double>();
  }
  GSS_TRY(Y2.alloc(sizeof(double) * (size_t)(R * ns)));
  DevBuf Ypart;
  if (ns) {
    ProfScope ps("lugs_gemm", s);
    // Y2 (ns x R, column-major) = L22 * W                                                      // lu.jl:211
    // A tall product with few columns is a handful of workgroups that each walk the whole of K (one memory round
    // trip per 16-deep stage: 1.2 ms for 12 288 x 100, an order of magnitude above the time to read L22).  It is
    // therefore cut along K into LUGS_KC column blocks of L22 -- block c only reaches the rows from its first column
    // down --, the blocks run side by side on the caller's and three helper streams into partial results, and a
    // kernel adds the partials in fixed order (the result does not depend on timing).
    static const bool split_on = [] {
      const char* e = std::getenv("GSS_LUGS_SPLITK");
      return !(e && e[0] == '0');
    }();
    hipStream_t hs[4] = {s, nullptr, nullptr, nullptr};
    bool split = split_on && ns >= 4096 && R <= 512;
    for (int i = 1; i < 4 && split; ++i) {
      hs[i] = realize_stream(i);
      if (!hs[i]) split = false;
    }
    if (!split) {
      GSS_TRY(gemm_f64(ns, R, ns, 1.0, h->L22(), 1, ns, weff, 1, ns, 0.0, Y2.as<double>(), 1, ns, false, s,
                       4 /* L22 is lower triangular: row tile i0 needs k < i0 + T only */));
    } else {
      const int64_t kc = ((ns + LUGS_KC - 1) / LUGS_KC + 127) / 128 * 128;   // columns per block, multiple of 128
      const int nchunk = (int)((ns + kc - 1) / kc);
      GSS_TRY(Ypart.alloc(sizeof(double) * (size_t)(R * ns) * (size_t)nchunk));
      ScopedEvent e0, e1[4];
      GSS_HIP(e0.create());
      GSS_HIP(hipEventRecord(e0, s));
      for (int i = 1; i < 4; ++i) GSS_HIP(hipStreamWaitEvent(hs[i], e0, 0));
      int32_t rc = GSS_OK;
      for (int c = 0; c < nchunk && rc == GSS_OK; ++c) {
        const int64_t c0 = (int64_t)c * kc, k = (ns - c0) < kc ? (ns - c0) : kc;
        rc = gemm_f64(ns - c0, R, k, 1.0, h->L22() + c0 + c0 * ns, 1, ns, weff + c0, 1, ns, 0.0,
                      Ypart.as<double>() + (int64_t)c * R * ns + c0, 1, ns, false, hs[c % 4], 4);
      }
      for (int i = 1; i < 4; ++i) {   // the caller's stream comes back behind the helpers whatever happened above
        GSS_HIP(e1[i].create());
        GSS_HIP(hipEventRecord(e1[i], hs[i]));
        GSS_HIP(hipStreamWaitEvent(s, e1[i], 0));
      }
      GSS_TRY(rc);
      hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)((ns + 255) / 256), (unsigned)R), dim3(256), 0, s,
                         Ypart.as<double>(), nchunk, kc, ns, R, Y2.as<double>());
      GSS_HIP(hipGetLastError());
    }
  }
  const double add = nd == 0 ? h->mean : 0.0;                                                   // lu.jl:221
  hipLaunchKernelGGL(lugs_scatter_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)R), dim3(256), 0, s,
                     Y2.as<double>(), h->d2(), h->slocs.as<int64_t>(), ns, h->z1.as<double>(),
                     h->dlocs.as<int64_t>(), nd, add, N, out);
  GSS_HIP(hipGetLastError());
  return GSS_OK;
}

int32_t gss_lugs_realize(gss_lugs_t* h, uint64_t seed, int64_t first_real, int64_t nreals, const double* noise,
                         double rho, const double* w1, double* out, double* w_out, int32_t mem, void* stream) {
  GSS_ENTRY();
  GSS_REQUIRE(h != nullptr && out != nullptr && nreals >= 0 && first_real >= 0, "gss_lugs_realize: bad arguments");
  GSS_REQUIRE(h->ready, "handle has no factor");
  GSS_REQUIRE(w1 == nullptr || (rho >= -1.0 && rho <= 1.0), "correlation %g outside [-1, 1]", rho);
  if (nreals == 0) return GSS_OK;
  hipStream_t s = to_stream(stream);
  const int64_t ns = h->ns, N = h->N;
  if (mem == GSS_MEM_DEVICE) {
    GSS_TRY(lugs_realize_block(h, seed, first_real, nreals, noise, rho, w1, out, w_out, s));
    GSS_HIP(hipStreamSynchronize(s));  // scratch buffers are freed on return
    return GSS_OK;
  }
  // Host arrays (lu.jl:217-221 returns host vectors): blocks of realisations that fit one chunk of the output ring;
  // block b's results (and normals) cross the bus while block b + 1 is computed, and HBM holds at most three blocks.
  const int64_t rb = OutStream::default_chunk(sizeof(double) * (size_t)N, nreals);
  OutStream os, osw;
  GSS_TRY(os.begin(out, sizeof(double) * (size_t)N, nreals, mem, s, rb));
  GSS_TRY(osw.begin(w_out, sizeof(double) * (size_t)ns, nreals, mem, s, rb));
  DevBuf nbuf, w1buf;
  if (noise) GSS_TRY(nbuf.alloc(sizeof(double) * (size_t)(rb * ns)));
  if (w1) GSS_TRY(w1buf.alloc(sizeof(double) * (size_t)(rb * ns)));
  for (int64_t r0 = 0; r0 < nreals; r0 += rb) {
    const int64_t n = nreals - r0 < rb ? nreals - r0 : rb;
    if (noise && ns)
      GSS_HIP(hipMemcpyAsync(nbuf.p, noise + r0 * ns, sizeof(double) * (size_t)(n * ns), hipMemcpyHostToDevice, s));
    if (w1 && ns)
      GSS_HIP(hipMemcpyAsync(w1buf.p, w1 + r0 * ns, sizeof(double) * (size_t)(n * ns), hipMemcpyHostToDevice, s));
    double *dout = nullptr, *dw = nullptr;
    GSS_TRY(os.slot(r0, s, &dout));
    if (w_out) GSS_TRY(osw.slot(r0, s, &dw));
    GSS_TRY(lugs_realize_block(h, seed, first_real + r0, n, noise ? nbuf.as<double>() : nullptr, rho,
                               w1 ? w1buf.as<double>() : nullptr, dout, dw, s));
    GSS_TRY(os.done(r0 + n - 1, s));
    GSS_TRY(osw.done(r0 + n - 1, s));
  }
  GSS_TRY(os.finish(s));
  GSS_TRY(osw.finish(s));
  return GSS_OK;
}

}  // extern "C"
