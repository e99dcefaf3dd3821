// FFTGS: FFT-based unconditional Gaussian simulation on Cartesian grids (Gutjahr 1997).
// Replaces preprocess (/root/reference/src/simulation/fft.jl:62-103) and solvesingle
// (fft.jl:145-173).  rocFFT does the transforms; everything else is fused into three kernels.
//
// preprocess, fft.jl:96-103:
//     C  = sill - gamma(centre cell, every cell)             fftgs_cov_kernel          (K6)
//     F  = sqrt(|fft(fftshift(C))|), F[1] = 0                rocFFT R2C + fftgs_amp_kernel
//   |fft| is invariant to the circular placement of C, so the fftshift is not materialised, and C is
//   real, so the real-to-complex transform gives the same |.| on the Hermitian half.
// solvesingle, fft.jl:163-170:
//     P  = F .* exp(im * angle(fft(rand)))                   noise kernel + R2C + fftgs_phase_kernel (K7)
//     Z  = real(ifft(P)); Z *= sqrt(sill / var(Z, mean=0)); Z += mean      rocFFT C2R
//   P is Hermitian, hence by Parseval sum(Z^2) = sum(F^2) / N for EVERY realisation: the rescale is
//   the constant s = sqrt(sill N (N-1) / sum F^2), folded with the 1/N of the unnormalised inverse
//   into the stored half-spectrum Fh = F s / N; the mean enters as the DC coefficient.
// HBM traffic per realisation (FP64, N cells): noise 8N (w) + R2C + phase 8N+4N (r) 8N (w) + C2R;
// the algorithmic floor is 32 N bytes.
#include "gss_internal.h"
#include "fftgs_fused.h"
#include "fftgs_generic.h"

#include <sys/stat.h>
#include <cerrno>
#include <string>
#include "philox.h"

#include <rocfft/rocfft.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <array>
#include <list>
#include <memory>
#include <mutex>
#include <utility>
#include <vector>

namespace gss {

#define GSS_FFT(call)                                                              \
  do {                                                                             \
    rocfft_status st__ = (call);                                                   \
    if (st__ != rocfft_status_success) {                                           \
      ::gss::set_error("%s:%d: %s failed: rocfft status %d", __FILE__, __LINE__, #call, (int)st__); \
      return GSS_ERR_HIP;                                                          \
    }                                                                              \
  } while (0)

struct GridSpec {
  int64_t n1, n2, n3;  // n1 fastest
  int64_t nh;          // n1 / 2 + 1
  int64_t c1, c2, c3;  // 0-based centre cell (dims div 2, 1-based, fft.jl:69)
  double s1, s2, s3;   // spacing
};

// C[e] = cov(|lag|) with lag = (cell - centre) * spacing, written as a contiguous real array
__global__ __launch_bounds__(256) void fftgs_cov_kernel(VgDev vg, GridSpec g, double* __restrict__ C) {
  const int64_t N = g.n1 * g.n2 * g.n3;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < N; e += (int64_t)gridDim.x * 256) {
    const int64_t i1 = e % g.n1, i2 = (e / g.n1) % g.n2, i3 = e / (g.n1 * g.n2);
    double a[3] = {(double)(i1 - g.c1) * g.s1, (double)(i2 - g.c2) * g.s2, (double)(i3 - g.c3) * g.s3};
    const double zero[3] = {0.0, 0.0, 0.0};
    C[e] = cov_pair<3>(vg, a, zero);
  }
}

constexpr int RED_BLOCKS = 1024;

// Fh[idx] = sqrt(|X[idx]|) (DC = 0); partial[b] = sum over this block's elements of w * F^2 where w
// counts how many full-spectrum entries the half-spectrum entry stands for
__global__ __launch_bounds__(256) void fftgs_amp_kernel(GridSpec g, const double2* __restrict__ X,
                                                        double* __restrict__ Fh, double* __restrict__ partial) {
  __shared__ double red[256];
  const int64_t NH = g.nh * g.n2 * g.n3;
  double acc = 0.0;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < NH; idx += (int64_t)gridDim.x * 256) {
    const int64_t k1 = idx % g.nh;
    const double2 x = X[idx];
    double f = sqrt(sqrt(x.x * x.x + x.y * x.y));
    if (idx == 0) f = 0.0;  // fft.jl:103
    Fh[idx] = f;
    const bool self = (k1 == 0) || (2 * k1 == g.n1);
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

// scal[0] = sum F^2, scal[1] = s / N with s = sqrt(sill N (N-1) / sum F^2)
__global__ void fftgs_scale_kernel(const double* __restrict__ partial, int nparts, double sill, double N,
                                   double* __restrict__ scal) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < nparts; ++i) s += partial[i];
    scal[0] = s;
    scal[1] = sqrt(sill * N * (N - 1.0) / s) / N;
  }
}

__global__ __launch_bounds__(256) void fftgs_apply_scale_kernel(double* __restrict__ Fh, int64_t NH,
                                                                const double* __restrict__ scal) {
  const double f = scal[1];
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < NH; idx += (int64_t)gridDim.x * 256)
    Fh[idx] *= f;
}

// K7: X <- Fh * X / |X| (phase of the noise spectrum, amplitude of the covariance); DC <- mean
// (blockIdx.y: member of a batch of realisations, half spectra NH apart)
__global__ __launch_bounds__(256) void fftgs_phase_kernel(double2* __restrict__ X, const double* __restrict__ Fh,
                                                          int64_t NH, double mean) {
  X += (int64_t)blockIdx.y * NH;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < NH; idx += (int64_t)gridDim.x * 256) {
    const double2 x = X[idx];
    const double f = Fh[idx];
    const double mag2 = x.x * x.x + x.y * x.y;
    double2 p;
    if (mag2 > 0.0) {
      const double inv = f / sqrt(mag2);
      p.x = x.x * inv;
      p.y = x.y * inv;
    } else {  // angle(0) = 0
      p.x = f;
      p.y = 0.0;
    }
    if (idx == 0) {
      p.x = mean;
      p.y = 0.0;
    }
    X[idx] = p;
  }
}

// full-size F from the scaled half-spectrum (parity checks): F(k) = Fh(hermitian partner) / scal[1]
__global__ __launch_bounds__(256) void fftgs_expand_kernel(GridSpec g, const double* __restrict__ Fh,
                                                           const double* __restrict__ scal,
                                                           double* __restrict__ F) {
  const int64_t N = g.n1 * g.n2 * g.n3;
  const double inv = 1.0 / scal[1];
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < N; e += (int64_t)gridDim.x * 256) {
    int64_t k1 = e % g.n1, k2 = (e / g.n1) % g.n2, k3 = e / (g.n1 * g.n2);
    if (k1 >= g.nh) {
      k1 = g.n1 - k1;
      k2 = (g.n2 - k2) % g.n2;
      k3 = (g.n3 - k3) % g.n3;
    }
    F[e] = Fh[k1 + g.nh * (k2 + g.n2 * k3)] * inv;
  }
}

// (blockIdx.y: member of a batch of realisations, sources sbs and destinations n doubles apart)
__global__ __launch_bounds__(256) void gather_kernel(const double* __restrict__ src, const int64_t* __restrict__ inds,
                                                     int64_t n, double* __restrict__ dst, int64_t sbs = 0) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[(int64_t)blockIdx.y * n + i] = src[(int64_t)blockIdx.y * sbs + inds[i]];
}

static std::once_flag g_rocfft_once;

}  // namespace gss

using namespace gss;

// The two rocFFT plans of a grid size.  A plan describes the transform and is executed with the execution info of the
// handle that runs it (its own work buffer and stream), so handles of the same size share one pair; the library keeps
// the pairs of the last eight sizes alive between handles -- creating a pair costs 2-3 ms even when rocFFT finds its
// compiled kernels, which is most of a `solve` on the 100 x 100 grids of the reference's tests
// (tools/small_problem_latency.py).
struct FftPlans {
  rocfft_plan fwd = nullptr, inv = nullptr;
  size_t work_bytes = 0;
  // the same transforms for `batch` realisations at a time (small grids: the pipeline is four to ten launches whatever
  // the grid, created on the first call with enough realisations)
  rocfft_plan fwdB = nullptr, invB = nullptr;
  int batch = 0;
  size_t work_bytesB = 0;
  ~FftPlans() {
    if (fwd) rocfft_plan_destroy(fwd);
    if (inv) rocfft_plan_destroy(inv);
    if (fwdB) rocfft_plan_destroy(fwdB);
    if (invB) rocfft_plan_destroy(invB);
  }
};

struct gss_fftgs {
  VgDev vg;
  GridSpec g;
  int ndim = 0;
  int64_t N = 0, NH = 0;
  double mean = 0.0;
  // rocFFT pipeline (general grids): plans (shared with other handles of the same grid size: fft_plans), work buffer,
  // U and Xn are created on first use
  std::shared_ptr<struct FftPlans> plans;
  rocfft_plan fwd = nullptr, inv = nullptr;   // = plans->fwd / inv
  rocfft_execution_info info = nullptr;
  DevBuf state;  // Fh (NH doubles) followed by scal[2]
  DevBuf U, Xn, work, Z;
  bool ready = false;  // state holds a spectrum (computed here or adopted after a broadcast)
  // fused pipeline (power-of-two 3-D grids)
  bool fused = false;
  FusedGrid fg;
  DevBuf X, tw1, tw2, tw3, Fh_tiled, covsrc;
  // realisation pipeline: P1 (instruction-issue bound) of realisation r+1 runs on a helper stream beside P2..P5
  // (HBM bound) of realisation r; two half-spectrum buffers alternate
  DevBuf X2;
  double2* Xcur = nullptr;      // buffer the launch helpers work on
  hipStream_t s2 = nullptr;
  hipEvent_t ev_p1[2] = {nullptr, nullptr}, ev_p5[2] = {nullptr, nullptr}, ev_in = nullptr;
  int overlap = 1;
  DevBuf xtw;                   // per-pass twiddle tables of the Stockham x passes
  int x_rows = 8;               // x lines per workgroup of the Stockham x passes (rows * M / 8 <= 256)
  int txy_log = 3, txz_log = 3; // log2 of the tile width (columns) of the y and z passes
  // generic pipeline (2-D grids, sizes 2^a 3^b 5^c 7^d: fftgs_generic.h): Stockham plans per axis and their twiddle tables;
  // the half-spectrum buffer is X, the amplitudes are read from the state in their natural layout
  bool generic = false;
  GenGrid gg;
  GenPlan gp[3];
  DevBuf gtab[3];
  int g_rows = 1, g_txlog[3] = {3, 3, 3};
  bool g_tg = false;   // x passes read their tables from global memory (long lines)
  int g_batch = 1;     // realisations per launch (small grids: X holds g_batch half-spectrum buffers, g_xbs elements apart)
  int64_t g_xbs = 0;
  // long y lines of 2-D grids (1 024 < n2 <= 4 096): n2 = L1 L2, plans and tables of the two short transforms, W_n2
  bool g_long = false;
  GenLong gl;
  GenPlan gpl[2];
  DevBuf gltab[2], gtwl;
  // slab order of the strided passes: slab i runs on stream i mod ns (0 = the caller's, the others are helper streams
  // of the process, slab_stream()); the events that fence them belong to the handle
  static constexpr int SLAB_MAX_STREAMS = 4;
  hipEvent_t slab_e0 = nullptr, slab_e[SLAB_MAX_STREAMS] = {nullptr, nullptr, nullptr, nullptr};
  double* Fh() const { return state.as<double>(); }
  double* scal() const { return state.as<double>() + NH; }
  ~gss_fftgs() {
    if (s2) {
      (void)hipStreamSynchronize(s2);
      (void)hipStreamDestroy(s2);
    }
    for (int b = 0; b < 2; ++b) {
      if (ev_p1[b]) (void)hipEventDestroy(ev_p1[b]);
      if (ev_p5[b]) (void)hipEventDestroy(ev_p5[b]);
    }
    if (ev_in) (void)hipEventDestroy(ev_in);
    for (int i = 0; i < SLAB_MAX_STREAMS; ++i)
      if (slab_e[i]) {
        (void)hipEventSynchronize(slab_e[i]);   // the helper streams may still be working on this handle's buffers
        (void)hipEventDestroy(slab_e[i]);
      }
    if (slab_e0) (void)hipEventDestroy(slab_e0);
    if (info) rocfft_execution_info_destroy(info);   // (the plans go with the last holder of `plans`)
  }
};

static int grid_blocks(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b < 256 * 16 ? (b < 1 ? 1 : b) : 256 * 16);
}

static int32_t fft_exec(gss_fftgs* h, rocfft_plan plan, void* in, void* out, hipStream_t s) {
  GSS_FFT(rocfft_execution_info_set_stream(h->info, s));
  void* ib[1] = {in};
  void* ob[1] = {out};
  GSS_FFT(rocfft_execute(plan, ib, ob, h->info));
  return GSS_OK;
}

// rocFFT plans (run-time compiled: ~1.5 s the first time a size is seen), their work buffer and the natural-layout
// buffers of the general pipeline; the fused pipeline never needs them
// rocFFT compiles the kernels of a grid size at run time (0.5 - 1.7 s per new size) and, left alone, forgets them
// when the process ends.  Unless the user chose a place (ROCFFT_RTC_CACHE_PATH), the compiled kernels are kept in
// $XDG_CACHE_HOME or ~/.cache, gss_hip/rocfft_kernels.db: the first plan of a later process then costs 0.1 s and
// further sizes a few milliseconds (measured: 1 024 x 1 024 cells 1.7 s -> 0.11 s, 100 x 100 cells 1.2 s -> 4 ms).
static void rocfft_kernel_cache_default() {
  if (std::getenv("ROCFFT_RTC_CACHE_PATH")) return;
  std::string dir;
  if (const char* x = std::getenv("XDG_CACHE_HOME")) dir = x;
  else if (const char* hm = std::getenv("HOME")) dir = std::string(hm) + "/.cache";
  if (dir.empty()) return;
  (void)mkdir(dir.c_str(), 0700);            // an existing directory is fine; a failure leaves rocFFT on its own
  dir += "/gss_hip";
  if (mkdir(dir.c_str(), 0700) != 0 && errno != EEXIST) return;
  (void)setenv("ROCFFT_RTC_CACHE_PATH", (dir + "/rocfft_kernels.db").c_str(), 0);
}

static int32_t ensure_rocfft(gss_fftgs* h) {
  if (h->fwd) return GSS_OK;
  std::call_once(g_rocfft_once, [] {
    rocfft_kernel_cache_default();
    rocfft_setup();
  });
  using Key = std::array<size_t, 5>;   // rocFFT plans belong to the device that was current when they were made
  // most recent first; calls hold the library lock; never destroyed (at exit rocFFT's own statics may be gone first)
  static auto& cache = *new std::list<std::pair<Key, std::shared_ptr<FftPlans>>>();
  int dev = 0;
  GSS_HIP(hipGetDevice(&dev));
  const Key key = {(size_t)h->ndim, (size_t)h->g.n1, (size_t)h->g.n2, (size_t)h->g.n3, (size_t)dev};
  for (auto it = cache.begin(); it != cache.end(); ++it)
    if (it->first == key) {
      h->plans = it->second;
      cache.splice(cache.begin(), cache, it);
      break;
    }
  if (!h->plans) {
    auto pl = std::make_shared<FftPlans>();
    size_t lengths[3] = {(size_t)h->g.n1, (size_t)h->g.n2, (size_t)h->g.n3};
    GSS_FFT(rocfft_plan_create(&pl->fwd, rocfft_placement_notinplace, rocfft_transform_type_real_forward,
                               rocfft_precision_double, (size_t)h->ndim, lengths, 1, nullptr));
    GSS_FFT(rocfft_plan_create(&pl->inv, rocfft_placement_notinplace, rocfft_transform_type_real_inverse,
                               rocfft_precision_double, (size_t)h->ndim, lengths, 1, nullptr));
    size_t w1 = 0, w2 = 0;
    GSS_FFT(rocfft_plan_get_work_buffer_size(pl->fwd, &w1));
    GSS_FFT(rocfft_plan_get_work_buffer_size(pl->inv, &w2));
    pl->work_bytes = w1 > w2 ? w1 : w2;
    cache.emplace_front(key, pl);
    if (cache.size() > 8) cache.pop_back();   // handles that still use the pair keep it alive
    h->plans = std::move(pl);
  }
  h->fwd = h->plans->fwd;
  h->inv = h->plans->inv;
  const size_t wb = h->plans->work_bytes;
  GSS_FFT(rocfft_execution_info_create(&h->info));
  if (wb > 0) {
    GSS_TRY(h->work.alloc(wb));
    GSS_FFT(rocfft_execution_info_set_work_buffer(h->info, h->work.p, wb));
  }
  GSS_TRY(h->U.alloc(sizeof(double) * (size_t)h->N));
  GSS_TRY(h->Xn.alloc(sizeof(double) * 2 * (size_t)h->NH));
  return GSS_OK;
}

static int env_int(const char* name, int dflt) {
  const char* e = std::getenv(name);
  return e && *e ? std::atoi(e) : dflt;
}

// Batched plans of the rocFFT pipeline: 64 realisations per execution when the call has that many and 64 noise arrays
// and half spectra fit 96 MiB, else 16, else none (0).  The plans live with the single ones (shared per grid size).
static int32_t ensure_rocfft_batch(gss_fftgs* h, int64_t nreals, int* batch_out) {
  *batch_out = 0;
  const int env_b = env_int("GSS_FFTGS_ROCFFT_BATCH", -1);   // 0: off; n: that batch size (read per call)
  const size_t per = sizeof(double) * ((size_t)h->N + 2 * (size_t)h->NH);
  int want = 0;
  for (int b : {64, 16})
    if (nreals >= b && per * (size_t)b <= ((size_t)96 << 20)) { want = b; break; }
  if (env_b == 0) want = 0;
  if (env_b > 0) want = nreals >= env_b ? env_b : 0;
  if (want == 0) return GSS_OK;
  FftPlans* pl = h->plans.get();
  if (pl->batch != want) {
    if (pl->fwdB) rocfft_plan_destroy(pl->fwdB);
    if (pl->invB) rocfft_plan_destroy(pl->invB);
    pl->fwdB = pl->invB = nullptr;
    pl->batch = 0;
    size_t lengths[3] = {(size_t)h->g.n1, (size_t)h->g.n2, (size_t)h->g.n3};
    GSS_FFT(rocfft_plan_create(&pl->fwdB, rocfft_placement_notinplace, rocfft_transform_type_real_forward,
                               rocfft_precision_double, (size_t)h->ndim, lengths, (size_t)want, nullptr));
    GSS_FFT(rocfft_plan_create(&pl->invB, rocfft_placement_notinplace, rocfft_transform_type_real_inverse,
                               rocfft_precision_double, (size_t)h->ndim, lengths, (size_t)want, nullptr));
    size_t w1 = 0, w2 = 0;
    GSS_FFT(rocfft_plan_get_work_buffer_size(pl->fwdB, &w1));
    GSS_FFT(rocfft_plan_get_work_buffer_size(pl->invB, &w2));
    pl->work_bytesB = w1 > w2 ? w1 : w2;
    pl->batch = want;
  }
  if (pl->work_bytesB > h->work.bytes) {
    GSS_TRY(h->work.alloc(pl->work_bytesB));
    GSS_FFT(rocfft_execution_info_set_work_buffer(h->info, h->work.p, pl->work_bytesB));
  }
  if (h->U.bytes < sizeof(double) * (size_t)h->N * (size_t)want) GSS_TRY(h->U.alloc(sizeof(double) * (size_t)h->N * (size_t)want));
  if (h->Xn.bytes < sizeof(double) * 2 * (size_t)h->NH * (size_t)want)
    GSS_TRY(h->Xn.alloc(sizeof(double) * 2 * (size_t)h->NH * (size_t)want));
  *batch_out = want;
  return GSS_OK;
}

static bool pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }
static int ilog2(int64_t v) {
  int l = 0;
  while ((1LL << l) < v) ++l;
  return l;
}

static int32_t upload_twiddles(DevBuf& buf, int L, hipStream_t s) {
  std::vector<double> t((size_t)L);  // L/2 complex: exp(-2 pi i k / L)
  const long double two_pi = 6.283185307179586476925286766559005768L;
  for (int k = 0; k < L / 2; ++k) {
    const long double a = two_pi * (long double)k / (long double)L;
    t[(size_t)(2 * k)] = (double)cosl(a);
    t[(size_t)(2 * k + 1)] = (double)(-sinl(a));
  }
  GSS_TRY(buf.alloc(sizeof(double) * (size_t)L));
  GSS_HIP(hipMemcpyAsync(buf.p, t.data(), sizeof(double) * (size_t)L, hipMemcpyHostToDevice, s));
  GSS_HIP(hipStreamSynchronize(s));
  return GSS_OK;
}

static size_t ff_xfwd2_lds(int M, int logM, int rows) { return sizeof(double2) * (size_t)(M + x_table_len(logM) + rows * M); }
static size_t ff_xinv2_lds(int M, int logM, int rows) { return sizeof(double2) * (size_t)(x_table_len(logM) + rows * M); }

// per-pass twiddle tables of the Stockham x passes (forward sign; the inverse conjugates): middle passes
// T[(r-1) Ns + k] = exp(-2 pi i r k / (8 Ns)), then the last pass exp(-2 pi i r k / M)
static int32_t upload_x_tables(DevBuf& buf, int logM, hipStream_t s) {
  const int M = 1 << logM;
  const XPlan plan = x_plan(logM);
  std::vector<double> t((size_t)(2 * x_table_len(logM)));
  const long double two_pi = 6.283185307179586476925286766559005768L;
  size_t o = 0;
  int Ns = 8;
  for (int m = 0; m < plan.nmid; ++m) {
    for (int r = 1; r < 8; ++r)
      for (int k = 0; k < Ns; ++k) {
        const long double a = two_pi * (long double)(r * k) / (long double)(8 * Ns);
        t[o++] = (double)cosl(a);
        t[o++] = (double)(-sinl(a));
      }
    Ns <<= 3;
  }
  const int R = 1 << plan.ns_last;
  for (int r = 1; r < R; ++r)
    for (int k = 0; k < M / R; ++k) {
      const long double a = two_pi * (long double)(r * k) / (long double)M;
      t[o++] = (double)cosl(a);
      t[o++] = (double)(-sinl(a));
    }
  GSS_TRY(buf.alloc(sizeof(double) * t.size()));
  GSS_HIP(hipMemcpyAsync(buf.p, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice, s));
  GSS_HIP(hipStreamSynchronize(s));
  return GSS_OK;
}

static size_t ff_axis2_lds(int L, int txlog) { return sizeof(double2) * (size_t)(L / 2 + (L << txlog) + FF2_PAD); }

// strided pass `mode` (0 forward, 1 inverse, 2 forward-phase-inverse) along y (axis 1) or z (axis 2)
template <int MODE>
static int32_t launch_axis_mode(gss_fftgs* h, int axis, hipStream_t s, int slab_t0 = 0, int slab_nt = 0) {
  FusedGrid f = h->fg;
  f.slab_t0 = slab_t0;
  f.slab_nt = slab_nt;
  const int L = axis == 1 ? f.n2 : f.n3, logL = axis == 1 ? f.l2 : f.l3, nouter = axis == 1 ? f.n3 : f.n2;
  const double2* tw = axis == 1 ? h->tw2.as<double2>() : h->tw3.as<double2>();
  const int64_t ostride = axis == 1 ? (int64_t)f.n2 * f.nhp : (int64_t)f.nhp;
  const int64_t lstride = axis == 1 ? (int64_t)f.nhp : (int64_t)f.n2 * f.nhp;
  double2* X = h->Xcur;
  const double* fh = MODE == 2 ? h->Fh_tiled.as<double>() : nullptr;
  const double mean = MODE == 2 ? h->mean : 0.0;
  const int txlog = axis == 1 ? h->txy_log : h->txz_log;
  const unsigned blocks = (unsigned)(nouter * (slab_nt > 0 ? slab_nt : (f.nhp >> txlog)));
  const size_t lds = ff_axis2_lds(L, txlog);
  if (slab_nt > 0 && !(txlog == 3 && logL == 9)) return GSS_ERR_UNSUPPORTED;
  if (txlog == 3 && logL == 9)      // 512-point lines: the pass-by-pass kernel with every load issued up front
    hipLaunchKernelGGL((ff_axis2_fast_kernel<MODE, 3, 512, 9>), dim3(blocks), dim3(512), lds, s, f, tw, ostride, lstride,
                       X, fh, mean);
  else if (txlog == 3)
    hipLaunchKernelGGL((ff_axis2_kernel<MODE, 3, 512, -1>), dim3(blocks), dim3(512), lds, s, f, logL, tw, ostride, lstride,
                       X, fh, mean);
  else
    hipLaunchKernelGGL((ff_axis2_kernel<MODE, 2, 256, -1>), dim3(blocks), dim3(256), lds, s, f, logL, tw, ostride, lstride,
                       X, fh, mean);
  GSS_HIP(hipGetLastError());
  return GSS_OK;
}

// The fused five-pass pipeline serves 3-D grids whose sizes are powers of two (32..1024 along x, 16..1024 along
// y and z); everything else, and GSS_FFTGS_PATH=rocfft, stays on the rocFFT pipeline.
static int32_t fftgs_setup_fused(gss_fftgs* h, hipStream_t s) {
  const char* e = std::getenv("GSS_FFTGS_PATH");
  if (e && (std::strcmp(e, "rocfft") == 0 || std::strcmp(e, "generic") == 0)) return GSS_OK;
  const GridSpec& g = h->g;
  if (h->ndim != 3 || !pow2(g.n1) || !pow2(g.n2) || !pow2(g.n3)) return GSS_OK;
  if (g.n1 < 32 || g.n1 > 1024 || g.n2 < 16 || g.n2 > 1024 || g.n3 < 16 || g.n3 > 1024) return GSS_OK;
  // Small grids are left to the generic passes, where eight and more realisations share every launch (measured, ms per
  // realisation, fused / generic: 32^3 0.019 / 0.003, 64^3 0.024 / 0.011, 128 x 128 x 64 0.049 / 0.034, 128^3 0.064 /
  // 0.059, 256 x 128 x 128 0.095 / 0.127): up to a 12 MiB half spectrum.  GSS_FFTGS_PATH=fused keeps them here.
  const bool force = e && std::strcmp(e, "fused") == 0;
  if (!force && sizeof(double2) * (size_t)((g.nh + 7) / 8 * 8) * g.n2 * g.n3 <= ((size_t)12 << 20)) return GSS_OK;
  FusedGrid f;
  f.n1 = (int)g.n1; f.n2 = (int)g.n2; f.n3 = (int)g.n3;
  f.l1 = ilog2(g.n1); f.l2 = ilog2(g.n2); f.l3 = ilog2(g.n3);
  f.nh = (int)g.nh;
  f.nhp = (f.nh + FF_PITCH_ALIGN - 1) / FF_PITCH_ALIGN * FF_PITCH_ALIGN;
  f.ntx = f.nhp / FF_TX;
  f.slab_t0 = f.slab_nt = 0;
  h->fg = f;
  GSS_TRY(upload_twiddles(h->tw1, f.n1, s));
  GSS_TRY(upload_twiddles(h->tw2, f.n2, s));
  GSS_TRY(upload_twiddles(h->tw3, f.n3, s));
  const int64_t nt = (int64_t)f.n2 * f.ntx * f.n3 * FF_TX;
  GSS_TRY(h->Fh_tiled.alloc(sizeof(double) * (size_t)nt));
  const int M = f.n1 / 2;
  const int lmax = f.n2 > f.n3 ? f.n2 : f.n3;
  // Stockham x passes: as many lines per workgroup as give every thread one radix-8 item
  h->x_rows = M <= 256 ? 8 : 4;
  GSS_TRY(upload_x_tables(h->xtw, f.l1 - 1, s));
#define GSS_X2_ATTR(ROWS, LOGM)                                                                                       \
  do {                                                                                                                \
    GSS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ff_x_fwd2_kernel<FF_SRC_PHILOX, ROWS, 256, LOGM>),      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)ff_xfwd2_lds(M, f.l1 - 1, ROWS)));  \
    GSS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ff_x_fwd2_kernel<FF_SRC_ARRAY, ROWS, 256, LOGM>),       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)ff_xfwd2_lds(M, f.l1 - 1, ROWS)));  \
    GSS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ff_x_fwd2_kernel<FF_SRC_COV, ROWS, 256, LOGM>),         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)ff_xfwd2_lds(M, f.l1 - 1, ROWS)));  \
    GSS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ff_x_inv2_kernel<ROWS, 256, LOGM>),                    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)ff_xinv2_lds(M, f.l1 - 1, ROWS)));  \
  } while (0)
  // 512-cell lines (M = 256) have their own instantiation with the length fixed at compile time
  if (f.l1 == 9) GSS_X2_ATTR(8, 8);
  else if (h->x_rows == 8) GSS_X2_ATTR(8, -1);
  else GSS_X2_ATTR(4, -1);
#undef GSS_X2_ATTR
  // the half-spectrum buffer of the fused path has the padded row pitch; padding columns stay zero
  GSS_TRY(h->X.alloc(sizeof(double2) * (size_t)f.nhp * f.n2 * f.n3));
  GSS_TRY(dev_zero_bytes(h->X.p, h->X.bytes, s));
  h->Xcur = h->X.as<double2>();
  // measured (profiles/r02_fftgs_overlap.txt): the two-stream pipeline changes nothing at 512^3 (2.378 against
  // 2.367 ms per realisation) -- the passes fill the chip and the queues take turns; kept as an A/B switch, off
  h->overlap = env_int("GSS_FFTGS_OVERLAP", 0) != 0;
  // strided passes: 8-column tiles (4 when a 1024-point line would not leave room in LDS)
  h->txy_log = f.n2 > 512 ? 2 : 3;
  h->txz_log = f.n3 > 512 ? 2 : 3;
#define GSS_A2_ATTR(MODE, TXL, NT, LOGL)                                                                            \
  GSS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ff_axis2_kernel<MODE, TXL, NT, LOGL>),                  \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)ff_axis2_lds(lmax, TXL)))
  if (h->txy_log == 3 || h->txz_log == 3) {
    GSS_A2_ATTR(0, 3, 512, -1); GSS_A2_ATTR(1, 3, 512, -1); GSS_A2_ATTR(2, 3, 512, -1);
    if (lmax >= 512) {  // 512-point lines have their own kernel (length fixed at compile time)
      GSS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ff_axis2_fast_kernel<0, 3, 512, 9>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)ff_axis2_lds(512, 3)));
      GSS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ff_axis2_fast_kernel<1, 3, 512, 9>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)ff_axis2_lds(512, 3)));
      GSS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ff_axis2_fast_kernel<2, 3, 512, 9>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)ff_axis2_lds(512, 3)));
    }
  }
  GSS_A2_ATTR(0, 2, 256, -1); GSS_A2_ATTR(1, 2, 256, -1); GSS_A2_ATTR(2, 2, 256, -1);
#undef GSS_A2_ATTR
  h->fused = true;
  return GSS_OK;
}

// ---- generic pipeline (fftgs_generic.h) ---------------------------------------------------------------------------------
static hipStream_t slab_stream(int i);   // helper streams of the slab order (defined with the fused pipeline below)
static bool gen_plan(int L, GenPlan* pl) {
  std::memset(pl, 0, sizeof(*pl));
  pl->L = L;
  int n = L, np = 0;
  while (n % 7 == 0 && np < GEN_MAX_PASSES - 4) { pl->radix[np++] = 7; n /= 7; }
  while (n % 5 == 0) { pl->radix[np++] = 5; n /= 5; if (np >= GEN_MAX_PASSES - 4) break; }
  while (n % 3 == 0 && np < GEN_MAX_PASSES - 4) { pl->radix[np++] = 3; n /= 3; }
  int e = 0;
  while (n % 2 == 0) { ++e; n /= 2; }
  if (n != 1 || L < 2) return false;
  while (e >= 3 && np < GEN_MAX_PASSES) { pl->radix[np++] = 8; e -= 3; }
  if (e == 2 && np < GEN_MAX_PASSES) { pl->radix[np++] = 4; e = 0; }
  if (e == 1 && np < GEN_MAX_PASSES) { pl->radix[np++] = 2; e = 0; }
  if (e != 0) return false;
  pl->npass = np;
  int off = 0, Ns = 1;
  for (int p = 0; p < np; ++p) {
    pl->toff[p] = off;
    off += (pl->radix[p] - 1) * Ns;
    Ns *= pl->radix[p];
  }
  pl->tlen = off;
  pl->has7 = np > 0 && pl->radix[0] == 7;
  return Ns == L;
}

static int32_t gen_upload_table(DevBuf& buf, const GenPlan& pl, hipStream_t s) {
  std::vector<double> t((size_t)2 * (pl.tlen > 0 ? pl.tlen : 1), 0.0);
  const long double two_pi = 6.283185307179586476925286766559005768L;
  int Ns = 1;
  for (int p = 0; p < pl.npass; ++p) {
    const int R = pl.radix[p];
    for (int r = 1; r < R; ++r)
      for (int k = 0; k < Ns; ++k) {
        const long double a = two_pi * (long double)r * (long double)k / ((long double)Ns * (long double)R);
        const size_t e = (size_t)(pl.toff[p] + (r - 1) * Ns + k);
        t[2 * e] = (double)cosl(a);
        t[2 * e + 1] = (double)(-sinl(a));
      }
    Ns *= R;
  }
  GSS_TRY(buf.alloc(sizeof(double) * t.size()));
  GSS_HIP(hipMemcpyAsync(buf.p, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice, s));
  GSS_HIP(hipStreamSynchronize(s));
  return GSS_OK;
}

static size_t gen_x_lds(const GenPlan& pl, int rows, bool tg) { return sizeof(double2) * (size_t)((tg ? 0 : pl.L + pl.tlen) + rows * pl.L); }
static size_t gen_axis_lds(const GenPlan& pl, int txlog) { return sizeof(double2) * (size_t)(pl.tlen + (pl.L << txlog)); }

// 1-D and 2-D grids and 3-D grids that the power-of-two pipeline does not take, sizes 2^a 3^b 5^c 7^d: n1 even with n1 / 2 <= 2 048,
// the other axes <= 1 024.  Everything else stays on rocFFT.
static int32_t fftgs_setup_generic(gss_fftgs* h, hipStream_t s) {
  const char* e = std::getenv("GSS_FFTGS_PATH");
  if (e && std::strcmp(e, "rocfft") == 0) return GSS_OK;
  const GridSpec& g = h->g;
  if (h->fused || (g.n1 & 1) || g.n1 / 2 > 2048 || g.n1 < 4 || g.n3 > 1024) return GSS_OK;
  const bool lng = h->ndim == 2 && g.n2 > 1024;
  if (g.n2 > (lng ? 4096 : 1024)) return GSS_OK;
  GenPlan p1, p2, p3;
  std::memset(&p2, 0, sizeof(p2));
  if (!gen_plan((int)(g.n1 / 2), &p1)) return GSS_OK;
  if (h->ndim >= 2 && !gen_plan((int)g.n2, &p2)) return GSS_OK;   // (1-D grids: the x passes and an elementwise phase step)
  if (lng) {
    // n2 = L1 L2 with both factors <= 64 (L2 the largest such divisor); tiles of at most 4 096 elements
    GenLong& gl = h->gl;
    gl.L2 = 0;
    for (int d = 64; d >= 2; --d)
      if (g.n2 % d == 0 && g.n2 / d <= 64) { gl.L2 = d; break; }
    if (gl.L2 == 0) return GSS_OK;
    gl.L1 = (int)g.n2 / gl.L2;
    if (!gen_plan(gl.L1, &h->gpl[0]) || !gen_plan(gl.L2, &h->gpl[1])) return GSS_OK;
    gl.NB = gl.NC = 1;
    for (int d = 1; d <= gl.L2; ++d)
      if (gl.L2 % d == 0 && gl.L1 * d <= (h->gpl[0].has7 ? 448 : 512)) gl.NB = d;
    for (int d = 1; d <= gl.L1; ++d)
      if (gl.L1 % d == 0 && gl.L2 * d <= (h->gpl[1].has7 ? 448 : 512)) gl.NC = d;
  }
  if (h->ndim == 3 && !gen_plan((int)g.n3, &p3)) return GSS_OK;
  if (h->ndim < 3) std::memset(&p3, 0, sizeof(p3));
  h->gp[0] = p1; h->gp[1] = p2; h->gp[2] = p3;
  GenGrid& gg = h->gg;
  gg.n1 = (int)g.n1; gg.n2 = (int)g.n2; gg.n3 = (int)g.n3;
  gg.nh = (int)g.nh;
  gg.nhp = (gg.nh + 7) / 8 * 8;
  gg.ndim = h->ndim;
  gg.c1 = (int)g.c1; gg.c2 = (int)g.c2; gg.c3 = (int)g.c3;
  gg.s1 = g.s1; gg.s2 = g.s2; gg.s3 = g.s3;
  h->g_rows = (int)((p1.has7 ? 1792 : 2048) / p1.L);
  if (h->g_rows > 16) h->g_rows = 16;
  if (h->g_rows < 1) return GSS_OK;          // (a line of 7 x 2^k > 1 792 elements: rocFFT)
  {
    // tables in the LDS unless they cost occupancy: more workgroups per CU (of 160 KB, at most 8) without them
    const int with = (int)((size_t)(160 << 10) / gen_x_lds(p1, h->g_rows, false));
    const int without = (int)((size_t)(160 << 10) / gen_x_lds(p1, h->g_rows, true));
    static const int tg_env = env_int("GSS_FFTGS_GEN_TG", -1);
    h->g_tg = tg_env >= 0 ? tg_env != 0 : (with <= 2 && without > with);   // (three and more: measured neutral, 1 000^2)
  }
  h->g_txlog[1] = p2.L <= (p2.has7 ? 448 : 512) ? 3 : 2;
  h->g_txlog[2] = (h->ndim == 3 && p3.L > (p3.has7 ? 448 : 512)) ? 2 : 3;
  if (!lng && p2.has7 && p2.L > 896) return GSS_OK;      // (980, 1 008: four columns of the line exceed 3 584 elements)
  if (h->ndim == 3 && p3.has7 && p3.L > 896) return GSS_OK;
  GSS_TRY(upload_twiddles(h->tw1, gg.n1, s));
  GSS_TRY(gen_upload_table(h->gtab[0], p1, s));
  if (lng) {
    GSS_TRY(gen_upload_table(h->gltab[0], h->gpl[0], s));
    GSS_TRY(gen_upload_table(h->gltab[1], h->gpl[1], s));
    std::vector<double> w((size_t)2 * g.n2);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (int64_t k = 0; k < g.n2; ++k) {
      const long double a = two_pi * (long double)k / (long double)g.n2;
      w[2 * k] = (double)cosl(a);
      w[2 * k + 1] = (double)(-sinl(a));
    }
    GSS_TRY(h->gtwl.alloc(sizeof(double) * w.size()));
    GSS_HIP(hipMemcpyAsync(h->gtwl.p, w.data(), sizeof(double) * w.size(), hipMemcpyHostToDevice, s));
    GSS_HIP(hipStreamSynchronize(s));
  } else if (h->ndim >= 2) {
    GSS_TRY(gen_upload_table(h->gtab[1], p2, s));
  }
  if (h->ndim == 3) GSS_TRY(gen_upload_table(h->gtab[2], p3, s));
  // Small grids do not fill the device (100 x 100: seven workgroups in GP1) and their three or five launches cost more
  // than their work: up to g_batch realisations share every launch (blockIdx.y), each with a half-spectrum buffer of its
  // own -- as many as fit 96 MiB (the buffers of a batch stay in the memory-side cache between the passes), at most 64.
  h->g_xbs = (int64_t)gg.nhp * gg.n2 * gg.n3;
  {
    static const int batch_env = env_int("GSS_FFTGS_GEN_BATCH", 0);
    int64_t b = ((int64_t)96 << 20) / (int64_t)(sizeof(double2) * (size_t)h->g_xbs);
    if (b > 64) b = 64;
    if (batch_env > 0) b = batch_env;
    if (b < 1) b = 1;
    h->g_batch = (int)b;
  }
  GSS_TRY(h->X.alloc(sizeof(double2) * (size_t)h->g_xbs * h->g_batch));
  GSS_TRY(dev_zero_bytes(h->X.p, h->X.bytes, s));   // the padding columns stay zero
  GSS_TRY(h->Fh_tiled.alloc(sizeof(double) * (size_t)gg.nhp * gg.n2 * gg.n3));   // amplitudes in the last pass's tile order
  const int lx = 104 * 1024;   // (one fixed bound for every handle: the attribute belongs to the function, not to the launch)
  GSS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gen_x_fwd_kernel<FF_SRC_PHILOX, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lx));
  GSS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gen_x_fwd_kernel<FF_SRC_ARRAY, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lx));
  GSS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gen_x_fwd_kernel<FF_SRC_COV, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lx));
  GSS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gen_x_inv_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lx));
  const int la = 96 * 1024;
#define GSS_GEN_ATTR(MODE, TXL) \
  GSS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gen_axis_kernel<MODE, TXL>), hipFuncAttributeMaxDynamicSharedMemorySize, la))
  GSS_GEN_ATTR(0, 3); GSS_GEN_ATTR(1, 3); GSS_GEN_ATTR(2, 3); GSS_GEN_ATTR(0, 2); GSS_GEN_ATTR(1, 2); GSS_GEN_ATTR(2, 2);
#undef GSS_GEN_ATTR
  GSS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gen_long_outer_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, la));
  GSS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gen_long_outer_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, la));
  GSS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gen_long_inner_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, la));
  GSS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gen_long_inner_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, la));
  h->g_long = lng;
  h->generic = true;
  return GSS_OK;
}

// (nb: realisations in this launch, grid y)
template <int SRC>
static void gen_launch_p1(gss_fftgs* h, uint64_t seed, uint32_t real, const double* noise, hipStream_t s, int nb = 1) {
  const GenGrid& g = h->gg;
  const int64_t nrows = (int64_t)g.n2 * g.n3;
  const unsigned gx = (unsigned)((nrows + h->g_rows - 1) / h->g_rows);
  if (h->g_tg)
    hipLaunchKernelGGL((gen_x_fwd_kernel<SRC, true>), dim3(gx, nb), dim3(GEN_XNT), gen_x_lds(h->gp[0], h->g_rows, true), s, g, h->gp[0],
                       h->g_rows, h->tw1.as<double2>(), h->gtab[0].as<double2>(), seed, real, noise, h->X.as<double2>(), h->vg,
                       h->g_xbs);
  else
    hipLaunchKernelGGL((gen_x_fwd_kernel<SRC, false>), dim3(gx, nb), dim3(GEN_XNT), gen_x_lds(h->gp[0], h->g_rows, false), s, g, h->gp[0],
                       h->g_rows, h->tw1.as<double2>(), h->gtab[0].as<double2>(), seed, real, noise, h->X.as<double2>(), h->vg,
                       h->g_xbs);
}
static void gen_launch_p5(gss_fftgs* h, double* z, hipStream_t s, int nb = 1) {
  const GenGrid& g = h->gg;
  const int64_t nrows = (int64_t)g.n2 * g.n3;
  const unsigned gx = (unsigned)((nrows + h->g_rows - 1) / h->g_rows);
  if (h->g_tg)
    hipLaunchKernelGGL(gen_x_inv_kernel<true>, dim3(gx, nb), dim3(GEN_XNT), gen_x_lds(h->gp[0], h->g_rows, true), s, g, h->gp[0], h->g_rows,
                       h->tw1.as<double2>(), h->gtab[0].as<double2>(), h->X.as<double2>(), z, h->g_xbs, h->N);
  else
    hipLaunchKernelGGL(gen_x_inv_kernel<false>, dim3(gx, nb), dim3(GEN_XNT), gen_x_lds(h->gp[0], h->g_rows, false), s, g, h->gp[0], h->g_rows,
                       h->tw1.as<double2>(), h->gtab[0].as<double2>(), h->X.as<double2>(), z, h->g_xbs, h->N);
}
// strided pass `MODE` along y (axis 1) or z (axis 2)
template <int MODE>
static void gen_launch_axis(gss_fftgs* h, int axis, hipStream_t s, int slab_t0 = 0, int slab_nt = 0, int nb = 1) {
  const GenGrid& g = h->gg;
  const GenPlan& pl = h->gp[axis];
  const int txlog = h->g_txlog[axis];
  const int nouter = axis == 1 ? g.n3 : g.n2;
  const int64_t ostride = axis == 1 ? (int64_t)g.n2 * g.nhp : (int64_t)g.nhp;
  const int64_t lstride = axis == 1 ? (int64_t)g.nhp : (int64_t)g.n2 * g.nhp;
  const unsigned blocks = (unsigned)(nouter * (slab_nt > 0 ? slab_nt : (g.nhp >> txlog)));
  const size_t lds = gen_axis_lds(pl, txlog);
  if (txlog == 3)
    hipLaunchKernelGGL((gen_axis_kernel<MODE, 3>), dim3(blocks, nb), dim3(GEN_ANT), lds, s, g, pl, axis, h->gtab[axis].as<double2>(),
                       ostride, lstride, h->X.as<double2>(), h->Fh_tiled.as<double>(), h->mean, slab_t0, slab_nt, h->g_xbs);
  else
    hipLaunchKernelGGL((gen_axis_kernel<MODE, 2>), dim3(blocks, nb), dim3(GEN_ANT), lds, s, g, pl, axis, h->gtab[axis].as<double2>(),
                       ostride, lstride, h->X.as<double2>(), h->Fh_tiled.as<double>(), h->mean, slab_t0, slab_nt, h->g_xbs);
}

// the y pass of a 2-D grid with long lines (fftgs_generic.h: outer / inner / outer); MODE 0: forward only, left in the
// order frequency c + L1 d at row L2 c + d
template <int MODE>
static void gen_launch_long(gss_fftgs* h, hipStream_t s, int nb = 1) {
  const GenGrid& g = h->gg;
  const GenLong& gl = h->gl;
  const unsigned ntx = (unsigned)(g.nhp >> 3);
  const size_t lo = sizeof(double2) * (size_t)(h->gpl[0].tlen + gl.L1 * gl.NB * 8);
  const size_t li = sizeof(double2) * (size_t)(h->gpl[1].tlen + gl.L2 * gl.NC * 8);
  hipLaunchKernelGGL(gen_long_outer_kernel<false>, dim3(ntx * (unsigned)(gl.L2 / gl.NB), nb), dim3(GEN_ANT), lo, s, g, h->gpl[0], gl,
                     h->gltab[0].as<double2>(), h->gtwl.as<double2>(), h->X.as<double2>(), h->g_xbs);
  hipLaunchKernelGGL(gen_long_inner_kernel<MODE>, dim3(ntx * (unsigned)(gl.L1 / gl.NC), nb), dim3(GEN_ANT), li, s, g, h->gpl[1], gl,
                     h->gltab[1].as<double2>(), h->X.as<double2>(), h->Fh_tiled.as<double>(), h->mean, h->g_xbs);
  if (MODE == 2)
    hipLaunchKernelGGL(gen_long_outer_kernel<true>, dim3(ntx * (unsigned)(gl.L2 / gl.NB), nb), dim3(GEN_ANT), lo, s, g, h->gpl[0], gl,
                       h->gltab[0].as<double2>(), h->gtwl.as<double2>(), h->X.as<double2>(), h->g_xbs);
}

// fft.jl:96-103 on the generic passes
static int32_t fftgs_spectrum_generic(gss_fftgs* h, double* partial, hipStream_t s) {
  gen_launch_p1<FF_SRC_COV>(h, 0, 0, nullptr, s);
  if (h->g_long) gen_launch_long<0>(h, s);
  else if (h->ndim >= 2) gen_launch_axis<0>(h, 1, s);
  if (h->ndim == 3) gen_launch_axis<0>(h, 2, s);
  hipLaunchKernelGGL(gen_amp_kernel, dim3(RED_BLOCKS), dim3(256), 0, s, h->gg, h->X.as<double2>(), h->Fh(), partial,
                     h->g_long ? h->gl.L1 : 0, h->g_long ? h->gl.L2 : 0);
  GSS_HIP(hipGetLastError());
  return GSS_OK;
}

// one realisation (fft.jl:163-170): five passes on 3-D grids, three on 2-D grids
// nb > 1: realisations real .. real + nb - 1 in the same launches (noise arrays and outputs N doubles apart)
static int32_t fftgs_generic_realize(gss_fftgs* h, uint64_t seed, int64_t real, const double* noise, double* z, hipStream_t s,
                                     int nb = 1) {
  ProfScope ps("fftgs_generic", s);
  if (noise) gen_launch_p1<FF_SRC_ARRAY>(h, seed, (uint32_t)real, noise, s, nb);
  else gen_launch_p1<FF_SRC_PHILOX>(h, seed, (uint32_t)real, nullptr, s, nb);
  if (h->ndim == 3) {
    // The slab order of the power-of-two pipeline (fftgs_fused_rest: the three strided passes slab by slab over the x
    // tiles, slabs alternating over the caller's and the helper streams) is available here as an A/B switch only:
    // measured (round 4, bit-identical fields) 500^3 2.96 ms with it against 2.79 without, 400^3 1.62 / 1.64 -- these
    // passes are not bound by memory alone (GP3 is two transforms and the phase step per trip), so what the cache
    // saves the extra launches give back.  GSS_FFTGS_GEN_SLAB=<tiles of the 512^3 measure per slab> switches it on.
    static const int slab = env_int("GSS_FFTGS_GEN_SLAB", 0);
    static const int slab_streams = env_int("GSS_FFTGS_SLAB_STREAMS", 3);
    const GenGrid& g = h->gg;
    const bool can_slab = slab > 0 && nb == 1 && h->g_txlog[1] == 3 && h->g_txlog[2] == 3 && h->X.bytes > ((size_t)200 << 20);
    if (can_slab) {
      const int ntx = g.nhp >> 3;
      // slabs of about the same bytes as two tiles of the 512^3 buffer (67 MB)
      int per = (int)(((size_t)64 << 20) / ((size_t)g.n2 * g.n3 * 128)) * slab / 2;
      if (per < 1) per = 1;
      int ns = slab_streams < 1 ? 1 : (slab_streams > gss_fftgs::SLAB_MAX_STREAMS ? gss_fftgs::SLAB_MAX_STREAMS : slab_streams);
      hipStream_t side[gss_fftgs::SLAB_MAX_STREAMS] = {nullptr, nullptr, nullptr, nullptr};
      for (int i = 1; i < ns; ++i) {
        side[i] = slab_stream(i);
        if (!side[i]) ns = i;
      }
      if (ns > 1 && !h->slab_e0) {
        GSS_HIP(hipEventCreateWithFlags(&h->slab_e0, hipEventDisableTiming));
        for (int i = 1; i < ns; ++i) GSS_HIP(hipEventCreateWithFlags(&h->slab_e[i], hipEventDisableTiming));
      }
      if (ns > 1) {
        GSS_HIP(hipEventRecord(h->slab_e0, s));
        for (int i = 1; i < ns; ++i) GSS_HIP(hipStreamWaitEvent(side[i], h->slab_e0, 0));
      }
      int islab = 0;
      for (int t0 = 0; t0 < ntx; t0 += per, ++islab) {
        const int nt = t0 + per <= ntx ? per : ntx - t0;
        hipStream_t st = (islab % ns) ? side[islab % ns] : s;
        gen_launch_axis<0>(h, 1, st, t0, nt);
        gen_launch_axis<2>(h, 2, st, t0, nt);
        gen_launch_axis<1>(h, 1, st, t0, nt);
      }
      for (int i = 1; i < ns; ++i) {
        GSS_HIP(hipEventRecord(h->slab_e[i], side[i]));
        GSS_HIP(hipStreamWaitEvent(s, h->slab_e[i], 0));
      }
    } else {
      gen_launch_axis<0>(h, 1, s, 0, 0, nb);
      gen_launch_axis<2>(h, 2, s, 0, 0, nb);
      gen_launch_axis<1>(h, 1, s, 0, 0, nb);
    }
  } else if (h->ndim == 1) {
    hipLaunchKernelGGL(gen_phase1d_kernel, dim3((unsigned)((h->gg.nh + 255) / 256), nb), dim3(256), 0, s, h->gg, h->X.as<double2>(),
                       h->Fh(), h->mean, h->g_xbs);
  } else if (h->g_long) {
    gen_launch_long<2>(h, s, nb);
  } else {
    gen_launch_axis<2>(h, 1, s, 0, 0, nb);
  }
  gen_launch_p5(h, z, s, nb);
  GSS_HIP(hipGetLastError());
  return GSS_OK;
}

// launches of the forward passes shared by the realisations and the spectrum build
template <int SRC>
static void launch_p1_src(gss_fftgs* h, uint64_t seed, uint32_t real, const double* noise, hipStream_t s) {
  const FusedGrid& f = h->fg;
  const int M = f.n1 / 2;
  const int64_t nrows = (int64_t)f.n2 * f.n3;
  double2* X = h->Xcur;
  const CovSrc* cs = h->covsrc.as<CovSrc>();
  const unsigned gx = (unsigned)((nrows + h->x_rows - 1) / h->x_rows);
  const size_t lds = ff_xfwd2_lds(M, f.l1 - 1, h->x_rows);
  if (f.l1 == 9)
    hipLaunchKernelGGL((ff_x_fwd2_kernel<SRC, 8, 256, 8>), dim3(gx), dim3(256), lds, s, f, h->tw1.as<double2>(),
                       h->xtw.as<double2>(), seed, real, noise, X, cs);
  else if (h->x_rows == 8)
    hipLaunchKernelGGL((ff_x_fwd2_kernel<SRC, 8, 256, -1>), dim3(gx), dim3(256), lds, s, f, h->tw1.as<double2>(),
                       h->xtw.as<double2>(), seed, real, noise, X, cs);
  else
    hipLaunchKernelGGL((ff_x_fwd2_kernel<SRC, 4, 256, -1>), dim3(gx), dim3(256), lds, s, f, h->tw1.as<double2>(),
                       h->xtw.as<double2>(), seed, real, noise, X, cs);
}

// launches of the forward passes shared by the realisations and the spectrum build
static void launch_p1(gss_fftgs* h, int src, uint64_t seed, uint32_t real, const double* noise, hipStream_t s) {
  if (src == FF_SRC_COV) launch_p1_src<FF_SRC_COV>(h, seed, real, noise, s);
  else if (src == FF_SRC_ARRAY) launch_p1_src<FF_SRC_ARRAY>(h, seed, real, noise, s);
  else launch_p1_src<FF_SRC_PHILOX>(h, seed, real, noise, s);
}

static void launch_p5(gss_fftgs* h, double* z, hipStream_t s) {
  const FusedGrid& f = h->fg;
  const int M = f.n1 / 2;
  const int64_t nrows = (int64_t)f.n2 * f.n3;
  const double2* X = h->Xcur;
  const unsigned gx = (unsigned)((nrows + h->x_rows - 1) / h->x_rows);
  const size_t lds = ff_xinv2_lds(M, f.l1 - 1, h->x_rows);
  if (f.l1 == 9)
    hipLaunchKernelGGL((ff_x_inv2_kernel<8, 256, 8>), dim3(gx), dim3(256), lds, s, f, h->tw1.as<double2>(),
                       h->xtw.as<double2>(), X, z);
  else if (h->x_rows == 8)
    hipLaunchKernelGGL((ff_x_inv2_kernel<8, 256, -1>), dim3(gx), dim3(256), lds, s, f, h->tw1.as<double2>(),
                       h->xtw.as<double2>(), X, z);
  else
    hipLaunchKernelGGL((ff_x_inv2_kernel<4, 256, -1>), dim3(gx), dim3(256), lds, s, f, h->tw1.as<double2>(),
                       h->xtw.as<double2>(), X, z);
}

static void launch_p2(gss_fftgs* h, hipStream_t s) { (void)launch_axis_mode<0>(h, 1, s); }

// fft.jl:96-103 on the fused passes: covariance rows are produced inside P1 (no N-sized input array), then the y and
// z forward passes; the amplitude kernel undoes the bit reversal while it writes the natural-order state.
static int32_t fftgs_spectrum_fused(gss_fftgs* h, double* partial, hipStream_t s) {
  const FusedGrid& f = h->fg;
  CovSrc cs;
  cs.vg = h->vg;
  cs.c1 = (int)h->g.c1; cs.c2 = (int)h->g.c2; cs.c3 = (int)h->g.c3;
  cs.s1 = h->g.s1; cs.s2 = h->g.s2; cs.s3 = h->g.s3;
  GSS_TRY(h->covsrc.alloc(sizeof(CovSrc)));
  GSS_HIP(hipMemcpyAsync(h->covsrc.p, &cs, sizeof(CovSrc), hipMemcpyHostToDevice, s));
  GSS_HIP(hipStreamSynchronize(s));  // `cs` is a stack object
  launch_p1(h, FF_SRC_COV, 0, 0, nullptr, s);
  launch_p2(h, s);
  GSS_TRY(launch_axis_mode<0>(h, 2, s));
  hipLaunchKernelGGL(ff_amp_kernel, dim3(RED_BLOCKS), dim3(256), 0, s, f, h->Xcur, h->Fh(), partial);
  GSS_HIP(hipGetLastError());
  return GSS_OK;
}

// the state is complete (computed or adopted): derive what the realisation kernels read
static int32_t fftgs_finish_state(gss_fftgs* h, hipStream_t s) {
  double hs[2];
  GSS_HIP(hipMemcpyAsync(hs, h->scal(), sizeof(hs), hipMemcpyDeviceToHost, s));
  GSS_HIP(hipStreamSynchronize(s));
  GSS_REQUIRE(hs[0] > 0.0 && std::isfinite(hs[1]) && hs[1] > 0.0, "degenerate spectrum (sum F^2 = %g)", hs[0]);
  if (h->fused) {
    const FusedGrid& f = h->fg;
    const int64_t nt = (int64_t)f.n2 * f.ntx * f.n3 * FF_TX;
    if (h->txz_log == 3)
      hipLaunchKernelGGL(ff_tile_fh2_kernel<3>, dim3(grid_blocks(nt)), dim3(256), 0, s, f, h->Fh(), h->Fh_tiled.as<double>());
    else
      hipLaunchKernelGGL(ff_tile_fh2_kernel<2>, dim3(grid_blocks(nt)), dim3(256), 0, s, f, h->Fh(), h->Fh_tiled.as<double>());
    GSS_HIP(hipGetLastError());
  }
  if (h->generic && h->ndim >= 2) {
    const int axis = h->ndim == 3 ? 2 : 1;
    const int64_t nt = (int64_t)h->gg.nhp * h->gg.n2 * h->gg.n3;
    if (h->g_long)
      hipLaunchKernelGGL(gen_tile_fh_long_kernel, dim3(grid_blocks(nt)), dim3(256), 0, s, h->gg, h->gl, h->Fh(),
                         h->Fh_tiled.as<double>());
    else if (h->g_txlog[axis] == 3)
      hipLaunchKernelGGL(gen_tile_fh_kernel<3>, dim3(grid_blocks(nt)), dim3(256), 0, s, h->gg, axis, h->gp[axis].L, h->Fh(),
                         h->Fh_tiled.as<double>());
    else
      hipLaunchKernelGGL(gen_tile_fh_kernel<2>, dim3(grid_blocks(nt)), dim3(256), 0, s, h->gg, axis, h->gp[axis].L, h->Fh(),
                         h->Fh_tiled.as<double>());
    GSS_HIP(hipGetLastError());
  }
  h->ready = true;
  return GSS_OK;
}

// one realisation through the fused pipeline; `noise` (N uniforms) may be NULL; z receives N doubles
static int32_t fftgs_fused_p1(gss_fftgs* h, uint64_t seed, int64_t real, const double* noise, hipStream_t s) {
  ProfScope ps("fftgs_p1", s);
  launch_p1(h, noise ? FF_SRC_ARRAY : FF_SRC_PHILOX, seed, (uint32_t)real, noise, s);
  GSS_HIP(hipGetLastError());
  return GSS_OK;
}

// helper streams of the slab order, shared by every handle of the process (creating a stream costs milliseconds);
// what runs on them is fenced by the calling handle's events on both sides
static hipStream_t slab_stream(int i) { return helper_stream(HELPER_GEN0 + (i - 1) % 3); }   // i = 1, 2, ...

static int32_t fftgs_fused_rest(gss_fftgs* h, double* z, hipStream_t s) {
  // The three strided passes run slab by slab over the x tiles: P2, P3, P4 on the tiles of slab 0, then slab 1, ...
  // A slab (GSS_FFTGS_SLAB tiles; 2 tiles = 67 MB of the 1.1 GB half spectrum at 512^3) written by one pass is read
  // back by the next from the 256 MB memory-side cache; consecutive slabs go to GSS_FFTGS_SLAB_STREAMS streams in
  // turn (the caller's and helper streams fenced by events), so that the short launches of a slab -- two rounds of
  // resident workgroups each -- overlap with those of its neighbours instead of draining the device 45 times per
  // realisation.  Measured per 512^3 realisation: one launch per pass 2.23 ms; slabs of 7 tiles on one stream 2.09
  // (1 tile: 2.52, the launches are too small); slabs of 2 tiles on 3 streams 1.93 (2 streams 1.93-1.97, 4 streams 1.96;
  // 1 tile on 4 streams 1.95, 3 tiles on 3 streams 2.02).  GSS_FFTGS_SLAB=0: whole buffer per pass.
  static const int slab = env_int("GSS_FFTGS_SLAB", 2);
  static const int slab_streams = env_int("GSS_FFTGS_SLAB_STREAMS", 3);
  const FusedGrid& f = h->fg;
  const bool can_slab = slab > 0 && h->txy_log == 3 && h->txz_log == 3 && f.l2 == 9 && f.l3 == 9;
  if (can_slab) {
    ProfScope ps("fftgs_p234", s);
    const int ntx = f.nhp >> 3;
    int ns = slab_streams < 1 ? 1 : (slab_streams > gss_fftgs::SLAB_MAX_STREAMS ? gss_fftgs::SLAB_MAX_STREAMS : slab_streams);
    hipStream_t side[gss_fftgs::SLAB_MAX_STREAMS] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 1; i < ns; ++i) {
      side[i] = slab_stream(i);
      if (!side[i]) ns = i;   // no more helper streams to be had: the slabs share what there is
    }
    if (ns > 1 && !h->slab_e0) {
      GSS_HIP(hipEventCreateWithFlags(&h->slab_e0, hipEventDisableTiming));
      for (int i = 1; i < ns; ++i) GSS_HIP(hipEventCreateWithFlags(&h->slab_e[i], hipEventDisableTiming));
    }
    if (ns > 1) {
      GSS_HIP(hipEventRecord(h->slab_e0, s));   // P1 (and whatever else the caller's stream holds) comes first
      for (int i = 1; i < ns; ++i) GSS_HIP(hipStreamWaitEvent(side[i], h->slab_e0, 0));
    }
    int islab = 0;
    for (int t0 = 0; t0 < ntx; t0 += slab, ++islab) {
      const int nt = t0 + slab <= ntx ? slab : ntx - t0;
      hipStream_t st = (islab % ns) ? side[islab % ns] : s;
      GSS_TRY(launch_axis_mode<0>(h, 1, st, t0, nt));
      GSS_TRY(launch_axis_mode<2>(h, 2, st, t0, nt));
      GSS_TRY(launch_axis_mode<1>(h, 1, st, t0, nt));
    }
    for (int i = 1; i < ns; ++i) {              // P5 waits for every slab
      GSS_HIP(hipEventRecord(h->slab_e[i], side[i]));
      GSS_HIP(hipStreamWaitEvent(s, h->slab_e[i], 0));
    }
  } else {
  {
    ProfScope ps("fftgs_p2", s);
    launch_p2(h, s);
  }
  {
    ProfScope ps("fftgs_p3", s);
    GSS_TRY(launch_axis_mode<2>(h, 2, s));
  }
  {
    ProfScope ps("fftgs_p4", s);
    GSS_TRY(launch_axis_mode<1>(h, 1, s));
  }
  }
  {
    ProfScope ps("fftgs_p5", s);
    launch_p5(h, z, s);
  }
  GSS_HIP(hipGetLastError());
  return GSS_OK;
}

// helper stream, events and the second half-spectrum buffer of the realisation pipeline (first multi-realisation call)
static int32_t fftgs_pipeline_setup(gss_fftgs* h, hipStream_t s) {
  if (h->s2) return GSS_OK;
  const FusedGrid& f = h->fg;
  GSS_TRY(h->X2.alloc(sizeof(double2) * (size_t)f.nhp * f.n2 * f.n3));
  GSS_TRY(dev_zero_bytes(h->X2.p, h->X2.bytes, s));   // padding columns stay zero, as in X
  GSS_HIP(hipStreamCreateWithFlags(&h->s2, hipStreamNonBlocking));
  for (int b = 0; b < 2; ++b) {
    GSS_HIP(hipEventCreateWithFlags(&h->ev_p1[b], hipEventDisableTiming));
    GSS_HIP(hipEventCreateWithFlags(&h->ev_p5[b], hipEventDisableTiming));
  }
  GSS_HIP(hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming));
  return GSS_OK;
}

extern "C" {

int32_t gss_fftgs_create(gss_fftgs_t** out, const gss_variogram_t* vg, int32_t ndim, const int64_t* dims,
                         const double* spacing, double mean, int32_t flags, void* stream) {
  GSS_ENTRY();
  GSS_REQUIRE(out != nullptr, "gss_fftgs_create: out is NULL");
  *out = nullptr;
  GSS_REQUIRE(ndim >= 1 && ndim <= 3 && dims != nullptr, "FFTGS needs a 1-D, 2-D or 3-D Cartesian grid");
  gss_variogram_t v3 = *vg;
  GSS_REQUIRE(vg->dim == ndim, "variogram dimension %d does not match the grid dimension %d", vg->dim, ndim);
  v3.dim = 3;  // lags of absent axes are zero
  for (int k = ndim; k < 3; ++k) {
    v3.inv_radii[k] = 1.0;
    for (int e = 0; e < 3; ++e) v3.extra[e].inv_radii[k] = 1.0;
  }
  gss_fftgs* h = new (std::nothrow) gss_fftgs();
  if (!h) return GSS_ERR_ALLOC;
  struct Guard {
    gss_fftgs* h;
    ~Guard() { delete h; }
  } guard{h};
  GSS_REQUIRE(vg_is_stationary(&v3), "variogram model must be stationary");  // fft.jl:91, lu.jl:110
  GSS_TRY(make_vgdev(&v3, &h->vg));
  int64_t d[3] = {1, 1, 1};
  double sp[3] = {1.0, 1.0, 1.0};
  for (int k = 0; k < ndim; ++k) {
    GSS_REQUIRE(dims[k] >= 2, "every grid axis needs at least two cells");
    d[k] = dims[k];
    sp[k] = spacing ? spacing[k] : 1.0;
  }
  h->ndim = ndim;
  h->mean = mean;
  h->g = GridSpec{d[0], d[1], d[2], d[0] / 2 + 1, d[0] / 2 - 1, d[1] / 2 - 1, d[2] / 2 - 1, sp[0], sp[1], sp[2]};
  // axes beyond ndim have size 1: their centre offset must be zero
  if (ndim < 2) h->g.c2 = 0;
  if (ndim < 3) h->g.c3 = 0;
  h->N = d[0] * d[1] * d[2];
  h->NH = h->g.nh * d[1] * d[2];
  GSS_REQUIRE(h->N >= 2, "FFTGS needs at least two cells");
  hipStream_t s = to_stream(stream);

  GSS_TRY(h->state.alloc(sizeof(double) * (size_t)(h->NH + 2)));
  GSS_TRY(fftgs_setup_fused(h, s));  // decides the pipeline; allocates its buffers (no rocFFT plan on that path)
  GSS_TRY(fftgs_setup_generic(h, s));
  if (flags & GSS_FFTGS_NO_SPECTRUM) {  // the state arrives by broadcast (gss_fftgs_adopt_state)
    GSS_HIP(hipStreamSynchronize(s));
    guard.h = nullptr;
    *out = h;
    return GSS_OK;
  }

  // spectrum: C -> forward transform -> sqrt|.| -> Parseval scale
  DevBuf partial;
  GSS_TRY(partial.alloc(sizeof(double) * RED_BLOCKS));
  if (h->fused) {
    GSS_TRY(fftgs_spectrum_fused(h, partial.as<double>(), s));
  } else if (h->generic) {
    GSS_TRY(fftgs_spectrum_generic(h, partial.as<double>(), s));
  } else {
    GSS_TRY(ensure_rocfft(h));
    hipLaunchKernelGGL(fftgs_cov_kernel, dim3(grid_blocks(h->N)), dim3(256), 0, s, h->vg, h->g, h->U.as<double>());
    GSS_HIP(hipGetLastError());
    GSS_TRY(fft_exec(h, h->fwd, h->U.p, h->Xn.p, s));
    hipLaunchKernelGGL(fftgs_amp_kernel, dim3(RED_BLOCKS), dim3(256), 0, s, h->g, h->Xn.as<double2>(), h->Fh(),
                       partial.as<double>());
  }
  hipLaunchKernelGGL(fftgs_scale_kernel, dim3(1), dim3(64), 0, s, partial.as<double>(), RED_BLOCKS, h->vg.sill,
                     (double)h->N, h->scal());
  hipLaunchKernelGGL(fftgs_apply_scale_kernel, dim3(grid_blocks(h->NH)), dim3(256), 0, s, h->Fh(), h->NH,
                     h->scal());
  GSS_HIP(hipGetLastError());
  GSS_TRY(fftgs_finish_state(h, s));
  guard.h = nullptr;
  *out = h;
  return GSS_OK;
}

int32_t gss_fftgs_destroy(gss_fftgs_t* h) {
  GSS_ENTRY();
  delete h;
  return GSS_OK;
}

int32_t gss_fftgs_spectrum(gss_fftgs_t* h, double* f_out, int32_t mem, void* stream) {
  GSS_ENTRY();
  GSS_REQUIRE(h != nullptr && f_out != nullptr, "gss_fftgs_spectrum: NULL argument");
  GSS_REQUIRE(h->ready, "handle has no spectrum");
  hipStream_t s = to_stream(stream);
  Staged so;
  GSS_TRY(so.out(f_out, sizeof(double) * (size_t)h->N, mem));
  hipLaunchKernelGGL(fftgs_expand_kernel, dim3(grid_blocks(h->N)), dim3(256), 0, s, h->g, h->Fh(), h->scal(),
                     so.as<double>());
  GSS_HIP(hipGetLastError());
  return so.back(f_out, sizeof(double) * (size_t)h->N, mem, s);
}

int32_t gss_fftgs_state_buffer(gss_fftgs_t* h, void** dev_ptr, int64_t* bytes) {
  GSS_ENTRY();
  GSS_REQUIRE(h != nullptr && dev_ptr != nullptr && bytes != nullptr, "NULL argument");
  *dev_ptr = h->state.p;
  *bytes = (int64_t)(sizeof(double) * (size_t)(h->NH + 2));
  return GSS_OK;
}

int32_t gss_fftgs_adopt_state(gss_fftgs_t* h, void* stream) {
  GSS_ENTRY();
  GSS_REQUIRE(h != nullptr, "NULL handle");
  return fftgs_finish_state(h, to_stream(stream));
}

int32_t gss_fftgs_realize(gss_fftgs_t* h, uint64_t seed, int64_t first_real, int64_t nreals, const double* noise,
                          const int64_t* inds, int64_t ninds, double* out, int32_t mem, void* stream) {
  GSS_ENTRY();
  GSS_REQUIRE(h != nullptr && out != nullptr && nreals >= 0 && first_real >= 0, "gss_fftgs_realize: bad arguments");
  GSS_REQUIRE(h->ready, "handle has no spectrum");
  if (nreals == 0) return GSS_OK;
  hipStream_t s = to_stream(stream);
  int rbatch = 0;
  if (!h->fused && !h->generic) {
    GSS_TRY(ensure_rocfft(h));
    GSS_TRY(ensure_rocfft_batch(h, nreals, &rbatch));
  }
  const int64_t N = h->N;
  const int64_t npts = inds ? ninds : N;
  // Host arrays: the realisations leave chunk by chunk through a ring of at most three chunks (OutStream: the
  // reference returns every realisation as a host vector, fft.jl:173,197 -- 256 realisations of 512^3 cells are 256 GiB
  // and never sit in HBM together); supplied noise arrives one realisation at a time on the caller's stream.
  Staged sn, si;
  OutStream os;
  const bool host = mem == GSS_MEM_HOST;
  if (noise && host) {
    GSS_TRY(sn.own.alloc(sizeof(double) * (size_t)N));
    sn.p = sn.own.p;
  } else {
    sn.p = const_cast<double*>(noise);
  }
  GSS_TRY(si.in(inds, sizeof(int64_t) * (size_t)(inds ? ninds : 0), mem, s));
  GSS_TRY(os.begin(out, sizeof(double) * (size_t)npts, nreals, mem, s));
  if (inds && h->Z.bytes < sizeof(double) * (size_t)N) GSS_TRY(h->Z.alloc(sizeof(double) * (size_t)N));

  // fused pipeline with more than one realisation: P1 of realisation r+1 on the helper stream beside P2..P5 of r
  const bool piped = h->fused && h->overlap && nreals > 1 && !(noise && host);
  if (piped) {
    GSS_TRY(fftgs_pipeline_setup(h, s));
    GSS_HIP(hipEventRecord(h->ev_in, s));               // everything queued so far (inputs, earlier calls) ...
    GSS_HIP(hipStreamWaitEvent(h->s2, h->ev_in, 0));    // ... precedes the helper stream's first kernel
  }
  for (int64_t r = 0; r < nreals; ++r) {
    double* dst = nullptr;                              // realisation r's place (caller's HBM or a slot of the ring)
    GSS_TRY(os.slot(r, s, &dst));
    const double* nz = nullptr;
    if (noise) {
      nz = host ? sn.as<double>() : noise + r * N;
      if (host) GSS_HIP(hipMemcpyAsync(sn.p, noise + r * N, sizeof(double) * (size_t)N, hipMemcpyHostToDevice, s));
    }
    if (h->fused) {
      double* zf = inds ? h->Z.as<double>() : dst;
      if (piped) {
        const int b = (int)(r & 1);
        h->Xcur = b ? h->X2.as<double2>() : h->X.as<double2>();
        if (r >= 2) GSS_HIP(hipStreamWaitEvent(h->s2, h->ev_p5[b], 0));   // buffer b is free again
        GSS_TRY(fftgs_fused_p1(h, seed, first_real + r, nz, h->s2));
        GSS_HIP(hipEventRecord(h->ev_p1[b], h->s2));
        GSS_HIP(hipStreamWaitEvent(s, h->ev_p1[b], 0));
        GSS_TRY(fftgs_fused_rest(h, zf, s));
        GSS_HIP(hipEventRecord(h->ev_p5[b], s));
      } else {
        h->Xcur = h->X.as<double2>();
        GSS_TRY(fftgs_fused_p1(h, seed, first_real + r, nz, s));
        GSS_TRY(fftgs_fused_rest(h, zf, s));
      }
      if (inds) {
        hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((ninds + 255) / 256)), dim3(256), 0, s, zf, si.as<int64_t>(),
                           ninds, dst);
        GSS_HIP(hipGetLastError());
      }
      GSS_TRY(os.done(r, s));
      continue;
    }
    if (h->generic) {
      // a batch: consecutive realisations whose outputs are consecutive (the caller's array, or one chunk of the ring)
      int64_t nb = h->g_batch;
      if (noise && host) nb = 1;                           // (host noise is staged one realisation at a time)
      if (nb > nreals - r) nb = nreals - r;
      if (os.on && nb > os.chunk - r % os.chunk) nb = os.chunk - r % os.chunk;
      if (inds && nb > 1 && h->Z.bytes < sizeof(double) * (size_t)N * (size_t)nb)
        GSS_TRY(h->Z.alloc(sizeof(double) * (size_t)N * (size_t)h->g_batch));
      double* zf = inds ? h->Z.as<double>() : dst;
      GSS_TRY(fftgs_generic_realize(h, seed, first_real + r, nz, zf, s, (int)nb));
      if (inds) {
        hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((ninds + 255) / 256), (unsigned)nb), dim3(256), 0, s, zf,
                           si.as<int64_t>(), ninds, dst, N);
        GSS_HIP(hipGetLastError());
      }
      GSS_TRY(os.done(r + nb - 1, s));
      r += nb - 1;
      continue;
    }
    // rocFFT pipeline; `rbatch` realisations per execution while that many are left (and lie in one chunk of the ring)
    int64_t nb = 1;
    if (rbatch > 1 && !(noise && host) && nreals - r >= rbatch && (!os.on || os.chunk - r % os.chunk >= rbatch)) nb = rbatch;
    if (inds && nb > 1 && h->Z.bytes < sizeof(double) * (size_t)N * (size_t)nb) GSS_TRY(h->Z.alloc(sizeof(double) * (size_t)N * (size_t)nb));
    double* u = h->U.as<double>();
    if (noise) {
      u = const_cast<double*>(nz);  // the forward transform does not overwrite its input
    } else {
      ProfScope ps("fftgs_noise", s);
      GSS_TRY(philox_uniform_dev(seed, first_real + r, N, u, N, N, s, (int)nb, N));
    }
    {
      ProfScope ps("fftgs_fwd", s);
      GSS_TRY(fft_exec(h, nb > 1 ? h->plans->fwdB : h->fwd, u, h->Xn.p, s));
    }
    {
      ProfScope ps("fftgs_phase", s);
      hipLaunchKernelGGL(fftgs_phase_kernel, dim3(grid_blocks(h->NH), (unsigned)nb), dim3(256), 0, s, h->Xn.as<double2>(),
                         h->Fh(), h->NH, h->mean);
      GSS_HIP(hipGetLastError());
    }
    double* z = inds ? h->Z.as<double>() : dst;
    {
      ProfScope ps("fftgs_inv", s);
      GSS_TRY(fft_exec(h, nb > 1 ? h->plans->invB : h->inv, h->Xn.p, z, s));
    }
    if (inds) {
      hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((ninds + 255) / 256), (unsigned)nb), dim3(256), 0, s, z,
                         si.as<int64_t>(), ninds, dst, N);
      GSS_HIP(hipGetLastError());
    }
    GSS_TRY(os.done(r + nb - 1, s));
    r += nb - 1;
  }
  return os.finish(s);
}

}  // extern "C"
