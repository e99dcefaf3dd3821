// Internal declarations shared by the translation units of libgss_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include <mutex>

#include "gss.h"

namespace gss {

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);

#define GSS_HIP(call)                                                                         \
  do {                                                                                        \
    hipError_t e__ = (call);                                                                  \
    if (e__ != hipSuccess) {                                                                  \
      ::gss::set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
      return GSS_ERR_HIP;                                                                     \
    }                                                                                         \
  } while (0)

#define GSS_TRY(call)                 \
  do {                                \
    int32_t s__ = (call);             \
    if (s__ != GSS_OK) return s__;    \
  } while (0)

#define GSS_REQUIRE(cond, ...)          \
  do {                                  \
    if (!(cond)) {                      \
      ::gss::set_error(__VA_ARGS__);    \
      return GSS_ERR_INVALID;           \
    }                                   \
  } while (0)

// Every exported function takes the library's lock for its whole duration (GSS_ENTRY, first statement): calls from
// several host threads are serialised, so the stream chain (to_stream), the block cache and the process-wide work
// buffers see one call at a time whichever handles and streams the threads use.  Recursive: exports call each other.
std::recursive_mutex& api_mutex();
// ... and, when the outermost export returns, records the library's chain event on the stream the call used (the
// stream is certainly alive then); the next call on a DIFFERENT stream waits for that event (to_stream), so the library
// never touches a stream of an earlier call again -- its owner may have destroyed it.
struct EntryGuard {
  std::lock_guard<std::recursive_mutex> lock;
  EntryGuard();
  ~EntryGuard();
};
#define GSS_ENTRY() ::gss::EntryGuard gss_entry_guard__

// ---------------------------------------------------------------------------------------------
// optional per-kernel timing with HIP events (gss_profile_*)
// ---------------------------------------------------------------------------------------------
bool prof_enabled();
void prof_begin(const char* name, hipStream_t s);
void prof_end(const char* name, hipStream_t s);
struct ProfScope {
  const char* name;
  hipStream_t s;
  bool on;
  ProfScope(const char* n, hipStream_t st) : name(n), s(st), on(prof_enabled()) {
    if (on) prof_begin(name, s);
  }
  ~ProfScope() {
    if (on) prof_end(name, s);
  }
};

// ---------------------------------------------------------------------------------------------
// device buffers
// ---------------------------------------------------------------------------------------------
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;   // what was asked for
  size_t cap = 0;     // size of the block behind it (a cached block may be up to a quarter larger)
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  int32_t alloc(size_t nbytes);
  void release();
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// A pointer that is either borrowed (caller's device memory) or a staged copy of host memory.
struct Staged {
  DevBuf own;
  void* p = nullptr;
  // input array: copy host->device when mem == HOST
  int32_t in(const void* src, size_t bytes, int32_t mem, hipStream_t s);
  // output array: allocate device scratch when mem == HOST
  int32_t out(void* dst, size_t bytes, int32_t mem);
  // copy back to host destination (no-op for DEVICE)
  int32_t back(void* dst, size_t bytes, int32_t mem, hipStream_t s);
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// Host arrays of a large call, handed over in pieces: the copy of piece i + 1 (host -> device, copy stream 0 of the
// process) and the results of piece i - 1 (device -> host, copy stream 1) cross the bus while piece i is computed on
// the caller's stream.  The device images of the arrays are whole-call scratch (Staged::out); `stride` = bytes per
// point.  A copy to or from pageable memory holds the calling thread until it is done, which is why the results of
// a piece only leave once the NEXT piece has been queued.  Usage: begin(); add_in / add_out; per piece fetch(),
// launches, deliver(); finish().  `on` is false (and every method a no-op) for device arrays, short calls,
// GSS_HOST_PIPELINE=0, or when the copy streams cannot be had: the caller then copies in one go as before.
struct HostPipe {
  static constexpr int64_t PIECE = 131072;
  struct Arr {
    const char* host_in;
    char* host_out;
    char* dev;
    size_t stride;
  };
  Arr ins[4], outs[6];
  int nin = 0, nout = 0;
  hipStream_t cin = nullptr, cout = nullptr;
  hipEvent_t ev_in = nullptr, ev_done = nullptr;
  int64_t pend_off = 0, pend_n = 0;
  bool on = false;
  HostPipe() = default;
  HostPipe(const HostPipe&) = delete;
  HostPipe& operator=(const HostPipe&) = delete;
  ~HostPipe();
  int32_t begin(int32_t mem, int64_t m, hipStream_t s);
  void add_in(const void* host, void* dev, size_t stride);
  void add_out(void* host, void* dev, size_t stride);   // NULL host or dev: skipped
  int32_t fetch(int64_t off, int64_t n, hipStream_t s);
  int32_t deliver(int64_t off, int64_t n, hipStream_t s);
  int32_t finish(hipStream_t s);
};

// Per-realisation outputs of the simulation calls (gss_fftgs_realize, gss_lugs_realize, gss_sgs_realize).  The
// reference hands every realisation back as a host vector (fft.jl:173,197; lu.jl:217-221); with a host destination the
// device therefore stages at most DEPTH chunks of realisations (a chunk: as many realisations as fit ~256 MiB, at least
// one -- GSS_OUT_CHUNK_MB) in a ring, and chunk c crosses the bus on a copy stream of the process while chunk c + 1 is
// computed on the caller's stream.  A pinned destination (hipHostMalloc / hipHostRegister, e.g. a torch pinned tensor)
// is written by the DMA engine directly; a pageable one goes through two pinned bounce buffers of the process (pieces of
// 32 MiB alternating over two copy streams, each followed in stream order by a host function that copies the piece to
// its destination with a few threads: the transfer of piece p + 1 overlaps the host copy of piece p).  With a device
// destination every method is a no-op around `dst + r * real_bytes`.  Usage: begin(); per realisation r (ascending):
// slot(r) -> kernels on s -> done(r); finish().  The destructor joins the copy streams (error paths).
struct OutStream {
  static constexpr int DEPTH = 3;
  bool on = false, issued = false;
  char* dst = nullptr;          // destination (host when on, device otherwise)
  size_t real_bytes = 0;
  int64_t nreals = 0, chunk = 1;
  bool pinned_dst = false;
  DevBuf ring[DEPTH];
  hipEvent_t ev_done[DEPTH] = {nullptr, nullptr, nullptr}, ev_free[DEPTH] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_join = nullptr;
  hipStream_t cs[2] = {nullptr, nullptr};
  int64_t piece_no = 0;
  OutStream() = default;
  OutStream(const OutStream&) = delete;
  OutStream& operator=(const OutStream&) = delete;
  ~OutStream();
  // chunk_reals > 0 fixes the realisations per chunk (callers that compute block-wise and feed several outputs)
  int32_t begin(void* dst_, size_t real_bytes_, int64_t nreals_, int32_t mem, hipStream_t s, int64_t chunk_reals = 0);
  static int64_t default_chunk(size_t real_bytes_, int64_t nreals_);
  // device address of realisation r's output; the first realisation of a chunk makes s wait until the slot's previous
  // contents have left
  int32_t slot(int64_t r, hipStream_t s, double** out);
  int32_t done(int64_t r, hipStream_t s);   // realisation r has been queued on s
  int32_t finish(hipStream_t s);            // host destination: returns when every byte has arrived
  size_t staged_bytes() const;              // HBM held by the ring
};

// Every C-ABI entry converts its `stream` argument here.  Scratch memory (the DevBuf pool, the kriging workspace) is
// recycled without per-block events, which is only safe if everything the library queues is ordered; when a call
// arrives on a different stream than the previous one, the new stream is made to wait for the event the previous call
// left behind at its exit (EntryGuard above): one stream wait, nothing when the process keeps to one stream.
hipStream_t to_stream(void* s);
inline int64_t round_up(int64_t x, int64_t q) { return (x + q - 1) / q * q; }

// ---------------------------------------------------------------------------------------------
// variogram on device
// ---------------------------------------------------------------------------------------------
struct VgExtra {
  int kind;
  int aniso;
  double cs;         // contribution (partial sill) of this structure
  double inv_range;
  double mscale;
  double pw;
  double ir[3];
};

struct VgDev {
  int kind;      // GSS_VG_* ; MATERN split into 30..35 for nu = 1/2, 3/2, 5/2, 1, 2, 3
  int dim;
  int aniso;
  int nextra;        // additional nested structures
  double sill;       // total sill (all structures + nugget)
  double cs;         // contribution of the first structure
  double inv_range;  // 1 / range (1 when aniso)
  double mscale;     // Matern: sqrt(2 nu) * 3; power: scaling / cs
  double pw;         // power: exponent / 2 (applied to the squared distance); VG_MATERN_NU: the order nu
  double ir[3];      // inverse radii (aniso) or 1
  VgExtra ex[3];
};
enum { VG_MATERN12 = 30, VG_MATERN32 = 31, VG_MATERN52 = 32, VG_MATERN1 = 33, VG_MATERN2 = 34, VG_MATERN3 = 35,
       VG_MATERN_NU = 36 };  // any other positive order: pw carries nu

int32_t make_vgdev(const gss_variogram_t* vg, VgDev* out);
// fft.jl:91, lu.jl:110: the simulation solvers need a finite sill
inline bool vg_is_stationary(const gss_variogram_t* vg) { return vg->kind != GSS_VG_POWER; }

// squared (possibly Mahalanobis) distance, dimension order, no FMA contraction (kNN tie contract)
template <int DIM>
__device__ __forceinline__ double sqdist_nofma(const double* a, const double* b, const double* ir, bool aniso) {
#pragma clang fp contract(off)
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < DIM; ++k) {
    double t = a[k] - b[k];
    if (aniso) t = t * ir[k];
    double tt = t * t;
    acc = acc + tt;
  }
  return acc;
}

// Ranking key of the neighbour search for the `distance` solver parameter (krig.jl:72, idw.jl:54, lwr.jl:57):
// monotone in the distance, accumulated in dimension order with one rounding per operation.
//   EUCLIDEAN: squared (Mahalanobis) distance; CITYBLOCK: sum |t|; CHEBYSHEV: max |t|;
//   HAVERSINE: sin^2(dlat/2) + cos(lat1) cos(lat2) sin^2(dlon/2) for (lon, lat) in degrees ([DEP] Distances.jl)
template <int DIM, int METRIC>
__device__ __forceinline__ double metric_key(const double* a, const double* b, const double* ir, bool aniso) {
#pragma clang fp contract(off)
  if (METRIC == GSS_METRIC_CITYBLOCK) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < DIM; ++k) acc = acc + fabs(a[k] - b[k]);
    return acc;
  } else if (METRIC == GSS_METRIC_CHEBYSHEV) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
      const double t = fabs(a[k] - b[k]);
      acc = t > acc ? t : acc;
    }
    return acc;
  } else if (METRIC == GSS_METRIC_HAVERSINE) {
    const double D = 0.017453292519943295;  // pi / 180
    const double s1 = sin(((b[DIM > 1 ? 1 : 0] - a[DIM > 1 ? 1 : 0]) * 0.5) * D);
    const double s2 = sin(((b[0] - a[0]) * 0.5) * D);
    const double cc = cos(a[DIM > 1 ? 1 : 0] * D) * cos(b[DIM > 1 ? 1 : 0] * D);
    const double t1 = s1 * s1, t2 = cc * (s2 * s2);
    return t1 + t2;
  } else {
    return sqdist_nofma<DIM>(a, b, ir, aniso);
  }
}

// distance from the ranking key
__device__ __forceinline__ double metric_dist(int metric, double key, double param) {
  if (metric == GSS_METRIC_EUCLIDEAN) return sqrt(key);
  if (metric == GSS_METRIC_HAVERSINE) {
    const double r = sqrt(key);
    return 2.0 * param * asin(r < 1.0 ? r : 1.0);
  }
  return key;
}

// exp and sqrt for the covariance kernels.  The libm versions cost 26 and 19 FP64 instructions; the covariance
// assembly (K1: 10^9 evaluations per 10^6 points, K5: 2 144 per point) is bound by exactly those.
//   gss_exp : x = n ln2/64 + r, exp(x) = 2^(n>>6) * T[n&63] * (1 + r + ... + r^5/120); Cody-Waite reduction with a
//             26-bit high part (n L_HI exact for |x| < 745); 64-entry table of correctly rounded 2^(j/64).
//             Max relative error 2.2e-16 on 10^6 arguments in [-745, -1e-8] (tools/probe_fastmath.hip).
//   gss_sqrt: v_rsq_f64 seed + two Goldschmidt steps; bit-identical to libm sqrt on the same sample.
static __device__ const double GSS_EXP2_TAB[64] = {
    1.0, 1.0108892860517005, 1.0218971486541166, 1.0330248790212284,
    1.0442737824274138, 1.0556451783605572, 1.0671404006768237, 1.0787607977571199,
    1.0905077326652577, 1.102382583307841, 1.1143867425958924, 1.1265216186082418,
    1.1387886347566916, 1.1511892299529827, 1.1637248587775775, 1.1763969916502812,
    1.189207115002721, 1.202156731452703, 1.215247359980469, 1.22848053610687,
    1.241857812073484, 1.255380757024691, 1.2690509571917332, 1.2828700160787783,
    1.2968395546510096, 1.3109612115247644, 1.3252366431597413, 1.339667524053303,
    1.3542555469368927, 1.3690024229745905, 1.383909881963832, 1.3989796725383112,
    1.4142135623730951, 1.42961333839197, 1.4451808069770467, 1.460917794180647,
    1.4768261459394993, 1.4929077282912648, 1.5091644275934228, 1.5255981507445384,
    1.5422108254079407, 1.559004400237837, 1.5759808451078865, 1.593142151342267,
    1.6104903319492543, 1.6280274218573478, 1.645755478153965, 1.6636765803267364,
    1.681792830507429, 1.7001063537185235, 1.718619298122478, 1.7373338352737062,
    1.7562521603732995, 1.7753764925265212, 1.7947090750031072, 1.8142521755003989,
    1.8340080864093424, 1.8539791250833855, 1.8741676341103, 1.8945759815869656,
    1.9152065613971474, 1.9360617934922943, 1.9571441241754002, 1.978456026387951,
};

__device__ __forceinline__ double gss_exp(double x) {
  x = fmax(x, -800.0);                                     // exp underflows to 0 from -745.2 on; keeps n in int range
  const double n = __builtin_rint(x * 92.33248261689366);  // 64 / ln 2
  double r = fma(-n, 0.010830424493178725, x);             // ln2/64, 27 trailing bits cleared
  r = fma(-n, 2.030704202170295e-10, r);                    // ln2/64 - high part
  double p = fma(r, 1.0 / 120.0, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = p * r;  // exp(r) - 1
  const int ni = (int)n;
  const double t = GSS_EXP2_TAB[ni & 63];
  return __builtin_amdgcn_ldexp(fma(t, p, t), ni >> 6);
}

// Table-free form for kernels that run few waves per SIMD (K5, SGS weights): a degree-13 polynomial costs four more
// FMAs than the table lookup but no memory round trip.  Same 2.2e-16 maximum relative error.
__device__ __forceinline__ double gss_exp_poly(double x) {
  x = fmax(x, -800.0);
  const double n = __builtin_rint(x * 1.4426950408889634);
  double r = fma(-n, 6.93147180369123816490e-01, x);
  r = fma(-n, 1.90821492927058770002e-10, r);
  double p = 1.0 / 6227020800.0;
  p = fma(p, r, 1.0 / 479001600.0);
  p = fma(p, r, 1.0 / 39916800.0);
  p = fma(p, r, 1.0 / 3628800.0);
  p = fma(p, r, 1.0 / 362880.0);
  p = fma(p, r, 1.0 / 40320.0);
  p = fma(p, r, 1.0 / 5040.0);
  p = fma(p, r, 1.0 / 720.0);
  p = fma(p, r, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return __builtin_amdgcn_ldexp(p, (int)n);
}

// gss_exp_poly for arguments that are finite and <= 0 (minus a distance): the clamp at -800 only protects the integer
// conversion of n, which saturates by itself, and ldexp of a huge negative exponent is the 0 that exp underflows to.
__device__ __forceinline__ double gss_exp_poly_neg(double x) {
  const double n = __builtin_rint(x * 1.4426950408889634);
  double r = fma(-n, 6.93147180369123816490e-01, x);
  r = fma(-n, 1.90821492927058770002e-10, r);
  double p = 1.0 / 6227020800.0;
  p = fma(p, r, 1.0 / 479001600.0);
  p = fma(p, r, 1.0 / 39916800.0);
  p = fma(p, r, 1.0 / 3628800.0);
  p = fma(p, r, 1.0 / 362880.0);
  p = fma(p, r, 1.0 / 40320.0);
  p = fma(p, r, 1.0 / 5040.0);
  p = fma(p, r, 1.0 / 720.0);
  p = fma(p, r, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return __builtin_amdgcn_ldexp(p, (int)n);
}

__device__ __forceinline__ double gss_sqrt(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double s = x * y;
  const double h = 0.5 * y;
  s = fma(fma(-s, s, x), h, s);
  s = fma(fma(-s, s, x), h, s);
  return x > 0.0 ? s : 0.0;
}

// Modified Bessel functions K0, K1 for the Matern models of integer order (the reference's default MaternVariogram
// has order 1).  Chebyshev expansions generated with mpmath at 40 digits (coefficients below 3e-18 dropped):
//   x <= 2:  K0 = -ln(x/2) I0 + P0(x^2/2 - 1),   x K1 = x ln(x/2) I1 + P1(x^2/2 - 1),   I0, I1 by their power series
//   x >  2:  K0 = e^-x / sqrt(x) Q0(4/x - 1),    K1 = e^-x / sqrt(x) Q1(4/x - 1)
// Maximum relative error against mpmath on (1e-6, 700): 8e-16 (K0), 5e-16 (K1).
constexpr int GSS_BK_P0_N = 10;
static __device__ const double GSS_BK_P0[10] = {
    -0.2676636966169514, 0.3442898999246285, 0.0359799365153615,
    0.001264615411446926, 2.286212103119452e-05, 2.5347910790261494e-07,
    1.904516377220209e-09, 1.0349695257633625e-11, 4.2598161427910826e-14,
    1.3744654358807508e-16};
constexpr int GSS_BK_P1_N = 11;
static __device__ const double GSS_BK_P1[11] = {
    0.7626501136694739, -0.3531559607765449, -0.12261118082265715,
    -0.006975723859639864, -0.0001730288957513052, -2.4334061415659684e-06,
    -2.213387630734726e-08, -1.4114883926335278e-10, -6.666901694199329e-13,
    -2.427449850519366e-15, -7.023863479386288e-18};
constexpr int GSS_BK_Q0_N = 25;
static __device__ const double GSS_BK_Q0[25] = {
    1.2201515410329777, -0.0314481013119645, 0.0015698838857300533,
    -0.00012849549581627802, 1.39498137188765e-05, -1.8317555227191195e-06,
    2.766813639445015e-07, -4.660489897687948e-08, 8.574034017414225e-09,
    -1.6975345093890614e-09, 3.5773972814003283e-10, -7.957489244477396e-11,
    1.8559491149549264e-11, -4.514597883374519e-12, 1.1403405882073441e-12,
    -2.9800969231481784e-13, 8.032890775068373e-14, -2.2275133267462946e-14,
    6.3400764762765995e-15, -1.848593377920796e-15, 5.512055999401639e-16,
    -1.6782311257483392e-16, 5.210391777482758e-17, -1.6475805935875518e-17,
    5.3004337613219474e-18};
constexpr int GSS_BK_Q1_N = 25;
static __device__ const double GSS_BK_Q1[25] = {
    1.3603130952422213, 0.10392373657681724, -0.002857816859622779,
    0.00019521551847135162, -1.936197974166083e-05, 2.406484947837217e-06,
    -3.5019606030878126e-07, 5.7410841254500495e-08, -1.0345762465678097e-08,
    2.0150497551970347e-09, -4.1903547593419254e-10, 9.218315187605315e-11,
    -2.129967838427791e-11, 5.139639673482343e-12, -1.2891739609498229e-12,
    3.348419666052243e-13, -8.976705182010145e-14, 2.4771544242195966e-14,
    -7.01983708921472e-15, 2.038703166239744e-15, -6.057047270640184e-16,
    1.8380935752361397e-16, -5.689462849024281e-17, 1.7940510474681617e-17,
    -5.756744471675435e-18};

template <int N>
__device__ __forceinline__ double gss_clenshaw(const double (&c)[N], double u) {
  double b1 = 0.0, b2 = 0.0;
#pragma unroll
  for (int j = N - 1; j >= 1; --j) {
    const double t = fma(2.0 * u, b1, c[j] - b2);
    b2 = b1;
    b1 = t;
  }
  return fma(u, b1, c[0] - b2);
}

// k0 = K0(x), xk1 = x K1(x) for x > 0 (x K1 -> 1 as x -> 0, which is what the Matern shapes need)
__device__ __attribute__((noinline)) static void gss_bessel_k0_xk1(double x, double* k0, double* xk1) {
  if (x <= 2.0) {
    const double y = 0.25 * x * x;
    double s0 = 1.0, t0 = 1.0, s1 = 1.0, t1 = 1.0;
#pragma unroll
    for (int k = 1; k <= 15; ++k) {
      t0 = t0 * y * (1.0 / (double)(k * k));
      s0 += t0;
      t1 = t1 * y * (1.0 / (double)(k * (k + 1)));
      s1 += t1;
    }
    const double lg = log(0.5 * x);
    const double u = 0.5 * x * x - 1.0;
    *k0 = gss_clenshaw(GSS_BK_P0, u) - lg * s0;
    *xk1 = fma(x * lg, 0.5 * x * s1, gss_clenshaw(GSS_BK_P1, u));
  } else {
    const double u = 4.0 / x - 1.0;
    const double e = gss_exp(-x) / gss_sqrt(x);
    *k0 = e * gss_clenshaw(GSS_BK_Q0, u);
    *xk1 = x * (e * gss_clenshaw(GSS_BK_Q1, u));
  }
}

// Matern correlation of any positive order: 2^(1-nu) / Gamma(nu) d^nu K_nu(d), d > 0.  K_nu follows Temme (1975):
// nu = n + mu with |mu| <= 1/2; K_mu and K_(mu+1) from the power series (d <= 2) or Steed's continued fraction (d > 2),
// then upward recurrence.  Gamma1 / Gamma2 are Temme's auxiliary functions, expanded in T_k(8 mu^2 - 1); tables and a
// float64 transcription checked against mpmath (5.6e-15 max relative error over nu in [0.05, 20], d in [1e-9, 690]):
// tools/gen_matern_general.py.
static __device__ const double GSS_TEMME_G1[8] = {
    -0.571011340185584, 0.006516511267073688, 0.0003087090173085368,
    -3.470626964904318e-06, 6.943766448667449e-09, 3.67795398857441e-11,
    -1.3563951023664248e-13, -3.680298480635798e-17};
static __device__ const double GSS_TEMME_G2[8] = {
    0.9218702936504527, -0.07685284084478668, 0.0012719271366545622,
    -4.9717367041957395e-06, -3.3126119768180853e-08, 2.42309579004827e-10,
    -1.702377664251273e-13, -1.4943667065169001e-15};

__device__ __attribute__((noinline)) static double gss_matern_general(double d, double nu) {
  if (d > 700.0) return 0.0;
  const int n = (int)floor(nu + 0.5);
  const double mu = nu - (double)n;
  // (2 / d)^nu would overflow: the correlation differs from 1 by O(d^2) there (only reachable with nu > 1)
  if (nu * log(2.0 / d) > 690.0) return 1.0;
  const double t = 8.0 * mu * mu - 1.0;
  const double g1 = gss_clenshaw(GSS_TEMME_G1, t), g2 = gss_clenshaw(GSS_TEMME_G2, t);
  const double gampl = g2 - mu * g1;  // 1 / Gamma(1 + mu)
  const double gammi = g2 + mu * g1;  // 1 / Gamma(1 - mu)
  double kmu, kmu1;
  if (d <= 2.0) {
    const double pimu = 3.14159265358979323846 * mu;
    const double fact = fabs(pimu) < 1e-15 ? 1.0 : pimu / sin(pimu);
    const double dl = -log(0.5 * d);
    const double e = mu * dl;
    const double fact2 = fabs(e) < 1e-15 ? 1.0 : sinh(e) / e;
    double ff = fact * (g1 * cosh(e) + g2 * fact2 * dl);
    double s = ff;
    const double ee = exp(e);
    double p = 0.5 * ee / gampl;
    double q = 0.5 / (ee * gammi);
    double c = 1.0;
    const double dd = 0.25 * d * d;
    double s1 = p;
    for (int i = 1; i < 60; ++i) {
      const double fi = (double)i;
      ff = (fi * ff + p + q) / (fi * fi - mu * mu);
      c *= dd / fi;
      p /= (fi - mu);
      q /= (fi + mu);
      const double de = c * ff;
      s += de;
      s1 += c * (p - fi * ff);
      if (fabs(de) < fabs(s) * 1e-17) break;
    }
    kmu = s;
    kmu1 = s1 * 2.0 / d;
  } else {
    double b = 2.0 * (1.0 + d);
    double dd = 1.0 / b;
    double h = dd, delh = dd;
    double q1 = 0.0, q2 = 1.0;
    const double a1 = 0.25 - mu * mu;
    double q = a1, c = a1;
    double a = -a1;
    double s = 1.0 + q * delh;
    for (int i = 2; i < 500; ++i) {
      a -= 2.0 * (double)(i - 1);
      c = -a * c / (double)i;
      const double qn = (q1 - b * q2) / a;
      q1 = q2;
      q2 = qn;
      q += c * qn;
      b += 2.0;
      dd = 1.0 / (b + a * dd);
      delh = (b * dd - 1.0) * delh;
      h += delh;
      const double dels = q * delh;
      s += dels;
      if (fabs(dels) < fabs(s) * 1e-17) break;
    }
    h = a1 * h;
    kmu = sqrt(3.14159265358979323846 / (2.0 * d)) * exp(-d) / s;
    kmu1 = kmu * (mu + d + 0.5 - h) / d;
  }
  double km = kmu, kp = kmu1;
  double inv_gamma = n == 0 ? gampl * mu : gampl;  // 1 / Gamma(nu)
  for (int j = 1; j < n; ++j) {
    const double kn = km + 2.0 * (mu + (double)j) / d * kp;
    km = kp;
    kp = kn;
    inv_gamma /= (mu + (double)j);
  }
  const double knu = n == 0 ? kmu : kp;
  return exp2(1.0 - nu) * inv_gamma * pow(d, nu) * knu;
}

// g(h) = 1 - f(h / range): normalised covariance shape of one structure, from the squared distance (d2 > 0)
__device__ __forceinline__ double vg_shape(int kind, double d2, double inv_range, double mscale, double pw) {
  switch (kind) {
    case GSS_VG_GAUSSIAN: return gss_exp(-3.0 * (d2 * inv_range * inv_range));
    case GSS_VG_EXPONENTIAL: return gss_exp(-3.0 * (gss_sqrt(d2) * inv_range));
    case GSS_VG_SPHERICAL: {
      const double x = gss_sqrt(d2) * inv_range;
      return x < 1.0 ? 1.0 - (1.5 * x - 0.5 * x * x * x) : 0.0;
    }
    case VG_MATERN12: return gss_exp(-(mscale * (gss_sqrt(d2) * inv_range)));
    case VG_MATERN32: {
      const double d = mscale * (gss_sqrt(d2) * inv_range);
      return (1.0 + d) * gss_exp(-d);
    }
    case VG_MATERN52: {
      const double d = mscale * (gss_sqrt(d2) * inv_range);
      return (1.0 + d + d * d * (1.0 / 3.0)) * gss_exp(-d);
    }
    case VG_MATERN1:
    case VG_MATERN2:
    case VG_MATERN3: {  // 2^(1-nu) / Gamma(nu) d^nu K_nu(d) with K_2 = K_0 + 2 K_1 / d, K_3 = K_1 + 4 K_2 / d
      const double d = mscale * (gss_sqrt(d2) * inv_range);
      if (d > 700.0) return 0.0;
      double k0, dk1;
      gss_bessel_k0_xk1(d, &k0, &dk1);
      if (kind == VG_MATERN1) return dk1;                                   // d K1
      const double d2k2 = fma(d * d, k0, 2.0 * dk1);                         // d^2 K2
      if (kind == VG_MATERN2) return 0.5 * d2k2;
      return 0.125 * fma(d * d, dk1, 4.0 * d2k2);                            // d^3 K3 / 8
    }
    case VG_MATERN_NU: return gss_matern_general(mscale * (gss_sqrt(d2) * inv_range), pw);
    case GSS_VG_CUBIC: {
      const double x = gss_sqrt(d2) * inv_range;
      const double x2 = x * x, x3 = x2 * x;
      return x < 1.0 ? 1.0 - (7.0 * x2 - 8.75 * x3 + 3.5 * x3 * x2 - 0.75 * x3 * x3 * x) : 0.0;
    }
    case GSS_VG_SINEHOLE: {
      const double t = 3.14159265358979323846 * (gss_sqrt(d2) * inv_range);
      return sin(t) / t;
    }
    case GSS_VG_POWER: return 1.0 - mscale * pow(d2, pw);  // pseudo-covariance A - gamma(h), see gss.h
    default: {  // GSS_VG_PENTASPHERICAL
      const double x = gss_sqrt(d2) * inv_range;
      const double x2 = x * x, x3 = x2 * x;
      return x < 1.0 ? 1.0 - (1.875 * x - 1.25 * x3 + 0.375 * x3 * x2) : 0.0;
    }
  }
}

// C(a, b) = sill - gamma(a, b).  Per structure C_i = c_i g_i(h_i) for h > 0, which is algebraically identical to
// c_i - c_i f_i and avoids the cancellation; a zero lag returns the total sill (nugget included).
template <int DIM>
__device__ __forceinline__ double cov_pair(const VgDev& v, const double* a, const double* b) {
  const double d2 = sqdist_nofma<DIM>(a, b, v.ir, v.aniso != 0);
  if (d2 <= 0.0) return v.sill;  // positive radii: d2 == 0 iff the points coincide
  double c = v.cs * vg_shape(v.kind, d2, v.inv_range, v.mscale, v.pw);
  for (int e = 0; e < v.nextra; ++e) {
    const double d2e = sqdist_nofma<DIM>(a, b, v.ex[e].ir, v.ex[e].aniso != 0);
    c += v.ex[e].cs * vg_shape(v.ex[e].kind, d2e, v.ex[e].inv_range, v.ex[e].mscale, v.ex[e].pw);
  }
  return c;
}

// Out-of-line evaluation of the less common models (spherical family, sine hole, power): sin / pow / long polynomials
// would otherwise be inlined four times per call site of cov_pair4, and the moving-neighbourhood kernel has eleven
// such sites -- its code then no longer fits the instruction cache.
__device__ __attribute__((noinline)) static double vg_shape_call(int kind, double d2, double inv_range, double mscale,
                                                                 double pw) {
  return vg_shape(kind, d2, inv_range, mscale, pw);
}

// Four covariances at once (same arithmetic as cov_pair).  The model switch is taken once and the four
// evaluations sit in one basic block, so their sqrt / exp dependency chains overlap instead of running one after
// the other -- what a kernel with few waves per SIMD needs (krig_local.hip).
__device__ __forceinline__ void vg_shape4(int kind, const double* d2, double inv_range, double mscale, double pw,
                                          double* g) {
#define GSS_SHAPE4(EXPR)                                   \
  _Pragma("unroll") for (int u = 0; u < 4; ++u) {          \
    const double q2 = d2[u];                               \
    g[u] = (EXPR);                                         \
  }
  switch (kind) {
    case GSS_VG_GAUSSIAN: GSS_SHAPE4(gss_exp_poly(-3.0 * (q2 * inv_range * inv_range))); break;
    case GSS_VG_EXPONENTIAL: GSS_SHAPE4(gss_exp_poly(-3.0 * (gss_sqrt(q2) * inv_range))); break;
    case VG_MATERN12: GSS_SHAPE4(gss_exp_poly(-(mscale * (gss_sqrt(q2) * inv_range)))); break;
    case VG_MATERN32: {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double d = mscale * (gss_sqrt(d2[u]) * inv_range);
        g[u] = (1.0 + d) * gss_exp_poly(-d);
      }
      break;
    }
    case VG_MATERN52: {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double d = mscale * (gss_sqrt(d2[u]) * inv_range);
        g[u] = (1.0 + d + d * d * (1.0 / 3.0)) * gss_exp_poly(-d);
      }
      break;
    }
    default: GSS_SHAPE4(vg_shape_call(kind, q2, inv_range, mscale, pw)); break;
  }
#undef GSS_SHAPE4
}

// Single-structure models fixed at compile time (KIND = device kind): the moving-neighbourhood kernel is instantiated
// per model so that each of its covariance call sites carries one formula instead of all of them (instruction cache).
template <int KIND>
__device__ __forceinline__ double vg_shape_k(double d2, double inv_range, double mscale, double pw) {
  if (KIND == GSS_VG_GAUSSIAN) return gss_exp_poly(-3.0 * (d2 * inv_range * inv_range));
  if (KIND == GSS_VG_EXPONENTIAL) return gss_exp_poly(-3.0 * (gss_sqrt(d2) * inv_range));
  if (KIND == VG_MATERN12) return gss_exp_poly(-(mscale * (gss_sqrt(d2) * inv_range)));
  if (KIND == VG_MATERN32) {
    const double d = mscale * (gss_sqrt(d2) * inv_range);
    return (1.0 + d) * gss_exp_poly(-d);
  }
  if (KIND == VG_MATERN52) {
    const double d = mscale * (gss_sqrt(d2) * inv_range);
    return (1.0 + d + d * d * (1.0 / 3.0)) * gss_exp_poly(-d);
  }
  if (KIND == GSS_VG_SPHERICAL) {
    const double x = gss_sqrt(d2) * inv_range;
    return x < 1.0 ? 1.0 - (1.5 * x - 0.5 * x * x * x) : 0.0;
  }
  return vg_shape(KIND, d2, inv_range, mscale, pw);
}

template <int DIM, int KIND, bool UNIT>
__device__ __forceinline__ double cov_pair_k(const VgDev& v, const double* a, const double* b);
template <int DIM, int KIND, bool UNIT>
__device__ __forceinline__ void cov_pair4_k(const VgDev& v, const double (*a)[DIM], const double* b, double* out);

// out[u] = C(a_u, b) for four points a_u and one point b
template <int DIM>
__device__ __forceinline__ void cov_pair4(const VgDev& v, const double (*a)[DIM], const double* b, double* out) {
  double d2[4], g[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) d2[u] = sqdist_nofma<DIM>(a[u], b, v.ir, v.aniso != 0);
  vg_shape4(v.kind, d2, v.inv_range, v.mscale, v.pw, g);
  double c[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) c[u] = v.cs * g[u];
  for (int e = 0; e < v.nextra; ++e) {
    double d2e[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) d2e[u] = sqdist_nofma<DIM>(a[u], b, v.ex[e].ir, v.ex[e].aniso != 0);
    vg_shape4(v.ex[e].kind, d2e, v.ex[e].inv_range, v.ex[e].mscale, v.ex[e].pw, g);
#pragma unroll
    for (int u = 0; u < 4; ++u) c[u] += v.ex[e].cs * g[u];
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) out[u] = d2[u] <= 0.0 ? v.sill : c[u];
}

// KIND < 0: any model (the general functions above); otherwise one structure of device kind KIND, v.nextra == 0
// KIND >= 0 (single structure): the coordinates arrive already divided by the radii of the structure's ball (the caller
// scales each point once, krig_local.hip), so a pair costs three differences and one fused sum of squares instead of
// a multiply and a select per coordinate; the square root runs on max(d2, 1e-300) without its zero guard (a zero lag
// is replaced by the total sill afterwards anyway).  KIND < 0: any model, coordinates as they are.
// UNIT: the caller has also multiplied the coordinates by the model's own scale (kpos_scale below), so the scaled
// distance IS the argument of the shape -- two multiplications fewer per pair (the moving-neighbourhood kernel scales its
// 64 neighbours once and evaluates 2 048 pairs).
template <int KIND>
__device__ __forceinline__ double kpos_scale(const VgDev& v) {
  if (KIND == GSS_VG_GAUSSIAN) return 1.7320508075688772 * v.inv_range;   // exp(-3 (h / r)^2) = exp(-|sqrt(3) h / r|^2)
  if (KIND == GSS_VG_EXPONENTIAL) return 3.0 * v.inv_range;
  if (KIND == GSS_VG_SPHERICAL) return v.inv_range;
  return v.mscale * v.inv_range;                                           // Matern 1/2, 3/2, 5/2
}
template <int KIND, bool UNIT = false>
__device__ __forceinline__ double vg_shape_kpos(double d2, double inv_range, double mscale, double pw) {
  if (KIND == GSS_VG_GAUSSIAN) return gss_exp_poly_neg(UNIT ? -d2 : -3.0 * (d2 * inv_range * inv_range));
  const double y = __builtin_amdgcn_rsq(d2);  // gss_sqrt without the zero guard
  double sq = d2 * y;
  const double hy = 0.5 * y;
  sq = fma(fma(-sq, sq, d2), hy, sq);
  sq = fma(fma(-sq, sq, d2), hy, sq);
  if (KIND == GSS_VG_EXPONENTIAL) return gss_exp_poly_neg(UNIT ? -sq : -3.0 * (sq * inv_range));
  if (KIND == VG_MATERN12) return gss_exp_poly_neg(UNIT ? -sq : -(mscale * (sq * inv_range)));
  if (KIND == VG_MATERN32) {
    const double d = UNIT ? sq : mscale * (sq * inv_range);
    return (1.0 + d) * gss_exp_poly_neg(-d);
  }
  if (KIND == VG_MATERN52) {
    const double d = UNIT ? sq : mscale * (sq * inv_range);
    return (1.0 + d + d * d * (1.0 / 3.0)) * gss_exp_poly_neg(-d);
  }
  if (KIND == GSS_VG_SPHERICAL) {
    const double x = UNIT ? sq : sq * inv_range;
    return x < 1.0 ? 1.0 - (1.5 * x - 0.5 * x * x * x) : 0.0;
  }
  return vg_shape(KIND, d2, inv_range, mscale, pw);
}

// a * b rounded once and never fused into a following addition: the scaled coordinates of a sample and of an estimation
// point that coincide must be the SAME doubles, so that their distance is exactly zero (C(0) = sill, not sill - nugget)
__device__ __forceinline__ double mul_rounded(double a, double b) {
#pragma clang fp contract(off)
  const double r = a * b;
  return r;
}

template <int DIM>
__device__ __forceinline__ double sqdist_scaled(const double* a, const double* b) {
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < DIM; ++k) {
    const double t = a[k] - b[k];
    acc = fma(t, t, acc);
  }
  return acc;
}

template <int DIM, int KIND, bool UNIT = false>
__device__ __forceinline__ double cov_pair_k(const VgDev& v, const double* a, const double* b) {
  if (KIND < 0) return cov_pair<DIM>(v, a, b);
  const double d2 = sqdist_scaled<DIM>(a, b);
  const double g = vg_shape_kpos<(KIND < 0 ? 0 : KIND), UNIT>(fmax(d2, 1e-300), v.inv_range, v.mscale, v.pw);
  return d2 <= 0.0 ? v.sill : v.cs * g;
}

template <int DIM, int KIND, bool UNIT = false>
__device__ __forceinline__ void cov_pair4_k(const VgDev& v, const double (*a)[DIM], const double* b, double* out) {
  if (KIND < 0) {
    cov_pair4<DIM>(v, a, b, out);
    return;
  }
  double d2[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) d2[u] = sqdist_scaled<DIM>(a[u], b);
#pragma unroll
  for (int u = 0; u < 4; ++u)
    out[u] = v.cs * vg_shape_kpos<(KIND < 0 ? 0 : KIND), UNIT>(fmax(d2[u], 1e-300), v.inv_range, v.mscale, v.pw);
  // C(0) = sill.  At a zero lag the shape evaluated at 1e-300 is exactly 1 for every one of these models, i.e. the value
  // above is cs = sill - nugget: only a model WITH a nugget needs the select (wave-uniform: a scalar branch)
  if (v.cs != v.sill) {
#pragma unroll
    for (int u = 0; u < 4; ++u) out[u] = d2[u] <= 0.0 ? v.sill : out[u];
  }
}

// four independent pairs (a[u], b[u]): the paired diagonal tiles of the moving-neighbourhood kernel, where the column
// a lane works on depends on the row (krig_local.hip)
template <int DIM, int KIND, bool UNIT = false>
__device__ __forceinline__ void cov_pairs4_k(const VgDev& v, const double (*a)[DIM], const double (*b)[DIM], double* out) {
  if (KIND < 0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) out[u] = cov_pair<DIM>(v, a[u], b[u]);
    return;
  }
  double d2[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) d2[u] = sqdist_scaled<DIM>(a[u], b[u]);
#pragma unroll
  for (int u = 0; u < 4; ++u)
    out[u] = v.cs * vg_shape_kpos<(KIND < 0 ? 0 : KIND), UNIT>(fmax(d2[u], 1e-300), v.inv_range, v.mscale, v.pw);
  if (v.cs != v.sill) {   // (see cov_pair4_k)
#pragma unroll
    for (int u = 0; u < 4; ++u) out[u] = d2[u] <= 0.0 ? v.sill : out[u];
  }
}

// ---------------------------------------------------------------------------------------------
// dense FP64 toolkit (dense_la.hip, gemm_f64.hip)
// ---------------------------------------------------------------------------------------------
// D(i,j) = alpha * sum_k A(i,k) B(k,j) + beta * D(i,j), arbitrary element strides.
int32_t gemm_f64(int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t sa_i, int64_t sa_k,
                 const double* B, int64_t sb_k, int64_t sb_j, double beta, double* D, int64_t sd_i,
                 int64_t sd_j, bool lower_only, hipStream_t s, int tri = 0);
// `tri`: zero structure of an operand (k-tiles that only meet zeros are skipped): 1 = B(k,j) = 0 for k > j,
// 2 = B(k,j) = 0 for k < j, 4 = A(i,k) = 0 for k > i (GEMM_TRI_* in dense_la.hip)
// y = op(A) x for column-major A (m x n, lda); trans: y = A' x.  work: gemv_work_doubles(trans, m, n) doubles.
int64_t gemv_work_doubles(bool trans, int64_t m, int64_t n);
int32_t gemv_f64(bool trans, int64_t m, int64_t n, const double* A, int64_t lda, const double* x, double* y,
                 double* work, hipStream_t s);
// in-place Cholesky of the lower triangle (column-major); *d_info (device int) set to 1+row on failure.
// dinv: workspace of potrf_dinv_doubles(n) doubles that receives inv() of every 64 x 64 diagonal leaf block.
// Asynchronous on s.
int64_t potrf_dinv_doubles(int64_t n);
int32_t potrf_f64(double* A, int64_t n, int64_t lda, int* d_info, double* dinv, hipStream_t s);
// Factor and inverse together (A = L L', W = inv(L)); W's strict upper triangle must be zero on entry, scr holds
// max(n * n, potrf_inverse_work_doubles(n)) doubles.  *d_info = -1: the single-launch panel gave up at a grid barrier.  *d_info as potrf_f64.  Asynchronous on s.
// keep_L: also leave the complete factor L in the lower triangle of A (otherwise only its diagonal leaf blocks).
int64_t potrf_inverse_work_doubles(int64_t n);
// after a *d_info of -1: no more single-launch panels in this process (callers retry on the launch-per-block recursion)
void potrf_panel_disable();
// Settings that HIP keeps per device (function attributes, library plans): true the first time this is called with
// `mask` on the current device.  Exports hold the library lock, so the word needs no atomics.
inline bool first_on_this_device(uint64_t& mask) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (mask & bit) return false;
  mask |= bit;
  return true;
}
int comm_last_ipc_route();   // comm.hip
int panel_giveups();          // dense_la.hip: times a single-launch factorisation gave up at a grid barrier
void lu_grid_disable();   // lu.hip: 32-column LU panels on one workgroup instead of the grid
// padded16: the caller guarantees that rows and columns n .. 16 ceil(n / 16) - 1 of A and W exist in memory and are
// zero (they stay zero): a size that is not a multiple of 16 then still runs on the single-launch panel kernel.
int32_t potrf_inverse_f64(double* A, int64_t n, int64_t lda, double* W, int64_t ldw, double* scr, int* d_info,
                          bool keep_L, hipStream_t s, bool padded16 = false);
// Cholesky of a large block without its inverse, by panels (work: potrf_blocked_work_doubles(n) doubles).
int64_t potrf_blocked_work_doubles(int64_t n);
int32_t potrf_blocked_f64(double* A, int64_t n, int64_t lda, int* d_info, double* work, hipStream_t s);
// Data columns of the LUGS preprocess as one blocked factorisation of [C11 . ; C21 C22] (dense_la.hip): C11 <- L11,
// C21 (mb >= ns rows, leading dimension ld21) <- C21 L11^-T, lower tiles of C22 <- C22 - A21 A21' (first ns rows of C21).
// An event that lives for one call (timing disabled): created on demand, destroyed when the scope ends -- also on the
// early returns of GSS_TRY / GSS_HIP.  (Destroying an event that a stream still waits for is allowed: the runtime
// releases it once the wait has been satisfied.)
struct ScopedEvent {
  hipEvent_t e = nullptr;
  ScopedEvent() = default;
  ScopedEvent(const ScopedEvent&) = delete;
  ScopedEvent& operator=(const ScopedEvent&) = delete;
  ~ScopedEvent() {
    if (e) (void)hipEventDestroy(e);
  }
  hipError_t create() { return e ? hipSuccess : hipEventCreateWithFlags(&e, hipEventDisableTiming); }
  operator hipEvent_t() const { return e; }
};

// the process's low-priority helper stream of those look-aheads (nullptr if it cannot be created); work put on it
// must be fenced by events against the caller's stream on both sides
// helper streams of the process (common.hip), created together in a fixed order
enum { HELPER_LOOKAHEAD = 0, HELPER_GEN0 = 1, HELPER_GEN1 = 2, HELPER_GEN2 = 3, HELPER_FIT = 4, HELPER_COUNT = 5 };
hipStream_t helper_stream(int which);
inline hipStream_t lookahead_stream() { return helper_stream(HELPER_LOOKAHEAD); }
int64_t potrf_joint_work_doubles(int64_t nd, int64_t mb);
int32_t potrf_joint_f64(double* C11, int64_t nd, double* C21, int64_t mb, int64_t ld21, double* C22, int64_t ns,
                        int* d_info, double* work, hipStream_t s);
// lu.hip: A <- unit lower-triangular L of the partial-pivot LU P A = L U (`lu(A).L`, lu.jl:70); ipiv: n device ints,
// *d_info (zeroed by the caller) = 1 + column of an exactly zero pivot.  mirror_lower: upper triangle <- lower'.
int32_t getrf_unit_lower_f64(double* A, int64_t n, int64_t lda, int* ipiv, int* d_info, hipStream_t s);
int32_t mirror_lower_f64(double* A, int64_t n, int64_t lda, hipStream_t s);
// W = inv(L), lower, column-major; W's strict upper triangle must be zero on entry; T is scratch of
// at least (n/2+64)^2 doubles; dinv (nullable) = cached leaf inverses from potrf_f64
int32_t trtri_f64(const double* L, int64_t n, int64_t ldl, double* W, int64_t ldw, double* T, const double* dinv,
                  hipStream_t s);
// X <- X * inv(L)' (X is m x n column-major); scratch >= 64*64 doubles (unused when dinv is given)
int32_t trsm_right_lt_f64(double* X, int64_t m, int64_t n, int64_t ldx, const double* L, int64_t ldl,
                          double* scratch, const double* dinv, hipStream_t s);

// Fill / copy kernels used instead of hipMemsetAsync / hipMemcpyAsync inside sequences that may be captured into a
// hipGraph: with ROCm 7.2 a graph's memset nodes running next to plain memsets of another stream were observed to
// write a stale fill pattern (0x24242424 into a word that was cleared with 0).
int32_t dev_zero_bytes(void* p, size_t bytes, hipStream_t s);             // p 4-byte aligned, bytes multiple of 4
int32_t dev_copy_f64(double* dst, const double* src, int64_t n, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// covariance assembly (cov_kernels.hip)
// ---------------------------------------------------------------------------------------------
// out[i * ldo + j] = C(a_i, b_j); device pointers
int32_t cov_pairwise_dev(const VgDev& vg, const double* a, int64_t na, const double* b, int64_t nb, double* out,
                         int64_t ldo, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// neighbour search (knn.hip)
// ---------------------------------------------------------------------------------------------
struct KnnIndex {
  DevBuf xs, perm, lo, hi;  // k-d ordered coordinates (n x dim), original indices, batch box corners (nb x dim)
  DevBuf lo1, hi1;          // boxes of groups of 64 consecutive batches (nb1 x dim)
  int64_t n = 0;
  int nb = 0, nb1 = 0, dim = 0;
};
int32_t knn_index_build(const double* xhost, int64_t n, int dim, KnnIndex* ix, hipStream_t s);
int32_t knn_index_build_from_device(const double* xdev, int64_t n, int dim, KnnIndex* ix, hipStream_t s);
// the same ordering rule carried out on the device (knn_build.hip); used from KNN_DEVICE_BUILD_MIN samples
int32_t knn_index_build_device(const double* xdev, int64_t n, int dim, KnnIndex* ix, hipStream_t s);
constexpr int64_t KNN_DEVICE_BUILD_MIN = 16384;
// rank / qrank / bminrank (all or none): masked search of sequential simulation, see knn.hip
int32_t knn_search_indexed(const KnnIndex& ix, const double* centers, int64_t m, int k, double radius,
                           const double* inv_radii_host, int* idx, int* count, hipStream_t s,
                           const int* rank = nullptr, const int* qrank = nullptr, const int* bminrank = nullptr,
                           int metric = 0 /* Euclidean, Cityblock or Chebyshev */, const double* lowd = nullptr,
                           const int* lowi = nullptr /* per-query lower bound of the accepted keys */);
// any k <= n: passes of 64 neighbours, each bounded below by the last key of the pass before (idx is m x k);
// xdata = the samples in their original order (device)
// rank / qrank / bminrank: the mask of SGS (candidates whose rank is below the query's), as in knn_search_indexed
int32_t knn_search_indexed_any(const KnnIndex& ix, const double* xdata, const double* centers, int64_t m, int k,
                               double radius, const double* inv_radii_host, int* idx, int* count, hipStream_t s,
                               int metric = 0, const int* rank = nullptr, const int* qrank = nullptr,
                               const int* bminrank = nullptr);
// Haversine: exhaustive kernel (no box bounds for that key); non-Euclidean metrics do not combine with balls
int32_t knn_search_dev(const double* xdata, int64_t n, int dim, const double* centers, int64_t m, int k,
                       double radius, const double* inv_radii_host, int* idx, int* count, hipStream_t s,
                       int metric = 0);
int32_t check_metric(int metric, double metric_param, int dim, double radius, const double* inv_radii);

// ---------------------------------------------------------------------------------------------
// noise (noise.hip)
// ---------------------------------------------------------------------------------------------
// (nreals > 1: realisations real .. real + nreals - 1 in one launch, outputs bstride doubles apart)
int32_t philox_uniform_dev(uint64_t seed, int64_t real, int64_t n, double* out, int64_t ld_pad_n1, int64_t n1,
                           hipStream_t s, int nreals = 1, int64_t bstride = 0);
int32_t philox_normal_dev(uint64_t seed, int64_t real, int64_t n, double* out, hipStream_t s);

}  // namespace gss
