// HBM access-pattern probe for the FFTGS passes (FFTGS).
//   1. plain streams: copy / read-only / write-only with 16 B per lane and several loads in flight per lane
//      (the anchor the passes are graded against; MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy);
//   2. the tile shapes of the strided passes, as pure in-place copies through registers (no LDS, no arithmetic):
//      a tile = L rows of TX complex doubles (TX * 16 B contiguous) at the row stride of the y pass (one padded
//      x row) or of the z pass (one padded xy plane) of a 512^3 half-spectrum buffer.
// Tells how much of a pass's time is the memory system's answer to its access pattern and how much is the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int U>
__global__ __launch_bounds__(256) void copy_kernel(const double2* __restrict__ in, double2* __restrict__ out, size_t n) {
  const size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  double2 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) v[u] = in[base + (size_t)u * 256];
#pragma unroll
  for (int u = 0; u < U; ++u) out[base + (size_t)u * 256] = v[u];
}

template <int U>
__global__ __launch_bounds__(256) void read_kernel(const double2* __restrict__ in, double* __restrict__ sink, size_t n) {
  const size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  double acc = 0.0;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const double2 v = in[base + (size_t)u * 256];
    acc += v.x + v.y;
  }
  if (acc == 1.2345e300) sink[0] = acc;
}

template <int U>
__global__ __launch_bounds__(256) void write_kernel(double2* __restrict__ out, size_t n, double val) {
  const size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
#pragma unroll
  for (int u = 0; u < U; ++u) out[base + (size_t)u * 256] = make_double2(val, val + u);
}

// in-place tile copy: tile (o, t) = rows j = 0..L-1 at X + o * ostride + t * TX + j * lstride, TX complex each
template <int TXLOG, int E /* elements per thread = L * TX / 256 */, bool XCDPAIR>
__global__ __launch_bounds__(256) void tile_copy_kernel(double2* __restrict__ X, int ntx, int64_t ostride, int64_t lstride,
                                                        double scale) {
  constexpr int TX = 1 << TXLOG;
  int tile = blockIdx.x;
  if (XCDPAIR && TXLOG < 3) {
    constexpr int GL = 3 - TXLOG;   // tiles that share one 128-B line go to blocks b, b+8, ... (same XCD)
    const int bb = blockIdx.x;
    tile = (bb & ~((8 << GL) - 1)) + ((bb & 7) << GL) + ((bb >> 3) & ((1 << GL) - 1));
  }
  const int t = tile % ntx, o = tile / ntx;
  double2* base = X + (int64_t)o * ostride + (int64_t)t * TX;
  double2 v[E];
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const int e = threadIdx.x + i * 256;
    const int c = e & (TX - 1), j = e >> TXLOG;
    v[i] = base[(int64_t)j * lstride + c];
  }
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const int e = threadIdx.x + i * 256;
    const int c = e & (TX - 1), j = e >> TXLOG;
    base[(int64_t)j * lstride + c] = make_double2(v[i].x * scale, v[i].y * scale);
  }
}

template <class F>
static int timeit(const char* name, double bytes, F launch) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep >= 1 && ms < best) best = ms;
  }
  CK(hipGetLastError());
  printf("%-58s %8.3f ms  %7.1f GB/s\n", name, best, bytes / best / 1e6);
  return 0;
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs %d\n", p.gcnArchName, p.multiProcessorCount);
  const size_t n = (size_t)1 << 27;  // 2 GiB as double2
  double2 *a, *b;
  double* sink;
  CK(hipMalloc(&a, n * sizeof(double2)));
  CK(hipMalloc(&b, n * sizeof(double2)));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(a, 0, n * sizeof(double2)));
  CK(hipMemset(b, 0, n * sizeof(double2)));
  const double GB2 = 2.0 * n * 16, GB1 = 1.0 * n * 16;
#define COPY(U) if (timeit("copy 2 GiB -> 2 GiB, " #U " x 16 B in flight per lane (r+w bytes)", GB2, [&] { \
    hipLaunchKernelGGL(copy_kernel<U>, dim3((unsigned)(n / 256 / U)), dim3(256), 0, 0, a, b, n); })) return 1;
  COPY(1) COPY(2) COPY(4) COPY(8)
#define READ(U) if (timeit("read 2 GiB, " #U " x 16 B in flight per lane", GB1, [&] { \
    hipLaunchKernelGGL(read_kernel<U>, dim3((unsigned)(n / 256 / U)), dim3(256), 0, 0, a, sink, n); })) return 1;
  READ(4) READ(8)
#define WRITE(U) if (timeit("write 2 GiB, " #U " x 16 B per lane", GB1, [&] { \
    hipLaunchKernelGGL(write_kernel<U>, dim3((unsigned)(n / 256 / U)), dim3(256), 0, 0, b, n, 1.5); })) return 1;
  WRITE(4) WRITE(8)

  // 512^3 half spectrum: nh = 257 -> pitch 264 complex, n2 = n3 = 512
  const int L = 512, nhp = 264, n2 = 512, n3 = 512;
  const double tile_bytes = 2.0 * (double)nhp * n2 * n3 * 16;
  double2* X = a;   // 264 * 512 * 512 * 16 B = 1.03 GiB fits
  printf("-- in-place tile copies over the padded 512^3 half spectrum (%.2f GB read + written)\n", tile_bytes / 1e9);
#define TILE(TXLOG, PAIR, GEO) { \
    constexpr int TX = 1 << TXLOG; constexpr int E = 512 * TX / 256; const int ntx = nhp / TX; \
    const bool ygeo = (GEO) == 0; \
    const int64_t ostride = ygeo ? (int64_t)n2 * nhp : (int64_t)nhp, lstride = ygeo ? (int64_t)nhp : (int64_t)n2 * nhp; \
    const unsigned blocks = (unsigned)((ygeo ? n3 : n2) * ntx); \
    char nm[128]; snprintf(nm, sizeof nm, "%s pass tiles, TX = %d (%d B rows), %s", ygeo ? "y" : "z", TX, TX * 16, \
                           PAIR ? "line-sharing blocks b, b+8" : "tiles in block order"); \
    if (timeit(nm, 2.0 * (double)ntx * TX * n2 * n3 * 16, [&] { hipLaunchKernelGGL((tile_copy_kernel<TXLOG, E, PAIR>), dim3(blocks), dim3(256), 0, 0, \
                                                       X, ntx, ostride, lstride, 1.0); })) return 1; }
  TILE(1, true, 0) TILE(1, false, 0) TILE(2, true, 0) TILE(2, false, 0) TILE(3, false, 0) TILE(4, false, 0)
  TILE(1, true, 1) TILE(1, false, 1) TILE(2, true, 1) TILE(2, false, 1) TILE(3, false, 1) TILE(4, false, 1)
  (void)L;
  return 0;
}
