// Probe for pairing two FFTGS passes INSIDE one launch through an XCD's L2.
//
// The pair P1|P2 of the fused pipeline (x lines, then y lines of the same z plane) as pure data movement, no
// arithmetic: phase A writes the 512 rows of a plane of the padded half spectrum (264 complex each) into a scratch
// plane, phase B reads that plane back as 32 (+1) tiles of 8 columns x 512 rows and writes them to their place in the
// 1.1 GB half-spectrum buffer X.  Variants:
//   separate : two launches, the plane going through X itself (what the library does today; 1.1 GB written, 1.1 GB read
//              back, 1.1 GB written)
//   paired   : ONE persistent launch; the workgroups that find themselves on one XCD (HW_REG_XCC_ID, read at run time --
//              placement is not assumed) form a team that takes whole planes: rows -> team scratch plane (2.16 MB, plain
//              stores: they stay in that XCD's L2) -> every storing wave drains (s_waitcnt vmcnt(0)) -> team barrier (one
//              agent-scope atomic add per workgroup, sc1 poll) -> tiles read with sc1 loads (served by the L2, never by a
//              stale L1) -> X.  Single scratch plane + two barriers per plane, or two scratch planes + one barrier.
// Every word that phase B reads is checked against what phase A wrote for THAT plane (a stale line shows as the value
// of an earlier plane), so the run is also the correctness test of the same-XCD hand-off under load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int NH = 257, NHP = 264, N2 = 512, N3 = 512;
constexpr int NT = 512;
constexpr int TILES = NHP / 8;   // 33
typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xfu; }
__device__ __forceinline__ double2 val(int plane, int row, int col) {
  return make_double2((double)(plane * 1000003 + row * 517 + col), (double)(col * 7919 - row + plane));
}

struct Ctl {                // zeroed before every launch
  unsigned registered;      // workgroups that know their team
  unsigned next_plane;      // dynamic plane queue (one dequeue per team and plane)
  unsigned errors;
  unsigned timeouts;
  unsigned pad0[28];
  unsigned team_size[8][32];   // [xcc][0] on a line of its own
  unsigned team_bar[8][32];    // arrival counter of the team barrier
  unsigned team_plane[8][32];  // plane the team works on (written by its leader before the barrier)
};

__device__ __forceinline__ bool spin_ge(unsigned* p, unsigned target, unsigned* tmo) {
  unsigned spins = 0;
  while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    if (++spins > 20000000u) {
      atomicAdd(tmo, 1u);
      return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  return true;
}

// team barrier: every wave has drained its stores before the workgroup barrier; one lane arrives and polls
__device__ __forceinline__ bool team_barrier(unsigned* cnt, unsigned& epoch, unsigned size, unsigned* tmo, int* s_ok) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  ++epoch;
  if (threadIdx.x == 0) {
    atomicAdd(cnt, 1u);
    *s_ok = spin_ge(cnt, epoch * size, tmo) ? 1 : 0;
  }
  __syncthreads();
  return *s_ok != 0;
}

// MODE 0: one scratch plane, barriers A->B and B->A.  MODE 1: two scratch planes, one barrier per plane.
template <int MODE, bool SC1>
__global__ __launch_bounds__(NT) void paired_kernel(Ctl* ctl, double2* __restrict__ S, double2* __restrict__ X, int nplanes) {
  __shared__ int s_ok;
  __shared__ unsigned s_team, s_rank, s_size, s_plane;
  const int tid = threadIdx.x;
  if (tid == 0) {
    const unsigned x = xcc_id();
    s_team = x;
    s_rank = atomicAdd(&ctl->team_size[x][0], 1u);
    atomicAdd(&ctl->registered, 1u);
    s_ok = spin_ge(&ctl->registered, gridDim.x, &ctl->timeouts) ? 1 : 0;   // every workgroup of the grid is resident
    s_size = __hip_atomic_load(&ctl->team_size[x][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!s_ok) return;
  const unsigned team = s_team, rank = s_rank, size = s_size;
  unsigned* bar = &ctl->team_bar[team][0];
  unsigned epoch = 0;
  double2* Steam = S + (size_t)team * 2 * N2 * NHP;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(Steam, 0, (int)(2 * N2 * NHP * sizeof(double2)), 0x00020000);
  // the team's leader takes the first plane; the others learn it behind a barrier
  if (rank == 0 && tid == 0)
    __hip_atomic_store(&ctl->team_plane[team][0], atomicAdd(&ctl->next_plane, 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (!team_barrier(bar, epoch, size, &ctl->timeouts, &s_ok)) return;
  for (int it = 0;; ++it) {
    if (tid == 0) s_plane = __hip_atomic_load(&ctl->team_plane[team][it & 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int plane = (int)s_plane;
    if (plane >= nplanes) return;
    const int sbase = MODE == 1 ? (it & 1) * N2 * NHP : 0;
    double2* Sp = Steam + sbase;
    // ---- phase A: rows rank, rank + size, ... ; 264 complex per row, 16 B per lane, plain stores
    for (int row = (int)rank; row < N2; row += (int)size)
      for (int c = tid; c < NHP; c += NT) Sp[(size_t)row * NHP + c] = val(plane, row, c);
    // (the leader takes the next plane; its peers read the word behind the barrier, at the top of the next iteration)
    if (rank == 0 && tid == 0)
      __hip_atomic_store(&ctl->team_plane[team][(it + 1) & 1], atomicAdd(&ctl->next_plane, 1u), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    if (!team_barrier(bar, epoch, size, &ctl->timeouts, &s_ok)) return;
    // ---- phase B: tiles rank, rank + size, ... of 8 columns x 512 rows: 8 rows per thread in flight
    for (int t = (int)rank; t < TILES; t += (int)size) {
      const int c = tid & 7, r = tid >> 3;   // 64 row groups
      double2 v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int row = r + 64 * q;
        if (SC1) {
          const v4i w = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((sbase + row * NHP + t * 8 + c) * sizeof(double2)), 0, 16);
          v[q] = __builtin_bit_cast(double2, w);
        } else {
          v[q] = Sp[(size_t)row * NHP + t * 8 + c];
        }
      }
      double2* xo = X + (size_t)plane * N2 * NHP + t * 8 + c;
#pragma unroll
      for (int q = 0; q < 8; ++q) xo[(size_t)(r + 64 * q) * NHP] = v[q];
    }
    // one scratch plane: phase B of this plane must be over everywhere before the next plane's rows are stored
    if (MODE == 0 && !team_barrier(bar, epoch, size, &ctl->timeouts, &s_ok)) return;
  }
}

// ---- the two separate launches of today
__global__ __launch_bounds__(256) void rows_kernel(double2* __restrict__ X) {   // 4 rows per workgroup
  const size_t row0 = (size_t)blockIdx.x * 4;
  for (int e = threadIdx.x; e < 4 * NHP; e += 256) {
    const int r = e / NHP, c = e - r * NHP;
    const size_t row = row0 + r;
    X[row * NHP + c] = val((int)(row / N2), (int)(row % N2), c);
  }
}
__global__ __launch_bounds__(NT) void tiles_kernel(double2* __restrict__ X, unsigned* errors, int check) {
  const int t = blockIdx.x % TILES, plane = blockIdx.x / TILES;
  const int c = threadIdx.x & 7, r = threadIdx.x >> 3;
  double2* xo = X + (size_t)plane * N2 * NHP + t * 8 + c;
  double2 v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = xo[(size_t)(r + 64 * q) * NHP];
  unsigned nerr = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    if (check) {
      const double2 e = val(plane, r + 64 * q, t * 8 + c);
      nerr += (v[q].x != e.x || v[q].y != e.y) ? 1u : 0u;
    }
    xo[(size_t)(r + 64 * q) * NHP] = make_double2(v[q].x + 1.0, v[q].y);
  }
  if (nerr) atomicAdd(errors, nerr);
}

// verification of the paired runs on the result: X must hold every plane's values
__global__ __launch_bounds__(256) void verify_kernel(const double2* __restrict__ X, unsigned* errors) {
  const size_t row = blockIdx.x;
  unsigned nerr = 0;
  for (int c = threadIdx.x; c < NHP; c += 256) {
    const double2 e = val((int)(row / N2), (int)(row % N2), c), v = X[row * NHP + c];
    nerr += (v.x != e.x || v.y != e.y) ? 1u : 0u;
  }
  if (nerr) atomicAdd(errors, nerr);
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 5;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s CUs %d\n", prop.gcnArchName, prop.multiProcessorCount);
  const size_t nX = (size_t)N3 * N2 * NHP;
  double2 *X, *S;
  Ctl* ctl;
  unsigned* err;
  CK(hipMalloc(&X, nX * sizeof(double2)));
  CK(hipMalloc(&S, (size_t)8 * 2 * N2 * NHP * sizeof(double2)));
  CK(hipMalloc(&ctl, sizeof(Ctl)));
  CK(hipMalloc(&err, 4));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  auto report = [&](const char* name, float ms, double bytes) {
    unsigned herr = 0;
    Ctl h;
    (void)hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost);
    printf("%-78s %7.3f ms  %7.1f GB/s on %.2f GB  wrong words %u  timeouts %u  teams", name, ms, bytes / ms / 1e6, bytes / 1e9,
           herr, h.timeouts);
    for (int x = 0; x < 8; ++x) printf(" %u", h.team_size[x][0]);
    printf("\n");
  };
  // ---- separate launches
  for (int rep = 0; rep < reps; ++rep) {
    CK(hipMemset(err, 0, 4));
    CK(hipMemset(ctl, 0, sizeof(Ctl)));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(rows_kernel, dim3(N3 * N2 / 4), dim3(256), 0, 0, X);
    hipLaunchKernelGGL(tiles_kernel, dim3(N3 * TILES), dim3(NT), 0, 0, X, err, 1);
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    if (rep == reps - 1) report("separate: rows kernel + in-place tile kernel (plane through X)", ms, 3.0 * nX * 16);
  }
  // ---- paired, one persistent launch
  struct Var { const char* name; int mode; bool sc1; int per_cu; };
  const Var vars[] = {{"paired: 1 scratch plane, 2 barriers/plane, sc1 loads, 1 WG/CU", 0, true, 1},
                      {"paired: 2 scratch planes, 1 barrier/plane, sc1 loads, 1 WG/CU", 1, true, 1},
                      {"paired: 2 scratch planes, 1 barrier/plane, sc1 loads, 2 WG/CU", 1, true, 2},
                      {"paired: 1 scratch plane, 2 barriers/plane, PLAIN loads (expected stale), 1 WG/CU", 0, false, 1}};
  for (const Var& v : vars) {
    for (int rep = 0; rep < reps; ++rep) {
      CK(hipMemset(err, 0, 4));
      CK(hipMemset(ctl, 0, sizeof(Ctl)));
      CK(hipMemset(X, 0xff, nX * sizeof(double2)));
      CK(hipDeviceSynchronize());
      const int grid = prop.multiProcessorCount * v.per_cu;
      CK(hipEventRecord(a));
      if (v.mode == 0 && v.sc1) hipLaunchKernelGGL((paired_kernel<0, true>), dim3(grid), dim3(NT), 0, 0, ctl, S, X, N3);
      else if (v.mode == 1 && v.sc1) hipLaunchKernelGGL((paired_kernel<1, true>), dim3(grid), dim3(NT), 0, 0, ctl, S, X, N3);
      else hipLaunchKernelGGL((paired_kernel<0, false>), dim3(grid), dim3(NT), 0, 0, ctl, S, X, N3);
      CK(hipEventRecord(b));
      CK(hipDeviceSynchronize());
      float ms = 0;
      CK(hipEventElapsedTime(&ms, a, b));
      hipLaunchKernelGGL(verify_kernel, dim3(N3 * N2), dim3(256), 0, 0, X, err);
      CK(hipDeviceSynchronize());
      if (rep == reps - 1) report(v.name, ms, 1.0 * nX * 16);
    }
  }
  return 0;
}
