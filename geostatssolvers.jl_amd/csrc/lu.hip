// Partial-pivot LU for LUGS' `factorization = lu` (/root/reference/src/simulation/lu.jl:70,107,128,134,139).
//
// The reference calls `lu(Symmetric(C)).L`: LAPACK getrf on the dense matrix (P C = L U, pivot = the first row of
// maximal |a| in the column) and keeps only the unit lower-triangular factor -- the row permutation and U are
// dropped, so L is NOT a square root of C; the solver reproduces what the reference computes, not what Alabert's
// method intends (LUGS).  Right-looking blocked factorisation, column-major, in place:
//   panel (LU_NB columns): one workgroup, per column an arg-max reduction, the row interchange inside the panel,
//                          the scaling by the reciprocal pivot (as LAPACK's getrf2) and the rank-1 update of the
//                          rest of the panel;
//   interchanges applied to the columns left and right of the panel (laswp);
//   U12 = inv(L11) A12: one thread per column, L11 in LDS;
//   A22 -= L21 U12 on the FP64-MFMA GEMM (dense_la.hip).
// Round 3: two levels of blocking.  Outer panels of LU_OUTER columns end in ONE trailing product with k = LU_OUTER
// (the k = 32 updates of round 1 moved the whole trailing matrix through HBM once per 32 columns); inside an outer
// panel the 32-column panels are factorised by getrf_panel_grid_kernel: sixteen workgroups (more for panels taller than 16 384 rows), one matrix row per thread
// with its 32 panel entries in registers, and ONE grid barrier per column -- every workgroup publishes its best
// candidate row before the barrier and reads the winner's row (the update's multipliers) after it; the rows c and p
// change places in their owners' registers.  Partial pivoting needs that global decision per column, so n barriers
// of ~2 us are the floor of this factorisation.  Results are those of the single-workgroup kernel (same pivot rule:
// the first row of maximal |a|), which remains the fall-back for panels taller than the grid holds.
// The option exists for parity with the reference's parameter surface; its own test runs it on 100 cells
// (test/simulation/lu.jl:72).
#include "gss_internal.h"
#include "tile16.h"   // static_for

#include <atomic>

namespace gss {

constexpr int LU_NB = 32;
constexpr int LU_NT = 1024;

// factorises the m x jb panel at P (leading dimension lda); ipiv[c] = j0 + (row of the pivot of column c)
__global__ __launch_bounds__(LU_NT) void getrf_panel_kernel(double* __restrict__ P, int64_t m, int jb, int64_t lda,
                                                            int64_t j0, int* __restrict__ ipiv, int* __restrict__ info) {
  __shared__ double s_val[LU_NT / 64];
  __shared__ long long s_row[LU_NT / 64];
  __shared__ double s_urow[LU_NB];
  __shared__ long long s_piv;
  __shared__ double s_inv;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  for (int c = 0; c < jb && c < m; ++c) {
    double* col = P + (int64_t)c * lda;
    // ---- pivot: first row of maximal |a| among rows c .. m-1 (idamax)
    double best = -1.0;
    long long brow = m;
    for (int64_t r = c + tid; r < m; r += LU_NT) {
      const double v = fabs(col[r]);
      if (v > best) {  // rows are visited in increasing order per thread: strict > keeps the first
        best = v;
        brow = r;
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const double ov = __shfl_xor(best, off);
      const long long orow = __shfl_xor(brow, off);
      if (ov > best || (ov == best && orow < brow)) {
        best = ov;
        brow = orow;
      }
    }
    if (lane == 0) {
      s_val[wave] = best;
      s_row[wave] = brow;
    }
    __syncthreads();
    if (tid == 0) {
      double b = s_val[0];
      long long br = s_row[0];
      for (int w = 1; w < LU_NT / 64; ++w)
        if (s_val[w] > b || (s_val[w] == b && s_row[w] < br)) {
          b = s_val[w];
          br = s_row[w];
        }
      if (!(b > 0.0)) {  // exactly singular (or NaN): report like LAPACK's info > 0, keep going with no interchange
        if (*info == 0) *info = (int)(j0 + c + 1);
        br = c;
      }
      s_piv = br;
      ipiv[c] = (int)(j0 + br);
    }
    __syncthreads();
    const long long p = s_piv;
    // ---- interchange rows c and p inside the panel; keep row c (columns c .. jb-1) for the update
    if (tid < jb) {
      double* a = P + (int64_t)tid * lda;
      const double vc = a[c], vp = a[p];
      if (p != c) {
        a[c] = vp;
        a[p] = vc;
      }
      s_urow[tid] = vp;
      if (tid == c) s_inv = vp != 0.0 ? 1.0 / vp : 0.0;
    }
    __syncthreads();
    const double inv = s_inv;
    // ---- scale the column below the pivot and update the rest of the panel (row-wise: coalesced over rows)
    for (int64_t r = c + 1 + tid; r < m; r += LU_NT) {
      const double l = col[r] * inv;
      col[r] = l;
      for (int c2 = c + 1; c2 < jb; ++c2) {
        double* a = P + (int64_t)c2 * lda + r;
        *a = fma(-l, s_urow[c2], *a);
      }
    }
    __syncthreads();
  }
}


// ---- the 32-column panel on a grid of workgroups ----------------------------------------------------------------
constexpr int LUG_WG = 16;        // workgroups up to 16 384 rows (a barrier of sixteen arrivals costs ~1.5 us, of sixty-four
constexpr int LUG_WG_MAX = 64;    // ~5 us): 32 / 64 workgroups for taller panels, one row per thread throughout
constexpr int LUG_NT = 1024;      // threads per workgroup
struct LuCand {                   // what a workgroup publishes per column: its best row and that row's panel entries
  double val;
  long long row;
  double a[LU_NB];
};

// One grid barrier.  bar[0] counts arrivals; its top bit says "this panel is given up".  A workgroup gives up by a
// compare-and-swap against the count it last saw, so the bit can only be set while the barrier it waits at is incomplete:
// once ANY workgroup has passed the last barrier of a panel nobody can give up any more, and once the bit is set nobody
// passes a barrier -- the verdict is the same for every workgroup of the launch (and, through LDS, for every thread of a
// workgroup).  Returns true when the panel is dead.
constexpr unsigned LU_DEAD = 0x80000000u;
__device__ __forceinline__ bool lu_grid_sync(unsigned* cnt, unsigned& epoch, int G, unsigned spin_limit, unsigned* s_dead) {
  __syncthreads();
  ++epoch;
  if (threadIdx.x == 0) {
    __threadfence();
    const unsigned target = epoch * (unsigned)G;
    unsigned v = atomicAdd(cnt, 1u) + 1u;
    unsigned spins = 0, dead = 0u;
    for (;;) {
      if (v & LU_DEAD) {
        dead = 1u;
        break;
      }
      if (v >= target) break;
      if (++spins > spin_limit) {   // the workgroups are not all resident: give up, the host repeats the factorisation
        const unsigned old = atomicCAS(cnt, v, v | LU_DEAD);
        if (old == v) {
          dead = 1u;
          break;
        }
        v = old;                    // somebody arrived (or gave up) meanwhile: look again
        continue;
      }
      __builtin_amdgcn_s_sleep(1);
      v = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __threadfence();
    *s_dead = dead;
  }
  __syncthreads();
  return *s_dead != 0u;
}

// bar[0] arrivals (+ LU_DEAD), bar[1] a panel of this factorisation was given up (sticky: the later panels of the same
// factorisation return at once, the host repeats it on the single-workgroup panels), bar[2] departures; cand: 2 x
// (LUG_WG_MAX candidates + row c).  A panel that is given up leaves P as it found it, identity interchanges in ipiv
// and *info = -1.
__global__ __launch_bounds__(LUG_NT) void getrf_panel_grid_kernel(double* __restrict__ P, int64_t m, int jb, int64_t lda,
                                                                  int64_t j0, int* __restrict__ ipiv,
                                                                  int* __restrict__ info, unsigned* __restrict__ bar,
                                                                  LuCand* __restrict__ cand, unsigned spin_limit) {
  __shared__ double s_val[LUG_NT / 64];
  __shared__ long long s_row[LUG_NT / 64];
  __shared__ double s_urow[LU_NB];
  __shared__ double s_crow[LU_NB];
  __shared__ double s_rec[(LUG_WG_MAX + 1) * (LU_NB + 2)];
  __shared__ long long s_piv;
  __shared__ unsigned s_dead;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int RPT = 1;
  const int wg = blockIdx.x;
  const int G = gridDim.x;
  unsigned epoch = 0;
  // (bar[1] was written by an earlier launch of this stream: every workgroup of this launch reads the same value)
  if (__hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
    if (wg == 0 && tid < jb) ipiv[tid] = (int)(j0 + tid);
    return;
  }
  // rows of this thread: r = (q * LUG_WG + wg) * LUG_NT + tid  (consecutive threads, consecutive rows: coalesced)
  double a[RPT][LU_NB];
  int64_t row[RPT];
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    row[q] = ((int64_t)q * G + wg) * LUG_NT + tid;
#pragma unroll
    for (int c = 0; c < LU_NB; ++c) a[q][c] = (row[q] < m && c < jb) ? P[row[q] + (int64_t)c * lda] : 0.0;
  }
  bool dead = false;
  // (the column loop is unrolled: the panel entries live in registers, whose indices must be constants)
  static_for<0, LU_NB>([&](auto CC) {
    constexpr int c = decltype(CC)::value;
    if (c < jb && c < m && !dead) {
      LuCand* slot = cand + (size_t)(c & 1) * (LUG_WG_MAX + 1);
      // ---- this workgroup's candidate: first row of maximal |a(., c)| among its rows >= c
      double best = -1.0;
      long long brow = m;
#pragma unroll
      for (int q = 0; q < RPT; ++q) {
        const double v = fabs(a[q][c]);
        if (row[q] >= c && row[q] < m && v > best) {   // q ascending = rows ascending: strict > keeps the first
          best = v;
          brow = row[q];
        }
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const double ov = __shfl_xor(best, off);
        const long long orow = __shfl_xor(brow, off);
        if (ov > best || (ov == best && orow < brow)) {
          best = ov;
          brow = orow;
        }
      }
      if (lane == 0) {
        s_val[wave] = best;
        s_row[wave] = brow;
      }
      __syncthreads();
      if (wave == 0) {   // the sixteen per-wave candidates, one per lane, reduced across lanes (not a serial scan by one thread)
        constexpr int NW = LUG_NT / 64;
        double b = lane < NW ? s_val[lane] : -2.0;
        long long br = lane < NW ? s_row[lane] : m;
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) {
          const double ov = __shfl_xor(b, off);
          const long long orow = __shfl_xor(br, off);
          if (ov > b || (ov == b && orow < br)) {
            b = ov;
            br = orow;
          }
        }
        if (lane == 0) {
          s_piv = br;
          // (write-through stores: nothing dirty is left in this XCD's L2 for the barrier's release to write back)
          __hip_atomic_store(&slot[wg].val, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&slot[wg].row, br, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      __syncthreads();
      {
        const long long mine = s_piv;
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
          if (row[q] == mine) {
#pragma unroll
            for (int c2 = 0; c2 < LU_NB; ++c2)
              __hip_atomic_store(&slot[wg].a[c2], a[q][c2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if (row[q] == c) {   // the owner of row c publishes it too: the pivot row's owner takes it over
#pragma unroll
            for (int c2 = 0; c2 < LU_NB; ++c2)
              __hip_atomic_store(&slot[LUG_WG_MAX].a[c2], a[q][c2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
      dead = lu_grid_sync(bar, epoch, G, spin_limit, &s_dead);
    }
    // (a dead panel decides nothing: its peers' records may never have been written)
    if (c < jb && c < m && !dead) {
      LuCand* slot = cand + (size_t)(c & 1) * (LUG_WG_MAX + 1);
      // ---- the decision, the same on every workgroup: best candidate, ties to the smaller row.  Every candidate
      // record and row c come into LDS with ONE round trip to memory (the records were written by other XCDs:
      // agent-scope loads), the choice is made there
      {
        constexpr int RECW = LU_NB + 2;   // doubles per record
        const int nel = (G + 1) * RECW;
        for (int e = tid; e < nel; e += LUG_NT) {
          const int rI = e / RECW, f = e - rI * RECW;
          const LuCand* rec = rI < G ? &slot[rI] : &slot[LUG_WG_MAX];
          const double* src = reinterpret_cast<const double*>(rec) + f;   // {val, row, a[32]}: 34 x 8 bytes
          s_rec[rI * RECW + f] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      __syncthreads();
      if (wave == 0) {
        constexpr int RECW = LU_NB + 2;
        double b = lane < G ? s_rec[lane * RECW] : -2.0;
        long long br = lane < G ? __builtin_bit_cast(long long, s_rec[lane * RECW + 1]) : m;
        int w = lane;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
          const double ov = __shfl_xor(b, off);
          const long long orow = __shfl_xor(br, off);
          const int ow = __shfl_xor(w, off);
          if (ov > b || (ov == b && orow < br)) {
            b = ov;
            br = orow;
            w = ow;
          }
        }
        b = __shfl(b, 0);
        br = __shfl(br, 0);
        w = __shfl(w, 0);
        const bool singular = !(b > 0.0);   // exactly singular (or NaN): LAPACK's info > 0, no interchange
        if (singular) br = c;
        if (lane < LU_NB) {
          s_crow[lane] = s_rec[G * RECW + 2 + lane];
          s_urow[lane] = singular ? s_crow[lane] : s_rec[w * RECW + 2 + lane];
        }
        if (lane == 0) {
          s_piv = br;
          if (wg == 0) {
            ipiv[c] = (int)(j0 + br);
            if (singular && *info == 0) *info = (int)(j0 + c + 1);
          }
        }
      }
      __syncthreads();
      const long long p = s_piv;
      const double piv = s_urow[c];
      const double inv = piv != 0.0 ? 1.0 / piv : 0.0;
#pragma unroll
      for (int q = 0; q < RPT; ++q) {
        // interchange in the owners' registers: row p takes the old row c, row c takes the pivot row
        if (p != c && row[q] == p) {
#pragma unroll
          for (int c2 = 0; c2 < LU_NB; ++c2) a[q][c2] = s_crow[c2];
        }
        if (row[q] == c) {
#pragma unroll
          for (int c2 = 0; c2 < LU_NB; ++c2) a[q][c2] = s_urow[c2];
        }
        if (row[q] > c && row[q] < m) {   // scale by the reciprocal pivot (as getrf2) and update the rest of the panel
          const double l = a[q][c] * inv;
          a[q][c] = l;
#pragma unroll
          for (int c2 = c + 1; c2 < LU_NB; ++c2) a[q][c2] = fma(-l, s_urow[c2], a[q][c2]);
        }
      }
      // (no barrier: s_urow / s_crow are rewritten by the next column's decision, which every wave reaches only through
      //  that column's barriers)
    }
  });
  if (!dead) {
#pragma unroll
    for (int q = 0; q < RPT; ++q)
#pragma unroll
      for (int c = 0; c < LU_NB; ++c)
        if (row[q] < m && c < jb) P[row[q] + (int64_t)c * lda] = a[q][c];
  } else if (wg == 0 && tid < jb) {
    ipiv[tid] = (int)(j0 + tid);   // whatever the columns before the give-up chose: P was not touched
  }
  // the last workgroup out resets the barrier words for the next panel on this stream
  __syncthreads();
  if (tid == 0) {
    __threadfence();
    if (atomicAdd(&bar[2], 1u) == (unsigned)G - 1u) {
      if (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & LU_DEAD) {
        atomicExch(info, -1);      // unconditionally: a "singular" verdict of this factorisation means nothing now
        __hip_atomic_store(&bar[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __hip_atomic_store(&bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&bar[2], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// apply the jb interchanges of a panel to `ncols` other columns starting at A0 (row indices are global)
__global__ __launch_bounds__(256) void laswp_kernel(double* __restrict__ A0, int64_t ncols, int64_t lda, int64_t j0,
                                                    int jb, const int* __restrict__ ipiv) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= ncols) return;
  double* a = A0 + j * lda;
  for (int c = 0; c < jb; ++c) {
    const int64_t r = j0 + c, p = ipiv[c];
    if (p != r) {
      const double t = a[r];
      a[r] = a[p];
      a[p] = t;
    }
  }
}

// B <- inv(L11) B, L11 unit lower triangular jb x jb (strict lower of the panel's top block), B is jb x ncols
__global__ __launch_bounds__(256) void trsm_unit_lower_kernel(const double* __restrict__ L11, int jb, int64_t lda,
                                                              double* __restrict__ B, int64_t ncols) {
  __shared__ double sL[LU_NB * LU_NB];
  for (int e = threadIdx.x; e < jb * jb; e += 256) {
    const int i = e % jb, k = e / jb;
    sL[i * LU_NB + k] = L11[i + (int64_t)k * lda];
  }
  __syncthreads();
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= ncols) return;
  double* b = B + j * lda;
  double x[LU_NB];
#pragma unroll
  for (int i = 0; i < LU_NB; ++i) x[i] = i < jb ? b[i] : 0.0;
#pragma unroll
  for (int i = 1; i < LU_NB; ++i) {
    if (i < jb) {
      double acc = x[i];
#pragma unroll
      for (int k = 0; k < LU_NB; ++k)
        if (k < i) acc = fma(-sL[i * LU_NB + k], x[k], acc);
      x[i] = acc;
    }
  }
#pragma unroll
  for (int i = 0; i < LU_NB; ++i)
    if (i < jb) b[i] = x[i];
}

// A <- unit lower triangular factor: zero above the diagonal, ones on it (`.L` of the LU object)
__global__ __launch_bounds__(256) void unit_lower_kernel(double* __restrict__ A, int64_t n, int64_t lda) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t j = blockIdx.y;
  if (i < n && i <= j) A[i + j * lda] = (i == j) ? 1.0 : 0.0;
}

// upper triangle <- transpose of the lower triangle (the SYRK-shaped update writes lower tiles only)
__global__ __launch_bounds__(256) void mirror_lower_kernel(double* __restrict__ A, int64_t n, int64_t lda) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // row, i > j
  const int64_t j = blockIdx.y;
  if (i < n && i > j) A[j + i * lda] = A[i + j * lda];
}

int32_t mirror_lower_f64(double* A, int64_t n, int64_t lda, hipStream_t s) {
  if (n <= 1) return GSS_OK;
  hipLaunchKernelGGL(mirror_lower_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)n), dim3(256), 0, s, A, n, lda);
  GSS_HIP(hipGetLastError());
  return GSS_OK;
}

// A (n x n, column-major, full) <- L of P A = L U as a dense unit lower-triangular matrix.  *d_info (device int,
// zeroed by the caller) receives 1 + column of the first exactly-zero pivot.  ipiv: device scratch of n ints.
constexpr int LU_OUTER = 512;
// set when a grid panel reported that its workgroups did not gather (*d_info = -1): single-workgroup panels from then on
static std::atomic<bool> g_lu_grid_off{false};
void lu_grid_disable() { g_lu_grid_off.store(true); }

// one 32-column panel (m rows at P): on the grid when its rows fit, else -- or when the grid once failed to gather --
// on the single workgroup
static int32_t lu_panel(double* P, int64_t m, int jb, int64_t lda, int64_t j0, int* ipiv, int* d_info, unsigned* bar,
                        LuCand* cand, bool* grid_ok, hipStream_t s) {
  if (*grid_ok && m > 1024 && m <= (int64_t)LUG_WG_MAX * LUG_NT) {
    const int G = m <= (int64_t)LUG_WG * LUG_NT ? LUG_WG : (m <= (int64_t)32 * LUG_NT ? 32 : LUG_WG_MAX);
    // test hook: GSS_LU_PANEL_FAIL=1 gives the first grid panel of the process no patience at its barriers, so that its
    // workgroups really take the give-up path (tests/test_gpu_dense.py)
    static std::atomic<bool> fail_once{std::getenv("GSS_LU_PANEL_FAIL") != nullptr};
    const unsigned spin_limit = fail_once.exchange(false) ? 0u : 4000000u;
    hipLaunchKernelGGL(getrf_panel_grid_kernel, dim3((unsigned)G), dim3(LUG_NT), 0, s, P, m, jb, lda, j0, ipiv, d_info, bar,
                       cand, spin_limit);
  } else {
    hipLaunchKernelGGL(getrf_panel_kernel, dim3(1), dim3(LU_NT), 0, s, P, m, jb, lda, j0, ipiv, d_info);
  }
  GSS_HIP(hipGetLastError());
  return GSS_OK;
}

int32_t getrf_unit_lower_f64(double* A, int64_t n, int64_t lda, int* ipiv, int* d_info, hipStream_t s) {
  DevBuf barbuf, candbuf;
  GSS_TRY(barbuf.alloc(64));
  GSS_TRY(candbuf.alloc(sizeof(LuCand) * 2 * (LUG_WG_MAX + 1)));
  GSS_TRY(dev_zero_bytes(barbuf.p, 64, s));
  GSS_TRY(dev_zero_bytes(candbuf.p, sizeof(LuCand) * 2 * (LUG_WG_MAX + 1), s));
  unsigned* bar = barbuf.as<unsigned>();
  LuCand* cand = candbuf.as<LuCand>();
  bool grid_ok = !g_lu_grid_off.load();
  for (int64_t o0 = 0; o0 < n; o0 += LU_OUTER) {
    const int ob = (int)((n - o0) < LU_OUTER ? (n - o0) : LU_OUTER);   // columns of this outer panel
    // ---- the outer panel, by 32-column panels; everything is confined to its ob columns
    for (int64_t j0 = o0; j0 < o0 + ob; j0 += LU_NB) {
      const int jb = (int)((o0 + ob - j0) < LU_NB ? (o0 + ob - j0) : LU_NB);
      const int64_t m = n - j0;
      double* P = A + j0 + j0 * lda;
      GSS_TRY(lu_panel(P, m, jb, lda, j0, ipiv + j0, d_info, bar, cand, &grid_ok, s));
      const int64_t nl = j0 - o0;                 // columns of the outer panel left of this panel
      const int64_t nr = o0 + ob - j0 - jb;       // ... and right of it
      if (nl > 0)
        hipLaunchKernelGGL(laswp_kernel, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, s, A + o0 * lda, nl, lda, j0, jb,
                           ipiv + j0);
      if (nr > 0) {
        double* A12 = A + (j0 + jb) * lda;
        hipLaunchKernelGGL(laswp_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, s, A12, nr, lda, j0, jb, ipiv + j0);
        hipLaunchKernelGGL(trsm_unit_lower_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, s, P, jb, lda,
                           A12 + j0, nr);
        GSS_HIP(hipGetLastError());
        if (m - jb > 0)
          GSS_TRY(gemm_f64(m - jb, nr, jb, -1.0, P + jb, 1, lda, A12 + j0, 1, lda, 1.0, A12 + j0 + jb, 1, lda, false, s));
      }
    }
    // ---- the outer panel's interchanges on the columns left and right of it
    if (o0 > 0)
      hipLaunchKernelGGL(laswp_kernel, dim3((unsigned)((o0 + 255) / 256)), dim3(256), 0, s, A, o0, lda, o0, ob, ipiv + o0);
    const int64_t nright = n - o0 - ob;
    if (nright > 0) {
      double* A12 = A + (o0 + ob) * lda;          // column o0 + ob, row 0
      hipLaunchKernelGGL(laswp_kernel, dim3((unsigned)((nright + 255) / 256)), dim3(256), 0, s, A12, nright, lda, o0, ob,
                         ipiv + o0);
      // U12 = inv(L11) A12 by block rows of 32: rows b of A12 first lose what the block rows above contribute
      for (int64_t b0 = 0; b0 < ob; b0 += LU_NB) {
        const int bb = (int)((ob - b0) < LU_NB ? (ob - b0) : LU_NB);
        double* Lrow = A + (o0 + b0) + o0 * lda;  // rows o0 + b0 .., columns o0 .. of L11
        if (b0 > 0)
          GSS_TRY(gemm_f64(bb, nright, b0, -1.0, Lrow, 1, lda, A12 + o0, 1, lda, 1.0, A12 + o0 + b0, 1, lda, false, s));
        hipLaunchKernelGGL(trsm_unit_lower_kernel, dim3((unsigned)((nright + 255) / 256)), dim3(256), 0, s,
                           A + (o0 + b0) + (o0 + b0) * lda, bb, lda, A12 + o0 + b0, nright);
      }
      GSS_HIP(hipGetLastError());
      // A22 -= L21 U12, k = ob
      GSS_TRY(gemm_f64(nright, nright, ob, -1.0, A + (o0 + ob) + o0 * lda, 1, lda, A12 + o0, 1, lda, 1.0,
                       A12 + o0 + ob, 1, lda, false, s));
    }
  }
  // (a grid panel whose workgroups did not all arrive reports *d_info = -1 and the later grid panels of this call return
  // at once: A is then neither the input nor a factor.  The caller calls lu_grid_disable() and repeats the factorisation
  // from the original matrix, as for the single-launch Cholesky panel)
  hipLaunchKernelGGL(unit_lower_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)n), dim3(256), 0, s, A, n, lda);
  GSS_HIP(hipGetLastError());
  return GSS_OK;
}

}  // namespace gss

using namespace gss;

extern "C" {

int32_t gss_dev_getrf_l(double* a, int64_t n, int64_t lda, void* stream) {
  GSS_ENTRY();
  GSS_REQUIRE(a != nullptr && n >= 1 && lda >= n, "gss_dev_getrf_l: bad arguments");
  hipStream_t s = to_stream(stream);
  DevBuf info, ipiv;
  GSS_TRY(info.alloc(sizeof(int)));
  GSS_TRY(ipiv.alloc(sizeof(int) * (size_t)n));
  GSS_TRY(dev_zero_bytes(info.p, sizeof(int), s));
  GSS_TRY(getrf_unit_lower_f64(a, n, lda, ipiv.as<int>(), info.as<int>(), s));
  int h = 0;
  GSS_HIP(hipMemcpyAsync(&h, info.p, sizeof(int), hipMemcpyDeviceToHost, s));
  GSS_HIP(hipStreamSynchronize(s));
  if (h < 0) {
    lu_grid_disable();
    set_error("LU factorisation: the grid panel gave up waiting at a barrier and is switched off for this process; the "
              "matrix was overwritten in place -- restore the input, then call again");
    return GSS_ERR_HIP;
  }
  if (h != 0) {
    set_error("LU factorisation: exactly singular matrix (zero pivot in column %d)", h - 1);
    return GSS_ERR_NOT_POSDEF;
  }
  return GSS_OK;
}

}  // extern "C"
