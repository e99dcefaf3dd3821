// SGS: sequential Gaussian simulation.
// Replaces SGS.preprocess (/root/reference/src/simulation/sgs.jl:56-85) and the SeqSim path loop
// (/root/reference/src/simulation/seq.jl:76-141) for the estimator/marginal pair SGS always builds
// (SimpleKriging(variogram, mean), Normal(mean, sqrt(sill)), sgs.jl:62-67).
//
// The reference walks the path once per realisation and, at every node, searches the already simulated
// cells (seq.jl:105), fits a simple-kriging system (seq.jl:121) and draws from Normal(mu, sigma) (seq.jl:129).
// Which cells are "already simulated" depends on the path and on the data cells only, never on the drawn
// values, and simple-kriging weights do not depend on the values either.  So the device splits the work:
//
//   stage A (gss_sgs_create, data parallel over the N nodes, done once for every realisation):
//     masked exact k-NN (rank[neighbour] < rank[node]; knn.hip)  -> neighbours of every node
//     one wave per node: k x k covariance, Cholesky, lambda = C^-1 c0, sigma^2 = sill - |L^-1 c0|^2
//     fewer than minneighbors neighbours, or a failed factorisation (`status(fitted)`, seq.jl:124-128)
//     -> marginal: no weights, sigma = sqrt(sill)
//   stage B (gss_sgs_realize, parallel over realisations, sequential along the path):
//     z[node] = mean + sum_j lambda_j (z[nb_j] - mean) + sigma * eps(seed, realisation, node)
//     one lane per realisation; the field is kept node-major [N][R] so that the gathers coalesce.
#include "gss_internal.h"
#include "philox.h"

#include <hipcub/hipcub.hpp>

#include <climits>
#include <cmath>
#include <cstring>
#include <vector>

namespace gss {

constexpr int SGS_MAX_K = 64;

__device__ __forceinline__ int sgs_tri(int i) { return (i * (i + 1)) >> 1; }

__device__ __forceinline__ double sgs_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// lowest visiting rank inside each batch of the search index
__global__ __launch_bounds__(256) void sgs_batch_minrank_kernel(const int* __restrict__ perm,
                                                                const int* __restrict__ rank, int n, int nb,
                                                                int* __restrict__ bmin) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (b >= nb) return;
  const int j = b * 64 + lane;
  int r = j < n ? rank[perm[j]] : INT_MAX;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const int o = __shfl_xor(r, off);
    r = o < r ? o : r;
  }
  if (lane == 0) bmin[b] = r;
}

// stage A: simple-kriging weights of one node per wave
template <int DIM>
__global__ __launch_bounds__(64) void sgs_weights_kernel(VgDev vg, const double* __restrict__ cent,
                                                         const int* __restrict__ rank, int64_t N, int k,
                                                         int minneighbors, const int* __restrict__ idx,
                                                         const int* __restrict__ count, int* __restrict__ ncond,
                                                         double* __restrict__ w_out, double* __restrict__ sigma_out,
                                                         int* __restrict__ idx_rw, int filter_after) {
  __shared__ double Lp[SGS_MAX_K * (SGS_MAX_K + 1) / 2];
  __shared__ double nx[SGS_MAX_K][3];
  __shared__ int cidx[SGS_MAX_K];
  const int64_t p = blockIdx.x;
  const int lane = threadIdx.x;
  if (rank[p] < 0) {  // data cell: never simulated (seq.jl:103)
    if (lane == 0) {
      ncond[p] = 0;
      sigma_out[p] = 0.0;
    }
    return;
  }
  int cnt = count[p];
  if (filter_after) {
    // the search was NOT masked: of the k nearest cells (the node itself included) keep, in order, those already
    // simulated when the node is visited (data cells have rank -1) -- `mask=simulated` applied after the search
    const bool in = lane < cnt;
    const int nb = in ? idx[p * k + lane] : 0;
    const bool keep = in && rank[nb] < rank[p];
    const unsigned long long bm = __ballot(keep);
    if (keep) cidx[__popcll(bm & ((1ull << lane) - 1ull))] = nb;
    __syncthreads();
    cnt = __popcll(bm);
    if (lane < k) idx_rw[p * k + lane] = lane < cnt ? cidx[lane] : -1;
    __syncthreads();
  }
  const double smarg = sqrt(vg.sill);  // sgs.jl:66
  if (cnt < minneighbors || cnt <= 0) {  // seq.jl:107-109
    if (lane == 0) {
      ncond[p] = 0;
      sigma_out[p] = smarg;
    }
    return;
  }
  double c0[DIM], xj[DIM];
#pragma unroll
  for (int a = 0; a < DIM; ++a) c0[a] = cent[p * DIM + a];
  const bool act = lane < cnt;
  const int nj = act ? (filter_after ? cidx[lane] : idx[p * k + lane]) : 0;
#pragma unroll
  for (int a = 0; a < DIM; ++a) {
    xj[a] = act ? cent[(int64_t)nj * DIM + a] : 0.0;
    nx[lane][a] = xj[a];
  }
  double b = act ? cov_pair<DIM>(vg, xj, c0) : 0.0;
  __syncthreads();
  const int nent = sgs_tri(cnt);
  for (int e = lane; e < nent; e += 64) {
    int i = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
    while (sgs_tri(i + 1) <= e) ++i;
    while (sgs_tri(i) > e) --i;
    const int c = e - sgs_tri(i);
    double xi[DIM], xc[DIM];
#pragma unroll
    for (int a = 0; a < DIM; ++a) {
      xi[a] = nx[i][a];
      xc[a] = nx[c][a];
    }
    Lp[e] = cov_pair<DIM>(vg, xi, xc);
  }
  __syncthreads();
  // Cholesky, left-looking by columns: lane i owns row i
  bool bad = false;
  const double* rowi = Lp + sgs_tri(lane < cnt ? lane : 0);
  for (int j = 0; j < cnt; ++j) {
    double acc = 0.0;
    const bool mine = lane >= j && lane < cnt;
    if (mine) {
      const double* rowj = Lp + sgs_tri(j);
      acc = rowi[j];
      for (int c = 0; c < j; ++c) acc = fma(-rowi[c], rowj[c], acc);
    }
    const double d = __shfl(acc, j);
    if (!(d > 0.0)) {
      bad = true;
      break;
    }
    const double sq = sqrt(d);
    if (mine) Lp[sgs_tri(lane) + j] = (lane == j) ? sq : acc / sq;
    __syncthreads();
  }
  if (bad) {  // status(fitted) == false -> marginal (seq.jl:124-128)
    if (lane == 0) {
      ncond[p] = 0;
      sigma_out[p] = smarg;
    }
    return;
  }
  // y = L^-1 c0 (column sweep)
  for (int j = 0; j < cnt; ++j) {
    const double ljj = Lp[sgs_tri(j) + j];
    const double lij = (lane > j && lane < cnt) ? rowi[j] : 0.0;
    const double yj = __shfl(b, j) / ljj;
    if (lane == j) b = yj;
    else b = fma(-lij, yj, b);
  }
  const double q = sgs_wave_sum(act ? b * b : 0.0);
  // lambda = L^-T y (row j of L is contiguous: lane i < j reads L[j][i])
  for (int j = cnt - 1; j >= 0; --j) {
    const double ljj = Lp[sgs_tri(j) + j];
    const double lj = __shfl(b, j) / ljj;
    if (lane == j) b = lj;
    else if (lane < j) b = fma(-Lp[sgs_tri(j) + lane], lj, b);
  }
  if (act) w_out[p * k + lane] = b;
  if (lane == 0) {
    const double v = vg.sill - q;
    ncond[p] = cnt;
    sigma_out[p] = sqrt(v > 0.0 ? v : 0.0);
  }
}

// ---------------------------------------------------------------------------------------------
// More than 64 neighbours (seq.jl:91-98 accepts any maxneighbors; available with the search-then-filter reading of
// `mask = simulated`, whose search is unmasked and runs in passes of 64, knn.hip).  One workgroup of 256 threads per
// node: the kept neighbours in order (wave 0: ballot compaction by 64s), the packed covariance triangle in LDS while
// it fits (up to 180 neighbours) or in a slab of HBM, Cholesky by columns, forward and back substitution.  A
// functional path like krig_local_big_kernel: two barriers per column.
// ---------------------------------------------------------------------------------------------
constexpr int SGS_BIG_NT = 256;
constexpr int SGS_BIG_LDS_DOUBLES = 16384 + 256;   // packed triangle + the right-hand side of <= 180 neighbours
constexpr int SGS_BIG_MAX_K = 1024;

template <int DIM>
__global__ __launch_bounds__(SGS_BIG_NT) void sgs_weights_big_kernel(
    VgDev vg, const double* __restrict__ cent, const int* __restrict__ rank, int64_t N, int k, int minneighbors,
    const int* __restrict__ idx, const int* __restrict__ count, int* __restrict__ ncond, double* __restrict__ w_out,
    double* __restrict__ sigma_out, int* __restrict__ idx_rw, double* __restrict__ scratch, int64_t slab, int use_lds) {
  extern __shared__ double sgs_big_sm[];
  __shared__ int s_cnt, s_bad;
  __shared__ double s_red[SGS_BIG_NT / 64];
  const int tid = threadIdx.x;
  const double smarg = sqrt(vg.sill);  // sgs.jl:66
  for (int64_t p = blockIdx.x; p < N; p += gridDim.x) {
    __syncthreads();  // the previous node's shared words have been read
    if (rank[p] < 0) {  // data cell: never simulated (seq.jl:103)
      if (tid == 0) {
        ncond[p] = 0;
        sigma_out[p] = 0.0;
      }
      continue;
    }
    if (tid < 64) {  // keep, in order, the neighbours already simulated when the node is visited
      const int cnt0 = count[p];
      const int rp = rank[p];
      int off = 0;
      for (int base = 0; base < cnt0; base += 64) {
        const int j = base + tid;
        const bool in = j < cnt0;
        const int nb = in ? idx[p * k + j] : 0;
        const bool keep = in && rank[nb] < rp;
        const unsigned long long bm = __ballot(keep);
        if (keep) idx_rw[p * k + off + __popcll(bm & ((1ull << tid) - 1ull))] = nb;
        off += __popcll(bm);
      }
      for (int j = off + tid; j < k; j += 64) idx_rw[p * k + j] = -1;
      if (tid == 0) {
        s_cnt = off;
        s_bad = 0;
      }
    }
    __syncthreads();
    const int c = s_cnt;
    if (c < minneighbors || c <= 0) {  // seq.jl:107-109
      if (tid == 0) {
        ncond[p] = 0;
        sigma_out[p] = smarg;
      }
      continue;
    }
    const int* nb = idx_rw + p * k;
    double* M = use_lds ? sgs_big_sm : scratch + (int64_t)blockIdx.x * slab;
    const int ntri = c * (c + 1) / 2;
    double* b = M + ntri;
    double c0[DIM];
#pragma unroll
    for (int a = 0; a < DIM; ++a) c0[a] = cent[p * DIM + a];
    for (int e = tid; e < ntri; e += SGS_BIG_NT) {
      int i = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
      while (sgs_tri(i + 1) <= e) ++i;
      while (sgs_tri(i) > e) --i;
      const int j = e - sgs_tri(i);
      const int ni = nb[i], nj = nb[j];
      double xi[DIM], xj[DIM];
#pragma unroll
      for (int a = 0; a < DIM; ++a) {
        xi[a] = cent[(int64_t)ni * DIM + a];
        xj[a] = cent[(int64_t)nj * DIM + a];
      }
      M[e] = cov_pair<DIM>(vg, xi, xj);
    }
    for (int j = tid; j < c; j += SGS_BIG_NT) {
      double xj[DIM];
#pragma unroll
      for (int a = 0; a < DIM; ++a) xj[a] = cent[(int64_t)nb[j] * DIM + a];
      b[j] = cov_pair<DIM>(vg, xj, c0);
    }
    __syncthreads();
    // Cholesky, left-looking by columns; rows are dealt to the threads
    for (int j = 0; j < c; ++j) {
      const double* rowj = M + sgs_tri(j);
      for (int i = j + tid; i < c; i += SGS_BIG_NT) {
        double* rowi = M + sgs_tri(i);
        double acc = rowi[j];
        for (int q = 0; q < j; ++q) acc = fma(-rowi[q], rowj[q], acc);
        rowi[j] = acc;
      }
      __syncthreads();
      const double d = rowj[j];
      if (!(d > 0.0)) {
        if (tid == 0) s_bad = 1;
        break;   // d is read by every thread alike: the whole workgroup leaves the loop
      }
      __syncthreads();
      const double sq = sqrt(d);
      for (int i = j + tid; i < c; i += SGS_BIG_NT) {
        double* rowi = M + sgs_tri(i);
        rowi[j] = (i == j) ? sq : rowi[j] / sq;
      }
      __syncthreads();
    }
    __syncthreads();
    if (s_bad) {  // status(fitted) == false -> marginal (seq.jl:124-128)
      if (tid == 0) {
        ncond[p] = 0;
        sigma_out[p] = smarg;
      }
      continue;
    }
    // y = L^-1 c0 by columns
    for (int j = 0; j < c; ++j) {
      const double yj = b[j] / M[sgs_tri(j) + j];
      __syncthreads();
      if (tid == 0) b[j] = yj;
      for (int i = j + 1 + tid; i < c; i += SGS_BIG_NT) b[i] = fma(-M[sgs_tri(i) + j], yj, b[i]);
      __syncthreads();
    }
    double q = 0.0;
    for (int j = tid; j < c; j += SGS_BIG_NT) q = fma(b[j], b[j], q);
    q = sgs_wave_sum(q);
    if ((tid & 63) == 0) s_red[tid >> 6] = q;
    // lambda = L^-T y (row j of L is contiguous)
    for (int j = c - 1; j >= 0; --j) {
      const double lj = b[j] / M[sgs_tri(j) + j];
      __syncthreads();
      if (tid == 0) b[j] = lj;
      const double* rowj = M + sgs_tri(j);
      for (int i = tid; i < j; i += SGS_BIG_NT) b[i] = fma(-rowj[i], lj, b[i]);
      __syncthreads();
    }
    for (int j = tid; j < c; j += SGS_BIG_NT) w_out[p * k + j] = b[j];
    if (tid == 0) {
      double qq = 0.0;
      for (int wv = 0; wv < SGS_BIG_NT / 64; ++wv) qq += s_red[wv];
      const double v = vg.sill - qq;
      ncond[p] = c;
      sigma_out[p] = sqrt(v > 0.0 ? v : 0.0);
    }
  }
}

// stage B for more than 64 neighbours, shared visiting order: the same recursion without the lane-held lists
__global__ __launch_bounds__(64) void sgs_sweep_big_kernel(const int64_t* __restrict__ path, const int* __restrict__ rank,
                                                           const int* __restrict__ idx, const int* __restrict__ ncond,
                                                           const double* __restrict__ w, const double* __restrict__ sigma,
                                                           int k, int64_t N, int R, double mean, double* __restrict__ zt) {
  const int r = blockIdx.x * 64 + threadIdx.x;
  const bool live = r < R;
  const int rr = live ? r : R - 1;
  for (int64_t t = 0; t < N; ++t) {
    const int64_t node = path[t];
    if (rank[node] < 0) continue;  // conditioning cell
    const int c = ncond[node];
    const int* nb = idx + node * k;
    const double* ww = w + node * k;
    double acc = 0.0;
    int j = 0;
    for (; j + 4 <= c; j += 4) {   // four gathers in flight
      const double z0 = zt[(int64_t)nb[j] * R + rr], z1 = zt[(int64_t)nb[j + 1] * R + rr];
      const double z2 = zt[(int64_t)nb[j + 2] * R + rr], z3 = zt[(int64_t)nb[j + 3] * R + rr];
      acc = fma(ww[j], z0 - mean, acc);
      acc = fma(ww[j + 1], z1 - mean, acc);
      acc = fma(ww[j + 2], z2 - mean, acc);
      acc = fma(ww[j + 3], z3 - mean, acc);
    }
    for (; j < c; ++j) acc = fma(ww[j], zt[(int64_t)nb[j] * R + rr] - mean, acc);
    const double v = mean + acc + sigma[node] * zt[node * R + rr];
    if (live) zt[node * R + r] = v;
  }
}

// ---- level schedule of the path recursion (shared visiting order) ----------------------------------------------
// z[node] only needs the values of the node's own neighbours, all visited earlier: the recursion is a sparse
// triangular solve, and its dependency graph is shallow -- a node's level is 1 + the highest level among its
// neighbours (conditioning cells: level 0); a random path over 512 x 512 cells with 16 neighbours has about 250
// levels of about 1 000 nodes.  Nodes of one level are independent, so stage B runs level by level with one wave per
// (node, 64 realisations) instead of one wave per 64 realisations walking all N nodes: the same sums in the same
// order (bit-identical fields), N k gathers per realisation spread over the whole device.
__global__ __launch_bounds__(256) void sgs_level_init_kernel(const int* __restrict__ rank, int64_t N /* all paths */,
                                                             int* __restrict__ level, int* __restrict__ node) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p < N) {
    level[p] = rank[p] < 0 ? 0 : -1;   // -1: not known yet
    node[p] = (int)p;
  }
}

// Levels in one launch: thread t owns the t-th node of the path and polls its neighbours' levels until all of them
// are known (-1 = not yet), then publishes its own.  A node only waits for nodes visited earlier, i.e. for threads
// of its own or of earlier workgroups, and workgroups start in index order: the earliest unfinished workgroup never
// waits for one that is not resident.  The store sits INSIDE the polling loop -- lanes of one wave wait for each
// other (consecutive nodes of a path are neighbours), and a lane parked behind the loop could not publish.
__global__ __launch_bounds__(256) void sgs_level_chain_kernel(const int64_t* __restrict__ path,
                                                              const int* __restrict__ rank, const int* __restrict__ idx,
                                                              const int* __restrict__ ncond, int k, int64_t N,
                                                              int64_t npaths, int* level, int* __restrict__ gave_up) {
  // Consecutive nodes of an order are neighbours of each other, and a level that a lane of the same wave has just
  // found is taken from its register instead of waiting for it to come back from memory: up to SC of a thread's
  // neighbours (the first ones of its list that lanes of its own wave own -- the list is sorted by distance, so these
  // are the nodes just before it in a row) are followed that way, the others are polled.
  constexpr int SC = 6;
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;   // paths one after the other, N nodes each
  const bool inside = g < N * npaths;
  const int64_t gg = inside ? g : N * npaths - 1;
  const int64_t base = gg / N * N;                   // first entry of this thread's path in the per-path arrays
  const int64_t p = base + path[gg];                 // (path, node) as one index
  const bool mine = inside && rank[p] >= 0;          // not a conditioning cell (level 0, sgs_level_init_kernel)
  const int c = mine ? ncond[p] : 0;
  const int* nb = idx + p * k;                       // node numbers within the path
  level += base;
  const int lane = threadIdx.x & 63;
  int so[SC], sj[SC];                                // owner lane and list position of the followed neighbours
  unsigned long long intra = 0;                      // their positions as a mask (all within the first 64)
#pragma unroll
  for (int u = 0; u < SC; ++u) so[u] = sj[u] = -1;
  {
    const int cl = c < 64 ? c : 64;
    int nf = 0;
    for (int j = 0; j < cl && nf < SC; ++j) {
      const int rj = rank[base + nb[j]];
      const int64_t og = base + rj;                  // thread that owns the neighbour: its rank in the order
      if (rj >= 0 && (og >> 6) == (gg >> 6)) {
#pragma unroll
        for (int u = 0; u < SC; ++u)
          if (u == nf) {
            so[u] = (int)(og & 63);
            sj[u] = j;
          }
        intra |= 1ull << j;
        ++nf;
      }
    }
  }
  // neighbours are polled 64 at a time and a neighbour whose level is known is never asked again: with hundreds of
  // thousands of threads in flight, polling everything on every round makes the polls the bottleneck
  // The loop is left by the whole wave at once (__all): a lane that has published must not be parked behind the
  // loop's exit while lanes of its own wave still wait for what it published -- and a store on a path that leaves
  // the loop is, for the compiler, a store after the loop.
  bool done = !mine;
  int mylevel = mine ? -1 : 0;                       // (a conditioning cell's level is 0; lanes outside are never asked)
  int polls = 0, mx = 0, j0 = 0;
  int cc = c < 64 ? c : 64;
  unsigned long long pend = cc == 64 ? ~0ull : ((1ull << cc) - 1ull);
  auto settle = [&]() {   // pend == 0: next 64 neighbours, or publish
    j0 += 64;
    if (j0 >= c) {
      mylevel = mx + 1;
      __hip_atomic_store(&level[p - base], mx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      done = true;
    } else {
      cc = c - j0 < 64 ? c - j0 : 64;
      pend = cc == 64 ? ~0ull : ((1ull << cc) - 1ull);
    }
  };
  bool news = false;   // a lane of this wave has found its level since the last exchange inside the wave (uniform)
  int idle = 0;        // rounds in a row without any answer (uniform)
  for (;;) {
    // inside the wave: every lane offers its level, every lane takes what its first 64 neighbours in this wave have;
    // repeated while that lets another lane finish (a row of the row-by-row order resolves lane after lane).  Only
    // when there is news: a wave that waits for other waves keeps to its polls.
    for (int sweep = 0; news && sweep < 64; ++sweep) {
      bool fresh = false;
      int lv[SC];
#pragma unroll
      for (int u = 0; u < SC; ++u) lv[u] = __shfl(mylevel, so[u] >= 0 ? so[u] : lane);   // every lane takes part
#pragma unroll
      for (int u = 0; u < SC; ++u) {
        if (!done && so[u] >= 0 && lv[u] >= 0 && ((pend >> sj[u]) & 1ull) && j0 == 0) {
          mx = lv[u] > mx ? lv[u] : mx;
          pend &= ~(1ull << sj[u]);
        }
      }
      if (!done && pend == 0) {
        settle();
        fresh = done;
      }
      if (!__any(fresh)) break;
    }
    news = false;
    const bool was_done = done;
    const unsigned long long pend_before = pend;
    const int j0_before = j0;
    if (!done) {
      // up to eight of the unknown neighbours per round, their loads in flight together: a round is one memory
      // round trip, and a round is what a hop of the chain costs between waves
      // (what a lane of this wave owns arrives through the exchange above and is not asked for in memory: with
      // hundreds of thousands of threads polling, the polls themselves are what the kernel waits for)
      unsigned long long m = j0 == 0 ? pend & ~intra : pend;
      int js[8], ls[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        js[u] = -1;
        ls[u] = -1;
        if (m) {
          js[u] = __builtin_ctzll(m);
          m &= m - 1;
          ls[u] = __hip_atomic_load(&level[nb[j0 + js[u]]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (js[u] >= 0 && ls[u] >= 0) {
          mx = ls[u] > mx ? ls[u] : mx;
          pend &= ~(1ull << js[u]);
        }
      }
      if (pend == 0) {
        settle();
      } else if (++polls > (1 << 18)) {
        // never seen; the wait is bounded all the same: publish, report, and the host drops the schedule
        mylevel = 1 << 29;
        __hip_atomic_store(&level[p - base], 1 << 29, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *gave_up = 1;
        done = true;
      }
    }
    if (__all(done)) break;
    news = __any(done && !was_done);
    // A wave that learns nothing backs off (up to ~16 us between polls): with a quarter of a million threads asking,
    // the polls of the waves far behind the front are what the waves at the front wait for.  Any answer -- one
    // neighbour's level is enough -- brings the wave back to polling at once.
    if (__any(pend != pend_before || j0 != j0_before || done != was_done)) idle = 0;
    else ++idle;
    if (!news) {
      if (idle < 4) {
        __builtin_amdgcn_s_sleep(8);
      } else {
        const int reps = idle - 3 < 4 ? idle - 3 : 4;
        for (int i = 0; i < reps; ++i) __builtin_amdgcn_s_sleep(127);
      }
    }
  }
}

// first position of every level in the level-sorted node list
__global__ __launch_bounds__(256) void sgs_level_offsets_kernel(const int* __restrict__ lvl_sorted, int64_t N,
                                                                int* __restrict__ off) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const int l = lvl_sorted[i];
  if (i == 0 || lvl_sorted[i - 1] != l) off[l] = (int)i;
}

// one level: wave = (node of the level, block of 64 realisations); the recursion step of sgs_sweep_kernel
__global__ __launch_bounds__(256) void sgs_level_sweep_kernel(const int* __restrict__ order, int first, int count,
                                                              const int* __restrict__ idx, const int* __restrict__ ncond,
                                                              const double* __restrict__ w,
                                                              const double* __restrict__ sigma, int k, int R, int rb,
                                                              double mean, double* __restrict__ zt) {
  const int lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= (int64_t)count * rb) return;   // whole wave
  const int64_t node = order[first + (int)(item / rb)];
  const int r = (int)(item % rb) * 64 + lane;
  const bool live = r < R;
  const int rr = live ? r : R - 1;
  const int c = ncond[node];
  const int* nb = idx + node * k;      // wave-uniform: scalar loads
  const double* ww = w + node * k;
  const double eps = zt[node * R + rr];   // the cell's own slot holds its normal until it is simulated
  double acc = 0.0;
  int j = 0;
  for (; j + 8 <= c; j += 8) {   // eight gathers in flight
    double zz[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) zz[u] = zt[(int64_t)nb[j + u] * R + rr];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = fma(ww[j + u], zz[u] - mean, acc);
  }
  for (; j < c; ++j) acc = fma(ww[j], zt[(int64_t)nb[j] * R + rr] - mean, acc);
  const double v = mean + acc + sigma[node] * eps;
  if (live) zt[node * R + r] = v;
}

// ---- short levels: the whole schedule in one launch (sgs_level_team_kernel) -------------------------------------------
// An order whose levels are short (the row-by-row LinearPath: ~3 100 levels of ~85 nodes on 512 x 512 cells) spends its
// sweep on launches: ~5 us per level whatever the level holds.  Realisations do not depend on each other, so a
// workgroup (a team) takes RL of them through every level by itself and the step from one level to the next is a
// workgroup barrier.  The team path keeps its own layout of everything it touches:
//   field    zq[team][position in the schedule][RL]: a team's part is contiguous, a 128-byte line belongs to ONE
//            team (in the node-major [N][R] field two teams on different XCDs share a line and every gather of a
//            half-written line goes to memory: 2.7 us per level), a level's normals and results are one contiguous run;
//   lists    neighbours as schedule positions, weights, sigma -- in schedule order, entries beyond a node's ncond
//            replaced by (the node itself, weight 0): exact zeros at the end of the same sum, so the values are those of
//            sgs_level_sweep_kernel.
__global__ __launch_bounds__(256) void sgs_team_inverse_kernel(const int* __restrict__ order, int64_t N,
                                                               int* __restrict__ inv) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < N) inv[order[i]] = (int)i;
}

__global__ __launch_bounds__(256) void sgs_team_lists_kernel(const int* __restrict__ order, const int* __restrict__ inv,
                                                             int64_t N, int k, int kn, int kw,
                                                             const int* __restrict__ ncond,
                                                             const int* __restrict__ idx, const double* __restrict__ w,
                                                             const double* __restrict__ sigma, int* __restrict__ nb_s,
                                                             double* __restrict__ w_s, double* __restrict__ sg_s) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= N * kn) return;   // kn >= kw: rows of kn ints and kw doubles, padded so that the rows of a wave's node slots
  const int64_t pos = e / kn;   // start in different LDS banks (sgs_team_strides)
  const int j = (int)(e - pos * kn);
  const int64_t node = order[pos];
  const bool kept = j < ncond[node];   // ncond <= k
  nb_s[e] = kept ? inv[idx[node * k + j]] : (int)pos;
  if (j < kw) w_s[pos * kw + j] = kept ? w[node * k + j] : 0.0;
  if (j == 0) sg_s[pos] = sigma[node];
}

// zq[team][pos][rl] = eps(seed, realisation, cell at pos); realisations beyond R (the last team's padding) get zeros
__global__ __launch_bounds__(256) void sgs_team_noise_kernel(uint64_t seed, int64_t first_real, int64_t N, int R, int RL,
                                                             int nteams, const double* __restrict__ noise,
                                                             const int* __restrict__ order, double* __restrict__ zq) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)nteams * N * RL) return;
  const int rl = (int)(e % RL);
  const int64_t pos = (e / RL) % N;
  const int r = (int)(e / ((int64_t)RL * N)) * RL + rl;
  const int64_t cell = order[pos];
  zq[e] = r < R ? (noise ? noise[(int64_t)r * N + cell] : philox_normal(seed, (uint32_t)(first_real + r), (uint64_t)cell))
                : 0.0;
}

__global__ __launch_bounds__(256) void sgs_team_seed_data_kernel(const int64_t* __restrict__ dlocs,
                                                                 const double* __restrict__ zd, int64_t nd,
                                                                 const int* __restrict__ inv, int64_t N, int R, int RL,
                                                                 double* __restrict__ zq) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= nd * R) return;
  const int64_t j = e / R;
  const int r = (int)(e % R);
  zq[((int64_t)(r / RL) * N + inv[dlocs[j]]) * RL + r % RL] = zd[j];
}

// out[r][cell] from the team layout through an LDS tile: 64 cells of one team; RL * 8 bytes per cell come in, 512-byte
// runs go out
__global__ __launch_bounds__(256) void sgs_team_out_kernel(const double* __restrict__ zq, const int* __restrict__ inv,
                                                           int64_t N, int R, int RL, double* __restrict__ out) {
  __shared__ double tile[64][65];
  const int64_t c0 = (int64_t)blockIdx.x * 64;
  const int team = blockIdx.y;
  for (int e = threadIdx.x; e < 64 * RL; e += 256) {
    const int i = e / RL, rl = e % RL;
    if (c0 + i < N) tile[rl][i] = zq[((int64_t)team * N + inv[c0 + i]) * RL + rl];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * RL; e += 256) {
    const int rl = e >> 6, i = e & 63;
    const int r = team * RL + rl;
    if (c0 + i < N && r < R) out[(int64_t)r * N + c0 + i] = tile[rl][i];
  }
}

// Fourteen waves compute: lane = (node slot, realisation), 896 / RL nodes per round, every gather of a node in flight
// at once.  Two waves stage: what a level needs besides its neighbours' values -- lists, weights, sigma, the level's
// normals -- does not depend on the sweep and is copied into LDS three chunks ahead with LDS-direct loads (no registers,
// so the copies stay in flight across the barriers: the staging waves wait with a COUNTED vmcnt, the barrier is the raw
// instruction; a wave's ordinary loads return in order, which is why the computing waves do not fetch any of this
// themselves).  Levels are cut into chunks of at most `cap` nodes (chunk_off); all LDS is one array.
constexpr int SGS_TEAM_THREADS = 1024;
constexpr int SGS_TEAM_STAGERS = 2;            // staging waves
constexpr int SGS_TEAM_ENTRIES = 2560;         // list entries per chunk: cap * (padded row) <= 2560
constexpr int SGS_TEAM_NORMALS = 1024;         // normals per chunk: cap * RL <= 1024
constexpr int SGS_TEAM_CAP = 128;              // nodes per chunk at most
constexpr int SGS_TEAM_AHEAD = 3;              // chunks in flight; LDS holds one stage more
// Row lengths of the staged lists (k a multiple of 4): the node slots of a wave read the same column of consecutive rows
// at once, so a row of k doubles (k = 16: 128 bytes) would put all of them into the same LDS banks.  kw = k + 2 doubles
// and kn = 4 mod 8 ints (both whole 16-byte pieces) spread eight consecutive rows over the 32 banks.
inline void sgs_team_strides(int k, int* kn, int* kw) {
  *kw = k + 2;
  *kn = k % 8 == 0 ? k + 4 : k + 8;
}
constexpr int SGS_TEAM_STAGES = SGS_TEAM_AHEAD + 1;
struct SgsTeamStage {
  double ww[SGS_TEAM_ENTRIES];
  double eps[SGS_TEAM_NORMALS];
  double sg[SGS_TEAM_CAP];
  int nb[SGS_TEAM_ENTRIES];
};
static_assert(SGS_TEAM_STAGES * sizeof(SgsTeamStage) <= 160 * 1024, "LDS of a CU");
// LDS-direct copies issued per staging wave and chunk; the staging waves' vmcnt leaves AHEAD - 1 chunks in flight
constexpr int SGS_TEAM_GLDS = (SGS_TEAM_ENTRIES * 4 + SGS_TEAM_ENTRIES * 8 + SGS_TEAM_NORMALS * 8) / 1024 / SGS_TEAM_STAGERS +
                              SGS_TEAM_CAP * 8 / 256 / SGS_TEAM_STAGERS;
static_assert(SGS_TEAM_GLDS == 21 && SGS_TEAM_AHEAD == 3, "the waits below are written for 21 copies and 3 chunks: vmcnt(42), (21)");
static_assert(SGS_TEAM_GLDS * SGS_TEAM_AHEAD <= 63, "vmcnt counts to 63");

// `bytes` valid bytes from src to dst (LDS), the staging waves together: pieces of 64 lanes x SIZE bytes, a piece per
// instruction, lanes beyond the end repeat the last SIZE bytes (every instruction is issued: the vmcnt count is fixed)
template <int SIZE, int PIECES>
__device__ __forceinline__ void sgs_team_copy(void* dst, const void* src, int bytes, int sw, int lane) {
  static_assert(PIECES % SGS_TEAM_STAGERS == 0 && (SIZE == 16 || SIZE == 4), "pieces divide among the staging waves");
#pragma unroll
  for (int u = 0; u < PIECES / SGS_TEAM_STAGERS; ++u) {
    const int piece = u * SGS_TEAM_STAGERS + sw;
    const int off = piece * 64 * SIZE + lane * SIZE;
    const char* g = static_cast<const char*>(src) + (off < bytes ? off : bytes - SIZE);
    auto gp = (const __attribute__((address_space(1))) void*)g;
    auto lp = (__attribute__((address_space(3))) void*)(static_cast<char*>(dst) + piece * 64 * SIZE);
    if constexpr (SIZE == 16)   // the builtin wants a literal
      __builtin_amdgcn_global_load_lds(gp, lp, 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds(gp, lp, 4, 0, 0);
  }
}

template <int RL>
__global__ __launch_bounds__(SGS_TEAM_THREADS) void sgs_level_team_kernel(
    const int* __restrict__ nb_s, const double* __restrict__ w_s, const double* __restrict__ sg_s,
    const int* __restrict__ chunk_off, int nchunks, int k, int kn, int kw, int64_t N, double mean,
    double* __restrict__ zq) {
  extern __shared__ __attribute__((aligned(16))) unsigned char team_lds[];
  SgsTeamStage* stages = reinterpret_cast<SgsTeamStage*>(team_lds);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int CW = SGS_TEAM_THREADS / 64 - SGS_TEAM_STAGERS;   // computing waves
  constexpr int NPW = 64 / RL, NPR = CW * NPW;
  const int q = lane / RL, rl = lane % RL;
  double* zb = zq + (int64_t)blockIdx.x * N * RL;   // this team's realisations, [position][RL]
  const int sw = __builtin_amdgcn_readfirstlane(wave) - CW;   // >= 0: staging wave

  auto issue = [&](int c) {
    SgsTeamStage& S = stages[c % SGS_TEAM_STAGES];
    const int first = chunk_off[c], cnt = chunk_off[c + 1] - first;
    sgs_team_copy<16, SGS_TEAM_ENTRIES * 4 / 1024>(S.nb, nb_s + (int64_t)first * kn, cnt * kn * 4, sw, lane);
    sgs_team_copy<16, SGS_TEAM_ENTRIES * 8 / 1024>(S.ww, w_s + (int64_t)first * kw, cnt * kw * 8, sw, lane);
    sgs_team_copy<16, SGS_TEAM_NORMALS * 8 / 1024>(S.eps, zb + (int64_t)first * RL, cnt * RL * 8, sw, lane);
    sgs_team_copy<4, SGS_TEAM_CAP * 8 / 256>(S.sg, sg_s + first, cnt * 8, sw, lane);
  };
  // all but the newest `left` chunks' copies have landed (left = chunks issued after the one that is needed)
  auto landed = [&](int left) {
    if (left >= 2) asm volatile("s_waitcnt vmcnt(42)" ::: "memory");
    else if (left == 1) asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  if (sw >= 0) {
    const int pre = nchunks < SGS_TEAM_AHEAD ? nchunks : SGS_TEAM_AHEAD;
    for (int c = 0; c < pre; ++c) issue(c);
    landed(pre - 1);   // chunk 0
  }
  int first = chunk_off[0], end = chunk_off[1];
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  for (int c = 0; c < nchunks; ++c) {
    const int end_next = chunk_off[c + 2 <= nchunks ? c + 2 : nchunks];   // boundaries one chunk ahead: off the chain
    if (sw >= 0) {
      // stage (c + AHEAD) % STAGES held chunk c - 1, and the barrier behind that chunk has been passed
      if (c + SGS_TEAM_AHEAD < nchunks) issue(c + SGS_TEAM_AHEAD);
      const int issued = c + SGS_TEAM_AHEAD < nchunks ? c + SGS_TEAM_AHEAD : nchunks - 1;   // newest chunk on its way
      landed(issued - (c + 1) > 0 ? issued - (c + 1) : 0);                                   // chunk c + 1 has landed
    } else {
      const SgsTeamStage& S = stages[c % SGS_TEAM_STAGES];
      const int cnt = end - first;
      for (int base = 0; base < cnt; base += NPR) {
        const int i = base + wave * NPW + q;
        const bool on = i < cnt;
        const int ii = on ? i : 0;
        const double sg = S.sg[ii];
        const double eps = S.eps[ii * RL + rl];   // the cell's own slot held its normal
        double acc = 0.0;
        const int* nbr = S.nb + ii * kn;
        const double* wwr = S.ww + ii * kw;
        auto batch = [&](int j, auto count) {     // `count` gathers in flight, then their terms in list order
          constexpr int C = decltype(count)::value;
          double zz[C];
#pragma unroll
          for (int u = 0; u < C; ++u) zz[u] = zb[(int64_t)nbr[j + u] * RL + rl];
#pragma unroll
          for (int u = 0; u < C; ++u) acc = fma(wwr[j + u], zz[u] - mean, acc);
        };
        int j = 0;
        for (; j + 16 <= k; j += 16) batch(j, std::integral_constant<int, 16>{});
        switch (k - j) {                          // k is a multiple of 4
          case 12: batch(j, std::integral_constant<int, 12>{}); break;
          case 8: batch(j, std::integral_constant<int, 8>{}); break;
          case 4: batch(j, std::integral_constant<int, 4>{}); break;
          default: break;
        }
        if (on) zb[(int64_t)(first + i) * RL + rl] = mean + acc + sg * eps;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the level's results have left before a wave of this CU gathers them
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    first = end;
    end = end_next;
  }
}

// one level of every visiting order at once (one order per realisation): thread = (path, node) of the level; the
// recursion step of sgs_sweep_paths_kernel on the realisation-major field
__global__ __launch_bounds__(256) void sgs_level_sweep_paths_kernel(const int* __restrict__ order, int first, int count,
                                                                    const int* __restrict__ idx,
                                                                    const int* __restrict__ ncond,
                                                                    const double* __restrict__ w,
                                                                    const double* __restrict__ sigma, int k, int64_t N,
                                                                    int R, int64_t path0, double mean,
                                                                    double* __restrict__ zr) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  const int64_t g = order[first + i];
  const int64_t pi = g / N, node = g - pi * N;
  const int64_t r = pi - path0;              // realisation r of this call walks path path0 + r
  if (r < 0 || r >= R) return;
  const int c = ncond[g];
  const int* nb = idx + g * k;
  const double* ww = w + g * k;
  double* z = zr + r * N;
  double acc = 0.0;
  int j = 0;
  for (; j + 4 <= c; j += 4) {                // four gathers in flight
    const double z0 = z[nb[j]], z1 = z[nb[j + 1]], z2 = z[nb[j + 2]], z3 = z[nb[j + 3]];
    acc = fma(ww[j], z0 - mean, acc);
    acc = fma(ww[j + 1], z1 - mean, acc);
    acc = fma(ww[j + 2], z2 - mean, acc);
    acc = fma(ww[j + 3], z3 - mean, acc);
  }
  for (; j < c; ++j) acc = fma(ww[j], z[nb[j]] - mean, acc);
  z[node] = mean + acc + sigma[g] * z[node];  // the cell's own slot holds its normal until it is simulated
}

// zt[dloc][r] = zdata for the conditioning cells
__global__ __launch_bounds__(256) void sgs_seed_data_kernel(const int64_t* __restrict__ dlocs,
                                                            const double* __restrict__ zd, int64_t nd, int64_t N,
                                                            int R, double* __restrict__ zt) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= nd * R) return;
  const int64_t j = e / R;
  const int r = (int)(e % R);
  const int64_t c = dlocs[j];
  zt[c * R + r] = zd[j];
}

// zt[cell][r] = eps(seed, realisation, cell): the standard normals the sweep consumes, drawn up front by the whole
// device (a single wave walking the path would otherwise spend half of every step inside Philox + Box-Muller)
__global__ __launch_bounds__(256) void sgs_noise_kernel(uint64_t seed, int64_t first_real, int64_t N, int R,
                                                        const double* __restrict__ noise, double* __restrict__ zt) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= N * R) return;
  const int64_t cell = e / R;
  const int r = (int)(e % R);
  zt[e] = noise ? noise[(int64_t)r * N + cell] : philox_normal(seed, (uint32_t)(first_real + r), (uint64_t)cell);
}

// out[r][cell] = zt[cell][r] through a 64 x 64 LDS tile (both sides coalesced)
__global__ __launch_bounds__(256) void sgs_transpose_kernel(const double* __restrict__ zt, int64_t N, int R,
                                                            double* __restrict__ out) {
  __shared__ double tile[64][65];
  const int64_t c0 = (int64_t)blockIdx.x * 64;
  const int r0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int64_t c = c0 + i;
    const int r = r0 + tx;
    if (c < N && r < R) tile[i][tx] = zt[c * R + r];
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i;
    const int64_t c = c0 + tx;
    if (c < N && r < R) out[(int64_t)r * N + c] = tile[tx][i];
  }
}

// stage B: the path recursion, one lane per realisation (all lanes visit the same node, so the neighbour list
// and the weights are wave-uniform loads and the field gathers are contiguous across lanes)
__global__ __launch_bounds__(64) void sgs_sweep_kernel(const int64_t* __restrict__ path, const int* __restrict__ rank,
                                                       const int* __restrict__ idx, const int* __restrict__ ncond,
                                                       const double* __restrict__ w, const double* __restrict__ sigma,
                                                       int k, int64_t N, int R, double mean,
                                                       double* __restrict__ zt) {
  const int lane = threadIdx.x;
  const int r = blockIdx.x * 64 + lane;
  const bool live = r < R;
  const int rr = live ? r : R - 1;  // idle lanes shadow the last realisation (no stores): the wave stays uniform
  // Everything that does not depend on simulated values is fetched one node ahead (two for the path itself):
  // lane u holds neighbour u's index and weight of the NEXT node, read back with readlane when it is consumed.
  int64_t node_n = path[0];
  int64_t node_nn = N > 1 ? path[1] : 0;
  int pc = ncond[node_n], prank = rank[node_n];
  double psg = sigma[node_n];
  int pidx = lane < k ? idx[node_n * k + lane] : 0;
  double pw = lane < k ? w[node_n * k + lane] : 0.0;
  double peps = zt[node_n * R + rr];  // the cell's own slot holds its normal until it is simulated
  for (int64_t t = 0; t < N; ++t) {
    const int64_t node = node_n;
    const int c = pc, rk = prank, myidx = pidx;
    const double sg = psg, myw = pw, eps = peps;
    if (t + 1 < N) {
      node_n = node_nn;
      node_nn = t + 2 < N ? path[t + 2] : 0;
      pc = ncond[node_n];
      prank = rank[node_n];
      psg = sigma[node_n];
      pidx = lane < k ? idx[node_n * k + lane] : 0;
      pw = lane < k ? w[node_n * k + lane] : 0.0;
      peps = zt[node_n * R + rr];
    }
    if (rk < 0) continue;  // conditioning cell
    double acc = 0.0;
    // sixteen neighbours per trip, all gathers of a trip in flight together: a node costs about one memory round trip
    for (int j0 = 0; j0 < c; j0 += 16) {
      double ww[16], zz[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int j = j0 + u < c ? j0 + u : c - 1;
        const int id = __builtin_amdgcn_readlane(myidx, j);
        const int wl = __builtin_amdgcn_readlane(__double2loint(myw), j);
        const int wh = __builtin_amdgcn_readlane(__double2hiint(myw), j);
        ww[u] = j0 + u < c ? __hiloint2double(wh, wl) : 0.0;
        zz[u] = zt[(int64_t)id * R + rr];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) acc = fma(ww[u], zz[u] - mean, acc);
    }
    const double v = mean + acc + sg * eps;
    if (live) zt[node * R + r] = v;
  }
}

// ---- one visiting order per realisation (seq.jl:99-102 calls traverse inside solvesingle: a RandomPath differs
//      from realisation to realisation).  Stage A runs once per path; the sweep gives every realisation a lane of its
//      own: nothing is shared between lanes any more, so the field is realisation-major [R][N] (= the output layout)
//      and every access is a per-lane gather.
__global__ __launch_bounds__(256) void sgs_noise_rows_kernel(uint64_t seed, int64_t first_real, int64_t N, int R,
                                                             const double* __restrict__ noise,
                                                             double* __restrict__ zr) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= N * R) return;
  const int64_t r = e / N, cell = e - r * N;
  zr[e] = noise ? noise[e] : philox_normal(seed, (uint32_t)(first_real + r), (uint64_t)cell);
}

__global__ __launch_bounds__(256) void sgs_seed_data_rows_kernel(const int64_t* __restrict__ dlocs,
                                                                 const double* __restrict__ zd, int64_t nd, int64_t N,
                                                                 int R, double* __restrict__ zr) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= nd * R) return;
  const int64_t r = e / nd, j = e - r * nd;
  zr[r * N + dlocs[j]] = zd[j];
}

__global__ __launch_bounds__(64) void sgs_sweep_paths_kernel(const int64_t* __restrict__ paths,
                                                             const int* __restrict__ ranks,
                                                             const int* __restrict__ idx,
                                                             const int* __restrict__ ncond,
                                                             const double* __restrict__ w,
                                                             const double* __restrict__ sigma, int k, int64_t N, int R,
                                                             int64_t path0, double mean, double* __restrict__ zr) {
  const int r = blockIdx.x * 64 + threadIdx.x;
  if (r >= R) return;
  const int64_t pi = path0 + r;                 // realisation r of this call walks path path0 + r
  const int64_t* path = paths + pi * N;
  const int* rank = ranks + pi * N;
  const int* nbr = idx + pi * N * k;
  const int* nc = ncond + pi * N;
  const double* wt = w + pi * N * k;
  const double* sg = sigma + pi * N;
  double* z = zr + (int64_t)r * N;
  int64_t node_n = path[0];
  for (int64_t t = 0; t < N; ++t) {
    const int64_t node = node_n;
    if (t + 1 < N) node_n = path[t + 1];        // next node's address is known a step ahead
    if (rank[node] < 0) continue;               // conditioning cell
    const int c = nc[node];
    const int* nb = nbr + node * k;
    const double* ww = wt + node * k;
    double acc = 0.0;
    int j = 0;
    for (; j + 4 <= c; j += 4) {                // four gathers in flight
      const double z0 = z[nb[j]], z1 = z[nb[j + 1]], z2 = z[nb[j + 2]], z3 = z[nb[j + 3]];
      acc = fma(ww[j], z0 - mean, acc);
      acc = fma(ww[j + 1], z1 - mean, acc);
      acc = fma(ww[j + 2], z2 - mean, acc);
      acc = fma(ww[j + 3], z3 - mean, acc);
    }
    for (; j < c; ++j) acc = fma(ww[j], z[nb[j]] - mean, acc);
    z[node] = mean + acc + sg[node] * z[node];  // the cell's own slot holds its normal until it is simulated
  }
}

}  // namespace gss

using namespace gss;

struct gss_sgs {
  VgDev vg;
  int dim = 0, k = 0;
  int64_t N = 0, nd = 0;
  double mean = 0.0;
  int filter_after = 0;   // GSS_SGS_MASK_AFTER_SEARCH
  int64_t npaths = 1, path_base = 0;  // npaths > 1: one visiting order per realisation, path p <-> realisation path_base + p
  DevBuf path, rank, idx, ncond, w, sigma, dlocs, zd;   // per path: N entries (path, rank, ncond, sigma), N k (idx, w)
  DevBuf field;  // node-major [N][R] working field of gss_sgs_realize, kept between calls (grows to the largest R seen)
  DevBuf order;  // shared visiting order: nodes sorted by level of the dependency graph
  std::vector<int> lvl_off;   // lvl_off[l] .. lvl_off[l + 1] - 1: positions of level l in `order` (level 0 = conditioning cells)
  // short levels (sgs_level_team_kernel): schedule-ordered padded lists / weights / sigma, chunk boundaries in `order`
  DevBuf nb_sched, w_sched, sg_sched, inv_sched, chunk_off;
  int team_chunks = 0, team_cap = 0, team_rl = 0;
};

// levels of the dependency graph of path 0 (see sgs_level_sweep_kernel); leaves h->lvl_off empty when switched off
static int32_t sgs_build_levels(gss_sgs* h, hipStream_t s) {
  static const bool enabled = !(std::getenv("GSS_SGS_LEVELS") && std::getenv("GSS_SGS_LEVELS")[0] == '0');
  h->lvl_off.clear();
  if (!enabled || h->N * h->npaths >= ((int64_t)1 << 30)) return GSS_OK;
  const int64_t N = h->N * h->npaths;   // (path, node) pairs, path-major like the per-path arrays
  DevBuf level, node, lvl_sorted, off, tmp, flag;
  GSS_TRY(level.alloc(sizeof(int) * (size_t)N));
  GSS_TRY(node.alloc(sizeof(int) * (size_t)N));
  GSS_TRY(lvl_sorted.alloc(sizeof(int) * (size_t)N));
  GSS_TRY(h->order.alloc(sizeof(int) * (size_t)N));
  const dim3 grid((unsigned)((N + 255) / 256));
  hipLaunchKernelGGL(sgs_level_init_kernel, grid, dim3(256), 0, s, h->rank.as<int>(), N, level.as<int>(), node.as<int>());
  GSS_HIP(hipGetLastError());
  GSS_TRY(flag.alloc(sizeof(int)));
  GSS_HIP(hipMemsetAsync(flag.p, 0, sizeof(int), s));
  hipLaunchKernelGGL(sgs_level_chain_kernel, grid, dim3(256), 0, s, h->path.as<int64_t>(), h->rank.as<int>(),
                     h->idx.as<int>(), h->ncond.as<int>(), h->k, h->N, h->npaths, level.as<int>(), flag.as<int>());
  GSS_HIP(hipGetLastError());
  int gave_up = 0;
  GSS_HIP(hipMemcpyAsync(&gave_up, flag.p, sizeof(int), hipMemcpyDeviceToHost, s));
  GSS_HIP(hipStreamSynchronize(s));
  if (gave_up) {   // stage B then walks the path as before
    h->order.release();
    return GSS_OK;
  }
  size_t tb = 0;
  GSS_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, level.as<int>(), lvl_sorted.as<int>(), node.as<int>(),
                                             h->order.as<int>(), (int)N, 0, 32, s));
  GSS_TRY(tmp.alloc(tb > 0 ? tb : 16));
  GSS_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, level.as<int>(), lvl_sorted.as<int>(), node.as<int>(),
                                             h->order.as<int>(), (int)N, 0, 32, s));
  int L = 0;
  GSS_HIP(hipMemcpyAsync(&L, lvl_sorted.as<int>() + (N - 1), sizeof(int), hipMemcpyDeviceToHost, s));
  GSS_HIP(hipStreamSynchronize(s));
  GSS_TRY(off.alloc(sizeof(int) * (size_t)(L + 2)));
  GSS_HIP(hipMemsetAsync(off.p, 0xff, sizeof(int) * (size_t)(L + 2), s));
  hipLaunchKernelGGL(sgs_level_offsets_kernel, grid, dim3(256), 0, s, lvl_sorted.as<int>(), N, off.as<int>());
  GSS_HIP(hipGetLastError());
  std::vector<int> ho((size_t)L + 2);
  GSS_HIP(hipMemcpyAsync(ho.data(), off.p, sizeof(int) * (size_t)(L + 2), hipMemcpyDeviceToHost, s));
  GSS_HIP(hipStreamSynchronize(s));
  ho[(size_t)L + 1] = (int)N;
  for (int l = L; l >= 0; --l)   // a level without nodes (only level 0 can be: no conditioning cells) starts where the next does
    if (ho[(size_t)l] < 0) ho[(size_t)l] = ho[(size_t)l + 1];
  h->lvl_off = std::move(ho);
  // Orders with short levels sweep in one launch (sgs_level_team_kernel).  The average level size decides: a launch per
  // level costs ~5 us whatever it holds, a team's level ~2.8 us up to 112 nodes (measured).
  h->team_chunks = 0;
  static const char* team_env = std::getenv("GSS_SGS_TEAM");   // "0": never, "1": whenever it applies (measurements)
  const int64_t nsim = N - h->lvl_off[1];
  const double avg = L > 0 ? (double)nsim / (double)L : 0.0;
  const bool want = team_env ? team_env[0] != '0' : avg <= 128.0;
  // (any maxneighbors up to 64 -- the reference's default is 10 --: the staged rows are padded to a multiple of four,
  //  the padding adds exact zeros like the entries beyond a node's ncond)
  if (h->npaths == 1 && h->k <= 64 && L > 0 && want) {
    h->team_rl = avg <= 14.0 ? 64 : avg <= 28.0 ? 32 : avg <= 56.0 ? 16 : 8;   // one round (896 / RL nodes) per level where possible
    int kn = 0, kw = 0;
    sgs_team_strides((h->k + 3) & ~3, &kn, &kw);
    int cap = SGS_TEAM_ENTRIES / kn < SGS_TEAM_CAP ? SGS_TEAM_ENTRIES / kn : SGS_TEAM_CAP;
    if (cap > SGS_TEAM_NORMALS / h->team_rl) cap = SGS_TEAM_NORMALS / h->team_rl;
    std::vector<int> co;
    for (int l = 1; l <= L; ++l)
      for (int a = h->lvl_off[(size_t)l]; a < h->lvl_off[(size_t)l + 1]; a += cap) co.push_back(a);
    co.push_back((int)N);
    GSS_TRY(h->chunk_off.alloc(sizeof(int) * co.size()));
    GSS_HIP(hipMemcpyAsync(h->chunk_off.p, co.data(), sizeof(int) * co.size(), hipMemcpyHostToDevice, s));
    GSS_TRY(h->nb_sched.alloc(sizeof(int) * (size_t)(N * kn)));
    GSS_TRY(h->w_sched.alloc(sizeof(double) * (size_t)(N * kw)));
    GSS_TRY(h->sg_sched.alloc(sizeof(double) * (size_t)N));
    GSS_TRY(h->inv_sched.alloc(sizeof(int) * (size_t)N));
    hipLaunchKernelGGL(sgs_team_inverse_kernel, grid, dim3(256), 0, s, h->order.as<int>(), N, h->inv_sched.as<int>());
    hipLaunchKernelGGL(sgs_team_lists_kernel, dim3((unsigned)((N * kn + 255) / 256)), dim3(256), 0, s,
                       h->order.as<int>(), h->inv_sched.as<int>(), N, h->k, kn, kw, h->ncond.as<int>(), h->idx.as<int>(),
                       h->w.as<double>(), h->sigma.as<double>(), h->nb_sched.as<int>(), h->w_sched.as<double>(),
                       h->sg_sched.as<double>());
    GSS_HIP(hipGetLastError());
    GSS_HIP(hipStreamSynchronize(s));   // co is a local
    h->team_chunks = (int)co.size() - 1;
    h->team_cap = cap;
  }
  return GSS_OK;
}

extern "C" {

int32_t gss_sgs_create_paths(gss_sgs_t** out, const gss_variogram_t* vg, double mean, const double* centroids,
                             int64_t N, int32_t dim, const int64_t* path, int64_t npaths, int64_t path_base,
                             const int64_t* dlocs, const double* zdata, int64_t nd, int32_t maxneighbors,
                             int32_t minneighbors, double radius, const double* inv_radii, int32_t flags,
                             void* stream) {
  GSS_ENTRY();
  GSS_REQUIRE(out != nullptr, "gss_sgs_create: out is NULL");
  GSS_REQUIRE(npaths >= 1 && path_base >= 0 && (npaths == 1 || path != nullptr), "gss_sgs_create_paths: bad path set");
  *out = nullptr;
  GSS_REQUIRE(vg && centroids, "gss_sgs_create: NULL argument");
  GSS_REQUIRE(N >= 1 && N < INT_MAX && dim >= 1 && dim <= 3, "gss_sgs_create: bad sizes");
  GSS_REQUIRE(nd >= 0 && nd <= N && (nd == 0 || (dlocs && zdata)), "gss_sgs_create: bad conditioning data");
  GSS_REQUIRE(maxneighbors >= 1 && maxneighbors <= N, "maxneighbors %d outside 1..%lld (searcher_ui clamps it, "
              "ui.jl:18-20)", maxneighbors, (long long)N);
  GSS_REQUIRE(maxneighbors <= SGS_BIG_MAX_K, "maxneighbors = %d: at most %d neighbours in SGS", maxneighbors,
              SGS_BIG_MAX_K);
  hipStream_t s = to_stream(stream);
  auto* h = new gss_sgs();
  struct Guard {
    gss_sgs* p;
    ~Guard() { delete p; }
  } guard{h};
  GSS_REQUIRE(vg_is_stationary(vg), "variogram model must be stationary");  // fft.jl:91, lu.jl:110
  GSS_TRY(make_vgdev(vg, &h->vg));
  GSS_REQUIRE(h->vg.dim == dim, "variogram dimension %d != domain dimension %d", h->vg.dim, dim);
  h->dim = dim;
  h->k = maxneighbors;
  h->N = N;
  h->nd = nd;
  h->mean = mean;
  h->npaths = npaths;
  h->path_base = path_base;
  h->filter_after = (flags & GSS_SGS_MASK_AFTER_SEARCH) ? 1 : 0;
  const int metric = (flags >> GSS_SGS_METRIC_SHIFT) & 7;
  // Haversine(r) (seq.jl:91-98 hands `distance` to the searcher): its ranking key does not depend on r and has no box
  // bounds, so it runs on the exhaustive search -- which exists unmasked only: available with the mask applied to the
  // search result (GSS_SGS_MASK_AFTER_SEARCH, the front-ends' default), not for the masked search
  GSS_REQUIRE(metric == GSS_METRIC_EUCLIDEAN || metric == GSS_METRIC_CITYBLOCK || metric == GSS_METRIC_CHEBYSHEV ||
                  (metric == GSS_METRIC_HAVERSINE && h->filter_after),
              "gss_sgs_create: search distance %d -- Euclidean, Cityblock or Chebyshev; Haversine only with "
              "GSS_SGS_MASK_AFTER_SEARCH (there is no masked exhaustive search)", metric);
  GSS_TRY(check_metric(metric, 1.0, dim, radius, inv_radii));   // a ball only with the Euclidean distance (ui.jl:25-31)
  const int64_t P = npaths;

  // visiting rank of every cell (-1 = conditioning cell) per path; each path must be a permutation of 0..N-1
  std::vector<int64_t> hpath((size_t)(N * P));
  std::vector<int> hrank((size_t)(N * P), INT_MAX);
  for (int64_t pp = 0; pp < P; ++pp) {
    int* rk = hrank.data() + pp * N;
    for (int64_t t = 0; t < N; ++t) {
      const int64_t c = path ? path[pp * N + t] : t;  // LinearPath (seq.jl:33)
      GSS_REQUIRE(c >= 0 && c < N && rk[c] == INT_MAX, "path %lld is not a permutation of the domain (step %lld)",
                  (long long)pp, (long long)t);
      hpath[(size_t)(pp * N + t)] = c;
      rk[c] = (int)t;
    }
    for (int64_t j = 0; j < nd; ++j) {
      GSS_REQUIRE(dlocs[j] >= 0 && dlocs[j] < N, "data location %lld outside the domain", (long long)dlocs[j]);
      GSS_REQUIRE(rk[dlocs[j]] >= 0, "data location %lld given twice", (long long)dlocs[j]);
      rk[dlocs[j]] = -1;
    }
  }

  DevBuf cent, bmin, cnt;
  GSS_TRY(cent.alloc(sizeof(double) * (size_t)(N * dim)));
  GSS_TRY(h->path.alloc(sizeof(int64_t) * (size_t)(N * P)));
  GSS_TRY(h->rank.alloc(sizeof(int) * (size_t)(N * P)));
  GSS_TRY(h->idx.alloc(sizeof(int) * (size_t)(N * P * h->k)));
  GSS_TRY(h->ncond.alloc(sizeof(int) * (size_t)(N * P)));
  GSS_TRY(cnt.alloc(sizeof(int) * (size_t)N));
  GSS_TRY(h->w.alloc(sizeof(double) * (size_t)(N * P * h->k)));
  GSS_TRY(h->sigma.alloc(sizeof(double) * (size_t)(N * P)));
  GSS_HIP(hipMemcpyAsync(cent.p, centroids, cent.bytes, hipMemcpyHostToDevice, s));
  GSS_HIP(hipMemcpyAsync(h->path.p, hpath.data(), h->path.bytes, hipMemcpyHostToDevice, s));
  GSS_HIP(hipMemcpyAsync(h->rank.p, hrank.data(), h->rank.bytes, hipMemcpyHostToDevice, s));
  GSS_HIP(hipMemsetAsync(h->w.p, 0, h->w.bytes, s));
  if (nd > 0) {
    GSS_TRY(h->dlocs.alloc(sizeof(int64_t) * (size_t)nd));
    GSS_TRY(h->zd.alloc(sizeof(double) * (size_t)nd));
    GSS_HIP(hipMemcpyAsync(h->dlocs.p, dlocs, h->dlocs.bytes, hipMemcpyHostToDevice, s));
    GSS_HIP(hipMemcpyAsync(h->zd.p, zdata, h->zd.bytes, hipMemcpyHostToDevice, s));
  }
  KnnIndex ix;
  if (N >= KNN_DEVICE_BUILD_MIN) GSS_TRY(knn_index_build_device(cent.as<double>(), N, dim, &ix, s));  // already in HBM
  else GSS_TRY(knn_index_build(centroids, N, dim, &ix, s));
  GSS_TRY(bmin.alloc(sizeof(int) * (size_t)ix.nb));
  DevBuf rawidx;   // GSS_SGS_MASK_AFTER_SEARCH: the unmasked neighbour lists, shared by every path
  DevBuf bigscr;   // more than 180 neighbours: covariance triangles of the workgroups of sgs_weights_big_kernel
  for (int64_t pp = 0; pp < P; ++pp) {   // stage A once per visiting order
    int* rk = h->rank.as<int>() + pp * N;
    int* idxp = h->idx.as<int>() + pp * N * h->k;
    hipLaunchKernelGGL(sgs_batch_minrank_kernel, dim3((unsigned)((ix.nb + 3) / 4)), dim3(256), 0, s, ix.perm.as<int>(),
                       rk, (int)N, ix.nb, bmin.as<int>());
    GSS_HIP(hipGetLastError());
    {
      ProfScope ps("sgs_search", s);
      if (h->filter_after) {   // one unmasked search serves every path: k nearest cells of the whole domain
        if (pp == 0) {
          GSS_TRY(rawidx.alloc(sizeof(int) * (size_t)(N * h->k)));
          if (metric == GSS_METRIC_HAVERSINE)   // exhaustive (passes of 64 beyond 64 neighbours)
            GSS_TRY(knn_search_dev(cent.as<double>(), N, dim, cent.as<double>(), N, h->k, -1.0, nullptr, rawidx.as<int>(),
                                   cnt.as<int>(), s, metric));
          else if (h->k > SGS_MAX_K)
            GSS_TRY(knn_search_indexed_any(ix, cent.as<double>(), cent.as<double>(), N, h->k, radius, inv_radii,
                                           rawidx.as<int>(), cnt.as<int>(), s, metric));
          else
            GSS_TRY(knn_search_indexed(ix, cent.as<double>(), N, h->k, radius, inv_radii, rawidx.as<int>(), cnt.as<int>(), s,
                                       nullptr, nullptr, nullptr, metric));
        }
      } else if (h->k > SGS_MAX_K) {   // masked search in passes of 64; the lists go through the big weights kernel
        if (pp == 0) GSS_TRY(rawidx.alloc(sizeof(int) * (size_t)(N * h->k)));
        GSS_TRY(knn_search_indexed_any(ix, cent.as<double>(), cent.as<double>(), N, h->k, radius, inv_radii,
                                       rawidx.as<int>(), cnt.as<int>(), s, metric, rk, rk, bmin.as<int>()));
      } else {
        GSS_TRY(knn_search_indexed(ix, cent.as<double>(), N, h->k, radius, inv_radii, idxp, cnt.as<int>(), s, rk, rk,
                                   bmin.as<int>(), metric));
      }
    }
    {
      ProfScope ps("sgs_weights", s);
#define GSS_SGS_ARGS h->vg, cent.as<double>(), rk, N, h->k, minneighbors, \
                     (h->filter_after ? rawidx.as<int>() : idxp), cnt.as<int>(), \
                     h->ncond.as<int>() + pp * N, h->w.as<double>() + pp * N * h->k, h->sigma.as<double>() + pp * N, \
                     idxp, h->filter_after
      if (h->k > SGS_MAX_K) {
        // one workgroup per node, grid-stride; the triangle in LDS when it fits, else a slab of HBM per workgroup
        const int64_t need = (int64_t)h->k * (h->k + 1) / 2 + h->k;
        const int use_lds = need <= SGS_BIG_LDS_DOUBLES ? 1 : 0;
        int64_t nwg = N < 1024 ? N : 1024;
        if (!use_lds) {
          const int64_t cap = ((int64_t)1 << 30) / (int64_t)(sizeof(double) * (size_t)need);   // 1 GiB of slabs
          if (nwg > cap) nwg = cap > 1 ? cap : 1;
          if (bigscr.bytes < sizeof(double) * (size_t)(need * nwg)) {
            bigscr.release();
            GSS_TRY(bigscr.alloc(sizeof(double) * (size_t)(need * nwg)));
          }
        }
        const size_t lds = use_lds ? sizeof(double) * (size_t)need : 0;
#define GSS_SGS_BIG(D)                                                                                                  \
  do {                                                                                                                   \
    if (lds > 48 * 1024)                                                                                                 \
      GSS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sgs_weights_big_kernel<D>),                              \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                               \
    hipLaunchKernelGGL((sgs_weights_big_kernel<D>), dim3((unsigned)nwg), dim3(SGS_BIG_NT), lds, s, h->vg,              \
                       cent.as<double>(), rk, N, h->k, minneighbors, rawidx.as<int>(), cnt.as<int>(),                   \
                       h->ncond.as<int>() + pp * N, h->w.as<double>() + pp * N * h->k, h->sigma.as<double>() + pp * N,  \
                       idxp, bigscr.as<double>(), need, use_lds);                                                        \
  } while (0)
        switch (dim) {
          case 1: GSS_SGS_BIG(1); break;
          case 2: GSS_SGS_BIG(2); break;
          default: GSS_SGS_BIG(3); break;
        }
#undef GSS_SGS_BIG
      } else {
      switch (dim) {
        case 1: hipLaunchKernelGGL((sgs_weights_kernel<1>), dim3((unsigned)N), dim3(64), 0, s, GSS_SGS_ARGS); break;
        case 2: hipLaunchKernelGGL((sgs_weights_kernel<2>), dim3((unsigned)N), dim3(64), 0, s, GSS_SGS_ARGS); break;
        default: hipLaunchKernelGGL((sgs_weights_kernel<3>), dim3((unsigned)N), dim3(64), 0, s, GSS_SGS_ARGS); break;
      }
      }
#undef GSS_SGS_ARGS
      GSS_HIP(hipGetLastError());
    }
  }
  {
    ProfScope ps("sgs_levels", s);
    GSS_TRY(sgs_build_levels(h, s));
  }
  GSS_HIP(hipStreamSynchronize(s));  // host staging vectors and scratch are released on return
  guard.p = nullptr;
  *out = h;
  return GSS_OK;
}

int32_t gss_sgs_create(gss_sgs_t** out, const gss_variogram_t* vg, double mean, const double* centroids, int64_t N,
                       int32_t dim, const int64_t* path, const int64_t* dlocs, const double* zdata, int64_t nd,
                       int32_t maxneighbors, int32_t minneighbors, double radius, const double* inv_radii,
                       int32_t flags, void* stream) {
  GSS_ENTRY();
  return gss_sgs_create_paths(out, vg, mean, centroids, N, dim, path, 1, 0, dlocs, zdata, nd, maxneighbors, minneighbors,
                              radius, inv_radii, flags, stream);
}

int32_t gss_sgs_destroy(gss_sgs_t* h) {
  GSS_ENTRY();
  delete h;
  return GSS_OK;
}

int32_t gss_sgs_weights(gss_sgs_t* h, int32_t* idx, int32_t* ncond, double* w, double* sigma, int32_t mem,
                        void* stream) {
  GSS_ENTRY();
  GSS_REQUIRE(h != nullptr, "NULL handle");
  hipStream_t s = to_stream(stream);
  const hipMemcpyKind kind = mem == GSS_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  // (the first visiting order when the handle holds several)
  const size_t nk = (size_t)(h->N * h->k), n1 = (size_t)h->N;
  if (idx) GSS_HIP(hipMemcpyAsync(idx, h->idx.p, sizeof(int) * nk, kind, s));
  if (ncond) GSS_HIP(hipMemcpyAsync(ncond, h->ncond.p, sizeof(int) * n1, kind, s));
  if (w) GSS_HIP(hipMemcpyAsync(w, h->w.p, sizeof(double) * nk, kind, s));
  if (sigma) GSS_HIP(hipMemcpyAsync(sigma, h->sigma.p, sizeof(double) * n1, kind, s));
  if (mem == GSS_MEM_HOST) GSS_HIP(hipStreamSynchronize(s));
  return GSS_OK;
}

// Realisations first_real .. first_real + nreals - 1 with every array in HBM; asynchronous on s.
static int32_t sgs_realize_block(gss_sgs_t* h, uint64_t seed, int64_t first_real, int64_t nreals, const double* noise,
                                 double* out, hipStream_t s) {
  const int64_t N = h->N;
  const int R = (int)nreals;
  if (h->npaths > 1) {
    // one visiting order per realisation: realisation first_real + i walks path first_real + i - path_base
    const int64_t p0 = first_real - h->path_base;
    GSS_REQUIRE(p0 >= 0 && p0 + nreals <= h->npaths, "realisations %lld..%lld have no visiting order in this handle "
                "(paths cover %lld..%lld)", (long long)first_real, (long long)(first_real + nreals - 1),
                (long long)h->path_base, (long long)(h->path_base + h->npaths - 1));
    double* zr = out;   // realisation-major working field = the output itself
    {
      ProfScope ps("sgs_noise", s);
      hipLaunchKernelGGL(sgs_noise_rows_kernel, dim3((unsigned)((N * R + 255) / 256)), dim3(256), 0, s, seed, first_real,
                         N, R, noise ? noise : nullptr, zr);
      GSS_HIP(hipGetLastError());
    }
    if (h->nd > 0) {
      hipLaunchKernelGGL(sgs_seed_data_rows_kernel, dim3((unsigned)((h->nd * R + 255) / 256)), dim3(256), 0, s,
                         h->dlocs.as<int64_t>(), h->zd.as<double>(), h->nd, N, R, zr);
      GSS_HIP(hipGetLastError());
    }
    if (!h->lvl_off.empty()) {   // level by level, every visiting order at once
      ProfScope ps("sgs_sweep", s);
      const int L = (int)h->lvl_off.size() - 2;
      for (int l = 1; l <= L; ++l) {
        const int first = h->lvl_off[(size_t)l], count = h->lvl_off[(size_t)l + 1] - first;
        if (count <= 0) continue;
        hipLaunchKernelGGL(sgs_level_sweep_paths_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s,
                           h->order.as<int>(), first, count, h->idx.as<int>(), h->ncond.as<int>(), h->w.as<double>(),
                           h->sigma.as<double>(), h->k, N, R, p0, h->mean, zr);
      }
      GSS_HIP(hipGetLastError());
    } else {
      ProfScope ps("sgs_sweep", s);
      hipLaunchKernelGGL(sgs_sweep_paths_kernel, dim3((unsigned)((R + 63) / 64)), dim3(64), 0, s, h->path.as<int64_t>(),
                         h->rank.as<int>(), h->idx.as<int>(), h->ncond.as<int>(), h->w.as<double>(),
                         h->sigma.as<double>(), h->k, N, R, p0, h->mean, zr);
      GSS_HIP(hipGetLastError());
    }
    return GSS_OK;
  }
  // the working field is GBs (8 N R bytes): allocating and freeing it on every call costs more than a short sweep
  const bool team = !h->lvl_off.empty() && h->team_chunks > 0;
  const int RL = team ? h->team_rl : 1;
  const int nteams = (R + RL - 1) / RL;
  const size_t field_doubles = team ? (size_t)nteams * (size_t)N * (size_t)RL : (size_t)(N * R);
  if (h->field.bytes < sizeof(double) * field_doubles) {
    h->field.release();
    GSS_TRY(h->field.alloc(sizeof(double) * field_doubles));
  }
  DevBuf& zt = h->field;
  if (team) {   // short levels: one launch, the field in the team layout (sgs_level_team_kernel)
    {
      ProfScope ps("sgs_noise", s);
      hipLaunchKernelGGL(sgs_team_noise_kernel, dim3((unsigned)((field_doubles + 255) / 256)), dim3(256), 0, s, seed,
                         first_real, N, R, RL, nteams, noise ? noise : nullptr, h->order.as<int>(), zt.as<double>());
      GSS_HIP(hipGetLastError());
    }
    if (h->nd > 0) {
      hipLaunchKernelGGL(sgs_team_seed_data_kernel, dim3((unsigned)((h->nd * R + 255) / 256)), dim3(256), 0, s,
                         h->dlocs.as<int64_t>(), h->zd.as<double>(), h->nd, h->inv_sched.as<int>(), N, R, RL,
                         zt.as<double>());
      GSS_HIP(hipGetLastError());
    }
    {
      ProfScope ps("sgs_sweep", s);
      int kn = 0, kw = 0;
      const int kp = (h->k + 3) & ~3;   // list length the kernel walks: the rows are padded beyond it
      sgs_team_strides(kp, &kn, &kw);
#define GSS_SGS_TEAM_LAUNCH(W)                                                                                         \
  do {                                                                                                                 \
    static uint64_t attr_set = 0;                                                                                      \
    if (first_on_this_device(attr_set)) {                                                                              \
      GSS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sgs_level_team_kernel<W>),                             \
                                  hipFuncAttributeMaxDynamicSharedMemorySize,                                          \
                                  (int)(SGS_TEAM_STAGES * sizeof(SgsTeamStage))));                                     \
    }                                                                                                                  \
    hipLaunchKernelGGL((sgs_level_team_kernel<W>), dim3((unsigned)nteams), dim3(SGS_TEAM_THREADS),                     \
                       SGS_TEAM_STAGES * sizeof(SgsTeamStage), s, h->nb_sched.as<int>(), h->w_sched.as<double>(),      \
                       h->sg_sched.as<double>(), h->chunk_off.as<int>(), h->team_chunks, kp, kn, kw, N, h->mean,       \
                       zt.as<double>());                                                                               \
  } while (0)
      switch (RL) {
        case 64: GSS_SGS_TEAM_LAUNCH(64); break;
        case 32: GSS_SGS_TEAM_LAUNCH(32); break;
        case 16: GSS_SGS_TEAM_LAUNCH(16); break;
        default: GSS_SGS_TEAM_LAUNCH(8); break;
      }
#undef GSS_SGS_TEAM_LAUNCH
      GSS_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(sgs_team_out_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)nteams), dim3(256), 0, s,
                       zt.as<double>(), h->inv_sched.as<int>(), N, R, RL, out);
    GSS_HIP(hipGetLastError());
    return GSS_OK;
  }
  {
    ProfScope ps("sgs_noise", s);
    hipLaunchKernelGGL(sgs_noise_kernel, dim3((unsigned)((N * R + 255) / 256)), dim3(256), 0, s, seed, first_real, N, R,
                       noise ? noise : nullptr, zt.as<double>());
    GSS_HIP(hipGetLastError());
  }
  if (h->nd > 0) {
    hipLaunchKernelGGL(sgs_seed_data_kernel, dim3((unsigned)((h->nd * R + 255) / 256)), dim3(256), 0, s,
                       h->dlocs.as<int64_t>(), h->zd.as<double>(), h->nd, N, R, zt.as<double>());
    GSS_HIP(hipGetLastError());
  }
  if (!h->lvl_off.empty()) {
    ProfScope ps("sgs_sweep", s);
    const int L = (int)h->lvl_off.size() - 2;
    const int rb = (R + 63) / 64;
    for (int l = 1; l <= L; ++l) {
      const int first = h->lvl_off[(size_t)l], count = h->lvl_off[(size_t)l + 1] - first;
      if (count <= 0) continue;
      const int64_t waves = (int64_t)count * rb;
      hipLaunchKernelGGL(sgs_level_sweep_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, h->order.as<int>(),
                         first, count, h->idx.as<int>(), h->ncond.as<int>(), h->w.as<double>(), h->sigma.as<double>(),
                         h->k, R, rb, h->mean, zt.as<double>());
    }
    GSS_HIP(hipGetLastError());
  } else {
    ProfScope ps("sgs_sweep", s);
    if (h->k > SGS_MAX_K)
      hipLaunchKernelGGL(sgs_sweep_big_kernel, dim3((unsigned)((R + 63) / 64)), dim3(64), 0, s, h->path.as<int64_t>(),
                         h->rank.as<int>(), h->idx.as<int>(), h->ncond.as<int>(), h->w.as<double>(),
                         h->sigma.as<double>(), h->k, N, R, h->mean, zt.as<double>());
    else
      hipLaunchKernelGGL(sgs_sweep_kernel, dim3((unsigned)((R + 63) / 64)), dim3(64), 0, s, h->path.as<int64_t>(),
                         h->rank.as<int>(), h->idx.as<int>(), h->ncond.as<int>(), h->w.as<double>(),
                         h->sigma.as<double>(), h->k, N, R, h->mean, zt.as<double>());
    GSS_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(sgs_transpose_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)((R + 63) / 64)), dim3(256), 0, s,
                     zt.as<double>(), N, R, out);
  GSS_HIP(hipGetLastError());
  return GSS_OK;
}

int32_t gss_sgs_realize(gss_sgs_t* h, uint64_t seed, int64_t first_real, int64_t nreals, const double* noise,
                        double* out, int32_t mem, void* stream) {
  GSS_ENTRY();
  GSS_REQUIRE(h != nullptr, "NULL handle");
  GSS_REQUIRE(nreals >= 0 && first_real >= 0 && nreals < INT_MAX, "bad realisation range");
  if (nreals == 0) return GSS_OK;
  GSS_REQUIRE(out != nullptr, "gss_sgs_realize: out is NULL");
  hipStream_t s = to_stream(stream);
  const int64_t N = h->N;
  if (h->npaths > 1) {
    const int64_t p0 = first_real - h->path_base;
    GSS_REQUIRE(p0 >= 0 && p0 + nreals <= h->npaths, "realisations %lld..%lld have no visiting order in this handle "
                "(paths cover %lld..%lld)", (long long)first_real, (long long)(first_real + nreals - 1),
                (long long)h->path_base, (long long)(h->path_base + h->npaths - 1));
  }
  if (mem == GSS_MEM_DEVICE) {
    GSS_TRY(sgs_realize_block(h, seed, first_real, nreals, noise, out, s));
    GSS_HIP(hipStreamSynchronize(s));
    return GSS_OK;
  }
  // Host arrays (seq.jl:137-141 returns host vectors): blocks of realisations -- whole groups of 64 lanes for a shared
  // visiting order -- through the output ring; block b crosses the bus while block b + 1 is swept.
  int64_t rb = OutStream::default_chunk(sizeof(double) * (size_t)N, nreals);
  if (h->npaths <= 1 && rb < nreals) rb = rb < 64 ? 64 : rb / 64 * 64;
  if (rb > nreals) rb = nreals;
  OutStream os;
  GSS_TRY(os.begin(out, sizeof(double) * (size_t)N, nreals, mem, s, rb));
  DevBuf nbuf;
  if (noise) GSS_TRY(nbuf.alloc(sizeof(double) * (size_t)(rb * N)));
  for (int64_t r0 = 0; r0 < nreals; r0 += rb) {
    const int64_t n = nreals - r0 < rb ? nreals - r0 : rb;
    if (noise)
      GSS_HIP(hipMemcpyAsync(nbuf.p, noise + r0 * N, sizeof(double) * (size_t)(n * N), hipMemcpyHostToDevice, s));
    double* dout = nullptr;
    GSS_TRY(os.slot(r0, s, &dout));
    GSS_TRY(sgs_realize_block(h, seed, first_real + r0, n, noise ? nbuf.as<double>() : nullptr, dout, s));
    GSS_TRY(os.done(r0 + n - 1, s));
  }
  return os.finish(s);
}

}  // extern "C"
