/*
 * gss.h -- C-ABI of libgss_hip.so: MI355X (gfx950) kernels for the kriging-estimation and
 * Gaussian-simulation hot path of juliohm/GeoStatsSolvers.jl (KrigingSolver, FFTGS, LUGS) and, on the
 * same kernels, IDWSolver, LWRSolver and SGS.
 *
 * This is the drop-in boundary.  The reference is
 * pure Julia and has no FFI of its own; each entry point below replaces the arithmetic that the
 * cited reference lines delegate to their Julia dependencies, and is what a `ccall` from the
 * reference's `solve` / `preprocess` / `solvesingle` methods binds.
 *
 * Conventions
 *   - every function returns an int32 status (GSS_OK == 0); no C++ exception crosses the ABI;
 *     gss_last_error() returns the message of the last failure on the calling thread.
 *   - coordinates are "point-major": point i occupies d consecutive doubles.  This is the memory
 *     layout of a Julia d x n column-major matrix (PointSet coordinates), so Julia passes them
 *     without a copy.
 *   - all floating point is FP64, indices are int32/int64 as declared, 0-based on this side.
 *   - `mem` says where the LARGE arrays of the call live: GSS_MEM_HOST (library copies through
 *     PCIe; gss_krig_predict_global overlaps the copies with the computation, piece by piece) or GSS_MEM_DEVICE (pointers are HBM addresses on the current device; nothing is
 *     copied and the call is asynchronous on `stream`).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls may arrive on different streams:
 *     the library recycles its scratch memory across calls, so a call on a new stream is ordered (event wait,
 *     on the device) behind everything the library queued on the stream of the previous call.
 *     Some calls spread independent pieces of their work over helper streams of the process (gss_fftgs_realize:
 *     slabs of the strided FFT passes; gss_lugs_create / gss_lugs_realize: trailing updates beside the next panel,
 *     column blocks of L22 W; the estimation calls on host arrays: copies beside the computation; gss_krig_create
 *     with GSS_KRIG_ASYNC_FIT: the fit).  The process has five such streams, created together at the first use.
 *     Those streams are fenced by events on both sides: everything such a call does starts
 *     after what `stream` held at the call and is complete, in stream order, before whatever is put on `stream` next.
 *   - host threads: every export takes one process-wide lock for its whole duration, so calls from several host
 *     threads (Julia `Threads.@threads` over variables, finalizers on the GC thread) are SAFE and run one after
 *     another -- whichever handles and streams they use; the stream chain above then orders their device work.  What
 *     is NOT provided is concurrency: a call with host arrays holds the lock until its last byte has arrived.  A
 *     handle may be used from any thread, by one call at a time (which the lock guarantees).  `gss_last_error` is
 *     per thread.
 *   - host results of the simulation calls (gss_fftgs_realize, gss_lugs_realize, gss_sgs_realize with GSS_MEM_HOST)
 *     leave the device in chunks of ~256 MiB while the following realisations are computed; at most three chunks are
 *     staged in HBM whatever `nreals` is.  A page-locked destination (hipHostMalloc / hipHostRegister) is written by
 *     the DMA engine directly; a pageable one goes through pinned bounce buffers of the library.
 *   - handles are opaque and owned by the library; the caller owns every buffer it passes and the library never
 *     returns memory it allocated.
 */
#ifndef GSS_H
#define GSS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSS_VERSION 100

/* ---- status codes ------------------------------------------------------------------------ */
enum {
  GSS_OK = 0,
  GSS_ERR_INVALID = 1,      /* bad argument (maps to ArgumentError / AssertionError in the shim) */
  GSS_ERR_HIP = 2,          /* HIP / rocFFT runtime failure                                      */
  GSS_ERR_NOT_POSDEF = 3,   /* covariance factorisation hit a non-positive pivot                  */
  GSS_ERR_UNSUPPORTED = 4,  /* feature outside the hot-path scope           */
  GSS_ERR_NO_DEVICE = 5,
  GSS_ERR_ALLOC = 6
};

enum { GSS_MEM_HOST = 0, GSS_MEM_DEVICE = 1 };

/* ---- variogram model: replaces Variography.jl objects at fft.jl:91,98; lu.jl:110,124,131,132;
 *      krig.jl:65 (solver parameter `variogram`) -------------------------------------------- */
enum {
  GSS_VG_GAUSSIAN = 0,
  GSS_VG_EXPONENTIAL = 1,
  GSS_VG_SPHERICAL = 2,
  GSS_VG_MATERN = 3,         /* any order nu in (0, 50]; 1 is the reference's default order                   */
  GSS_VG_CUBIC = 4,
  GSS_VG_PENTASPHERICAL = 5,
  GSS_VG_SINEHOLE = 6,       /* gamma = (sill - nugget) (1 - sin(pi h/r) / (pi h/r)) + nugget          */
  GSS_VG_POWER = 7           /* gamma = scaling h^exponent + nugget: NOT stationary.  Fields: range = scaling,
                              * nu = exponent in (0,2), sill = constant A of the pseudo-covariance A - gamma(h)
                              * (any A >= max gamma over the data; ordinary / universal / external-drift
                              * kriging results do not depend on it).  Rejected by simple kriging and by the
                              * simulation solvers (fft.jl:91, lu.jl:110).                                  */
};

typedef struct gss_variogram {
  int32_t kind;        /* GSS_VG_*                                                   */
  int32_t dim;         /* embedding dimension d, 1..3                                */
  double sill;
  double nugget;
  double range;        /* isotropic range; ignored (==1) when aniso != 0             */
  double nu;           /* Matern order                                               */
  int32_t aniso;       /* 1: Mahalanobis distance with inv_radii (MetricBall((a,b))) */
  int32_t reserved;
  double inv_radii[3]; /* 1 / ball radii                                             */
  /* nested models gamma = sum_i c_i gamma_i ([DEP] Variography NestedVariogram): the fields above describe the
   * first structure (its `sill` includes the TOTAL nugget), each extra structure adds `sill` = its own contribution
   * with its own kind / range / anisotropy.  Total sill = sill + sum extra[i].sill.                              */
  int32_t nextra;      /* 0..3 */
  int32_t reserved2;
  struct {
    int32_t kind;
    int32_t aniso;
    double sill;
    double range;
    double nu;
    double inv_radii[3];
  } extra[3];
} gss_variogram_t;

/* ---- kriging variant: replaces ui.jl:40-50 (kriging_ui) model choice ---------------------- */
enum { GSS_KRIG_SIMPLE = 0, GSS_KRIG_ORDINARY = 1, GSS_KRIG_UNIVERSAL = 2, GSS_KRIG_EXTDRIFT = 3 };

/* per-point status byte written by the predict calls (krig.jl:213-214 `missing, missing`) */
enum { GSS_PT_OK = 0, GSS_PT_MISSING = 1, GSS_PT_SINGULAR = 2 };

typedef struct gss_krig gss_krig_t;
typedef struct gss_fftgs gss_fftgs_t;
typedef struct gss_lugs gss_lugs_t;
typedef struct gss_sgs gss_sgs_t;

/* ---- library ------------------------------------------------------------------------------ */
int32_t gss_version(void);
int32_t gss_device_count(int32_t* count);
int32_t gss_init(int32_t device);          /* bind the calling process to `device` (one process per GPU) */
int32_t gss_shutdown(void);
int32_t gss_last_error(char* buf, int32_t len);
int32_t gss_synchronize(void* stream);
/* Copy `bytes` from device memory (produced on `stream`) to host memory at the rate the simulation calls deliver their
 * results: a page-locked destination directly, a pageable one through the library's pinned bounce buffers (pieces of
 * 32 MiB, the host copy of one piece beside the transfer of the next).  For hosts that compose results on the device
 * (the conditional branch fft.jl:176-192) and hand the reference's host vectors back.  Returns when the data are there. */
int32_t gss_dev_to_host(void* dst, const void* src_dev, int64_t bytes, void* stream);
/* Released device blocks are cached for re-use (up to GSS_POOL_MAX_MB, default a quarter of the device memory that
 * was free at the first allocation, at most 16 GiB); gss_trim_pool gives them all back to the driver -- for a host
 * program whose own allocator (torch, RCCL, rocFFT) has just failed.  Synchronises the device. */
int32_t gss_trim_pool(void);
/* Counters for tests and diagnostics: "pool_bytes" (device bytes in the block cache), "out_ring_bytes" (HBM staged for
 * the host outputs of the last simulation call), "out_chunks" (chunks that call moved), "panel_giveups" (times a
 * single-launch factorisation left through its bounded wait and was repeated on the launch-per-block path),
 * "ipc_route" (how the last gss_state_ipc_import reached the owner's device: 0 same device, 1 visible peer, 2 not among
 * the visible devices, 3 refused -- visible but not peer-accessible). */
int32_t gss_stat(const char* name, int64_t* value);

/* ---- multi-GPU: one process per GPU, the preprocess state of rank 0 replicated to the peers -----------------
 *      The reference runs `preprocess` once and maps `solvesingle` over realisations, on worker processes if asked
 *      (fft.jl:62,145; lu.jl:76,171); kriging shards domain points over a replicated factor (krig.jl:180).  Realisation
 *      r depends on (seed, r) only, so the shards need nothing else from each other.  Two routes, both behind this ABI:
 *
 *      RCCL ("ncclBroadcast of the factor over xGMI"): rank 0 calls gss_comm_unique_id, the HOST carries
 *      the GSS_COMM_ID_BYTES bytes to the peers (Julia `remotecall`, an MPI / torch.distributed broadcast, a file ...),
 *      every rank calls gss_comm_init after gss_init(device); then gss_state_bcast(kind, handle, root, stream) on every
 *      rank -- the root with the handle it computed, the peers with one created GSS_*_NO_FACTOR / _NO_SPECTRUM from the
 *      same inputs -- is one ncclBroadcast of the state buffer followed by the peers' adopt.  RCCL is loaded at run time.
 *
 *      HIP IPC (no communicator, works between processes that share one device as well): the owner writes a token with
 *      gss_state_ipc_export, the host carries its GSS_IPC_TOKEN_BYTES bytes, a peer calls gss_state_ipc_import: it maps
 *      the owner's buffer, pulls it with one device-to-device copy over its own xGMI link and adopts.  The owner keeps
 *      its handle alive until the peers have imported.  The token names the owner's device (PCI domain / bus / device):
 *      the importer enables peer access when that device is another visible one, and refuses -- GSS_ERR_UNSUPPORTED, the
 *      message names gss_state_bcast and recomputation -- when it is visible but not peer-accessible.             */
enum { GSS_STATE_KRIG = 0, GSS_STATE_FFTGS = 1, GSS_STATE_LUGS = 2 };   /* which create call `handle` came from */
#define GSS_COMM_ID_BYTES 128
#define GSS_IPC_TOKEN_BYTES 96
int32_t gss_comm_unique_id(uint8_t* id);                               /* id[GSS_COMM_ID_BYTES], root only */
int32_t gss_comm_init(const uint8_t* id, int32_t rank, int32_t nranks);
int32_t gss_comm_info(int32_t* rank, int32_t* nranks);                 /* -1, 0 without a communicator */
int32_t gss_comm_destroy(void);
int32_t gss_state_bcast(int32_t kind, void* handle, int32_t root, void* stream);
int32_t gss_state_ipc_export(int32_t kind, void* handle, uint8_t* token);          /* token[GSS_IPC_TOKEN_BYTES] */
int32_t gss_state_ipc_import(int32_t kind, void* handle, const uint8_t* token, void* stream);

/* ---- kernel timing (bench.py's roofline leg): when enabled every launch of a named hot kernel is
 *      bracketed by HIP events on the stream it is launched on; gss_profile_read synchronises
 *      those events and returns the summed duration and the launch count for `name`
 *      ("krig_rhs", "krig_quadform", "fftgs_noise", "fftgs_fwd", "fftgs_phase", "fftgs_inv", ...). */
int32_t gss_profile_enable(int32_t on);
int32_t gss_profile_reset(void);
int32_t gss_profile_read(const char* name, double* total_ms, int64_t* launches);

/* ---- pairwise covariance: replaces `sill(g) .- Variography.pairwise(g, A, B)` at
 *      fft.jl:98, lu.jl:124,131,132.  out is na x nb, row-major with leading dimension ldo.
 *      b == NULL computes the symmetric na x na matrix. ---------------------------------------- */
int32_t gss_cov_pairwise(const gss_variogram_t* vg, const double* a, int64_t na, const double* b,
                         int64_t nb, double* out, int64_t ldo, int32_t mem, void* stream);

/* ---- neighbour search: replaces `search!(neighbors, center, searcher)` krig.jl:210 and the
 *      KNearestSearch / KBallSearch construction ui.jl:27,30.  Exact; neighbours ordered by
 *      ascending (FP64 squared distance accumulated in dimension order without FMA, index).
 *      radius < 0: plain k-NN.  radius >= 0: only neighbours with d^2 <= radius^2 (isotropic) or
 *      Mahalanobis d^2 <= 1 when inv_radii != NULL.  idx is m x k (int32, -1 padded). ---------- */
enum {                        /* solver parameter `distance` (krig.jl:72, idw.jl:54, lwr.jl:57), [DEP] Distances.jl */
  GSS_METRIC_EUCLIDEAN = 0,   /* default; with inv_radii: Mahalanobis ball                                       */
  GSS_METRIC_CITYBLOCK = 1,
  GSS_METRIC_CHEBYSHEV = 2,
  GSS_METRIC_HAVERSINE = 3    /* points are (longitude, latitude) in degrees, metric_param = sphere radius       */
};
/* metric != EUCLIDEAN cannot be combined with a ball (searcher_ui uses either the ball or the metric, ui.jl:25-31). */
int32_t gss_knn_search(const double* xdata, int64_t n, int32_t dim, const double* centers, int64_t m,
                       int32_t k, double radius, const double* inv_radii, int32_t metric, double metric_param,
                       int32_t* idx, int32_t* count, int32_t mem, void* stream);

/* ---- KrigingSolver ------------------------------------------------------------------------
 * gss_krig_create replaces preprocess (krig.jl:76-128) + GeoStatsModels.fit of exactsolve
 * (krig.jl:176): it uploads the non-missing samples and, unless GSS_KRIG_NO_FACTOR is set,
 * factorises the (n+nc)^2 kriging system on the device.
 *   xdata n x d point-major, z n values, drift_data n x ndrift (EXTDRIFT only, row per point).
 *   xdata, z and drift_data are HOST arrays (the preprocess of krig.jl:76-128 is host logic: the library reads the
 *   coordinates for the drift centring / scaling and the search index, then uploads them); only the predict calls
 *   take `mem`.
 */
enum {
  GSS_KRIG_NO_FACTOR = 1, /* moving-neighbourhood use only, or factor arrives by broadcast */
  /* gss_krig_create returns once the fit is queued (on a stream of the library, behind what `stream` holds): the
   * first gss_krig_predict_global assembles its right-hand sides beside it, and reports the fit's status
   * (GSS_ERR_NOT_POSDEF) itself; any other use of the factor waits for the fit first.  The reference's `solve`
   * fits and predicts in one call (krig.jl:166-186), so nothing changes for it. */
  GSS_KRIG_ASYNC_FIT = 2
};

int32_t gss_krig_create(gss_krig_t** out, const gss_variogram_t* vg, int32_t variant, double sk_mean,
                        int32_t degree, int32_t ndrift, const double* xdata, const double* z,
                        const double* drift_data, int64_t n, int32_t flags, void* stream);
int32_t gss_krig_destroy(gss_krig_t* h);

/* number of constraints nc and system size n+nc */
int32_t gss_krig_info(const gss_krig_t* h, int64_t* n, int32_t* nc);

/* device address + size of the factor state (inverse block factor W' and dual weights) so that
 * the host can broadcast it to peer GPUs over RCCL; after a broadcast into
 * a handle created with GSS_KRIG_NO_FACTOR call gss_krig_adopt_factor. */
int32_t gss_krig_factor_buffer(gss_krig_t* h, void** dev_ptr, int64_t* bytes);
int32_t gss_krig_adopt_factor(gss_krig_t* h);

/* global neighbourhood: replaces the predictprob loop krig.jl:180-183.
 * xdom m x d point-major, drift_dom m x ndrift; outputs mean[m], var[m], status[m]. */
int32_t gss_krig_predict_global(gss_krig_t* h, const double* xdom, const double* drift_dom, int64_t m,
                                double* mean, double* var, uint8_t* status, int32_t mem, void* stream);

/* Block support (optional; default: point support at the centroids).
 * krig.jl:180 hands `pdomain[ind]` -- on a grid the CELL -- to predictprob, and [RECALL] the reference's dependencies
 * then average the covariances over sample points inside the cell.  After this call gss_krig_predict_global and
 * gss_krig_predict_knn treat xdom as centroids of cells of size cell[0..d-1] and regularises by the midpoint rule with nsub points per axis
 * (nsub^d samples s): c0_i = mean_s C(x_i, c + s), variance = mean_{s,s'} C(s, s') - c0 . weights.  Simple / ordinary
 * kriging and drifts of degree <= 1 (whose cell average is the centroid value).  nsub = 0 or cell = NULL: back to
 * point support.  The dependency's own sampling scheme is version dependent and not in the tree: this one is the
 * library's documented convention. */
int32_t gss_krig_set_block_support(gss_krig_t* h, const double* cell, int32_t nsub, void* stream);

/* moving neighbourhood: replaces approxsolve krig.jl:188-234 (search + fit + predictprob per point).
 * k = maxneighbors (already clamped by searcher_ui; 1..4096: up to 64 on the MFMA-tile kernel, beyond that the
 * search runs in passes of 64 and one workgroup solves each point's system), radius / inv_radii / metric as gss_knn_search
 * (the metric only ranks neighbours; covariances keep the variogram's own distance).
 * idx_out (m x k int32) and count_out (m) may be NULL. */
int32_t gss_krig_predict_knn(gss_krig_t* h, const double* xdom, const double* drift_dom, int64_t m,
                             int32_t k, int32_t minneighbors, double radius, const double* inv_radii,
                             int32_t metric, double metric_param, double* mean, double* var, uint8_t* status,
                             int32_t* idx_out, int32_t* count_out, int32_t mem, void* stream);

/* re-use a factorised handle with new data values (same locations): the conditional-FFTGS
 * pattern fft.jl:176-188 where one kriging system serves every realisation.
 * zbatch is nbatch x n; mean_out nbatch x m (row per batch).  No variances. */
int32_t gss_krig_predict_global_batch(gss_krig_t* h, const double* xdom, int64_t m, const double* zbatch,
                                      int64_t nbatch, double* mean_out, int32_t mem, void* stream);

/* ---- IDWSolver / LWRSolver on the neighbour-search kernel -------------
 * gss_idw_predict replaces the estimation loop idw.jl:111-142: neighbours by gss_knn_search's rule,
 *   w_i = 1 / d_i^exponent, mean = sum w_i z_i / sum w_i, dist = min d_i; a zero distance copies that
 *   sample and reports dist = 0 (idw.jl:131-134).  Output column `dist` is `<var>_distance` (idw.jl:149).
 * gss_lwr_predict replaces lwr.jl:114-147: delta_i = d_i / max d, W = diag(weight(delta_i)),
 *   theta = (X'WX)^-1 X'Wz with X = [1 x], mean = theta . [1; x0], var = |W X (X'WX)^-1 [1; x0]|
 *   (stored under `<var>_variance` exactly as lwr.jl:145,154 does).
 *   weight(h) = exp(-weight_a * h^weight_p) for GSS_WEIGHT_EXP (reference default a = 3, p = 2,
 *   lwr.jl:58) or (1 - h^3)^3 for GSS_WEIGHT_TRICUBE.
 * k = number of neighbours the searcher returns (ui.jl:16-23): 1..n; k == n is `maxneighbors = nothing` (every
 * sample, no search); beyond 64 the search runs in passes of 64 (haversine: on the exhaustive kernel).  radius / inv_radii /
 * metric as gss_knn_search;
 * the weights use the distances of that metric (searchdists!, idw.jl:120).
 * status: GSS_PT_MISSING when fewer than minneighbors were found (idw.jl:123, lwr.jl:126),
 * GSS_PT_SINGULAR when the LWR normal equations are not positive definite (the reference throws).
 * xdata n x d and xdom m x d point-major; status may be NULL. */
enum { GSS_WEIGHT_EXP = 0, GSS_WEIGHT_TRICUBE = 1 };

int32_t gss_idw_predict(const double* xdata, const double* z, int64_t n, int32_t dim, const double* xdom,
                        int64_t m, int32_t k, int32_t minneighbors, double radius, const double* inv_radii,
                        int32_t metric, double metric_param, double exponent, double* mean, double* dist,
                        uint8_t* status, int32_t mem, void* stream);
int32_t gss_lwr_predict(const double* xdata, const double* z, int64_t n, int32_t dim, const double* xdom,
                        int64_t m, int32_t k, int32_t minneighbors, double radius, const double* inv_radii,
                        int32_t metric, double metric_param, int32_t weight_kind, double weight_a,
                        double weight_p, double* mean, double* var, uint8_t* status, int32_t mem, void* stream);
/* Several value columns on ONE search and ONE weight vector per point.  The reference's estimation loop is generic over
 * the value type (idw.jl:128-141: `mu = sum(ws[i] * vs[i])`, exercised with CoDa compositions in
 * test/estimation/idw.jl:47-65, whose arithmetic is linear in the log-parts) and several variables measured on the same
 * samples share everything but the values.  z: nz columns of n (column c at z + c * n); mean: nz columns of m (column c
 * at mean + c * m); dist / var / status: one per point, as in the single-column calls (which are these with nz = 1). */
int32_t gss_idw_predict_cols(const double* xdata, const double* z, int64_t n, int32_t dim, int32_t nz,
                             const double* xdom, int64_t m, int32_t k, int32_t minneighbors, double radius,
                             const double* inv_radii, int32_t metric, double metric_param, double exponent,
                             double* mean, double* dist, uint8_t* status, int32_t mem, void* stream);
int32_t gss_lwr_predict_cols(const double* xdata, const double* z, int64_t n, int32_t dim, int32_t nz,
                             const double* xdom, int64_t m, int32_t k, int32_t minneighbors, double radius,
                             const double* inv_radii, int32_t metric, double metric_param, int32_t weight_kind,
                             double weight_a, double weight_p, double* mean, double* var, uint8_t* status,
                             int32_t mem, void* stream);
/* LWR with the neighbours' weights supplied by the caller, for weight functions that cannot cross the ABI -- an arbitrary
 * `weightfun` closure (lwr.jl:58,136): the host searches (gss_knn_search), evaluates delta = d / max d and w = f(delta)
 * itself and hands over idx (m x k, 0-based, as the search wrote it), count (m) and weights (m x k); the rest of
 * lwr.jl:137-145 (normal equations about the estimation point, norm(r)) runs here.  The lists are the caller's: a count
 * above k is clamped and a point whose list holds an index outside 0 .. n-1 is reported GSS_PT_SINGULAR, never gathered. */
int32_t gss_lwr_predict_weights(const double* xdata, const double* z, int64_t n, int32_t dim, const double* xdom, int64_t m,
                                int32_t k, int32_t minneighbors, const int32_t* idx, const int32_t* count,
                                const double* weights, double* mean, double* var, uint8_t* status, int32_t mem,
                                void* stream);

/* ---- FFTGS ------------------------------------------------------------------------------
 * gss_fftgs_create replaces preprocess fft.jl:62-103 (unconditional part): covariance to the
 * centre cell, F = sqrt(|fft(fftshift(C))|), F[1] = 0.  dims[0] is the fastest axis (Julia
 * column-major), ndim in 1..3.
 */
enum { GSS_FFTGS_NO_SPECTRUM = 1 /* the spectrum arrives by broadcast: allocate the state, compute nothing */ };
int32_t gss_fftgs_create(gss_fftgs_t** out, const gss_variogram_t* vg, int32_t ndim, const int64_t* dims,
                         const double* spacing, double mean, int32_t flags, void* stream);
int32_t gss_fftgs_destroy(gss_fftgs_t* h);
/* full-size spectral amplitude F (prod(dims) doubles, element order) for parity checks */
int32_t gss_fftgs_spectrum(gss_fftgs_t* h, double* f_out, int32_t mem, void* stream);
/* device address + size of the state (rescaled half-spectrum amplitude, then sum F^2 and the rescale factor) so that
 * rank 0's preprocess (fft.jl:62-103, run once) can be broadcast to the peer GPUs over RCCL, which then realise their
 * share (fft.jl:145); after a broadcast into a handle created with GSS_FFTGS_NO_SPECTRUM call gss_fftgs_adopt_state. */
int32_t gss_fftgs_state_buffer(gss_fftgs_t* h, void** dev_ptr, int64_t* bytes);
int32_t gss_fftgs_adopt_state(gss_fftgs_t* h, void* stream);
/* replaces solvesingle fft.jl:145-173 for realisations first_real .. first_real+nreals-1.
 * noise == NULL: Philox4x32-10 uniform noise keyed by (seed, realisation) generated on device;
 * else noise is nreals x N uniform values supplied by the caller (parity-test mode).
 * inds (ninds int64, 0-based parent indices, may be NULL) gathers a grid view (fft.jl:152,173).
 * out is nreals x (ninds or N). */
int32_t gss_fftgs_realize(gss_fftgs_t* h, uint64_t seed, int64_t first_real, int64_t nreals,
                          const double* noise, const int64_t* inds, int64_t ninds, double* out,
                          int32_t mem, void* stream);

/* ---- LUGS -------------------------------------------------------------------------------
 * gss_lugs_create replaces preprocess lu.jl:105-147 for one variable: covariance blocks,
 * L11, B12 = L11 \ C12, d2, L22 = chol(C22 - B12'B12).  centroids N x d point-major;
 * dlocs nd sorted 0-based data locations with values z1 (after initbuff, lu.jl:86,113-114).
 */
enum {
  GSS_LUGS_NO_FACTOR = 1, /* L22 / d2 arrive by broadcast (lu.jl:76 runs on one rank): allocate, do not factorise */
  GSS_LUGS_FACT_LU = 2    /* solver parameter `factorization = lu` (lu.jl:70,107): L11 and L22 are the unit lower
                           * factors `lu(Symmetric(.)).L` of a partial-pivot LU, exactly as the reference takes them
                           * (they are not square roots of the covariance); default is `cholesky` */
};
int32_t gss_lugs_create(gss_lugs_t** out, const gss_variogram_t* vg, const double* centroids, int64_t N,
                        const int64_t* dlocs, const double* z1, int64_t nd, double mean, int32_t flags,
                        void* stream);
int32_t gss_lugs_destroy(gss_lugs_t* h);
int32_t gss_lugs_info(const gss_lugs_t* h, int64_t* ns, int64_t* nd);
/* copy out L22 (ns x ns, column-major, lower) and d2 (ns) for parity checks; either may be NULL */
int32_t gss_lugs_factor(gss_lugs_t* h, double* l22, double* d2, int32_t mem, void* stream);
/* device address + size of the state (L22 then d2) for the RCCL broadcast of SURVEY.md section 8e: rank 0 runs the
 * preprocess, the peers create their handle with GSS_LUGS_NO_FACTOR, receive the state and call gss_lugs_adopt_state. */
int32_t gss_lugs_state_buffer(gss_lugs_t* h, void** dev_ptr, int64_t* bytes);
int32_t gss_lugs_adopt_state(gss_lugs_t* h);
/* replaces solvesingle / lusim lu.jl:171-224: y2 = d2 + L22 * w  (w = rho*w1 + sqrt(1-rho^2)*w2
 * when w1 != NULL), scatter to dlocs/slocs, add mean when unconditional.
 * noise == NULL: w2 = Philox normals keyed by (seed, realisation); else nreals x ns supplied.
 * out nreals x N; w_out (nreals x ns, may be NULL) returns the w2 used (for co-simulation). */
int32_t gss_lugs_realize(gss_lugs_t* h, uint64_t seed, int64_t first_real, int64_t nreals,
                         const double* noise, double rho, const double* w1, double* out, double* w_out,
                         int32_t mem, void* stream);

/* ---- SGS -----------------------------------------------------------
 * gss_sgs_create replaces SGS.preprocess (sgs.jl:56-85: SimpleKriging(variogram, mean) and the marginal
 *   Normal(mean, sqrt(sill))) and the realisation-independent part of the SeqSim path loop: for every
 *   node of `path` the masked search among already simulated cells (seq.jl:105), the fit (seq.jl:121)
 *   and the simple-kriging weights / standard deviation behind predictprob (seq.jl:126).  Nodes with
 *   fewer than minneighbors simulated neighbours, or a failed fit, draw from the marginal (seq.jl:107-109,
 *   124-128).  centroids N x d (host), path = visiting order (N 0-based cell indices, NULL = LinearPath),
 *   dlocs / zdata = conditioning cells and their values (initbuff with NearestInit, seq.jl:85);
 *   maxneighbors <= 1024 (beyond 64: the search in passes of 64, one workgroup per node for the weights);
 *   radius / inv_radii as gss_knn_search.  All realisations of a handle share the path.  The handle also keeps the
 *   levels of the recursion's dependency graph (a node's level = 1 + the highest level among its neighbours): the
 *   realisations are then simulated level by level, every node of a level at once, with the same sums in the same
 *   order as the walk along the path (bit-identical fields).
 * gss_sgs_realize replaces solvesingle (seq.jl:76-141) for realisations first_real..first_real+nreals-1:
 *   z[node] = mean + sum_j lambda_j (z[nb_j] - mean) + sigma eps, eps = Philox normal (seed, realisation,
 *   cell) or noise[r * N + cell] when given.  out is nreals x N.  The handle keeps its node-major working field
 *   (8 N nreals bytes for the largest nreals seen) until gss_sgs_destroy.
 * gss_sgs_weights (test support): per node the neighbour list (N x k), the number of conditioning
 *   neighbours actually used (0 = marginal or data cell), the weights (N x k) and sigma (N). */
/* flags: how `search!(neighbors, p, searcher, mask=simulated)` (seq.jl:105) treats the mask.  0: the k nearest AMONG the
 * already simulated cells.  GSS_SGS_MASK_AFTER_SEARCH: the k nearest cells of the whole domain (the node itself
 * included), of which the already simulated ones are kept -- what [DEP] Meshes' KNearestSearch / KBallSearch are recalled
 * to do (the mask is applied to the result of the tree query); the front-ends pass it by default. */
enum { GSS_SGS_MASK_AFTER_SEARCH = 1 };
/* bits 4..6 of flags: GSS_METRIC_* of the neighbour search (the solver parameter `distance`, seq.jl:91-98):
 * Euclidean (0, the only one that combines with a ball), Cityblock or Chebyshev; Haversine (its key has no box bounds:
 * exhaustive search) with GSS_SGS_MASK_AFTER_SEARCH only -- the masked search has no exhaustive variant. */
#define GSS_SGS_METRIC_SHIFT 4
int32_t gss_sgs_create(gss_sgs_t** out, const gss_variogram_t* vg, double mean, const double* centroids, int64_t N,
                       int32_t dim, const int64_t* path, const int64_t* dlocs, const double* zdata, int64_t nd,
                       int32_t maxneighbors, int32_t minneighbors, double radius, const double* inv_radii,
                       int32_t flags, void* stream);
/* One visiting order per realisation -- what the reference does for a RandomPath, whose `traverse` is called inside
 * solvesingle (seq.jl:99-102): paths = npaths x N cell indices, path p belongs to realisation path_base + p; stage A
 * (search, fit, weights) runs once per path and gss_sgs_realize(first_real, nreals) needs
 * path_base <= first_real and first_real + nreals <= path_base + npaths.  npaths == 1 is gss_sgs_create (every
 * realisation shares the order and one stage A serves them all). */
int32_t gss_sgs_create_paths(gss_sgs_t** out, const gss_variogram_t* vg, double mean, const double* centroids,
                             int64_t N, int32_t dim, const int64_t* paths, int64_t npaths, int64_t path_base,
                             const int64_t* dlocs, const double* zdata, int64_t nd, int32_t maxneighbors,
                             int32_t minneighbors, double radius, const double* inv_radii, int32_t flags,
                             void* stream);
int32_t gss_sgs_destroy(gss_sgs_t* h);
int32_t gss_sgs_weights(gss_sgs_t* h, int32_t* idx, int32_t* ncond, double* w, double* sigma, int32_t mem,
                        void* stream);
int32_t gss_sgs_realize(gss_sgs_t* h, uint64_t seed, int64_t first_real, int64_t nreals, const double* noise,
                        double* out, int32_t mem, void* stream);

/* ---- noise (test support): the Philox streams used above, n values for one realisation ----- */
int32_t gss_philox_uniform(uint64_t seed, int64_t real, int64_t n, double* out, int32_t mem, void* stream);
int32_t gss_philox_normal(uint64_t seed, int64_t real, int64_t n, double* out, int32_t mem, void* stream);

/* ---- dense FP64 building blocks on the MFMA path (exported for unit tests) ------------------
 * column-major, lower triangle; potrf overwrites the lower triangle of a with L; trtri writes
 * inv(L) (lower) to w.  Device pointers only. */
int32_t gss_dev_potrf(double* a, int64_t n, int64_t lda, void* stream);
/* a (n x n, full) <- unit lower-triangular L of the partial-pivot LU P a = L U (LAPACK getrf pivoting rule) */
int32_t gss_dev_getrf_l(double* a, int64_t n, int64_t lda, void* stream);
int32_t gss_dev_trtri(const double* l, int64_t n, int64_t ldl, double* w, int64_t ldw, void* stream);
/* factor and inverse in one go (the fit of gss_krig_fit and of gss_lugs_create): lower triangle of a <- L,
 * w <- inv(L) with zeros above the diagonal */
int32_t gss_dev_potrf_inverse(double* a, int64_t n, int64_t lda, double* w, int64_t ldw, void* stream);
/* D = alpha * A * B + beta * D with arbitrary element strides (A is M x K, B is K x N) */
int32_t gss_dev_gemm(int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t sa_i,
                     int64_t sa_k, const double* B, int64_t sb_k, int64_t sb_j, double beta, double* D,
                     int64_t sd_i, int64_t sd_j, int32_t lower_only, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GSS_H */
