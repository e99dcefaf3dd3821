"""Solver front-ends with the reference's parameter surface and `solve(problem, solver)` entry.

    KrigingSolver  <- /root/reference/src/estimation/krig.jl:64-234  (+ ui.jl:11-50)
    IDWSolver      <- /root/reference/src/estimation/idw.jl:49-153
    LWRSolver      <- /root/reference/src/estimation/lwr.jl:53-158
    SGS            <- /root/reference/src/simulation/sgs.jl:45-89 + seq.jl:42-141
    FFTGS          <- /root/reference/src/simulation/fft.jl:51-198
    LUGS           <- /root/reference/src/simulation/lu.jl:67-224

This is host logic only (parameter handling, missing values, model choice, sharding); all
arithmetic is delegated to the engine (gfx950 kernels through the C-ABI).  Julia's
`Solver(:z => (variogram=..., maxneighbors=3))` becomes `Solver(("z", dict(variogram=..., maxneighbors=3)))`
or `Solver(z=dict(...))`; joint parameters `(:z, :y) => (correlation=0.95,)` become
`(("z", "y"), dict(correlation=0.95))`.
"""
from __future__ import annotations

import warnings
from typing import Dict, Optional

import numpy as np

from . import parallel
from .engine import EDK, OK, SK, UK, default_engine
from .geo import Composition, Ensemble, GeoTable, PointSet, georef, parent, parentindices
from .problems import EstimationProblem, SimulationProblem
from .variograms import GaussianVariogram, MetricBall

_GLOBAL_KEYS = {"rng", "threads", "init", "engine", "share", "mask"}


class _Solver:
    PARAMS: Dict[str, object] = {}
    JPARAMS: Dict[str, object] = {}
    GLOBALS: Dict[str, object] = {}

    def __init__(self, *pairs, **kw):
        self.vparams: Dict[str, dict] = {}
        self.jparams: Dict[frozenset, dict] = {}
        self.globals = dict(self.GLOBALS)
        for k in list(kw):
            if k in self.GLOBALS or k in _GLOBAL_KEYS:
                self.globals[k] = kw.pop(k)
        items = list(pairs) + list(kw.items())
        for key, val in items:
            val = dict(val)
            if isinstance(key, (tuple, list, frozenset)):
                bad = set(val) - set(self.JPARAMS)
                if bad:
                    raise ValueError(f"invalid joint parameters {sorted(bad)}")
                self.jparams[frozenset(key)] = {**self.JPARAMS, **val}
                self._jorder = getattr(self, "_jorder", {})
                self._jorder[frozenset(key)] = tuple(key)
            else:
                bad = set(val) - set(self.PARAMS)
                if bad:
                    raise ValueError(f"invalid parameters {sorted(bad)} for variable {key}")
                self.vparams[key] = {**self.PARAMS, **val}

    def params(self, var: str) -> dict:
        return self.vparams.get(var, dict(self.PARAMS))

    def covariables(self, problem):
        """[DEP] GeoStatsBase.covariables: variables tied by joint parameters form one group."""
        names = list(problem.variables)
        groups, seen = [], set()
        for key in self.jparams:
            grp = tuple(v for v in getattr(self, "_jorder", {}).get(key, tuple(key)) if v in names)
            if len(grp) == len(key):
                groups.append(grp)
                seen.update(grp)
        for v in names:
            if v not in seen:
                groups.append((v,))
        return groups

    @property
    def engine(self):
        return self.globals.get("engine") or default_engine()

    def _share(self, default):
        """Multi-rank policy for the preprocess state: "broadcast" (rank 0 computes, RCCL broadcast to the peers) or
        "recompute" (every rank computes its own copy) -- SURVEY.md section 8e offers both, picked by size."""
        return self.globals.get("share") or default


def _seed_from(rng) -> int:
    """`rng` may be an int seed, a numpy Generator (consumed once per solve) or None."""
    if rng is None:
        return int(np.random.default_rng().integers(0, 2 ** 63 - 1))
    if isinstance(rng, (int, np.integer)):
        return int(rng)
    return int(rng.integers(0, 2 ** 63 - 1))


def _run_state(solver, problem):
    """Realisation bookkeeping shared by `solve` (batched) and `solvesingle` (GeoStatsBase's loop calls it with four
    positional arguments and NO realisation index -- test/dummy.jl:22, fft.jl:145, lu.jl:171, seq.jl:76): the seed is
    drawn from `rng` once per preprocess and every covariable group counts its own calls, so the k-th call of the
    loop produces realisation k - 1 -- exactly what `realize(seed, 0, nreals)` produces in one device call."""
    return dict(seed=_seed_from(solver.globals.get("rng")), next={g: 0 for g in solver.covariables(problem)},
                vindex={v: i for i, v in enumerate(problem.variables)})


def _next_real(preproc, conames) -> int:
    run = preproc["_run"]
    r = run["next"][tuple(conames)]
    run["next"][tuple(conames)] = r + 1
    return r


def simulate_with_generic_loop(problem, solver):
    """[DEP] GeoStatsBase `solve(::SimulationProblem, ::SimulationSolver)` as the reference's solvers
    are driven by it: `preprocess` once, then `solvesingle(problem, covars, solver, preproc)` nreals times per
    covariable group, no index passed.  The product's `solve` batches the same realisations into one device call;
    tests/test_gpu_generic_loop.py checks that both give the same ensemble."""
    preproc = solver.preprocess(problem)
    reals = {v: [] for v in problem.variables}
    for conames in solver.covariables(problem):
        for _ in range(problem.nreals):
            out = solver.solvesingle(problem, conames, preproc)
            for v in conames:
                reals[v].append(out[v])
    return Ensemble(problem.domain, reals)


def _distance(p):
    """The search metric only matters without a neighbourhood (searcher_ui, ui.jl:25-31)."""
    from ._lib import metric_spec
    d = p["distance"]
    metric_spec(d)                       # validates / raises NotImplementedError for unknown metrics
    return None if p["neighborhood"] is not None else d


def _nearest_cells(engine, pdom, cent, x):
    """Index of the domain element whose centroid is nearest to each row of `x` (`KNearestSearch(domain, 1)`:
    fft.jl:129-132, NearestInit of lu.jl:86 / seq.jl:85).  On a full CartesianGrid the answer is arithmetic -- the cell
    that contains the point or one of its neighbours along an axis -- decided by the search's own rule (squared
    distance accumulated in dimension order, ties to the lower index), so no search index over millions of cells is
    built (33 ms for 128^3 cells); views and point sets go through the search."""
    import itertools
    g = parent(pdom) if pdom is not None else None
    x = np.asarray(x, dtype=np.float64)
    if g is None or not hasattr(g, "dims") or parentindices(pdom) is not None or x.shape[0] == 0:
        c = cent() if callable(cent) else cent
        if getattr(c, "is_cuda", False):
            import torch
            x = torch.as_tensor(x, device="cuda")
        idx, _ = engine.knn_search(c, x, 1)
        idx = idx[:, 0]
        return idx.cpu().numpy() if hasattr(idx, "cpu") else np.asarray(idx)
    d = len(g.dims)
    x = x.reshape(-1, d)
    axes = [g.origin[a] + (np.arange(g.dims[a]) + 0.5) * g.spacing[a] for a in range(d)]     # CartesianGrid.centroids
    with np.errstate(invalid="ignore"):
        base = [np.clip(np.nan_to_num(np.floor((x[:, a] - g.origin[a]) / g.spacing[a])), 0, g.dims[a] - 1).astype(np.int64)
                for a in range(d)]
    best_d = np.full(x.shape[0], np.inf)
    best_i = np.full(x.shape[0], np.iinfo(np.int64).max, dtype=np.int64)
    for offs in itertools.product((-1, 0, 1), repeat=d):
        ia = [np.clip(base[a] + offs[a], 0, g.dims[a] - 1) for a in range(d)]
        acc = np.zeros(x.shape[0])
        lin = np.zeros(x.shape[0], dtype=np.int64)
        stride = 1
        for a in range(d):
            t = axes[a][ia[a]] - x[:, a]
            acc = acc + t * t
            lin = lin + ia[a] * stride
            stride *= g.dims[a]
        better = (acc < best_d) | ((acc == best_d) & (lin < best_i))
        best_d = np.where(better, acc, best_d)
        best_i = np.where(better, lin, best_i)
    return best_i


def _initbuff(engine, init, cent, pdata, var, pdom=None):
    """`initbuff(domain, vars, init; data)` (lu.jl:86, seq.jl:85; [DEP] GeoStatsBase): which cells receive which data.
    "nearest" (NearestInit, the default): every non-missing datum goes to the cell whose centroid is nearest, later
    rows overwrite earlier ones; ("explicit", orig, dest) (ExplicitInit): row orig[i] of the data goes to cell dest[i]
    (0-based here).  Returns the occupied cells in ascending order and their values."""
    if pdata is None or var not in pdata.table:
        return np.empty(0, dtype=np.int64), np.empty(0)
    zv = np.asarray(pdata[var], dtype=np.float64)
    buff = {}
    if init is None or (isinstance(init, str) and init == "nearest"):
        keep = ~np.isnan(zv)
        idx = _nearest_cells(engine, pdom, cent, pdata.domain.centroids()[keep])
        for j, v in zip(idx, zv[keep]):
            buff[int(j)] = v
    elif isinstance(init, (tuple, list)) and len(init) == 3 and init[0] == "explicit":
        orig, dest = np.asarray(init[1], dtype=np.int64), np.asarray(init[2], dtype=np.int64)
        if orig.shape != dest.shape or orig.ndim != 1:
            raise ValueError("ExplicitInit: orig and dest must be index vectors of one length")
        if orig.size and (orig.min() < 0 or orig.max() >= zv.size or dest.min() < 0 or dest.max() >= cent.shape[0]):
            raise ValueError("ExplicitInit: index outside the data or the domain")
        for i, j in zip(orig, dest):
            if not np.isnan(zv[i]):
                buff[int(j)] = zv[i]
    else:
        raise NotImplementedError(f"init={init!r}: 'nearest' or ('explicit', orig, dest)")
    dlocs = np.array(sorted(buff), dtype=np.int64)
    return dlocs, np.array([buff[j] for j in dlocs])


def multigrid_order(dims):
    """Coarse-to-fine visiting order of a Cartesian grid ([DEP] Meshes `MultiGridPath`, test/estimation/krig.jl:87):
    cells whose indices are all multiples of the largest power-of-two stride first, then stride / 2, ... down to 1,
    each level in linear (first axis fastest) order.  The package's exact order is not vendored; an estimation
    solver only permutes its output columns by it (krig.jl:179-183), a simulation solver conditions on it."""
    dims = tuple(int(d) for d in dims)
    idx = np.indices(dims[::-1])[::-1]                      # idx[a]: index along axis a, element order (axis 0 fastest)
    flat = [i.reshape(-1) for i in idx]
    stride = 1
    while stride * 2 < max(dims):
        stride *= 2
    level = np.zeros(flat[0].shape, dtype=np.int64)
    s, lv = stride, 0
    assigned = np.zeros(flat[0].shape, dtype=bool)
    while s >= 1:
        on = np.ones_like(assigned)
        for f in flat:
            on &= (f % s) == 0
        new = on & ~assigned
        level[new] = lv
        assigned |= new
        s //= 2
        lv += 1
    return np.argsort(level, kind="stable")


def _path_order(path, n, domain=None):
    """Traversal order of the estimation loop (krig.jl:179, idw.jl:112, lwr.jl:115): None for LinearPath, else a
    permutation of 0..n-1 -- ("random", seed), "multigrid" (Cartesian grids) or an explicit visiting order.  The
    reference stores its results in traversal order (`pred = map(inds) do ind ...`, krig.jl:180-183), so the columns
    are permuted accordingly."""
    if path is None or (isinstance(path, str) and path == "linear"):
        return None
    if isinstance(path, tuple) and len(path) == 2 and path[0] == "random":
        return np.random.default_rng(path[1]).permutation(n)
    if isinstance(path, str) and path == "multigrid":
        g = parent(domain) if domain is not None else None
        if g is None or not hasattr(g, "dims") or parentindices(domain) is not None:
            raise ValueError("path='multigrid' needs a CartesianGrid domain")
        return multigrid_order(g.dims)
    if isinstance(path, str):
        raise NotImplementedError(f"path {path!r}: give 'linear', 'multigrid', ('random', seed) or a visiting order")
    order = np.asarray(path, dtype=np.int64)
    if order.shape != (n,) or not np.array_equal(np.sort(order), np.arange(n)):
        raise ValueError("path must be a permutation of the domain elements")
    return order


def _mask_missing(a, st):
    """`missing` (krig.jl:213-214, idw.jl:123-124) as NaN, in place: the arrays are the call's own outputs, and a pass of
    `np.where` over 10^6 - 10^7 estimates that almost never hold a missing value costs more than the device call."""
    bad = st != 0
    if bad.any():
        a[..., bad] = np.nan
    return a


def _ball(neighborhood):
    if neighborhood is None:
        return None, None
    if not isinstance(neighborhood, MetricBall):
        raise TypeError("neighborhood must be a MetricBall")
    if neighborhood.isotropic:
        return neighborhood.radii[0], None
    return None, neighborhood.radii


# ------------------------------------------------------------------------------------------
# ui.jl
# ------------------------------------------------------------------------------------------
def kriging_ui(domain, variogram, mean, degree, drifts):
    """ui.jl:40-50: drifts > degree > mean > ordinary."""
    if drifts is not None:
        return EDK
    if degree is not None:
        return UK
    if mean is not None:
        return SK
    return OK


def searcher_ui(domain, maxneighbors, metric, neighborhood):
    """ui.jl:11-32 -> (kind, nmax) with the reference's warning text."""
    nelem = domain.nelements()
    if maxneighbors is None:
        nmax = nelem
    elif maxneighbors < 1 or maxneighbors > nelem:
        warnings.warn(f"Invalid maximum number of neighbors. Adjusting to {nelem}...")
        nmax = nelem
    else:
        nmax = maxneighbors
    return ("KNearestSearch" if neighborhood is None else "KBallSearch"), nmax


# ------------------------------------------------------------------------------------------
# KrigingSolver
# ------------------------------------------------------------------------------------------
class KrigingSolver(_Solver):
    PARAMS = dict(variogram=GaussianVariogram(), mean=None, degree=None, drifts=None, minneighbors=1,
                  maxneighbors=None, neighborhood=None, distance="euclidean", path="linear",   # krig.jl:64-74
                  # not a parameter of the reference, which has no choice: predictprob gets the cell on a grid
                  # (krig.jl:180) and its dependencies regularise over it.  "point" (default) estimates at the
                  # centroids; ("block", nsub) regularises over the cells of a CartesianGrid by
                  # the midpoint rule with nsub points per axis (gss.h, gss_krig_set_block_support), global and
                  # moving neighbourhoods alike
                  support="point")

    def preprocess(self, problem: EstimationProblem):
        """krig.jl:76-128."""
        pre = {}
        ddom = problem.data.domain
        coords = ddom.centroids()
        for var in problem.variables:
            p = self.params(var)
            z = np.asarray(problem.data[var], dtype=np.float64)
            inds = np.flatnonzero(~np.isnan(z))                       # krig.jl:97
            if inds.size == 0:
                raise AssertionError(f"all samples of {var} are missing, aborting...")   # krig.jl:100-102
            _distance(p)
            _path_order(p["path"], problem.domain.nelements(), problem.domain)
            vdom = PointSet(coords[inds])
            variant = kriging_ui(problem.domain, p["variogram"], p["mean"], p["degree"], p["drifts"])
            kind, nmax = searcher_ui(vdom, p["maxneighbors"], p["distance"], p["neighborhood"])
            pre[var] = dict(x=vdom.coords, z=z[inds], variant=variant, minneighbors=p["minneighbors"],
                            maxneighbors=p["maxneighbors"], nmax=nmax, searcher=kind, params=p)
        return pre

    def solve(self, problem: EstimationProblem, gather: bool = True):
        """krig.jl:130-164 with the domain points sharded over ranks (parallel.shard_range).  With several ranks each
        one returns the estimates of its own block of domain points (no collective on the data path);
        `gather=True` (default: the reference's `solve` returns the whole solution) reassembles the full table on every
        rank; `gather=False` leaves every rank with its own block of points, in domain order."""
        pre = self.preprocess(problem)
        pdom = problem.domain
        m = pdom.nelements()
        rank, ws = parallel.world()
        lo, hi = parallel.shard_range(m, rank, ws)
        need_host = any(pre[v]["variant"] == EDK for v in problem.variables)
        xdom = _domain_points(self.engine, pdom, lo, hi, need_host)
        cols = {}
        # Variables of one problem that share everything but their values (same samples, same variogram object, same
        # variant / mean / degree / support, global neighbourhood) share the kriging system: the first one is fitted and
        # predicted, the others are one batched product with the weights it implies (gss_krig_predict_global_batch, the
        # conditional-FFTGS pattern) and take its variances -- the reference repeats the fit and the solves per variable
        # (krig.jl:141-161), the numbers are the same.
        shared = {}
        for var in problem.variables:
            q = pre[var]
            p = q["params"]
            skey = None
            if (q["maxneighbors"] is None and q["variant"] != EDK and p.get("support", "point") == "point"
                    and hasattr(self.engine.Krig, "predict_global_batch")):
                skey = (id(p["variogram"]), q["variant"], p["mean"], p["degree"], repr(p.get("support", "point")),
                        q["x"].shape, q["x"].tobytes() if q["x"].size <= 1 << 20 else id(q["x"]))
            if skey is not None and skey in shared and hi > lo:
                h0, var0, st0 = shared[skey]
                zrow = np.ascontiguousarray(q["z"], dtype=np.float64)[None, :]
                if hasattr(xdom, "is_cuda"):
                    import torch
                    zrow = torch.as_tensor(zrow, device=xdom.device)
                mu = _host(h0.predict_global_batch(xdom, zrow))[0]
                mu = _mask_missing(mu, st0)
                var_ = var0
                if gather and ws > 1:
                    mu = parallel.all_gather_concat(mu, m)
                order = _path_order(p["path"], m, pdom)
                if order is not None and (gather or ws == 1):
                    mu = mu[order]
                cols[var] = mu
                cols[f"{var}_variance"] = var_.copy()
                continue
            drift_data = drift_dom = None
            if q["variant"] == EDK:
                drift_data = np.stack([[f(c) for f in p["drifts"]] for c in q["x"]]).astype(np.float64)
                drift_dom = np.stack([[f(c) for f in p["drifts"]] for c in xdom]).astype(np.float64)
            exact = q["maxneighbors"] is None                          # krig.jl:151
            mk = lambda compute: self.engine.Krig(p["variogram"], q["variant"], q["x"], q["z"], mean=p["mean"],  # noqa: E731
                                                  degree=p["degree"], drift_data=drift_data, factor=compute,
                                                  async_fit=exact)   # fit and the first assembly side by side
            # the fit (krig.jl:176) is replicated by default: below n ~ 2 000 recomputing the factor on every GPU is
            # cheaper than any collective; share="broadcast" sends rank 0's factor instead
            h = parallel.replicate_state(mk, self._share("recompute")) if exact else mk(False)
            sup = p.get("support", "point")
            if sup != "point":
                nsub = 3 if sup == "block" else int(sup[1])
                g = parent(pdom)
                if not hasattr(g, "spacing"):
                    raise ValueError("support='block' needs a Cartesian grid domain (the cells to average over)")
                h.set_block_support(g.spacing, nsub)
            try:
                if hi > lo:
                    if exact:
                        mu, var_, st = h.predict_global(xdom, drift_dom)                 # krig.jl:166-186
                    else:
                        radius, radii = _ball(p["neighborhood"])
                        mu, var_, st = h.predict_knn(xdom, q["nmax"], q["minneighbors"], radius, radii, drift_dom,
                                                     distance=_distance(p))
                    mu, var_, st = _host(mu), _host(var_), _host(st)
                else:
                    mu, var_, st = np.empty(0), np.empty(0), np.empty(0, dtype=np.uint8)
            except BaseException:
                h.close()
                for h0, _, _ in shared.values():
                    h0.close()
                raise
            keep = skey is not None and hi > lo and len(problem.variables) > 1
            if not keep:
                h.close()
            mu = _mask_missing(mu, st)                                 # `missing` krig.jl:213-214
            var_ = _mask_missing(var_, st)
            if gather and ws > 1:
                mu = parallel.all_gather_concat(mu, m)
                var_ = parallel.all_gather_concat(var_, m)
            order = _path_order(p["path"], m, pdom)
            if order is not None and (gather or ws == 1):              # results in traversal order, krig.jl:179-183
                mu, var_ = mu[order], var_[order]
            if keep:
                shared[skey] = (h, var_, st)                           # (variances already gathered and ordered)
            cols[var] = mu
            cols[f"{var}_variance"] = var_                             # krig.jl:160
        for h0, _, _ in shared.values():
            h0.close()
        if gather or ws == 1:
            return georef(cols, pdom)                                  # krig.jl:163
        return georef(cols, PointSet(_host(xdom)))


# ------------------------------------------------------------------------------------------
# IDWSolver / LWRSolver
# ------------------------------------------------------------------------------------------
class ExpWeight:
    """Weight function h -> exp(-a h^p); the reference default is ExpWeight(3, 2) (lwr.jl:58)."""

    def __init__(self, a=3.0, p=2.0):
        self.a, self.p = float(a), float(p)

    def spec(self):
        return (0, self.a, self.p)

    def __call__(self, h):
        return np.exp(-self.a * np.asarray(h, dtype=np.float64) ** self.p)


class TricubeWeight:
    """Cleveland's tricube h -> (1 - h^3)^3."""

    def spec(self):
        return (1, 0.0, 0.0)

    def __call__(self, h):
        return (1.0 - np.asarray(h, dtype=np.float64) ** 3) ** 3


class _NeighborEstimator(_Solver):
    """Shared body of idw.jl:58-153 and lwr.jl:61-158 (they differ in the per-point arithmetic only)."""
    AUX = ""

    def _estimate(self, p, x, z, xdom, nmax, radius, radii):
        raise NotImplementedError

    def solve(self, problem: EstimationProblem, gather: bool = True):
        pdom = problem.domain
        coords = problem.data.domain.centroids()
        m = pdom.nelements()
        rank, ws = parallel.world()
        lo, hi = parallel.shard_range(m, rank, ws)
        xdom = _domain_points(self.engine, pdom, lo, hi)
        cols, aux = {}, {}
        # Scalar variables with the same parameters and the same valid samples share ONE search and ONE weight vector per
        # point (value columns of gss_idw_predict_cols / gss_lwr_predict_cols): the first of them computes all of them.
        # The reference repeats search and weights per variable (idw.jl:111-142); the numbers are the same.
        def _key(v):
            pv = self.params(v)
            cv = problem.data[v]
            if getattr(cv, "dtype", None) == object:
                return None
            valid = np.flatnonzero(~np.isnan(np.asarray(cv, dtype=np.float64)))
            ident = lambda o: o if isinstance(o, (int, float, str, type(None))) else id(o)   # noqa: E731
            return tuple(sorted((k, ident(x)) for k, x in pv.items())) + (valid.tobytes(),)
        keys = {v: _key(v) for v in problem.variables}
        batched = {}                                                     # var -> (mu, ax, st) computed with its group
        for var in problem.variables:
            p = self.params(var)
            col = problem.data[var]
            # a column of compositions (test/estimation/idw.jl:47-65): the loop is generic over the value type
            # (idw.jl:138 `sum(ws[i] * vs[i])` = perturbation of powers), and both operations are linear in the
            # log-parts -- one device value column per part, ONE search and ONE weight vector for all of them
            comp = getattr(col, "dtype", None) == object and any(isinstance(v, Composition) for v in col)
            if comp:
                keep = np.array([isinstance(v, Composition) and bool(np.all(np.isfinite(v.parts)) and np.all(v.parts > 0))
                                 for v in col])
                inds = np.flatnonzero(keep)
                zall = None
            else:
                zall = np.asarray(col, dtype=np.float64)
                inds = np.flatnonzero(~np.isnan(zall))                   # idw.jl:77, lwr.jl:80
            n = inds.size
            assert n > 0, "estimation requires data"                      # idw.jl:95
            _distance(p)
            order = _path_order(p["path"], m, pdom)
            nmin = p["minneighbors"]
            nmax = n if p["maxneighbors"] is None else min(p["maxneighbors"], n)      # idw.jl:93
            self._check(p)
            assert nmin <= nmax, "invalid min/max number of neighbors"    # idw.jl:97
            vdom = PointSet(coords[inds])
            _, k = searcher_ui(vdom, p["maxneighbors"], p["distance"], p["neighborhood"])   # idw.jl:100
            radius, radii = _ball(p["neighborhood"])
            if comp:
                zin = np.log(np.stack([col[i].parts for i in inds], axis=1))     # (parts, n): log-parts as value columns
            else:
                zin = zall[inds]
            group = [v for v in problem.variables if keys[v] is not None and keys[v] == keys[var]] if not comp else [var]
            if var in batched:
                mu, ax, st = batched.pop(var)
            elif hi > lo and len(group) > 1:
                zs = np.stack([np.asarray(problem.data[v], dtype=np.float64)[inds] for v in group])
                mus, ax, st = self._estimate(p, vdom.coords, zs, xdom, k, nmin, radius, radii)
                mus, ax, st = _host(mus), _host(ax), _host(st)
                for j, v in enumerate(group):
                    batched[v] = (np.asarray(mus[j]), ax, st)
                mu, ax, st = batched.pop(var)
            elif hi > lo:
                mu, ax, st = self._estimate(p, vdom.coords, zin, xdom, k, nmin, radius, radii)
                mu, ax, st = _host(mu), _host(ax), _host(st)
            else:
                mu = np.empty((zin.shape[0], 0)) if comp else np.empty(0)
                ax, st = np.empty(0), np.empty(0, dtype=np.uint8)
            mu = _mask_missing(mu, st)                                    # `missing`
            ax = _mask_missing(ax, st)
            if gather and ws > 1:
                mu = (np.stack([parallel.all_gather_concat(r, m) for r in mu]) if comp
                      else parallel.all_gather_concat(mu, m))
                ax = parallel.all_gather_concat(ax, m)
            if order is not None and (gather or ws == 1):              # results in traversal order, idw.jl:112-113
                mu, ax = mu[..., order], ax[order]
            if comp:                                                    # back from the log-parts; `missing` stays None
                parts = np.exp(mu)
                out = np.empty(parts.shape[1], dtype=object)
                for j in range(parts.shape[1]):
                    out[j] = Composition(parts[:, j]) if np.all(np.isfinite(parts[:, j])) else None
                mu = out
            cols[var] = mu
            aux[f"{var}_{self.AUX}"] = ax
        cols.update(aux)                                                  # (; mus..., sigmas...) idw.jl:152
        if gather or ws == 1:
            return georef(cols, pdom)
        return georef(cols, PointSet(_host(xdom)))

    def _check(self, p):
        pass


class IDWSolver(_NeighborEstimator):
    PARAMS = dict(minneighbors=1, maxneighbors=None, neighborhood=None, distance="euclidean", exponent=1,
                  path="linear")                                                               # idw.jl:49-56
    AUX = "distance"                                                                           # idw.jl:149

    def _check(self, p):
        assert p["exponent"] > 0, "exponent must be positive"                                  # idw.jl:96

    def _estimate(self, p, x, z, xdom, k, nmin, radius, radii):
        return self.engine.idw(x, z, xdom, k, nmin, float(p["exponent"]), radius, radii, distance=_distance(p))


class LWRSolver(_NeighborEstimator):
    PARAMS = dict(minneighbors=1, maxneighbors=None, neighborhood=None, distance="euclidean", weightfun=None,
                  path="linear")                                                               # lwr.jl:53-60
    AUX = "variance"                                                                           # lwr.jl:154

    def _estimate(self, p, x, z, xdom, k, nmin, radius, radii):
        wf = p["weightfun"] or ExpWeight()
        if not hasattr(wf, "spec"):
            # an arbitrary callable h -> weight (lwr.jl:58): it cannot cross the C-ABI, so the weights are evaluated on
            # the host between the device's search and the device's normal equations (engine.lwr_callable)
            if not callable(wf) or not hasattr(self.engine, "lwr_callable"):
                raise NotImplementedError("weightfun must be ExpWeight(a, p), TricubeWeight() or a callable h -> weight")
            return self.engine.lwr_callable(x, z, _host(xdom), k, nmin, wf, radius, radii, distance=_distance(p))
        return self.engine.lwr(x, z, xdom, k, nmin, wf.spec(), radius, radii, distance=_distance(p))


# ------------------------------------------------------------------------------------------
# FFTGS
# ------------------------------------------------------------------------------------------
class FFTGS(_Solver):
    PARAMS = dict(variogram=GaussianVariogram(), mean=0.0, minneighbors=1, maxneighbors=None, neighborhood=None,
                  distance="euclidean")                                                        # fft.jl:51-60
    GLOBALS = dict(threads=None, rng=None)

    def preprocess(self, problem: SimulationProblem):
        """fft.jl:62-143."""
        pdom = problem.domain
        pgrid = parent(pdom)
        if not hasattr(pgrid, "dims"):
            raise ValueError("FFTGS is limited to simulations on Cartesian grids")
        pre = {}
        for (var,) in [g for g in self.covariables(problem)]:
            p = self.params(var)
            vg = p["variogram"]
            if not vg.isstationary():
                raise ValueError("variogram model must be stationary")              # fft.jl:91-93
            # fft.jl:96-103 on rank 0, state broadcast to the peers (preprocess once, realise many: fft.jl:62,145)
            h = parallel.replicate_state(
                lambda compute: self.engine.FFTGS(vg, pgrid.dims, pgrid.spacing, p["mean"],
                                                  **({} if compute else {"spectrum": False})),
                self._share("broadcast"))
            zbar = krig = dinds = cdev = None
            pdata = problem.data
            if pdata is not None and var in pdata.table:                             # fft.jl:106-135
                xd = pdata.domain.centroids()
                zd = np.asarray(pdata[var], dtype=np.float64)
                cent = None
                krig = KrigingSolver((var, dict(variogram=vg, mean=p["mean"], minneighbors=p["minneighbors"],
                                                maxneighbors=p["maxneighbors"], neighborhood=p["neighborhood"],
                                                distance=p["distance"])), engine=self.globals.get("engine"))
                if p["maxneighbors"] is None and getattr(self.engine, "device_resident", False):
                    # global neighbourhood on the device engine: centroids go to HBM once and serve the kriging of
                    # the data (fft.jl:125), the cell lookup (fft.jl:129-132) and every realisation in solve()
                    import torch
                    cdev = _centroids_device(pdom)
                    keep = ~np.isnan(zd)                                              # krig.jl:97
                    if not keep.any():
                        raise AssertionError(f"all samples of {var} are missing, aborting...")
                    kh = self.engine.Krig(vg, SK, xd[keep], zd[keep], mean=p["mean"])
                    try:
                        # only the mean is used (fft.jl:126): the batched means-only path, no quadratic form
                        zbar = kh.predict_global_batch(cdev, torch.as_tensor(zd[keep][None, :], device="cuda"))[0]
                    finally:
                        kh.close()
                    found = _nearest_cells(self.engine, pdom, lambda: cdev, xd)      # fft.jl:129-132
                else:
                    cent = pdom.centroids()
                    kdom = PointSet(cent)
                    ksol = _solve_local(krig, georef({var: zd}, xd), kdom, var)       # fft.jl:125
                    zbar = ksol[var]
                    found = _nearest_cells(self.engine, pdom, kdom.coords, xd)        # fft.jl:129-132
                _, first = np.unique(found, return_index=True)
                dinds = found[np.sort(first)]
                pre[var] = dict(cent=cent, cdev=cdev)
            pre[var] = dict(dict(cent=None, cdev=None), **pre.get(var, {}), vg=vg, mean=p["mean"], handle=h, zbar=zbar,
                            krig=krig, dinds=dinds)
        pre["_run"] = _run_state(self, problem)
        return pre

    def _block(self, problem, pre, var, lo, count):
        """Realisations lo .. lo+count-1 of `var` on the problem domain (fft.jl:145-198), shape (count, npts)."""
        pdom = problem.domain
        inds = parentindices(pdom)
        q = pre[var]
        seed = pre["_run"]["seed"] + pre["_run"]["vindex"][var]
        if count <= 0:
            return np.empty((0, pdom.nelements()))
        cond = q["krig"] is not None
        if cond and q["krig"].params(var)["maxneighbors"] is None and getattr(self.engine, "device_resident", False):
            return self._condition_on_device(q, q["cent"], q["dinds"], seed, lo, count, inds)
        zu = q["handle"].realize(seed, lo, count, inds=inds)
        if not cond:
            return zu
        cent, dinds = q["cent"], q["dinds"]                                       # fft.jl:176-192
        if q["krig"].params(var)["maxneighbors"] is None:
            # one kriging system (same locations) serves every realisation: factor once, batch the data
            h = self.engine.Krig(q["vg"], SK, cent[dinds], zu[0, dinds], mean=q["mean"])
            try:
                zbar_u = h.predict_global_batch(cent, np.ascontiguousarray(zu[:, dinds]))
            finally:
                h.close()
            return q["zbar"][None, :] + (zu - zbar_u)                             # fft.jl:191
        out = np.empty_like(zu)
        for r in range(zu.shape[0]):
            kdat = georef({var: zu[r, dinds]}, cent[dinds])
            zbar_u = _solve_local(q["krig"], kdat, PointSet(cent), var)[var]
            out[r] = q["zbar"] + (zu[r] - zbar_u)
        return out

    def solvesingle(self, problem: SimulationProblem, covars, preproc):
        """fft.jl:145-198 with the reference's signature: one realisation per call, no index (see _run_state)."""
        r = _next_real(preproc, covars)
        return {var: self._block(problem, preproc, var, r, 1)[0] for var in covars}

    def solve(self, problem: SimulationProblem, gather: bool = True):
        """GeoStatsBase's realisation loop ([DEP]) batched: realisations are sharded
        over ranks, each rank produces its block in one device call (fft.jl:145-198) and returns an Ensemble of its
        realisations -- all of them on every rank by default, as the reference's `solve` returns them; `gather=False`
        keeps the rank's own block (no collective on the data path)."""
        pre = self.preprocess(problem)
        rank, ws = parallel.world()
        lo, hi = parallel.shard_range(problem.nreals, rank, ws)
        reals = {}
        for var in problem.variables:
            zu = self._block(problem, pre, var, lo, hi - lo)
            pre[var]["handle"].close()
            if gather and ws > 1:
                zu = parallel.all_gather_concat(zu, problem.nreals)
            reals[var] = [zu[r] for r in range(zu.shape[0])]
        return Ensemble(problem.domain, reals)


def _domain_points(engine, pdom, lo, hi, need_host=False):
    """Coordinates of domain elements lo .. hi-1 for an estimation call: for (views of) Cartesian grids on an engine that
    takes device arrays, the tensor product of the per-axis centroids is formed in HBM (bit-identical values) -- 10^7
    cells are 240 MB that would otherwise be generated on the host and copied; everything else, and `need_host`
    (external drift functions are evaluated point by point in Python), gives the host array."""
    import os
    least = int(os.environ.get("GSS_SOLVE_DEVICE_POINTS", "200000"))      # elements from which the device forms them; 0: never
    if (not need_host and least > 0 and getattr(engine, "device_resident", False) and hasattr(parent(pdom), "spacing")
            and pdom.nelements() >= least):
        c = _centroids_device(pdom)
        return c[lo:hi] if (lo, hi) != (0, c.shape[0]) else c
    return pdom.centroids()[lo:hi]


def _host(a):
    """numpy view of an engine result.  Device tensors come back in one copy; large ones through page-locked memory
    (torch keeps such blocks for re-use), which the copy engine fills at the bus rate instead of the pageable path's
    fraction of it."""
    if not hasattr(a, "is_cuda"):
        return np.asarray(a)
    if not a.is_cuda:
        return a.numpy()
    if a.numel() * a.element_size() >= (4 << 20):
        import torch
        h = torch.empty(a.shape, dtype=a.dtype, pin_memory=True)
        h.copy_(a, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return h.numpy()
    return a.cpu().numpy()


def _centroids_device(pdom):
    """`pdom.centroids()` assembled in HBM: the per-axis coordinates are computed on the host exactly as
    CartesianGrid.centroids does (so the values are bit-identical), only the tensor product happens on the device."""
    import torch
    g = parent(pdom)
    d = len(g.dims)
    axes = [torch.as_tensor(g.origin[a] + (np.arange(g.dims[a]) + 0.5) * g.spacing[a], device="cuda") for a in range(d)]
    mesh = torch.meshgrid(*axes[::-1], indexing="ij")
    c = torch.stack([t.reshape(-1) for t in mesh[::-1]], dim=1).contiguous()
    inds = parentindices(pdom)
    return c if inds is None else c.index_select(0, torch.as_tensor(np.asarray(inds, dtype=np.int64), device="cuda"))


def _fftgs_condition_on_device(self, q, cent, dinds, seed, first, count, inds):
    """fft.jl:176-192 with the realisations, the kriged fields and their combination kept in HBM: the only
    transfer is the conditioned block going out (torch is the device-memory plumbing)."""
    import torch
    zu = q["handle"].realize(seed, first, count, inds=inds, device=True)
    ddev = torch.as_tensor(np.asarray(dinds, dtype=np.int64), device=zu.device)
    zdat = zu.index_select(1, ddev).contiguous()
    xdat = q["cdev"].index_select(0, ddev).cpu().numpy()
    h = self.engine.Krig(q["vg"], SK, xdat, np.zeros(len(dinds)), mean=q["mean"])
    try:
        zbar_u = h.predict_global_batch(q["cdev"], zdat)
    finally:
        h.close()
    zu -= zbar_u
    zu += q["zbar"][None, :]                                                          # fft.jl:191
    from .engine import to_host
    return to_host(zu)


FFTGS._condition_on_device = _fftgs_condition_on_device


def _solve_local(krig: KrigingSolver, data: GeoTable, dom, var):
    """Nested kriging solve of the conditional path: every rank needs the full field, so no sharding."""
    saved = parallel.world
    parallel.world = lambda: (0, 1)
    try:
        return krig.solve(EstimationProblem(data, dom, var))
    finally:
        parallel.world = saved


# ------------------------------------------------------------------------------------------
# LUGS
# ------------------------------------------------------------------------------------------
class LUGS(_Solver):
    PARAMS = dict(variogram=GaussianVariogram(), mean=None, factorization="cholesky")          # lu.jl:67-74
    JPARAMS = dict(correlation=0.0)
    GLOBALS = dict(init="nearest", rng=None)

    def preprocess(self, problem: SimulationProblem):
        """lu.jl:76-169."""
        pdom = problem.domain
        cent = pdom.centroids()
        N = cent.shape[0]
        init = self.globals.get("init", "nearest")
        pre = {}
        for conames in self.covariables(problem):
            assert len(conames) in (1, 2), "invalid number of covariables"          # lu.jl:96
            co = {}
            for var in conames:
                p = self.params(var)
                vg = p["variogram"]
                assert vg.isstationary(), "variogram model must be stationary"      # lu.jl:110
                fact = p["factorization"]                                            # lu.jl:107
                if fact not in ("cholesky", "lu"):
                    raise ValueError(f"factorization={fact!r}: 'cholesky' or 'lu' (lu.jl:70)")
                dlocs, z1 = _initbuff(self.engine, init, cent, problem.data, var, pdom)   # lu.jl:86,113-114
                if p["mean"] is not None and dlocs.size > 0:
                    warnings.warn("mean can only be specified in unconditional simulation")   # lu.jl:142-144
                mu = 0.0 if p["mean"] is None else float(p["mean"])                   # lu.jl:147
                # lu.jl:124-139 on rank 0, (L22, d2) broadcast to the peers (preprocess once: lu.jl:76,171)
                co[var] = parallel.replicate_state(
                    lambda compute: self.engine.LUGS(vg, cent, dlocs, z1, mu, **({} if compute else {"factor": False}),
                                                     **({} if fact == "cholesky" else {"factorization": fact})),
                    self._share("broadcast"))
            rho = None
            if len(conames) == 2:
                rho = self.jparams[frozenset(conames)]["correlation"]                 # lu.jl:154-163
            pre[conames] = dict(handles=co, rho=rho)
        pre["_run"] = _run_state(self, problem)
        return pre

    def _block(self, pre, conames, lo, count):
        """lu.jl:171-196 for realisations lo .. lo+count-1 of one covariable group -> {var: (count, N)}."""
        q = pre[conames]
        run = pre["_run"]
        v1 = conames[0]
        y1, w1 = q["handles"][v1].realize(run["seed"] + run["vindex"][v1], lo, count)          # lu.jl:183-185
        out = {v1: y1}
        if len(conames) == 2:
            v2 = conames[1]
            y2, _ = q["handles"][v2].realize(run["seed"] + run["vindex"][v2], lo, count, rho=q["rho"], w1=w1)  # :188-193
            out[v2] = y2
        return out

    def solvesingle(self, problem: SimulationProblem, covars, preproc):
        """lu.jl:171-196 with the reference's signature: one realisation per call, no index (see _run_state)."""
        r = _next_real(preproc, covars)
        return {v: y[0] for v, y in self._block(preproc, tuple(covars), r, 1).items()}

    def solve(self, problem: SimulationProblem, gather: bool = True):
        pre = self.preprocess(problem)
        rank, ws = parallel.world()
        lo, hi = parallel.shard_range(problem.nreals, rank, ws)
        reals = {}
        for conames in self.covariables(problem):
            for v, y in self._block(pre, conames, lo, hi - lo).items():
                pre[conames]["handles"][v].close()
                if gather and ws > 1:
                    y = parallel.all_gather_concat(y, problem.nreals)
                reals[v] = [y[r] for r in range(y.shape[0])]
        return Ensemble(problem.domain, {v: reals[v] for v in problem.variables})


# ------------------------------------------------------------------------------------------
# SGS
# ------------------------------------------------------------------------------------------
SGS_PATHS_PER_HANDLE = 64      # realisations (= visiting orders) per device handle when every realisation has its own


class _SGSPlan:
    """What SGS.preprocess hands to solvesingle / solve for one variable.  A shared visiting order (LinearPath, an
    explicit order) is one device handle: stage A once, lanes = realisations.  `path=("random", seed)` is a RandomPath:
    the reference draws a new permutation in every solvesingle (seq.jl:99-102), so realisation r gets the permutation
    `default_rng([seed, r])` and the handles are built per block of realisations, on demand."""

    def __init__(self, engine, make_args, N, path_seed=None, order=None, mask_after=True, distance=None):
        self.engine, self.args, self.N = engine, make_args, N
        self.path_seed, self.order = path_seed, order
        self.kw = dict(mask_after_search=True) if mask_after else {}
        if distance not in (None, "euclidean"):
            self.kw["distance"] = distance
        self.shared = None

    def path_of(self, r):
        return np.random.default_rng([int(self.path_seed), int(r)]).permutation(self.N)

    def realize(self, seed, first, count):
        vg, cent, dlocs, zd, mean, nmax, nmin, radius, radii = self.args
        if count <= 0:
            return np.empty((0, self.N))
        if self.path_seed is None:
            if self.shared is None:
                self.shared = self.engine.SGS(vg, cent, self.order, dlocs, zd, mean, nmax, nmin, radius, radii, **self.kw)
            return self.shared.realize(seed, first, count)
        out = []
        for a in range(first, first + count, SGS_PATHS_PER_HANDLE):
            b = min(a + SGS_PATHS_PER_HANDLE, first + count)
            paths = np.stack([self.path_of(r) for r in range(a, b)])
            h = self.engine.SGS(vg, cent, paths, dlocs, zd, mean, nmax, nmin, radius, radii, path_base=a, **self.kw)
            try:
                out.append(h.realize(seed, a, b - a))
            finally:
                h.close()
        return np.concatenate(out, axis=0)

    def close(self):
        if self.shared is not None:
            self.shared.close()
            self.shared = None


class SGS(_Solver):
    """sgs.jl:45-89 on top of seq.jl:42-141.  `path` is "linear" (LinearPath), an explicit visiting order (shared by
    every realisation: the device computes the neighbour lists and simple-kriging weights of the path once) or
    ("random", seed) (RandomPath: a new permutation per realisation, as `traverse` inside the reference's
    solvesingle gives, seq.jl:99-102).
    Global `mask`: how `search!(..., mask=simulated)` (seq.jl:105) is read -- "after" (default): the searcher returns its
    k nearest cells of the whole domain and the mask keeps the simulated ones ([DEP] Meshes' KNearestSearch / KBallSearch
    as recalled; unverifiable here); "during": the k nearest among the simulated cells."""
    PARAMS = dict(variogram=GaussianVariogram(), mean=0.0, path="linear", minneighbors=1, maxneighbors=10,
                  neighborhood=None, distance="euclidean")                                     # sgs.jl:45-55
    GLOBALS = dict(init="nearest", rng=None, mask="after")

    def preprocess(self, problem: SimulationProblem):
        pdom = problem.domain
        cent = pdom.centroids()
        N = cent.shape[0]
        init = self.globals.get("init", "nearest")
        pre = {}
        for (var,) in [g for g in self.covariables(problem)]:
            p = self.params(var)
            dist = _distance(p)                                                        # seq.jl:91-98
            if dist is not None and not isinstance(dist, str) and self.globals.get("mask", "after") != "after":
                raise NotImplementedError("the haversine search distance needs mask='after' in SGS on the device "
                                          "(the masked search has no exhaustive variant)")
            path = p["path"]
            order = path_seed = None
            if path is None or (isinstance(path, str) and path == "linear"):
                order = None
            elif isinstance(path, tuple) and path[0] == "random":
                path_seed = int(path[1])
            elif isinstance(path, str) and path == "multigrid":
                order = _path_order(path, N, pdom)
            elif isinstance(path, str):
                raise NotImplementedError(f"path {path!r}: give 'linear', 'multigrid', ('random', seed) or a visiting order")
            else:
                order = np.asarray(path, dtype=np.int64)
            dlocs, zd = _initbuff(self.engine, init, cent, problem.data, var, pdom)   # seq.jl:85
            _, nmax = searcher_ui(pdom, p["maxneighbors"], p["distance"], p["neighborhood"])   # seq.jl:65
            radius, radii = _ball(p["neighborhood"])
            mask = self.globals.get("mask", "after")
            if mask not in ("after", "during"):
                raise ValueError(f"mask={mask!r}: 'after' or 'during'")
            pre[var] = _SGSPlan(self.engine, (p["variogram"], cent, dlocs, zd, float(p["mean"]), nmax,
                                              p["minneighbors"], radius, radii), N, path_seed, order, mask == "after",
                                dist)
        pre["_run"] = _run_state(self, problem)
        return pre

    def solvesingle(self, problem: SimulationProblem, covars, preproc):
        """seq.jl:76-141 with the reference's signature: one realisation per call, no index (see _run_state)."""
        r = _next_real(preproc, covars)
        run = preproc["_run"]
        return {var: preproc[var].realize(run["seed"] + run["vindex"][var], r, 1)[0] for var in covars}

    def solve(self, problem: SimulationProblem, gather: bool = True):
        pre = self.preprocess(problem)
        run = pre["_run"]
        rank, ws = parallel.world()
        lo, hi = parallel.shard_range(problem.nreals, rank, ws)
        reals = {}
        for var in problem.variables:
            plan = pre[var]
            y = plan.realize(run["seed"] + run["vindex"][var], lo, hi - lo)
            plan.close()
            if gather and ws > 1:
                y = parallel.all_gather_concat(y, problem.nreals)
            reals[var] = [y[r] for r in range(y.shape[0])]
        return Ensemble(problem.domain, reals)


def solve(problem, solver, **kw):
    """solve(problem, solver) -- the GeoStatsBase entry point extended at GeoStatsSolvers.jl:28."""
    return solver.solve(problem, **kw)
