"""Minimal geospatial containers mirroring the objects the reference solvers receive.

Only what the hot path touches ([DEP] Meshes.jl / GeoTables.jl as used in
/root/reference/src/estimation/krig.jl:78-81,163 and src/simulation/fft.jl:64-70,152):
PointSet, CartesianGrid, views of them, geo-referenced tables and ensembles.
Indices are 0-based on the Python side (the Julia shim shifts by one).
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np


class Domain:
    def nelements(self) -> int:
        raise NotImplementedError

    def embeddim(self) -> int:
        raise NotImplementedError

    def centroids(self) -> np.ndarray:
        """(nelements, d) point-major centroids: the build's point-support contract."""
        raise NotImplementedError

    def __len__(self):
        return self.nelements()


class PointSet(Domain):
    """PointSet(coords) with coords (n, d) point-major -- memory-identical to Julia's d x n matrix."""

    def __init__(self, coords):
        c = np.asarray(coords, dtype=np.float64)
        if c.ndim == 1:
            c = c[:, None]
        self.coords = np.ascontiguousarray(c)

    def nelements(self):
        return self.coords.shape[0]

    def embeddim(self):
        return self.coords.shape[1]

    def centroids(self):
        return self.coords

    def __eq__(self, other):
        return isinstance(other, PointSet) and np.array_equal(self.coords, other.coords)


class CartesianGrid(Domain):
    """CartesianGrid(dims[, origin, spacing]); `CartesianGrid(100, 100)` also accepted."""

    def __init__(self, *args):
        if len(args) >= 1 and isinstance(args[0], (tuple, list)):
            dims = tuple(int(x) for x in args[0])
            origin = args[1] if len(args) > 1 else None
            spacing = args[2] if len(args) > 2 else None
        else:
            dims = tuple(int(x) for x in args)
            origin = spacing = None
        d = len(dims)
        self.dims = dims
        self.origin = np.zeros(d) if origin is None else np.asarray(origin, dtype=np.float64)
        self.spacing = np.ones(d) if spacing is None else np.asarray(spacing, dtype=np.float64)

    def nelements(self):
        return int(np.prod(self.dims))

    def embeddim(self):
        return len(self.dims)

    def size(self):
        return self.dims

    def centroids(self):
        d = len(self.dims)
        axes = [self.origin[a] + (np.arange(self.dims[a]) + 0.5) * self.spacing[a] for a in range(d)]
        mesh = np.meshgrid(*axes[::-1], indexing="ij")
        return np.ascontiguousarray(np.stack([m.ravel() for m in mesh[::-1]], axis=1))

    def __eq__(self, other):
        return (isinstance(other, CartesianGrid) and self.dims == other.dims
                and np.array_equal(self.origin, other.origin) and np.array_equal(self.spacing, other.spacing))


class DomainView(Domain):
    """view(domain, inds): subset of a parent domain (fft.jl:66 `parent`, :152 `parentindices`)."""

    def __init__(self, parent: Domain, inds):
        self.parent = parent
        self.inds = np.asarray(inds, dtype=np.int64)

    def nelements(self):
        return int(self.inds.size)

    def embeddim(self):
        return self.parent.embeddim()

    def centroids(self):
        return np.ascontiguousarray(self.parent.centroids()[self.inds])

    def __eq__(self, other):
        return isinstance(other, DomainView) and self.parent == other.parent and np.array_equal(self.inds, other.inds)


def view(domain: Domain, inds) -> DomainView:
    if isinstance(inds, (range, slice)):
        inds = np.arange(domain.nelements())[inds] if isinstance(inds, slice) else np.asarray(list(inds))
    return DomainView(domain, inds)


def parent(domain: Domain) -> Domain:
    return domain.parent if isinstance(domain, DomainView) else domain


def parentindices(domain: Domain) -> Optional[np.ndarray]:
    return domain.inds if isinstance(domain, DomainView) else None


class Composition:
    """CoDa.jl's `Composition(parts...)` as far as the estimation loops use it (idw.jl:128-141 is generic over the value
    type; test/estimation/idw.jl:47-65 runs it on compositions): `w * c` is powering (parts .^ w), `c1 + c2` is
    perturbation (parts .* parts) and nothing is closed on the way -- a composition is an equivalence class under
    positive scaling, which `aitchison` respects ([RECALL] of CoDa's definitions; the package is not in the tree).
    Both operations are linear in the log-parts, which is how the device estimates them (one value column per part,
    one search and one weight vector: gss_idw_predict_cols)."""

    __slots__ = ("parts",)

    def __init__(self, *parts):
        if len(parts) == 1 and np.ndim(parts[0]) == 1:
            parts = tuple(parts[0])
        self.parts = np.asarray(parts, dtype=np.float64)

    def __rmul__(self, w):
        return Composition(self.parts ** float(w))

    __mul__ = __rmul__

    def __add__(self, other):
        return Composition(self.parts * other.parts)

    def __radd__(self, other):                  # sum(...) starts from 0
        return self if isinstance(other, (int, float)) and other == 0 else NotImplemented

    def closure(self):
        return Composition(self.parts / self.parts.sum())

    def clr(self):
        lg = np.log(self.parts)
        return lg - lg.mean()

    def __repr__(self):
        return "Composition(" + ", ".join("%.6g" % v for v in self.parts) + ")"


def aitchison(c1: Composition, c2: Composition) -> float:
    """Aitchison distance: the Euclidean norm of the difference of the centred log-ratio coordinates."""
    return float(np.linalg.norm(c1.clr() - c2.clr()))


class GeoTable:
    """Table + domain (`georef`).  Columns are accessed as attributes or items."""

    def __init__(self, table: Dict[str, np.ndarray], domain: Domain):
        self.table = dict(table)
        self.domain = domain

    def __getattr__(self, name):
        t = self.__dict__.get("table", {})
        if name in t:
            return t[name]
        raise AttributeError(name)

    def __getitem__(self, name):
        return self.table[name]

    def names(self):
        return list(self.table.keys())


def georef(table, domain) -> GeoTable:
    """georef(table, domain | coordinates): coordinates may be a list of tuples or an (n, d) array."""
    if not isinstance(domain, Domain):
        domain = PointSet(np.asarray(domain, dtype=np.float64))
    cols = {}
    for k, v in dict(table).items():
        numeric = isinstance(v, np.ndarray) and v.dtype != object     # (never walk a million floats looking for objects)
        if not numeric and len(v) and any(isinstance(x, Composition) for x in v):
            a = np.empty(len(v), dtype=object)          # a column of compositions (None = missing)
            a[:] = list(v)
        else:
            a = np.asarray([np.nan if x is None else x for x in v] if isinstance(v, (list, tuple)) else v)
        cols[k] = a
        if a.shape[0] != domain.nelements():
            raise ValueError(f"column {k} has {a.shape[0]} rows for {domain.nelements()} elements")
    return GeoTable(cols, domain)


class Ensemble:
    """Ensemble(domain, Dict(var => [realisation vectors])) -- cookie.jl:82, indexing as in
    test/simulation/fft.jl:22 (`sol[1].z`) and test/simulation/sgs.jl:16 (`sol[:z]`)."""

    def __init__(self, domain: Domain, reals: Dict[str, Sequence]):
        self.domain = domain
        self.reals = reals

    def __getitem__(self, key):
        if isinstance(key, str):
            return self.reals[key]
        return GeoTable({v: r[key] for v, r in self.reals.items()}, self.domain)

    def __len__(self):
        return len(next(iter(self.reals.values()))) if self.reals else 0


def domain(obj):
    return obj.domain


def asarray(sol: GeoTable, var: str) -> np.ndarray:
    """asarray(sol, :z): values reshaped to the grid size with Julia axis order Z[i, j(, k)]."""
    g = parent(sol.domain)
    return np.asarray(sol[var]).reshape(g.dims[::-1]).transpose()
