"""Variogram / covariance models (oracle side; test infrastructure only).

Restates [DEP] Variography.jl 0.22 as used at the reference call sites
`/root/reference/src/simulation/fft.jl:91,98`, `src/simulation/lu.jl:110,124,131,132`
and inside `GeoStatsModels.fit/predictprob` (`src/estimation/krig.jl:176,180,223,226`).

    gamma(h) = (sill - nugget) * f(h / range) + nugget * (h > 0)
    cov(h)   = sill - gamma(h)                      (every reference call site)

with the "practical range" convention (factor 3) of Variography:
    Gaussian        f = 1 - exp(-3 x^2)
    Exponential     f = 1 - exp(-3 x)
    Spherical       f = 1.5 x - 0.5 x^3          (x < 1), 1 otherwise
    Matern(nu)      f = 1 - 2^(1-nu)/Gamma(nu) * d^nu * K_nu(d),  d = sqrt(2 nu) * 3 x
    Cubic           f = 7x^2 - 8.75x^3 + 3.5x^5 - 0.75x^7   (x < 1), 1 otherwise
    Pentaspherical  f = 1.875x - 1.25x^3 + 0.375x^5         (x < 1), 1 otherwise
    SineHole        f = 1 - sin(pi x) / (pi x)
and the one non-stationary model of the zoo,
    Power           gamma(h) = scaling * h^exponent + nugget * (h > 0)        (no sill; `range` holds the
                    scaling and `nu` the exponent; kriging then uses the variogram form of the system)

Anisotropy: `MetricBall((a, b, ...))` gives a Mahalanobis distance
h = sqrt(sum(((x_i - y_i) / r_i)^2)) with range 1
(`/root/reference/test/simulation/fft.jl:11`, `test/simulation/lu.jl:59-60`).

Gaussian-model regularisation ([RECALL] Variography `GaussianVariogram`: "add small eps to nugget for
numerical stability", `n = nugget + 1e-6` inside the evaluation, the `nugget(gamma)` accessor keeps the raw value):
    Gaussian        gamma(h) = (sill - n) * (1 - exp(-3 x^2)) + n * (h > 0),   n = nugget + 1e-6
Evidence in the tree: the reference's own suite factors 10^4 x 10^4 Gaussian covariances with `nugget` left at 0
(`/root/reference/test/simulation/lu.jl:29-64`, `src/simulation/lu.jl:124,128`); LAPACK refuses those matrices without
the epsilon and factors them with it (`tests/test_oracle_reference_cases.py::test_lugs_2d_and_anisotropic_inputs`).
The value 1e-6 itself is recollection, not in the tree; `regularize=False` switches it off.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

GAUSSIAN_NUGGET_EPS = 1e-6       # [RECALL] Variography GaussianVariogram: n = nugget + 1e-6

KINDS = ("gaussian", "exponential", "spherical", "matern", "cubic", "pentaspherical", "sinehole", "power")


@dataclass
class Variogram:
    kind: str = "gaussian"          # default model of every solver: krig.jl:65, fft.jl:52, lu.jl:68
    sill: float = 1.0
    nugget: float = 0.0
    range: float = 1.0
    nu: float = 1.0                 # Matern order (Variography default order = 1)
    radii: Optional[Sequence[float]] = None   # anisotropic MetricBall radii -> range := 1
    regularize: bool = True         # Gaussian model only: evaluate with nugget + 1e-6 (module docstring)

    def __post_init__(self):
        if self.kind not in KINDS:
            raise ValueError(f"unknown variogram kind {self.kind!r}")
        if self.radii is not None:
            self.radii = tuple(float(r) for r in self.radii)
            if len(self.radii) == 1:       # MetricBall(r) is isotropic with range r
                self.range = self.radii[0]
                self.radii = None
            elif self.kind != "power":
                self.range = 1.0
        if self.kind == "power":
            self.sill = float("inf")


@dataclass
class Nested:
    """[DEP] Variography NestedVariogram: gamma = sum_i c_i gamma_i (each structure keeps its own range / ball)."""
    terms: Sequence  # of (coefficient, Variogram)

    @property
    def sill(self) -> float:
        return float(sum(c * v.sill for c, v in self.terms))

    @property
    def nugget(self) -> float:
        return float(sum(c * v.nugget for c, v in self.terms))


def effective_nugget(vg) -> float:
    """The nugget the evaluation uses: `nugget + 1e-6` for a regularised Gaussian model, the raw value otherwise."""
    if isinstance(vg, Nested):
        return float(sum(c * effective_nugget(v) for c, v in vg.terms))
    return vg.nugget + (GAUSSIAN_NUGGET_EPS if vg.kind == "gaussian" and vg.regularize else 0.0)


def isstationary(vg) -> bool:
    """Finite sill (fft.jl:91, lu.jl:110): everything but the power model."""
    return getattr(vg, "kind", None) != "power"


def sill(vg: Variogram) -> float:
    return vg.sill


def distance(vg: Variogram, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Pairwise distance |a|x|b| between point-major coordinate arrays (n x d)."""
    a = np.atleast_2d(np.asarray(a, dtype=np.float64))
    b = np.atleast_2d(np.asarray(b, dtype=np.float64))
    diff = a[:, None, :] - b[None, :, :]
    if vg.radii is not None:
        diff = diff / np.asarray(vg.radii, dtype=np.float64)[None, None, :]
    return np.sqrt(np.sum(diff * diff, axis=-1))


def _shape(vg: Variogram, x: np.ndarray) -> np.ndarray:
    """f(x) with x = h / range."""
    k = vg.kind
    if k == "gaussian":
        return 1.0 - np.exp(-3.0 * x * x)
    if k == "exponential":
        return 1.0 - np.exp(-3.0 * x)
    if k == "spherical":
        return np.where(x < 1.0, 1.5 * x - 0.5 * x ** 3, 1.0)
    if k == "cubic":
        return np.where(x < 1.0, 7 * x ** 2 - 8.75 * x ** 3 + 3.5 * x ** 5 - 0.75 * x ** 7, 1.0)
    if k == "pentaspherical":
        return np.where(x < 1.0, 1.875 * x - 1.25 * x ** 3 + 0.375 * x ** 5, 1.0)
    if k == "sinehole":
        t = np.pi * np.asarray(x, dtype=np.float64)
        with np.errstate(invalid="ignore", divide="ignore"):
            return np.where(t > 0, 1.0 - np.sin(t) / t, 0.0)
    if k == "matern":
        nu = float(vg.nu)
        d = np.sqrt(2.0 * nu) * 3.0 * x
        if nu == 0.5:
            return 1.0 - np.exp(-d)
        if nu == 1.5:
            return 1.0 - (1.0 + d) * np.exp(-d)
        if nu == 2.5:
            return 1.0 - (1.0 + d + d * d / 3.0) * np.exp(-d)
        from scipy.special import gamma, kv
        with np.errstate(invalid="ignore", over="ignore"):
            val = 1.0 - (2.0 ** (1.0 - nu) / gamma(nu)) * d ** nu * kv(nu, d)
        return np.where(d > 0, val, 0.0)
    raise ValueError(k)


def gamma_h(vg: Variogram, h: np.ndarray) -> np.ndarray:
    h = np.asarray(h, dtype=np.float64)
    if vg.kind == "power":
        return vg.range * h ** vg.nu + vg.nugget * (h > 0)
    n = effective_nugget(vg)
    return (vg.sill - n) * _shape(vg, h / vg.range) + n * (h > 0)


def cov_h(vg: Variogram, h: np.ndarray) -> np.ndarray:
    """C(h) = sill - gamma(h) (fft.jl:98, lu.jl:124,131,132)."""
    return vg.sill - gamma_h(vg, h)


def pairwise(vg, a: np.ndarray, b: Optional[np.ndarray] = None) -> np.ndarray:
    """[DEP] Variography.pairwise(gamma, A[, B]) -> |A| x |B| matrix of gamma."""
    if b is None:
        b = a
    if isinstance(vg, Nested):
        return sum(c * pairwise(v, a, b) for c, v in vg.terms)
    return gamma_h(vg, distance(vg, a, b))


def cov_pairwise(vg, a: np.ndarray, b: Optional[np.ndarray] = None) -> np.ndarray:
    return vg.sill - pairwise(vg, a, b)
