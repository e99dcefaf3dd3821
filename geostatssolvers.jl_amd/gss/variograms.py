"""Variogram model constructors with the reference's keyword surface
(`GaussianVariogram(range=35., nugget=0.)`, `SphericalVariogram(range=10.)`,
`GaussianVariogram(MetricBall((20., 5.)))` -- /root/reference/test/estimation/krig.jl:10,
test/simulation/lu.jl:11,59-60, test/simulation/fft.jl:11).  They only carry parameters;
evaluation happens on the device (csrc/gss_internal.h cov_from_d2).

Gaussian model: [RECALL] Variography evaluates `GaussianVariogram` with `nugget + 1e-6` ("small eps ... for numerical
stability") -- which is why the reference's own suite can factor Gaussian covariances of 10^4 cells with
the nugget left at 0 (test/simulation/lu.jl:29-64).  `effective_nugget` applies that rule where the parameters are
handed to the device (`engine._vg_struct`); `GaussianVariogram(..., regularize=False)` opts out.  The kernels know
nothing of it: it is a parameter change."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

GAUSSIAN_NUGGET_EPS = 1e-6


@dataclass(frozen=True)
class MetricBall:
    radii: Tuple[float, ...]

    def __init__(self, radii):
        if not isinstance(radii, (tuple, list)):
            radii = (radii,)
        object.__setattr__(self, "radii", tuple(float(r) for r in radii))

    @property
    def isotropic(self):
        return len(self.radii) == 1


@dataclass(frozen=True)
class VariogramModel:
    kind: str
    sill: float = 1.0
    nugget: float = 0.0
    range: float = 1.0
    nu: float = 1.0
    radii: Optional[Tuple[float, ...]] = None
    regularize: bool = True      # Gaussian model only (module docstring)

    def isstationary(self):
        return self.kind != "power"

    @property
    def effective_nugget(self):
        """The nugget the evaluation uses; `nugget` stays what the user gave (as Variography's `nugget(γ)`)."""
        return self.nugget + (GAUSSIAN_NUGGET_EPS if self.kind == "gaussian" and self.regularize else 0.0)

    # gamma1 + gamma2 and c * gamma build a NestedVariogram ([DEP] Variography)
    def __add__(self, other):
        return NestedVariogram(((1.0, self),)) + other

    def __rmul__(self, c):
        return NestedVariogram(((float(c), self),))

    __mul__ = __rmul__


@dataclass(frozen=True)
class NestedVariogram:
    """gamma = sum_i c_i gamma_i; sill and nugget are the weighted sums of the structures'."""
    terms: Tuple[Tuple[float, VariogramModel], ...]
    kind: str = "nested"

    def __add__(self, other):
        o = other.terms if isinstance(other, NestedVariogram) else ((1.0, other),)
        return NestedVariogram(self.terms + tuple(o))

    def __rmul__(self, c):
        return NestedVariogram(tuple((float(c) * w, m) for w, m in self.terms))

    __mul__ = __rmul__

    @property
    def sill(self):
        return sum(w * m.sill for w, m in self.terms)

    @property
    def nugget(self):
        return sum(w * m.nugget for w, m in self.terms)

    @property
    def effective_nugget(self):
        return sum(w * m.effective_nugget for w, m in self.terms)

    def isstationary(self):
        return True


def _make(kind, ball=None, *, sill=1.0, nugget=0.0, range=1.0, order=None, nu=None, regularize=True):
    radii = None
    if ball is not None:
        if not isinstance(ball, MetricBall):
            raise TypeError("positional argument must be a MetricBall")
        if ball.isotropic:
            range = ball.radii[0]
        else:
            radii, range = ball.radii, 1.0
    o = order if order is not None else (nu if nu is not None else 1.0)
    # gamma(h) = (sill - nugget) f(h) + nugget for h > 0: a nugget beyond the sill would make the structured part
    # negative (no valid model; inside a nested model the device would otherwise drop the structure and keep its nugget)
    if not (float(sill) > 0.0 and 0.0 <= float(nugget) <= float(sill)):
        raise ValueError(f"variogram needs sill > 0 and 0 <= nugget <= sill (got sill={sill}, nugget={nugget})")
    return VariogramModel(kind, float(sill), float(nugget), float(range), float(o), radii, bool(regularize))


def GaussianVariogram(ball=None, **kw):
    return _make("gaussian", ball, **kw)


def ExponentialVariogram(ball=None, **kw):
    return _make("exponential", ball, **kw)


def SphericalVariogram(ball=None, **kw):
    return _make("spherical", ball, **kw)


def MaternVariogram(ball=None, **kw):
    return _make("matern", ball, **kw)


def CubicVariogram(ball=None, **kw):
    return _make("cubic", ball, **kw)


def PentasphericalVariogram(ball=None, **kw):
    return _make("pentaspherical", ball, **kw)


def SineHoleVariogram(ball=None, **kw):
    return _make("sinehole", ball, **kw)


def PowerVariogram(*, scaling=1.0, nugget=0.0, exponent=1.0):
    """gamma(h) = scaling h^exponent + nugget ([DEP] Variography PowerVariogram): not stationary, so only the
    constrained kriging variants accept it.  Stored as range := scaling, nu := exponent, sill := inf."""
    if not 0.0 < exponent < 2.0:
        raise ValueError("exponent must be in (0, 2)")
    return VariogramModel("power", float("inf"), float(nugget), float(scaling), float(exponent), None)
