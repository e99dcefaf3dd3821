"""gss -- host-side twin of GeoStatsSolvers.jl's KrigingSolver / FFTGS / LUGS over libgss_hip.so.

The package holds only what the hot path needs: the ctypes binding of the C-ABI (`_lib`), the
array-level engine (`engine`), the minimal geospatial containers the solvers read (`geo`), and the
solver front-ends (`solvers`) that mirror the reference's `solve(problem, solver)` API.
"""
from .geo import (CartesianGrid, Composition, DomainView, Ensemble, GeoTable, PointSet, aitchison, asarray, domain,
                  georef, parent, parentindices, view)
from .problems import EstimationProblem, SimulationProblem
from .solvers import (FFTGS, LUGS, SGS, ExpWeight, IDWSolver, KrigingSolver, LWRSolver, TricubeWeight, kriging_ui,
                      searcher_ui, simulate_with_generic_loop, solve)
from .variograms import (CubicVariogram, ExponentialVariogram, GaussianVariogram, MaternVariogram, MetricBall,
                         NestedVariogram, PentasphericalVariogram, PowerVariogram, SineHoleVariogram,
                         SphericalVariogram)

__all__ = [n for n in dir() if not n.startswith("_")]
