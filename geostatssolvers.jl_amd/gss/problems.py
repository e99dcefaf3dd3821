"""EstimationProblem / SimulationProblem ([DEP] GeoStatsBase) as the reference solvers read them:
`data(problem)`, `domain(problem)`, `variables(problem)`, `nreals(problem)`
(/root/reference/src/estimation/krig.jl:78-81,140; src/simulation/lu.jl:78-80)."""
from __future__ import annotations

from typing import Optional, Sequence, Tuple, Union

from .geo import Domain, GeoTable


def _varnames(spec) -> Tuple[str, ...]:
    if isinstance(spec, str):
        return (spec,)
    if isinstance(spec, dict):
        return tuple(spec.keys())
    out = []
    for v in spec:
        out.append(v[0] if isinstance(v, (tuple, list)) else v)   # ("z", float) pairs ~ :z => Float64
    return tuple(out)


class EstimationProblem:
    def __init__(self, data: GeoTable, domain: Domain, variables: Union[str, Sequence[str]]):
        self.data = data
        self.domain = domain
        self.variables = _varnames(variables)
        for v in self.variables:
            if v not in data.table:
                raise ValueError(f"variable {v} not in data")


class SimulationProblem:
    """SimulationProblem([data,] domain, vars, nreals)."""

    def __init__(self, *args):
        if isinstance(args[0], GeoTable):
            self.data: Optional[GeoTable] = args[0]
            self.domain, variables, self.nreals = args[1], args[2], int(args[3])
        else:
            self.data = None
            self.domain, variables, self.nreals = args[0], args[1], int(args[2])
        if isinstance(variables, tuple) and len(variables) == 2 and not isinstance(variables[1], (str, tuple, list)):
            variables = (variables,)          # ("z", float) single pair
        self.variables = _varnames(variables)
