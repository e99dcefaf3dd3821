"""CPU oracle for the kriging / FFTGS / LUGS hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain numpy/scipy (FP64) restatement of the algorithm that
juliohm/GeoStatsSolvers.jl v0.7.16 runs for `KrigingSolver`, `FFTGS` and `LUGS`.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it; the product path (`geostatssolvers.jl_amd/`) never does and fails
loudly when the HIP library is missing.

PARITY PINNING STATUS
---------------------
The reference is pure Julia and its arithmetic lives in un-vendored dependencies
(Variography 0.22, GeoStatsModels 0.2, Meshes 0.37, GeoStatsBase 0.42, FFTW 1.5;
`/root/reference/Project.toml:27-43`). There is no Julia toolchain in the build
container or on the GPU box, so the reference cannot be executed and no outputs of
it exist.  The oracle is therefore pinned ONLY by
  * the reference's own numerical test assertions for this path
    (`test/estimation/krig.jl:35-37,50-52,70-72`, `test/simulation/fft.jl:21-22`,
    `test/ui.jl:6-37`), reproduced in `tests/test_oracle_reference_cases.py`, and
  * implementation-independent known-answer tests (`tests/test_oracle_kat.py`).
Beyond those assertions: **parity unpinned** (see DESIGN.md section 3).

Every function cites the reference file:line it follows; behaviour of the
un-vendored dependencies is restated from their published algorithms
 and marked [DEP].
"""
from . import variogram, kriging, fftgs, lugs, philox  # noqa: F401
