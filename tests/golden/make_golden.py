#!/usr/bin/env python3
"""Generates tests/golden/*.npz.

PROVENANCE: these vectors are outputs of the build's own CPU oracle (oracle/, numpy + LAPACK) on the
reference's test inputs and on BASELINE config 1.  They are NOT outputs of the Julia reference: it
cannot be executed in this environment (no Julia toolchain, un-vendored dependencies), so no
reference-generated vector exists.  They freeze the oracle (regression guard) and give the GPU
tests fixed inputs/expected outputs that do not need the oracle at run time.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import fftgs, kriging as K, lugs, philox  # noqa: E402
from oracle.variogram import Variogram  # noqa: E402


def main():
    # --- reference test inputs: test/estimation/krig.jl:22-72 (3 data -> 100x100 grid, Gaussian range 35) ----
    vg = Variogram("gaussian", range=35.0, nugget=0.0)
    x = np.array([(25.0, 25.0), (50.0, 75.0), (75.0, 50.0)])
    z = np.array([1.0, 0.0, 1.0])
    grid = fftgs.grid_centroids((100, 100), (0.5, 0.5), (1.0, 1.0))
    mu_g, var_g = K.exactsolve(K.OK, vg, x, z, grid)
    mu_n, var_n, st_n, idx_n, cnt_n = K.approxsolve(K.OK, vg, x, z, grid, 3, return_idx=True)
    np.savez_compressed(os.path.join(HERE, "krig_reference_2d.npz"), x=x, z=z, grid=grid, mu_global=mu_g,
                        var_global=var_g, mu_nearest=mu_n, var_nearest=var_n, idx_nearest=idx_n)

    # --- BASELINE config 1: OK, 100 2-D data -> 64x64 grid, Gaussian range 20 nugget 1e-6 ---------------------
    x1 = np.random.default_rng(1).uniform(0.0, 64.0, (100, 2))
    z1 = np.random.default_rng(1001).normal(size=100)
    g1 = fftgs.grid_centroids((64, 64))
    vg1 = Variogram("gaussian", range=20.0, sill=1.0, nugget=1e-6)
    mu1, var1 = K.exactsolve(K.OK, vg1, x1, z1, g1)
    np.savez_compressed(os.path.join(HERE, "krig_config1.npz"), x=x1, z=z1, grid=g1, mu=mu1, var=var1)

    # --- config-5-shaped moving neighbourhood, small: UK degree 1, k = 16, Matern-3/2 -------------------------
    x5 = np.random.default_rng(6).uniform(0.0, 100.0, (300, 3))
    z5 = 1.0 + 0.03 * x5[:, 0] - 0.02 * x5[:, 1] + np.random.default_rng(60).normal(size=300)
    d5 = np.random.default_rng(7).uniform(0.0, 100.0, (200, 3))
    vg5 = Variogram("matern", range=30.0, nu=1.5)
    mu5, var5, st5, idx5, cnt5 = K.approxsolve(K.UK, vg5, x5, z5, d5, 16, degree=1, return_idx=True)
    np.savez_compressed(os.path.join(HERE, "krig_local_uk.npz"), x=x5, z=z5, dom=d5, mu=mu5, var=var5, idx=idx5)

    # --- FFTGS: 24x20x16 exponential, Philox seed 4 -------------------------------------------------------------
    pre = fftgs.preprocess(Variogram("exponential", range=6.0, sill=1.5), (24, 20, 16), mean=0.5)
    zf = fftgs.realize(pre, 4, 0, 2)
    np.savez_compressed(os.path.join(HERE, "fftgs_24x20x16.npz"), F=pre.F.ravel(), z=zf,
                        noise0=philox.uniform(4, 0, 24 * 20 * 16)[:64])

    # --- LUGS: test/simulation/lu.jl:8-16 inputs (5 data on a 100-cell line, spherical range 10) ---------------
    cent = fftgs.grid_centroids((100,))
    p = lugs.preprocess(Variogram("spherical", range=10.0), cent, np.array([[0.0], [25.0], [50.0], [75.0], [100.0]]),
                        np.array([0.0, 1.0, 0.0, 1.0, 0.0]))
    y, w = lugs.realize(p, 123, 0, 2)
    np.savez_compressed(os.path.join(HERE, "lugs_line100.npz"), dlocs=p.dlocs, z1=p.z1, d2=p.d2,
                        L22_diag=np.diag(p.L22), L22_row50=p.L22[50], y=y, w=w)
    idw_lwr_vectors()
    print("golden vectors written to", HERE)


def idw_lwr_vectors():
    """test/estimation/idw.jl:3-9 and test/estimation/lwr.jl:18-27 inputs (3 / 4 data -> 100x100 grid) plus a
    seeded 3-D k = 12 case; `python tests/golden/make_golden.py idw_lwr` writes only this file."""
    from oracle import idw_lwr as E
    grid = fftgs.grid_centroids((100, 100), (0.5, 0.5), (1.0, 1.0))
    x3 = np.array([(25.0, 25.0), (50.0, 75.0), (75.0, 50.0)])
    z3 = np.array([1.0, 0.0, 1.0])
    imu, isd, _ = E.idw(x3, z3, grid, 3)
    x4 = np.array([(25.0, 25.0), (50.0, 75.0), (75.0, 50.0), (75.0, 25.0)])
    z4 = np.array([1.0, 0.0, 1.0, 0.0])
    lmu3, lvar3, _ = E.lwr(x4, z4, grid, 3)
    lmu4, lvar4, _ = E.lwr(x4, z4, grid, 4)
    xs = np.random.default_rng(12).uniform(0.0, 50.0, (400, 3))
    zs = np.cos(xs[:, 0] / 9.0) + 0.02 * xs[:, 1] * xs[:, 2] / 50.0
    dom = np.random.default_rng(13).uniform(0.0, 50.0, (300, 3))
    smu_i, ssd_i, _ = E.idw(xs, zs, dom, 12, exponent=2)
    smu_l, svar_l, _ = E.lwr(xs, zs, dom, 12)
    np.savez_compressed(os.path.join(HERE, "idw_lwr.npz"), grid=grid, x3=x3, z3=z3, idw_mu=imu, idw_dist=isd, x4=x4,
                        z4=z4, lwr_mu3=lmu3, lwr_var3=lvar3, lwr_mu4=lmu4, lwr_var4=lvar4, xs=xs, zs=zs, dom=dom,
                        idw_mu_s=smu_i, idw_dist_s=ssd_i, lwr_mu_s=smu_l, lwr_var_s=svar_l)


if __name__ == "__main__":
    if sys.argv[1:] == ["idw_lwr"]:
        idw_lwr_vectors()
    else:
        main()
