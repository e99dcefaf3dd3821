#!/usr/bin/env python3
"""Generates the Chebyshev tables of Temme's Gamma1 / Gamma2 used by `gss_matern_general` (gss_internal.h) and checks a
float64 transcription of the device algorithm against mpmath (40 digits).

    Gamma1(mu) = (1/Gamma(1-mu) - 1/Gamma(1+mu)) / (2 mu),   Gamma2(mu) = (1/Gamma(1-mu) + 1/Gamma(1+mu)) / 2

are even in mu; on |mu| <= 1/2 they are expanded in T_k(8 mu^2 - 1).  The Bessel function itself follows Temme (1975),
"On the numerical evaluation of the modified Bessel function of the third kind": power series for x <= 2, Steed's
continued fraction for x > 2, then upward recurrence in the order.

    python tools/gen_matern_general.py            # prints the tables and the maximum relative error
"""
import math
import sys

import mpmath as mp

mp.mp.dps = 40


def cheb_coeffs(f, n):
    """c_0..c_{n-1} of f(t) ~ c_0 + sum_{k>=1} c_k T_k(t) on [-1, 1] (Chebyshev-Gauss nodes)."""
    nodes = [mp.cos(mp.pi * (j + mp.mpf(1) / 2) / n) for j in range(n)]
    fv = [f(t) for t in nodes]
    out = []
    for k in range(n):
        s = sum(fv[j] * mp.cos(mp.pi * k * (j + mp.mpf(1) / 2) / n) for j in range(n))
        out.append((2 if k else 1) * s / n)
    return out


def gam1_t(t):
    mu = mp.sqrt((t + 1) / 8)
    return (1 / mp.gamma(1 - mu) - 1 / mp.gamma(1 + mu)) / (2 * mu)


def gam2_t(t):
    mu = mp.sqrt((t + 1) / 8)
    return (1 / mp.gamma(1 - mu) + 1 / mp.gamma(1 + mu)) / 2


def tables(n=20, drop=3e-18):
    c1 = [float(c) for c in cheb_coeffs(gam1_t, n) if abs(c) > drop]
    c2 = [float(c) for c in cheb_coeffs(gam2_t, n) if abs(c) > drop]
    return c1, c2


def clenshaw(c, u):
    b1 = b2 = 0.0
    for j in range(len(c) - 1, 0, -1):
        b1, b2 = 2.0 * u * b1 + c[j] - b2, b1
    return u * b1 + c[0] - b2


def matern_general(d, nu, c1, c2):
    """float64 transcription of the device routine: 2^(1-nu)/Gamma(nu) d^nu K_nu(d)."""
    n = int(math.floor(nu + 0.5))
    mu = nu - n
    t = 8.0 * mu * mu - 1.0
    g1, g2 = clenshaw(c1, t), clenshaw(c2, t)
    gampl, gammi = g2 - mu * g1, g2 + mu * g1
    if d <= 2.0:
        pimu = math.pi * mu
        fact = 1.0 if abs(pimu) < 1e-15 else pimu / math.sin(pimu)
        dl = -math.log(0.5 * d)
        e = mu * dl
        fact2 = 1.0 if abs(e) < 1e-15 else math.sinh(e) / e
        ff = fact * (g1 * math.cosh(e) + g2 * fact2 * dl)
        s = ff
        ee = math.exp(e)
        p = 0.5 * ee / gampl
        q = 0.5 / (ee * gammi)
        c = 1.0
        dd = 0.25 * d * d
        s1 = p
        for i in range(1, 60):
            ff = (i * ff + p + q) / (i * i - mu * mu)
            c *= dd / i
            p /= (i - mu)
            q /= (i + mu)
            de = c * ff
            s += de
            s1 += c * (p - i * ff)
            if abs(de) < abs(s) * 1e-17:
                break
        kmu, kmu1 = s, s1 * 2.0 / d
    else:
        b = 2.0 * (1.0 + d)
        dd = 1.0 / b
        h = delh = dd
        q1, q2 = 0.0, 1.0
        a1 = 0.25 - mu * mu
        q = c = a1
        a = -a1
        s = 1.0 + q * delh
        for i in range(2, 500):
            a -= 2 * (i - 1)
            c = -a * c / i
            qn = (q1 - b * q2) / a
            q1, q2 = q2, qn
            q += c * qn
            b += 2.0
            dd = 1.0 / (b + a * dd)
            delh = (b * dd - 1.0) * delh
            h += delh
            dels = q * delh
            s += dels
            if abs(dels / s) < 1e-17:
                break
        h = a1 * h
        kmu = math.sqrt(math.pi / (2.0 * d)) * math.exp(-d) / s
        kmu1 = kmu * (mu + d + 0.5 - h) / d
    km, kp = kmu, kmu1
    for j in range(1, n):
        km, kp = kp, km + 2.0 * (mu + j) / d * kp
    knu = kmu if n == 0 else kp
    inv_gamma = gampl * mu if n == 0 else gampl
    for j in range(1, n):
        inv_gamma /= (mu + j)
    return 2.0 ** (1.0 - nu) * inv_gamma * d ** nu * knu


def exact(d, nu):
    d, nu = mp.mpf(d), mp.mpf(nu)
    return mp.mpf(2) ** (1 - nu) / mp.gamma(nu) * d ** nu * mp.besselk(nu, d)


def fmt(name, c):
    rows = ",\n    ".join(", ".join(repr(x) for x in c[i:i + 3]) for i in range(0, len(c), 3))
    return "static __device__ const double %s[%d] = {\n    %s};" % (name, len(c), rows)


if __name__ == "__main__":
    c1, c2 = tables()
    print(fmt("GSS_TEMME_G1", c1))
    print(fmt("GSS_TEMME_G2", c2))
    worst = 0.0
    for nu in (0.05, 0.3, 0.49, 0.5, 0.51, 0.8, 1.0, 1.3, 1.5, 1.75, 2.4, 3.0, 4.2, 7.5, 12.3, 20.0):
        for d in (1e-9, 1e-5, 1e-3, 0.1, 0.5, 1.0, 1.99, 2.0, 2.01, 3.0, 5.0, 10.0, 30.0, 100.0, 400.0, 690.0):
            got, ref = matern_general(d, nu, c1, c2), exact(d, nu)
            err = abs((mp.mpf(got) - ref) / ref)
            if err > worst:
                worst, at = float(err), (nu, d)
    print("max relative error vs mpmath: %.2e at (nu, d) = %s" % (worst, at), file=sys.stderr)
