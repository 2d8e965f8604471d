#!/usr/bin/env python3
"""Generates the polynomial coefficients used by the f64 device math in
csrc/sfw_math.h:  atan(q) = q*P(q^2) on q in [0,1]  and  exp(r) on |r| <= ln2/2.
Near-minimax (Chebyshev interpolation), converted to the monomial basis and
verified with a float64 Horner evaluation."""
import numpy as np
from numpy.polynomial import Chebyshev, Polynomial


def fit(fn, lo, hi, deg):
    ch = Chebyshev.interpolate(fn, deg, domain=[lo, hi])
    return ch.convert(kind=Polynomial, domain=[-1, 1], window=[-1, 1]).coef


def horner(c, x):
    y = np.full_like(x, c[-1])
    for a in c[-2::-1]:
        y = y * x + a
    return y


def main():
    def g(z):
        q = np.sqrt(np.maximum(z, 0.0))
        return np.where(q > 1e-8, np.arctan(q) / np.where(q > 0, q, 1.0), 1.0 - z / 3.0)

    ca = fit(g, 0.0, 1.0, 15)
    q = np.linspace(0, 1, 400001)
    err = np.abs(q * horner(ca, q * q) - np.arctan(q))
    print("// atan(q) = q * P(q*q), q in [0,1]; max abs err %.2e (float64 Horner)" % err.max())
    print("constexpr double kAtanP[%d] = {" % len(ca))
    print(",\n".join("    %.17e" % v for v in ca) + "};")
    h = np.log(2.0) / 2
    ce = fit(np.exp, -h * 1.0001, h * 1.0001, 9)
    r = np.linspace(-h, h, 400001)
    err = np.abs(horner(ce, r) / np.exp(r) - 1)
    print("// exp(r), |r| <= ln2/2; max rel err %.2e" % err.max())
    print("constexpr double kExpP[%d] = {" % len(ce))
    print(",\n".join("    %.17e" % v for v in ce) + "};")
    # float versions (f32 mode): atan deg 7 in z, exp via v_exp_f32
    ca32 = fit(g, 0.0, 1.0, 7)
    err = np.abs(q * horner(ca32, q * q) - np.arctan(q))
    print("// float atan: max abs err %.2e" % err.max())
    print("constexpr float kAtanPf[%d] = {" % len(ca32))
    print(",\n".join("    %.9ef" % v for v in ca32) + "};")


if __name__ == "__main__":
    main()
