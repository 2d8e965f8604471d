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

    # half-angle form: phi in [0, pi/4] is evaluated as phi = t * P(t*t) with
    # t = tan(phi/2) = min/(max + hypot) in [0, tan(pi/8)]  (P includes the factor 2)
    tmax = np.tan(np.pi / 8)

    def g(z):  # noqa: F811
        t = np.sqrt(np.maximum(z, 0.0))
        return np.where(t > 1e-8, 2 * np.arctan(t) / np.where(t > 0, t, 1.0), 2.0 - 2 * z / 3.0)

    ca = fit(g, 0.0, tmax * tmax * 1.0001, 8)
    q = np.linspace(0, tmax, 400001)
    err = np.abs(q * horner(ca, q * q) - 2 * np.arctan(q))
    print("// 2*atan(t) = t * P(t*t), t in [0, tan(pi/8)]; max abs err %.2e (float64 Horner)" % err.max())
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
    ca32 = fit(g, 0.0, tmax * tmax * 1.0001, 4)
    err = np.abs(q * horner(ca32, q * q) - 2 * np.arctan(q))
    print("// float atan: max abs err %.2e" % err.max())
    print("constexpr float kAtanPf[%d] = {" % len(ca32))
    print(",\n".join("    %.9ef" % v for v in ca32) + "};")


if __name__ == "__main__":
    main()
