"""Coefficients of the transcendental-free erf-GELU of csrc/common.h (gelu_erf2).

gelu(x) = x * Phi(x),  Phi(x) ~= 0.5 + xc * Q(xc^2),  xc = clamp(x, -c, c).  Q is a weighted minimax fit (Lawson iteration on Chebyshev
nodes, weight = the factor x * xc that multiplies Q's error in the GELU value), converted to the monomial basis in s = xc^2 and checked in
emulated fp32 Horner arithmetic, i.e. as the kernel evaluates it.  Prints the error table for a few (c, degree) pairs and the coefficients
of the shipped pair (c = 4.5, degree 8).

    python tools/fit_gelu.py
"""
import numpy as np
from numpy.polynomial import chebyshev as Ch, polynomial as P
from scipy.special import erf


def phi(x):
    return 0.5 * (1 + erf(x / np.sqrt(2)))


def fit(c, deg):
    x = np.cos(np.linspace(0, np.pi, 6000)) * c / 2 + c / 2
    x = x[x > 1e-6]
    t = 2 * x * x / (c * c) - 1
    target = (phi(x) - 0.5) / x
    w = np.ones_like(x)
    for _ in range(200):
        V = Ch.chebvander(t, deg)
        W = w * x * np.maximum(x, 0.3)
        coef = np.linalg.lstsq(V * W[:, None], target * W, rcond=None)[0]
        err = (V @ coef - target) * x * np.maximum(x, 0.3)
        w = w * (np.abs(err) / np.abs(err).max() + 1e-3) ** 0.7
        w /= w.max()
    q_t, lin, q_s = Ch.cheb2poly(coef), np.array([-1, 2 / (c * c)]), np.zeros(1)
    for k, a in enumerate(q_t):   # t = 2 s / c^2 - 1
        q_s = P.polyadd(q_s, a * P.polypow(lin, k))
    return q_s


def eval_fp32(q_s, c, xx):
    x = xx.astype(np.float32)
    xc = np.clip(x, np.float32(-c), np.float32(c))
    s = (xc * xc).astype(np.float32)
    acc = np.full_like(s, np.float32(q_s[-1]))
    for a in q_s[-2::-1]:
        acc = (acc * s + np.float32(a)).astype(np.float32)
    return (x * (xc * acc + np.float32(0.5)).astype(np.float32)).astype(np.float32)


if __name__ == "__main__":
    xx = np.concatenate([np.linspace(-12, 12, 400001), np.random.default_rng(0).normal(size=200000)])
    gt = xx * phi(xx)
    for c, d in ((4.25, 7), (4.5, 7), (4.5, 8), (4.75, 8), (5.0, 9)):
        q = fit(c, d)
        e = np.abs(eval_fp32(q, c, xx) - gt)
        pos = xx > 0.5
        print(f"c={c} degree={d}: max |err| {e.max():.2e} (at x={xx[e.argmax()]:.3f}), inside the clamp {e[np.abs(xx) < c].max():.2e}, "
              f"max relative for x > 0.5: {(e[pos] / gt[pos]).max():.2e}")
        if (c, d) == (4.5, 8):
            print("   shipped coefficients (s^0 .. s^8):", ", ".join("%.9ef" % a for a in q))
