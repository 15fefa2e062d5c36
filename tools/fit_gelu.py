"""Coefficients of the GELU used by the GEMM epilogues (csrc/cpx_gemm.hip, gelu_erf):

    gelu(x) = x Phi(x) = max(x, 0) - |x| q(|x|),   q(a) = 0.5 erfc(a / sqrt 2) = 2 ** P(a)

P = a polynomial fitted to log2(0.5 erfc(a / sqrt 2)) with the weight a q(a) ln 2, i.e. minimising the ABSOLUTE error of the
product |x| q -- the only place q enters.  One v_exp_f32 and deg fused multiply-adds replace the v_rcp + v_exp + 5-term
polynomial of Abramowitz-Stegun 7.1.26.  The script fits (Chebyshev basis, iteratively re-weighted towards the minimax
solution), then checks the float32 evaluation (single-rounding Horner steps, as v_fma_f32 does them) against float64 erfc on a
dense grid and far outside the fitting interval (the leading coefficient is negative: P -> -inf, q -> 0 without a clamp)."""
import numpy as np
from numpy.polynomial import chebyshev as C, polynomial as Pn
from scipy.special import erfc

f32 = np.float32


def fit(deg, X, n=200001, iters=40):
    xs = np.linspace(0, X, n)
    f = np.log2(0.5 * erfc(xs / np.sqrt(2)))
    base_w = xs * 2.0 ** f * np.log(2) + 1e-10
    t = 2 * xs / X - 1
    w = base_w.copy()
    for _ in range(iters):
        c = C.chebfit(t, f, deg, w=w)
        e = np.abs((C.chebval(t, c) - f) * base_w)
        w = w * (1 + 2 * e / e.max())
    pt = C.cheb2poly(c)
    px, acc = np.zeros(1), np.array([1.0])
    for ck in pt:
        px = Pn.polyadd(px, ck * acc)
        acc = Pn.polymul(acc, np.array([-1.0, 2.0 / X]))
    return px


def eval32(px, x):
    x = x.astype(f32)
    ax = np.abs(x)
    p = np.full_like(x, f32(px[-1]))
    for c in px[-2::-1]:
        p = (p.astype(np.float64) * ax.astype(np.float64) + np.float64(f32(c))).astype(f32)
    q = np.exp2(p.astype(np.float64)).astype(f32)
    return (np.maximum(x, f32(0)).astype(np.float64) - ax.astype(np.float64) * q.astype(np.float64)).astype(f32)


if __name__ == "__main__":
    deg, X = 5, 6.25
    px = fit(deg, X)
    xs = np.concatenate([np.linspace(-14, 14, 2800001), np.random.default_rng(0).normal(0, 1.5, 2000000)])
    x64 = xs.astype(f32).astype(np.float64)
    ref = 0.5 * x64 * erfc(-x64 / np.sqrt(2))
    err = np.abs(eval32(px, xs).astype(np.float64) - ref)
    print("degree", deg, "fitted on [0, %g]" % X)
    print("coefficients c0..c%d:" % deg, ", ".join("%.9ef" % c for c in px))
    print("max |error| %.3e at x = %.3f;   max |error| / |x| for |x| > 0.5: %.3e" % (
        err.max(), xs[err.argmax()], (err / np.maximum(np.abs(xs), 1e-9))[np.abs(xs) > 0.5].max()))
    far = np.concatenate([np.linspace(X, 40, 100001), np.geomspace(40, 1e18, 2000)])
    p = np.polyval(px[::-1], far)
    print("beyond the interval: P decreasing:", bool(np.all(np.diff(p) < 0)), "  max |x| 2^P:", float((far * 2.0 ** p).max()))
    big = np.array([-1e30, -3e38, 1e30, 3e38, 0.0, -0.0], f32)
    print("extremes:", eval32(px, big))
