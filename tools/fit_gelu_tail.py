#!/usr/bin/env python3
"""Coefficients of csrc/common.h::normal_tail and their error bounds.

q(a) = 1 - Phi(a) = 0.5 erfc(a / sqrt 2) is evaluated as exp2(-1 + a (c1 + a (c2 + ... + a c6))): a degree-6 polynomial fit of
log2 q on [0, 6] (reweighted least squares towards the minimax fit; the constant term is pinned to log2 q(0) = -1).  Prints
the coefficients and, evaluated in float32 exactly as the kernels do, the worst errors of gelu and gelu' against the erf form.
    python tools/fit_gelu_tail.py"""
import numpy as np
from scipy.special import erfc

x = np.linspace(0, 6.0, 60001)
tg = np.log2(0.5 * erfc(x / np.sqrt(2))) + 1.0
V = np.vander(x, 7, increasing=True)[:, 1:]
w, best = np.ones_like(x), None
for _ in range(3000):
    c, *_ = np.linalg.lstsq(V * w[:, None], tg * w, rcond=None)
    e = np.abs(V @ c - tg)
    if best is None or e.max() < best[0]:
        best = (e.max(), c.copy())
    w = w * (1 + 0.5 * e / e.max())
    w /= w.max()
c = best[1]
print("max |log2 error| %.3e  ->  relative error of q %.3e" % (best[0], best[0] * np.log(2)))
print("c1..c6 =", ", ".join("%.10e" % v for v in c))
shipped = np.array([-1.1504803413e+00, -4.6086354093e-01, -5.1418609203e-02, 7.2603524696e-03, -6.1973935318e-04, 2.3433251092e-05], dtype=np.float32)
xf = np.linspace(-9, 9, 900001).astype(np.float32)
ax = np.abs(xf)
a = np.minimum(ax, np.float32(6.0))
p = np.full_like(a, shipped[5])
for k in (4, 3, 2, 1, 0):
    p = (p * a + shipped[k]).astype(np.float32)
q = np.exp2((p * a - np.float32(1)).astype(np.float32)).astype(np.float32)
gelu = (np.maximum(xf, 0) - ax * q).astype(np.float32)
xd = xf.astype(np.float64)
cdf = 0.5 * erfc(-xd / np.sqrt(2))
ref = xd * cdf
err = np.abs(gelu - ref)
big = np.abs(ref) > 1e-6
print("shipped coefficients, float32: gelu max abs err %.3e (x = %.3f), max rel err where |gelu| > 1e-6: %.3e" % (
    err.max(), xf[err.argmax()], (err[big] / np.abs(ref[big])).max()))
phi = np.exp2((np.float32(-0.72134752044448170368) * xf * xf).astype(np.float32)).astype(np.float32)
gp = (np.where(xf >= 0, 1 - q, q) + xf * np.float32(0.3989422804014327) * phi).astype(np.float32)
refp = cdf + xd * np.exp(-xd * xd / 2) / np.sqrt(2 * np.pi)
print("                               gelu' max abs err %.3e" % np.abs(gp - refp).max())
