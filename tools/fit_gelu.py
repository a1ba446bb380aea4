#!/usr/bin/env python
"""Coefficients of csrc/common.h::gelu_erf_f:  erfc(t / sqrt 2) ~= 2^(-t (c1 + t (c2 + ... + t c6)))  for t >= 0, fitted by
iteratively re-weighted least squares towards the minimax error of erf, then checked in emulated fp32 in the form the kernel
evaluates (gelu(g) = max(g, 0) - |g| / 2 * 2^P(|g|)).  Prints the coefficients and the errors quoted in the kernel comment.
CPU only (numpy / scipy)."""
import numpy as np
from scipy.optimize import least_squares
from scipy.special import erf

DEG = 6
t = np.linspace(0, 8, 40001)
target = erf(t / np.sqrt(2))


def model(c, t):
    p = np.zeros_like(t)
    for ck in c[::-1]:
        p = (p + ck) * t
    return 1 - np.exp2(-np.minimum(p, 200.0))


c = np.zeros(DEG)
c[0], c[1] = 1.151, 0.46
w = np.ones_like(t)
for _ in range(60):
    c = least_squares(lambda c: w * (model(c, t) - target), c, method="lm", xtol=1e-15, ftol=1e-15).x
    e = np.abs(model(c, t) - target)
    w = w * (1 + 2 * e / e.max())
    w /= w.mean()
c32 = (-c).astype(np.float32)
print("P(t) = t * (c1 + t * (c2 + ...)), coefficients (negated: erfc = 2^P):")
print("  " + ", ".join("%.9ef" % v for v in c32))
print("max |erf error| (float64 evaluation):", np.abs(model(c, t) - target).max())
g = np.concatenate([np.linspace(-12, 12, 2000001), np.random.default_rng(0).standard_normal(1000000) * 3]).astype(np.float32)
a = np.abs(g)
q = np.float32(c32[DEG - 1]) * a + c32[DEG - 2]
for k in range(DEG - 3, -1, -1):
    q = (q * a + c32[k]).astype(np.float32)
r = (np.maximum(g, 0) - (np.float32(0.5) * a) * np.exp2((a * q).astype(np.float32)).astype(np.float32)).astype(np.float32)
ref = 0.5 * g.astype(np.float64) * (1 + erf(g.astype(np.float64) / np.sqrt(2)))
print("max |gelu error| in emulated fp32:", np.abs(r - ref).max())
