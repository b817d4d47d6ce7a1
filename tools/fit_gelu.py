#!/usr/bin/env python3
"""Coefficients of csrc/ffn.hip's GELU: degree-5 weighted minimax-style fit of log2 Phi(-z) on [0, 7], the weight being the
sensitivity of x Phi(x) = max(x, 0) - |x| Phi(-|x|) to an error in the exponent (z Phi(-z) ln 2); prints the coefficients
(constant term first) and the fp32-evaluated maximum error of the GELU itself."""
import numpy as np
from scipy.special import log_ndtr

deg, zmax = 5, 7.0
z = np.linspace(0, zmax, 40001)
lh = log_ndtr(-z) / np.log(2)
w = z * 2.0 ** lh * np.log(2) + 1e-9
c = np.polynomial.polynomial.polyfit(z, lh, deg, w=w)
for _ in range(80):                                  # iterated reweighting towards the minimax solution
    e = np.abs(np.polynomial.polynomial.polyval(z, c) - lh) * w
    w = w * (1 + 2 * e / e.max())
    c = np.polynomial.polynomial.polyfit(z, lh, deg, w=w)
print(", ".join("%.9ef" % v for v in c))
zz = np.linspace(0, 12, 240001).astype(np.float32)
zc = np.minimum(zz, np.float32(zmax))
cf = c.astype(np.float32)
q = np.zeros_like(zc) + cf[-1]
for k in range(len(cf) - 2, -1, -1):
    q = (q * zc + cf[k]).astype(np.float32)
r = np.exp2(q).astype(np.float32)
print("max |gelu error|: %.3e" % np.abs(zz * (r - np.exp(log_ndtr(-zz.astype(np.float64))))).max())
