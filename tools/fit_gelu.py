#!/usr/bin/env python3
"""Coefficients of csrc/linear3.hip::l3_gelu2 (developer tool; CPU only).

erfc(t) = 2^(-t Q(t)) on t in (0, 4]: Q = -log2(erfc t) / t is smooth, a degree-8 polynomial fitted by iteratively
re-weighted least squares with the weight d erf / d Q = erfc(t) ln2 t (so that the error is flat in erf, not in Q).
Prints the coefficients and the error of the f32 evaluation (Horner, exp2, as the kernel does it) of erf and of
GELU(v) = 0.5 v (1 + erf(v / sqrt 2)) against f64, next to the reference's own expression evaluated in f32."""
import numpy as np
from scipy.special import erf, erfc

T, DEG = 4.0, 8
t = np.linspace(1e-6, T, 400001)
f = -np.log2(erfc(t)) / t
w = erfc(t) * np.log(2) * t
V = np.polynomial.chebyshev.chebvander(2 * t / T - 1, DEG)
ww = w.copy()
for _ in range(60):
    c, *_ = np.linalg.lstsq(V * ww[:, None], f * ww, rcond=None)
    err = (V @ c - f) * w
    ww = ww * (1 + 3 * np.abs(err) / np.abs(err).max())
    ww /= ww.max()
P = np.polynomial.Polynomial(np.polynomial.chebyshev.cheb2poly(c))(np.polynomial.Polynomial([-1, 2 / T]))
co = P.coef.astype(np.float32)
print("Q coefficients, low to high:", ", ".join("%.9e" % k for k in co))

v = np.linspace(-8, 8, 1600001).astype(np.float32)
z = (v * np.float32(0.70710678118654752440)).astype(np.float32)
tc = np.minimum(np.abs(z), np.float32(T))
q = np.zeros_like(tc) + co[-1]
for k in co[-2::-1]:
    q = (q * tc + k).astype(np.float32)
e = np.exp2((-(tc * q)).astype(np.float32)).astype(np.float32)
print("max |erf error|:", np.abs((np.float32(1) - e).astype(np.float64) - erf(tc.astype(np.float64))).max())
h = ((np.float32(0.5) * v).astype(np.float32) * e).astype(np.float32)
out = np.where(v > 0, (v - h).astype(np.float32), h)
vd = v.astype(np.float64)
true = 0.5 * vd * (1 + erf(vd / np.sqrt(2)))
ref32 = (np.float32(0.5) * v * (np.float32(1) + erf(z.astype(np.float64)).astype(np.float32))).astype(np.float32)
for name, y in (("this form", out), ("0.5 v (1 + erf) in f32, perfect erf", ref32)):
    d = np.abs(y - true)
    print(f"GELU, {name}: max abs error {d.max():.3e}, max error / max(|v|, 1) {(d / np.maximum(np.abs(vd), 1)).max():.3e}")
