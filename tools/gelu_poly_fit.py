"""Coefficients of the erf polynomial behind `gelu_erf_poly2_n` (stamp_amd/csrc/common.h):  python tools/gelu_poly_fit.py [zmax] [degree]

erf(z) ~ zc P(u), zc = clamp(z, +-zmax), u = 2 zc^2 / zmax^2 - 1, with the constraint zmax P(1) = 1 (the approximation saturates at exactly +-1, so
gelu is exactly 0 or x beyond the clamp).  Near-minimax by Lawson-weighted least squares in the Chebyshev basis; printed in the monomial basis of u
with the 1/sqrt2 of x = z sqrt2 folded in (the kernel clamps x, not z), followed by an fp32 emulation of the kernel's Horner evaluation."""
import sys

import numpy as np
from numpy.polynomial import chebyshev as C
from numpy.polynomial import polynomial as P
from scipy.special import erf

zmax = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 8
z = np.linspace(1e-7, zmax, 40001)
u = 2 * z * z / zmax ** 2 - 1
# P(u) = 1 / zmax + (u - 1) R(u): the constraint by construction
A = (z * (u - 1))[:, None] * C.chebvander(u, deg - 1)
b = erf(z) - z / zmax
w, best = np.ones_like(z), None
for _ in range(80):
    c, *_ = np.linalg.lstsq(A * w[:, None], b * w, rcond=None)
    r = np.abs(A @ c - b)
    if best is None or r.max() < best[0]:
        best = (r.max(), c.copy())
    w = w * np.sqrt(np.maximum(r, 1e-12) / r.max() + 1e-3)
    w /= w.max()
err, c = best
Pm = P.polyadd([1 / zmax], P.polymul([-1, 1], C.cheb2poly(c)))
Q = (Pm / np.sqrt(2)).astype(np.float32)
XMAX, US = np.float32(zmax * np.sqrt(2)), np.float32(1 / zmax ** 2)
x = np.linspace(-8, 8, 400001).astype(np.float32)
xc = np.clip(x, -XMAX, XMAX)
uu = (xc * xc * US - np.float32(1)).astype(np.float32)
p = np.full_like(x, Q[-1])
for q in Q[-2::-1]:
    p = (p * uu + q).astype(np.float32)
h = x * np.float32(0.5)
g = (h * (xc * p) + h).astype(np.float32)
ref = 0.5 * x.astype(np.float64) * (1 + erf(x.astype(np.float64) / np.sqrt(2)))
ae = np.abs(g - ref)
print(f"zmax {zmax}, degree {deg}: |erf error| <= {err:.2e} (1 - erf(zmax) = {1 - erf(zmax):.2e})")
print(f"fp32 Horner: |gelu error| <= {ae.max():.2e} at x = {x[ae.argmax()]:.2f};  max error / |x| = {np.max(ae / np.maximum(np.abs(x), 1e-3)):.2e};  |x| > 6: {ae[np.abs(x) > 6].max():.2e}")
print("GELU_XMAX =", repr(float(XMAX)), " GELU_USCALE =", repr(float(US)))
print("GELU_Q = {" + ", ".join(f"{v:.9e}f" for v in Q) + "}")

# ---- the form the kernel evaluates: gelu = x (1/2 + xc R(t)), t = xc^2 -- P re-expanded in t, 1/sqrt2 and 1/2 folded in; then R[0], R[1] moved by a few
# ulps so that the fp32 FMA chain gives xc R(t_max) = -+1/2 at the clamp (what is left there is multiplied by x: it must not grow with |x|)
f32 = np.float32


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + np.float64(c)).astype(f32)


R, pw = np.zeros(1), np.array([1.0])
for q in Pm / np.sqrt(2) * 0.5:
    R = P.polyadd(R, q * pw)
    pw = P.polymul(pw, np.array([-1.0, 1.0 / zmax ** 2]))
R32 = R.astype(f32)


def gelu_t(xv, Rt):
    xcv = np.clip(xv, -XMAX, XMAX).astype(f32)
    t = (xcv * xcv).astype(f32)
    pv = np.full_like(xv, Rt[-1])
    for r in Rt[-2::-1]:
        pv = fma(pv, t, r)
    e = fma(xcv, pv, f32(0.5))
    return (xv * e).astype(f32), e


def nudge(v, k):
    for _ in range(abs(k)):
        v = np.nextafter(v, f32(np.inf) if k > 0 else f32(-np.inf))
    return v


ends, best = np.array([-XMAX, XMAX], dtype=f32), None
for k0 in range(-12, 13):
    for k1 in range(-3, 4):
        Rt = R32.copy()
        Rt[0], Rt[1] = nudge(R32[0], k0), nudge(R32[1], k1)
        e = gelu_t(ends, Rt)[1]
        d = abs(float(e[0])) + abs(float(e[1]) - 1.0)
        if best is None or d < best[0]:
            best = (d, k0, k1, Rt.copy())
d, k0, k1, Rt = best
gt = gelu_t(x, Rt)[0]
aet = np.abs(gt - ref)
print(f"x^2 form, R[0] {k0:+d} ulp, R[1] {k1:+d} ulp: residue at the clamp {d:.2e};  |gelu error| <= {aet.max():.2e};  max error / |x| = {np.max(aet / np.maximum(np.abs(x), 1e-3)):.2e};  "
      f"|x| > 6: {aet[np.abs(x) > 6].max():.2e}")
print("GELU_R = {" + ", ".join(f"{v:.9e}f" for v in Rt) + "}")
