#!/usr/bin/env python3
"""Generates the 16-sample spectral->linear-sRGB table used by the thin-film (KHR_materials_iridescence) factor.

The reference evaluates iridescence through nvshaders' `thin_film_factor` (external to /root/reference), which
integrates 16 wavelengths between 400 and 700 nm against CIE 1931 colour-matching functions.  The tabulated CMF
values are not available offline, so this script derives them from the published analytic fit of
Wyman, Sloan, Shirley, "Simple Analytic Approximations to the CIE XYZ Color Matching Functions", JCGT 2013,
converts to linear Rec.709 and normalises so that a flat unit reflectance integrates to (1,1,1).
The printed literals are pasted verbatim into oracle/oracle_pt.cpp and csrc/device/pt_bsdf.hip.h.
"""
import math

def g(lam, mu, s1, s2):
    s = s1 if lam < mu else s2
    return math.exp(-0.5 * ((lam - mu) / s) ** 2)

def cmf(lam):
    x = 1.056 * g(lam, 599.8, 37.9, 31.0) + 0.362 * g(lam, 442.0, 16.0, 26.7) - 0.065 * g(lam, 501.1, 20.4, 26.2)
    y = 0.821 * g(lam, 568.8, 46.9, 40.5) + 0.286 * g(lam, 530.9, 16.3, 31.1)
    z = 1.217 * g(lam, 437.0, 11.8, 36.0) + 0.681 * g(lam, 459.0, 26.0, 13.8)
    return x, y, z

M = [[3.2406, -1.5372, -0.4986], [-0.9689, 1.8758, 0.0415], [0.0557, -0.2040, 1.0570]]
N = 16
step = 300.0 / N
lams = [400.0 + (i + 0.5) * step for i in range(N)]
rgb = []
for lam in lams:
    x, y, z = cmf(lam)
    rgb.append([M[r][0] * x + M[r][1] * y + M[r][2] * z for r in range(3)])
white = [sum(c[r] for c in rgb) / N for r in range(3)]
print("// lambda_i = 400 + (i + 0.5) * 18.75 nm; rows are linear Rec.709, mean over rows = (1,1,1)")
for lam, c in zip(lams, rgb):
    print("  {%.9ef, %.9ef, %.9ef},  // %.3f nm" % (c[0] / white[0], c[1] / white[1], c[2] / white[2], lam))
