#!/usr/bin/env python3
"""Generates the 2^(j/64) table and the reduction constants of gpismap_amd/csrc/exp_tab.h with 80-digit decimal arithmetic, and
checks the scheme (in Python doubles) against the 80-digit exponential: worst error in ulps, how many results are not the
correctly rounded ones -- beside the same count for the C library's exp."""
from decimal import Decimal, getcontext
import math
import random
import struct

getcontext().prec = 80
LN2 = Decimal(2).ln()
N = 64
rows = []
for j in range(N):
    t = (LN2 * Decimal(j) / Decimal(N)).exp()
    hi = float(t)
    rows.append((hi, float(t - Decimal(hi))))
inv = float(Decimal(N) / LN2)
c = LN2 / Decimal(N)
bits = struct.unpack("<Q", struct.pack("<d", float(c)))[0] & ~((1 << 24) - 1)
chi = struct.unpack("<d", struct.pack("<Q", bits))[0]
clo = float(c - Decimal(chi))


def fexp(x):
    kd = float(round(x * inv))
    r = x - kd * chi          # exact: chi has 24 trailing zero bits
    r = r - kd * clo
    k = int(kd)
    p = 1 / 720
    for cc in (1 / 120, 1 / 24, 1 / 6, 0.5):
        p = p * r + cc
    p = p * r * r + r
    hi, lo = rows[k & 63]
    return math.ldexp(hi + (hi * p + lo), k >> 6)


if __name__ == "__main__":
    for j, (hi, lo) in enumerate(rows):
        print("    {%s, %s}," % (hi.hex(), lo.hex()))
    print("// 64 / ln 2 = %s ; ln 2 / 64 = %s + %s" % (inv.hex(), chi.hex(), clo.hex()))
    random.seed(1)
    worst, bad, badc = 0.0, 0, 0
    n = 20000
    for _ in range(n):
        x = -random.random() * 12
        ref = Decimal(x).exp()
        got = fexp(x)
        worst = max(worst, float(abs(Decimal(got) - ref) / Decimal(math.ulp(got))))
        bad += float(ref) != got
        badc += float(ref) != math.exp(x)
    print("// worst error %.3f ulp; not correctly rounded: %d of %d (C library: %d)" % (worst, bad, n, badc))
