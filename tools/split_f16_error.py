#!/usr/bin/env python3
"""CPU emulation: can an fp32 contraction run on the fp16 matrix cores (16x the exact-fp32 MFMA rate on gfx950) without losing
fp32 accuracy?  Split each operand into two halfs, x = hi + lo (hi = fp16(x), lo = fp16(x - hi)), and sum the three products
hi*hi + hi*lo + lo*hi in fp32 -- the "3-term split" known from 3xTF32 / Ozaki-style schemes.  One GEMM stands in for a Winograd
position: M[tile, cout] = sum_k V[tile, k] * U[k, cout] with V like transformed activations (order 1) and U like transformed
weights (order 1e-2 .. 1e-3; a power-of-two pre-scale keeps their low halves out of fp16's subnormal range).

Prints max |result - float64 reference| / max |reference| for: plain fp32 accumulation, the 3-term split, the 4-term split, one
fp16 product alone.  No GPU needed.  Recorded in profiles/r03_split_f16_error.txt; the kernel it argues for is not built (open items)."""
import numpy as np

rng = np.random.default_rng(0)


def run(K, uscale, shift_u):
    V = (np.maximum(rng.standard_normal((256, K)), 0) * 3 + rng.standard_normal((256, K)) * 0.3).astype(np.float32)
    U = (rng.standard_normal((K, 64)) * uscale).astype(np.float32)
    ref = V.astype(np.float64) @ U.astype(np.float64)

    def split(x, sh=0):
        xs = x * np.float32(2.0 ** sh)
        hi = xs.astype(np.float16)
        lo = (xs - hi.astype(np.float32)).astype(np.float16)
        return hi.astype(np.float32), lo.astype(np.float32)

    vh, vl = split(V)
    uh, ul = split(U, shift_u)
    back = np.float32(2.0 ** -shift_u)
    three = (vh @ uh + vh @ ul + vl @ uh) * back
    four = three + (vl @ ul) * back
    m = np.abs(ref).max()
    return [np.abs(r - ref).max() / m for r in (V @ U, three, four, (vh @ uh) * back)]


print("%5s %8s %6s | %10s %10s %10s %10s" % ("K", "|U|~", "shift", "fp32", "3-term f16", "4-term f16", "1-term f16"))
for K in (64, 512, 1024):
    for us, sh in ((0.05, 0), (0.05, 8), (0.005, 0), (0.005, 10)):
        print("%5d %8g %6d | %10.2e %10.2e %10.2e %10.2e" % ((K, us, sh) + tuple(run(K, us, sh))))
