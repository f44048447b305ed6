#!/usr/bin/env python3
"""Derive the polynomial coefficients used by csrc/detmath.h.

Nothing here comes from the reference: the reference calls the OpenCL built-ins
cos/log (src/render.cl:151-153), whose precision is implementation-defined.
detmath pins ONE bit-reproducible realisation; this script documents where its
constants come from (near-minimax fits through Chebyshev nodes, evaluated with
mpmath at 200 bits, rounded to binary32).
"""
import mpmath as mp
import numpy as np
import struct

mp.mp.prec = 200


def cheb_fit(f, a, b, deg):
    """Interpolate f on [a,b] at deg+1 Chebyshev nodes; return monomial coeffs."""
    n = deg + 1
    xs = [(a + b) / 2 + (b - a) / 2 * mp.cos(mp.pi * (2 * k + 1) / (2 * n)) for k in range(n)]
    A = mp.matrix(n, n)
    y = mp.matrix(n, 1)
    for i, x in enumerate(xs):
        for j in range(n):
            A[i, j] = x ** j
        y[i] = f(x)
    c = mp.lu_solve(A, y)
    return [c[i] for i in range(n)]


def f32(v):
    f = np.float32(float(v))
    return f, struct.unpack("<I", struct.pack("<f", f))[0]


def show(name, coeffs):
    for i, c in enumerate(coeffs):
        f, bits = f32(c)
        print(f"{name}{i} = {float(f):.10e}f  /* 0x{bits:08x} */")


zmax = (mp.pi / 4) ** 2 * mp.mpf("1.02")

# sin(r) = r + r*z*(S0 + S1 z + S2 z^2),  z = r^2
def sin_tail(z):
    if z == 0:
        return mp.mpf(-1) / 6
    r = mp.sqrt(z)
    return (mp.sin(r) / r - 1) / z

show("S", cheb_fit(sin_tail, mp.mpf(0), zmax, 2))

# cos(r) = 1 - z/2 + z^2*(C0 + C1 z + C2 z^2)
def cos_tail(z):
    if z == 0:
        return mp.mpf(1) / 24
    r = mp.sqrt(z)
    return (mp.cos(r) - 1 + z / 2) / (z * z)

show("C", cheb_fit(cos_tail, mp.mpf(0), zmax, 2))

# log(1+f) = 2s + s*R(z), s = f/(2+f), z = s^2, R(z) = z*(L0 + L1 z + L2 z^2 + L3 z^3)
smax = (mp.sqrt(2) - 1) / (mp.sqrt(2) + 1)
def log_tail(z):
    if z == 0:
        return mp.mpf(2) / 3
    s = mp.sqrt(z)
    return (mp.log((1 + s) / (1 - s)) - 2 * s) / (s * z)

show("L", cheb_fit(log_tail, mp.mpf(0), smax ** 2 * mp.mpf("1.02"), 3))

# atan(a)/pi = a * (A0 + A1 z + ... + A9 z^9), z = a^2 in [0,1]
def atanpi_tail(z):
    if z == 0:
        return 1 / mp.pi
    s = mp.sqrt(z)
    return mp.atan(s) / (mp.pi * s)

show("A", cheb_fit(atanpi_tail, mp.mpf(0), mp.mpf(1), 9))
