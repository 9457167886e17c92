#!/usr/bin/env python3
"""Offline generator of code-robchar_amd/csrc/mt19937_jump_poly.h: the jump-ahead polynomial of MT19937 for a fixed
distance B (in 32-bit words), used by the device-side legacy stream to start P sub-streams in parallel.

Theory (Haramoto, Matsumoto, Nishimura, Panneton, L'Ecuyer, "Efficient jump ahead for F2-linear random number
generators", INFORMS J. Comput. 2008): the word sequence x[t] of MT19937 is an F2-linear recurring sequence whose
characteristic polynomial phi (degree 19937) is primitive, so for g(x) = x^B mod phi(x)

        x[t + B] = XOR over the set coefficients i of g of  x[t + i]            (every bit position separately).

Hence the 624-word window at distance B is window_B[j] = XOR_i g_i x[i + j], j = 0..623, computable from the first
19937 + 624 words of the stream.  (Word 0 of a window only contributes its top bit to the recurrence; the jumped
window reproduces exactly those bits that matter - see mt19937_jump_step_kernel in csrc/k_draws.inc.h.)

Steps here: (1) phi by Berlekamp-Massey on 2 x 19937 bits of the generator; (2) g = x^B mod phi by square and
multiply on Python integers used as GF(2)[x] polynomials; (3) self-check against a directly generated stream;
(4) write the set-bit indices of g as a C array.

usage: python scripts/mt_jump_poly.py            (takes a few seconds)
B = 512 state blocks since late round 3 (2048 before), polynomials for B, 4 B, 16 B, 64 B.
"""
import os
import sys

import numpy as np

N, M = 624, 397
DEG = 19937
BLOCKS_PER_JUMP = 512                        # (2048 until late round 3: the raw-word kernel walks a sub-stream with ONE wave)
B = N * BLOCKS_PER_JUMP                      # jump distance in words


def mt_words(key: np.ndarray, nwords: int) -> np.ndarray:
    """Raw (untempered) words x[0 .. nwords): x[0..624) = key, then the recurrence, 227 words at a time."""
    x = np.empty(nwords + 256, dtype=np.uint32)
    x[:N] = key
    i = N
    while i < nwords:
        n = min(227, nwords - i)
        a, b, m = x[i - 624:i - 624 + n], x[i - 623:i - 623 + n], x[i - 227:i - 227 + n]
        y = (a & np.uint32(0x80000000)) | (b & np.uint32(0x7FFFFFFF))
        x[i:i + n] = m ^ (y >> np.uint32(1)) ^ np.where(b & np.uint32(1), np.uint32(0x9908B0DF), np.uint32(0))
        i += n
    return x[:nwords]


def berlekamp_massey(bits):
    """Minimal polynomial of a binary sequence; polynomials as Python ints (bit i = coefficient of x^i).
    Returns C with C(x) = sum c_i x^i, c_0 = 1, such that sum_i c_i s[n - i] = 0 - the connection polynomial - and L."""
    n_bits = len(bits)
    s = 0
    for i, b in enumerate(bits):
        if b:
            s |= 1 << i
    C, Bp, L, m = 1, 1, 0, 1
    # reversed-prefix trick: discrepancy d = sum_{i=0..L} c_i s[n-i] = parity( C & reverse-window )
    # keep R_n = integer whose bit i is s[n - i]  (i.e. the sequence reversed up to n)
    R = 0
    for n in range(n_bits):
        R = (R << 1) | bits[n]
        d = (C & R).bit_count() & 1
        if d:
            T = C
            C ^= Bp << m
            if 2 * L <= n:
                L, Bp, m = n + 1 - L, T, 1
            else:
                m += 1
        else:
            m += 1
    return C, L


def poly_mulmod_x(p, phi, deg):
    p <<= 1
    if (p >> deg) & 1:
        p ^= phi
    return p


_SPREAD = [int("".join(c + "0" for c in format(b, "08b"))[:-1] or "0", 2) for b in range(256)]


def poly_square(p):
    """Squaring in GF(2)[x] = spreading the bits apart."""
    out, shift = 0, 0
    data = p.to_bytes((p.bit_length() + 7) // 8 or 1, "little")
    parts = []
    for byte in data:
        parts.append(_SPREAD[byte])
    for i, v in enumerate(parts):
        if v:
            out |= v << (16 * i)
    return out


def poly_mod(p, phi, deg):
    """p mod phi for deg(p) < 2 deg."""
    for k in range(p.bit_length() - 1, deg - 1, -1):
        if (p >> k) & 1:
            p ^= phi << (k - deg)
    return p


def x_pow_mod(e, phi, deg):
    result = 1
    for bit in bin(e)[2:]:
        result = poly_mod(poly_square(result), phi, deg)
        if bit == "1":
            result = poly_mulmod_x(result, phi, deg)
    return result


def main():
    rng = np.random.RandomState(20220714)
    key = rng.randint(0, 2 ** 32, size=N, dtype=np.uint64).astype(np.uint32)
    key[0] |= np.uint32(0x80000000)
    x = mt_words(key, N + 2 * DEG + 64)
    # (1) characteristic polynomial from one bit sequence (bit 0 of x[t], t >= 1: a functional of the true state)
    bits = [int(v & 1) for v in x[1:1 + 2 * DEG + 2]]
    C, L = berlekamp_massey(bits)
    assert L == DEG, L
    # connection polynomial C: sum_i c_i s[n-i] = 0  ->  characteristic polynomial phi(x) = x^L C(1/x):
    # sum_k phi_k s[t+k] = 0 with phi_k = c_{L-k}
    phi = 0
    for i in range(L + 1):
        if (C >> i) & 1:
            phi |= 1 << (L - i)
    assert (phi >> DEG) & 1 and phi & 1
    # sanity: the recurrence annihilates another bit position too
    idx = [i for i in range(DEG + 1) if (phi >> i) & 1]
    for t in (1, 5, 1000):
        acc = np.uint32(0)
        for i in idx:
            acc ^= x[t + i]
        assert acc == 0, "phi does not annihilate the word sequence"
    print("phi: degree", DEG, "weight", len(idx))
    # (2) jump polynomials for B, 4 B, 16 B and 64 B words (the device starts its sub-streams in log-many rounds with them)
    polys = {}
    xs = mt_words(key, 64 * B + N + DEG + 8)
    for mult in (1, 4, 16, 64):
        g = x_pow_mod(mult * B, phi, DEG)
        gi = [i for i in range(DEG) if (g >> i) & 1]
        print("g = x^%d mod phi: weight %d" % (mult * B, len(gi)))
        # (3) self-check against a directly generated stream
        for toff in (1, 3):
            want = xs[mult * B + toff:mult * B + toff + N]
            got = np.zeros(N, dtype=np.uint32)
            for i in gi:
                got ^= xs[toff + i:toff + i + N]
            assert np.array_equal(got, want), "jump polynomial check failed"
        # window at t = 0: everything but the low 31 bits of word 0
        got = np.zeros(N, dtype=np.uint32)
        for i in gi:
            got ^= xs[i:i + N]
        want = xs[mult * B:mult * B + N]
        assert np.array_equal(got[1:], want[1:]) and (got[0] ^ want[0]) & np.uint32(0x80000000) == 0
        polys[mult] = gi
    print("self-check ok (windows at distance B, 4 B, 16 B, 64 B reproduced)")
    # (4) header
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "code-robchar_amd", "csrc",
                       "mt19937_jump_poly.h")
    with open(out, "w") as fh:
        fh.write("// GENERATED by scripts/mt_jump_poly.py - do not edit.  Set coefficients of g_m(x) = x^(m B) mod phi(x), m = 1, 4, 16, 64;\n"
                 "// phi = the characteristic polynomial of MT19937 (degree 19937), B = %d words = %d state blocks:\n"
                 "//     x[t + m B] = XOR_{i in table m} x[t + i]   for the raw word sequence of the generator.\n"
                 "#pragma once\n" % (B, BLOCKS_PER_JUMP))
        fh.write("constexpr long long kMtJumpWords = %dLL;\n" % B)
        for mult, gi in polys.items():
            fh.write("constexpr int kMtJumpTerms%d = %d;\n" % (mult, len(gi)))
            fh.write("#define RC_MT_JUMP_IDX%d_VALUES \\\n" % mult)
            lines = []
            for k in range(0, len(gi), 16):
                lines.append("    " + ", ".join(str(v) for v in gi[k:k + 16]))
            fh.write(", \\\n".join(lines) + "\n")
    print("wrote", out)


if __name__ == "__main__":
    sys.setrecursionlimit(10000)
    main()
