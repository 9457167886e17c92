"""Host regeneration of the device's counter-based Gaussian stream  --  TEST INFRASTRUCTURE ONLY.

Restates `philox_normal_kernel` (code-robchar_amd/csrc/k_draws.inc.h) in NumPy: Philox4x32-10 (Salmon et al.,
"Parallel random numbers: as easy as 1, 2, 3", SC'11; multipliers 0xD2511F53 / 0xCD9E8D57, Weyl constants
0x9E3779B9 / 0xBB67AE85) keyed by the 64-bit seed, counter = element index >> 1, two 53-bit uniforms, Box-Muller,
element parity selects cos / sin.  This mode has no counterpart in the reference (which draws from numpy's
legacy MT19937 stream); it exists so that device-generated draws can be checked element by element.
"""
import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr_lo, ctr_hi, key_lo, key_hi):
    c0 = ctr_lo.astype(np.uint32); c1 = ctr_hi.astype(np.uint32)
    c2 = np.zeros_like(c0); c3 = np.zeros_like(c0)
    k0 = np.uint32(key_lo); k1 = np.uint32(key_hi)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _M0 * c0.astype(np.uint64)
            p1 = _M1 * c2.astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ k0
            n1 = (p1 & _MASK).astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ k1
            n3 = (p0 & _MASK).astype(np.uint32)
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = np.uint32((int(k0) + int(_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def philox_normal(seed: int, offset: int, n: int, scale: float = 1.0) -> np.ndarray:
    """Elements offset .. offset+n-1 of the Gaussian stream `seed`, times `scale`."""
    e = np.arange(offset, offset + n, dtype=np.uint64)
    ctr = e >> np.uint64(1)
    w0, w1, w2, w3 = philox4x32_10(ctr & _MASK, ctr >> np.uint64(32), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    a = ((w1.astype(np.uint64) << np.uint64(32)) | w0.astype(np.uint64)) >> np.uint64(11)
    b = ((w3.astype(np.uint64) << np.uint64(32)) | w2.astype(np.uint64)) >> np.uint64(11)
    u1 = (a.astype(np.float64) + 0.5) * 2.0 ** -53
    u2 = (b.astype(np.float64) + 0.5) * 2.0 ** -53
    rad = np.sqrt(-2.0 * np.log(u1))
    ang = 6.283185307179586476925286766559 * u2
    return scale * rad * np.where((e & np.uint64(1)).astype(bool), np.sin(ang), np.cos(ang))
