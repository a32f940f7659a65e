"""NumPy restatement of the device noise generator (csrc/synth.hip) -- TEST INFRASTRUCTURE.
Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11) + Box-Muller."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over arrays of uint32 counters; scalar uint32 keys."""
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3)]
    k0, k1 = np.uint32(k0), np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ k0
            n1 = (p1 & MASK).astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ k1
            n3 = (p0 & MASK).astype(np.uint32)
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0, k1 = np.uint32(k0 + W0), np.uint32(k1 + W1)
    return c0, c1, c2, c3


def _unit(x):
    return (x.astype(np.float64) + 0.5) * 2.3283064365386963e-10


def normals(first, n, seed):
    """N(0,1) for the global elements first .. first + n - 1 of the stream keyed by seed."""
    q0, q1 = first // 4, (first + n + 3) // 4
    ctr = np.arange(q0, q1, dtype=np.uint64)
    v = philox4x32_10((ctr & MASK).astype(np.uint32), (ctr >> np.uint64(32)).astype(np.uint32), np.zeros(len(ctr), np.uint32),
                      np.zeros(len(ctr), np.uint32), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    r0, r1 = np.sqrt(-2.0 * np.log(_unit(v[0]))), np.sqrt(-2.0 * np.log(_unit(v[2])))
    a0, a1 = 2.0 * np.pi * _unit(v[1]), 2.0 * np.pi * _unit(v[3])
    z = np.stack([r0 * np.cos(a0), r0 * np.sin(a0), r1 * np.cos(a1), r1 * np.sin(a1)], axis=1).reshape(-1)
    s = first - 4 * q0
    return z[s:s + n]


def nan_mask(first, n, seed, fraction):
    q0, q1 = first // 4, (first + n + 3) // 4
    ctr = np.arange(q0, q1, dtype=np.uint64)
    v = philox4x32_10((ctr & MASK).astype(np.uint32), (ctr >> np.uint64(32)).astype(np.uint32), np.ones(len(ctr), np.uint32),
                      np.zeros(len(ctr), np.uint32), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u = np.stack([_unit(x) for x in v], axis=1).reshape(-1)
    s = first - 4 * q0
    return u[s:s + n] < fraction
