"""numpy statement of the noise stream the kernels generate in-register ("noise
stream"): Philox4x32-R + Box-Muller, R = ROUNDS = 7 (the round count of csrc/common.h: kPhiloxRounds).
TEST INFRASTRUCTURE ONLY (checks reparam.hip / layout.hip).  Pinned by the Random123 known-answer vectors for
philox4x32 with 7 and with 10 rounds (tests/test_oracle_golden.py::test_philox_known_answers)."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)
ROUNDS = 7


def philox4x32(group, offset, seed, rounds=ROUNDS):
    """group: uint64 array; returns uint32 array [len(group), 4].  Counter = (group_lo, group_hi, offset_lo,
    offset_hi), key = (seed_lo, seed_hi)."""
    g = np.asarray(group, dtype=np.uint64)
    c0, c1 = (g & MASK).astype(np.uint32), (g >> np.uint64(32)).astype(np.uint32)
    c2 = np.full_like(c0, np.uint32(offset & 0xFFFFFFFF))
    c3 = np.full_like(c0, np.uint32((offset >> 32) & 0xFFFFFFFF))
    k0, k1 = np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        for _ in range(rounds):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            lo0, hi0 = (p0 & MASK).astype(np.uint32), (p0 >> np.uint64(32)).astype(np.uint32)
            lo1, hi1 = (p1 & MASK).astype(np.uint32), (p1 >> np.uint64(32)).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0, k1 = np.uint32(k0 + W0), np.uint32(k1 + W1)
    return np.stack([c0, c1, c2, c3], axis=1)


def _u01(x):
    return ((x >> np.uint32(8)).astype(np.float64) + 0.5) * 2.0 ** -24


def normals(n_groups, seed, offset, scale=1.0):
    """[n_groups, 4] standard normals (times `scale`) in stream order z0..z3."""
    x = philox4x32(np.arange(n_groups, dtype=np.uint64), offset, seed)
    u = _u01(x)
    r0, r1 = np.sqrt(-2 * np.log(u[:, 0])), np.sqrt(-2 * np.log(u[:, 2]))
    a0, a1 = 2 * np.pi * u[:, 1], 2 * np.pi * u[:, 3]
    return scale * np.stack([r0 * np.cos(a0), r0 * np.sin(a0), r1 * np.cos(a1), r1 * np.sin(a1)], 1)


def real_noise(n, seed, offset):
    return normals((n + 3) // 4, seed, offset).reshape(-1)[:n]


def cplx_noise(n, seed, offset):
    z = normals((n + 1) // 2, seed, offset, scale=np.sqrt(0.5)).reshape(-1, 2)[:n]
    return z[:, 0], z[:, 1]
