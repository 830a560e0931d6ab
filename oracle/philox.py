"""Test infrastructure (never imported by the package): the counter-based generator behind `randomized=True`, restated in numpy.

The reference draws `torch.rand(batch, num_samples + 1)` for the stratified jitter (intern/ray.py:104) and for the randomized inverse CDF
(intern/ray.py:31).  libm360 draws those uniforms inside its kernels with Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random
numbers: as easy as 1, 2, 3", SC'11 - the generator of Random123 / cuRAND / torch's device RNG), key = seed, counter = (offset lo, offset hi,
element lo, stream << 28 | element hi), and uses the first output word's top 24 bits as a uniform in [0, 1)
(mipnerf360_amd/csrc/m360_common.hip.h: philox4x32_10_x / philox_uniform; include/m360.h: m360_hyper_t.rng_seed).
Pinned by Random123's published known-answer vectors (tests/test_oracle_golden.py::test_philox_known_answers).
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter, key):
    """counter: 4 arrays (or ints) of uint32, key: 2 -> 4 arrays of uint32 (all ten rounds)."""
    c = [np.asarray(v, dtype=np.uint64) & MASK for v in counter]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c[0]
        p1 = M1 * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return [v.astype(np.uint32) for v in c]


def uniform(seed: int, offset: int, stream_id: int, n: int) -> np.ndarray:
    """float32[n]: what the kernels draw for elements 0..n-1 of stream `stream_id` (0 = t_rand, 1 = u_rand) under (seed, offset)."""
    e = np.arange(n, dtype=np.uint64)
    c0 = np.full(n, offset & 0xFFFFFFFF, np.uint64)
    c1 = np.full(n, (offset >> 32) & 0xFFFFFFFF, np.uint64)
    c3 = np.uint64(stream_id << 28) | ((e >> np.uint64(32)) & np.uint64(0x0FFFFFFF))
    x = philox4x32_10((c0, c1, e & MASK, c3), (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))[0]
    return ((x >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)
