"""Platform-independent pseudo-random arrays for fixtures: pure integer hashing, so the
inputs of a golden case can be rebuilt bit-for-bit anywhere instead of being stored."""
import numpy as np


def hash_u32(n, seed):
    i = np.arange(n, dtype=np.uint64)
    x = i + np.uint64((int(seed) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)  # array ops wrap mod 2^64
    x ^= x >> np.uint64(30)
    x = x * np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(27)
    x = x * np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(31)
    return (x >> np.uint64(32)).astype(np.uint32)


def uniform(shape, seed):
    """float32 in [0, 1) with 24 random bits (exact in float32)."""
    n = int(np.prod(shape))
    return ((hash_u32(n, seed) >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)).reshape(shape)


def normalish(shape, seed):
    """Zero-mean, unit-variance-ish (Irwin-Hall of 4 uniforms); every op exact/IEEE in float32."""
    u = uniform((4,) + tuple(shape), seed)
    return ((u[0] + u[1]) + (u[2] + u[3]) - np.float32(2.0)) * np.float32(1.7320508)
