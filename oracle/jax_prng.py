"""TEST INFRASTRUCTURE ONLY -- NumPy restatement of JAX's threefry2x32 PRNG.

The only hard-coded golden numbers in the reference live in
``tests/test_reference_results.py:9-140``; their inputs are drawn with
``jax.random.PRNGKey(42)`` / ``split`` / ``normal`` (``:11-16``).  JAX is not
installed in this image, so the *published* Threefry-2x32 generator (Salmon et
al., "Parallel random numbers: as easy as 1, 2, 3", SC'11, 20 rounds) and the
way ``jax.random`` maps its output to float64 normals are restated here.  Two
bit-layouts are provided because ``jax_threefry_partitionable`` flipped its
default in JAX 0.5.0 and the JAX version that produced the reference's golden
numbers is not recorded; ``tests/test_oracle_golden.py`` shows which of them
reproduces the reference's expected values (self-validating: a wrong PRNG can
not reproduce 30 pinned predictions to 1e-5).

Nothing in the product package imports this module.
"""
import numpy as np
from scipy.special import erfinv

_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))
_U32 = np.uint32


def _rotl(x, r):
    return (x << _U32(r)) | (x >> _U32(32 - r))


def threefry2x32(key, x0, x1):
    """20-round Threefry-2x32 block function on two uint32 count arrays."""
    k0, k1 = _U32(key[0]), _U32(key[1])
    ks = (k0, k1, k0 ^ k1 ^ _U32(0x1BD11BDA))
    x0 = x0.astype(_U32).copy()
    x1 = x1.astype(_U32).copy()
    with np.errstate(over="ignore"):
        x0 += ks[0]
        x1 += ks[1]
        for i in range(5):
            for r in _ROT[i % 2]:
                x0 += x1
                x1 = _rotl(x1, r)
                x1 ^= x0
            x0 += ks[(i + 1) % 3]
            x1 += ks[(i + 2) % 3] + _U32(i + 1)
    return x0, x1


def prng_key(seed):
    """``jax.random.PRNGKey(seed)`` for a non-negative 64-bit seed: [hi, lo]."""
    seed = int(seed)
    return np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], dtype=_U32)


def split(key, num=2, partitionable=False):
    """``jax.random.split(key, num)`` -> (num, 2) uint32."""
    if partitionable:
        hi = np.zeros(num, dtype=_U32)
        lo = np.arange(num, dtype=_U32)
        o0, o1 = threefry2x32(key, hi, lo)
        return np.stack([o0, o1], axis=1)
    counts = np.arange(2 * num, dtype=_U32)
    o0, o1 = threefry2x32(key, counts[:num], counts[num:])
    return np.concatenate([o0, o1]).reshape(num, 2)


def random_bits64(key, shape, partitionable=False):
    """``jax._src.prng.threefry_random_bits(key, 64, shape)`` -> uint64."""
    size = int(np.prod(shape))
    if partitionable:
        idx = np.arange(size, dtype=np.uint64)
        hi = (idx >> np.uint64(32)).astype(_U32)
        lo = (idx & np.uint64(0xFFFFFFFF)).astype(_U32)
        b1, b2 = threefry2x32(key, hi, lo)
        bits = (b1.astype(np.uint64) << np.uint64(32)) | b2.astype(np.uint64)
        return bits.reshape(shape)
    counts = np.arange(2 * size, dtype=_U32)
    o0, o1 = threefry2x32(key, counts[:size], counts[size:])
    bits = (o0.astype(np.uint64) << np.uint64(32)) | o1.astype(np.uint64)
    return bits.reshape(shape)


def uniform64(key, shape, minval, maxval, partitionable=False):
    """``jax.random.uniform(key, shape, float64, minval, maxval)``."""
    bits = random_bits64(key, shape, partitionable)
    one = np.float64(1.0).view(np.uint64)
    fbits = (bits >> np.uint64(64 - 52)) | one
    floats = fbits.view(np.float64) - 1.0
    return np.maximum(minval, floats * (maxval - minval) + minval)


def normal64(key, shape, partitionable=False):
    """``jax.random.normal(key, shape)`` under ``jax_enable_x64``."""
    lo = np.nextafter(np.float64(-1.0), np.float64(0.0))
    u = uniform64(key, shape, lo, np.float64(1.0), partitionable)
    return np.sqrt(2.0) * erfinv(u)
