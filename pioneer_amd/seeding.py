"""The reference's seed -> reset-draw stream for the single-env façade.

PioneerKinematicEnv.seed() (pioneer_knm_env.py:107-109) calls ``gym.utils.seeding.np_random(seed)`` and reset_world()
(:80-90) draws ``np_random.uniform(r_lo, r_hi)`` (six joint angles) and then ``np_random.uniform(target_lo, target_hi)``
from it.  gym is a third-party dependency that is absent from the reference tree and UNPINNED in its requirements.txt
(``gym[atari]``); the reference's era (ray 0.8.x, torch >= 1.4: early 2020) is gym 0.15-0.17, whose seeding module stayed
the same through gym 0.21.  Its published algorithm, restated here:

  * a given seed must be a non-negative int; it is reduced modulo 2**64 (``create_seed``; None -> 8 bytes of os.urandom);
  * ``hash_seed``: the first 8 bytes of sha512(str(seed)) read as two little-endian uint32 words (after gym's padding
    quirk, which appends FOUR zero bytes to an 8-byte string, i.e. a third word 0), combined little-endian into one int;
  * that int is split back into base-2**32 digits, least significant first, and given to
    ``numpy.random.RandomState.seed`` (MT19937 ``init_by_array``).

NumPy's legacy RandomState stream is frozen by NumPy's compatibility policy, so the draws below are what the
reference's env produces for the same ``env.seed(s)`` under gym <= 0.21.  (gym >= 0.22 switched to PCG64 Generators; with
such a gym the reference itself draws differently.)  Parity unpinned: neither gym nor the reference can run here.
"""
import hashlib
import os
import struct
from typing import List, Optional, Tuple

import numpy as np


def _bigint_from_bytes(data: bytes) -> int:
    sizeof_int = 4
    padding = sizeof_int - len(data) % sizeof_int            # 4, not 0, for a multiple of four: gym's quirk, kept
    data += b"\0" * padding
    words = struct.unpack("{}I".format(len(data) // sizeof_int), data)
    return sum(2 ** (sizeof_int * 8 * i) * w for i, w in enumerate(words))


def _int_list_from_bigint(bigint: int) -> List[int]:
    if bigint < 0:
        raise ValueError("Seed must be non-negative, not {}".format(bigint))
    if bigint == 0:
        return [0]
    ints = []
    while bigint > 0:
        bigint, mod = divmod(bigint, 2 ** 32)
        ints.append(mod)
    return ints


def create_seed(a: Optional[int] = None, max_bytes: int = 8) -> int:
    if a is None:
        return _bigint_from_bytes(os.urandom(max_bytes))
    if isinstance(a, (int, np.integer)):
        return int(a) % 2 ** (8 * max_bytes)
    raise ValueError("Invalid type for seed: {} ({})".format(type(a), a))


def hash_seed(seed: Optional[int] = None, max_bytes: int = 8) -> int:
    if seed is None:
        seed = create_seed(max_bytes=max_bytes)
    return _bigint_from_bytes(hashlib.sha512(str(seed).encode("utf8")).digest()[:max_bytes])


def np_random(seed: Optional[int] = None) -> Tuple[np.random.RandomState, int]:
    """gym.utils.seeding.np_random of gym <= 0.21: (RandomState, the seed actually used)."""
    if seed is not None and not (isinstance(seed, (int, np.integer)) and 0 <= seed):
        raise ValueError("Seed must be a non-negative integer or omitted, not {}".format(seed))
    seed = create_seed(seed)
    rng = np.random.RandomState()
    rng.seed(_int_list_from_bigint(hash_seed(seed)))
    return rng, seed
