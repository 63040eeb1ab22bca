"""numpy Philox4x32-10, word-addressed the way the device kernels use it (TEST INFRASTRUCTURE ONLY).

word(seed, offset, idx) = element idx & 3 of the block with counter (idx >> 2 [64 bit], offset [64 bit]) and key
seed [64 bit] -- hopwise_amd/csrc/hw_common.h:philox_bits.  No reference counterpart (the reference draws from
torch / numpy global generators); used to check the device's random decisions exactly."""

import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox_words(seed, offset, idx):
    idx = np.asarray(idx, np.uint64)
    blk = idx >> np.uint64(2)
    c0, c1 = blk & MASK, (blk >> np.uint64(32)) & MASK
    c2 = np.full_like(c0, np.uint64(offset & 0xFFFFFFFF))
    c3 = np.full_like(c0, np.uint64((offset >> 32) & 0xFFFFFFFF))
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)
        n1 = p1 & MASK
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
        n3 = p0 & MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    w = np.stack([c0, c1, c2, c3], -1)
    return np.take_along_axis(w, (idx & np.uint64(3)).astype(np.int64)[..., None], -1)[..., 0].astype(np.uint32)


def keep_bits_packed(seed, offset, n):
    """Keep bits (0 / 1, p = 1/2) of elements 0 .. n - 1 of the packed dropout stream of csrc/pgpr_policy.hip
    (hw_pgpr_policy_fused, drop_mode 3): element i keeps iff bit (i & 31) of word (i >> 5) & 3 of Philox block i >> 7 is
    set -- 128 keep bits per Philox4x32-10 block instead of philox_words' 4."""
    i = np.arange(n, dtype=np.uint64)
    words = philox_words(seed, offset, ((i >> np.uint64(7)) << np.uint64(2)) | ((i >> np.uint64(5)) & np.uint64(3)))
    return ((words >> (i & np.uint64(31)).astype(np.uint32)) & np.uint32(1)).astype(np.uint8)
