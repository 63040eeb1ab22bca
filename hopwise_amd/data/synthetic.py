"""Synthetic knowledge graphs with the shape statistics of the public datasets BASELINE.json names
(the datasets themselves are not in the reference tree and there is no network).

Id conventions are the reference's: row 0 of every table is [PAD]; items are entity ids [0, n_items)
(kg_dataset.py:556-588); [UI-Relation] is the last relation id (kg_dataset.py:626-631).
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np

SHAPES = {
    # name: (n_users, n_items, n_entities, n_relations incl. PAD + UI, kg triples, interactions)
    "ml-1m-small": (6041, 2985, 13805, 13, 65_000, 932_000),
    "yelp2018": (45920, 45539, 90962, 44, 1_853_704, 1_185_068),
    "alibaba-ifashion": (114738, 30041, 59157, 53, 279_155, 1_781_093),
    "tiny": (64, 48, 160, 6, 900, 1500),
}


@dataclass
class SyntheticKG:
    name: str
    n_users: int
    n_items: int
    n_entities: int
    n_relations: int
    inter_users: np.ndarray  # [n_inter] int64 in [1, n_users)
    inter_items: np.ndarray  # [n_inter] int64 in [1, n_items)
    heads: np.ndarray  # [n_triples] int64 in [1, n_entities)
    relations: np.ndarray  # [n_triples] int64 in [1, n_relations - 1)
    tails: np.ndarray  # [n_triples]

    @property
    def ui_relation(self):
        return self.n_relations - 1


def zipf_ids(rng, n, lo, hi, s=1.0):
    """n ids in [lo, hi) with rank-frequency ~ 1/rank^s (rank order shuffled over the id space)."""
    m = hi - lo
    w = 1.0 / np.power(np.arange(1, m + 1, dtype=np.float64), s)
    w /= w.sum()
    ranks = rng.choice(m, size=n, p=w)
    perm = rng.permutation(m)
    return (lo + perm[ranks]).astype(np.int64)


def make(name="ml-1m-small", seed=2024, n_triples=None, n_inter=None) -> SyntheticKG:
    nU, nI, nE, nR, nT, nX = SHAPES[name]
    nT = nT if n_triples is None else n_triples
    nX = nX if n_inter is None else n_inter
    rng = np.random.default_rng(seed)
    users = rng.integers(1, nU, nX).astype(np.int64)
    items = zipf_ids(rng, nX, 1, nI, 1.0)
    ui = np.unique(np.stack([users, items], 1), axis=0)  # one interaction per (user, item)
    heads = zipf_ids(rng, nT, 1, nE, 0.8)
    tails = rng.integers(1, nE, nT).astype(np.int64)
    rels = rng.integers(1, nR - 1, nT).astype(np.int64)
    return SyntheticKG(name, nU, nI, nE, nR, ui[:, 0].copy(), ui[:, 1].copy(), heads, rels, tails)


def used_csr(keys, values, n_keys):
    """CSR of the per-key used sets (sorted, de-duplicated) that the device sampler rejects against
    (AbstractSampler.get_used_ids, sampler.py:229-252 / :321-336)."""
    kv = np.unique(np.stack([np.asarray(keys, np.int64), np.asarray(values, np.int64)], 1), axis=0)
    ptr = np.zeros(n_keys + 1, np.int64)
    np.add.at(ptr, kv[:, 0] + 1, 1)
    np.cumsum(ptr, out=ptr)
    return ptr, kv[:, 1].copy()
