"""numpy restatement of the reference negative samplers (TEST INFRASTRUCTURE ONLY).

Reference: /root/reference/hopwise/sampler/sampler.py
  AbstractSampler._build_alias_table :68-94, _pop_sampling :96-116, sample_by_key_ids :140-183
  Sampler (user -> item) :186-291, KGSampler (head -> tail) :294-357

The restatement draws from numpy's *global* RNG in the same call order as the reference, so
under ``np.random.seed(s)`` it is stream-exact (pinned by tests/golden/sampler.npz).  The
device sampler of the product is distribution-equal, not stream-equal.
"""

from __future__ import annotations

import numpy as np


class AliasTable:
    """Walker alias table over the distinct ids of ``candidates`` with p ~ count^alpha (sampler.py:68-94)."""

    def __init__(self, candidates, alpha):
        cand = np.asarray(candidates)
        # Counter() keeps first-seen order; the reference then iterates dicts in that order.
        _, first = np.unique(cand, return_index=True)
        self.keys = cand[np.sort(first)]
        pos = {int(k): i for i, k in enumerate(self.keys)}
        cnt = np.zeros(len(self.keys), np.float64)
        for c in cand:
            cnt[pos[int(c)]] += 1
        prob = [pow(c / len(cand), alpha) for c in cnt]
        norm = sum(prob)
        n = len(prob)
        alias = [-1] * n
        large, small = [], []
        for i in range(n):
            prob[i] = prob[i] / norm * n
            if prob[i] > 1:
                large.append(i)
            elif prob[i] < 1:
                small.append(i)
        while large and small:
            lq = large.pop(0)
            sq = small.pop(0)
            alias[sq] = lq
            prob[lq] = prob[lq] - (1 - prob[sq])
            if prob[lq] < 1:
                small.append(lq)
            elif prob[lq] > 1:
                large.append(lq)
        self.prob = np.array(prob, np.float64)
        # alias -1 stays the literal id -1 in the reference (dict value), reachable only through rounding
        self.alias = np.array([self.keys[a] if a >= 0 else -1 for a in alias], np.int64)

    def sample(self, n):
        idx = np.random.randint(0, len(self.keys), n)
        pr = np.random.random(n)
        take = self.prob[idx] > pr
        return np.where(take, self.keys[idx], self.alias[idx]).astype(np.int64)


class OracleSampler:
    """sample_by_key_ids over ids in [1, value_num) rejecting ``used[key]`` (sampler.py:140-183)."""

    def __init__(self, used_sets, value_num, distribution="uniform", alpha=1.0, candidates=None):
        self.used = used_sets  # list[set[int]] indexed by key id
        self.value_num = int(value_num)
        self.distribution = distribution
        if distribution == "popularity":
            self.alias = AliasTable(candidates, alpha)
        for s in used_sets:
            if len(s) + 1 == self.value_num:  # sampler.py:245-251, :330-335
                raise ValueError("a key has used every id; cannot sample negatives for it")

    def sampling(self, n):
        if self.distribution == "uniform":
            return np.random.randint(1, self.value_num, n)
        return self.alias.sample(n)

    def sample_by_key_ids(self, key_ids, num):
        key_ids = np.array(key_ids)
        key_num = len(key_ids)
        total = key_num * num
        for k in key_ids:
            if k < 0 or k >= len(self.used):
                raise ValueError(f"key id [{k}] not exist.")
        if (key_ids == key_ids[0]).all():
            used = np.array(list(self.used[key_ids[0]]))
            value_ids = self.sampling(total)
            check = np.arange(total)[np.isin(value_ids, used)]
            while len(check) > 0:
                value_ids[check] = value = self.sampling(len(check))
                check = check[np.isin(value, used)]
        else:
            value_ids = np.zeros(total, dtype=np.int64)
            check = np.arange(total)
            tiled = np.tile(key_ids, num)
            while len(check) > 0:
                value_ids[check] = self.sampling(len(check))
                check = np.array([i for i in check if value_ids[i] in self.used[tiled[i]]], dtype=np.int64)
        return value_ids.astype(np.int64)


def kg_used_sets(heads, tails, entity_num):
    """KGSampler.get_used_ids (sampler.py:321-336): used[head] = set of its tails."""
    used = [set() for _ in range(int(entity_num))]
    for h, t in zip(heads, tails):
        used[int(h)].add(int(t))
    return used


def rec_used_sets(phase_users, phase_items, user_num):
    """Sampler.get_used_ids (sampler.py:229-252): phase-cumulative used item sets, one list per phase."""
    out = []
    last = [set() for _ in range(int(user_num))]
    for users, items in zip(phase_users, phase_items):
        cur = [set(s) for s in last]
        for u, i in zip(users, items):
            cur[int(u)].add(int(i))
        out.append(cur)
        last = cur
    return out


def make_kg_sampler(heads, tails, entity_num, distribution="uniform", alpha=1.0):
    cand = np.concatenate([np.asarray(heads), np.asarray(tails)])  # sampler.py:318-319
    return OracleSampler(kg_used_sets(heads, tails, entity_num), entity_num, distribution, alpha, cand)


def make_rec_samplers(phase_users, phase_items, user_num, item_num, distribution="uniform", alpha=1.0):
    cand = np.concatenate([np.asarray(x) for x in phase_items])  # sampler.py:220-224 (all phases)
    return [
        OracleSampler(u, item_num, distribution, alpha, cand) for u in rec_used_sets(phase_users, phase_items, user_num)
    ]
