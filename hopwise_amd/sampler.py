"""Negative samplers with the reference's class names and call signatures, drawing on the device.

Reference: /root/reference/hopwise/sampler/sampler.py -- AbstractSampler :19-183, Sampler :186-291,
KGSampler :294-357.

Same: constructor arguments, ``sample_by_user_ids`` / ``sample_by_entity_ids`` / ``set_phase``, output layout
(value_ids[j * len(keys) + i] is the j-th sample of keys[i]), the support ([1, N) minus the key's used set),
phase-cumulative used sets, the ``ValueError``s (unknown key; a key that has used every id), the
``popularity`` distribution p ~ count^alpha over the ids that occur.

Different by design: draws come from a counter-based Philox stream on the GPU (one independent stream per output
position, rejection inside the kernel) instead of numpy's global RNG, so results are distribution-equal, not
stream-equal, to the reference; used sets live on the device as a sorted CSR instead of Python sets.
"""

from __future__ import annotations

import copy

import numpy as np
import torch

from . import ops
from .utils import seed_for_rank


def _np(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def build_alias_table(candidates, alpha):
    """Walker/Vose alias table of p_i ~ (count_i / total)^alpha over the distinct candidate ids
    (sampler.py:68-94).  Returns (keys int64, prob float32, alias int64)."""
    cand = _np(candidates).astype(np.int64)
    keys, counts = np.unique(cand, return_counts=True)
    p = np.power(counts / len(cand), alpha)
    n = len(keys)
    q = p / p.sum() * n
    alias = np.arange(n)
    small = [i for i in range(n) if q[i] < 1.0]
    large = [i for i in range(n) if q[i] > 1.0]
    q = q.copy()
    while small and large:
        s, l = small.pop(), large.pop()
        alias[s] = l
        q[l] -= 1.0 - q[s]
        if q[l] < 1.0:
            small.append(l)
        elif q[l] > 1.0:
            large.append(l)
    return keys, np.minimum(q, 1.0).astype(np.float32), keys[alias]


class AbstractSampler:
    def __init__(self, distribution, alpha, device, seed=2020):
        self.distribution = ""
        self.alpha = alpha
        self.device = torch.device(device)
        self.seed = seed_for_rank(seed)  # rank folded into the Philox key: ranks draw different negatives
        self._calls = 0  # Philox offset; persisted through state_dict() so a resumed run does not replay epoch 1's stream
        self._alias = None
        self.set_distribution(distribution)

    def set_distribution(self, distribution):
        if distribution not in ("uniform", "popularity"):
            raise NotImplementedError(f"The sampling distribution [{distribution}] is not implemented.")
        self.distribution = distribution
        if distribution == "popularity":
            keys, prob, alias = build_alias_table(self._get_candidates_list(), self.alpha)
            self._alias = (torch.from_numpy(prob).to(self.device), torch.from_numpy(keys).to(self.device),
                           torch.from_numpy(alias).to(self.device))

    def _get_candidates_list(self):
        raise NotImplementedError

    @staticmethod
    def _csr(keys, values, n_keys, value_num, what):
        kv = np.unique(np.stack([_np(keys).astype(np.int64), _np(values).astype(np.int64)], 1), axis=0)
        ptr = np.zeros(n_keys + 1, np.int64)
        np.add.at(ptr, kv[:, 0] + 1, 1)
        np.cumsum(ptr, out=ptr)
        # "[pad] is a item": a key whose used set covers every real id cannot be sampled for (sampler.py:245-251,330-335)
        real = kv[kv[:, 1] != 0]
        cnt = np.bincount(real[:, 0], minlength=n_keys)
        if (cnt + 1 >= value_num).any():
            raise ValueError(f"Some {what} have used all ids, which we can not sample negatives for them.")
        return ptr, kv[:, 1].copy()

    def _set_used(self, ptr, idx):
        self._used_ptr = torch.from_numpy(ptr).to(self.device)
        self._used_idx = torch.from_numpy(idx).to(self.device)

    def sample_by_key_ids(self, key_ids, num, out=None):
        """Device draw; ``key_ids`` may be a host array or a device tensor.  Returns an int64 device tensor."""
        if not isinstance(key_ids, torch.Tensor):
            key_ids = torch.as_tensor(np.asarray(key_ids), dtype=torch.int64)
        keys = key_ids.to(self.device, torch.int64)
        if keys.numel() == 0 and out is None:  # sampler.py:140-183 on no key: np.tile of nothing -> an empty draw
            return torch.empty(0, dtype=torch.int64, device=self.device)
        self._calls += 1
        before = ops.oob_count(self.device) if self.check_keys else 0
        res = ops.neg_sample(self._used_ptr, self._used_idx, keys, num, self.value_num, self.seed, self._calls,
                             alias=self._alias, out=out)
        if self.check_keys and ops.oob_count(self.device) != before:
            raise IndexError("key id out of range")
        return res

    def state_dict(self):
        return {"calls": int(self._calls)}

    def load_state_dict(self, state):
        self._calls = int(state["calls"])

    check_keys = True  # costs one device sync per call; the driver can switch it off after validating a dataset


class Sampler(AbstractSampler):
    """user -> negative items, per phase (sampler.py:186-291)."""

    def __init__(self, phases, datasets, distribution="uniform", alpha=1.0, device="cuda", seed=2020):
        if not isinstance(phases, list):
            phases = [phases]
        if not isinstance(datasets, list):
            datasets = [datasets]
        if len(phases) != len(datasets):
            raise ValueError(f"Phases {phases} and datasets {datasets} should have the same length.")
        self.phases, self.datasets = phases, datasets
        self.uid_field, self.iid_field = datasets[0].uid_field, datasets[0].iid_field
        self.user_num, self.item_num = datasets[0].user_num, datasets[0].item_num
        self.value_num = self.item_num
        super().__init__(distribution, alpha, device, seed)
        self._phase_csr = {}
        us, its = [], []
        for phase, ds in zip(phases, datasets):  # phase-cumulative used sets (sampler.py:229-252)
            us.append(_np(ds.inter_feat[self.uid_field]))
            its.append(_np(ds.inter_feat[self.iid_field]))
            self._phase_csr[phase] = (np.concatenate(us), np.concatenate(its))
        last_u, last_i = self._phase_csr[phases[-1]]
        self._csr(last_u, last_i, self.user_num, self.item_num, "users")  # raises like the reference
        self.phase = None

    def _get_candidates_list(self):
        return np.concatenate([_np(ds.inter_feat[self.iid_field]) for ds in self.datasets])

    def set_phase(self, phase):
        if phase not in self.phases:
            raise ValueError(f"Phase [{phase}] not exist.")
        new = copy.copy(self)
        new.phase = phase
        u, i = self._phase_csr[phase]
        ptr, idx = self._csr(u, i, self.user_num, self.item_num + 1, "users")
        new._set_used(ptr, idx)
        return new

    def sample_by_user_ids(self, user_ids, item_ids, num, out=None):
        try:
            return self.sample_by_key_ids(user_ids, num, out=out)
        except IndexError:
            for user_id in _np(user_ids):
                if user_id < 0 or user_id >= self.user_num:
                    raise ValueError(f"user_id [{user_id}] not exist.")
            raise


class KGSampler(AbstractSampler):
    """head entity -> negative tails (sampler.py:294-357)."""

    def __init__(self, dataset, distribution="uniform", alpha=1.0, device="cuda", seed=2020):
        self.dataset = dataset
        self.hid_list, self.tid_list = _np(dataset.head_entities), _np(dataset.tail_entities)
        self.head_entities = set(self.hid_list.tolist())
        self.entity_num = dataset.entity_num
        self.value_num = self.entity_num
        super().__init__(distribution, alpha, device, seed)
        self._set_used(*self._csr(self.hid_list, self.tid_list, self.entity_num, self.entity_num, "head entities"))

    def _get_candidates_list(self):
        return np.concatenate([self.hid_list, self.tid_list])

    def sample_by_entity_ids(self, head_entity_ids, num=1, out=None):
        try:
            return self.sample_by_key_ids(head_entity_ids, num, out=out)
        except IndexError:
            for h in _np(head_entity_ids):
                if int(h) not in self.head_entities:
                    raise ValueError(f"head_entity_id [{h}] not exist.")
            raise
