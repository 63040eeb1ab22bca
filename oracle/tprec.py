"""numpy restatement of TPRec on top of the PGPR oracle (TEST INFRASTRUCTURE ONLY; see oracle/__init__.py).

Reference: /root/reference/hopwise/model/knowledge_aware_recommender/tprec.py
  _get_actions :408-483 (trim scores with the per-user temporal vector), _get_reward :532-557, _build_interacted_matrix
  :621-632 (one shared accumulator), collect_scores :704-771, calculate_loss_transe :295-334, expand_paths / _has_pattern
  :168-197, :387-389 (no pattern can match).
Pinned by tests/golden/tprec.npz (the reference class on its PGPR fixture with fabricated temporal weights).
"""

from __future__ import annotations

import numpy as np

from .pgpr import USER
from .pgpr_train import PGPRTrainOracle

F32 = np.float32
PAIRWISE_EPS = 1e-6


def uc_weight_from_arrays(users, clusters, weights):
    """{user: {cluster: weight}} with the insertion order of the flat arrays (the reference dict's iteration order)."""
    out = {}
    for u, c, w in zip(users, clusters, weights):
        out.setdefault(int(u), {})[int(c)] = float(w)
    return out


def temporal_vector(uc_weight, clust_emb, uid):
    acc = None
    for c, w in uc_weight[uid].items():
        term = np.asarray(clust_emb, F32)[c] * w
        acc = term if acc is None else acc + term
    return acc


def interacted_constant(uc_weight, clust_emb):
    acc = None
    for uid in range(1, len(uc_weight) + 1):
        for c, w in uc_weight[uid].items():
            term = np.asarray(clust_emb, F32)[c] * w
            if acc is None:
                acc = term.copy()
            else:
                acc += term
    return acc


class TPRecOracle(PGPRTrainOracle):
    def __init__(self, *args, uc_weight=None, clust_emb=None, **kw):
        super().__init__(*args, **kw)
        self.uc_weight, self.C = uc_weight, np.asarray(clust_emb, F32)
        self.patterns = set()  # nothing matches in the reference
        self.const = interacted_constant(uc_weight, self.C)

    def _cand_scores(self, path, cand):  # tprec.py:450-471
        ctype = path[-1][1]
        uid = path[0][2]
        u = self.U[uid]
        item_emb = temporal_vector(self.uc_weight, self.C, uid)
        scores = np.empty(len(cand), np.float64)
        for i, (r, n) in enumerate(cand):
            nt = self.g.next_type(ctype, r)
            if nt == USER:
                src, emb = u, self.U[n]
            elif n < self.n_items:
                src, emb = u + item_emb, self.E[n]
            else:
                src, emb = u + item_emb + self.R[r], self.E[n]
            scores[i] = np.matmul(src, emb)
        return scores

    def path_scores(self):  # tprec.py:706-710
        inter = np.concatenate([np.zeros((1, self.d)), np.tile(self.const, (len(self.U) - 1, 1))])
        return np.dot(self.U + inter, self.E[: self.n_items].T)


def pretrain_loss(U, E, R, C, labels, inter, margin):
    """calculate_loss_transe (tprec.py:295-334): (loss float64, d loss / d C float64)."""
    U, E, R = (np.asarray(x, np.float64) for x in (U, E, R))
    C64 = np.asarray(C, F32).astype(np.float64)
    user, pos, neg = inter["user_id"], inter["item_id"], inter["neg_item_id"]
    lab = [labels[(int(u), int(i))] for u, i in zip(user, pos)]
    num_id, num_index, num_n = np.unique(lab, return_index=True, return_counts=True)
    assert len(num_id) > 1, "the reference raises on single-cluster batches"

    def dist(a, b):
        return np.sqrt(((a - b + PAIRWISE_EPS) ** 2).sum(-1))

    def unit(a, b):
        d = a - b + PAIRWISE_EPS
        n = np.sqrt((d * d).sum(-1, keepdims=True))
        return np.where(n > 0, d / np.where(n > 0, n, 1), 0.0)

    loss, gC = 0.0, np.zeros_like(C64)
    for c, s, cnt in zip(num_id, num_index, num_n):
        sl = slice(int(s), int(s) + int(cnt))
        a = U[user[sl]] + C64[c]
        p, n = E[pos[sl]], E[neg[sl]]
        l = dist(a, p) - dist(a, n) + margin
        act = (l > 0).astype(np.float64) / len(l)
        loss += np.maximum(l, 0).mean()
        gC[c] += (act[:, None] * (unit(a, p) - unit(a, n))).sum(0)
    h = np.concatenate([U[user] + R[-1], E[inter["head_id"]] + R[inter["relation_id"]]])
    pt = np.concatenate([E[pos], E[inter["tail_id"]]])
    nt = np.concatenate([E[neg], E[inter["neg_tail_id"]]])
    loss += np.maximum(dist(h, pt) - dist(h, nt) + margin, 0).mean()
    return loss, gC
