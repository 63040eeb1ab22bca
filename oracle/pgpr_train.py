"""numpy restatement of PGPR training: action dropout, Categorical step, rewards, update() and its gradients.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned against the reference by tests/golden/pgpr_train_*.npz
(reference calculate_loss + backward on its own fixture with every random draw recorded).

Reference: /root/reference/hopwise/model/knowledge_aware_recommender/pgpr.py
  select_action :120-135, update :137-173, calculate_loss :189-203, _get_reward :339-357, batch_step :363-382,
  batch_action_mask :384-396; torch.distributions.Categorical (clamp_probs / entropy) for log_prob and entropy.
"""

from __future__ import annotations

import numpy as np

from .pgpr import ENTITY, F32, SELF_LOOP, USER, PGPROracle, elu
from .philox import philox_words

EPS = np.finfo(np.float32).eps


def dropout_keep_mask(n_acts, act_dim, dropout, seed, offset):
    """The device's batch_action_mask: keep action 0 and the int((n-1)(1-p)) others with the smallest Philox keys."""
    B = len(n_acts)
    mask = np.zeros((B, act_dim), np.uint8)
    for row, n in enumerate(n_acts):
        n = int(min(n, act_dim))
        if not dropout > 0 or n < 5:
            mask[row, :n] = 1
            continue
        keep = int((n - 1) * (1.0 - dropout))
        keys = philox_words(seed, offset, row * act_dim + np.arange(n)).astype(np.int64)
        order = np.lexsort((np.arange(1, n), keys[1:]))  # by key, then index
        mask[row, 0] = 1
        mask[row, 1 + order[:keep]] = 1
    return mask


def inverse_cdf_draw(probs, seed, offset):
    """The device's Categorical draw: u = top 24 bits of word `row`; first j with float64 cumsum_j > u * total."""
    B = probs.shape[0]
    u = (philox_words(seed, offset, np.arange(B)) >> np.uint32(8)).astype(np.float64) / 16777216.0
    p = probs.astype(np.float64)
    cs = np.cumsum(p, axis=1)
    thr = u * cs[:, -1]
    acts = np.empty(B, np.int64)
    for r in range(B):
        hit = np.nonzero((p[r] > 0) & (cs[r] > thr[r]))[0]
        acts[r] = hit[0] if len(hit) else np.nonzero(p[r] > 0)[0][-1]
    return acts, u


class PGPRTrainOracle(PGPROracle):
    def __init__(self, *args, patterns=(), gamma=0.99, weight_factor=1e-3, max_path_len=3, **kw):
        super().__init__(*args, **kw)
        self.start_user_revisitable = True
        self.patterns = {tuple(int(x) for x in p[1 : int(p[0])]) for p in patterns}  # relation ids after the self loop
        self.pattern_lens = {int(p[0]) for p in patterns}
        self.gamma, self.weight_factor, self.T = float(gamma), float(weight_factor), int(max_path_len)
        ups = np.dot(self.U + self.R[self.g.ui_relation], self.E[: self.n_items].T)
        self.u_p_scales = ups.max(axis=1)  # pgpr.py:85-88

    def reward(self, path):  # pgpr.py:339-357
        if len(path) <= 2:
            return 0.0
        if tuple(int(v[0]) for v in path[1:]) not in self.patterns:
            return 0.0
        _, t, n = path[-1]
        if t == ENTITY and n < self.n_items:
            uid = path[0][2]
            s = np.dot(self.U[uid] + self.R[self.g.ui_relation], self.E[n]) / self.u_p_scales[uid]
            return max(float(s), 0.0)
        return 0.0

    def forward_full(self, state, mask, drop1, drop2):
        W = self.W
        pre1 = state.astype(F32) @ W["l1.weight"].T + W["l1.bias"]
        x1 = elu(pre1) * (1 if drop1 is None else drop1)
        pre2 = x1 @ W["l2.weight"].T + W["l2.bias"]
        x2 = elu(pre2) * (1 if drop2 is None else drop2)
        logits = (x2 @ W["actor.weight"].T + W["actor.bias"]).astype(F32)
        z = np.where(mask, logits, F32(-np.inf)).astype(F32)
        e = np.exp(z - z.max(axis=1, keepdims=True))
        probs = (e / e.sum(axis=1, keepdims=True)).astype(F32)
        value = (x2 @ W["critic.weight"].T + W["critic.bias"])[:, 0]
        return dict(state=state.astype(F32), pre1=pre1, x1=x1.astype(F32), pre2=pre2, x2=x2.astype(F32), probs=probs,
                    value=value.astype(F32), drop1=drop1, drop2=drop2)

    def episode(self, users, choose, masks=None, drops=None):
        """choose(t, probs, mask) -> acts [B]; masks[t] uint8 [B, act_dim] (None = every available action);
        drops[t] = (drop1, drop2) or None."""
        paths = [[(SELF_LOOP, USER, int(u))] for u in users]
        steps = []
        for t in range(self.T):
            acts_pool = [self.get_actions(p) for p in paths]
            avail = self.acts_to_mask(acts_pool)
            mask = avail if masks is None else np.asarray(masks[t]).astype(bool)
            state = np.vstack([self.get_state(p) for p in paths])
            d1, d2 = drops[t] if drops is not None else (None, None)
            fw = self.forward_full(state, mask, d1, d2)
            acts = np.asarray(choose(t, fw["probs"], mask), np.int64).copy()
            acts[~mask[np.arange(len(acts)), acts]] = 0  # pgpr.py:129-131
            p = fw["probs"].astype(F32)
            pn = (p / p.sum(axis=1, keepdims=True, dtype=F32)).astype(F32)
            lg = np.log(np.clip(pn, EPS, 1 - EPS)).astype(F32)
            fw["logp"] = lg[np.arange(len(acts)), acts]
            fw["entropy"] = -(lg * pn).sum(axis=1)
            for i, a in enumerate(acts):
                rel, nid = acts_pool[i][a]
                ctype = paths[i][-1][1]
                ntype = ctype if rel == SELF_LOOP else self.g.next_type(ctype, rel)
                paths[i] = paths[i] + [(rel, ntype, nid)]
            fw.update(acts=acts, mask=mask, n_acts=np.array([len(a) for a in acts_pool]),
                      reward=np.array([self.reward(p) for p in paths], np.float64))
            steps.append(fw)
        return steps, paths

    def update(self, steps):
        """(loss, actor, critic, entropy) float64 and the gradients of the eight policy tensors (float64)."""
        T, B = len(steps), len(steps[0]["acts"])
        R = np.stack([s["reward"] for s in steps], 1).copy()  # [B, T]
        for i in range(1, T):
            R[:, T - i - 1] += self.gamma * R[:, T - i]
        W = {k: v.astype(np.float64) for k, v in self.W.items()}
        grads = {k: np.zeros_like(v) for k, v in W.items()}
        actor = critic = ent = 0.0
        for t, s in enumerate(steps):
            adv = R[:, t] - s["value"].astype(np.float64)
            actor += (-s["logp"].astype(np.float64) * adv).mean()
            critic += (adv ** 2).mean()
            ent += (-s["entropy"].astype(np.float64)).mean()
            # backward, float64: dz = p (g - sum p g) with g from the clamped-log terms
            p = s["probs"].astype(np.float64)
            inr = (p >= EPS) & (p <= 1 - EPS)
            lp = np.log(np.clip(p, EPS, 1 - EPS))
            g = (self.weight_factor / B) * (lp + inr)
            rows = np.arange(B)
            a = s["acts"]
            g[rows, a] -= np.where(inr[rows, a], (adv / B) / p[rows, a], 0.0)
            dz = p * (g - (p * g).sum(axis=1, keepdims=True))
            dv = (-2.0 * adv / B)[:, None]
            x2, x1, st = (s[k].astype(np.float64) for k in ("x2", "x1", "state"))
            grads["actor.weight"] += dz.T @ x2
            grads["actor.bias"] += dz.sum(0)
            grads["critic.weight"] += dv.T @ x2
            grads["critic.bias"] += dv.sum(0)
            dx2 = dz @ W["actor.weight"] + dv @ W["critic.weight"]
            m2 = 1.0 if s["drop2"] is None else s["drop2"]
            dpre2 = dx2 * m2 * np.where(s["pre2"] > 0, 1.0, np.exp(np.minimum(s["pre2"], 0)))
            grads["l2.weight"] += dpre2.T @ x1
            grads["l2.bias"] += dpre2.sum(0)
            dx1 = dpre2 @ W["l2.weight"]
            m1 = 1.0 if s["drop1"] is None else s["drop1"]
            dpre1 = dx1 * m1 * np.where(s["pre1"] > 0, 1.0, np.exp(np.minimum(s["pre1"], 0)))
            grads["l1.weight"] += dpre1.T @ st
            grads["l1.bias"] += dpre1.sum(0)
        loss = actor + critic + self.weight_factor * ent
        return np.array([loss, actor, critic, ent]), grads
