"""numpy restatement of PGPR inference: actions, states, policy forward, beam search, collect_scores.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned against the reference by
tests/golden/pgpr_*.npz (reference run on its own tests/test_data fixture, dropout patched to identity
or with recorded masks).

Reference: /root/reference/hopwise/model/knowledge_aware_recommender/pgpr.py
  _get_actions :226-290, _get_next_node_type :209-213, _get_state :296-333, KGState :555-580,
  forward :175-187, _batch_acts_to_masks :398-403, beam_search :440-484, collect_scores :486-552
and /root/reference/hopwise/data/dataset/kg_dataset.py:1044-1095 (ckg_dict_graph).

Encoding used here and by the device path: node type 0 = user, 1 = entity; the self-loop
"relation" is -1; a path is a list of (relation, type, id) triples starting with (-1, 0, uid).
"""

from __future__ import annotations

from collections import defaultdict

import numpy as np

USER, ENTITY = 0, 1
SELF_LOOP = -1
F32 = np.float32


class CKG:
    """Collaborative KG adjacency with multiplicity (ckg_dict_graph): user->item and item->user under the
    UI relation, every KG triple in both directions under the SAME relation id.  Stored as two CSRs whose
    rows are sorted by (relation, neighbour) -- the order PGPR puts un-trimmed actions in (pgpr.py:259)."""

    def __init__(self, n_users, n_entities, ui_relation, inter_users, inter_items, heads, rels, tails,
                 ui_bidirectional=True):
        self.n_users, self.n_entities, self.ui_relation = int(n_users), int(n_entities), int(ui_relation)
        iu, ii = np.asarray(inter_users, np.int64), np.asarray(inter_items, np.int64)
        h, r, t = np.asarray(heads, np.int64), np.asarray(rels, np.int64), np.asarray(tails, np.int64)
        ui = np.full(len(iu), ui_relation, np.int64)
        nx, nt = len(iu), len(h)
        # user rows: (user) -[ui]-> item, appended in interaction order
        self.user_ptr, self.user_rel, self.user_dst, self.user_ins = self._csr(iu, ui, ii, np.arange(nx), n_users)
        # entity rows: item -[ui]-> user (interaction k at sequence k); triple k appends h->t then t->h
        if ui_bidirectional:
            src = np.concatenate([ii, h, t])
            rel = np.concatenate([ui, r, r])
            dst = np.concatenate([iu, t, h])
            seq = np.concatenate([np.arange(nx), nx + 2 * np.arange(nt), nx + 2 * np.arange(nt) + 1])
        else:  # CAFE: dataset.ckg_dict_graph(ui_bidirectional=False), cafe.py:86
            src, rel, dst = np.concatenate([h, t]), np.concatenate([r, r]), np.concatenate([t, h])
            seq = np.concatenate([2 * np.arange(nt), 2 * np.arange(nt) + 1])
        self.ent_ptr, self.ent_rel, self.ent_dst, self.ent_ins = self._csr(src, rel, dst, seq, n_entities)

    @staticmethod
    def _csr(src, rel, dst, seq, n):
        order = np.lexsort((dst, rel, src))
        s_src, s_rel, s_dst, s_seq = src[order], rel[order], dst[order], seq[order]
        ptr = np.zeros(n + 1, np.int64)
        np.add.at(ptr, s_src + 1, 1)
        np.cumsum(ptr, out=ptr)
        # ckg_dict_graph order inside a row: relations by first appearance, then append order
        ins = np.empty(len(order), np.int64)
        for node in np.unique(s_src):
            lo, hi = ptr[node], ptr[node + 1]
            first = {}
            for rr, q in zip(s_rel[lo:hi], s_seq[lo:hi]):
                first[rr] = min(first.get(rr, q), q)
            keys = sorted(range(hi - lo), key=lambda i: (first[s_rel[lo + i]], s_seq[lo + i]))
            ins[lo:hi] = lo + np.array(keys, np.int64)
        return ptr, s_rel.copy(), s_dst.copy(), ins

    def neighbours(self, ntype, nid, order="sorted"):
        """(relations, neighbours) of a node, sorted by (relation, neighbour) or in the reference's
        dict-of-lists insertion order."""
        if ntype == USER:
            ptr, rel, dst, ins = self.user_ptr, self.user_rel, self.user_dst, self.user_ins
        else:
            ptr, rel, dst, ins = self.ent_ptr, self.ent_rel, self.ent_dst, self.ent_ins
        lo, hi = ptr[nid], ptr[nid + 1]
        if order == "sorted":
            return rel[lo:hi], dst[lo:hi]
        sel = ins[lo:hi]
        return rel[sel], dst[sel]

    def next_type(self, cur_type, rel):  # pgpr.py:209-213
        return USER if (cur_type == ENTITY and rel == self.ui_relation) else ENTITY


def elu(x):
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0))).astype(F32)


class PGPROracle:
    def __init__(self, graph: CKG, n_items, user_emb, entity_emb, relation_emb, positives, weights,
                 max_acts=250, state_history=1, fix_scores_sorting_bug=False, topk=(10,), tie_rule="stable"):
        # tie_rule decides which candidates survive the top-max_acts trim when float64 scores tie
        # (frequent: an item reached through two relations scores the same, pgpr.py:271-272):
        #   "numpy"  -- candidates in the reference's insertion order through np.argsort's default (unstable)
        #               sort: what the reference does on THIS machine; used to pin the oracle to the goldens;
        #   "stable" -- the build's convention, platform independent: candidates in (relation, neighbour)
        #               order, rank by (score, position) -- the later position wins a tie.
        self.tie_rule = tie_rule
        self.start_user_revisitable = False
        self.g = graph
        self.n_items = int(n_items)
        self.U = np.asarray(user_emb, np.float64)
        self.E = np.asarray(entity_emb, np.float64)
        self.R = np.asarray(relation_emb, np.float64)
        self.d = self.U.shape[1]
        self.positives = [set(int(x) for x in row) for row in positives]  # history_item_matrix rows (0 = pad incl.)
        self.W = {k: np.asarray(v, F32) for k, v in weights.items()}
        self.max_acts = int(max_acts)
        self.act_dim = self.max_acts + 1
        self.state_history = int(state_history)
        self.fix_bug = bool(fix_scores_sorting_bug)
        self.topk = tuple(topk)

    # ---- A10 -------------------------------------------------------------------------------------
    def _candidates(self, path, order):
        _, ctype, cid = path[-1]
        rels, dsts = self.g.neighbours(ctype, cid, order)
        # calculate_loss keeps the start user id as a 0-dim tensor (reset(), pgpr.py:215-216): its hash never equals an
        # int's, so training episodes (start_user_revisitable) do not count path[0] as visited; beam_search does.
        # The tensor id also rides along self-loop actions taken at the start (pgpr.py:229), so the whole leading
        # run of self loops is invisible to the filter.
        lead = 0
        if self.start_user_revisitable:
            lead = 1
            while lead < len(path) and path[lead][0] == SELF_LOOP:
                lead += 1
        visited = {(t, n) for _, t, n in path[lead:]}
        return [(int(r), int(n)) for r, n in zip(rels, dsts) if (self.g.next_type(ctype, r), int(n)) not in visited]

    def _cand_scores(self, path, cand):
        ctype = path[-1][1]
        u = self.U[path[0][2]]
        ui = self.R[self.g.ui_relation]
        scores = np.empty(len(cand), np.float64)
        for i, (r, n) in enumerate(cand):
            nt = self.g.next_type(ctype, r)
            if nt == USER:
                src, emb = u, self.U[n]
            elif n < self.n_items:
                src, emb = u + ui, self.E[n]
            else:
                src, emb = u + ui + self.R[r], self.E[n]
            scores[i] = np.matmul(src, emb)
        return scores

    def get_actions(self, path):
        """[(rel, node)] with action 0 the self loop (pgpr.py:226-290)."""
        actions = [(SELF_LOOP, int(path[-1][2]))]
        cand = self._candidates(path, "insertion" if self.tie_rule == "numpy" else "sorted")
        if not cand:
            return actions
        if len(cand) <= self.max_acts:
            actions.extend(sorted(cand))
            return actions
        scores = self._cand_scores(path, cand)
        if self.tie_rule == "numpy":
            idxs = np.argsort(scores)[-self.max_acts:]
        else:
            idxs = np.argsort(scores, kind="stable")[-self.max_acts:]
        if self.fix_bug:
            actions.extend(cand[i] for i in idxs[::-1])
        else:
            actions.extend(sorted(cand[i] for i in idxs))
        return actions

    def action_scores(self, path):
        """(candidates, float64 scores) of the trimming branch -- lets tests reason about ties."""
        cand = self._candidates(path, "sorted")
        return cand, self._cand_scores(path, cand)

    # ---- A11 -------------------------------------------------------------------------------------
    def _emb(self, ntype, nid):
        return self.U[nid] if ntype == USER else self.E[nid]

    def get_state(self, path):
        """float64 state vector (pgpr.py:296-333, KGState :568-580)."""
        z = np.zeros(self.d)
        u = self.U[path[0][2]]
        if len(path) == 1:
            parts = [u, u, z, z, z, z]
        else:
            older_rel, ltype, lid = path[-2]
            last_rel, ctype, cid = path[-1]
            cur, last = self._emb(ctype, cid), self._emb(ltype, lid)
            lrel = self.R[last_rel] if last_rel != SELF_LOOP else z
            if len(path) == 2:
                parts = [u, cur, last, lrel, z, z]
            else:
                _, otype, oid = path[-3]
                orel = z if older_rel == SELF_LOOP else self.R[older_rel]
                parts = [u, cur, last, lrel, self._emb(otype, oid), orel]
        n = {0: 2, 1: 4, 2: 6}[self.state_history]
        return np.concatenate(parts[:n])

    # ---- A12 -------------------------------------------------------------------------------------
    def forward(self, state, mask, drop1=None, drop2=None):
        """probs [n, act_dim] float32, values [n, 1].  drop* are the multiplicative dropout masks
        (entries 0 or 1/(1-p)); None = dropout patched to identity."""
        W = self.W
        x = state.astype(F32) @ W["l1.weight"].T + W["l1.bias"]
        x = elu(x)
        if drop1 is not None:
            x = x * drop1
        x = x @ W["l2.weight"].T + W["l2.bias"]
        x = elu(x)
        if drop2 is not None:
            x = x * drop2
        logits = x @ W["actor.weight"].T + W["actor.bias"]
        logits = np.where(mask, logits, F32(-np.inf)).astype(F32)
        m = logits.max(axis=1, keepdims=True)
        e = np.exp(logits - m)
        probs = (e / e.sum(axis=1, keepdims=True)).astype(F32)
        values = x @ W["critic.weight"].T + W["critic.bias"]
        return probs, values

    def acts_to_mask(self, acts_pool):
        m = np.zeros((len(acts_pool), self.act_dim), bool)
        for i, a in enumerate(acts_pool):
            m[i, : len(a)] = True
        return m

    # ---- A13 -------------------------------------------------------------------------------------
    def beam_search(self, users, hops=(25, 5, 1), drops=None):
        """Returns (path_pool, probs_pool, trace).  ``drops``: optional list per hop of (drop1, drop2)."""
        path_pool = [[(SELF_LOOP, USER, int(u))] for u in users]
        probs_pool = [[] for _ in users]
        state_pool = np.vstack([self.get_state(p) for p in path_pool])
        trace = []
        for hop, k in enumerate(hops):
            acts_pool = [self.get_actions(p) for p in path_pool]
            mask = self.acts_to_mask(acts_pool)
            d1, d2 = drops[hop] if drops is not None else (None, None)
            probs, _ = self.forward(state_pool.astype(F32), mask, d1, d2)
            probs = (probs + mask.astype(F32)).astype(F32)  # pgpr.py:453
            order = np.lexsort((np.broadcast_to(np.arange(probs.shape[1]), probs.shape), -probs.astype(np.float64)),
                               axis=1)[:, :k]
            topv = np.take_along_axis(probs, order, axis=1)
            trace.append(dict(acts=acts_pool, probs=probs, topk_idx=order, topk_val=topv))
            new_paths, new_probs = [], []
            for row in range(order.shape[0]):
                path, pr = path_pool[row], probs_pool[row]
                for idx, p in zip(order[row], topv[row]):
                    if idx >= len(acts_pool[row]):
                        continue
                    rel, nid = acts_pool[row][idx]
                    ntype = path[-1][1] if rel == SELF_LOOP else self.g.next_type(path[-1][1], rel)
                    new_paths.append(path + [(rel, ntype, nid)])
                    new_probs.append(pr + [p])
            path_pool, probs_pool = new_paths, new_probs
            if hop < 2:  # pgpr.py:481 (hard-wired to 3 hops)
                state_pool = np.vstack([self.get_state(p) for p in path_pool]) if path_pool else np.zeros((0, 1))
        return path_pool, probs_pool, trace

    # ---- A14 -------------------------------------------------------------------------------------
    def path_scores(self):
        return np.dot(self.U + self.R[self.g.ui_relation], self.E[: self.n_items].T)

    def collect_scores(self, users, paths, probs):
        """(results [len(users), n_items] float32 with -inf, explanations list) -- pgpr.py:486-552 incl. its quirks:
        rows are numbered by order of first user with a surviving item, the padded items are paired with the
        wrong paths and truncated by zip()."""
        kmax = max(self.topk)
        users = [int(u) for u in users]
        pred = {u: defaultdict(list) for u in users}
        ps = self.path_scores()
        for path, prob in zip(paths, probs):
            if any(node[0] == SELF_LOOP for node in path[1:]):
                continue
            if path[-1][1] != ENTITY:
                continue
            uid, pid = path[0][2], path[-1][2]
            if uid not in pred or not (pid < self.n_items) or pid in self.positives[uid]:
                continue
            pp = F32(1.0)
            first = True
            for x in prob:  # reduce(lambda x, y: x * y, prob) over np.float32 values
                pp = F32(x) if first else F32(pp * F32(x))
                first = False
            pred[uid][pid].append((ps[uid][pid], pp, path))
        best = defaultdict(list)
        for u, per_item in pred.items():
            for item, lst in per_item.items():
                best[u].append(sorted(lst, key=lambda x: x[1], reverse=True)[0])
        results = np.full((len(users), self.n_items), -np.inf, F32)
        out = []
        for i, u in enumerate(best):
            sp = sorted(best[u], key=lambda x: (x[0], x[1]), reverse=True)
            top_items = [[p[-1][2], s] for s, _, p in sp][:kmax]
            top_paths = [p for _, _, p in sp][:kmax]
            if len(top_items) < kmax:
                for c in np.argsort(ps[u])[::-1]:
                    if int(c) in self.positives[u]:
                        continue
                    top_items.append([int(c), ps[u][c]])
                    if len(top_items) >= kmax:
                        break
            for (item, score), path in zip(top_items[::-1], top_paths[::-1]):
                results[i, item] = score
                out.append([u, int(item), float(score), path])
        return results, out


def safe_leaves(orc, users, hops, tol=2e-6, with_trace=False):
    """(paths, probs, safe[, trace]): the oracle's beam search plus, per leaf, whether NONE of its hop selections sits
    within `tol` of the value of the first action that was cut off at that hop -- those selections cannot be flipped by
    float32 rounding differences, so a correct device search must emit every safe leaf; a leaf it does not emit must be
    unsafe (the parity rule of tests/test_gpu_pgpr.py and bench_pgpr.py)."""
    paths, probs, trace = orc.beam_search(users, hops)
    # replay to know each leaf's parent row per hop
    safe = []
    row_safe = [True] * len(users)
    for hop, k in enumerate(hops):
        t = trace[hop]
        new_safe = []
        for row in range(len(t["acts"])):
            na = len(t["acts"][row])
            vals = np.sort(t["probs"][row][:na].astype(np.float64))[::-1]
            cut = vals[k] if na > k else None
            for idx, v in zip(t["topk_idx"][row], t["topk_val"][row]):
                if idx >= na:
                    continue
                ok = row_safe[row] and (cut is None or float(v) - cut > tol)
                new_safe.append(ok)
        row_safe = new_safe
    if with_trace:
        return paths, probs, row_safe, trace
    return paths, probs, row_safe
