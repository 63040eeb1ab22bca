"""numpy restatement of CAFE inference (TEST INFRASTRUCTURE ONLY; see oracle/__init__.py).

Reference: /root/reference/hopwise/model/knowledge_aware_recommender/cafe.py
  DeepRelationModule.forward :686-699, RelationModule.forward :656-667, SymbolicNetwork._forward :514-524,
  infer_with_path :587-642, KGMask.get_ids :734-750, MetaProgramExecutor.execute :790-857, collect_results :859-867,
  CAFE.run_program :406-447, NeuralProgramLayout :873-888.
Pinned by tests/golden/cafe_*.npz (reference run on its own tests/test_data fixture, eval mode, child permutation
patched to the identity).  Ties between equal log-probabilities follow (value desc, id asc), the build's convention.
"""

from __future__ import annotations

import numpy as np

from .pgpr import CKG

F32 = np.float32


def _bn(x, p, pre, eps=1e-5):
    return ((x - p[pre + ".running_mean"]) / np.sqrt(p[pre + ".running_var"] + eps) * p[pre + ".weight"] + p[pre + ".bias"]).astype(F32)


def module_forward(p, rel, eh_vec, user_vec, deep=True):
    """Eval-mode forward of the module of relation ``rel`` (dropout off, BatchNorm running statistics)."""
    pre = f"model.{rel}"
    feat = np.concatenate([eh_vec, user_vec], axis=-1).astype(F32)
    if deep:
        x = np.maximum(feat @ p[pre + ".fc1.weight"].T + p[pre + ".fc1.bias"], 0).astype(F32)
        x = _bn(x, p, pre + ".bn1")
        x = np.maximum(x @ p[pre + ".fc2.weight"].T + p[pre + ".fc2.bias"] + feat, 0).astype(F32)
        x = _bn(x, p, pre + ".bn2")
        return (x @ p[pre + ".fc3.weight"].T + p[pre + ".fc3.bias"]).astype(F32)
    x = np.maximum(feat @ p[pre + ".fc1.weight"].T + p[pre + ".fc1.bias"], 0).astype(F32)
    x = _bn(x, p, pre + ".bn1")
    return (x @ p[pre + ".fc2.weight"].T + p[pre + ".fc2.bias"] + eh_vec).astype(F32)


def _bn_train(x, p, pre, eps=1e-5):
    """BatchNorm1d in training mode: batch mean / biased variance."""
    mu, var = x.mean(0, dtype=np.float64), x.var(0, dtype=np.float64)
    return (((x - mu) / np.sqrt(var + eps)) * p[pre + ".weight"] + p[pre + ".bias"]).astype(F32)


def module_forward_train(p, rel, eh_vec, user_vec, masks, deep=True):
    """Train-mode forward (cafe.py:656-667, :686-699): ``masks`` = the multiplicative dropout masks in call order."""
    pre = f"model.{rel}"
    feat = np.concatenate([eh_vec, user_vec], axis=-1).astype(F32)
    x = np.maximum(feat @ p[pre + ".fc1.weight"].T + p[pre + ".fc1.bias"], 0).astype(F32)
    x = _bn_train(x * masks[0], p, pre + ".bn1")
    if not deep:
        return (x @ p[pre + ".fc2.weight"].T + p[pre + ".fc2.bias"] + eh_vec).astype(F32)
    x = np.maximum(x @ p[pre + ".fc2.weight"].T + p[pre + ".fc2.bias"] + feat, 0).astype(F32)
    x = _bn_train(x * masks[1], p, pre + ".bn2")
    return (x @ p[pre + ".fc3.weight"].T + p[pre + ".fc3.bias"]).astype(F32)


def symbolic_forward_loss(p, metapath, pos_paths, neg_pids, tables, masks, deep=True):
    """(reg_loss, rank_loss) of SymbolicNetwork.forward (cafe.py:526-555); tables = {"user": U, "entity": E}."""
    user_vec = tables["user"][pos_paths[:, 0]]
    x, reg, rank = user_vec, 0.0, None
    per = 2 if deep else 1
    for i, (rel, et) in enumerate(metapath[1:]):
        x = module_forward_train(p, rel, x, user_vec, masks[per * i : per * (i + 1)], deep)
        scores = (x @ tables[et].T).astype(np.float64)
        lse = np.log(np.exp(scores - scores.max(1, keepdims=True)).sum(1)) + scores.max(1)
        rows = np.arange(len(scores))
        reg += float(np.mean(lse - scores[rows, pos_paths[:, i + 1]]))
        if i == len(metapath) - 2:
            z = scores[rows, neg_pids] - scores[rows, pos_paths[:, -1]]
            rank = float(np.mean(1.0 / (1.0 + np.exp(-z))))
    return reg, rank


def log_softmax(x):
    x = x.astype(F32)
    m = x.max()
    return ((x - m) - np.log(np.sum(np.exp(x - m), dtype=F32))).astype(F32)


class CafeOracle:
    def __init__(self, graph: CKG, params, user_emb, entity_emb, relation2rid, ui_relation_name, n_items, positives,
                 metapaths, deep=True):
        self.g, self.p = graph, params
        self.U, self.E = np.asarray(user_emb, F32), np.asarray(entity_emb, F32)
        self.rid = relation2rid
        self.ui = ui_relation_name
        self.n_items = int(n_items)
        self.positives = np.asarray(positives)
        self.metapaths = metapaths
        self.deep = deep

    def eh_type(self, rel):
        return 0 if rel == self.ui else 1  # relation_info: the UI relation goes user -> entity (cafe.py:96-100)

    def valid_ids(self, rel, last):
        """KGMask.get_ids(eh_name, last, rid): the lookup type is the relation's head type."""
        t = self.eh_type(rel)
        n = self.g.n_users if t == 0 else self.g.n_entities
        if not 0 <= last < n:
            return set()
        rels, dsts = self.g.neighbours(t, last)
        return set(int(x) for x in dsts[rels == self.rid[rel]])

    def node_logprobs(self, uid, chain):
        """log-prob vectors of every prefix of the relation chain for this user (SymbolicNetwork._forward)."""
        uv = self.U[uid][None, :]
        vec, out = uv, []
        for rel in chain:
            vec = module_forward(self.p, rel, vec, uv, self.deep)
            out.append(log_softmax((vec @ self.E.T)[0]))
        return out

    @staticmethod
    def _topk(ids, lp, k):
        ids = sorted(ids)
        order = sorted(ids, key=lambda i: (-float(lp[i]), i))[:k]
        return order

    def infer_with_path(self, mp, uid, excluded, topk_paths):
        chain = [r for r, _ in mp[1:]]
        lps = self.node_logprobs(uid, chain)
        if len(self.valid_ids(chain[0], uid)) <= 0:
            return []
        sizes = [topk_paths, 5, 1]
        result = [([uid], [])]
        for i, rel in enumerate(chain):
            tmp, visited = [], set()
            for path, value in result:
                valid = self.valid_ids(rel, path[-1]) - visited
                # cafe.py:606-607 subtracts `excluded_pids`, but that is a torch tensor there: set.difference() walks
                # its 0-dim tensor elements, which hash by identity, so NOTHING is removed.  Reproduced: no exclusion.
                if not valid:
                    continue
                for nid in self._topk(valid, lps[i], min(sizes[i], len(valid))):
                    tmp.append((path + [nid], value + [lps[i][nid]]))
                    visited.add(nid)
            if not tmp:
                return []
            result = tmp
        return result

    # ---- executor over the layout tree, children visited in dict order (identity permutation) -----------------
    def layout(self):
        tree = {}  # chain tuple -> list of child chains, insertion ordered
        for mp in self.metapaths:
            chain = ()
            for rel, _ in mp[1:]:
                child = chain + (rel,)
                tree.setdefault(chain, [])
                if child not in tree[chain]:
                    tree[chain].append(child)
                tree.setdefault(child, [])
                chain = child
        return tree

    def execute(self, uid, excluded, manual_topk=5):
        tree = self.layout()
        excluded = set(int(x) for x in excluded)
        data = {(): [([uid], [])]}
        lp_cache = {}
        queue = list(tree[()])
        while queue:
            chain = queue.pop(0)
            queue.extend(tree[chain])
            lps = lp_cache.get(chain)
            if lps is None:
                lps = self.node_logprobs(uid, list(chain))[-1]
                lp_cache[chain] = lps
            leaf = len(tree[chain]) == 0
            out, visited = [], set()
            for path, value in data[chain[:-1]]:
                valid = self.valid_ids(chain[-1], path[-1]) - visited
                if leaf and excluded:
                    valid = valid - excluded
                if not valid:
                    continue
                for nid in self._topk(valid, lps, min(manual_topk, len(valid))):
                    out.append((path + [nid], value + [lps[nid]]))
                    visited.add(nid)
                    if leaf:
                        excluded.add(nid)
            data[chain] = out
        # collect_results: leaves in BFS order
        res, queue = [], list(tree[()])
        while queue:
            chain = queue.pop(0)
            queue.extend(tree[chain])
            if not tree[chain]:
                res.extend((p, v, chain) for p, v in data[chain])
        return res

    def run_program(self, users):
        results = np.full((len(users), self.n_items), -np.inf, F32)
        collected = []
        for i, u in enumerate(users):
            u = int(u)
            pos = self.positives[u]
            paths = self.execute(u, pos[pos != 0])
            tmp, inst = [], {}
            for ids, vals, chain in paths:
                sc = F32(vals[0])
                for x in vals[1:]:
                    sc = F32(sc * F32(x))
                tmp.append((ids[-1], sc))
                inst[ids[-1]] = [("self_loop", "user", u)] + [(chain[j], "entity", ids[j + 1]) for j in range(len(chain))]
            for item, sc in sorted(tmp, key=lambda x: x[1], reverse=True):
                if item < self.n_items and results[i, item] < sc:
                    results[i, item] = sc
                    collected.append([u, item, sc, inst[item]])
        return results, collected


def collect_records(leaf_entries, n_items):
    """collect_results + the ranking loop of run_program for ONE user (cafe.py:417-441, :859-867), array-wise.

    ``leaf_entries``: per leaf, in collect_results order, (ids [c, depth] int, vals [c, depth] float32).  The reference
    appends every path to ``tmp`` in leaf order with score = reduce(mul) over its np.float32 values, ``inst[item]`` keeps the
    LAST path that reached an item, ``tmp`` is sorted by score (stable, descending) and an entry is emitted when it beats
    results[item] -- i.e. per item the first entry in sorted order, carrying inst[item]'s path.
    Returns (results [n_items] float32 with -inf holes, records [(item, score, leaf ordinal, row)] in emission order)."""
    items, scores, blk, row = [], [], [], []
    for l, (ids, vals) in enumerate(leaf_entries):
        c = len(ids)
        if c == 0:
            continue
        sc = vals[:, 0].astype(np.float32)
        for d in range(1, vals.shape[1]):
            sc = sc * vals[:, d].astype(np.float32)
        items.append(ids[:, -1].astype(np.int64))
        scores.append(sc)
        blk.append(np.full(c, l, np.int64))
        row.append(np.arange(c, dtype=np.int64))
    results = np.full(n_items, -np.inf, np.float32)
    if not items:
        return results, []
    items, scores, blk, row = (np.concatenate(x) for x in (items, scores, blk, row))
    n = len(items)
    last_of = {}
    for e in range(n):
        last_of[int(items[e])] = e
    srt = np.argsort(-scores, kind="stable")
    srt = srt[(scores[srt] > -np.inf) & (items[srt] < n_items) & (items[srt] >= 0)]
    seen, out = set(), []
    for k in srt.tolist():
        it = int(items[k])
        if it in seen:
            continue
        seen.add(it)
        results[it] = scores[k]
        src = last_of[it]
        out.append((it, scores[k], int(blk[src]), int(row[src])))
    return results, out
