"""PGPR path reasoning with the reference's constructor / methods / checkpoint names, running the
beam search on the gfx950 kernels of csrc/pgpr.hip.

Reference: /root/reference/hopwise/model/knowledge_aware_recommender/pgpr.py (class PGPR :27-552).

Same: ``PGPR(config, dataset)``; config keys (PGPR.yaml: state_history, max_acts, hidden_sizes,
beam_search_hop, fix_scores_sorting_bug, topk, ...); parameters ``l1, l2, actor, critic`` (state_dict
compatible); ``beam_search`` semantics hop by hop (actions with visited filter and float64 top-max_acts trim,
state = [u, cur, last node, last rel, ...], ELU MLP with dropout, masked softmax, ``probs + mask``, top-k,
children in row order; the state is refreshed only after hops 0 and 1, pgpr.py:481).

Different by design: the graph is one device CSR instead of a dict of dicts, a batch of users expands
as flat row arrays with device-side counts (no host round trip between hops), and results come back as
``int32 [n, 12]`` path records (decode with :meth:`decode_paths` for the reference's tuple format).

Dropout: the reference calls ``F.dropout(.., p=0.5)`` with the default ``training=True`` inside ``forward``
(pgpr.py:180,182), so its inference is stochastic.  ``policy_dropout`` selects: "reference" (default; the
same Bernoulli(1/2) x2 masks drawn from a device Philox stream -- distribution-equal), "identity", or
explicit masks passed to :meth:`beam_search` (what the parity tests use).
"""

from __future__ import annotations

import numpy as np
import torch
from torch import nn

from .. import pgpr_ops as P
from ..utils import InputType
from .abstract_recommender import ExplainableRecommender, KnowledgeRecommender
from .knowledge_graph_embedding_recommender import _cfg, xavier_normal_initialization


class BeamResult:
    """Leaf (or intermediate) beam rows on the device."""

    def __init__(self, paths, probs, n_rows):
        self.paths, self.probs, self._n = paths, probs, n_rows

    @property
    def n(self):
        return int(self._n.item())  # the only synchronisation of a search

    def numpy(self):
        n = self.n
        return self.paths[:n].cpu().numpy(), self.probs[:n].cpu().numpy()


class PGPR(KnowledgeRecommender, ExplainableRecommender):
    input_type = InputType.USERWISE

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        self.user_num = self.n_users
        self.topk = config["topk"]
        self.state_history = int(config["state_history"])
        self.max_acts = int(config["max_acts"])
        self.hidden_sizes = list(config["hidden_sizes"])
        self.act_dim = self.max_acts + 1
        self.beam_search_hop = list(config["beam_search_hop"])
        self.fix_scores_sorting_bug = bool(config["fix_scores_sorting_bug"])
        self.policy_dropout = _cfg(config, "policy_dropout", "reference")
        self.user_chunk = int(_cfg(config, "beam_user_chunk", 64))
        self.ui_relation_id = dataset.field2token_id["relation_id"][dataset.ui_relation]
        dev = torch.device(self.device)

        # graph: the same five arrays ckg_dict_graph() walks (kg_dataset.py:1050-1057)
        inter = dataset.inter_feat
        uids = np.asarray(inter[dataset.uid_field])
        iids = np.asarray(inter[dataset.iid_field])
        self.graph = P.DeviceCKG(self.n_users, self.n_items, self.n_entities, self.ui_relation_id, uids, iids,
                                 np.asarray(dataset.head_entities), np.asarray(dataset.relations),
                                 np.asarray(dataset.tail_entities), dev)
        self.positives = dataset.history_item_matrix()[0]
        self._pos_bits = P.positives_bitset(self.positives, self.n_users, self.n_items, dev)

        # pretrained KGE tables: float64 as the reference keeps them (dataset.py:568-581) + float32 copies
        ue = np.asarray(dataset.get_preload_weight("user_embedding_id"), np.float64)
        ee = np.asarray(dataset.get_preload_weight("entity_embedding_id"), np.float64)
        re_ = np.asarray(dataset.get_preload_weight("relation_embedding_id"), np.float64)
        self.embedding_size = ue.shape[1]
        self._t64 = tuple(torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (ue, ee, re_))
        self._t32 = tuple(t.to(torch.float32).contiguous() for t in self._t64)
        self._score_rel_entity = P.rel_entity_scores(self._t64[2], self._t64[1])  # R[r] . E[n], float64, once
        self.state_dim = (2 + 2 * self.state_history) * self.embedding_size

        self.l1 = nn.Linear(self.state_dim, self.hidden_sizes[0])
        self.l2 = nn.Linear(self.hidden_sizes[0], self.hidden_sizes[1])
        self.actor = nn.Linear(self.hidden_sizes[1], self.act_dim)
        self.critic = nn.Linear(self.hidden_sizes[1], 1)
        self._drop_calls = 0
        self._seed = int(_cfg(config, "seed", 2020))

    # ------------------------------------------------------------------------------------------------
    def _alloc(self, B, hops):
        dev = self._t64[0].device
        rows = [B]
        for k in hops:
            rows.append(rows[-1] * int(k))
        rmax = max(rows[:-1])
        i32 = lambda *s: torch.empty(s, dtype=torch.int32, device=dev)
        f32 = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        buf = dict(
            rows=rows,
            paths=[i32(r, P.ROW_INTS) for r in rows],
            probs=[f32(r, 3) for r in rows],
            n_rows=torch.zeros(len(rows), dtype=torch.int32, device=dev),
            acts=i32(rmax, self.act_dim, 2),
            n_acts=i32(rmax),
            state=f32(rmax, self.state_dim),
            x1=f32(rmax, self.hidden_sizes[0]),
            x2=f32(rmax, self.hidden_sizes[1]),
            logits=f32(rmax, self.act_dim),
            tk_idx=i32(rmax, max(hops)),
            tk_val=f32(rmax, max(hops)),
            offsets=i32(rmax),
            scratch=None,
        )
        if self.graph.max_degree > P.ACT_LDS_KEYS:
            buf["scratch"] = torch.empty((min(2048, rmax), self.graph.max_degree), dtype=torch.int64, device=dev)
        return buf

    def _policy_logits(self, buf, n_rows, rmax, drops, hop):
        """l1 -> ELU -> dropout -> l2 -> ELU -> dropout -> actor (pgpr.py:175-187), logits into buf['logits']."""
        modes = []
        for li in range(2):
            if drops is not None:
                modes.append((drops[hop][li], 1))
            elif self.policy_dropout == "reference":
                modes.append((None, 2))
            else:
                modes.append((None, 0))
        self._drop_calls += 1
        P.mlp_layer(buf["state"], self.l1.weight, self.l1.bias, buf["x1"], n_rows, rmax, True, modes[0][0], modes[0][1],
                    self._seed, 2 * self._drop_calls)
        P.mlp_layer(buf["x1"], self.l2.weight, self.l2.bias, buf["x2"], n_rows, rmax, True, modes[1][0], modes[1][1],
                    self._seed, 2 * self._drop_calls + 1)
        P.mlp_layer(buf["x2"], self.actor.weight, self.actor.bias, buf["logits"], n_rows, rmax, False)

    @torch.no_grad()
    def beam_search_device(self, users, hops=None, drops=None, keep_trace=False):
        """Expand ``users`` (int64 tensor on the device) hop by hop; returns a :class:`BeamResult` of leaf rows.
        ``drops``: optional per hop (mask1 [rows, h0], mask2 [rows, h1]) multiplicative masks."""
        hops = list(self.beam_search_hop if hops is None else hops)
        if not 1 <= len(hops) <= 3:
            raise ValueError("beam_search_hop must have 1..3 entries (paths hold at most 4 nodes)")
        if max(hops) > self.act_dim:
            raise RuntimeError("selected index k out of range")  # what torch.topk raises in the reference
        users = users.to(self._t64[0].device, torch.int64).contiguous()
        B = users.numel()
        buf = self._alloc(B, hops)
        g = self.graph
        U64, E64, R64 = self._t64
        U32, E32, R32 = self._t32
        nr = buf["n_rows"]
        P.init_paths(users, buf["paths"][0], buf["probs"][0], nr[0:1])
        score_user, score_entity = P.user_scores(g, U64, E64, R64, users)  # float64 trim tables of this batch
        trace = []
        for hop, k in enumerate(hops):
            rmax = buf["rows"][hop]
            paths, probs, n_rows = buf["paths"][hop], buf["probs"][hop], nr[hop : hop + 1]
            if hop <= 2:
                # the reference refreshes state_pool at the end of hops 0 and 1 only (pgpr.py:481); with at most
                # three hops every hop sees a fresh state, which is the same thing.
                P.state(g, U32, E32, R32, self.state_history, paths, n_rows, rmax, buf["state"])
            P.actions(g, score_user, score_entity, self._score_rel_entity, paths, n_rows, rmax, self.max_acts,
                      self.fix_scores_sorting_bug, buf["acts"], buf["n_acts"], buf["scratch"])
            self._policy_logits(buf, n_rows, rmax, drops, hop)
            P.policy_topk(buf["logits"], self.act_dim, buf["n_acts"], n_rows, rmax, k, buf["tk_idx"], buf["tk_val"])
            P.expand(g, paths, probs, buf["acts"], self.act_dim, buf["n_acts"], n_rows, rmax, buf["tk_idx"],
                     buf["tk_val"], k, buf["offsets"], buf["paths"][hop + 1], buf["probs"][hop + 1],
                     nr[hop + 1 : hop + 2], buf["rows"][hop + 1])
            if keep_trace:
                n = int(n_rows.item())
                trace.append(dict(acts=buf["acts"][:n].clone(), n_acts=buf["n_acts"][:n].clone(),
                                  state=buf["state"][:n].clone(), logits=buf["logits"][:n].clone(),
                                  tk_idx=buf["tk_idx"][:n, :k].clone(), tk_val=buf["tk_val"][:n, :k].clone()))
        res = BeamResult(buf["paths"][-1], buf["probs"][-1], nr[len(hops) : len(hops) + 1])
        res.n_rows_per_hop = nr
        res.trace = trace
        return res

    # -- reference-format helpers ----------------------------------------------------------------------
    @staticmethod
    def decode_paths(paths_np, probs_np):
        """int32 records -> the reference's [(rel, type, id), ...] lists ('self_loop' / 'user' / 'entity')."""
        tname = ("user", "entity")
        out_p, out_q = [], []
        for rec, pr in zip(paths_np, probs_np):
            n = int(rec[1])
            path = [("self_loop", "user", int(rec[0]))]
            for j in range(1, n):
                r, t, i = (int(x) for x in rec[2 + 3 * (j - 1) : 5 + 3 * (j - 1)])
                path.append(("self_loop" if r == P.SELF_LOOP else r, tname[t], i))
            out_p.append(path)
            out_q.append([np.float32(x) for x in pr[: n - 1]])
        return out_p, out_q

    def beam_search(self, users):
        """Reference signature (pgpr.py:440): users tensor -> (path_pool, probs_pool) Python lists."""
        out_p, out_q = [], []
        users = torch.as_tensor(users)
        for s in range(0, users.numel(), self.user_chunk):
            res = self.beam_search_device(users[s : s + self.user_chunk])
            p, q = self.decode_paths(*res.numpy())
            out_p += p
            out_q += q
        return out_p, out_q

    # -- A14 -------------------------------------------------------------------------------------------
    @torch.no_grad()
    def collect_scores_device(self, users, res, n_nodes=None):
        """scores [len(users), n_items] on the device + the selection arrays (pgpr.py:486-552)."""
        users = users.to(self._t64[0].device, torch.int64).contiguous()
        n_nodes = len(self.beam_search_hop) + 1 if n_nodes is None else n_nodes
        return P.collect(self.graph, *self._t64, users, self._pos_bits, res.paths, res.probs, res._n,
                         res.paths.shape[0], n_nodes, max(self.topk))

    def explanations(self, users, res, sel):
        """The reference's [user, item, score, path] list, in the order collect_scores appends them."""
        cnt, item, score, leaf = (sel[k].cpu().numpy() for k in ("count", "item", "score", "leaf"))
        ulist = users.cpu().numpy()
        need = np.unique(leaf[leaf >= 0])
        rows = res.paths[torch.from_numpy(need).to(res.paths.device).long()].cpu().numpy() if len(need) else np.zeros((0, 12))
        prs = np.zeros((len(need), 3), np.float32)
        dec, _ = self.decode_paths(rows, prs)
        by_leaf = dict(zip(need.tolist(), dec))
        out = []
        for slot in range(len(ulist)):
            for j in range(int(cnt[slot])):
                out.append([int(ulist[slot]), int(item[slot, j]), float(score[slot, j]), by_leaf[int(leaf[slot, j])]])
        return out

    def _search_and_collect(self, users):
        users = torch.as_tensor(users)
        outs, expl = [], []
        for s in range(0, users.numel(), self.user_chunk):
            chunk = users[s : s + self.user_chunk]
            res = self.beam_search_device(chunk)
            scores, sel = self.collect_scores_device(chunk, res)
            outs.append((chunk, res, scores, sel))
        return outs

    def full_sort_predict(self, interaction):  # pgpr.py:408-415
        return torch.cat([o[2] for o in self._search_and_collect(interaction[self.USER_ID])], dim=0)

    def explain(self, interaction):  # pgpr.py:417-435
        outs = self._search_and_collect(interaction[self.USER_ID])
        scores = torch.cat([o[2] for o in outs], dim=0)
        expl = []
        for chunk, res, _, sel in outs:
            expl += self.explanations(chunk, res, sel)
        for e in expl:
            e[-1] = self.decode_path(e[-1])
        return scores, expl

    def predict(self, interaction):
        return

    def decode_path(self, path):
        return path
