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

import os

import numpy as np
import torch
from torch import nn

from .. import ops
from .. import pgpr_ops as P
from ..lazy import LazyList
from ..philox import PhiloxCounter
from ..stream_cache import FencedValue
from ..stream_pool import concurrent_streams
from ..utils import InputType, seed_for_rank
from .abstract_recommender import ExplainableRecommender, KnowledgeRecommender
from .knowledge_graph_embedding_recommender import _cfg


_ACTION_STREAMS = {}  # (device index, caller stream handle) -> the stream a search's action kernels run on beside its policy launch


class BeamResult:
    """Leaf (or intermediate) beam rows on the device.

    Two layouts.  Full: ``paths`` int32 [rows, 12] records + ``probs`` float32 [rows, 3].  Compact (the default output of the
    last hop, ``leaves`` is not None): ``leaves`` int32 [rows, 4] = {parent row, relation, node, bits of the hop's value} next
    to the parent rows in ``parent_paths`` / ``parent_probs`` -- the 60 bytes of a full record repeat the parent's prefix once
    per child, 64 times over on a 64-wide beam.  ``paths`` / ``probs`` then materialise on first use
    (hw_pgpr_materialize_leaves); ``rows(idx)`` rebuilds only the asked records; the collector reads the compact form."""

    def __init__(self, paths, probs, n_rows, graph=None, leaves=None, child_offsets=None, n_parents=None):
        self._paths, self._probs, self._n = paths, probs, n_rows
        self.graph, self.leaves = graph, leaves
        # compact leaves: first leaf of every parent row + the device count of parent rows (the collector walks parents)
        self.child_offsets, self.n_parents = child_offsets, n_parents
        self.parent_paths, self.parent_probs = (paths, probs) if leaves is not None else (None, None)
        if leaves is not None:
            self._paths = self._probs = None

    @property
    def n(self):
        return int(self._n.item())  # the only synchronisation of a search

    def record_stream(self, stream):
        """The result was produced on another stream than the one that will read it: tell torch's allocator."""
        for t in (self._paths, self._probs, self._n, self.leaves, self.parent_paths, self.parent_probs,
                  self.child_offsets, self.n_parents, getattr(self, "n_rows_per_hop", None)):
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(stream)

    @property
    def capacity(self):
        return (self.leaves if self.leaves is not None else self._paths).shape[0]

    def _materialize(self):
        if self._paths is None:
            self._paths, self._probs = P.materialize_leaves(self.graph, self.parent_paths, self.parent_probs, self.leaves,
                                                            self._n)

    @property
    def paths(self):
        self._materialize()
        return self._paths

    @property
    def probs(self):
        self._materialize()
        return self._probs

    def rows(self, idx):
        """(paths, probs) of the leaf rows ``idx`` (int tensor / array) as device tensors."""
        idx = torch.as_tensor(idx)
        if self.leaves is not None and self._paths is None:
            return P.materialize_leaves(self.graph, self.parent_paths, self.parent_probs, self.leaves, self._n,
                                        sel=idx.to(self.leaves.device))
        idx = idx.to(self._paths.device).long()
        return self._paths[idx], self._probs[idx]

    def numpy(self):
        n = self.n
        if self.leaves is not None and self._paths is None:
            p, q = P.materialize_leaves(self.graph, self.parent_paths, self.parent_probs, self.leaves, self._n, n_out=n)
            return p.cpu().numpy(), q.cpu().numpy()
        return self._paths[:n].cpu().numpy(), self._probs[:n].cpu().numpy()


class _EpisodeLoss(torch.autograd.Function):
    """loss = PGPR.update() of one device episode; backward = the hand-written gradient pass."""

    @staticmethod
    def forward(ctx, model, users, *params):
        ep = model.episode_device(users)
        ctx.model, ctx.ep = model, ep
        model.last_loss_parts = ep["loss"]
        return ep["loss"][0].clone()

    @staticmethod
    def backward(ctx, grad_out):
        # the upstream gradient of the scalar loss enters at d loss / d head: every gradient behind it is linear in it
        grads = ctx.model._episode_backward(ctx.ep, upstream=grad_out.to(torch.float32).reshape(1))
        ctx.ep = None
        return (None, None) + tuple(grads)


class PGPR(KnowledgeRecommender, ExplainableRecommender):
    input_type = InputType.USERWISE
    other_parameter_name = ["_drop_calls"]  # trainer.py:296-304 saves other_parameter(); resume continues the stream

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
        self.l1_lookup = bool(_cfg(config, "l1_lookup", True))  # beam search: first policy layer from slot tables
        # beam search: the whole policy network of a hop in one launch when the shapes allow (HW_PGPR_FUSED=0: three launches)
        self.fused_policy = bool(_cfg(config, "fused_policy", os.environ.get("HW_PGPR_FUSED", "1") != "0"))
        # beam search: softmax + top-k + expand of a hop in one launch, the last hop's children as 16-byte leaves
        # (HW_PGPR_TAIL=0: top-k and expand as two launches, full 60-byte records everywhere)
        self.fused_tail = bool(_cfg(config, "fused_topk_expand", os.environ.get("HW_PGPR_TAIL", "1") != "0"))
        self.compact_leaves = bool(_cfg(config, "compact_leaves", self.fused_tail))
        self.acts_u16 = bool(_cfg(config, "acts_u16", os.environ.get("HW_PGPR_ACTS16", "1") != "0"))
        # users per search (config beam_user_chunk; 0 = sized from the beam).  The first hops of a search are launch-bound, so
        # few users per search waste the chip (64/64/64: 256 users per search expand 2x the paths per second of 64;
        # hopwise's default 25/5/1 with its 4 096-user evaluation batches: 0.51 ms per 256 users as ONE search against 1.18 in
        # chunks of 256, tools/bench_pgpr_eval.py); many users cost memory: the rows of the widest expanded hop, ~3 KB each
        self.user_chunk = int(_cfg(config, "beam_user_chunk", 0))
        # evaluation batches larger than user_chunk: consecutive searches are dealt over this many HIP streams (a search's
        # policy launch is matrix-bound, its action / ranking launches are not: +15 % on the bench); 1 = one after another
        self.search_streams = max(1, int(_cfg(config, "search_streams", os.environ.get("HW_PGPR_STREAMS", "3"))))
        # beam search: a hop's action kernels on a stream of their own beside its policy launch (HW_PGPR_OVERLAP=0: in line)
        self.score_cache_mb = int(_cfg(config, "score_cache_mb", os.environ.get("HW_PGPR_SCORE_CACHE_MB", "2048")))
        self.overlap_actions = bool(_cfg(config, "overlap_actions", os.environ.get("HW_PGPR_OVERLAP", "1") != "0"))
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
        # Philox offsets of the policy dropout / action draws: four per call (layer 1, layer 2, action dropout, action draw);
        # the call count rides in the checkpoint as `_drop_calls` (other_parameter); philox.py for the captured-step form
        self._philox = PhiloxCounter(stride=4)
        self._seed = seed_for_rank(_cfg(config, "seed", 2020))

        # training (pgpr.py:40-46, 90-104)
        self.gamma = float(_cfg(config, "gamma", 0.99))
        self.action_dropout = float(_cfg(config, "action_dropout", 0.0))
        self.weight_factor = float(_cfg(config, "weight_factor", 1e-3))
        self.max_path_len = int(_cfg(config, "max_path_len", 3))
        if not 1 <= self.max_path_len <= 3:
            raise ValueError("max_path_len must be 1..3 (a path record holds at most 4 nodes)")
        rel2id = dataset.field2token_id["relation_id"]
        pats = []
        for constraint in _cfg(config, "path_constraint", None) or []:
            names = [node[0] for node in constraint if node[0] is not None]
            names = [n[:-2] if n.endswith("_r") else n for n in names]  # the reference strips "_r" on the pattern side only
            if len(names) <= 3:
                pats.append([1 + len(names)] + [int(rel2id.get(n, -2)) for n in names] + [-2] * (3 - len(names)))
        self._patterns = torch.tensor(pats, dtype=torch.int32, device=dev).view(-1, 4) if pats else None
        self.last_loss_parts = None

    def capture_error_flags(self, device):
        """Device flags a captured training step may raise (driver.GraphedTrainStep allocates them before the capture and
        the epoch loops read them once per epoch): {name: (bool tensor [1], message)}."""
        if self.__dict__.get("_pad_user_flag") is None or self._pad_user_flag.device != torch.device(device):
            self.__dict__["_pad_user_flag"] = torch.zeros(1, dtype=torch.bool, device=device)
        return {"pad_user": (self._pad_user_flag, "a batch of a captured PGPR step held the [PAD] user 0, which the eager "
                                                   "step filters out (pgpr.py:189-203): its episode went into the loss")}

    @property
    def _drop_calls(self):
        return self._philox.calls

    @_drop_calls.setter
    def _drop_calls(self, v):
        self._philox.calls = int(v)

    # ------------------------------------------------------------------------------------------------
    def _alloc(self, B, hops, compact_leaves=False, acts_u16=False, mlp_buffers=True, topk_buffers=True):
        """Work buffers of one search.  ``mlp_buffers`` / ``topk_buffers`` = False leave out what only the three-launch policy
        network (state, x1, x2: 4.8 GB at 1 M beam rows) and the two-launch tail (top-k indices / values) write."""
        dev = self._t64[0].device
        rows = [B]
        for k in hops:
            rows.append(rows[-1] * int(k))
        rmax = max(rows[:-1])
        i32 = lambda *s: torch.empty(s, dtype=torch.int32, device=dev)
        f32 = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        full = rows[:-1] if compact_leaves else rows  # the last level is held as 16-byte leaves instead of 60-byte records
        buf = dict(
            rows=rows,
            paths=[i32(r, P.ROW_INTS) for r in full] + ([None] if compact_leaves else []),
            probs=[f32(r, 3) for r in full] + ([None] if compact_leaves else []),
            leaves=i32(rows[-1], 4) if compact_leaves else None,
            n_rows=torch.zeros(len(rows), dtype=torch.int32, device=dev),
            acts=(torch.empty((rmax, P.act16_ld(self.act_dim)), dtype=torch.int16, device=dev) if acts_u16
                  else i32(rmax, self.act_dim, 2)),
            n_acts=i32(rmax),
            state=f32(rmax, self.state_dim) if mlp_buffers else None,
            x1=f32(rmax, self.hidden_sizes[0]) if mlp_buffers else None,
            x2=f32(rmax, self.hidden_sizes[1]) if mlp_buffers else None,
            # rows padded to 256 floats: the fused policy kernel stores whole 16-byte pieces; the top-k reads [:, :act_dim]
            logits=f32(rmax, max(256, self.act_dim))[:, : self.act_dim],
            tk_idx=i32(rmax, max(hops)) if topk_buffers else None,
            tk_val=f32(rmax, max(hops)) if topk_buffers else None,
            offsets=i32(rmax),
            scratch=None,
        )
        if self.graph.max_degree > P.ACT_LDS_KEYS:
            buf["scratch"] = torch.empty((min(2048, rmax), self.graph.max_degree), dtype=torch.int64, device=dev)
        return buf

    def _fenced(self, name):
        fv = self.__dict__.get(name)
        if fv is None:
            fv = self.__dict__[name] = FencedValue()
        return fv

    def _l1_slot_tables(self):
        """Slot tables of the first layer for the current l1 weights (rebuilt when the parameter version changes).  Fenced:
        searches on several streams read them (stream_cache.py)."""
        key = (self.l1.weight._version, self.l1.weight.data_ptr())
        return self._fenced("_l1_cache").get(
            key, lambda: P.layer1_tables(self._t32, self.l1.weight.detach(), self.state_history))

    def _fused_policy_state(self):
        """(weight planes, zero row, padded actor bias) of hw_pgpr_policy_fused for the current l2 / actor parameters."""
        key = tuple((t._version, t.data_ptr()) for t in (self.l2.weight, self.actor.weight, self.actor.bias))

        def build():
            dev = self.l2.weight.device
            ba = torch.zeros(256, dtype=torch.float32, device=dev)
            ba[: self.act_dim] = self.actor.bias.detach()
            return (P.policy_prepare(self.l2.weight, self.actor.weight),
                    torch.zeros(self.hidden_sizes[0], dtype=torch.float32, device=dev), ba)

        return self._fenced("_fused_cache").get(key, build)

    def _warm_search_caches(self):
        """Build, on the CURRENT stream, everything a search builds lazily: called before searches are dealt over side
        streams, so that no side stream is the first (and only ordered) writer of a tensor the others read."""
        if self.fused_policy and P.policy_fused_supported(self.hidden_sizes, self.act_dim):
            self._fused_policy_state()
        self._l1_slot_tables()
        ops._OobCounter.get(self._t64[0].device)
        self._batch_scores(torch.zeros(1, dtype=torch.int64, device=self._t64[0].device))  # the all-users score cache

    def _policy_logits(self, buf, n_rows, rmax, drops, hop, paths=None):
        """l1 -> ELU -> dropout -> l2 -> ELU -> dropout -> actor (pgpr.py:175-187), logits into buf['logits'].
        With ``paths`` the first layer is the table-lookup form (no state tensor, no first GEMM)."""
        modes = []
        for li in range(2):
            if drops is not None:
                modes.append((drops[hop][li], 1))
            elif self.policy_dropout == "reference":
                modes.append((None, 2))
            else:
                modes.append((None, 0))
        call = self._philox.next()
        off1, off2 = self._philox.offset(call, 0), self._philox.offset(call, 1)
        if paths is not None and self.fused_policy and P.policy_fused_supported(self.hidden_sizes, self.act_dim):
            # lookup -> l2 -> actor in one launch (csrc/pgpr_policy.hip); its Philox dropout stream packs 128 keep bits per
            # block (drop_mode 3), the three-launch form below keeps hw_common.h:philox_bits' stream (drop_mode 2)
            planes, zero_row, ba_pad = self._fused_policy_state()
            P.policy_fused(self._l1_slot_tables(), zero_row, self.l1.bias.detach(), planes, self.l2.bias.detach(), ba_pad,
                           self.act_dim, self.state_history, paths, n_rows, rmax, buf["logits"],
                           drop_mode={0: 0, 1: 1, 2: 3}[modes[0][1]], mask1=modes[0][0], mask2=modes[1][0],
                           seed=self._seed, offset1=off1, offset2=off2)
            return
        if paths is not None:
            P.layer1_lookup(self._l1_slot_tables(), self.l1.bias.detach(), self.state_history, paths, n_rows, rmax,
                            buf["x1"], modes[0][0], modes[0][1], self._seed, off1)
        else:
            P.mlp_layer(buf["state"], self.l1.weight, self.l1.bias, buf["x1"], n_rows, rmax, True, modes[0][0],
                        modes[0][1], self._seed, off1)
        P.mlp_layer(buf["x1"], self.l2.weight, self.l2.bias, buf["x2"], n_rows, rmax, True, modes[1][0], modes[1][1],
                    self._seed, off2)
        P.mlp_layer(buf["x2"], self.actor.weight, self.actor.bias, buf["logits"], n_rows, rmax, False)

    @torch.no_grad()
    def beam_search_device(self, users, hops=None, drops=None, keep_trace=False, keep_buffers=False, stage_events=None,
                           fused_tail=None, compact_leaves=None, overlap_actions=None):
        """Expand ``users`` (int64 tensor on the device) hop by hop; returns a :class:`BeamResult` of leaf rows.
        ``drops``: optional per hop (mask1 [rows, h0], mask2 [rows, h1]) multiplicative masks.
        ``fused_tail`` (default: config ``fused_topk_expand`` / HW_PGPR_TAIL, on): softmax + top-k + expand of a hop as one
        launch where the shapes allow (act_dim <= 256, k <= 64); ``compact_leaves`` (default: with the fused tail, unless a
        trace is kept): the last hop writes 16-byte leaves instead of full records (see :class:`BeamResult`).
        ``stage_events``: a list that receives (stage, hop, HIP event) marks on the launch stream after every stage of
        every hop (bench_pgpr's per-stage times; nothing is recorded when None)."""
        hops = list(self.beam_search_hop if hops is None else hops)
        if not 1 <= len(hops) <= 3:
            raise ValueError("beam_search_hop must have 1..3 entries (paths hold at most 4 nodes)")
        if max(hops) > self.act_dim:
            raise RuntimeError("selected index k out of range")  # what torch.topk raises in the reference
        users = users.to(self._t64[0].device, torch.int64).contiguous()
        B = users.numel()
        if B == 0:  # nothing to expand: an empty result without a launch (the entry points refuse null buffers)
            dev = users.device
            res = BeamResult(torch.empty((0, P.ROW_INTS), dtype=torch.int32, device=dev),
                             torch.empty((0, 3), dtype=torch.float32, device=dev),
                             torch.zeros(1, dtype=torch.int32, device=dev), graph=self.graph)
            res.n_rows_per_hop = torch.zeros(len(hops) + 1, dtype=torch.int32, device=dev)
            res.trace, res.buffers = [], None
            return res
        if fused_tail is None:
            fused_tail = self.fused_tail
        fused_tail = bool(fused_tail) and not keep_trace
        if compact_leaves is None:
            compact_leaves = self.compact_leaves
        compact_leaves = bool(compact_leaves) and fused_tail and P.topk_expand_supported(self.act_dim, hops[-1])
        # action lists as 2-byte edge offsets when every hop's tail is the one-launch form (the only reader of that layout)
        acts_u16 = (fused_tail and self.acts_u16 and all(P.topk_expand_supported(self.act_dim, k) for k in hops)
                    and P.acts_u16_supported(self.graph, self.fix_scores_sorting_bug))
        lookup = self.l1_lookup and self.hidden_sizes[0] % 4 == 0
        one_launch_policy = (lookup and self.fused_policy and not keep_trace
                             and P.policy_fused_supported(self.hidden_sizes, self.act_dim))
        all_fused_tails = fused_tail and all(P.topk_expand_supported(self.act_dim, k) for k in hops)
        buf = self._alloc(B, hops, compact_leaves, acts_u16, mlp_buffers=not one_launch_policy,
                          topk_buffers=not all_fused_tails)
        g = self.graph
        U64, E64, R64 = self._t64
        U32, E32, R32 = self._t32
        nr = buf["n_rows"]
        P.init_paths(users, buf["paths"][0], buf["probs"][0], nr[0:1])
        score_user, score_entity = self._batch_scores(users)  # float64 trim tables
        trace = []

        def mark(stage, hop):
            if stage_events is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                stage_events.append((stage, hop, ev))

        # stage marks time the stages one after another; a trace reads buffers between them
        # (``overlap_actions``: a caller that already runs several searches side by side passes False -- their kernels fill
        # each other's gaps, and the extra stream only adds to the queue: 4.97 -> 4.64 ms for a lone search, 4.20 = 4.31 for three)
        overlap = ((self.overlap_actions if overlap_actions is None else bool(overlap_actions)) and stage_events is None
                   and not keep_trace and users.is_cuda and not torch.cuda.is_current_stream_capturing())
        if overlap:
            dev = users.device
            cur = torch.cuda.current_stream(dev)
            side = _ACTION_STREAMS.get((dev.index, cur.cuda_stream))
            if side is None:  # a stream on another hardware queue than the caller's (stream_pool: probed once)
                side = _ACTION_STREAMS[(dev.index, cur.cuda_stream)] = concurrent_streams(1, dev, beside=[cur])[0]
        mark("start", -1)
        for hop, k in enumerate(hops):
            rmax = buf["rows"][hop]
            paths, probs, n_rows = buf["paths"][hop], buf["probs"][hop], nr[hop : hop + 1]
            if hop <= 2 and (keep_trace or not lookup):
                # the reference refreshes state_pool at the end of hops 0 and 1 only (pgpr.py:481); with at most
                # three hops every hop sees a fresh state, which is the same thing.
                P.state(g, U32, E32, R32, self.state_history, paths, n_rows, rmax, buf["state"])
            if overlap:
                # the action lists and the policy logits of a hop both depend on the hop's path rows only -- and on different
                # units (the action kernels are scalar- / vector-issue bound, the policy launch matrix- and L2-bound): the action
                # kernels go to a stream of their own, the tail below waits for both
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    P.actions(g, score_user, score_entity, self._score_rel_entity, paths, n_rows, rmax, self.max_acts,
                              self.fix_scores_sorting_bug, buf["acts"], buf["n_acts"], buf["scratch"], u16=acts_u16)
                self._policy_logits(buf, n_rows, rmax, drops, hop, paths if lookup else None)
                cur.wait_stream(side)
            else:
                P.actions(g, score_user, score_entity, self._score_rel_entity, paths, n_rows, rmax, self.max_acts,
                          self.fix_scores_sorting_bug, buf["acts"], buf["n_acts"], buf["scratch"], u16=acts_u16)
                mark("actions", hop)
                self._policy_logits(buf, n_rows, rmax, drops, hop, paths if lookup else None)
                mark("policy", hop)
            if fused_tail and P.topk_expand_supported(self.act_dim, k):
                last = hop == len(hops) - 1
                mark("topk", hop)  # no launch of its own: the stage is part of "expand"
                P.topk_expand(g, paths, probs, buf["acts"], self.act_dim, buf["n_acts"], n_rows, rmax, buf["logits"], k,
                              buf["offsets"], buf["paths"][hop + 1], buf["probs"][hop + 1],
                              buf["leaves"] if (last and compact_leaves) else None, nr[hop + 1 : hop + 2],
                              buf["rows"][hop + 1], acts_u16=acts_u16)
            else:
                P.policy_topk(buf["logits"], self.act_dim, buf["n_acts"], n_rows, rmax, k, buf["tk_idx"], buf["tk_val"])
                mark("topk", hop)
                P.expand(g, paths, probs, buf["acts"], self.act_dim, buf["n_acts"], n_rows, rmax, buf["tk_idx"],
                         buf["tk_val"], k, buf["offsets"], buf["paths"][hop + 1], buf["probs"][hop + 1],
                         nr[hop + 1 : hop + 2], buf["rows"][hop + 1])
            mark("expand", hop)
            if keep_trace:
                n = int(n_rows.item())
                trace.append(dict(acts=buf["acts"][:n].clone(), n_acts=buf["n_acts"][:n].clone(),
                                  state=buf["state"][:n].clone(), logits=buf["logits"][:n].clone(),
                                  tk_idx=buf["tk_idx"][:n, :k].clone(), tk_val=buf["tk_val"][:n, :k].clone()))
        if compact_leaves:  # parents = the rows the last hop expanded
            res = BeamResult(buf["paths"][-2], buf["probs"][-2], nr[len(hops) : len(hops) + 1], graph=g,
                             leaves=buf["leaves"], child_offsets=buf["offsets"],
                             n_parents=nr[len(hops) - 1 : len(hops)])
        else:
            res = BeamResult(buf["paths"][-1], buf["probs"][-1], nr[len(hops) : len(hops) + 1], graph=g)
        res.n_rows_per_hop = nr
        res.trace = trace
        res.buffers = buf if keep_buffers else None  # per-hop path rows / action counts (bench_pgpr's byte accounting)
        return res

    def _batch_scores(self, users, plain=False, all_users=False):
        """(score_user [B, n_users], score_entity [B, n_entities]) float64 of a batch's root users: u . U[n] and (u + r_ui) . E[n]
        for every node n -- what the trims and the reward read (pgpr.py:268-276, 339-357).  ``plain``: the (U + R_ui) form
        even when the model swaps the [UI-Relation] row per user (TPRec's u_p_scales, tprec.py:146-149).

        The tables these are products of are frozen (pretrained TransE embeddings loaded at construction), so the scores of
        EVERY user are computed once -- ``hw_pgpr_user_scores`` over all users in chunks, the same fma chains -- when they fit
        ``score_cache_mb`` (default 2 048: ml-1m's 6 040 users x 19 845 nodes x 8 B = 0.96 GB), and a batch is two row gathers:
        0.19 -> 0.07 ms of a 1 024-user training step, 0.10 -> 0.02 ms of a 256-user search.  ``all_users``: the whole cached
        tables (None when they are not kept) -- a training episode reads them in place, its records carrying the user's row."""
        g = self.graph
        U64, E64, R64 = self._t64
        vec = None if plain else self._user_vec()
        budget = self.score_cache_mb << 20
        slot = "_score_cache_plain" if vec is None else "_score_cache"
        if (g.n_users * (g.n_users + g.n_entities) * 8 > budget
                or (U64.is_cuda and torch.cuda.is_current_stream_capturing() and not self._fenced(slot).has(0))):
            if all_users:
                return None
            return P.user_scores(g, U64, E64, R64, users, vec)  # too big to keep, or not yet built and no place to build it

        def build():
            su = torch.empty((g.n_users, g.n_users), dtype=torch.float64, device=U64.device)
            se = torch.empty((g.n_users, g.n_entities), dtype=torch.float64, device=U64.device)
            ids = torch.arange(g.n_users, device=U64.device)
            for lo in range(0, g.n_users, 2048):
                a, b = P.user_scores(g, U64, E64, R64, ids[lo : lo + 2048], vec)
                su[lo : lo + 2048], se[lo : lo + 2048] = a, b
            return su, se

        su, se = self._fenced(slot).get(0, build)
        if all_users:
            return su, se
        rows = users.clamp(0, g.n_users - 1)  # hw_pgpr_user_scores clamps the same way
        return su.index_select(0, rows), se.index_select(0, rows)

    def _all_user_scales(self):
        """u_p_scales of every user (pgpr.py:85-88: the largest item score of the plain tables), next to the score cache."""
        return self._fenced("_scale_cache").get(
            0, lambda: self._batch_scores(None, plain=True, all_users=True)[1][:, : self.n_items].max(dim=1).values.contiguous())

    def _user_vec(self):
        """Per-user replacement of the [UI-Relation] row in the trimming / reward scores (None: PGPR; TPRec overrides)."""
        return None

    def _collect_tables(self):
        """(U, E, R) float64 tables collect_scores reads; TPRec swaps the [UI-Relation] row."""
        return self._t64

    # -- training: one REINFORCE episode per batch (pgpr.py:120-203, 335-396) ---------------------------------------
    def _head_params(self):
        """(actor | critic) weights and biases stacked for the one head product of a step; rebuilt when a parameter moves."""
        ps = (self.actor.weight, self.critic.weight, self.actor.bias, self.critic.bias)
        build = lambda: (torch.cat([self.actor.weight, self.critic.weight], 0).detach().contiguous(),
                         torch.cat([self.actor.bias, self.critic.bias], 0).detach().contiguous())
        if ps[0].is_cuda and torch.cuda.is_current_stream_capturing():
            return build()  # a captured step must hold the stacking itself: a replay moves the weights without running this
        key = tuple((t._version, t.data_ptr()) for t in ps)
        return self._fenced("_head_cache").get(key, build)

    @torch.no_grad()
    def episode_device(self, users, forced_acts=None, act_masks=None, drops=None):
        """Roll the policy out for ``max_path_len`` steps on the device and evaluate update()'s loss.

        forced_acts (int32 [T, B]), act_masks (uint8 [T, B, act_dim]) and drops (per step (mask1, mask2)) replace the
        device random streams -- what the parity tests use to replay the reference's trajectory.  Returns the episode
        record that :meth:`_episode_backward` consumes."""
        dev = self._t64[0].device
        users = users.to(dev, torch.int64).contiguous()
        B, T, A = users.numel(), self.max_path_len, self.act_dim
        g = self.graph
        U64, E64, R64 = self._t64
        U32, E32, R32 = self._t32
        H1, H2 = self.hidden_sizes
        i32 = lambda *s: torch.empty(s, dtype=torch.int32, device=dev)
        f32 = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        ep = dict(B=B, T=T, state=f32(T * B, self.state_dim), x1=f32(T * B, H1), x2=f32(T * B, H2),
                  head=f32(T * B, A + 1), probs=f32(T * B, A), acts=i32(T * B), logp=f32(T * B), entropy=f32(T * B),
                  value=f32(T * B), mask=torch.empty((T * B, A), dtype=torch.uint8, device=dev),
                  rewards=torch.empty(T * B, dtype=torch.float64, device=dev), drop=[], calls=[], n_acts=i32(T * B),
                  paths=[i32(B, P.ROW_INTS) for _ in range(T + 1)])
        probs3 = [f32(B, 3) for _ in range(2)]
        nr = torch.zeros(T + 1, dtype=torch.int32, device=dev)
        acts_buf, offsets = i32(B, A, 2), i32(B)
        scratch = None
        if g.max_degree > P.ACT_LDS_KEYS:
            scratch = torch.empty((min(2048, B), g.max_degree), dtype=torch.int64, device=dev)
        head_w, head_b = self._head_params()
        P.init_paths(users, ep["paths"][0], probs3[0], nr[0:1])
        own_scales = self._user_vec() is None  # u_p_scales always come from the plain (U + R_ui) tables (tprec.py:146-149)
        cached = self._batch_scores(None, all_users=True)
        if cached is not None:
            # the scores of all users are kept (frozen embeddings): the episode reads them in place -- the records' last word,
            # which names the row of the score tables, is the user id instead of the slot in the batch (nothing else in an
            # episode reads that word) -- and the scales come from their cache: no table of the batch is made at all
            score_user, score_entity = cached
            scales = self._all_user_scales()
            ep["paths"][0][:, P.ROW_INTS - 1] = users.clamp(0, g.n_users - 1).to(torch.int32)
            own_scales = False  # nothing to compute inside the step
        else:
            score_user, score_entity = self._batch_scores(users)
            scales = torch.empty(B, dtype=torch.float64, device=dev)
            if not own_scales:
                _, plain = self._batch_scores(users, plain=True)
                P.reward(ep["paths"][0], B, None, plain, self.n_items, scales, True, ep["rewards"][:B])
        for t in range(T):
            sl = slice(t * B, (t + 1) * B)
            paths, n_rows = ep["paths"][t], nr[t : t + 1]
            n_acts = ep["n_acts"][sl]  # the step's action counts land in the episode record directly
            P.actions(g, score_user, score_entity, self._score_rel_entity, paths, n_rows, B, self.max_acts,
                      self.fix_scores_sorting_bug, acts_buf, n_acts, scratch, start_user_revisitable=True)
            call = self._philox.next()
            off = [self._philox.offset(call, j) for j in range(4)]
            ep["calls"].append(call)
            if act_masks is not None:
                ep["mask"][sl].copy_(act_masks[t])
            else:
                P.action_dropout(n_acts, n_rows, B, A, self.action_dropout, self._seed, off[2], ep["mask"][sl])
            P.state(g, U32, E32, R32, self.state_history, paths, n_rows, B, ep["state"][sl])
            if drops is not None:
                modes = [(drops[t][0], 1, 0), (drops[t][1], 1, 0)]
            elif self.policy_dropout == "reference":
                modes = [(None, 2, off[0]), (None, 2, off[1])]
            else:
                modes = [(None, 0, 0), (None, 0, 0)]
            ep["drop"].append(modes)
            P.mlp_layer(ep["state"][sl], self.l1.weight, self.l1.bias, ep["x1"][sl], n_rows, B, True, modes[0][0],
                        modes[0][1], self._seed, modes[0][2])
            P.mlp_layer(ep["x1"][sl], self.l2.weight, self.l2.bias, ep["x2"][sl], n_rows, B, True, modes[1][0],
                        modes[1][1], self._seed, modes[1][2])
            P.mlp_layer(ep["x2"][sl], head_w, head_b, ep["head"][sl], n_rows, B, False)
            P.policy_sample(ep["head"][sl], A, ep["mask"][sl], n_rows, B, ep["probs"][sl], ep["acts"][sl], ep["logp"][sl],
                            ep["entropy"][sl], ep["value"][sl], self._seed, off[3],
                            None if forced_acts is None else forced_acts[t].contiguous())
            P.expand(g, paths, probs3[t % 2], acts_buf, A, n_acts, n_rows, B, ep["acts"][sl].view(B, 1),
                     ep["logp"][sl].view(B, 1), 1, offsets, ep["paths"][t + 1], probs3[(t + 1) % 2], nr[t + 1 : t + 2], B)
            P.reward(ep["paths"][t + 1], B, self._patterns, score_entity, self.n_items, scales, own_scales and t == 0,
                     ep["rewards"][sl])
        ep["raw_rewards"] = ep["rewards"].clone()
        ep["loss"] = torch.empty(4, dtype=torch.float64, device=dev)
        ep["coef_actor"], ep["dvalue"] = f32(T * B), f32(T * B)
        P.reinforce_loss(ep["rewards"], ep["value"], ep["logp"], ep["entropy"], T, B, self.gamma, self.weight_factor,
                         ep["loss"], ep["coef_actor"], ep["dvalue"])
        ep["head_w"] = head_w
        return ep

    @torch.no_grad()
    def _episode_backward(self, ep, upstream=None):
        """Gradients of update()'s loss w.r.t. (l1.w, l1.b, l2.w, l2.b, actor.w, actor.b, critic.w, critic.b): the
        head gradient per row, then one stacked GEMM pass over all T * B rows."""
        from ..ops import gemm

        B, T, A = ep["B"], ep["T"], self.act_dim
        n = T * B
        dhead = torch.empty_like(ep["head"])
        P.policy_grad(ep["probs"], A, ep["acts"], ep["coef_actor"], self.weight_factor / B, ep["dvalue"], n, dhead, upstream)
        g_head_w = gemm(dhead, ep["x2"], a_k_contiguous=False, b_k_contiguous=False)
        g_head_b = P.colsum(dhead)
        dx2 = gemm(dhead, ep["head_w"], b_k_contiguous=False)  # [n, H2]
        for t in range(T):
            sl = slice(t * B, (t + 1) * B)
            mask, mode, off = ep["drop"][t][1]
            P.act_dropout_backward(dx2[sl], ep["x2"][sl], 1, mask, mode, self._seed, off, out=dx2[sl])
        g_w2 = gemm(dx2, ep["x1"], a_k_contiguous=False, b_k_contiguous=False)
        g_b2 = P.colsum(dx2)
        dx1 = gemm(dx2, self.l2.weight.detach(), b_k_contiguous=False)  # [n, H1]
        for t in range(T):
            sl = slice(t * B, (t + 1) * B)
            mask, mode, off = ep["drop"][t][0]
            P.act_dropout_backward(dx1[sl], ep["x1"][sl], 1, mask, mode, self._seed, off, out=dx1[sl])
        g_w1 = gemm(dx1, ep["state"], a_k_contiguous=False, b_k_contiguous=False)
        g_b1 = P.colsum(dx1)
        return (g_w1, g_b1, g_w2, g_b2, g_head_w[:A].contiguous(), g_head_b[:A].contiguous(),
                g_head_w[A:].contiguous(), g_head_b[A:].contiguous())

    def calculate_loss(self, interaction):
        """pgpr.py:189-203: roll out, update(); the float64 loss carries a grad_fn over the eight policy tensors."""
        users = interaction[self.USER_ID]
        if users.is_cuda and torch.cuda.is_current_stream_capturing():
            # a captured step has one shape: the [PAD] filter below (a host read) cannot run.  hopwise's train loaders never
            # yield user 0; should one appear the flag is raised on the device and driver.graphed_trainer / train_epoch_graphed
            # refuse the epoch when they read it (capture_error_flags)
            if self.__dict__.get("_pad_user_flag") is None or self._pad_user_flag.device != users.device:
                raise RuntimeError("PGPR: build the step with driver.GraphedTrainStep (it prepares the capture's flags)")
            self._pad_user_flag.logical_or_((users == 0).any())
        else:
            users = users[users != 0]
        if users.numel() == 0:
            return sum(p.sum() for p in self.parameters()).double() * 0.0
        params = (self.l1.weight, self.l1.bias, self.l2.weight, self.l2.bias, self.actor.weight, self.actor.bias,
                  self.critic.weight, self.critic.bias)
        return _EpisodeLoss.apply(self, users, *params)

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
        step = self._users_per_search()
        for s in range(0, users.numel(), step):
            res = self.beam_search_device(users[s : s + step])
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
        if getattr(res, "leaves", None) is not None:  # compact leaves: read in place, nothing is materialised
            return P.collect(self.graph, *self._collect_tables(), users, self._pos_bits, res.parent_paths, res.parent_probs,
                             res._n, res.leaves.shape[0], n_nodes, max(self.topk), leaves=res.leaves,
                             child_offsets=getattr(res, "child_offsets", None), n_parents=getattr(res, "n_parents", None))
        return P.collect(self.graph, *self._collect_tables(), users, self._pos_bits, res.paths, res.probs, res._n,
                         res.paths.shape[0], n_nodes, max(self.topk))

    def explanations(self, users, res, sel):
        """The reference's [user, item, score, path] list, in the order collect_scores appends them."""
        cnt, item, score, leaf = (sel[k].cpu().numpy() for k in ("count", "item", "score", "leaf"))
        ulist = users.cpu().numpy()
        need = np.unique(leaf[leaf >= 0])
        rows = res.rows(torch.from_numpy(need))[0].cpu().numpy() if len(need) else np.zeros((0, 12))
        prs = np.zeros((len(need), 3), np.float32)
        dec, _ = self.decode_paths(rows, prs)
        by_leaf = dict(zip(need.tolist(), dec))
        out = []
        for slot in range(len(ulist)):
            for j in range(int(cnt[slot])):
                out.append([int(ulist[slot]), int(item[slot, j]), float(score[slot, j]), by_leaf[int(leaf[slot, j])]])
        return out

    TARGET_BEAM_ROWS = 1 << 20  # rows of the widest expanded hop of one search when beam_user_chunk is left to the model

    def _users_per_search(self, hops=None):
        if self.user_chunk > 0:
            return min(self.user_chunk, 65535)
        hops = list(self.beam_search_hop if hops is None else hops)
        widest = 1
        for k in hops[:-1]:
            widest *= int(k)
        n = self.TARGET_BEAM_ROWS // max(widest, 1)
        # the float64 trim tables of a search are [users, n_users + n_entities]: at most 4 GB of them
        g = getattr(self, "graph", None)
        if g is not None:
            n = min(n, (4 << 30) // (8 * max(1, g.n_users + g.n_entities)))
            # the collector's [users, n_items] tables (best key u64 + score f64 + result f32 = 20 B per cell), held by up to
            # ``search_streams`` searches at a time: at most 8 GB of them
            n = min(n, (8 << 30) // (20 * max(1, self.n_items) * max(1, self.search_streams)))
        n = min(n, 65535)  # hw_pgpr_collect numbers a search's users in 16 bits
        return max(64, n // 64 * 64)

    def _search_and_collect(self, users):
        """[(user chunk, BeamResult, scores, selection)] for chunks of ``user_chunk`` users.  More than one chunk: the
        searches are dealt round-robin over ``search_streams`` side streams -- they are independent and hold no state
        between calls (tests/test_gpu_pgpr_streams.py) -- fenced against the caller's stream on both sides, so that the caller
        sees what it would see after the same searches one after another."""
        users = torch.as_tensor(users)
        step = self._users_per_search()
        chunks = [users[s : s + step] for s in range(0, users.numel(), step)]
        dev = self._t64[0].device
        n_streams = min(self.search_streams, len(chunks))
        if n_streams <= 1 or dev.type != "cuda":
            outs = []
            for chunk in chunks:
                res = self.beam_search_device(chunk)
                scores, sel = self.collect_scores_device(chunk, res)
                outs.append((chunk, res, scores, sel))
            return outs
        main = torch.cuda.current_stream(dev)
        # side streams come from a module-level pool per device, not from the model (stream objects do not survive the
        # copy.deepcopy / pickle a caller may apply to a model) -- and from streams that were PROBED to run beside each other
        # and beside the caller's: torch's pooled streams share four hardware queues, and three searches of which two took
        # turns on one queue measured 5.2 ms per 256 users against 4.8 (stream_pool.py)
        side = concurrent_streams(n_streams, dev, beside=[main])
        self._warm_search_caches()  # on the caller's stream, ahead of the fences below
        for st in side[:n_streams]:
            st.wait_stream(main)  # the ids, the weights and whatever else the caller queued are in place
        outs = []
        for i, chunk in enumerate(chunks):
            with torch.cuda.stream(side[i % n_streams]):
                res = self.beam_search_device(chunk, overlap_actions=False)
                scores, sel = self.collect_scores_device(chunk, res)
            outs.append((chunk, res, scores, sel))
        for st in side[:n_streams]:
            main.wait_stream(st)
        for _, res, scores, sel in outs:  # allocated on a side stream, used by the caller on its own from here on
            res.record_stream(main)
            scores.record_stream(main)
            for t in sel.values():
                if torch.is_tensor(t):
                    t.record_stream(main)
        return outs

    def full_sort_predict(self, interaction):  # pgpr.py:408-415
        return torch.cat([o[2] for o in self._search_and_collect(interaction[self.USER_ID])], dim=0)

    def explain(self, interaction):  # pgpr.py:417-435
        """(scores, [[user, item, score, path], ...]).  The record list is built when it is read (hopwise.lazy.LazyList):
        ExplainableCollector keeps it only for ``rec.paths`` metrics, and building it -- device-to-host copies of the
        selections, a Python loop over every (user, item) -- costs more than the searches."""
        outs = self._search_and_collect(interaction[self.USER_ID])
        scores = torch.cat([o[2] for o in outs], dim=0)

        def build():
            expl = []
            for chunk, res, _, sel in outs:
                expl += self.explanations(chunk, res, sel)
            for e in expl:
                e[-1] = self.decode_path(e[-1])
            return expl

        return scores, LazyList(build)

    def predict(self, interaction):
        return

    def decode_path(self, path):
        return path
