"""TransE / RotatE / ComplEx / DistMult with the reference's constructor, methods, config keys and
``state_dict`` names, computing on the gfx950 kernels.

Reference classes (paths under /root/reference/hopwise/model/knowledge_graph_embedding_recommender):
  TransE transe.py:21-154, RotatE rotate.py:24-220, ComplEx complex.py:21-219, DistMult distmult.py:21-146.

What is kept identical: ``Model(config, dataset)``; parameter names (checkpoint compatible);
xavier-normal init (+ U(0, 2pi) phases for RotatE, rotate.py:58-59); ``calculate_loss`` semantics per
model (TripletMarginLoss over cat(rec, kg) for TransE; BCEWithLogits on rec and kg separately, summed,
for RotatE/ComplEx; MarginRankingLoss over cat(rec, kg) for DistMult); ``predict``, ``predict_kg``,
``full_sort_predict``, ``full_sort_predict_kg``; which relation row the rec side uses
(``relation_embedding.weight[-1]`` vs the ``ui_relation`` token id, per method, as in the reference).

What differs by design: no [B, d] gathered temporaries and no [B, N, d] broadcast temporaries exist;
rows are gathered inside the scoring kernel, gradients are scattered straight into dense grad tables.

Extension (NOT in the reference, default off -- SURVEY.md §8 A9): ``self_adversarial: True`` with
``adversarial_temperature`` switches the rec-side loss of RotatE/ComplEx to self-adversarial negative
weighting over the ``train_neg_sample_args.sample_num`` negatives of each positive.
"""

from __future__ import annotations

import math
import os

import torch
from torch import nn

from .. import ops
from ..philox import PhiloxCounter
from ..stream_cache import FencedValue
from ..utils import InputType, seed_for_rank
from .abstract_recommender import KnowledgeRecommender


def xavier_normal_initialization(module):
    """model/init.py:13-29 of the reference."""
    if isinstance(module, nn.Embedding):
        nn.init.xavier_normal_(module.weight.data)
    elif isinstance(module, nn.Linear):
        nn.init.xavier_normal_(module.weight.data)
        if module.bias is not None:
            nn.init.constant_(module.bias.data, 0)


def _cfg(config, key, default):
    try:
        v = config[key]
    except (KeyError, IndexError):
        return default
    return default if v is None else v


class _GatherKGE(KnowledgeRecommender):
    """Shared driver of the four gather-bound scorers."""

    input_type = InputType.PAIRWISE
    MODEL = None  # "TransE" | "RotatE" | "ComplEx" | "DistMult"
    _narrow_rec_tails = True  # False when the tables handed to the kernels are compact lookups (parallel.py)
    USER_TABLES = ()  # attribute names, slot order
    ENTITY_TABLES = ()
    RELATION_TABLES = ()

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        self.embedding_size = config["embedding_size"]
        self.margin = float(_cfg(config, "margin", 0.0))
        self.ui_relation = dataset.field2token_id["relation_id"][dataset.ui_relation]
        for n in self.USER_TABLES:
            setattr(self, n, nn.Embedding(self.n_users, self.embedding_size))
        for n in self.ENTITY_TABLES:
            setattr(self, n, nn.Embedding(self.n_entities, self.embedding_size))
        for n in self.RELATION_TABLES:
            setattr(self, n, nn.Embedding(self.n_relations, self.embedding_size * self.REL_ROW_FACTOR()))
        # extension knobs (absent from reference configs -> reference behaviour)
        self.self_adversarial = bool(_cfg(config, "self_adversarial", False))
        self.adversarial_temperature = float(_cfg(config, "adversarial_temperature", 1.0))
        self.sample_num = int(_cfg(_cfg(config, "train_neg_sample_args", {}) or {}, "sample_num", 1))
        # sample_num > 1: trust the dataloader's tile order (row j*P + i = j-th negative of positive i); set False for
        # hand-made batches that repeat positives in another order
        self.rec_tiled_layout = bool(_cfg(config, "rec_tiled_layout", True))
        # calculate_loss as one autograd node (ops.FusedPairLoss); False keeps the per-op graph (scorer nodes + loss nodes)
        self.fused_loss = bool(_cfg(config, "fused_loss", True))
        # bilinear scorers (BILINEAR below): full sorts as one product on the MFMA GEMM (HW_FULLSORT_GEMM=0: the gather kernel)
        self.gemm_full_sort = bool(_cfg(config, "gemm_full_sort", os.environ.get("HW_FULLSORT_GEMM", "1") != "0"))
        # full_sort_topk of the product form: exact re-rank of the top candidates.  Unset: on for the distance scorers (whose
        # product scores carry the expansion's rounding) and "auto" for the bilinear ones -- their products are fp32-accurate as
        # they are, so the re-rank is taken where it is the FASTER route: wide tables, where the three-term ranking product with
        # its in-epilogue mask and stretch maxima beats the six-term product + mask + top-k (and returns the gather path's scores)
        rr = _cfg(config, "gemm_rerank", os.environ.get("HW_FULLSORT_RERANK"))
        self.gemm_rerank = ((True if self.GEMM_FORM == "distance" else "auto") if rr is None
                            else str(rr).lower() not in ("0", "false"))
        self.gemm_min_scores = int(_cfg(config, "gemm_min_scores", os.environ.get("HW_FULLSORT_GEMM_MIN", self.GEMM_MIN_SCORES)))
        # split terms of the evaluation top-k's ranking product (distance scorers, with the exact re-rank): 3 (hw_gemm_rank) or 6
        self.gemm_rank_planes = str(_cfg(config, "gemm_rank_planes", os.environ.get("HW_FULLSORT_RANK_PLANES", "1"))).lower() not in ("0", "false")
        self.gemm_rank_terms = 6 if str(_cfg(config, "gemm_rank_terms", os.environ.get("HW_FULLSORT_RANK_TERMS", "3"))) == "6" else 3
        self.apply(xavier_normal_initialization)

    def REL_ROW_FACTOR(self):
        return 1

    # -- table tuples in slot order ---------------------------------------------------------------
    def _w(self, names):
        return tuple(getattr(self, n).weight for n in names)

    def _tables(self, side):
        head = self._w(self.USER_TABLES if side == "rec" else self.ENTITY_TABLES)
        return head, self._w(self.RELATION_TABLES), self._w(self.ENTITY_TABLES)

    # which relation row the rec side reads, per method (mirrors each reference method)
    def _rec_rel_loss(self):
        return self.ui_relation

    def _rec_rel_full_sort(self):
        return self.ui_relation

    LOSS_MODEL = None  # scorer used inside calculate_loss when it differs from MODEL (TorusE trains as TransE)

    def _score(self, side, h_idx, r_idx, r_fixed, t_idx, eps=0.0, j_major_out=False, model=None):
        head, rel, tail = self._tables(side)
        # rec-side tails are items = the first n_items entity rows (kg_dataset.py:556-588)
        tail_rows = self.n_items if (side == "rec" and self._narrow_rec_tails) else None
        return ops.kge_score_grouped(model or self.MODEL, head, rel, tail, h_idx, r_idx, r_fixed, t_idx, self.margin, eps,
                                     j_major_out, tail_rows)

    @staticmethod
    def _pair(pos, neg):
        """[B, 2] index view (column 0 positive) over one [2, B] buffer."""
        return torch.stack([pos, neg], dim=0).t()

    def _pair_scores(self, interaction, eps=0.0):
        """([2, B_rec], [2, B_kg]) score matrices: row 0 positives, row 1 negatives."""
        user = interaction[self.USER_ID]
        m = self.LOSS_MODEL
        rec = self._score("rec", user, None, self._rec_rel_loss(),
                          self._pair(interaction[self.ITEM_ID], interaction[self.NEG_ITEM_ID]), eps, True, m)
        kg = self._score("kg", interaction[self.HEAD_ENTITY_ID], interaction[self.RELATION_ID], 0,
                         self._pair(interaction[self.TAIL_ENTITY_ID], interaction[self.NEG_TAIL_ENTITY_ID]), eps, True, m)
        return rec.t(), kg.t()

    def _fused_pair_loss(self, interaction, kind, eps=0.0, scale=1.0, group_rec=False):
        """The reference's calculate_loss as one autograd node (ops.FusedPairLoss): rec side (user, [UI-Relation], item /
        neg item) + kg side (head, relation, tail / neg tail).  ``group_rec``: the rec batch tiles each positive
        sample_num times (abstract_dataloader.py:192-198); score it as [P, 1 + K] groups, positives weighted K."""
        n_rec, n_kg = interaction[self.USER_ID].numel(), interaction[self.HEAD_ENTITY_ID].numel()
        if n_rec == 0 or n_kg == 0:
            # The reference on an empty side: its gathers give [0, d] rows and a mean over no element is NaN -- for the BCE
            # models, which take one mean per side (rotate.py:123-131), as soon as ONE side is empty; for the margin models,
            # which take one mean over the concatenated rows (transe.py:91-96), only when both are.  Trainer._check_nan
            # (trainer.py:337-339) turns the NaN into its ValueError; the kernels are not asked to score nothing.
            if kind == "bce" or (n_rec == 0 and n_kg == 0):
                return torch.full((), float("nan"), device=interaction[self.USER_ID].device)
            # a margin loss with one empty side: the mean over the other side's rows alone (an empty side is left out below)
        # table positions in the parameter list are static per model: built once (unless the tables are supplied per call,
        # as parallel.RowShardedKGE does with its compact tables)
        static = "_tables" not in self.__dict__
        cache = self.__dict__.get("_pair_layout") if static else None
        if cache is None:
            params, pos = [], {}

            def at(t):
                if id(t) not in pos:
                    pos[id(t)] = len(params)
                    params.append(t)
                return pos[id(t)]

            layout = {}
            for name in ("rec", "kg"):
                head, rel, tail = self._tables(name)
                layout[name] = ([at(t) for t in head], [at(t) for t in rel], [at(t) for t in tail])
            cache = (params, layout)
            if static:
                self.__dict__["_pair_layout"] = cache
        params, layout = cache

        def side(name, h_idx, r_idx, r_fixed, t_idx, pos_weight=1.0):
            tail_rows = self.n_items if (name == "rec" and self._narrow_rec_tails) else None
            return ops.PairSide(*layout[name], h_idx, r_idx, r_fixed, t_idx, tail_rows, pos_weight)

        sides = []
        if n_rec and group_rec:
            users, tails = self._rec_grouped_from_tiled(interaction)
            sides.append(side("rec", users, None, self._rec_rel_loss(), tails, float(self.sample_num)))
        elif n_rec:
            sides.append(side("rec", interaction[self.USER_ID], None, self._rec_rel_loss(),
                              self._pair(interaction[self.ITEM_ID], interaction[self.NEG_ITEM_ID])))
        if n_kg:
            sides.append(side("kg", interaction[self.HEAD_ENTITY_ID], interaction[self.RELATION_ID], 0,
                              self._pair(interaction[self.TAIL_ENTITY_ID], interaction[self.NEG_TAIL_ENTITY_ID])))
        return ops.fused_pair_loss(self.LOSS_MODEL or self.MODEL, kind, self.margin, eps, scale, sides, params)

    # -- reference API ------------------------------------------------------------------------------
    def forward(self, *rows):
        raise NotImplementedError(
            "hopwise_amd scorers gather inside the kernel; use predict / predict_kg / calculate_loss"
        )

    def predict(self, interaction):
        user = interaction[self.USER_ID]
        item = interaction[self.ITEM_ID]
        return self._score("rec", user, None, self._rec_rel_loss(), item.view(-1, 1)).view(-1)

    def predict_kg(self, interaction):
        head = interaction[self.HEAD_ENTITY_ID]
        return self._score("kg", head, interaction[self.RELATION_ID], 0,
                           interaction[self.TAIL_ENTITY_ID].view(-1, 1)).view(-1)

    # -- full sorts as one dense product ------------------------------------------------------------------------------
    # DistMult / ComplEx / Analogy score <query(h, r), t> with the tail's tables side by side (distmult.py:50-51,
    # complex.py:53-62, analogy.py:66-70): against EVERY candidate that is query [B, K] x table^T [K, N] -- the genuinely dense
    # contraction the MFMA GEMM is for.  TransE and hopwise's RotatE are L2 distances over the row (transe.py:55-57;
    # rotate.py:61-69: the Frobenius norm of the (re, im) differences), so ||q - t||^2 = ||q||^2 + ||t||^2 - 2 <q, t> is the
    # same product followed by one in-place pass (csrc/kge_fullsort_gemm.hip: hw_kge_dist_finish, which recomputes the few
    # entries where the expansion cancels).  The gather kernel streams every tail row past four queries held in registers
    # and is bound by vector-instruction issue (~20 wave instructions per score); the product on the split-bf16 GEMM
    # (fp32-level error) is 3-4x faster at Yelp2018's 45 539 items (tools/bench_fullsort_gemm.py).
    # The query rows come from the scorers' own fold in one launch (hw_kge_query_rows).
    GEMM_FORM = None  # "bilinear" | "distance": the scorer's full sort is a dense product
    RERANK_MARGIN = 6  # full_sort_topk of the product form re-scores the top-(k + margin, rounded up to 8, at least 16) exactly
    GEMM_MIN_SCORES = 1 << 21  # distance form: below B x N = 2 M scores the gather kernels' one or two launches win

    @classmethod
    def _rerank_candidates(cls, k, n):
        return min(max(16, (k + cls.RERANK_MARGIN + 7) // 8 * 8), 64, n)

    @property
    def BILINEAR(self):
        return self.GEMM_FORM == "bilinear"

    def _cat_tables(self, tail):
        """The tail tables side by side, [rows, K]: the table itself for one table, else ONE cached copy of the whole tables
        per version (optimizers bump the version, FusedAdam included), whatever slice is asked for -- items in rec, entities
        in kg, a shard's rows.  Fenced: evaluation may run on another stream than the one that built it."""
        if len(tail) == 1:
            return tail[0].detach()
        slot = ("cat",) + tuple(t.data_ptr() for t in tail)
        return self._table_cache(slot).get(tuple(t._version for t in tail),
                                           lambda: torch.cat([t.detach() for t in tail], dim=1).contiguous())

    def _table_cache(self, slot):
        cache = self.__dict__.setdefault("_gemm_tables", {})
        fv = cache.get(slot)
        if fv is None:
            for stale in [s for s in cache if s[0] == slot[0]]:
                del cache[stale]  # a table moved (resize, .to()): the old copy is dead weight
            fv = cache[slot] = FencedValue()
        return fv

    def _bilinear_table(self, tail, n):
        return self._cat_tables(tail)[:n]

    def _tail_sqnorm(self, tail):
        """||t||^2 of every candidate row, cached per table version (the distance form)."""
        slot = ("sqnorm",) + tuple(t.data_ptr() for t in tail)
        return self._table_cache(slot).get(tuple(t._version for t in tail), lambda: ops.rows_sqnorm(self._cat_tables(tail)))

    RANK_PLANES_MIN_K = 128  # from this inner width on the plane-image product beats the one that splits on the fly

    def _rank_planes(self, tail, n):
        """The first ``n`` candidate rows as (hi, mid) bf16 plane images for the ranking product (ops.rank_planes), cached per
        table version: an evaluation epoch scores every batch against the same table."""
        slot = ("rankplanes", n) + tuple(t.data_ptr() for t in tail)
        return self._table_cache(slot).get(tuple(t._version for t in tail), lambda: ops.rank_planes(self._cat_tables(tail)[:n]))

    def _rank_bias(self, tail):
        """(-||t||^2 / 2 per candidate row, max ||t||^2): the per-column bias that turns the product into the distance form's
        ranking value, and the bound the re-rank's proof needs; cached per table version."""
        slot = ("rankbias",) + tuple(t.data_ptr() for t in tail)
        ver = tuple(t._version for t in tail)

        def build():
            tn = self._tail_sqnorm(tail)
            return tn * -0.5, tn.max().reshape(1)

        bias, tmax = self._table_cache(slot).get(ver, build)
        cache = self.__dict__.setdefault("_tn_max_host", {})
        if (slot, ver) not in cache:  # one host read per table version (the re-rank kernel takes the bound by value)
            cache.clear()
            cache[(slot, ver)] = float(tmax.item())
        return bias, cache[(slot, ver)]

    def _full_sort_gemm(self, side, h_idx, r_idx, r_fixed, n, padded=False):
        """[B, n] scores of every candidate.  ``padded``: rows start 256-byte aligned (a view of a wider buffer) so that the
        product's epilogue, the finish pass and the top-k read and write 16 bytes per lane."""
        head, rel, tail = self._tables(side)
        dist = self.GEMM_FORM == "distance"
        q, qn = ops.kge_query_rows(self.MODEL, head, rel, tail, h_idx, r_idx, r_fixed, sqnorm=dist, margin=self.margin)
        table = self._cat_tables(tail)[:n]
        out = None
        if padded:
            out = torch.empty((q.shape[0], (n + 63) // 64 * 64), dtype=torch.float32, device=q.device)[:, :n]
        if dist:  # the product with the distance epilogue: one launch, the dot products never reach memory
            return ops.gemm_dist(q, table, qn, self._tail_sqnorm(tail)[:n], self.margin if self.MODEL == "RotatE" else 0.0,
                                 out=out)
        return ops.gemm(q, table, out=out)

    def _use_gemm_full_sort(self, n_queries, n=None):
        if (self.GEMM_FORM is None or not self.gemm_full_sort or n_queries <= 0 or "_tables" in self.__dict__
                or self.MODEL not in ops.GEMM_QUERY_MODELS):
            return False
        # the product form is ~10 launches (query rows, product, finish, mask, top-c, re-score, re-rank, the predicated
        # fallback): measured against the gather kernels it wins from a few million scores on (tools/bench_fullsort_dist.py)
        return self.GEMM_FORM == "bilinear" or n is None or n_queries * n >= self.gemm_min_scores

    def full_sort_predict(self, interaction):
        user = interaction[self.USER_ID]
        if self._use_gemm_full_sort(user.numel(), self.n_items):
            return self._full_sort_gemm("rec", user, None, self._rec_rel_full_sort(), self.n_items)
        head, rel, tail = self._tables("rec")
        return ops.kge_full_sort(self.MODEL, head, rel, tail, user, None, self._rec_rel_full_sort(), 0, self.n_items,
                                 self.margin)

    def full_sort_predict_kg(self, interaction):
        if self._use_gemm_full_sort(interaction[self.HEAD_ENTITY_ID].numel(), self.n_entities):
            return self._full_sort_gemm("kg", interaction[self.HEAD_ENTITY_ID], interaction[self.RELATION_ID], 0,
                                        self.n_entities)
        head, rel, tail = self._tables("kg")
        return ops.kge_full_sort(self.MODEL, head, rel, tail, interaction[self.HEAD_ENTITY_ID],
                                 interaction[self.RELATION_ID], 0, 0, self.n_entities, self.margin)

    def full_sort_topk(self, interaction, k, history_index=None, kg=False):
        """Top-k of the masked full sort (trainer.py:731-734 + collector.py:173-183): (values [B, k], columns [B, k]).

        Product form (GEMM_FORM): scores -> mask -> top-c.  With ``gemm_rerank`` (default for the distance scorers) the c
        candidates are re-scored by the gather scorer -- the kernel ``predict`` runs -- and ranked by (exact value desc,
        column asc); the re-rank proves on the device that no column outside the candidates can belong to the top-k, and the
        fused gather kernel, queued behind it under that flag as its predicate, redoes the batch in the rare case it cannot.
        The result is the gather path's, score for score."""
        side = "kg" if kg else "rec"
        h_idx = interaction[self.HEAD_ENTITY_ID if kg else self.USER_ID]
        r_idx = interaction[self.RELATION_ID] if kg else None
        r_fixed = 0 if kg else self._rec_rel_full_sort()
        n = self.n_entities if kg else self.n_items
        head, rel, tail = self._tables(side)
        if self._use_gemm_full_sort(h_idx.numel(), n):
            c = self._rerank_candidates(k, n)
            fast_rank = self.gemm_rank_terms == 3 and 128 * c <= n <= 65536  # the three-term product + stretch maxima apply
            rerank = (fast_rank if self.gemm_rerank == "auto" else bool(self.gemm_rerank)) and k <= c
            if rerank and (self.GEMM_FORM == "distance" or fast_rank):
                # Ranking needs no scores, and no exact ones either: the candidates are re-scored below.
                #  * distance form: with -||t||^2 / 2 as the product's per-column bias the product itself orders the candidates as
                #    the score does -- no finish pass over the [B, N] matrix; bilinear form: the product is the score;
                #  * the product runs on the three largest of its six split terms (hw_gemm_rank: half the matrix work).  Its error
                #    bound, 1.2e-5 sum |q_k t_k| <= 1.2e-5 ||q|| ||t|| <= 0.6e-5 (||q||^2 + ||t||^2), is what the re-rank's proof is
                #    given: doubled in d^2 = ||q||^2 - 2 p for the distance form (2e-5 (||q||^2 + max ||t||^2); 1e-5 with the
                #    six-term product, whose own error is ~1e-7), 2e-5 ||q|| max ||t|| for the bilinear one;
                #  * the history mask and the maximum of every 64-column stretch leave the product's epilogue with the scores: the
                #    candidates are then found from the maxima and the few stretches that can hold them (hw_topk_from_tile_max)
                #    instead of by a pass over the whole [B, N] matrix.
                dist = self.GEMM_FORM == "distance"
                q, qn = ops.kge_query_rows(self.MODEL, head, rel, tail, h_idx, r_idx, r_fixed, sqnorm=True, margin=self.margin)
                bias, tn_max = self._rank_bias(tail)
                bias = bias[:n] if dist else None
                buf = torch.empty((q.shape[0], (n + 63) // 64 * 64), dtype=torch.float32, device=q.device)[:, :n]
                lite = self.gemm_rank_terms == 3
                table = self._cat_tables(tail)[:n]
                p = flag = None
                if lite and 128 * c <= n <= 65536:  # at least 2 c stretches: the k-th largest stretch maximum means something
                    bits = ops.mask_bits(history_index, q.shape[0], n, device=q.device)  # column 0 and the history pairs
                    tmax = torch.empty((q.shape[0], (n + 63) // 64), dtype=torch.float32, device=q.device)
                    if q.shape[1] >= self.RANK_PLANES_MIN_K and q.shape[0] >= 512 and self.gemm_rank_planes:
                        # operands split once (the table per weight version, the queries per call), 256-wide tiles fed by DMA
                        p = ops.gemm_rank_planes(ops.rank_planes(q), self._rank_planes(tail, n), q.shape[0], n, q.shape[1],
                                                 bias, out=buf, mask=bits, tile_max=tmax)
                    else:
                        p = ops.gemm_rank(q, table, bias, out=buf, mask=bits, tile_max=tmax)
                    if p is not None:
                        approx, cand, flag = ops.topk_from_tile_max(p, tmax, c)
                if p is None:  # six terms asked for, a very wide table, or operands the split loop does not take
                    lite = False
                    p = ops.gemm(q, table, bias=bias, out=buf)
                    ops.mask_scores_(p, history_index)
                    approx, cand = ops.topk_rows(p, c)
                desc = ops.make_desc(self.MODEL, head, rel, tail, self.margin, 0.0, n)
                exact = ops.kge_score_grouped_raw(desc, h_idx, r_idx, r_fixed, cand)  # c <= n: every candidate is a column
                if dist:
                    val, idx, flag = ops.topk_rerank(cand, approx, exact, k, rel_slack=2e-5 if lite else 1e-5, abs_slack=1e-6,
                                                     unverified=flag, qn=qn, base=self.margin if self.MODEL == "RotatE" else 0.0,
                                                     tn_max=tn_max)
                else:
                    # every slack of the product form scales with ||q|| max ||t|| (a DistMult on freshly initialised tables
                    # scores ~1e-5: an absolute 1e-6 would be a quarter of the gaps it has to resolve)
                    val, idx, flag = ops.topk_rerank(cand, approx, exact, k, rel_slack=2.2e-5 if lite else 2e-6, abs_slack=0.0,
                                                     unverified=flag, qn=qn, tn_max=tn_max, dot=True)
                return ops.kge_full_sort_topk(self.MODEL, head, rel, tail, h_idx, r_idx, r_fixed, 0, n, k, history_index,
                                              True, self.margin, run_if=flag, out=(val, idx))
            if not rerank and self.GEMM_FORM == "bilinear" and 128 * k <= n <= 65536 and k <= 64:
                # the product IS the score: mask and stretch maxima in its epilogue, the top-k from the maxima (no mask pass, no
                # top-k pass over the [B, N] matrix); the gather kernel behind the overflow flag for rows of equal scores
                q, _ = ops.kge_query_rows(self.MODEL, head, rel, tail, h_idx, r_idx, r_fixed)
                bits = ops.mask_bits(history_index, q.shape[0], n, device=q.device)
                tmax = torch.empty((q.shape[0], (n + 63) // 64), dtype=torch.float32, device=q.device)
                p = ops.gemm_rank(q, self._bilinear_table(tail, n), None, mask=bits, tile_max=tmax, exact=True)
                if p is not None:
                    val, idx, flag = ops.topk_from_tile_max(p, tmax, k)
                    return ops.kge_full_sort_topk(self.MODEL, head, rel, tail, h_idx, r_idx, r_fixed, 0, n, k, history_index,
                                                  True, self.margin, run_if=flag, out=(val, idx))
            scores = self._full_sort_gemm(side, h_idx, r_idx, r_fixed, n, padded=True)
            ops.mask_scores_(scores, history_index)
            if not rerank:
                return ops.topk_rows(scores, k)
            approx, cand = ops.topk_rows(scores, c)
            desc = ops.make_desc(self.MODEL, head, rel, tail, self.margin, 0.0, n)
            exact = ops.kge_score_grouped_raw(desc, h_idx, r_idx, r_fixed, cand)  # c <= n: every candidate is a column
            val, idx, flag = ops.topk_rerank(cand, approx, exact, k)
            return ops.kge_full_sort_topk(self.MODEL, head, rel, tail, h_idx, r_idx, r_fixed, 0, n, k, history_index, True,
                                          self.margin, run_if=flag, out=(val, idx))
        return ops.kge_full_sort_topk(self.MODEL, head, rel, tail, h_idx, r_idx, r_fixed, 0, n, k, history_index, True,
                                      self.margin)

    # -- extension: grouped negatives ---------------------------------------------------------------
    def score_grouped(self, side, h_idx, r_idx, t_idx):
        """scores [B, 1+K] of B (h, r) groups against t_idx [B, 1+K] (column 0 = positive)."""
        r_fixed = self._rec_rel_loss() if r_idx is None else 0
        return self._score(side, h_idx, r_idx, r_fixed, t_idx)

    def _rec_grouped_from_tiled(self, interaction):
        """Undo the reference's tile layout (abstract_dataloader.py:192-198): the batch holds P positives
        tiled K times, row j*P + i = j-th negative of positive i.  Returns (users [P], tails [P, 1+K])."""
        K = self.sample_num
        user = interaction[self.USER_ID]
        P = user.shape[0] // K
        pos = interaction[self.ITEM_ID][:P]
        neg = interaction[self.NEG_ITEM_ID].view(K, P).t()
        return user[:P], torch.cat([pos.view(P, 1), neg], dim=1)


class TransE(_GatherKGE):
    MODEL = "TransE"
    USER_TABLES = ("user_embedding",)
    ENTITY_TABLES = ("entity_embedding",)
    RELATION_TABLES = ("relation_embedding",)

    def _rec_rel_loss(self):  # relation_embedding.weight[-1]  (transe.py:63,108,117)
        return self.n_relations - 1

    _rec_rel_full_sort = _rec_rel_loss
    GEMM_FORM = "distance"

    def calculate_loss(self, interaction):
        # TripletMarginLoss(margin, p=2) on (h + r, t+, t-) over cat(rec, kg) (transe.py:86-96); its
        # distance is ||a - b + 1e-6||, so the loss path scores with eps = 1e-6.
        if self.fused_loss:
            return self._fused_pair_loss(interaction, "margin", eps=ops.PAIRWISE_EPS)
        rec, kg = self._pair_scores(interaction, eps=ops.PAIRWISE_EPS)
        pos = torch.cat([rec[0], kg[0]])
        neg = torch.cat([rec[1], kg[1]])
        return ops.margin_ranking_mean(pos, neg, self.margin)


class DistMult(_GatherKGE):
    MODEL = "DistMult"
    USER_TABLES = ("user_embedding",)
    ENTITY_TABLES = ("entity_embedding",)
    RELATION_TABLES = ("relation_embedding",)

    def _rec_rel_loss(self):  # relation_embedding.weight[-1]  (distmult.py:57,103,123)
        return self.n_relations - 1

    _rec_rel_full_sort = _rec_rel_loss
    GEMM_FORM = "bilinear"

    def calculate_loss(self, interaction):  # distmult.py:69-95
        if self.fused_loss:
            return self._fused_pair_loss(interaction, "margin")
        rec, kg = self._pair_scores(interaction)
        pos = torch.cat([rec[0], kg[0]])
        neg = torch.cat([rec[1], kg[1]])
        return ops.margin_ranking_mean(pos, neg, self.margin)


class RESCAL(DistMult):
    """score = h^T M_r t with M_r = relation_embedding(r).view(d, d) (rescal.py:43,51-54); MarginRankingLoss over
    cat(rec, kg) (rescal.py:73-99); the rec side uses relation_embedding.weight[-1] (rescal.py:60,108,133)."""

    MODEL = "RESCAL"
    GEMM_FORM = None  # the query h^T M_r is a d x d product per (h, r): built inside the gather kernel

    def REL_ROW_FACTOR(self):
        return self.embedding_size


class _BceKGE(_GatherKGE):
    def calculate_loss(self, interaction):
        # BCEWithLogitsLoss on cat(pos, neg) with labels (1, 0), rec and kg separately, summed
        # (rotate.py:118-131, complex.py:115-128).
        if self.self_adversarial and self.sample_num > 1:
            users, tails = self._rec_grouped_from_tiled(interaction)
            rec_scores = self.score_grouped("rec", users, None, tails)
            rec_loss = ops.self_adversarial_mean(rec_scores, self.adversarial_temperature)
            kg = self._score("kg", interaction[self.HEAD_ENTITY_ID], interaction[self.RELATION_ID], 0,
                             self._pair(interaction[self.TAIL_ENTITY_ID], interaction[self.NEG_TAIL_ENTITY_ID]),
                             0.0, True).t()
            return rec_loss + ops.bce_logits_mean(kg, kg.shape[1])
        K = self.sample_num
        grouped = K > 1 and self.rec_tiled_layout and interaction[self.USER_ID].shape[0] % K == 0
        if self.fused_loss:
            return self._fused_pair_loss(interaction, "bce", group_rec=grouped)
        if grouped:
            # The batch tiles each of its P positives K times (abstract_dataloader.py:192-198), so 2 P K rows hold only
            # P distinct positive scores: score every (user, positive) once in a [P, 1+K] group and give it weight K in
            # the mean.  Same loss value; the gradient kernel sees P groups instead of K-fold repeated rows hammering the
            # same user / item rows with atomics.
            users, tails = self._rec_grouped_from_tiled(interaction)
            rec = self._score("rec", users, None, self._rec_rel_loss(), tails, 0.0, True).t()  # [1 + K, P]: positives first
            rec_loss = ops.bce_logits_mean(rec, rec.shape[1], pos_weight=K)
            kg = self._score("kg", interaction[self.HEAD_ENTITY_ID], interaction[self.RELATION_ID], 0,
                             self._pair(interaction[self.TAIL_ENTITY_ID], interaction[self.NEG_TAIL_ENTITY_ID]),
                             0.0, True).t()
            return rec_loss + ops.bce_logits_mean(kg, kg.shape[1])
        rec, kg = self._pair_scores(interaction)
        return ops.bce_logits_mean(rec, rec.shape[1]) + ops.bce_logits_mean(kg, kg.shape[1])


class RotatE(_BceKGE):
    MODEL = "RotatE"
    GEMM_FORM = "distance"  # rotate.py:61-69: margin - the L2 norm over the 2 d (re, im) differences
    USER_TABLES = ("user_embedding", "user_embedding_im")
    ENTITY_TABLES = ("entity_embedding", "entity_embedding_im")
    RELATION_TABLES = ("relation_embedding",)

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        nn.init.uniform_(self.relation_embedding.weight, 0, 2 * math.pi)  # rotate.py:59


class ComplEx(_BceKGE):
    MODEL = "ComplEx"
    USER_TABLES = ("user_re_embedding", "user_im_embedding")
    ENTITY_TABLES = ("entity_re_embedding", "entity_im_embedding")
    RELATION_TABLES = ("relation_re_embedding", "relation_im_embedding")

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        self.margin = 0.0  # ComplEx has no margin (complex.py:30-50)

    def _rec_rel_full_sort(self):  # relation_*_embedding.weight[-1]  (complex.py:167-168)
        return self.n_relations - 1

    GEMM_FORM = "bilinear"  # query (Scorer::fold): complex.py:54-59, the reference's t_im in the last term included


class TorusE(TransE):
    """toruse.py:21-172: predict / full sorts score on the torus (-4 sum min(x^2, 1 - x^2) of the frac parts); the
    reference's calculate_loss (toruse.py:81-103) is TransE's triplet loss on the raw rows -- its forward() is not
    used in training -- so the loss path runs the TransE scorer."""

    MODEL = "TorusE"
    GEMM_FORM = None  # not an L2 distance of (q, t) / not a fold of (h, r): the gather kernels
    LOSS_MODEL = "TransE"


class TransH(TransE):
    """transh.py:20-145.  project() multiplies by the SUM of the relation's norm vector element-wise (transh.py:74-75),
    so the projected difference is (h - t) * (1 - sum(w) w); TripletMarginLoss over cat(rec, kg); no kg-side predictors."""

    MODEL = "TransH"
    GEMM_FORM = None  # not an L2 distance of (q, t) / not a fold of (h, r): the gather kernels
    RELATION_TABLES = ("relation_embedding", "norm_vec")

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        if self.ui_relation != self.n_relations - 1:
            # the reference reads relation_embedding.weight[-1] and norm_vec(ui_relation): one row id only when
            # [UI-Relation] is the last relation, which kg_dataset.py:626-631 guarantees
            raise ValueError("TransH needs [UI-Relation] to be the last relation id")

    def predict_kg(self, interaction):
        raise NotImplementedError

    def full_sort_predict_kg(self, interaction):
        raise NotImplementedError


class TransD(TransE):
    """transd.py:20-217: p_r(e) = r_vec (e . e_vec) + e; TripletMarginLoss over cat(rec, kg)."""

    MODEL = "TransD"
    GEMM_FORM = None  # not an L2 distance of (q, t) / not a fold of (h, r): the gather kernels
    USER_TABLES = ("user_embedding", "user_vec_embedding")
    ENTITY_TABLES = ("entity_embedding", "entity_vec_embedding")
    RELATION_TABLES = ("relation_embedding", "relation_vec_embedding")

    def full_sort_predict_kg(self, interaction):
        # transd.py:200-201 reads the head's projection vector from entity_embedding (not entity_vec_embedding)
        ent = self.entity_embedding.weight
        _, rel, tail = self._tables("kg")
        return ops.kge_full_sort(self.MODEL, (ent, ent), rel, tail, interaction[self.HEAD_ENTITY_ID],
                                 interaction[self.RELATION_ID], 0, 0, self.n_entities, self.margin)

    def full_sort_topk(self, interaction, k, history_index=None, kg=False):
        if not kg:
            return super().full_sort_topk(interaction, k, history_index, False)
        ent = self.entity_embedding.weight
        _, rel, tail = self._tables("kg")
        return ops.kge_full_sort_topk(self.MODEL, (ent, ent), rel, tail, interaction[self.HEAD_ENTITY_ID],
                                      interaction[self.RELATION_ID], 0, 0, self.n_entities, k, history_index, True,
                                      self.margin)


class Analogy(_GatherKGE):
    """analogy.py:20-302: DistMult on (e) plus the complex product on (re, im), every table embedding_size wide
    (scalar_share only sizes two unused attributes); LogisticLoss on the negated scores, rec + kg."""

    MODEL = "Analogy"
    USER_TABLES = ("user_embedding", "user_re_embedding", "user_im_embedding")
    ENTITY_TABLES = ("entity_embedding", "entity_re_embedding", "entity_im_embedding")
    RELATION_TABLES = ("relation_embedding", "relation_re_embedding", "relation_im_embedding")

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        self.margin = 0.0
        self.scalar_share = _cfg(config, "scalar_share", 0.5)
        self.scalar_dim = int(self.embedding_size * self.scalar_share)
        self.complex_dim = int(self.embedding_size - self.scalar_dim)

    def _rec_rel_full_sort(self):  # relation_*_embedding.weight[-1]  (analogy.py:222-224)
        return self.n_relations - 1

    GEMM_FORM = "bilinear"  # query (Scorer::fold): analogy.py:66-70, DistMult on (e) + the complex product on (re, im)

    def calculate_loss(self, interaction):
        # LogisticLoss(-pos, -neg) = mean softplus(-pos) + mean softplus(neg) (loss.py:172-190): twice the BCE-with-logits
        # mean over cat(pos, neg) with labels (1, 0); rec and kg separately (analogy.py:196-198)
        if self.fused_loss:
            return self._fused_pair_loss(interaction, "bce", scale=2.0)
        rec, kg = self._pair_scores(interaction)
        return 2.0 * (ops.bce_logits_mean(rec, rec.shape[1]) + ops.bce_logits_mean(kg, kg.shape[1]))


class HolE(TransE):
    """hole.py:20-160: (h circular-convolved with r) . t through the rolling matrix R[i][j] = r[(j - i) mod d];
    MarginRankingLoss(sigmoid(pos), sigmoid(neg), 1) over cat(rec, kg)."""

    MODEL = "HolE"
    GEMM_FORM = None  # not an L2 distance of (q, t) / not a fold of (h, r): the gather kernels

    def calculate_loss(self, interaction):
        if self.fused_loss:
            return self._fused_pair_loss(interaction, "margin_sigmoid")
        rec, kg = self._pair_scores(interaction)
        pos = torch.cat([rec[0], kg[0]])
        neg = torch.cat([rec[1], kg[1]])
        return ops.margin_ranking_mean(pos, neg, self.margin, sigmoid=True)


class TransR(KnowledgeRecommender):
    """transr.py:20-166: entities are mapped by the relation's d x d matrix (``proj_mat_e`` rows) and scored like TransE
    in that space; TripletMarginLoss over cat(rec, kg).  The projections run on csrc/kge_transr.hip (the only d^2 part),
    the distances on the TransE gather kernels with the projected buffers as tables.  The reference defines predict,
    predict_kg and full_sort_predict (no full_sort_predict_kg)."""

    input_type = InputType.PAIRWISE

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        self.embedding_size = config["embedding_size"]
        self.margin = float(_cfg(config, "margin", 0.0))
        self.ui_relation = dataset.field2token_id["relation_id"][dataset.ui_relation]
        d = self.embedding_size
        self.user_embedding = nn.Embedding(self.n_users, d)
        self.entity_embedding = nn.Embedding(self.n_entities, d)
        self.relation_embedding = nn.Embedding(self.n_relations, d)
        self.proj_mat_e = nn.Embedding(self.n_relations, d * d)
        self.apply(xavier_normal_initialization)

    def _distances(self, head_table, h_idx, r_idx, proj_rel, tails, eps):
        """scores [n, T] = -||M h + r - M t_j + eps|| for tails [n, T]; r_idx None = the rec side
        (translation row relation_embedding.weight[-1], projection matrix of ui_relation)."""
        n, T = tails.shape
        mats = self.proj_mat_e.weight
        if r_idx is None:
            ph = ops.project_rows(mats, head_table, None, proj_rel, h_idx)
            pt = ops.project_rows(mats, self.entity_embedding.weight, None, proj_rel, tails.t().reshape(-1))
        else:
            ph = ops.project_rows(mats, head_table, r_idx, 0, h_idx)
            pt = ops.project_rows(mats, self.entity_embedding.weight, r_idx.repeat(T), 0, tails.t().reshape(-1))
        rows = torch.arange(n, device=tails.device)
        t_rows = (rows.view(1, n) + n * torch.arange(T, device=tails.device).view(T, 1)).t()  # [n, T] view, j-major
        return ops.kge_score_grouped("TransE", (ph,), (self.relation_embedding.weight,), (pt,), rows, r_idx,
                                     self.n_relations - 1, t_rows, 0.0, eps, True)

    def forward(self, *rows):
        raise NotImplementedError("hopwise_amd scorers gather inside the kernel; use predict / calculate_loss")

    def calculate_loss(self, interaction):  # transr.py:72-104
        rec = self._distances(self.user_embedding.weight, interaction[self.USER_ID], None, self.ui_relation,
                              torch.stack([interaction[self.ITEM_ID], interaction[self.NEG_ITEM_ID]], 1),
                              ops.PAIRWISE_EPS)
        kg = self._distances(self.entity_embedding.weight, interaction[self.HEAD_ENTITY_ID], interaction[self.RELATION_ID],
                             0, torch.stack([interaction[self.TAIL_ENTITY_ID], interaction[self.NEG_TAIL_ENTITY_ID]], 1),
                             ops.PAIRWISE_EPS)
        pos = torch.cat([rec[:, 0], kg[:, 0]])
        neg = torch.cat([rec[:, 1], kg[:, 1]])
        return ops.margin_ranking_mean(pos, neg, self.margin)

    def predict(self, interaction):  # transr.py:106-122
        return self._distances(self.user_embedding.weight, interaction[self.USER_ID], None, self.ui_relation,
                               interaction[self.ITEM_ID].view(-1, 1), 0.0).view(-1)

    def predict_kg(self, interaction):  # transr.py:124-138
        return self._distances(self.entity_embedding.weight, interaction[self.HEAD_ENTITY_ID],
                               interaction[self.RELATION_ID], 0, interaction[self.TAIL_ENTITY_ID].view(-1, 1), 0.0).view(-1)

    def _projected_items(self):
        d = self.embedding_size
        m_ui = self.proj_mat_e.weight[self.ui_relation].view(d, d)
        return ops.gemm(self.entity_embedding.weight[: self.n_items], m_ui)  # [n_items, d] = E M^T

    def full_sort_predict(self, interaction):  # transr.py:140-166
        user = interaction[self.USER_ID]
        ph = ops.project_rows(self.proj_mat_e.weight, self.user_embedding.weight, None, self.ui_relation, user)
        rows = torch.arange(user.numel(), device=user.device)
        return ops.kge_full_sort("TransE", (ph,), (self.relation_embedding.weight,), (self._projected_items(),), rows, None,
                                 self.n_relations - 1, 0, self.n_items, 0.0)

    def full_sort_topk(self, interaction, k, history_index=None, kg=False):
        if kg:
            raise NotImplementedError  # the reference has no full_sort_predict_kg for TransR
        user = interaction[self.USER_ID]
        ph = ops.project_rows(self.proj_mat_e.weight, self.user_embedding.weight, None, self.ui_relation, user)
        rows = torch.arange(user.numel(), device=user.device)
        return ops.kge_full_sort_topk("TransE", (ph,), (self.relation_embedding.weight,), (self._projected_items(),), rows,
                                      None, self.n_relations - 1, 0, self.n_items, k, history_index, True, 0.0)


class TuckER(ops.DropoutCounters, KnowledgeRecommender):
    """TuckER 1-vs-all scorer (tucker.py:21-172) on the fp32 matrix cores.

    Same parameters / state_dict keys as the reference (``user_embedding`` has n_users + n_items rows, ``weights`` is
    the [d, d, d] core, ``bn0``/``bn1``), same config keys (input_dropout, input_dropout1, input_dropout2,
    label_smoothing), same quirks: the training target marks the UNION of the batch's positives in every row
    (tucker.py:111-115), predict() indexes the user-table logits with the raw item id (tucker.py:139), there is no
    full_sort_predict_kg (tucker.py:164-172 is commented out).
    Dense products (r . W, x . E^T and their gradients) run on hw_gemm_f32; the [B, d, d] contraction with
    hidden_dropout1 and the sigmoid + BCE with its gradient are fused kernels (csrc/tucker.hip); BatchNorm1d and the two
    [B, d] dropouts run on csrc/batchnorm.hip (ops.BatchNorm1d / ops.Dropout, same state_dict keys)."""

    # Philox offsets of hidden_dropout1 and of the two [B, d] dropouts: saved with the checkpoint (trainer.py:296-304)
    other_parameter_name = ["_drop_calls", "_dropout_calls"]

    input_type = InputType.PAIRWISE

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        d = self.embedding_size = config["embedding_size"]
        self.ui_relation = dataset.field2token_id["relation_id"][dataset.ui_relation]
        self.label_smoothing = config["label_smoothing"]
        self.user_embedding = nn.Embedding(self.n_users + self.n_items, d)
        self.entity_embedding = nn.Embedding(self.n_entities, d)
        self.relation_embedding = nn.Embedding(self.n_relations, d)
        self.weights = nn.Parameter(torch.empty(d, d, d).uniform_(-1, 1))
        seed = _cfg(config, "seed", 2020)
        self.input_dropout = ops.Dropout(config["input_dropout"], seed, stream=0)
        self.hidden_dropout1 = nn.Dropout(config["input_dropout1"])  # only its p: the mask is drawn inside the contraction
        self.hidden_dropout2 = ops.Dropout(config["input_dropout2"], seed, stream=1)
        self.bn0 = ops.BatchNorm1d(d)
        self.bn1 = ops.BatchNorm1d(d)
        self.apply(xavier_normal_initialization)
        self.forced_masks = None  # tests: {"rec": (m0 [B,d], m1 [B,d,d], m2 [B,d]), "kg": (...)} multiplicative masks
        # False: the per-row mm(r, W) of the reference; True (default): one matrix per distinct relation (see _features)
        self.share_relation_matrices = bool(_cfg(config, "share_relation_matrices", True))
        self._seed = seed_for_rank(_cfg(config, "seed", 2020))
        self._philox = PhiloxCounter()  # hidden_dropout1's offsets (one per contraction); `_drop_calls` in the checkpoint

    @property
    def _drop_calls(self):
        return self._philox.calls

    @_drop_calls.setter
    def _drop_calls(self, v):
        self._philox.calls = int(v)

    def _drop(self, module, x, forced):
        if forced is not None:
            return x * forced
        return module(x)

    def logits(self, h_idx, r_idx, table, side):
        """Pre-sigmoid scores [B, rows(table)] of forward() (tucker.py:69-84)."""
        return ops.matmul_nt(self._features(h_idx, r_idx, side), table)

    def _features(self, h_idx, r_idx, side):
        """forward() up to the product with the candidate table (tucker.py:69-82)."""
        d = self.embedding_size
        fm = self.forced_masks[side] if self.forced_masks is not None else (None, None, None)
        h = ops.embedding(self.user_embedding.weight if side == "rec" else self.entity_embedding.weight, h_idx)
        x = self._drop(self.input_dropout, self.bn0(h), fm[0])
        n_rel = self.relation_embedding.weight.shape[0]
        # the relation-set form holds n_relations [d, d] matrices (and their gradient): only while that is no more than the
        # reference's own [B, d, d] -- a KG with thousands of relations and a small batch takes the per-row product below
        if self.share_relation_matrices and h_idx.numel() > 0 and (side == "rec" or n_rel <= h_idx.numel()):
            # mm(r, W) has one distinct row per distinct relation: ONE on the rec side (every row carries [UI-Relation],
            # self._ui) and n_relations at most on the kg side, against B = 2048 rows.  Form those few [d, d] matrices once
            # ([n_rel, d] x [d, d^2] instead of [B, d] x [d, d^2], forward and twice backward) and let each batch row pick
            # its matrix; the per-row dropout masks (explicit or Philox) are untouched, so the result is the reference's.
            if side == "rec":
                rows, index = ops.embedding(self.relation_embedding.weight, r_idx[:1]), None
            else:
                # the contraction kernels index the matrix set with this id unchecked: clamp it (one tiny launch) the way
                # the embedding gather clamps a bad relation id on the per-row path
                rows, index = self.relation_embedding.weight, r_idx.clamp(0, n_rel - 1)
            w_set = ops.matmul_nn(rows, self.weights.view(d, d * d)).view(-1, d, d)
            if fm[1] is None and self.training and self.hidden_dropout1.p > 0:
                off = self._philox.offset(self._philox.next())
                x = ops.tucker_contract_shared(x, w_set, index, None, self.hidden_dropout1.p, self._seed, off)
            else:
                x = ops.tucker_contract_shared(x, w_set, index, fm[1])
            return self._drop(self.hidden_dropout2, self.bn1(x), fm[2])
        r = ops.embedding(self.relation_embedding.weight, r_idx)
        w_mat = ops.matmul_nn(r, self.weights.view(d, d * d)).view(-1, d, d)
        if fm[1] is None and self.training and self.hidden_dropout1.p > 0:
            # hidden_dropout1 over the [B, d, d] tensor is drawn inside the contraction (Philox), never materialised
            off = self._philox.offset(self._philox.next())
            x = ops.tucker_contract_dropout(x, w_mat, self.hidden_dropout1.p, self._seed, off)
        else:
            x = ops.tucker_contract(x, w_mat, fm[1])
        return self._drop(self.hidden_dropout2, self.bn1(x), fm[2])

    def forward(self, h_idx, r_idx, table, side):
        return ops.sigmoid(self.logits(h_idx, r_idx, table, side))

    def _ui(self, user):
        return torch.full_like(user, self.ui_relation)

    def calculate_loss(self, interaction):
        user, item = interaction[self.USER_ID], interaction[self.ITEM_ID]
        head, rel, tail = interaction[self.HEAD_ENTITY_ID], interaction[self.RELATION_ID], interaction[self.TAIL_ENTITY_ID]
        dev = user.device
        item_flag = torch.zeros(self.n_users + self.n_items, dtype=torch.uint8, device=dev)
        # index_fill_, not `flag[idx] = 1`: index_put_ stages its scalar through the host, which a captured step may not do
        item_flag.index_fill_(0, item + self.n_users, 1)
        tail_flag = torch.zeros(self.n_entities, dtype=torch.uint8, device=dev)
        tail_flag.index_fill_(0, tail, 1)
        if self.label_smoothing:
            s_rec, a_rec = 1.0 - self.label_smoothing, 1.0 / self.n_items
            s_kg, a_kg = 1.0 - self.label_smoothing, 1.0 / self.n_entities
        else:
            s_rec = s_kg = 1.0
            a_rec = a_kg = 0.0
        # the kg forward runs first in the reference (tucker.py:108-109 gather order does not matter; BN running
        # statistics are updated by score_users first, then score_kg -- tucker.py:121-122)
        xu = self._features(user, self._ui(user), "rec")
        xk = self._features(head, rel, "kg")
        return (ops.score_all_bce_mean(xu, self.user_embedding.weight, item_flag, s_rec, a_rec)
                + ops.score_all_bce_mean(xk, self.entity_embedding.weight, tail_flag, s_kg, a_kg))

    def predict(self, interaction):
        user, item = interaction[self.USER_ID], interaction[self.ITEM_ID]
        score = self.forward(user, self._ui(user), self.user_embedding.weight, "rec")
        return score[torch.arange(user.size(0), device=user.device), item]

    def predict_kg(self, interaction):
        head, rel, tail = interaction[self.HEAD_ENTITY_ID], interaction[self.RELATION_ID], interaction[self.TAIL_ENTITY_ID]
        score = self.forward(head, rel, self.entity_embedding.weight, "kg")
        return score[torch.arange(head.size(0), device=head.device), tail]

    def full_sort_predict(self, interaction):
        user = interaction[self.USER_ID]
        score = self.forward(user, self._ui(user), self.user_embedding.weight, "rec")
        return score[:, self.n_users :]


class _PadEmbedding(torch.autograd.Function):
    """table[idx] whose backward skips ``padding_idx`` (nn.Embedding(padding_idx=0) semantics, conve.py:52-55)."""

    @staticmethod
    def forward(ctx, table, idx, padding_idx):
        ctx.save_for_backward(idx)
        ctx.shape, ctx.pad = table.shape, padding_idx
        return ops.gather_rows(table, idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        gt = torch.zeros(ctx.shape, dtype=torch.float32, device=g.device)
        keep = idx != ctx.pad
        ops.scatter_add_rows_(gt, idx[keep].contiguous(), g[keep].contiguous())
        return gt, None, None


class ConvE(ops.DropoutCounters, KnowledgeRecommender):
    """ConvE 1-vs-all scorer (conve.py:21-190): the head and relation rows are reshaped to [d1, d2] images, stacked,
    convolved with 32 3x3 filters, projected back to d and multiplied with every row of the entity table.

    Same parameters / state_dict keys as the reference (``user_embedding`` has n_users + n_items rows, ``relations_embeddings``,
    ``conv1``, ``bn0/bn1/bn2``, ``b_users``, ``b_entities``, ``fc``), config keys (embedding_shape, hidden_size, use_bias,
    input_dropout, hidden_dropout, feature_dropout, label_smoothing) and quirks: union-of-the-batch targets
    (conve.py:124-128), label smoothing with 1 / n_items and 1 / n_entities (:131-132), predict_kg returns
    ``score[:, tail]`` -- a [B, B] matrix (:162), full_sort_predict_kg reads attributes that do not exist (:183-186) and
    therefore raises AttributeError, lookups skip the gradient of id 0 (padding_idx) while the 1-vs-all product does not.
    The convolution (csrc/conv.hip), the two BatchNorm2d layers (csrc/batchnorm.hip), the two dense products and sigmoid +
    BCE are HIP, and so are BatchNorm1d and the element-wise dropouts (ops.BatchNorm1d / ops.Dropout); only the channel-wise
    Dropout2d is torch's."""

    other_parameter_name = ["_dropout_calls"]  # mask streams of inp_drop / hidden_drop resume with the checkpoint

    input_type = InputType.PAIRWISE

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        d = self.embedding_size = config["embedding_size"]
        self.label_smoothing = config["label_smoothing"]
        self.embedding_dim1 = config["embedding_shape"]
        self.embedding_dim2 = d // self.embedding_dim1
        self.hidden_size = config["hidden_size"]
        self.use_bias = config["use_bias"]
        self.ui_relation = dataset.field2token_id["relation_id"][dataset.ui_relation]
        self.user_embedding = nn.Embedding(self.n_users + self.n_items, d, padding_idx=0)
        self.entity_embedding = nn.Embedding(self.n_entities, d, padding_idx=0)
        self.relations_embeddings = nn.Embedding(self.n_relations, d, padding_idx=0)
        seed = _cfg(config, "seed", 2020)
        self.inp_drop = ops.Dropout(config["input_dropout"], seed, stream=0)
        self.hidden_drop = ops.Dropout(config["hidden_dropout"], seed, stream=1)
        self.feature_map_drop = nn.Dropout2d(config["feature_dropout"])
        self.conv1 = nn.Conv2d(1, 32, (3, 3), 1, 0, bias=self.use_bias)
        self.bn0 = ops.BatchNorm2d(1)
        self.bn1 = ops.BatchNorm2d(32)
        self.bn2 = ops.BatchNorm1d(d)
        self.register_parameter("b_users", nn.Parameter(torch.zeros(self.n_users + self.n_items)))
        self.register_parameter("b_entities", nn.Parameter(torch.zeros(self.n_entities)))
        self.fc = nn.Linear(self.hidden_size, d)
        self.apply(xavier_normal_initialization)
        self.forced_masks = None  # tests: {"rec": (m_inp, m_feature, m_hidden), "kg": (...)} multiplicative masks

    def _drop(self, module, x, forced):
        return x * forced if forced is not None else module(x)

    def logits(self, h_idx, r_idx, head_table, out_table, bias, side):
        """Pre-sigmoid scores [B, rows(out_table)] of forward() (conve.py:75-93)."""
        return ops.linear_act(self._features(h_idx, r_idx, head_table, side), out_table, bias)

    def _features(self, h_idx, r_idx, head_table, side):
        """forward() up to the product with the candidate table (conve.py:75-90)."""
        fm = self.forced_masks[side] if self.forced_masks is not None else (None, None, None)
        shape = (-1, 1, self.embedding_dim1, self.embedding_dim2)
        h = _PadEmbedding.apply(head_table, h_idx, 0).view(shape)
        r = _PadEmbedding.apply(self.relations_embeddings.weight, r_idx, 0).view(shape)
        x = self.bn0(torch.cat([h, r], 2))
        x = self._drop(self.inp_drop, x, fm[0])
        x = ops.conv2d_1ch(x, self.conv1.weight, self.conv1.bias)
        x = torch.relu(self.bn1(x))
        x = self._drop(self.feature_map_drop, x, fm[1])
        x = ops.linear_act(x.reshape(x.shape[0], -1), self.fc.weight, self.fc.bias)
        x = self._drop(self.hidden_drop, x, fm[2])
        return torch.relu(self.bn2(x))

    def forward(self, h_idx, r_idx, head_table, out_table, bias, side):
        return ops.sigmoid(self.logits(h_idx, r_idx, head_table, out_table, bias, side))

    def _rec(self, user):
        return (user, torch.full_like(user, self.ui_relation), self.user_embedding.weight, self.user_embedding.weight,
                self.b_users, "rec")

    def _kg(self, head, rel):
        return head, rel, self.entity_embedding.weight, self.entity_embedding.weight, self.b_entities, "kg"

    def calculate_loss(self, interaction):  # conve.py:110-137
        user, item = interaction[self.USER_ID], interaction[self.ITEM_ID]
        head, rel, tail = interaction[self.HEAD_ENTITY_ID], interaction[self.RELATION_ID], interaction[self.TAIL_ENTITY_ID]
        dev = user.device
        item_flag = torch.zeros(self.n_users + self.n_items, dtype=torch.uint8, device=dev)
        # index_fill_, not `flag[idx] = 1`: index_put_ stages its scalar through the host, which a captured step may not do
        item_flag.index_fill_(0, item + self.n_users, 1)
        tail_flag = torch.zeros(self.n_entities, dtype=torch.uint8, device=dev)
        tail_flag.index_fill_(0, tail, 1)
        if self.label_smoothing:
            s, a_rec, a_kg = 1.0 - self.label_smoothing, 1.0 / self.n_items, 1.0 / self.n_entities
        else:
            s, a_rec, a_kg = 1.0, 0.0, 0.0
        uh, ur, utab, uout, ub, _ = self._rec(user)
        kh, kr, ktab, kout, kb, _ = self._kg(head, rel)
        xu = self._features(uh, ur, utab, "rec")
        xk = self._features(kh, kr, ktab, "kg")
        return (ops.score_all_bce_mean(xu, uout, item_flag, s, a_rec, bias=ub)
                + ops.score_all_bce_mean(xk, kout, tail_flag, s, a_kg, bias=kb))

    def predict(self, interaction):  # conve.py:139-151
        user, item = interaction[self.USER_ID], interaction[self.ITEM_ID]
        score = self.forward(*self._rec(user))[:, self.n_users :]
        return score[torch.arange(user.size(0), device=user.device), item]

    def predict_kg(self, interaction):  # conve.py:153-164: ``score[:, tail]`` -> [B, len(tail)]
        score = self.forward(*self._kg(interaction[self.HEAD_ENTITY_ID], interaction[self.RELATION_ID]))
        return score[:, interaction[self.TAIL_ENTITY_ID]]

    def full_sort_predict(self, interaction):  # conve.py:166-176
        return self.forward(*self._rec(interaction[self.USER_ID]))[:, self.n_users :]

    def full_sort_predict_kg(self, interaction):  # conve.py:178-190 touches self.entities_embeddings, which is never set
        raise AttributeError("'ConvE' object has no attribute 'entities_embeddings'")


class ConvKB(ops.DropoutCounters, KnowledgeRecommender):
    """ConvKB (convkb.py:24-160): the (h, r, t) rows form a [d, 3] matrix that goes through BatchNorm2d(1), T filters of
    shape (kernel_size, 3), BatchNorm2d(T), ReLU, dropout and a bias-free Linear to one score.

    Same parameters / state_dict keys (``conv1_bn``, ``conv_layer``, ``conv2_bn``, ``fc_layer``), config keys
    (out_channels, kernel_size, dropout_prob, lambda) and loss: LogisticLoss on the negated scores with the
    regulariser lambda * (mean h^2 + mean t^2 + mean r^2 + ||conv W|| + ||conv b|| + ||fc W||) added to both terms,
    rec + kg; four forward passes per step (pos / neg x rec / kg), each with its own BatchNorm batch statistics, in the
    reference's order (convkb.py:131-134).  The convolution and the Linear run on csrc/conv.hip and the MFMA GEMM, the
    softplus means on the BCE kernel, the BatchNorm2d layers on csrc/batchnorm.hip; dropout / the tiny norms are torch.  No full-sort methods exist in
    the reference."""

    other_parameter_name = ["_dropout_calls"]

    input_type = InputType.PAIRWISE

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        d = self.embedding_size = config["embedding_size"]
        self.out_channels = config["out_channels"]
        self.kernel_size = config["kernel_size"]
        self.drop_prob = config["dropout_prob"]
        self.lmbda = config["lambda"]
        self.ui_relation = dataset.field2token_id["relation_id"][dataset.ui_relation]
        self.user_embedding = nn.Embedding(self.n_users, d)
        self.entity_embedding = nn.Embedding(self.n_entities, d)
        self.relation_embedding = nn.Embedding(self.n_relations, d)
        self.conv1_bn = ops.BatchNorm2d(1)
        self.conv_layer = nn.Conv2d(1, self.out_channels, (self.kernel_size, 3))
        self.conv2_bn = ops.BatchNorm2d(self.out_channels)
        self.dropout = ops.Dropout(self.drop_prob, _cfg(config, "seed", 2020), stream=0)
        self.fc_layer = nn.Linear((d - self.kernel_size + 1) * self.out_channels, 1, bias=False)
        self.apply(xavier_normal_initialization)
        self.forced_masks = None  # tests: list of multiplicative dropout masks, one per forward call

    def forward(self, head, relation, tail):
        """Scores of gathered rows [B, d] x 3 (convkb.py:65-82)."""
        x = torch.stack([head, relation, tail], dim=2).unsqueeze(1)  # [B, 1, d, 3]
        x = self.conv1_bn(x)
        x = ops.conv2d_1ch(x, self.conv_layer.weight, self.conv_layer.bias)
        x = torch.relu(self.conv2_bn(x))
        x = x.reshape(x.shape[0], -1)
        if self.forced_masks is not None:
            x = x * self.forced_masks.pop(0)
        else:
            x = self.dropout(x)
        return ops.linear_act(x, self.fc_layer.weight, None).view(-1)

    def _reg(self, h, r, t):  # convkb.py:84-88
        l2 = torch.mean(h**2) + torch.mean(t**2) + torch.mean(r**2)
        for w in (*self.conv_layer.parameters(), *self.fc_layer.parameters()):
            l2 = l2 + w.norm(2)
        return self.lmbda * l2

    def _rows(self, table, idx):
        return ops.embedding(table.weight, idx)

    def _logistic(self, pos, neg, pos_reg, neg_reg):
        # LogisticLoss(-pos, -neg, regs) (loss.py:172-190): mean softplus(-pos) + mean softplus(neg) (+ regs when both are
        # non-zero) = twice the BCE-with-logits mean over cat(pos, neg) with labels (1, 0)
        loss = 2.0 * ops.bce_logits_mean(torch.stack([pos, neg]), pos.shape[0])
        if bool(pos_reg) and bool(neg_reg):
            loss = loss + pos_reg + neg_reg
        return loss

    def calculate_loss(self, interaction):  # convkb.py:106-141
        ui = torch.full_like(interaction[self.USER_ID], self.ui_relation)
        u = self._rows(self.user_embedding, interaction[self.USER_ID])
        ur = self._rows(self.relation_embedding, ui)
        ip = self._rows(self.entity_embedding, interaction[self.ITEM_ID])
        ineg = self._rows(self.entity_embedding, interaction[self.NEG_ITEM_ID])
        h = self._rows(self.entity_embedding, interaction[self.HEAD_ENTITY_ID])
        r = self._rows(self.relation_embedding, interaction[self.RELATION_ID])
        tp = self._rows(self.entity_embedding, interaction[self.TAIL_ENTITY_ID])
        tn = self._rows(self.entity_embedding, interaction[self.NEG_TAIL_ENTITY_ID])
        s_pu, s_nu = self.forward(u, ur, ip), self.forward(u, ur, ineg)
        s_pk, s_nk = self.forward(h, r, tp), self.forward(h, r, tn)
        rec = self._logistic(s_pu, s_nu, self._reg(u, ur, ip), self._reg(u, ur, ineg))
        kg = self._logistic(s_pk, s_nk, self._reg(h, r, tp), self._reg(h, r, tn))
        return rec + kg

    def predict(self, interaction):
        users = interaction[self.USER_ID]
        return self.forward(self._rows(self.user_embedding, users),
                            self._rows(self.relation_embedding, torch.full_like(users, self.ui_relation)),
                            self._rows(self.entity_embedding, interaction[self.ITEM_ID]))

    def predict_kg(self, interaction):
        return self.forward(self._rows(self.entity_embedding, interaction[self.HEAD_ENTITY_ID]),
                            self._rows(self.relation_embedding, interaction[self.RELATION_ID]),
                            self._rows(self.entity_embedding, interaction[self.TAIL_ENTITY_ID]))
