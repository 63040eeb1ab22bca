"""TPRec (time-aware path reasoning) on the PGPR kernels.

Reference: /root/reference/hopwise/model/knowledge_aware_recommender/tprec.py (class TPRec :24-771).

TPRec is PGPR with the [UI-Relation] vector replaced, per user, by the temporal mixture
``V[u] = sum_c uc_weight[u][c] * ui_clust_relation_embedding[c]`` wherever PGPR adds ``relation_embedding[ui]`` to the
user vector: the top-``max_acts`` trim of ``_get_actions`` (tprec.py:450-471) and the soft reward (:540-552); its two
stages are a TransE-style pretraining of the cluster relation table (:295-334) and PGPR's REINFORCE policy training.
The device path is therefore PGPR's (csrc/pgpr.hip, pgpr_train.hip) with ``hw_pgpr_user_scores_tp``.

Kept as written in the reference:
  * ``_build_interacted_matrix`` (:621-632) never resets its accumulator and appends the SAME array for every user, so
    ``collect_scores`` adds one constant vector (the sum over all users and clusters) to every user but the pad row;
  * ``u_p_scales`` (:146-149) use the plain ``relation_embedding[ui]``, not the temporal vector;
  * ``expand_paths`` (:168-197) turns the patterns into lists without the leading "self_loop" while ``_has_pattern``
    (:387-389) looks up a tuple that starts with it, so no path ever matches and every reward is 0;
  * ``calculate_loss_transe`` slices the batch positionally ``[first index of cluster, + its count)`` (np.unique,
    :301,316-325) -- the slices follow batch order, not cluster membership -- and raises when a batch holds a single
    cluster (:313, :328: a 1-D row is concatenated with a 2-D tensor);
  * pretraining freezes the three TransE tables (``nn.Embedding.from_pretrained`` default) -- only the cluster table trains.
The temporal clustering itself (TPRecTimestampDataset) is dataset preprocessing and out of scope: ``dataset.temporal_weights``
must provide ``uc_weight``, ``timenum`` and ``timeClassifyLabel`` as the reference's does.
"""

from __future__ import annotations

import os

import numpy as np
import torch
from torch import nn

from .. import ops
from .knowledge_aware_recommender import PGPR
from .knowledge_graph_embedding_recommender import _cfg


def temporal_vectors(uc_weight, clust_emb, n_users):
    """V[u] accumulated like the reference: float32 products summed in the dict's iteration order (tprec.py:454-460)."""
    c32 = np.asarray(clust_emb, np.float32)
    v = np.zeros((n_users, c32.shape[1]), np.float32)
    for u, d in uc_weight.items():
        acc = None
        for c, w in d.items():
            term = c32[c] * w
            acc = term if acc is None else acc + term
        if acc is not None and 0 <= int(u) < n_users:
            v[int(u)] = acc
    return v


def interacted_constant(uc_weight, clust_emb):
    """The one vector _build_interacted_matrix ends up appending for every user (tprec.py:621-632)."""
    c32 = np.asarray(clust_emb, np.float32)
    acc = None
    for uid in range(1, len(uc_weight) + 1):
        for c, w in uc_weight[uid].items():
            term = c32[c] * w
            if acc is None:
                acc = term.copy()
            else:
                acc += term
    return acc


class _PretrainDataset:
    """What PGPR.__init__ reads, with the tables taken from a pretraining checkpoint instead of the dataset."""

    def __init__(self, dataset, tables):
        self._ds, self._tables = dataset, tables

    def __getattr__(self, k):
        return getattr(self._ds, k)

    def get_preload_weight(self, name):
        return self._tables[name]


class TPRec(PGPR):
    def __init__(self, config, dataset):
        self.train_stage = _cfg(config, "train_stage", "policy")
        tw = dataset.temporal_weights
        self.uc_weight, self.timenum, self.ui2label_dict = tw.uc_weight, tw.timenum, tw.timeClassifyLabel
        self.n_clusters = int(config["cluster_num"])
        if self.train_stage == "pretrain":
            nn.Module.__init__(self)
            self._init_pretrain(config, dataset)
            return
        weights = _cfg(config, "pretrained_weights", None)
        if weights is None:  # tprec.py:199-208
            path = os.path.join(config["checkpoint_dir"], "{}-{}-pretrained.pth".format(config["model"], config["dataset"]))
            weights = torch.load(path, weights_only=False, map_location="cpu")["state_dict"]
        tabs = {"user_embedding_id": weights["user_embedding.weight"].cpu().numpy(),
                "entity_embedding_id": weights["entity_embedding.weight"].cpu().numpy(),
                "relation_embedding_id": weights["relation_embedding.weight"].cpu().numpy()}
        super().__init__(config, _PretrainDataset(dataset, tabs))
        self.ui_clust_relation_embedding = weights["ui_clust_relation_embedding.weight"].cpu().numpy()
        dev = self._t64[0].device
        v32 = temporal_vectors(self.uc_weight, self.ui_clust_relation_embedding, self.n_users)
        self._v64 = torch.from_numpy(v32.astype(np.float64)).to(dev).contiguous()
        # rewards: expand_paths (tprec.py:168-197) returns LISTS of relation names without the leading "self_loop", and
        # _has_pattern (:387-389) looks a TUPLE that starts with "self_loop" up in them: nothing ever matches, every reward
        # of the reference's policy stage is 0.  Reproduced by keeping no pattern at all.
        self._patterns = None
        self._interacted = None

    # ---- policy stage hooks ------------------------------------------------------------------------------------------
    def _user_vec(self):
        return self._v64

    def set_temporal_weights(self, temporal_weight):
        """The evaluation-time weights handed to full_sort_predict / explain (tprec.py:610-619)."""
        const = interacted_constant(temporal_weight.uc_weight, self.ui_clust_relation_embedding)
        r = self._t64[2].clone()
        r[self.ui_relation_id] = torch.from_numpy(const.astype(np.float64)).to(r.device)
        self._interacted = (self._t64[0], self._t64[1], r.contiguous())

    def _collect_tables(self):
        if self._interacted is None:
            raise RuntimeError("TPRec.collect_scores needs the evaluation temporal weights: call set_temporal_weights "
                               "or pass (interaction, temporal_weights) to full_sort_predict")
        return self._interacted

    def full_sort_predict(self, interaction):
        if self.train_stage == "pretrain":
            return self.full_sort_predict_transe(interaction)
        interaction, temporal_weight = interaction
        self.set_temporal_weights(temporal_weight)
        return super().full_sort_predict(interaction)

    def explain(self, interaction):
        users, temporal_weight = interaction[self.USER_ID]
        self.set_temporal_weights(temporal_weight)
        return super().explain({self.USER_ID: users})

    def predict(self, interaction):
        if self.train_stage == "pretrain":
            return self.predict_transe(interaction)
        return

    def calculate_loss(self, interaction):
        if self.train_stage == "pretrain":
            return self.calculate_loss_transe(interaction)
        return super().calculate_loss(interaction)

    # ---- pretrain stage (tprec.py:295-366) --------------------------------------------------------------------------------
    def _init_pretrain(self, config, dataset):
        self.device = config["device"]
        for attr, key in (("USER_ID", "USER_ID_FIELD"), ("ITEM_ID", "ITEM_ID_FIELD"), ("ENTITY_ID", "ENTITY_ID_FIELD"),
                          ("RELATION_ID", "RELATION_ID_FIELD"), ("HEAD_ENTITY_ID", "HEAD_ENTITY_ID_FIELD"),
                          ("TAIL_ENTITY_ID", "TAIL_ENTITY_ID_FIELD")):
            setattr(self, attr, config[key])
        self.NEG_ITEM_ID = config["NEG_PREFIX"] + self.ITEM_ID
        self.NEG_TAIL_ENTITY_ID = config["NEG_PREFIX"] + self.TAIL_ENTITY_ID
        self.n_users, self.n_items = dataset.num(self.USER_ID), dataset.num(self.ITEM_ID)
        self.n_entities, self.n_relations = dataset.num(self.ENTITY_ID), dataset.num(self.RELATION_ID)
        self.margin = float(config["margin"])
        dev = torch.device(self.device)
        for name, key in (("user_embedding", "user_embedding_id"), ("entity_embedding", "entity_embedding_id"),
                          ("relation_embedding", "relation_embedding_id")):
            w = torch.from_numpy(np.asarray(dataset.get_preload_weight(key)))  # float64, frozen (from_pretrained)
            setattr(self, name, nn.Embedding.from_pretrained(w))
        d = self.user_embedding.weight.shape[1]
        self.ui_clust_relation_embedding = nn.Embedding(self.n_clusters, d)
        self.to(dev)
        self._f32 = None

    def _frozen32(self):
        if self._f32 is None:
            self._f32 = tuple(t.weight.detach().to(torch.float32).contiguous()
                              for t in (self.user_embedding, self.entity_embedding, self.relation_embedding))
        return self._f32

    def calculate_loss_transe(self, interaction):
        U, E, R = self._frozen32()
        user, pos, neg = interaction[self.USER_ID], interaction[self.ITEM_ID], interaction[self.NEG_ITEM_ID]
        labels = [self.ui2label_dict[(int(u), int(i))] for u, i in zip(user.tolist(), pos.tolist())]
        num_id, num_index, num_n = np.unique(labels, return_index=True, return_counts=True)
        if len(num_id) == 1:
            raise RuntimeError("Tensors must have same number of dimensions: got 1 and 2")  # what tprec.py:313,328 does
        C = self.ui_clust_relation_embedding.weight
        loss = 0
        n = user.shape[0]
        for c, s, cnt in zip(num_id, num_index, num_n):
            e = min(int(s) + int(cnt), n)  # positional slice, clipped like Python slicing
            tails = torch.stack([pos[s:e], neg[s:e]], 1)
            sc = ops.kge_score_grouped("TransE", (U,), (C,), (E,), user[s:e].contiguous(), None, int(c), tails, 0.0,
                                       ops.PAIRWISE_EPS, True)
            loss = loss + ops.margin_ranking_mean(sc[:, 0], sc[:, 1], self.margin)
        with torch.no_grad():  # every table of the joint term is frozen: it only adds to the value
            rec = ops.kge_score_grouped("TransE", (U,), (R,), (E,), user, None, self.n_relations - 1,
                                        torch.stack([pos, neg], 1), 0.0, ops.PAIRWISE_EPS, True)
            kg = ops.kge_score_grouped("TransE", (E,), (R,), (E,), interaction[self.HEAD_ENTITY_ID],
                                       interaction[self.RELATION_ID], 0,
                                       torch.stack([interaction[self.TAIL_ENTITY_ID], interaction[self.NEG_TAIL_ENTITY_ID]], 1),
                                       0.0, ops.PAIRWISE_EPS, True)
            joint = ops.margin_ranking_mean(torch.cat([rec[:, 0], kg[:, 0]]), torch.cat([rec[:, 1], kg[:, 1]]), self.margin)
        return loss + joint

    def predict_transe(self, interaction):
        U, E, R = self._frozen32()
        return ops.kge_score_grouped("TransE", (U,), (R,), (E,), interaction[self.USER_ID], None, self.n_relations - 1,
                                     interaction[self.ITEM_ID].view(-1, 1), 0.0).view(-1)

    def full_sort_predict_transe(self, interaction):
        U, E, R = self._frozen32()
        return ops.kge_full_sort("TransE", (U,), (R,), (E,), interaction[self.USER_ID], None, self.n_relations - 1, 0,
                                 self.n_items, 0.0)
