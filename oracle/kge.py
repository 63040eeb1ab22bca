"""numpy restatement of the reference KGE scorers, losses and analytic gradients.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Pinned against the
reference by ``tests/golden/kge_*.npz`` (made by ``tests/golden/make_golden.py``).

Reference (paths relative to /root/reference/hopwise/model/knowledge_graph_embedding_recommender):
  TransE    transe.py:55-57 (forward), :75-98 (calculate_loss), :112-126 / :139-154 (full sorts)
  RotatE    rotate.py:61-69, :98-131, :161-190 / :192-220
  ComplEx   complex.py:53-62, :95-128, :162-190 / :192-219
  DistMult  distmult.py:50-51, :69-95, :119-133 / :135-146
Losses: torch.nn.TripletMarginLoss (TransE), BCEWithLogitsLoss (RotatE, ComplEx),
MarginRankingLoss (DistMult), all with reduction="mean".

Table naming follows the reference ``state_dict`` keys.  Every model is described
by three *slots* (head, relation, tail), each a tuple of one or two tables:

  model     head (rec side / kg side)                              relation                    tail
  TransE    (user_embedding,) / (entity_embedding,)                (relation_embedding,)       (entity_embedding,)
  DistMult  same as TransE
  RotatE    (user_embedding, user_embedding_im) /                  (relation_embedding,)       (entity_embedding, entity_embedding_im)
            (entity_embedding, entity_embedding_im)
  ComplEx   (user_re_embedding, user_im_embedding) /               (relation_re_embedding,     (entity_re_embedding, entity_im_embedding)
            (entity_re_embedding, entity_im_embedding)              relation_im_embedding)
"""

from __future__ import annotations

import numpy as np

from . import kge_ext

MODELS = ("TransE", "RotatE", "ComplEx", "DistMult")
DENSE_MODELS = ("RESCAL",)  # same table slots as TransE; relation rows are d x d matrices (rescal.py:43)
EXT_MODELS = kge_ext.EXT_MODELS  # TorusE, TransH, TransD, Analogy, HolE (oracle/kge_ext.py)
# which of predict_kg / full_sort_predict_kg each reference class defines
HAS_KG_PREDICT = {"TransH": (False, False), "TransR": (True, False)}

F32 = np.float32
PAIRWISE_EPS = 1e-6  # torch.nn.functional.pairwise_distance default eps (used by TripletMarginLoss)


# --------------------------------------------------------------------------------------
# table slots
# --------------------------------------------------------------------------------------
def slots(model: str, side: str):
    """Names of the (head, relation, tail) tables for ``side`` in {"rec", "kg"}."""
    if model in EXT_MODELS:
        return kge_ext.slots(model, side)
    if model in ("TransE", "DistMult", "RESCAL"):
        head = ("user_embedding",) if side == "rec" else ("entity_embedding",)
        return head, ("relation_embedding",), ("entity_embedding",)
    if model == "RotatE":
        head = ("user_embedding", "user_embedding_im") if side == "rec" else ("entity_embedding", "entity_embedding_im")
        return head, ("relation_embedding",), ("entity_embedding", "entity_embedding_im")
    if model == "ComplEx":
        head = (
            ("user_re_embedding", "user_im_embedding")
            if side == "rec"
            else ("entity_re_embedding", "entity_im_embedding")
        )
        return head, ("relation_re_embedding", "relation_im_embedding"), ("entity_re_embedding", "entity_im_embedding")
    raise ValueError(model)


def table_names(model: str):
    names = []
    for side in ("rec", "kg"):
        for slot in slots(model, side):
            for n in slot:
                if n not in names:
                    names.append(n)
    return names


# --------------------------------------------------------------------------------------
# scorers on gathered rows.  H, R, T are tuples of float32 arrays that broadcast together
# over leading axes; the embedding axis is last.
# --------------------------------------------------------------------------------------
def score_rows(model: str, H, R, T, margin: float = 0.0, eps: float = 0.0):
    """Score of (h, r, t) rows, reduced over the last axis.

    ``eps`` is only meaningful for TransE: the loss path measures
    ``||h + r - t + eps||`` (pairwise_distance), the scorer uses eps = 0.
    """
    if model in EXT_MODELS:
        return kge_ext.score_rows(model, H, R, T, margin, eps)
    if model == "TransE":  # transe.py:55-57
        d = (H[0] + R[0]) - T[0]
        if eps:
            d = d + F32(eps)
        return -np.sqrt(np.sum(d * d, axis=-1, dtype=F32))
    if model == "DistMult":  # distmult.py:50-51
        return np.sum(H[0] * R[0] * T[0], axis=-1, dtype=F32)
    if model == "RESCAL":  # rescal.py:51-54: (h^T M_r) . t with M_r = relation row viewed [d, d]
        d = H[0].shape[-1]
        M = R[0].reshape(R[0].shape[:-1] + (d, d))
        hr = np.einsum("...i,...ij->...j", H[0], M).astype(F32)
        return np.sum(hr * T[0], axis=-1, dtype=F32)
    if model == "RotatE":  # rotate.py:61-69
        rr, ri = np.cos(R[0]).astype(F32), np.sin(R[0]).astype(F32)
        re = (rr * H[0] - ri * H[1]) - T[0]
        im = (rr * H[1] + ri * H[0]) - T[1]
        s = np.sum(re * re, axis=-1, dtype=F32) + np.sum(im * im, axis=-1, dtype=F32)
        return F32(margin) - np.sqrt(s)
    if model == "ComplEx":  # complex.py:53-62 (note: 4th term uses t_im, as the reference does)
        hr, hi = H
        rr, ri = R
        tr, ti = T
        return (
            np.sum(hr * rr * tr, axis=-1, dtype=F32)
            + np.sum(hi * rr * ti, axis=-1, dtype=F32)
            + np.sum(hr * ri * ti, axis=-1, dtype=F32)
            - np.sum(hi * ri * ti, axis=-1, dtype=F32)
        )
    raise ValueError(model)


def score_rows_vjp(model: str, H, R, T, g, margin: float = 0.0, eps: float = 0.0):
    """Row gradients of ``sum(g * score_rows(...))`` w.r.t. every H/R/T row (float64 math).

    Returns (gH, gR, gT) tuples shaped like the inputs (after broadcasting to the
    score shape + embedding axis).
    """
    if model in EXT_MODELS:
        return kge_ext.score_rows_vjp(model, H, R, T, g, margin, eps)
    H = [np.asarray(x, np.float64) for x in H]
    R = [np.asarray(x, np.float64) for x in R]
    T = [np.asarray(x, np.float64) for x in T]
    g = np.asarray(g, np.float64)[..., None]
    if model == "TransE":
        d = H[0] + R[0] - T[0] + eps
        n = np.sqrt(np.sum(d * d, axis=-1, keepdims=True))
        with np.errstate(invalid="ignore", divide="ignore"):
            u = np.where(n > 0, d / n, 0.0)
        # score = -n
        return (-(g * u),), (-(g * u),), ((g * u),)
    if model == "DistMult":
        return (g * R[0] * T[0],), (g * H[0] * T[0],), (g * H[0] * R[0],)
    if model == "RESCAL":
        d = H[0].shape[-1]
        M = R[0].reshape(R[0].shape[:-1] + (d, d))
        hr = np.einsum("...i,...ij->...j", H[0], M)
        gh = g * np.einsum("...ij,...j->...i", M, T[0])
        gM = g[..., None] * H[0][..., :, None] * T[0][..., None, :]
        return (gh,), (gM.reshape(gM.shape[:-2] + (d * d,)),), (g * hr,)
    if model == "RotatE":
        rr, ri = np.cos(R[0]), np.sin(R[0])
        re = rr * H[0] - ri * H[1] - T[0]
        im = rr * H[1] + ri * H[0] - T[1]
        n = np.sqrt(np.sum(re * re + im * im, axis=-1, keepdims=True))
        with np.errstate(invalid="ignore", divide="ignore"):
            ure = np.where(n > 0, re / n, 0.0)
            uim = np.where(n > 0, im / n, 0.0)
        # score = margin - n ; dn/dre = ure, dn/dim = uim
        g_hre = -g * (ure * rr + uim * ri)
        g_him = -g * (-ure * ri + uim * rr)
        dre_dth = -ri * H[0] - rr * H[1]
        dim_dth = -ri * H[1] + rr * H[0]
        g_th = -g * (ure * dre_dth + uim * dim_dth)
        return (g_hre, g_him), (g_th,), (g * ure, g * uim)
    if model == "ComplEx":
        hr, hi = H
        rr, ri = R
        tr, ti = T
        g_hr = g * (rr * tr + ri * ti)
        g_hi = g * (rr * ti - ri * ti)
        g_rr = g * (hr * tr + hi * ti)
        g_ri = g * (hr * ti - hi * ti)
        g_tr = g * (hr * rr)
        g_ti = g * (hi * rr + hr * ri - hi * ri)
        return (g_hr, g_hi), (g_rr, g_ri), (g_tr, g_ti)
    raise ValueError(model)


# --------------------------------------------------------------------------------------
# losses on score vectors (float32 like torch)
# --------------------------------------------------------------------------------------
def bce_with_logits_mean(x, y):
    """torch.nn.BCEWithLogitsLoss(reduction='mean'): max(x,0) - x*y + log1p(exp(-|x|))."""
    x = np.asarray(x, F32)
    y = np.asarray(y, F32)
    l = np.maximum(x, F32(0)) - x * y + np.log1p(np.exp(-np.abs(x)))
    return F32(np.mean(l, dtype=np.float64))


def bce_with_logits_grad(x, y):
    x = np.asarray(x, np.float64)
    return (1.0 / (1.0 + np.exp(-x)) - np.asarray(y, np.float64)) / x.size


def margin_ranking_mean(pos, neg, margin):
    """torch.nn.MarginRankingLoss(margin)(pos, neg, +1) = mean(max(0, -(pos-neg)+margin))."""
    l = np.maximum(F32(0), -(np.asarray(pos, F32) - np.asarray(neg, F32)) + F32(margin))
    return F32(np.mean(l, dtype=np.float64))


def triplet_margin_mean(d_pos, d_neg, margin):
    """TripletMarginLoss given the two distances: mean(max(d_pos - d_neg + margin, 0))."""
    l = np.maximum(np.asarray(d_pos, F32) - np.asarray(d_neg, F32) + F32(margin), F32(0))
    return F32(np.mean(l, dtype=np.float64))


# --------------------------------------------------------------------------------------
# calculate_loss restatement (reference batch layout: SURVEY §8 A8)
# --------------------------------------------------------------------------------------
def _gather(tables, names, idx):
    return tuple(tables[n][idx] for n in names)


def _rec_rel_rows(model, tables, ui_relation, n):
    # TransE / DistMult / (ComplEx full_sort) use relation_embedding.weight[-1]; RotatE/ComplEx use the
    # ui_relation token id, which is the last id (kg_dataset.py:626-631) -- same row.
    _, rn, _ = slots(model, "rec")
    return tuple(np.broadcast_to(tables[k][ui_relation], (n, tables[k].shape[1])) for k in rn)


def calculate_loss(model, tables, inter, margin=1.0, ui_relation=None, with_grad=False):
    """Loss (float32 scalar) of one RSKG batch; optionally dense float64 grads per table.

    ``inter`` keys: user_id, item_id, neg_item_id, head_id, relation_id, tail_id, neg_tail_id.
    """
    if ui_relation is None:
        ui_relation = tables[slots(model, "rec")[1][0]].shape[0] - 1
    u, pi, ni = inter["user_id"], inter["item_id"], inter["neg_item_id"]
    h, r, pt, nt = inter["head_id"], inter["relation_id"], inter["tail_id"], inter["neg_tail_id"]
    hn_rec, rn, tn = slots(model, "rec")
    hn_kg, _, _ = slots(model, "kg")

    H_rec = _gather(tables, hn_rec, u)
    R_rec = _rec_rel_rows(model, tables, ui_relation, len(u))
    H_kg = _gather(tables, hn_kg, h)
    R_kg = _gather(tables, rn, r)
    Tp_rec, Tn_rec = _gather(tables, tn, pi), _gather(tables, tn, ni)
    Tp_kg, Tn_kg = _gather(tables, tn, pt), _gather(tables, tn, nt)

    n_rec, n_kg = len(u), len(h)
    if model == "TorusE":  # toruse.py:81-103: the triplet loss of TransE on the raw rows; the torus scorer is unused
        model = "TransE"
    triplet = model in ("TransE", "TransH", "TransD", "TransR")  # TripletMarginLoss(h_proj + r, t_pos_proj, t_neg_proj)
    eps = PAIRWISE_EPS if triplet else 0.0
    sp_rec = score_rows(model, H_rec, R_rec, Tp_rec, margin, eps)
    sn_rec = score_rows(model, H_rec, R_rec, Tn_rec, margin, eps)
    sp_kg = score_rows(model, H_kg, R_kg, Tp_kg, margin, eps)
    sn_kg = score_rows(model, H_kg, R_kg, Tn_kg, margin, eps)

    if model == "HolE":  # hole.py:100-104: MarginRankingLoss(sigmoid(pos), sigmoid(neg), 1)
        pos = np.concatenate([sp_rec, sp_kg]).astype(np.float64)
        neg = np.concatenate([sn_rec, sn_kg]).astype(np.float64)
        sp, sn = 1.0 / (1.0 + np.exp(-pos)), 1.0 / (1.0 + np.exp(-neg))
        loss = margin_ranking_mean(sp.astype(F32), sn.astype(F32), margin)
        act = ((-(sp - sn) + margin) > 0).astype(np.float64) / (n_rec + n_kg)
        gp, gn = -act * sp * (1 - sp), act * sn * (1 - sn)
        g = dict(p_rec=gp[:n_rec], n_rec=gn[:n_rec], p_kg=gp[n_rec:], n_kg=gn[n_rec:])
    elif model == "Analogy":  # analogy.py:196-198 + loss.py:172-190: mean softplus(-pos) + mean softplus(neg), rec + kg
        def softplus(x):
            x = np.asarray(x, np.float64)
            return np.maximum(x, 0) + np.log1p(np.exp(-np.abs(x)))

        sig = lambda x: 1.0 / (1.0 + np.exp(-np.asarray(x, np.float64)))
        loss = F32(softplus(-sp_rec).mean() + softplus(sn_rec).mean() + softplus(-sp_kg).mean() + softplus(sn_kg).mean())
        g = dict(p_rec=-sig(-sp_rec) / n_rec, n_rec=sig(sn_rec) / n_rec, p_kg=-sig(-sp_kg) / n_kg, n_kg=sig(sn_kg) / n_kg)
    elif triplet:  # transe.py:91-96: one TripletMarginLoss over cat(rec, kg)
        dp = -np.concatenate([sp_rec, sp_kg])
        dn = -np.concatenate([sn_rec, sn_kg])
        loss = triplet_margin_mean(dp, dn, margin)
        act = ((dp.astype(np.float64) - dn + margin) > 0).astype(np.float64) / (n_rec + n_kg)
        # d loss / d score_pos = -act (score = -dist), d loss / d score_neg = +act
        gp, gn = -act, act
        g = dict(p_rec=gp[:n_rec], n_rec=gn[:n_rec], p_kg=gp[n_rec:], n_kg=gn[n_rec:])
    elif model in ("DistMult", "RESCAL"):  # distmult.py:86-93, rescal.py:89-97
        pos = np.concatenate([sp_rec, sp_kg])
        neg = np.concatenate([sn_rec, sn_kg])
        loss = margin_ranking_mean(pos, neg, margin)
        act = ((-(pos.astype(np.float64) - neg) + margin) > 0).astype(np.float64) / (n_rec + n_kg)
        gp, gn = -act, act
        g = dict(p_rec=gp[:n_rec], n_rec=gn[:n_rec], p_kg=gp[n_rec:], n_kg=gn[n_rec:])
    else:  # RotatE rotate.py:118-131, ComplEx complex.py:115-128: BCE on rec and kg separately, summed
        x_rec = np.concatenate([sp_rec, sn_rec])
        y_rec = np.concatenate([np.ones(n_rec, F32), np.zeros(n_rec, F32)])
        x_kg = np.concatenate([sp_kg, sn_kg])
        y_kg = np.concatenate([np.ones(n_kg, F32), np.zeros(n_kg, F32)])
        loss = F32(bce_with_logits_mean(x_rec, y_rec) + bce_with_logits_mean(x_kg, y_kg))
        g_rec = bce_with_logits_grad(x_rec, y_rec)
        g_kg = bce_with_logits_grad(x_kg, y_kg)
        g = dict(p_rec=g_rec[:n_rec], n_rec=g_rec[n_rec:], p_kg=g_kg[:n_kg], n_kg=g_kg[n_kg:])
    if not with_grad:
        return loss

    grads = {k: np.zeros(v.shape, np.float64) for k, v in tables.items()}

    def scatter(names, idx, rows):
        for nme, gr in zip(names, rows):
            np.add.at(grads[nme], idx, gr)

    for (Hh, Rr, Tt, hidx, ridx, tidx, hn, gs) in (
        (H_rec, R_rec, Tp_rec, u, None, pi, hn_rec, g["p_rec"]),
        (H_rec, R_rec, Tn_rec, u, None, ni, hn_rec, g["n_rec"]),
        (H_kg, R_kg, Tp_kg, h, r, pt, hn_kg, g["p_kg"]),
        (H_kg, R_kg, Tn_kg, h, r, nt, hn_kg, g["n_kg"]),
    ):
        gH, gR, gT = score_rows_vjp(model, Hh, Rr, Tt, gs, margin, eps)
        scatter(hn, hidx, gH)
        scatter(tn, tidx, gT)
        if ridx is None:
            for nme, gr in zip(rn, gR):
                grads[nme][ui_relation] += gr.sum(axis=0)
        else:
            scatter(rn, ridx, gR)
    return loss, grads


# --------------------------------------------------------------------------------------
# predict / full sort
# --------------------------------------------------------------------------------------
def predict(model, tables, inter, margin=1.0, ui_relation=None):
    """predict(): score (user, ui_relation, item) rows."""
    if ui_relation is None:
        ui_relation = tables[slots(model, "rec")[1][0]].shape[0] - 1
    hn, rn, tn = slots(model, "rec")
    u, i = inter["user_id"], inter["item_id"]
    return score_rows(
        model, _gather(tables, hn, u), _rec_rel_rows(model, tables, ui_relation, len(u)), _gather(tables, tn, i), margin
    )


def predict_kg(model, tables, inter, margin=1.0):
    hn, rn, tn = slots(model, "kg")
    return score_rows(
        model,
        _gather(tables, hn, inter["head_id"]),
        _gather(tables, rn, inter["relation_id"]),
        _gather(tables, tn, inter["tail_id"]),
        margin,
    )


def full_sort_predict(model, tables, users, n_items, margin=1.0, ui_relation=None):
    """[B, n_items] scores of every item (entity rows [0, n_items)) for each user."""
    if ui_relation is None:
        ui_relation = tables[slots(model, "rec")[1][0]].shape[0] - 1
    hn, rn, tn = slots(model, "rec")
    H = tuple(tables[n][users][:, None, :] for n in hn)
    R = tuple(tables[n][ui_relation][None, None, :] for n in rn)
    T = tuple(tables[n][:n_items][None, :, :] for n in tn)
    return score_rows(model, H, R, T, margin)


def full_sort_predict_kg(model, tables, heads, rels, margin=1.0):
    """[B, n_entities] scores of every entity as tail."""
    hn, rn, tn = slots(model, "kg")
    if model == "TransD":  # transd.py:200-201: ``head_e_vec = self.entity_embedding(user)``
        hn = ("entity_embedding", "entity_embedding")
    H = tuple(tables[n][heads][:, None, :] for n in hn)
    R = tuple(tables[n][rels][:, None, :] for n in rn)
    T = tuple(tables[n][None, :, :] for n in tn)
    return score_rows(model, H, R, T, margin)


def score_grouped(model, tables, side, h_idx, r_idx, t_idx, margin=1.0):
    """Scores [B, 1+K] of B (h, r) groups against ``t_idx[B, 1+K]`` tails (pos first, then K negatives)."""
    hn, rn, tn = slots(model, side)
    H = tuple(tables[n][h_idx][:, None, :] for n in hn)
    R = tuple(tables[n][r_idx][:, None, :] for n in rn)
    T = tuple(tables[n][t_idx] for n in tn)
    return score_rows(model, H, R, T, margin)


# --------------------------------------------------------------------------------------
# evaluation protocol: mask + top-k (trainer.py:731-734, collector.py:173-183)
# --------------------------------------------------------------------------------------
def mask_scores(scores, history_index=None):
    s = np.array(scores, F32, copy=True)
    s[:, 0] = -np.inf
    if history_index is not None:
        s[history_index[0], history_index[1]] = -np.inf
    return s


def topk_desc(scores, k):
    """Top-k per row, ordered (score desc, index asc) -- the build's tie-break convention."""
    s = np.asarray(scores)
    n = s.shape[1]
    order = np.lexsort((np.broadcast_to(np.arange(n), s.shape), -s.astype(np.float64)), axis=1)[:, :k]
    return np.take_along_axis(s, order, axis=1), order.astype(np.int64)


# --------------------------------------------------------------------------------------
# self-adversarial negative weighting (SURVEY §8 A9 -- NOT in the reference; parity unpinned by it)
# --------------------------------------------------------------------------------------
def self_adversarial_loss(scores, alpha):
    """RotatE-paper loss on grouped scores [B, 1+K]: column 0 positive.

    p_i = softmax_i(alpha * s_neg_i) (treated as a constant), loss_b = -logsig(s_pos) - sum_i p_i logsig(-s_neg_i);
    returns mean over b (float64 math, float32 result).
    """
    s = np.asarray(scores, np.float64)
    pos, neg = s[:, 0], s[:, 1:]
    z = alpha * neg
    z = z - z.max(axis=1, keepdims=True)
    p = np.exp(z)
    p /= p.sum(axis=1, keepdims=True)

    def logsig(x):
        return -(np.maximum(-x, 0) + np.log1p(np.exp(-np.abs(x))))

    l = -logsig(pos) - np.sum(p * logsig(-neg), axis=1)
    return F32(l.mean())


def self_adversarial_grad(scores, alpha):
    s = np.asarray(scores, np.float64)
    B = s.shape[0]
    pos, neg = s[:, 0], s[:, 1:]
    z = alpha * neg
    z = z - z.max(axis=1, keepdims=True)
    p = np.exp(z)
    p /= p.sum(axis=1, keepdims=True)
    sig = lambda x: 1.0 / (1.0 + np.exp(-x))
    g = np.empty_like(s)
    g[:, 0] = -(1.0 - sig(pos)) / B  # d(-logsig(x)) = -(1 - sig(x))
    g[:, 1:] = p * sig(neg) / B  # d(-logsig(-x)) = sig(x)
    return g
