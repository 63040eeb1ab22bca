"""numpy restatement of TuckER's forward and loss (TEST INFRASTRUCTURE ONLY; see oracle/__init__.py).

Reference: /root/reference/hopwise/model/knowledge_graph_embedding_recommender/tucker.py
  forward :69-84, calculate_loss :97-127 (targets: union of the batch's positives in every row + label smoothing),
  predict :129-140 (indexes the user-table scores with the raw item id), full_sort_predict :154-162.
Pinned by tests/golden/tucker_d*.npz.  Gradients are pinned directly against the reference's autograd output
(the goldens), not restated here.
"""

from __future__ import annotations

import numpy as np

F32 = np.float32


def batch_norm(x, p, prefix, training, eps=1e-5):
    if training:
        mean, var = x.mean(0), x.var(0)  # biased variance normalises (torch BatchNorm1d)
    else:
        mean, var = p[prefix + ".running_mean"], p[prefix + ".running_var"]
    return ((x - mean) / np.sqrt(var + eps) * p[prefix + ".weight"] + p[prefix + ".bias"]).astype(F32)


def forward(p, h, r, table, training=False, masks=(None, None, None)):
    """sigmoid scores [B, rows(table)]; masks = multiplicative dropout masks (x [B,d], W_mat [B,d,d], x [B,d])."""
    d = h.shape[1]
    x = batch_norm(h, p, "bn0", training)
    if masks[0] is not None:
        x = x * masks[0]
    w_mat = (r @ p["weights"].reshape(d, d * d)).reshape(-1, d, d).astype(F32)
    if masks[1] is not None:
        w_mat = w_mat * masks[1]
    x = np.einsum("bi,bij->bj", x, w_mat).astype(F32)
    x = batch_norm(x, p, "bn1", training)
    if masks[2] is not None:
        x = x * masks[2]
    logits = (x @ table.T).astype(F32)
    return (1.0 / (1.0 + np.exp(-logits.astype(np.float64)))).astype(F32)


def bce_mean(pred, target):
    pred = pred.astype(F32)
    lp = np.maximum(np.log(pred), F32(-100.0))
    l1p = np.maximum(np.log(F32(1.0) - pred), F32(-100.0))
    return F32(np.mean(-(target * lp + (1.0 - target) * l1p), dtype=np.float64))


def calculate_loss(p, inter, n_users, n_items, n_entities, ui_relation, label_smoothing, training=False, masks=None):
    user, item = inter["user_id"], inter["item_id"]
    head, rel, tail = inter["head_id"], inter["relation_id"], inter["tail_id"]
    item_new = np.zeros((len(item), n_users + n_items), F32)
    item_new[:, item + n_users] = 1.0
    tail_new = np.zeros((len(tail), n_entities), F32)
    tail_new[:, tail] = 1.0
    if label_smoothing:
        item_new = ((1.0 - label_smoothing) * item_new + 1.0 / n_items).astype(F32)
        tail_new = ((1.0 - label_smoothing) * tail_new + 1.0 / n_entities).astype(F32)
    U, E, R = p["user_embedding.weight"], p["entity_embedding.weight"], p["relation_embedding.weight"]
    mr = masks["rec"] if masks else (None, None, None)
    mk = masks["kg"] if masks else (None, None, None)
    su = forward(p, U[user], R[np.full(len(user), ui_relation)], U, training, mr)
    sk = forward(p, E[head], R[rel], E, training, mk)
    return F32(bce_mean(su, item_new) + bce_mean(sk, tail_new))


def predict(p, inter, ui_relation):
    U, R = p["user_embedding.weight"], p["relation_embedding.weight"]
    user, item = inter["user_id"], inter["item_id"]
    s = forward(p, U[user], R[np.full(len(user), ui_relation)], U)
    return s[np.arange(len(user)), item]


def predict_kg(p, inter):
    E, R = p["entity_embedding.weight"], p["relation_embedding.weight"]
    s = forward(p, E[inter["head_id"]], R[inter["relation_id"]], E)
    return s[np.arange(len(inter["head_id"])), inter["tail_id"]]


def full_sort_predict(p, users, n_users, ui_relation):
    U, R = p["user_embedding.weight"], p["relation_embedding.weight"]
    return forward(p, U[users], R[np.full(len(users), ui_relation)], U)[:, n_users:]
