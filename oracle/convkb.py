"""numpy restatement of ConvKB's forward and loss (TEST INFRASTRUCTURE ONLY; see oracle/__init__.py).

Reference: /root/reference/hopwise/model/knowledge_graph_embedding_recommender/convkb.py forward :65-82,
_get_regularization :84-88, calculate_loss :106-141; hopwise/model/loss.py LogisticLoss :172-190, RegLoss :104-116.
Pinned by tests/golden/convkb_d*.npz; gradients are pinned by the goldens directly.
"""

from __future__ import annotations

import numpy as np

from .conve import _bn, conv2d_1ch

F32 = np.float32


def forward(p, h, r, t, mask=None):
    """Scores of row triples [B, d] x 3; ``mask`` (multiplicative dropout mask) switches to train mode."""
    stats = "batch" if mask is not None else None
    x = np.stack([h, r, t], axis=2)[:, None].astype(F32)
    x = _bn(x, p, "conv1_bn", (0, 2, 3), stats)
    x = conv2d_1ch(x, p["conv_layer.weight"], p["conv_layer.bias"])
    x = np.maximum(_bn(x, p, "conv2_bn", (0, 2, 3), stats), 0)
    x = x.reshape(x.shape[0], -1)
    if mask is not None:
        x = x * mask
    return (x @ p["fc_layer.weight"].T).reshape(-1).astype(F32)


def regularization(p, h, r, t, lmbda):
    l2 = np.mean(h.astype(np.float64) ** 2) + np.mean(t.astype(np.float64) ** 2) + np.mean(r.astype(np.float64) ** 2)
    for k in ("conv_layer.weight", "conv_layer.bias", "fc_layer.weight"):
        l2 += np.sqrt((p[k].astype(np.float64) ** 2).sum())
    return lmbda * l2


def softplus(x):
    x = np.asarray(x, np.float64)
    return np.maximum(x, 0) + np.log1p(np.exp(-np.abs(x)))


def calculate_loss(p, inter, ui_relation, lmbda, masks=None):
    U, E, R = p["user_embedding.weight"], p["entity_embedding.weight"], p["relation_embedding.weight"]
    u, ur = U[inter["user_id"]], R[np.full_like(inter["user_id"], ui_relation)]
    ip, ineg = E[inter["item_id"]], E[inter["neg_item_id"]]
    h, r, tp, tn = E[inter["head_id"]], R[inter["relation_id"]], E[inter["tail_id"]], E[inter["neg_tail_id"]]
    mk = masks if masks is not None else [None] * 4
    s_pu, s_nu = forward(p, u, ur, ip, mk[0]), forward(p, u, ur, ineg, mk[1])
    s_pk, s_nk = forward(p, h, r, tp, mk[2]), forward(p, h, r, tn, mk[3])
    rec = softplus(-s_pu).mean() + softplus(s_nu).mean() + regularization(p, u, ur, ip, lmbda) + regularization(p, u, ur, ineg, lmbda)
    kg = softplus(-s_pk).mean() + softplus(s_nk).mean() + regularization(p, h, r, tp, lmbda) + regularization(p, h, r, tn, lmbda)
    return float(rec + kg)
