"""numpy restatement of ConvE's forward and loss (TEST INFRASTRUCTURE ONLY; see oracle/__init__.py).

Reference: /root/reference/hopwise/model/knowledge_graph_embedding_recommender/conve.py
  forward :75-93, calculate_loss :110-137 (union-of-the-batch targets, label smoothing), predict :139-151,
  predict_kg :153-164 (``score[:, tail]``), full_sort_predict :166-176.
Pinned by tests/golden/conve_d*.npz (the reference class in eval mode and in train mode with recorded dropout /
dropout2d masks).  Gradients are pinned by the goldens directly (the device path is compared with the reference's
autograd output); this file restates the forward only.
"""

from __future__ import annotations

import numpy as np

F32 = np.float32


def _bn(x, p, pre, axes, stats=None, eps=1e-5):
    shape = [1] * x.ndim
    shape[1] = -1
    if stats == "batch":
        mu = x.mean(axes, dtype=np.float64).reshape(shape)
        var = x.var(axes, dtype=np.float64).reshape(shape)
    else:
        mu, var = p[pre + ".running_mean"].reshape(shape), p[pre + ".running_var"].reshape(shape)
    return (((x - mu) / np.sqrt(var + eps)) * p[pre + ".weight"].reshape(shape) + p[pre + ".bias"].reshape(shape)).astype(F32)


def conv2d_1ch(x, w, b):
    """x [B, 1, H, W], w [C, 1, kh, kw] -> [B, C, H - kh + 1, W - kw + 1] (stride 1, no padding)."""
    B, _, H, W = x.shape
    C, _, kh, kw = w.shape
    out = np.zeros((B, C, H - kh + 1, W - kw + 1), np.float64)
    for u in range(kh):
        for v in range(kw):
            out += w[:, 0, u, v].reshape(1, C, 1, 1) * x[:, :, u : u + H - kh + 1, v : v + W - kw + 1]
    if b is not None:
        out += b.reshape(1, C, 1, 1)
    return out.astype(F32)


def forward(p, h_idx, r_idx, head_table, out_table, bias, shape, masks=None):
    """sigmoid scores [B, rows(out_table)]; masks = (inp, feature, hidden) multiplicative masks -> train mode."""
    d1 = int(shape)
    d2 = p[head_table].shape[1] // d1
    stats = "batch" if masks is not None else None
    h = p[head_table][h_idx].reshape(-1, 1, d1, d2)
    r = p["relations_embeddings.weight"][r_idx].reshape(-1, 1, d1, d2)
    x = _bn(np.concatenate([h, r], 2), p, "bn0", (0, 2, 3), stats)
    if masks is not None:
        x = x * masks[0]
    x = conv2d_1ch(x, p["conv1.weight"], p.get("conv1.bias"))
    x = np.maximum(_bn(x, p, "bn1", (0, 2, 3), stats), 0)
    if masks is not None:
        x = x * masks[1]
    x = x.reshape(x.shape[0], -1) @ p["fc.weight"].T + p["fc.bias"]
    if masks is not None:
        x = x * masks[2]
    x = np.maximum(_bn(x.astype(F32), p, "bn2", (0,), stats), 0)
    z = x @ p[out_table].T + p[bias]
    return (1.0 / (1.0 + np.exp(-z.astype(np.float64)))).astype(F32)


def bce_mean(pred, target):
    pr = pred.astype(np.float64)
    l = -(target * np.maximum(np.log(pr), -100) + (1 - target) * np.maximum(np.log1p(-pr), -100))
    return float(l.mean())


def calculate_loss(p, inter, n_users, n_items, n_entities, ui_relation, shape, label_smoothing, masks=None):
    """masks: ((inp, feature, hidden) for the rec pass, (...) for the kg pass) or None (eval mode)."""
    user, item = inter["user_id"], inter["item_id"]
    su = forward(p, user, np.full_like(user, ui_relation), "user_embedding.weight", "user_embedding.weight", "b_users", shape,
                 None if masks is None else masks[0])
    sk = forward(p, inter["head_id"], inter["relation_id"], "entity_embedding.weight", "entity_embedding.weight",
                 "b_entities", shape, None if masks is None else masks[1])
    items = np.zeros((len(user), n_users + n_items))
    items[:, item + n_users] = 1.0
    tails = np.zeros((len(inter["head_id"]), n_entities))
    tails[:, inter["tail_id"]] = 1.0
    if label_smoothing:
        items = (1.0 - label_smoothing) * items + 1.0 / n_items
        tails = (1.0 - label_smoothing) * tails + 1.0 / n_entities
    return bce_mean(su, items) + bce_mean(sk, tails)
