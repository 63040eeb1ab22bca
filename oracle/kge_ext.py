"""numpy restatement of the widened KGE scorer set: TorusE, TransH, TransD, Analogy, HolE.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Pinned by tests/golden/kge_<Model>_d*.npz, produced by the
reference classes themselves (tests/golden/make_golden.py).

Reference (paths relative to /root/reference/hopwise/model/knowledge_graph_embedding_recommender):
  TorusE   toruse.py:68-79 forward (frac_ then 4 * min(x^2, 1 - x^2)); calculate_loss :81-103 is TransE's triplet loss on the
           RAW rows (forward is not used in training)
  TransH   transh.py:51-55 forward, :74-75 project: ``ent - (ent * norm_vec(rel).sum(1).view(-1, 1)) * norm_vec(rel)`` --
           the scalar is the SUM of the norm vector, not a dot product; restated as written
  TransD   transd.py:89-94 forward (p(e) = r_vec (e . e_vec) + e); full_sort_predict_kg :196-217 takes the head "vec" row
           from entity_embedding (restated as written)
  Analogy  analogy.py:66-70 forward; LogisticLoss (model/loss.py:172-190) on negated scores, rec + kg
  TransR   transr.py:67-70 forward (p = M_r e with M_r = proj_mat_e(r).view(d, d)), :72-104 TripletMarginLoss; the
           reference defines no full_sort_predict_kg for it
  HolE     hole.py:53-61 forward with the rolling matrix R[i][j] = r[(j - i) mod d]; MarginRankingLoss on sigmoid(scores)
"""

from __future__ import annotations

import numpy as np

F32 = np.float32
EXT_MODELS = ("TorusE", "TransH", "TransD", "Analogy", "HolE", "TransR")


def slots(model, side):
    ent = "user" if side == "rec" else "entity"
    if model in ("TorusE", "HolE"):
        return (f"{ent}_embedding",), ("relation_embedding",), ("entity_embedding",)
    if model == "TransH":
        return (f"{ent}_embedding",), ("relation_embedding", "norm_vec"), ("entity_embedding",)
    if model == "TransR":
        return (f"{ent}_embedding",), ("relation_embedding", "proj_mat_e"), ("entity_embedding",)
    if model == "TransD":
        return ((f"{ent}_embedding", f"{ent}_vec_embedding"), ("relation_embedding", "relation_vec_embedding"),
                ("entity_embedding", "entity_vec_embedding"))
    if model == "Analogy":
        return ((f"{ent}_embedding", f"{ent}_re_embedding", f"{ent}_im_embedding"),
                ("relation_embedding", "relation_re_embedding", "relation_im_embedding"),
                ("entity_embedding", "entity_re_embedding", "entity_im_embedding"))
    raise ValueError(model)


def frac(x):
    return (x - np.trunc(x)).astype(F32)


def _roll_matrix(x):
    """[..., d] -> [..., d, d] with M[i][j] = x[(j - i) mod d] (hole.py:58-61)."""
    d = x.shape[-1]
    return np.stack([np.roll(x, i, axis=-1) for i in range(d)], axis=-2)


def score_rows(model, H, R, T, margin=0.0, eps=0.0):
    if model == "TorusE":
        x = (frac(H[0]) + frac(R[0])) - frac(T[0])
        x2 = x * x
        return -(F32(4) * np.sum(np.minimum(x2, F32(1) - x2), axis=-1, dtype=F32))
    if model == "TransH":
        w = R[1]
        s = np.sum(w, axis=-1, keepdims=True, dtype=F32)
        hp = H[0] - (H[0] * s) * w
        tp = T[0] - (T[0] * s) * w
        d = (hp + R[0]) - tp
        if eps:
            d = d + F32(eps)
        return -np.sqrt(np.sum(d * d, axis=-1, dtype=F32))
    if model == "TransD":
        hp = R[1] * np.sum(H[0] * H[1], axis=-1, keepdims=True, dtype=F32) + H[0]
        tp = R[1] * np.sum(T[0] * T[1], axis=-1, keepdims=True, dtype=F32) + T[0]
        d = (hp + R[0]) - tp
        if eps:
            d = d + F32(eps)
        return -np.sqrt(np.sum(d * d, axis=-1, dtype=F32))
    if model == "TransR":
        dd = H[0].shape[-1]
        M = R[1].reshape(R[1].shape[:-1] + (dd, dd))
        hp = np.einsum("...ck,...k->...c", M, H[0]).astype(F32)
        tp = np.einsum("...ck,...k->...c", M, T[0]).astype(F32)
        d = (hp + R[0]) - tp
        if eps:
            d = d + F32(eps)
        return -np.sqrt(np.sum(d * d, axis=-1, dtype=F32))
    if model == "Analogy":
        h, hr, hi = H
        r, rr, ri = R
        t, tr, ti = T
        return (np.sum(h * r * t, axis=-1, dtype=F32)
                + np.sum(hr * (rr * tr + ri * ti) + hi * (rr * ti - ri * tr), axis=-1, dtype=F32))
    if model == "HolE":
        hr = np.einsum("...i,...ij->...j", H[0], _roll_matrix(R[0])).astype(F32)
        return np.sum(hr * T[0], axis=-1, dtype=F32)
    raise ValueError(model)


def score_rows_vjp(model, H, R, T, g, margin=0.0, eps=0.0):
    """float64 row gradients of sum(g * score) for the differentiable members (TorusE's scorer is never differentiated)."""
    H = [np.asarray(x, np.float64) for x in H]
    R = [np.asarray(x, np.float64) for x in R]
    T = [np.asarray(x, np.float64) for x in T]
    g = np.asarray(g, np.float64)[..., None]

    def unit(d):
        n = np.sqrt(np.sum(d * d, axis=-1, keepdims=True))
        with np.errstate(invalid="ignore", divide="ignore"):
            return np.where(n > 0, d / n, 0.0)

    if model == "TransH":
        h, (r, w), t = H[0], R, T[0]
        s = w.sum(-1, keepdims=True)
        c = 1.0 - s * w
        d = h * c + r - t * c + eps
        u = g * unit(d)  # dL/dd = -u
        gc = -u * h + u * t
        gw = -s * gc - (gc * w).sum(-1, keepdims=True)
        return (-u * c,), (-u, gw), (u * c,)
    if model == "TransD":
        (h, hv), (r, rv), (t, tv) = H, R, T
        sh = (h * hv).sum(-1, keepdims=True)
        st = (t * tv).sum(-1, keepdims=True)
        d = h + rv * sh + r - t - rv * st + eps
        u = g * unit(d)
        a = -u  # dL/dd
        arv = (a * rv).sum(-1, keepdims=True)
        return ((a + arv * hv, arv * h), (a, a * sh - a * st), (-a - arv * tv, -arv * t))
    if model == "TransR":
        h, (r, m), t = H[0], R, T[0]
        dd = h.shape[-1]
        M = m.reshape(m.shape[:-1] + (dd, dd))
        d = np.einsum("...ck,...k->...c", M, h - t) + r + eps
        a = -g * unit(d)
        mta = np.einsum("...ck,...c->...k", M, a)
        gM = a[..., :, None] * (h - t)[..., None, :]
        return (mta,), (a, gM.reshape(gM.shape[:-2] + (dd * dd,))), (-mta,)
    if model == "Analogy":
        (h, hr, hi), (r, rr, ri), (t, tr, ti) = H, R, T
        return ((g * r * t, g * (rr * tr + ri * ti), g * (rr * ti - ri * tr)),
                (g * h * t, g * (hr * tr + hi * ti), g * (hr * ti - hi * tr)),
                (g * h * r, g * (hr * rr - hi * ri), g * (hr * ri + hi * rr)))
    if model == "HolE":
        h, r, t = H[0], R[0], T[0]
        gq = g * t
        Rm, Hm = _roll_matrix(r), _roll_matrix(h)
        gh = np.einsum("...j,...ij->...i", gq, Rm)
        grel = np.einsum("...j,...mj->...m", gq, Hm)
        hr = np.einsum("...i,...ij->...j", h, Rm)
        return (gh,), (grel,), (g * hr,)
    raise ValueError(model)
