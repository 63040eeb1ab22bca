"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE (build container only).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden.py [kge|sampler|pgpr|all]

The reference (/root/reference, tail-unica/hopwise v0.9.1.post1) is imported through
``_refimport.install()``; this script only *calls* reference classes on seeded inputs and
stores inputs + outputs as .npz / .json.  No reference source text is copied.  The GPU box
never runs this file (it has no /root/reference); it only reads the committed fixtures.

What is pinned
  kge_<Model>_d<dim>.npz   tables, RSKG batch, predict / predict_kg scores, calculate_loss value,
                           dense table grads, full_sort_predict(_kg) matrices, masked top-10
  kge_train_<Model>.npz    3 Adam steps through the reference model (losses + final tables) -- A18 protocol
  sampler_*.npz            KGSampler / Sampler draws under a fixed numpy seed (stream-exact for the oracle)
  pgpr_*.npz / .json       PGPR on the reference's own tests/test_data fixture: actions, states, MLP outputs,
                           beam-search paths/probs (dropout patched to identity), collect_scores matrix
"""

import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refimport  # noqa: E402

_refimport.install()

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.set_num_threads(1)

FIELDS = dict(
    USER_ID_FIELD="user_id",
    ITEM_ID_FIELD="item_id",
    NEG_PREFIX="neg_",
    ENTITY_ID_FIELD="entity_id",
    RELATION_ID_FIELD="relation_id",
    HEAD_ENTITY_ID_FIELD="head_id",
    TAIL_ENTITY_ID_FIELD="tail_id",
)


class FakeDataset:
    """The four things a KGE model constructor reads from a dataset (transe.py:34-53, rotate.py:36-59)."""

    ui_relation = "[UI-Relation]"

    def __init__(self, n_users, n_items, n_entities, n_relations):
        self._num = dict(user_id=n_users, item_id=n_items, entity_id=n_entities, relation_id=n_relations)
        self.field2token_id = {"relation_id": {self.ui_relation: n_relations - 1}}

    def num(self, field):
        return self._num[field]


def make_batch(rng, n_users, n_items, n_entities, n_relations, b_rec, b_kg):
    inter = dict(
        user_id=rng.integers(1, n_users, b_rec),
        item_id=rng.integers(1, n_items, b_rec),
        neg_item_id=rng.integers(1, n_items, b_rec),
        head_id=rng.integers(1, n_entities, b_kg),
        relation_id=rng.integers(1, n_relations - 1, b_kg),
        tail_id=rng.integers(1, n_entities, b_kg),
        neg_tail_id=rng.integers(1, n_entities, b_kg),
    )
    # PAD row 0 and repeated indices must be exercised (scatter-add of grads into the same row)
    inter["user_id"][:3] = [0, 5, 5]
    inter["item_id"][:3] = [0, 7, 7]
    inter["neg_item_id"][1:3] = 9
    inter["head_id"][:4] = [0, 11, 11, 11]
    inter["tail_id"][:4] = [0, 13, 13, 11]
    inter["neg_tail_id"][:2] = [0, 13]
    inter["relation_id"][:3] = [0, n_relations - 1, 2]
    return {k: v.astype(np.int64) for k, v in inter.items()}


def gen_kge(out_dir, only=None):
    from hopwise.model.knowledge_graph_embedding_recommender.complex import ComplEx
    from hopwise.model.knowledge_graph_embedding_recommender.rescal import RESCAL
    from hopwise.model.knowledge_graph_embedding_recommender.distmult import DistMult
    from hopwise.model.knowledge_graph_embedding_recommender.rotate import RotatE
    from hopwise.model.knowledge_graph_embedding_recommender.transe import TransE

    from hopwise.model.knowledge_graph_embedding_recommender.analogy import Analogy
    from hopwise.model.knowledge_graph_embedding_recommender.hole import HolE
    from hopwise.model.knowledge_graph_embedding_recommender.toruse import TorusE
    from hopwise.model.knowledge_graph_embedding_recommender.transd import TransD
    from hopwise.model.knowledge_graph_embedding_recommender.transh import TransH
    from hopwise.model.knowledge_graph_embedding_recommender.transr import TransR

    nU, nI, nE, nR = 20, 16, 48, 5
    classes = dict(TransE=TransE, RotatE=RotatE, ComplEx=ComplEx, DistMult=DistMult, RESCAL=RESCAL, TorusE=TorusE,
                   TransH=TransH, TransD=TransD, Analogy=Analogy, HolE=HolE, TransR=TransR)
    for name, cls in classes.items():
        if only and name not in only:
            continue
        for d in ((8, 32, 64) if name in ("RESCAL", "TransR") else (8, 64, 200)):
            torch.manual_seed(2024 + d)
            config = dict(FIELDS, embedding_size=d, margin=1.0, device=torch.device("cpu"), scalar_share=0.5)
            model = cls(config, FakeDataset(nU, nI, nE, nR))
            if name == "TorusE":  # xavier rows are all inside (-1, 1): spread them so that frac() matters
                with torch.no_grad():
                    for p_ in model.parameters():
                        p_.mul_(6.0)
            rng = np.random.default_rng(7 + d)
            inter_np = make_batch(rng, nU, nI, nE, nR, 24, 32)
            inter = {k: torch.from_numpy(v) for k, v in inter_np.items()}

            out = {f"table.{k}": v.detach().numpy().copy() for k, v in model.state_dict().items()}
            out.update({f"inter.{k}": v for k, v in inter_np.items()})
            with torch.no_grad():
                out["predict"] = model.predict(inter).numpy()
                try:
                    out["predict_kg"] = model.predict_kg(inter).numpy()
                except (NotImplementedError, AttributeError):  # TransH defines neither kg-side predictor
                    pass
            model.zero_grad()
            loss = model.calculate_loss(inter)
            loss.backward()
            out["loss"] = loss.detach().numpy()
            for k, p in model.named_parameters():
                out[f"grad.{k}"] = p.grad.numpy().copy()
            with torch.no_grad():
                users = torch.tensor([1, 5, 0, 19, 5])
                fs = model.full_sort_predict({"user_id": users}).view(len(users), nI)
                heads = torch.tensor([3, 0, 47, 11])
                rels = torch.tensor([1, 4, 0, 2])
                try:
                    fskg = model.full_sort_predict_kg({"head_id": heads, "relation_id": rels}).view(len(heads), nE)
                    out["fskg.scores"] = fskg.numpy()
                except (NotImplementedError, AttributeError):
                    pass
            out["fs.users"] = users.numpy()
            out["fs.scores"] = fs.numpy()
            out["fskg.heads"] = heads.numpy()
            out["fskg.rels"] = rels.numpy()
            # evaluation protocol (trainer.py:731-734, collector.py:173-183)
            hist_rows = rng.integers(0, len(users), 12)
            hist_cols = rng.integers(1, nI, 12)
            masked = fs.clone()
            masked[:, 0] = -np.inf
            masked[torch.from_numpy(hist_rows), torch.from_numpy(hist_cols)] = -np.inf
            tv, ti = torch.topk(masked, 10, dim=-1)
            out["eval.history_rows"] = hist_rows.astype(np.int64)
            out["eval.history_cols"] = hist_cols.astype(np.int64)
            out["eval.masked"] = masked.numpy()
            out["eval.topk_values"] = tv.numpy()
            out["eval.topk_idx"] = ti.numpy()
            out["margin"] = np.float32(1.0)
            np.savez_compressed(os.path.join(out_dir, f"kge_{name}_d{d}.npz"), **out)
            print("kge", name, d, float(loss))

        # A18: three optimizer steps the way Trainer._train_epoch drives the model (trainer.py:208-268):
        # zero_grad -> calculate_loss -> backward -> Adam(lr=1e-3).step, dense grads.
        torch.manual_seed(99)
        d = 16
        config = dict(FIELDS, embedding_size=d, margin=1.0, device=torch.device("cpu"), scalar_share=0.5)
        model = cls(config, FakeDataset(nU, nI, nE, nR))
        out = {f"init.{k}": v.detach().numpy().copy() for k, v in model.state_dict().items()}
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=0.0)
        rng = np.random.default_rng(123)
        losses = []
        model.train()
        for step in range(3):
            inter_np = make_batch(rng, nU, nI, nE, nR, 16, 16)
            for k, v in inter_np.items():
                out[f"step{step}.{k}"] = v
            inter = {k: torch.from_numpy(v) for k, v in inter_np.items()}
            opt.zero_grad()
            loss = model.calculate_loss(inter)
            loss.backward()
            opt.step()
            losses.append(float(loss))
        out["losses"] = np.array(losses, np.float32)
        out.update({f"final.{k}": v.detach().numpy().copy() for k, v in model.state_dict().items()})
        np.savez_compressed(os.path.join(out_dir, f"kge_train_{name}.npz"), **out)
        print("kge_train", name, losses)


# ---------------------------------------------------------------------------------------------
# sampler (sampler.py:140-183, :186-291, :294-357)
# ---------------------------------------------------------------------------------------------
class _FakeKG:
    head_entity_field = "head_id"
    tail_entity_field = "tail_id"

    def __init__(self, heads, tails, entity_num):
        self.head_entities = heads
        self.tail_entities = tails
        self.entity_num = entity_num


class _FakeInter:
    uid_field = "user_id"
    iid_field = "item_id"

    def __init__(self, users, items, user_num, item_num):
        self.inter_feat = {"user_id": torch.from_numpy(users), "item_id": torch.from_numpy(items)}
        self.user_num = user_num
        self.item_num = item_num


def gen_sampler(out_dir):
    from hopwise.sampler import KGSampler, Sampler

    rng = np.random.default_rng(5)
    nE = 60
    heads = rng.integers(1, nE, 400).astype(np.int64)
    tails = rng.integers(1, nE, 400).astype(np.int64)
    # one dense head: almost every tail used -> many rejection rounds
    heads[:50] = 7
    tails[:50] = np.arange(2, 52)
    out = dict(heads=heads, tails=tails, entity_num=np.int64(nE))
    for dist in ("uniform", "popularity"):
        s = KGSampler(_FakeKG(heads, tails, nE), distribution=dist, alpha=0.75)
        keys = heads[rng.integers(0, 400, 33)]
        keys[:4] = 7
        np.random.seed(2024)
        v1 = s.sample_by_entity_ids(keys, 1).numpy()
        v3 = s.sample_by_entity_ids(keys, 3).numpy()
        same = np.full(9, 7, np.int64)  # all-keys-equal fast path (sampler.py:157-165)
        vs = s.sample_by_entity_ids(same, 4).numpy()
        out[f"kg.{dist}.keys"] = keys
        out[f"kg.{dist}.num1"] = v1
        out[f"kg.{dist}.num3"] = v3
        out[f"kg.{dist}.same_keys"] = same
        out[f"kg.{dist}.same_num4"] = vs

    nU, nI = 25, 30
    phases = ["train", "valid", "test"]
    dsets = []
    for p in range(3):
        users = rng.integers(1, nU, 80).astype(np.int64)
        items = rng.integers(1, nI, 80).astype(np.int64)
        dsets.append(_FakeInter(users, items, nU, nI))
        out[f"rec.{phases[p]}.users"] = users
        out[f"rec.{phases[p]}.items"] = items
    out["rec.user_num"] = np.int64(nU)
    out["rec.item_num"] = np.int64(nI)
    for dist in ("uniform", "popularity"):
        s = Sampler(phases, dsets, distribution=dist, alpha=1.0)
        for ph in phases:
            sp = s.set_phase(ph)
            keys = rng.integers(1, nU, 8).astype(np.int64)
            np.random.seed(11)
            v = sp.sample_by_user_ids(keys, None, 5).numpy()
            out[f"rec.{dist}.{ph}.keys"] = keys
            out[f"rec.{dist}.{ph}.num5"] = v
    np.savez_compressed(os.path.join(out_dir, "sampler.npz"), **out)
    print("sampler ok")


def gen_tucker(out_dir):
    """TuckER: eval-mode outputs + loss + grads, and a train-mode step with recorded dropout masks."""
    import torch.nn.functional as F
    from hopwise.model.knowledge_graph_embedding_recommender.tucker import TuckER

    from make_golden_pgpr import RecordingDropout

    nU, nI, nE, nR = 20, 16, 48, 5
    real = F.dropout
    for d in (8, 32):
        torch.manual_seed(500 + d)
        config = dict(FIELDS, embedding_size=d, device=torch.device("cpu"), label_smoothing=0.1, input_dropout=0.3,
                      input_dropout1=0.4, input_dropout2=0.5)
        model = TuckER(config, FakeDataset(nU, nI, nE, nR))
        with torch.no_grad():  # non-trivial BatchNorm state
            for bn in (model.bn0, model.bn1):
                bn.weight.uniform_(0.5, 1.5)
                bn.bias.normal_(0, 0.2)
                bn.running_mean.normal_(0, 0.1)
                bn.running_var.uniform_(0.5, 2.0)
        rng = np.random.default_rng(40 + d)
        inter_np = make_batch(rng, nU, nI, nE, nR, 12, 14)
        inter_np["item_id"] = np.maximum(inter_np["item_id"], 0)
        inter = {k: torch.from_numpy(v) for k, v in inter_np.items()}
        out = {f"p.{k}": v.detach().numpy().copy() for k, v in model.state_dict().items()}
        out.update({f"inter.{k}": v for k, v in inter_np.items()})
        model.eval()
        with torch.no_grad():
            out["eval.predict"] = model.predict(inter).numpy()
            out["eval.predict_kg"] = model.predict_kg(inter).numpy()
            users = torch.tensor([1, 5, 0, 19])
            out["eval.fs_users"] = users.numpy()
            out["eval.fs"] = model.full_sort_predict({"user_id": users}).numpy()
        model.zero_grad()
        loss = model.calculate_loss(inter)
        loss.backward()
        out["eval.loss"] = loss.detach().numpy()
        for k, prm in model.named_parameters():
            out[f"eval.grad.{k}"] = prm.grad.numpy().copy()
        model.train()
        rec = RecordingDropout(7 + d)
        F.dropout = rec
        model.zero_grad()
        loss = model.calculate_loss(inter)
        loss.backward()
        F.dropout = real
        out["train.loss"] = loss.detach().numpy()
        for i, m in enumerate(rec.masks):
            out[f"train.mask{i}"] = m
        for k, prm in model.named_parameters():
            out[f"train.grad.{k}"] = prm.grad.numpy().copy()
        for k, v in model.state_dict().items():
            if "running" in k or "num_batches" in k:
                out[f"train.after.{k}"] = v.numpy().copy()
        np.savez_compressed(os.path.join(out_dir, f"tucker_d{d}.npz"), **out)
        print("tucker", d, float(out["eval.loss"]), float(out["train.loss"]), len(rec.masks))


WIDE = dict(nU=300, nI=170, nE=900, nR=9, d=200, b_rec=40, b_kg=44, core_seed=20200, core_scale=0.05, stride=5)


def wide_core(seed, d, scale):
    """The [d, d, d] core of the d = 200 golden, regenerated by the tests instead of stored (32 MB): numpy's frozen legacy
    MT19937 stream (RandomState is guaranteed stable across numpy versions), scaled like a trained core."""
    return (np.random.RandomState(seed).uniform(-1.0, 1.0, (d, d, d)) * scale).astype(np.float32)


def gen_tucker_wide(out_dir):
    """TuckER at BASELINE configs[4]'s width (d = 200) on a small ragged id space: the reference's eval-mode outputs, loss and
    gradients, and one training-mode step with recorded dropout masks.  Kept small by NOT storing what can be regenerated or
    summarised: the core tensor comes from `wide_core` (seed stored, probe values stored to validate the regeneration), the
    [B, d, d] dropout masks are stored as packed bits, and the core gradient as its [::5, ::5, ::5] sample plus float64 sums."""
    import torch.nn.functional as F
    from hopwise.model.knowledge_graph_embedding_recommender.tucker import TuckER

    from make_golden_pgpr import RecordingDropout

    W = WIDE
    nU, nI, nE, nR, d = W["nU"], W["nI"], W["nE"], W["nR"], W["d"]
    torch.manual_seed(777)
    config = dict(FIELDS, embedding_size=d, device=torch.device("cpu"), label_smoothing=0.1, input_dropout=0.3,
                  input_dropout1=0.4, input_dropout2=0.5)
    model = TuckER(config, FakeDataset(nU, nI, nE, nR))
    core = wide_core(W["core_seed"], d, W["core_scale"])
    with torch.no_grad():
        model.weights.copy_(torch.from_numpy(core))
        for bn in (model.bn0, model.bn1):
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.normal_(0, 0.2)
            bn.running_mean.normal_(0, 0.1)
            bn.running_var.uniform_(0.5, 2.0)
    rng = np.random.default_rng(4200)
    inter_np = make_batch(rng, nU, nI, nE, nR, W["b_rec"], W["b_kg"])
    inter_np["item_id"] = np.maximum(inter_np["item_id"], 0)
    inter = {k: torch.from_numpy(v) for k, v in inter_np.items()}
    out = {f"p.{k}": v.detach().numpy().copy() for k, v in model.state_dict().items() if k != "weights"}
    out["shape"] = np.array([nU, nI, nE, nR, d], dtype=np.int64)
    out["core.seed"], out["core.scale"] = np.int64(W["core_seed"]), np.float64(W["core_scale"])
    out["core.probe"] = core[::40, ::40, ::40].copy()
    out["core.sum"] = np.float64(core.astype(np.float64).sum())
    out.update({f"inter.{k}": v for k, v in inter_np.items()})
    s = W["stride"]

    def put_grads(prefix):
        for k, prm in model.named_parameters():
            gnp = prm.grad.numpy()
            if k == "weights":
                out[f"{prefix}.grad.weights.sample"] = gnp[::s, ::s, ::s].copy()
                out[f"{prefix}.grad.weights.sum"] = np.float64(gnp.astype(np.float64).sum())
                out[f"{prefix}.grad.weights.abs_sum"] = np.float64(np.abs(gnp.astype(np.float64)).sum())
                out[f"{prefix}.grad.weights.max_abs"] = np.float64(np.abs(gnp).max())
            else:
                out[f"{prefix}.grad.{k}"] = gnp.copy()

    model.eval()
    with torch.no_grad():
        out["eval.predict"] = model.predict(inter).numpy()
        out["eval.predict_kg"] = model.predict_kg(inter).numpy()
        users = torch.tensor([1, 5, 0, 299, 123])
        out["eval.fs_users"] = users.numpy()
        out["eval.fs"] = model.full_sort_predict({"user_id": users}).numpy()
    model.zero_grad()
    loss = model.calculate_loss(inter)
    loss.backward()
    out["eval.loss"] = loss.detach().numpy()
    put_grads("eval")
    model.train()
    rec = RecordingDropout(99)
    real = F.dropout
    F.dropout = rec
    model.zero_grad()
    loss = model.calculate_loss(inter)
    loss.backward()
    F.dropout = real
    out["train.loss"] = loss.detach().numpy()
    ps = (0.3, 0.4, 0.5, 0.3, 0.4, 0.5)
    for i, m in enumerate(rec.masks):  # values are 0 or 1 / (1 - p): one bit each
        out[f"train.maskbits{i}"] = np.packbits((m != 0).reshape(-1))
        out[f"train.maskshape{i}"] = np.array(m.shape, dtype=np.int64)
        out[f"train.maskp{i}"] = np.float64(ps[i])
        assert np.array_equal(m, (m != 0).astype(np.float32) / np.float32(1.0 - ps[i])), i
    put_grads("train")
    for k, v in model.state_dict().items():
        if "running" in k or "num_batches" in k:
            out[f"train.after.{k}"] = v.numpy().copy()
    np.savez_compressed(os.path.join(out_dir, "tucker_wide_d200.npz"), **out)
    print("tucker wide", d, float(out["eval.loss"]), float(out["train.loss"]), len(rec.masks),
          os.path.getsize(os.path.join(out_dir, "tucker_wide_d200.npz")))


def gen_conve(out_dir):
    """ConvE: eval-mode outputs + loss + grads, and a train-mode step with recorded dropout / dropout2d masks."""
    import torch.nn.functional as F
    from hopwise.model.knowledge_graph_embedding_recommender.conve import ConvE

    from make_golden_pgpr import RecordingDropout

    class RecordingDropout2d(RecordingDropout):
        def __call__(self, x, p=0.5, training=True, inplace=False):
            keep = (torch.rand(x.shape[:2], generator=self.gen) >= p).float().view(*x.shape[:2], 1, 1) / (1.0 - p)
            self.masks.append(keep.numpy().copy())
            return x * keep

    nU, nI, nE, nR = 20, 16, 48, 5
    real, real2 = F.dropout, F.dropout2d
    for d, shape in ((32, 4), (64, 8)):  # d = 200 would store 3 x 7.8 MB of fc weights and gradients
        torch.manual_seed(900 + d)
        d2 = d // shape
        config = dict(FIELDS, embedding_size=d, device=torch.device("cpu"), label_smoothing=0.1, input_dropout=0.2,
                      hidden_dropout=0.3, feature_dropout=0.2, embedding_shape=shape, hidden_size=32 * (2 * shape - 2) * (d2 - 2),
                      use_bias=True)
        model = ConvE(config, FakeDataset(nU, nI, nE, nR))
        with torch.no_grad():  # non-trivial BatchNorm state and biases
            for bn in (model.bn0, model.bn1, model.bn2):
                bn.weight.uniform_(0.5, 1.5)
                bn.bias.normal_(0, 0.2)
                bn.running_mean.normal_(0, 0.1)
                bn.running_var.uniform_(0.5, 2.0)
            model.b_users.normal_(0, 0.1)
            model.b_entities.normal_(0, 0.1)
        rng = np.random.default_rng(60 + d)
        inter_np = make_batch(rng, nU, nI, nE, nR, 12, 14)
        inter = {k: torch.from_numpy(v) for k, v in inter_np.items()}
        out = {f"p.{k}": v.detach().numpy().copy() for k, v in model.state_dict().items()}
        out.update({f"inter.{k}": v for k, v in inter_np.items()})
        out["embedding_shape"], out["hidden_size"] = np.int64(shape), np.int64(config["hidden_size"])
        model.eval()
        with torch.no_grad():
            out["eval.predict"] = model.predict(inter).numpy()
            out["eval.predict_kg"] = model.predict_kg(inter).numpy()
            users = torch.tensor([1, 5, 0, 19])
            out["eval.fs_users"] = users.numpy()
            out["eval.fs"] = model.full_sort_predict({"user_id": users}).numpy()
        model.zero_grad()
        loss = model.calculate_loss(inter)
        loss.backward()
        out["eval.loss"] = loss.detach().numpy()
        for k, prm in model.named_parameters():
            out[f"eval.grad.{k}"] = prm.grad.numpy().copy()
        model.train()
        rec, rec2 = RecordingDropout(7 + d), RecordingDropout2d(17 + d)
        F.dropout, F.dropout2d = rec, rec2
        model.zero_grad()
        loss = model.calculate_loss(inter)
        loss.backward()
        F.dropout, F.dropout2d = real, real2
        out["train.loss"] = loss.detach().numpy()
        for i, m in enumerate(rec.masks):
            out[f"train.mask{i}"] = m
        for i, m in enumerate(rec2.masks):
            out[f"train.mask2d{i}"] = m
        for k, prm in model.named_parameters():
            out[f"train.grad.{k}"] = prm.grad.numpy().copy()
        for k, v in model.state_dict().items():
            if "running" in k:
                out[f"train.after.{k}"] = v.numpy().copy()
        np.savez_compressed(os.path.join(out_dir, f"conve_d{d}.npz"), **out)
        print("conve", d, float(out["eval.loss"]), float(out["train.loss"]), len(rec.masks), len(rec2.masks))


def gen_convkb(out_dir):
    """ConvKB: eval-mode predict / predict_kg / loss / grads, and a train-mode step with recorded dropout masks."""
    import torch.nn.functional as F
    from hopwise.model.knowledge_graph_embedding_recommender.convkb import ConvKB

    from make_golden_pgpr import RecordingDropout

    nU, nI, nE, nR = 20, 16, 48, 5
    real = F.dropout
    for d, ks, T in ((16, 1, 8), (50, 1, 64), (24, 3, 5)):
        torch.manual_seed(1300 + d)
        config = {**FIELDS, "embedding_size": d, "device": torch.device("cpu"), "out_channels": T, "kernel_size": ks,
                  "dropout_prob": 0.3, "lambda": 0.2}
        model = ConvKB(config, FakeDataset(nU, nI, nE, nR))
        with torch.no_grad():
            for bn in (model.conv1_bn, model.conv2_bn):
                bn.weight.uniform_(0.5, 1.5)
                bn.bias.normal_(0, 0.2)
                bn.running_mean.normal_(0, 0.1)
                bn.running_var.uniform_(0.5, 2.0)
        rng = np.random.default_rng(70 + d)
        inter_np = make_batch(rng, nU, nI, nE, nR, 12, 14)
        inter = {k: torch.from_numpy(v) for k, v in inter_np.items()}
        out = {f"p.{k}": v.detach().numpy().copy() for k, v in model.state_dict().items()}
        out.update({f"inter.{k}": v for k, v in inter_np.items()})
        out["kernel_size"], out["out_channels"] = np.int64(ks), np.int64(T)
        model.eval()
        with torch.no_grad():
            out["eval.predict"] = model.predict(inter).numpy()
            out["eval.predict_kg"] = model.predict_kg(inter).numpy()
        model.zero_grad()
        loss = model.calculate_loss(inter)
        loss.backward()
        out["eval.loss"] = loss.detach().numpy()
        for k, prm in model.named_parameters():
            out[f"eval.grad.{k}"] = prm.grad.numpy().copy()
        model.train()
        rec = RecordingDropout(3 + d)
        F.dropout = rec
        model.zero_grad()
        loss = model.calculate_loss(inter)
        loss.backward()
        F.dropout = real
        out["train.loss"] = loss.detach().numpy()
        for i, m in enumerate(rec.masks):
            out[f"train.mask{i}"] = m
        for k, prm in model.named_parameters():
            out[f"train.grad.{k}"] = prm.grad.numpy().copy()
        for k, v in model.state_dict().items():
            if "running" in k:
                out[f"train.after.{k}"] = v.numpy().copy()
        np.savez_compressed(os.path.join(out_dir, f"convkb_d{d}.npz"), **out)
        print("convkb", d, float(out["eval.loss"]), float(out["train.loss"]), len(rec.masks))


CAFE_VARIANTS = {"default": {}, "shallow": {"deep_module": False}, "nodropout": {"use_dropout": False}}


def gen_cafe(out_dir, tag="default"):
    """CAFE inference on the reference fixture: module outputs, infer_with_path for every (user, metapath), executor +
    run_program with the child permutation patched to the identity, eval mode.  ``tag``: "default" (CAFE.yaml: DeepRelationModule
    with dropout), "shallow" (deep_module: False -> RelationModule, cafe.py:645-667), "nodropout" (use_dropout: False)."""
    import pickle

    from hopwise.config import Config
    from hopwise.data import create_dataset, data_preparation
    from hopwise.model.knowledge_aware_recommender.cafe import KGMask
    from hopwise.utils import get_model, init_seed

    os.makedirs("/tmp/scratch", exist_ok=True)
    os.chdir("/tmp/scratch")
    config = Config(model="CAFE", dataset="test", config_file_list=["/root/reference/tests/model/test_model.yaml"],
                    config_dict={"data_path": "/root/reference/tests/test_data/", "user_inter_num_interval": [1, 10],
                                 "device": "cpu", "checkpoint_dir": "/tmp/scratch/saved", **CAFE_VARIANTS[tag]})
    init_seed(config["seed"], config["reproducibility"])
    dataset = create_dataset(config)
    train_data, valid_data, test_data = data_preparation(config, dataset)
    ds = train_data._dataset
    model = get_model("CAFE")(config, ds).to("cpu")
    torch.manual_seed(77)
    with torch.no_grad():  # non-trivial BatchNorm statistics
        for m in model.model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 2.0)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.2)
    model.eval()

    class IdentityRng:
        def permutation(self, x):
            return list(x) if not isinstance(x, (int, np.integer)) else np.arange(x)

        def __getattr__(self, k):
            return getattr(np.random.default_rng(0), k)

    model.rng = IdentityRng()
    out = {}
    used = sorted({r for mp in model.metapaths for r, _ in mp[1:]})
    for k, v in model.state_dict().items():
        if any(k.startswith(f"model.{r}.") for r in used) and "num_batches" not in k:
            out[f"p.{k}"] = v.numpy().copy()
    out["relation_names"] = np.array([str(model.rid2relation[i]) for i in range(len(model.rid2relation))])
    out["metapaths"] = np.array([[str(r) for r, _ in mp[1:]] for mp in model.metapaths])
    out["inter_users"] = ds.inter_feat[ds.uid_field].numpy().astype(np.int64)
    out["inter_items"] = ds.inter_feat[ds.iid_field].numpy().astype(np.int64)
    out["heads"] = np.asarray(ds.head_entities).astype(np.int32)
    out["rels"] = np.asarray(ds.relations).astype(np.int32)
    out["tails"] = np.asarray(ds.tail_entities).astype(np.int32)
    out["n_users"], out["n_items"], out["n_entities"] = np.int64(ds.user_num), np.int64(ds.item_num), np.int64(ds.entity_num)
    out["ui_relation"] = np.int64(model.ui_relation_id)
    out["user_emb"], out["entity_emb"] = model.user_embedding.numpy(), model.entity_embedding.numpy()
    out["positives"] = model.positives.numpy().astype(np.int64)
    out["topk_paths"] = np.int64(model.topk_paths)
    active = np.unique(out["inter_users"])
    users = torch.from_numpy(np.concatenate([active[:10], [0]]).astype(np.int64))
    out["users"] = users.numpy()
    kgm = KGMask(model.graph_dict, model.ui_relation_id)
    with torch.no_grad():
        # module outputs for a few (relation, input) pairs
        uv = model.user_embedding[users]
        ev = model.entity_embedding[torch.arange(3, 3 + len(users))]
        for r in used[:6]:
            out[f"mod.{r}"] = getattr(model.model, r)((ev, uv)).numpy()
        out["mod.inputs_e"], out["mod.inputs_u"] = ev.numpy(), uv.numpy()
        infer = {}
        for u in users.tolist():
            for mpid, mp in enumerate(model.metapaths):
                res = model.model.infer_with_path(mp, u, kgm, excluded_pids=model.positives[u], topk_paths=model.topk_paths)
                infer[(u, mpid)] = ([list(map(int, p)) for p, _ in res], [[float(x) for x in v] for _, v in res])
        scores, collected = model.run_program(users, model._estimate_path_count(users), model._infer_paths(users, kgm))
    out["infer.keys"] = np.array([[u, m] for (u, m) in infer], np.int64)
    out["infer.off"] = np.cumsum([0] + [len(v[0]) for v in infer.values()]).astype(np.int64)
    out["infer.paths"] = np.array([p for v in infer.values() for p in v[0]], np.int64).reshape(-1, 4)
    out["infer.vals"] = np.array([p for v in infer.values() for p in v[1]], np.float32).reshape(-1, 3)
    out["run.scores"] = scores.numpy()
    out["run.uis"] = np.array([[c[0], c[1]] for c in collected], np.int64).reshape(-1, 2)
    out["run.score"] = np.array([c[2] for c in collected], np.float32)
    out["run.path_ids"] = np.array([[n[2] for n in c[3]] for c in collected], np.int64).reshape(-1, 4)
    out["run.path_rels"] = np.array([[str(n[0]) for n in c[3][1:]] for c in collected]).reshape(-1, 3)
    # training: one batch through the reference's sampler and SymbolicNetwork.forward in train mode (batch-statistics
    # BatchNorm, recorded dropout masks), losses + gradients of the modules on the sampled metapath
    import random

    import torch.nn.functional as F
    from make_golden_pgpr import RecordingDropout

    model.train()
    model.rng = np.random.default_rng(5)
    random.seed(5)
    tr_users = torch.from_numpy(active[:24].astype(np.int64))
    mpid, pos_paths, neg_pids = model._get_batch_by_user(tr_users)
    out["train.mpid"], out["train.pos_paths"], out["train.neg_pids"] = np.int64(mpid), pos_paths.astype(np.int64), neg_pids.astype(np.int64)
    real = F.dropout
    rec = RecordingDropout(31)
    F.dropout = rec
    model.zero_grad()
    reg_loss, rank_loss = model.model.forward(model.metapaths[mpid], torch.from_numpy(pos_paths), torch.from_numpy(neg_pids))
    (reg_loss + rank_loss * model.rank_weight).backward()
    F.dropout = real
    out["train.reg_loss"], out["train.rank_loss"] = reg_loss.detach().numpy(), rank_loss.detach().numpy()
    out["train.rank_weight"] = np.float64(model.rank_weight)
    for i, m in enumerate(rec.masks):
        out[f"train.mask{i}"] = m
    for k, prm in model.named_parameters():
        if prm.grad is not None:
            out[f"train.grad.{k}"] = prm.grad.numpy().copy()
    for k, v in model.state_dict().items():
        if "running" in k and any(k.startswith(f"model.{r}.") for r, _ in model.metapaths[mpid][1:]):
            out[f"train.after.{k}"] = v.numpy().copy()
    out["deep_module"], out["use_dropout"] = np.bool_(config["deep_module"]), np.bool_(config["use_dropout"])
    np.savez_compressed(os.path.join(out_dir, f"cafe_{tag}.npz"), **out)
    print("cafe", tag, len(infer), out["infer.paths"].shape, scores.shape, len(collected), "train", mpid, pos_paths.shape,
          float(reg_loss), float(rank_loss), len(rec.masks))


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("kge", "all"):
        gen_kge(HERE)
    if what.startswith("kge:"):
        gen_kge(HERE, only=what[4:].split(","))
    if what in ("sampler", "all"):
        gen_sampler(HERE)
    if what in ("cafe", "all"):
        for tag in CAFE_VARIANTS:
            gen_cafe(HERE, tag)
    if what.startswith("cafe:"):
        gen_cafe(HERE, what[5:])
    if what in ("convkb", "all"):
        gen_convkb(HERE)
    if what in ("conve", "all"):
        gen_conve(HERE)
    if what in ("tucker", "all"):
        gen_tucker(HERE)
    if what in ("tucker_wide", "all"):
        gen_tucker_wide(HERE)
    if what in ("pgpr", "all"):
        from make_golden_pgpr import gen_pgpr

        gen_pgpr(HERE)
    if what in ("tprec", "all"):
        from make_golden_tprec import gen_tprec

        gen_tprec(HERE)
    if what in ("pgpr_train", "all"):
        from make_golden_pgpr_train import gen_pgpr_train

        gen_pgpr_train(HERE)
