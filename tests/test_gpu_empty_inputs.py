"""Empty inputs at the model boundary: an evaluation chunk without queries gives empty outputs, a training batch without rows
gives the NaN loss the reference's means over nothing give (which Trainer._check_nan turns into its ValueError) -- instead of
an invalid-argument status from an entry point that was handed a null buffer."""

import pytest
import torch

import hopwise_amd.model as M
from hopwise_amd import ops
from tests.helpers import FakeDataset, make_config

pytestmark = pytest.mark.gpu
nU, nI, nE, nR = 60, 40, 120, 6


def _model(name):
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    return getattr(M, name)(make_config(dev, 16, 1.0), FakeDataset(nU, nI, nE, nR)).to(dev), dev


@pytest.mark.parametrize("name", ["TransE", "RotatE", "ComplEx", "DistMult", "RESCAL", "HolE"])
def test_empty_evaluation_inputs(name):
    m, dev = _model(name)
    e = torch.zeros(0, dtype=torch.int64, device=dev)
    it = {k: e for k in ("user_id", "item_id", "neg_item_id", "head_id", "relation_id", "tail_id", "neg_tail_id")}
    assert tuple(m.predict(it).shape) == (0,)
    assert tuple(m.predict_kg(it).shape) == (0,)
    assert m.full_sort_predict(it).view(-1, nI).shape[0] == 0
    assert m.full_sort_predict_kg(it).view(-1, nE).shape[0] == 0
    val, idx = m.full_sort_topk(it, 5)
    assert tuple(val.shape) == (0, 5) and idx.dtype == torch.int64
    scores = torch.zeros((0, nI), device=dev)
    assert ops.mask_scores_(scores, None) is scores
    v, i = ops.topk_rows(scores, 3)
    assert tuple(v.shape) == (0, 3) and tuple(i.shape) == (0, 3)


@pytest.mark.parametrize("name", ["TransE", "RotatE", "DistMult", "ComplEx"])
def test_empty_training_batch_is_nan_like_the_reference(name):
    m, dev = _model(name)
    e = torch.zeros(0, dtype=torch.int64, device=dev)
    it = {k: e for k in ("user_id", "item_id", "neg_item_id", "head_id", "relation_id", "tail_id", "neg_tail_id")}
    loss = m.calculate_loss(it)
    loss = sum(loss) if isinstance(loss, tuple) else loss
    assert torch.isnan(loss).item()
    # one empty side: NaN for the per-side means of the BCE models; the margin models average over the concatenated rows,
    # i.e. over the other side alone -- the loss of the same rows with the empty side's gathers skipped
    g = torch.Generator().manual_seed(1)
    some = {k: torch.randint(1, hi, (8,), generator=g).to(dev) for k, hi in (("head_id", nE), ("relation_id", nR - 1),
                                                                               ("tail_id", nE), ("neg_tail_id", nE))}
    half = dict(it, **some)
    if name in ("RotatE", "ComplEx"):
        assert torch.isnan(m.calculate_loss(half)).item()
    else:
        loss = m.calculate_loss(half)
        assert torch.isfinite(loss).item()
        loss.backward()
        assert m.entity_embedding.weight.grad is not None and torch.isfinite(m.entity_embedding.weight.grad).all().item()
        assert m.user_embedding.weight.grad is None or not m.user_embedding.weight.grad.any().item()  # no rec row touched
        if name == "DistMult":  # MarginRankingLoss over the kg rows alone (distmult.py:86-93 with an empty rec side)
            import numpy as np

            from oracle import kge as okge

            tables = {k: getattr(m, k).weight.detach().cpu().numpy() for k in okge.table_names(name)}
            inter = {k: v.cpu().numpy() for k, v in some.items()}
            pos = okge.predict_kg(name, tables, inter, margin=m.margin)
            neg = okge.predict_kg(name, tables, dict(inter, tail_id=inter["neg_tail_id"]), margin=m.margin)
            np.testing.assert_allclose(float(loss.detach()), okge.margin_ranking_mean(pos, neg, m.margin), rtol=1e-5)


def test_sampler_without_keys():
    import numpy as np

    from types import SimpleNamespace

    from hopwise_amd.sampler import KGSampler

    ds = SimpleNamespace(head_entities=np.array([1, 2, 3]), tail_entities=np.array([4, 5, 6]), entity_num=10)
    s = KGSampler(ds, device="cuda")
    calls = s.state_dict()["calls"]
    out = s.sample_by_entity_ids(np.zeros(0, np.int64), 1)
    assert tuple(out.shape) == (0,) and out.dtype == torch.int64
    assert s.state_dict()["calls"] == calls  # no draw: the stream position is untouched
