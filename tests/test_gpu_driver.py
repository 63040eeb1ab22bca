"""End-to-end plumbing on the device (BASELINE config 1 shape, scaled down): synthetic KG -> device samplers ->
KGBatcher -> model.calculate_loss -> Adam, then full-sort evaluation with history masking."""

import types

import numpy as np
import pytest
import torch

from tests.helpers import FakeDataset, make_config

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model_name,sample_num", [("TransE", 1), ("RotatE", 4), ("DistMult", 1), ("ComplEx", 2)])
def test_train_and_eval_protocol(model_name, sample_num):
    import hopwise_amd.model as M
    from hopwise_amd.data import synthetic
    from hopwise_amd.driver import KGBatcher, full_sort_topk, train_epoch
    from hopwise_amd.sampler import KGSampler, Sampler

    dev = torch.device("cuda", 0)
    kg = synthetic.make("tiny", seed=1)
    ds = types.SimpleNamespace(uid_field="user_id", iid_field="item_id", user_num=kg.n_users, item_num=kg.n_items,
                               inter_feat={"user_id": kg.inter_users, "item_id": kg.inter_items},
                               head_entities=kg.heads, tail_entities=kg.tails, entity_num=kg.n_entities)
    rec = Sampler("train", ds, device=dev).set_phase("train")
    kgs = KGSampler(ds, device=dev)
    torch.manual_seed(0)
    cfg = make_config(dev, 64, 1.0, train_neg_sample_args={"sample_num": sample_num})
    model = getattr(M, model_name)(cfg, FakeDataset(kg.n_users, kg.n_items, kg.n_entities, kg.n_relations)).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    b = KGBatcher(kg.inter_users, kg.inter_items, kg.heads, kg.relations, kg.tails, rec, kgs, train_batch_size=256,
                  sample_num=sample_num, device=dev)
    losses = [train_epoch(model, b, opt) / len(b) for _ in range(6)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0], losses

    users = torch.arange(1, 9, device=dev)
    hist_r = torch.tensor([0, 0, 3], device=dev)
    hist_c = torch.tensor([5, 7, 2], device=dev)
    val, idx, masked = full_sort_topk(model, {"user_id": users}, (hist_r, hist_c), k=10)
    assert tuple(masked.shape) == (8, kg.n_items)
    m = masked.cpu().numpy()
    assert np.isneginf(m[:, 0]).all() and np.isneginf(m[0, 5]) and np.isneginf(m[0, 7]) and np.isneginf(m[3, 2])
    i = idx.cpu().numpy()
    assert not (i == 0).any() and 5 not in i[0] and 7 not in i[0] and 2 not in i[3]
    v = val.cpu().numpy()
    assert (np.diff(v, axis=1) <= 0).all()
    np.testing.assert_array_equal(np.take_along_axis(m, i, 1), v)
    # link-prediction flavour
    heads = torch.tensor([3, 4], device=dev)
    rels = torch.tensor([1, 2], device=dev)
    val, idx, masked = full_sort_topk(model, {"head_id": heads, "relation_id": rels}, None, k=5, kg=True)
    assert tuple(masked.shape) == (2, kg.n_entities)


def test_train_from_atomic_files():
    """(f)2: .inter/.kg/.link -> KGFiles -> device samplers -> RotatE epoch -> fused full-sort top-k."""
    import os

    import hopwise_amd.model as M
    from hopwise_amd.data.kg_files import KGFiles
    from hopwise_amd.driver import KGBatcher, train_epoch
    from hopwise_amd.sampler import KGSampler, Sampler

    dev = torch.device("cuda", 0)
    ds = KGFiles(os.path.join(os.path.dirname(__file__), "golden", "atomic"), "kg_reverse_r", kg_reverse_r=True)
    rec = Sampler("train", ds, device=dev).set_phase("train")
    kgs = KGSampler(ds, device=dev)
    torch.manual_seed(0)
    model = M.RotatE(make_config(dev, 16, 2.0, train_neg_sample_args={"sample_num": 2}), ds).to(dev)
    assert model.n_entities == ds.entity_num and model.n_relations == ds.relation_num
    opt = torch.optim.Adam(model.parameters(), lr=5e-2)
    b = KGBatcher(ds.inter_feat["user_id"], ds.inter_feat["item_id"], ds.head_entities, ds.relations,
                  ds.tail_entities, rec, kgs, train_batch_size=4, sample_num=2, device=dev)
    losses = [train_epoch(model, b, opt) / len(b) for _ in range(8)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0], losses
    users = torch.arange(1, ds.user_num, device=dev)
    scores = model.full_sort_predict({"user_id": users}).view(len(users), -1)
    assert scores.shape[1] == ds.item_num and torch.isfinite(scores).all()


@pytest.mark.parametrize("model_name", ["RotatE", "TransE", "TransD"])
def test_graphed_train_step_equals_eager(model_name):
    """The HIP-graph replay of zero_grad / calculate_loss / backward / Adam gives the eager run's losses and weights."""
    import copy

    import hopwise_amd.model as M
    from hopwise_amd.driver import GraphedTrainStep

    dev = torch.device("cuda", 0)
    nU, nI, nE, nR, nb = 300, 200, 900, 7, 512
    torch.manual_seed(3)
    eager = getattr(M, model_name)(make_config(dev, 64, 1.0), FakeDataset(nU, nI, nE, nR)).to(dev)
    graphed = copy.deepcopy(eager)
    rng = np.random.default_rng(0)

    def batch():
        return {k: torch.from_numpy(rng.integers(1, hi, nb)).to(dev) for k, hi in (
            ("user_id", nU), ("item_id", nI), ("neg_item_id", nI), ("head_id", nE), ("relation_id", nR - 1),
            ("tail_id", nE), ("neg_tail_id", nE))}

    batches = [batch() for _ in range(6)]
    opt_e = torch.optim.Adam(eager.parameters(), lr=1e-3, capturable=True)
    opt_g = torch.optim.Adam(graphed.parameters(), lr=1e-3, capturable=True)
    before = {k: v.clone() for k, v in graphed.state_dict().items()}
    step = GraphedTrainStep(graphed, opt_g, batches[0])
    # the constructor's warm-up steps are undone before the capture: parameters and Adam state are what they were
    for k, v in graphed.state_dict().items():
        assert torch.equal(v, before[k]), k
    for st in opt_g.state.values():
        assert float(st["step"]) == 0 and not st["exp_avg"].any() and not st["exp_avg_sq"].any()
    for b in batches:
        opt_e.zero_grad()
        le = eager.calculate_loss(b)
        le.backward()
        opt_e.step()
        lg = step(b)
        np.testing.assert_allclose(float(lg), float(le.detach()), rtol=2e-4, atol=1e-6)
    for (k, a), (_, c) in zip(eager.state_dict().items(), graphed.state_dict().items()):
        np.testing.assert_allclose(c.cpu().numpy(), a.cpu().numpy(), rtol=0, atol=5e-4, err_msg=k)
    with pytest.raises(ValueError):
        GraphedTrainStep(graphed, torch.optim.Adam(graphed.parameters()), batches[0])


def test_graphed_trainer_mixin_equals_the_plain_loop():
    """driver.graphed_trainer(<a Trainer>) -- the two-line drop-in of INTEGRATION.md -- replays the steady-state steps from
    one HIP graph, runs the ragged last batch eagerly, and ends an epoch with the plain loop's loss and parameters.  (Its
    loop is checked against hopwise's own KGTrainer in tests/test_dropin_reference.py, where hopwise is importable.)"""
    import copy

    import hopwise_amd.model as M
    from hopwise_amd.driver import graphed_trainer
    from hopwise_amd.optim import FusedAdam

    dev = torch.device("cuda", 0)
    nU, nI, nE, nR, nb = 300, 200, 900, 7, 512
    torch.manual_seed(5)
    plain_m = M.RotatE(make_config(dev, 64, 1.0), FakeDataset(nU, nI, nE, nR)).to(dev)
    graph_m = copy.deepcopy(plain_m)
    rng = np.random.default_rng(1)

    class Batch(dict):
        def to(self, device):
            return self

    def batch(n):
        return Batch({k: torch.from_numpy(rng.integers(1, hi, n)).to(dev) for k, hi in (
            ("user_id", nU), ("item_id", nI), ("neg_item_id", nI), ("head_id", nE), ("relation_id", nR - 1),
            ("tail_id", nE), ("neg_tail_id", nE))})

    epoch = [batch(nb) for _ in range(5)] + [batch(77)]  # ragged last batch

    class Base:
        def __init__(self, config, model):
            self.config, self.model, self.device, self.clip_grad_norm = config, model, dev, None
            self.optimizer = FusedAdam(model.parameters(), lr=config["learning_rate"])

        def _check_nan(self, loss):
            if torch.isnan(loss):
                raise ValueError("Training loss is nan")

        def _train_epoch(self, train_data, epoch_idx, loss_func=None, show_progress=False):
            self.model.train()
            total = 0.0
            for inter in train_data:
                self.optimizer.zero_grad()
                loss = (loss_func or self.model.calculate_loss)(inter)
                total += loss.item()
                self._check_nan(loss)
                loss.backward()
                self.optimizer.step()
            return total

    cfg = {"learning_rate": 1e-3, "learner": "adam", "single_spec": True}
    plain, graphed = Base(cfg, plain_m), graphed_trainer(Base)(cfg, graph_m)
    assert type(graphed).__name__ == "GraphedBase" and isinstance(graphed.optimizer, FusedAdam)
    for ep in range(2):  # the same trajectory from the first batch on: GraphedTrainStep undoes its warm-up steps
        lp = plain._train_epoch(epoch, ep)
        lg = graphed._train_epoch(epoch, ep)
        np.testing.assert_allclose(lg, lp, rtol=2e-4)
    assert graphed._graph_step is not None
    for (k, a), (_, c) in zip(plain_m.state_dict().items(), graph_m.state_dict().items()):
        np.testing.assert_allclose(c.cpu().numpy(), a.cpu().numpy(), rtol=0, atol=5e-4, err_msg=k)
    # a custom loss function is not captured: the base loop runs
    assert graphed._train_epoch(epoch[:1], 2, loss_func=graph_m.calculate_loss) > 0
