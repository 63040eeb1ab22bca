"""CAFE's training step replayed from HIP graphs (driver.GraphedStagedTrainStep): the host path sampler of the reference runs as
it is on every call, the device half -- symbolic network of the sampled metapath, backward, optimizer -- is one capture per
(metapath, rows).  Replays against the eager loop on a second copy of the model: same sampled batches (one host generator
each, same seed), same dropout streams (Philox counters handed from capture to capture on the device), same parameters."""

import numpy as np
import pytest
import torch

from tests.test_gpu_cafe import build, dev

pytestmark = pytest.mark.gpu


def _pair(tag="default"):
    torch.manual_seed(0)  # the modules of relations the fixture's metapaths do not use keep their random initialisation
    z, _, a = build(tag)
    torch.manual_seed(0)
    _, _, b = build(tag)
    for m in (a, b):
        m.train()
        m.rng = np.random.default_rng(3)
    return z, a, b


def _batches(z, n, size=24):
    users = np.unique(z["inter_users"])
    rng = np.random.default_rng(11)
    out = []
    for i in range(n):
        u = rng.choice(users, size=size if i % 4 else size - 5, replace=False)  # a second batch size now and then
        u[0] = 0  # the padding user the loader can hand over: dropped on the host
        out.append(torch.from_numpy(u).to(dev()))
    return out


def _eager(m, opt, inter):
    opt.zero_grad()
    losses = m.calculate_loss(inter)
    loss = sum(losses)
    loss.backward()
    opt.step()
    return loss.detach()


@pytest.mark.parametrize("optim", ["fused", "torch"])
def test_cafe_step_replayed_per_metapath_equals_the_eager_loop(optim):
    from hopwise_amd.driver import GraphedStagedTrainStep, GraphedTrainStep, make_graphed_step
    from hopwise_amd.optim import FusedAdam
    from hopwise_amd.philox import counters_of

    z, eager, graphed = _pair()
    mk = (lambda m: FusedAdam(m.parameters(), lr=2e-3)) if optim == "fused" else (
        lambda m: torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=2e-3, capturable=True))
    o1, o2 = mk(eager), mk(graphed)
    with pytest.raises(ValueError, match="GraphedStagedTrainStep"):
        GraphedTrainStep(graphed, o2, {graphed.USER_ID: _batches(z, 1)[0]})
    step = make_graphed_step(graphed, o2, None)
    assert isinstance(step, GraphedStagedTrainStep)
    l1, l2 = [], []
    for u in _batches(z, 40):
        l1.append(float(_eager(eager, o1, {eager.USER_ID: u})))
        l2.append(float(step({graphed.USER_ID: u})))
    # more than one metapath and more than one batch size were met, most steps were replays of an existing capture
    keys = list(step.captures)
    assert len({k[0] for k in keys}) >= 2 and len({k[1] for k in keys}) >= 2, keys
    assert step.n_replayed == 40 - step.n_eager and step.n_captured == len(keys) and step.verified_no_memset_nodes
    np.testing.assert_allclose(l2, l1, rtol=2e-5)
    for (n1, p1), (n2, p2) in zip(eager.named_parameters(), graphed.named_parameters()):
        np.testing.assert_allclose(p2.detach().cpu().numpy(), p1.detach().cpu().numpy(), rtol=1e-4, atol=1e-6, err_msg=n1)
    for (n1, b1), (n2, b2) in zip(eager.named_buffers(), graphed.named_buffers()):
        np.testing.assert_allclose(b2.cpu().numpy(), b1.cpu().numpy(), rtol=1e-4, atol=1e-6, err_msg=n1)
    # the dropout streams moved as far in both, per relation module
    assert [(n, c.calls) for n, c in counters_of(eager)] == [(n, c.calls) for n, c in counters_of(graphed)]
    assert any(c.calls for _, c in counters_of(graphed))
    # the host generators were consumed alike
    assert eager.rng.bit_generator.state == graphed.rng.bit_generator.state


def test_cafe_replays_draw_fresh_dropout_masks():
    """The same batch replayed twice from one capture: different masks (the Philox words moved on the device), hence different
    losses; with the counters put back, the same loss again."""
    from hopwise_amd.driver import make_graphed_step
    from hopwise_amd.optim import FusedAdam
    from hopwise_amd.philox import counters_of

    z, _, m = _pair()
    pos = neg = None

    def fixed(users):  # one sampled batch, over and over
        return 0, pos, neg

    mpid, pos, neg = m._get_batch_by_user(np.unique(z["inter_users"])[:24])
    m._get_batch_by_user = lambda users: (mpid, pos, neg)
    opt = FusedAdam(m.parameters(), lr=0.0)  # nothing moves but the dropout streams
    step = make_graphed_step(m, opt, None)
    u = {m.USER_ID: torch.from_numpy(np.unique(z["inter_users"])[:24]).to(dev())}
    a = float(step(u))
    counts = [c.calls for _, c in counters_of(m)]
    b = float(step(u))
    assert step.n_captured == 1 and step.n_replayed == 2 and a != b
    for (_, c), v in zip(counters_of(m), counts):  # an eager interruption (here: putting the counts back) is followed
        c.calls = v
    assert float(step(u)) == b


def test_more_forms_than_kept_captures_drop_the_oldest():
    from hopwise_amd.driver import make_graphed_step
    from hopwise_amd.optim import FusedAdam

    z, eager, graphed = _pair()
    o1, o2 = FusedAdam(eager.parameters(), lr=2e-3), FusedAdam(graphed.parameters(), lr=2e-3)
    step = make_graphed_step(graphed, o2, None)
    step.MAX_CAPTURES = 2
    l1, l2 = [], []
    for u in _batches(z, 12):
        l1.append(float(_eager(eager, o1, {eager.USER_ID: u})))
        l2.append(float(step({graphed.USER_ID: u})))
    assert len(step.captures) == 2 and step.n_captured > 2 and step.n_eager == 0
    np.testing.assert_allclose(l2, l1, rtol=2e-5)


def test_graphed_trainer_mixin_on_cafe_takes_the_loaders_host_batches():
    """driver.graphed_trainer(<a Trainer>) with CAFE: the epoch loop hands the staged step the loader's HOST batch (the path sampler
    wants the ids there), the device half replays per metapath, the epoch ends with the plain loop's loss and parameters."""
    from hopwise_amd.driver import GraphedStagedTrainStep, graphed_trainer
    from hopwise_amd.optim import FusedAdam

    z, plain_m, graph_m = _pair()

    class Batch(dict):
        moved = 0

        def to(self, device):
            Batch.moved += 1
            return Batch({k: v.to(device) for k, v in self.items()})

    epoch = [Batch({plain_m.USER_ID: u.cpu()}) for u in _batches(z, 10)]

    class Base:
        def __init__(self, config, model):
            self.config, self.model, self.device, self.clip_grad_norm = config, model, dev(), None
            self.optimizer = FusedAdam(model.parameters(), lr=config["learning_rate"])

        def _check_nan(self, loss):
            if torch.isnan(loss):
                raise ValueError("Training loss is nan")

        def _train_epoch(self, train_data, epoch_idx, loss_func=None, show_progress=False):
            self.model.train()
            total = 0.0
            for inter in train_data:
                inter = inter.to(self.device)
                self.optimizer.zero_grad()
                losses = (loss_func or self.model.calculate_loss)(inter)
                loss = sum(losses)  # the trainer sums the (reg, rank) tuple (trainer.py:245-249)
                total += loss.item()
                self._check_nan(loss)
                loss.backward()
                self.optimizer.step()
            return total

    cfg = {"learning_rate": 2e-3, "learner": "adam", "single_spec": True}
    plain, graphed = Base(cfg, plain_m), graphed_trainer(Base)(cfg, graph_m)
    for ep in range(2):
        lp = plain._train_epoch(epoch, ep)
        moved = Batch.moved
        lg = graphed._train_epoch(epoch, ep)
        assert Batch.moved == moved  # the replayed loop left the batches on the host
        np.testing.assert_allclose(lg, lp, rtol=2e-5)
    step = graphed._graph_step
    assert isinstance(step, GraphedStagedTrainStep) and step.n_replayed == 20 and step.n_captured == len(step.captures)
    for (k, a), (_, c) in zip(plain_m.state_dict().items(), graph_m.state_dict().items()):
        np.testing.assert_allclose(c.cpu().numpy(), a.cpu().numpy(), rtol=1e-4, atol=1e-6, err_msg=k)
