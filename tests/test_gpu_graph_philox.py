"""Philox offsets that live on the device (csrc/hw_common.h: philox_offset; hopwise_amd/philox.py) and the training steps they
make capturable: TuckER's three dropouts and PGPR's policy dropout / action dropout / action draws inside driver.GraphedTrainStep.

The reference draws these from torch's generator (tucker.py:53-58 nn.Dropout; pgpr.py:180-187 F.dropout, :152-173
Categorical.sample / batch_action_mask), which advances on every call; a captured step must do the same.  Checked here:
an indirect offset names exactly the stream position the plain one does, kernel by kernel; replays draw a NEW mask each time
and the SAME masks as eager steps (graph replays == the eager loop, step for step); the host call counts follow, so a
checkpoint (other_parameter) resumes the stream; eager steps between replays keep the two in step."""

import copy

import numpy as np
import pytest
import torch

from tests.helpers import FakeDataset, make_config

pytestmark = pytest.mark.gpu
INDIRECT = 1 << 63


def dev():
    return torch.device("cuda", 0)


def _indirect(word, addend):
    return INDIRECT | (addend << 48) | word.data_ptr()


def test_an_indirect_offset_is_the_plain_offset_kernel_by_kernel():
    from hopwise_amd import _lib, ops
    from hopwise_amd import pgpr_ops as P

    seed, plain = 0x1234ABCD, 1000
    word = torch.tensor([plain - 5], dtype=torch.int64, device=dev())
    ind = _indirect(word, 5)
    g = torch.Generator(device="cpu").manual_seed(0)
    # hw_dropout
    x = torch.randn(10_001, generator=g).to(dev())
    assert torch.equal(ops._DropoutFn.apply(x, 0.3, seed, plain), ops._DropoutFn.apply(x, 0.3, seed, ind))
    # the GEMM's Philox epilogue (drop_mode 2), tile and split-K forms
    for m_rows, n, k in ((300, 512, 64), (40, 256, 1024)):
        a, w, b = torch.randn(m_rows, k, generator=g).to(dev()), torch.randn(n, k, generator=g).to(dev()), torch.randn(n, generator=g).to(dev())
        n_rows = torch.tensor([m_rows], dtype=torch.int32, device=dev())
        o1, o2 = torch.empty(m_rows, n, device=dev()), torch.empty(m_rows, n, device=dev())
        P.mlp_layer(a, w, b, o1, n_rows, m_rows, True, None, 2, seed, plain)
        P.mlp_layer(a, w, b, o2, n_rows, m_rows, True, None, 2, seed, ind)
        assert torch.equal(o1, o2) and 0.3 < float((o1 == 0).float().mean()) < 0.7
        # ... and the backward that regenerates the mask
        dy = torch.randn(m_rows, n, generator=g).to(dev())
        d1 = P.act_dropout_backward(dy, o1, 1, None, 2, seed, plain)
        d2 = P.act_dropout_backward(dy, o1, 1, None, 2, seed, ind)
        assert torch.equal(d1, d2)
    # TuckER's contraction with hidden_dropout1 inside (forward, and backward through autograd)
    B, d = 37, 24
    xs = [torch.randn(B, d, generator=g).to(dev()).requires_grad_() for _ in range(2)]
    ws = [torch.randn(B, d, d, generator=g).to(dev()).requires_grad_() for _ in range(2)]
    ws[1].data.copy_(ws[0].data), xs[1].data.copy_(xs[0].data)
    outs = [ops.tucker_contract_dropout(xs[i], ws[i], 0.4, seed, off) for i, off in enumerate((plain, ind))]
    assert torch.equal(outs[0], outs[1])
    go = torch.randn(B, d, generator=g).to(dev())
    for o in outs:
        o.backward(go)
    assert torch.equal(xs[0].grad, xs[1].grad) and torch.equal(ws[0].grad, ws[1].grad)
    mats = torch.randn(3, d, d, generator=g).to(dev())
    index = torch.from_numpy(np.random.default_rng(1).integers(0, 3, B)).to(dev())
    s1 = ops.tucker_contract_shared(xs[0].detach(), mats, index, None, 0.4, seed, plain)
    s2 = ops.tucker_contract_shared(xs[0].detach(), mats, index, None, 0.4, seed, ind)
    assert torch.equal(s1, s2)
    # PGPR's action dropout and action draws
    rows, A = 50, 64
    n_acts = torch.from_numpy(np.random.default_rng(2).integers(1, A + 1, rows).astype(np.int32)).to(dev())
    n_rows = torch.tensor([rows], dtype=torch.int32, device=dev())
    masks = [torch.empty((rows, A), dtype=torch.uint8, device=dev()) for _ in range(2)]
    P.action_dropout(n_acts, n_rows, rows, A, 0.5, seed, plain, masks[0])
    P.action_dropout(n_acts, n_rows, rows, A, 0.5, seed, ind, masks[1])
    assert torch.equal(masks[0], masks[1]) and int(masks[0].sum()) < int(n_acts.sum())
    head = torch.randn(rows, A + 1, generator=g).to(dev())
    res = []
    for off in (plain, ind):
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev())
        probs, acts = f(rows, A), torch.empty(rows, dtype=torch.int32, device=dev())
        logp, ent, val = f(rows), f(rows), f(rows)
        P.policy_sample(head, A, masks[0], n_rows, rows, probs, acts, logp, ent, val, seed, off)
        res.append((acts, logp))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    # the stand-alone negative sampler
    used_ptr = torch.arange(0, 41, 4, dtype=torch.int64, device=dev())
    used_idx = torch.from_numpy(np.sort(np.random.default_rng(3).integers(1, 500, (10, 4)), axis=1).reshape(-1)).to(dev())
    keys = torch.arange(10, dtype=torch.int64, device=dev())
    assert torch.equal(ops.neg_sample(used_ptr, used_idx, keys, 33, 500, seed, plain), ops.neg_sample(used_ptr, used_idx, keys, 33, 500, seed, ind))
    # the word moves, the draw moves
    word.add_(1)
    assert not torch.equal(ops._DropoutFn.apply(x, 0.3, seed, plain), ops._DropoutFn.apply(x, 0.3, seed, ind))
    assert torch.equal(ops._DropoutFn.apply(x, 0.3, seed, plain + 1), ops._DropoutFn.apply(x, 0.3, seed, ind))
    # entry points that do not resolve it say so instead of drawing from a wrong stream
    import ctypes as C

    lib = _lib.load()
    tabs = (C.c_void_p * 9)(*([16] * 9))
    assert lib.hw_pgpr_policy_fused(tabs, 16, 16, 512, 16, 16, 16, 251, 1, 16, 16, 1, 3, None, None, seed, ind, 0, 16, 256,
                                    None) == -2  # HW_ERR_UNSUPPORTED, before anything is launched


def test_a_captured_dropout_draws_a_new_mask_on_every_replay_and_the_eager_stream():
    """One ops.Dropout inside a raw HIP graph, the protocol of GraphedTrainStep by hand."""
    from hopwise_amd import ops
    from hopwise_amd.philox import DeviceCounters

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.drop = ops.Dropout(0.5, seed=7, stream=3)

    m, ref = M().train(), M().train()
    x = torch.ones(4096, device=dev())
    m.drop(x), ref.drop(x)  # one eager call first: the capture starts in mid-stream
    dc = DeviceCounters(m, dev())
    out = torch.empty_like(x)
    graph = torch.cuda.CUDAGraph()
    dc.begin_capture()
    with torch.cuda.graph(graph):
        out.copy_(m.drop(x))
        dc.bump_in_graph()
    dc.end_capture()
    assert m.drop.calls == 1 and dc.per_step == [1]
    seen = []
    for i in range(4):
        dc.before_replay()
        graph.replay()
        dc.after_replay()
        want = ref.drop(x)
        assert torch.equal(out, want), i
        seen.append(out.clone())
        if i == 1:  # an eager call in between moves both
            assert torch.equal(m.drop(x), ref.drop(x))
    assert m.drop.calls == ref.drop.calls == 6
    assert not any(torch.equal(seen[i], seen[j]) for i in range(4) for j in range(i))


def _tucker(d=24):
    from hopwise_amd.model import TuckER

    nU, nI, nE, nR = 120, 70, 300, 7
    cfg = make_config(dev(), d, 1.0, label_smoothing=0.1, input_dropout=0.2, input_dropout1=0.4, input_dropout2=0.3)
    torch.manual_seed(3)
    m = TuckER(cfg, FakeDataset(nU, nI, nE, nR)).to(dev())
    with torch.no_grad():
        m.weights.mul_(0.05)
    return m.train(), (nU, nI, nE, nR)


def _tucker_batch(shape, B, seed):
    nU, nI, nE, nR = shape
    rng = np.random.default_rng(seed)
    return {k: torch.from_numpy(v).to(dev()) for k, v in dict(
        user_id=rng.integers(1, nU, B), item_id=rng.integers(1, nI, B), head_id=rng.integers(1, nE, B),
        relation_id=rng.integers(1, nR - 1, B), tail_id=rng.integers(1, nE, B)).items()}


def test_tucker_graph_replays_equal_the_eager_loop():
    from hopwise_amd.driver import GraphedTrainStep
    from hopwise_amd.optim import FusedAdam

    eager, shape = _tucker()
    graphed = copy.deepcopy(eager)
    B = 64
    batches = [_tucker_batch(shape, B, s) for s in range(6)]
    o1, o2 = FusedAdam(eager.parameters(), lr=1e-2), FusedAdam(graphed.parameters(), lr=1e-2)
    step = GraphedTrainStep(graphed, o2, batches[0])
    assert step.verified_no_memset_nodes and len(step.counters.entries) == 3 and step.counters.per_step == [2, 2, 2]
    assert graphed._drop_calls == 0 and graphed._dropout_calls == {"input_dropout": 0, "hidden_dropout2": 0}
    losses_e, losses_g = [], []
    for i, it in enumerate(batches):
        o1.zero_grad()
        loss = eager.calculate_loss(it)
        loss.backward()
        o1.step()
        losses_e.append(float(loss.detach()))
        if i == 3:  # a ragged batch in mid-epoch: eager on both sides, the device words re-synchronise before the next replay
            assert not step.matches(_tucker_batch(shape, B - 3, 99))
            o2.zero_grad()
            l2 = graphed.calculate_loss(it)
            l2.backward()
            o2.step()
            losses_g.append(float(l2.detach()))
        else:
            losses_g.append(float(step(it)))
        assert graphed._drop_calls == eager._drop_calls == 2 * (i + 1)
        assert graphed._dropout_calls == eager._dropout_calls
    # same masks at every step: the losses agree to the summation order of the gradient atomics of earlier steps
    np.testing.assert_allclose(losses_g, losses_e, rtol=2e-4)
    assert len(set(losses_e)) == len(losses_e)
    for (n, p), (_, q) in zip(graphed.named_parameters(), eager.named_parameters()):
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=2e-3, atol=2e-5, err_msg=n)
    # a frozen mask would show: replaying ONE batch gives another loss every time beyond what the weight updates explain --
    # compare with a model whose dropout stream is held still
    frozen = copy.deepcopy(graphed)
    frozen.eval()
    with torch.no_grad():
        e1, e2 = float(frozen.calculate_loss(batches[0])), float(frozen.calculate_loss(batches[0]))
    assert e1 == e2
    # the checkpoint carries the counts (trainer.py:296-304 other_parameter / load_other_parameter)
    fresh, _ = _tucker()
    for name in fresh.other_parameter_name:
        setattr(fresh, name, getattr(graphed, name))
    assert fresh._drop_calls == graphed._drop_calls and fresh._dropout_calls == graphed._dropout_calls


def _pgpr(action_dropout=0.5):
    from tests.test_gpu_pgpr_train import build

    g, z, orc, m = build("a", dropout="reference", action_dropout=action_dropout)
    return g, z, m.train()


def test_pgpr_graph_replays_equal_the_eager_loop():
    """One REINFORCE episode per step with the reference's stochastic policy: dropout p = 0.5 in both hidden layers
    (pgpr.py:180-182), batch_action_mask (:152-165), Categorical draws (:167-173) -- every draw keyed by a device offset."""
    from hopwise_amd.driver import GraphedTrainStep
    from hopwise_amd.optim import FusedAdam

    g, z, eager = _pgpr()
    graphed = _pgpr()[2]
    graphed.load_state_dict(eager.state_dict())
    n_users, B = int(g["n_users"]), 48
    rng = np.random.default_rng(0)
    batches = [{"user_id": torch.from_numpy(rng.integers(1, n_users, B)).to(dev())} for _ in range(5)]
    o1, o2 = FusedAdam(eager.parameters(), lr=1e-3), FusedAdam(graphed.parameters(), lr=1e-3)
    step = GraphedTrainStep(graphed, o2, batches[0])
    assert step.verified_no_memset_nodes and step.counters.per_step == [eager.max_path_len]
    parts_e, parts_g = [], []
    for i, it in enumerate(batches):
        o1.zero_grad()
        loss = eager.calculate_loss(it)
        loss.backward()
        o1.step()
        parts_e.append(eager.last_loss_parts.cpu().numpy().copy())
        step(it)
        parts_g.append(graphed.last_loss_parts.cpu().numpy().copy())
        assert graphed._drop_calls == eager._drop_calls == eager.max_path_len * (i + 1)
    step.check_flags()
    # same trajectories (a different action anywhere changes the rewards, i.e. the loss, at once)
    np.testing.assert_allclose(np.array(parts_g), np.array(parts_e), rtol=1e-5, atol=1e-7)
    assert len({float(p[0]) for p in parts_e}) == len(parts_e)
    for (n, p), (_, q) in zip(graphed.named_parameters(), eager.named_parameters()):
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=1e-4, atol=1e-6, err_msg=n)
    # the [PAD] user cannot be filtered inside a capture: flagged on the device, refused when read
    bad = {"user_id": batches[0]["user_id"].clone()}
    bad["user_id"][3] = 0
    step(bad)
    with pytest.raises(ValueError, match="PAD"):
        step.check_flags()
    step.check_flags()  # cleared when reported
