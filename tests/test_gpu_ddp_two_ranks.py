"""The reference's one real parallel mode -- stock DistributedDataParallel, set up exactly as hopwise's trainer does
(trainer.py:79-84: SyncBatchNorm.convert_sync_batchnorm + DistributedDataParallel(model, device_ids=[local_rank]);
trainer.py:94-112, 244-261: set_reduce_hook() through an identity forward, the `sum(p) * 0` sync term, then
model.calculate_loss called directly and (loss + sync_loss).backward()) -- on the HIP models, with the real kernels: two
ranks share the box's one GPU under the gloo backend (RCCL refuses two ranks on one device).

What must hold for a drop-in: every gradient the HIP autograd nodes produce reaches DDP's reducer through AccumulateGrad,
so after backward every rank holds the MEAN of the ranks' gradients -- RotatE: the mean of the two per-rank gradients of a
single-process model; TuckER (BatchNorm -> SyncBatchNorm, statistics over both ranks' rows): the gradient of the single-process
model on the concatenated batch.  An optimizer step then leaves both replicas with the same weights.  FusedAdam's gradient
sinks bypass AccumulateGrad and are refused at world size 2; grad_sinks=False works unchanged."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
NU, NI, NE, NR, B = 300, 200, 500, 7, 256


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _kge_batch(rank):
    g = np.random.default_rng(50 + rank)
    return {k: g.integers(1, hi, B) for k, hi in (
        ("user_id", NU), ("item_id", NI), ("neg_item_id", NI), ("head_id", NE), ("relation_id", NR - 1), ("tail_id", NE),
        ("neg_tail_id", NE))}


def _tucker_batch(rank):
    """TuckER's 1-vs-all target marks the UNION of the batch's positives in every row (tucker.py:111-115): both ranks carry the
    same positive items / tails (in another order, with other users / heads / relations), so that the per-rank targets are the
    concatenated batch's."""
    g0 = np.random.default_rng(7)
    items, tails = g0.integers(1, NI, B), g0.integers(1, NE, B)
    g = np.random.default_rng(70 + rank)
    return {"user_id": g.integers(1, NU, B), "item_id": g.permutation(items), "neg_item_id": g.integers(1, NI, B),
            "head_id": g.integers(1, NE, B), "relation_id": g.integers(1, NR - 1, B), "tail_id": g.permutation(tails),
            "neg_tail_id": g.integers(1, NE, B)}


def _wrap_like_the_reference(model, dev):
    """trainer.py:82-84 + set_reduce_hook (:94-102) + sync_grad_loss (:104-112)."""
    from torch.nn.parallel import DistributedDataParallel

    model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
    ddp = DistributedDataParallel(model, device_ids=[dev.index])

    def set_reduce_hook():
        t = model.forward
        model.forward = lambda x: x
        ddp(torch.LongTensor([0]).to(dev))
        model.forward = t

    def sync_grad_loss():
        sync_loss = 0
        for params in model.parameters():
            sync_loss += torch.sum(params) * 0
        return sync_loss

    return model, ddp, set_reduce_hook, sync_grad_loss


def _worker(rank, world, port, tmp):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    import hopwise_amd.model as M
    from hopwise_amd import _lib
    from hopwise_amd.optim import FusedAdam
    from tests.helpers import FakeDataset, load_tables, make_config, random_tables

    to_dev = lambda it: {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in it.items()}
    try:
        # ---- RotatE (the north-star scorer): fused pair-loss autograd node under DDP ----------------------------------------
        dim = 64
        tables = random_tables("RotatE", NU, NE, NR, dim, seed=4)
        build = lambda: load_tables(M.RotatE(make_config(dev, dim, 1.0), FakeDataset(NU, NI, NE, NR)).to(dev), tables, dev)
        single = build()
        want = {n: torch.zeros_like(p) for n, p in single.named_parameters()}
        losses = []
        for r in range(world):
            single.zero_grad()
            loss = single.calculate_loss(to_dev(_kge_batch(r)))
            loss.backward()
            losses.append(float(loss.detach()))
            for n, p in single.named_parameters():
                want[n] += p.grad / world
        model, ddp, set_reduce_hook, sync_grad_loss = _wrap_like_the_reference(build(), dev)
        assert isinstance(model, M.RotatE)
        with pytest.raises(_lib.HopwiseAmdError):  # sinks would bypass the reducer: refused, not silently wrong
            FusedAdam(model.parameters(), lr=1e-2)
        opt = FusedAdam(model.parameters(), lr=1e-2, grad_sinks=False)
        ref_opt = torch.optim.Adam(single.parameters(), lr=1e-2)
        opt.zero_grad()
        set_reduce_hook()
        sync_loss = sync_grad_loss()
        loss = model.calculate_loss(to_dev(_kge_batch(rank)))
        assert abs(float(loss.detach()) - losses[rank]) <= 1e-6 * abs(losses[rank])
        (loss + sync_loss).backward()
        for n, p in model.named_parameters():
            w = want[n].cpu().numpy()
            np.testing.assert_allclose(p.grad.cpu().numpy(), w, rtol=2e-4, atol=2e-6 * np.abs(w).max() + 1e-9, err_msg=n)
        opt.step()
        single.zero_grad()
        for n, p in single.named_parameters():
            p.grad = want[n].clone()
        ref_opt.step()
        for (n, p), (_, q) in zip(model.named_parameters(), single.named_parameters()):
            np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=1e-4, atol=3e-6, err_msg=n)
        # both replicas hold the same weights after the step
        for n, p in model.named_parameters():
            mine = p.detach().cpu()
            both = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(both, mine)
            assert torch.equal(both[0], both[1]), n

        # ---- TuckER (configs[4]): GEMM + contraction + BCE nodes, BatchNorm1d -> SyncBatchNorm ------------------------------
        dim = 16
        cfg = dict(input_dropout=0.0, input_dropout1=0.0, input_dropout2=0.0, label_smoothing=0.1)
        torch.manual_seed(3)
        proto = M.TuckER(make_config(dev, dim, 1.0, **cfg), FakeDataset(NU, NI, NE, NR)).to(dev)
        state = {k: v.clone() for k, v in proto.state_dict().items()}

        def build_t():
            m = M.TuckER(make_config(dev, dim, 1.0, **cfg), FakeDataset(NU, NI, NE, NR)).to(dev)
            m.load_state_dict(state)
            return m.train()

        single = build_t()
        cat = {k: np.concatenate([_tucker_batch(r)[k] for r in range(world)]) for k in _tucker_batch(0)}
        loss_cat = single.calculate_loss(to_dev(cat))
        loss_cat.backward()
        model, ddp, set_reduce_hook, sync_grad_loss = _wrap_like_the_reference(build_t(), dev)
        assert isinstance(model.bn0, torch.nn.SyncBatchNorm) and isinstance(model.bn1, torch.nn.SyncBatchNorm)
        set_reduce_hook()
        sync_loss = sync_grad_loss()
        loss = model.calculate_loss(to_dev(_tucker_batch(rank)))
        (loss + sync_loss).backward()
        lsum = torch.tensor([float(loss.detach())], dtype=torch.float64)
        dist.all_reduce(lsum)
        assert abs(float(lsum) / world - float(loss_cat.detach())) <= 2e-5 * abs(float(loss_cat.detach()))
        for (n, p), (_, q) in zip(model.named_parameters(), single.named_parameters()):
            w = q.grad.cpu().numpy()
            np.testing.assert_allclose(p.grad.cpu().numpy(), w, rtol=5e-4, atol=5e-6 * np.abs(w).max() + 1e-9, err_msg=n)
        # SyncBatchNorm's running statistics: those of the concatenated batch
        np.testing.assert_allclose(model.bn0.running_mean.cpu().numpy(), single.bn0.running_mean.cpu().numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(model.bn1.running_var.cpu().numpy(), single.bn1.running_var.cpu().numpy(), rtol=1e-3, atol=1e-6)
        torch.cuda.synchronize()
        open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_reference_style_ddp_on_the_hip_models(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")
