"""PGPR training on the device: replay of the reference's recorded episode (golden + oracle), the device's own random
decisions against their numpy definitions, and calculate_loss / backward through the trainer-facing API."""

import numpy as np
import pytest
import torch

from oracle import pgpr_train as opt
from tests.helpers import FakeKGDataset, pgpr_config
from tests.test_gpu_pgpr import paths_to_records
from tests.test_oracle_golden_pgpr_train import load_train_oracle

pytestmark = pytest.mark.gpu
PARAMS = ("l1.weight", "l1.bias", "l2.weight", "l2.bias", "actor.weight", "actor.bias", "critic.weight", "critic.bias")


def dev():
    return torch.device("cuda", 0)


def build(tag, dropout="identity", action_dropout=None):
    from hopwise_amd.model.knowledge_aware_recommender import PGPR

    g, z, orc = load_train_oracle(tag, tie_rule="stable")
    ds = FakeKGDataset(int(g["n_users"]), int(g["n_items"]), int(g["n_entities"]), int(g["n_relations"]),
                       int(g["ui_relation"]), g["inter_users"], g["inter_items"], g["heads"], g["rels"], g["tails"],
                       g["positives"], g["user_emb"], g["entity_emb"], g["relation_emb"])
    cfg = pgpr_config(dev(), int(z["max_acts"]), 1, (25, 5, 1), False, (10,), dropout)
    cfg.update(gamma=float(z["gamma"]), weight_factor=float(z["weight_factor"]),
               action_dropout=float(z["action_dropout"]) if action_dropout is None else action_dropout)
    m = PGPR(cfg, ds).to(dev())
    m.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w.")})
    m._patterns = torch.from_numpy(z["patterns"].astype(np.int32)).to(dev())
    return g, z, orc, m


def grads_of(m, ep):
    return dict(zip(PARAMS, (x.cpu().numpy() for x in m._episode_backward(ep))))


def check_against_steps(ep, steps, paths, parts, grads, m, tol=2e-5):
    T, B = ep["T"], ep["B"]
    np.testing.assert_array_equal(ep["paths"][T].cpu().numpy()[:, :11], paths_to_records(paths)[:, :11])
    raw = ep["raw_rewards"].cpu().numpy().reshape(T, B)
    for t, s in enumerate(steps):
        sl = slice(t * B, (t + 1) * B)
        np.testing.assert_array_equal(ep["n_acts"][sl].cpu().numpy(), s["n_acts"])
        np.testing.assert_allclose(raw[t], s["reward"], rtol=1e-12, atol=0)
        np.testing.assert_allclose(ep["probs"][sl].cpu().numpy(), s["probs"], rtol=5e-5, atol=1e-7)
        np.testing.assert_allclose(ep["logp"][sl].cpu().numpy(), s["logp"], rtol=5e-5, atol=5e-6)
        np.testing.assert_allclose(ep["value"][sl].cpu().numpy(), s["value"], rtol=5e-5, atol=5e-6)
        np.testing.assert_allclose(ep["entropy"][sl].cpu().numpy(), s["entropy"], rtol=5e-5, atol=5e-6)
    np.testing.assert_allclose(ep["loss"].cpu().numpy(), parts, rtol=2e-5, atol=2e-6)
    got = grads_of(m, ep)
    for k in PARAMS:
        assert np.abs(got[k] - grads[k]).max() <= tol * max(1.0, np.abs(grads[k]).max()), k


@pytest.mark.parametrize("tag", ["a", "b"])
def test_replay_reference_episode(tag):
    g, z, orc, m = build(tag)
    users = z["users"][z["users"] != 0]
    T = z["acts"].shape[0]
    drops_np = [(z[f"drop1.{t}"], z[f"drop2.{t}"]) for t in range(T)]
    steps, paths = orc.episode(users, lambda t, probs, mask: z["acts"][t], masks=z["masks"], drops=drops_np)
    parts, grads = orc.update(steps)
    d = dev()
    ep = m.episode_device(torch.from_numpy(users).to(d),
                          forced_acts=torch.from_numpy(z["acts"].astype(np.int32)).to(d),
                          act_masks=torch.from_numpy(z["masks"]).to(d),
                          drops=[tuple(torch.from_numpy(x).to(d) for x in dd) for dd in drops_np])
    check_against_steps(ep, steps, paths, parts, grads, m)
    if tag == "b":  # no trimming ties at max_acts = 250: the device must also match the reference run itself
        np.testing.assert_allclose(ep["loss"].cpu().numpy(), z["loss_parts"], rtol=2e-5, atol=2e-6)
        np.testing.assert_array_equal(ep["paths"][T].cpu().numpy()[:, :11],
                                      paths_to_records([[tuple(int(v) for v in n) for n in p] for p in z["final_paths"]])[:, :11])
        got = grads_of(m, ep)
        for k in PARAMS:
            want = z[f"g.{k}"]
            assert np.abs(got[k] - want).max() <= 3e-5 * max(1.0, np.abs(want).max()), k


def test_device_random_decisions_match_their_definitions():
    g, z, orc, m = build("a", dropout="identity", action_dropout=0.5)
    users = z["users"][z["users"] != 0]
    ep = m.episode_device(torch.from_numpy(users).to(dev()))
    T, B, A = ep["T"], ep["B"], m.act_dim
    masks = ep["mask"].cpu().numpy().reshape(T, B, A)
    acts = ep["acts"].cpu().numpy().reshape(T, B)
    probs = ep["probs"].cpu().numpy().reshape(T, B, A)
    n_acts = ep["n_acts"].cpu().numpy().reshape(T, B)
    for t, call in enumerate(ep["calls"]):
        np.testing.assert_array_equal(masks[t], opt.dropout_keep_mask(n_acts[t], A, 0.5, m._seed, 4 * call + 2))
        want, _ = opt.inverse_cdf_draw(probs[t], m._seed, 4 * call + 3)
        np.testing.assert_array_equal(acts[t], want)
        assert (masks[t][np.arange(B), acts[t]] == 1).all()
    assert (n_acts >= 5).any()
    # the same trajectory through the oracle: everything downstream of the draws agrees
    steps, paths = orc.episode(users, lambda t, p, mk: acts[t], masks=masks)
    parts, grads = orc.update(steps)
    check_against_steps(ep, steps, paths, parts, grads, m)


def test_philox_dropout_backward_uses_the_forward_masks():
    g, z, orc, m = build("b", dropout="reference")
    users = z["users"][z["users"] != 0]
    ep = m.episode_device(torch.from_numpy(users).to(dev()))
    T, B = ep["T"], ep["B"]
    x1, x2 = ep["x1"].cpu().numpy().reshape(T, B, -1), ep["x2"].cpu().numpy().reshape(T, B, -1)
    drops = [((x1[t] != 0) * np.float32(2), (x2[t] != 0) * np.float32(2)) for t in range(T)]
    assert 0.4 < np.mean(drops[0][0] > 0) < 0.6
    acts = ep["acts"].cpu().numpy().reshape(T, B)
    masks = ep["mask"].cpu().numpy().reshape(T, B, -1)
    steps, paths = orc.episode(users, lambda t, p, mk: acts[t], masks=masks, drops=drops)
    parts, grads = orc.update(steps)
    check_against_steps(ep, steps, paths, parts, grads, m)


def test_sampling_distribution():
    """Many draws of one row follow its probabilities (chi-square well inside the 1e-6 tail)."""
    from hopwise_amd import pgpr_ops as P

    d = dev()
    A, n = 23, 200000
    torch.manual_seed(0)
    head = torch.randn(1, A + 1, device=d).repeat(n, 1).contiguous()
    mask = torch.ones(n, A, dtype=torch.uint8, device=d)
    mask[:, 5] = 0
    probs, acts = torch.empty(n, A, device=d), torch.empty(n, dtype=torch.int32, device=d)
    logp, ent, val = (torch.empty(n, device=d) for _ in range(3))
    P.policy_sample(head, A, mask, None, n, probs, acts, logp, ent, val, seed=123, offset=9)
    p = probs[0].double().cpu().numpy()
    cnt = np.bincount(acts.cpu().numpy(), minlength=A)
    assert cnt[5] == 0 and p[5] == 0
    keep = p > 0
    chi2 = (((cnt[keep] - n * p[keep]) ** 2) / (n * p[keep])).sum()
    assert chi2 < 80, chi2  # 21 dof
    np.testing.assert_allclose(logp.cpu().numpy(), np.log(p[acts.cpu().numpy()]), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ent.cpu().numpy(), -(p[keep] * np.log(p[keep])).sum(), rtol=1e-5)


def test_calculate_loss_trains_through_autograd():
    g, z, orc, m = build("b", dropout="reference", action_dropout=0.3)
    users = torch.from_numpy(z["users"]).to(dev())
    opt_ = torch.optim.Adam(m.parameters(), lr=1e-3)
    before = {k: v.detach().clone() for k, v in m.named_parameters()}
    losses = []
    for _ in range(3):
        opt_.zero_grad()
        loss = m.calculate_loss({m.USER_ID: users})
        assert loss.dtype == torch.float64 and loss.requires_grad
        loss.backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
        opt_.step()
        losses.append(float(loss))
        np.testing.assert_allclose(m.last_loss_parts.cpu().numpy()[0], losses[-1])
    assert np.isfinite(losses).all()
    assert all(not torch.equal(before[k], v) for k, v in m.named_parameters())


@pytest.mark.parametrize("m_rows,n,ld", [(6144, 512, 512), (6144, 252, 256), (33, 70, 70), (32, 64, 64), (1, 5, 8),
                                         (100000, 130, 136)])
def test_colsum_two_stage_is_deterministic_and_right(m_rows, n, ld):
    """hw_colsum_f32_ws (bias gradients): row chunks summed in parallel, then the chunks in fixed order."""
    from hopwise_amd import pgpr_ops as P

    g = torch.Generator().manual_seed(m_rows + n)
    buf = torch.randn((m_rows, ld), generator=g).to(dev())
    x = buf[:, :n]
    a = P.colsum(x)
    b = P.colsum(x)
    assert torch.equal(a, b)
    want = x.double().sum(0)
    np.testing.assert_allclose(a.cpu().numpy(), want.cpu().numpy(), rtol=2e-5, atol=1e-5 * float(np.sqrt(m_rows)))


def test_trim_tables_of_a_large_batch_equal_those_of_small_batches():
    """hw_pgpr_user_scores takes the register-tiled float64 kernel from 64 users on: one fma chain over ascending k per output
    in both kernels, so the [users, n_users | n_entities] tables are bit-identical (the action trims compare their entries)."""
    from hopwise_amd import pgpr_ops as P

    g, z, orc, m = build("b", dropout="identity", action_dropout=0.0)
    U64, E64, R64 = m._t64
    rng = np.random.default_rng(7)
    users = torch.from_numpy(rng.integers(1, m.graph.n_users, 333)).to(dev())
    su, se = P.user_scores(m.graph, U64, E64, R64, users)
    for lo in range(0, 333, 50):  # chunks below 64 users: the one-row-per-lane kernel
        a, b = P.user_scores(m.graph, U64, E64, R64, users[lo : lo + 50])
        assert torch.equal(su[lo : lo + 50], a) and torch.equal(se[lo : lo + 50], b)
    want = (U64[users] + R64[m.graph.ui_relation]) @ E64.T  # loose check against a float64 product
    np.testing.assert_allclose(se.cpu().numpy(), want.cpu().numpy(), rtol=1e-12, atol=1e-13)
