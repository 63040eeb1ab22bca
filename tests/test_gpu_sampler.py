"""Device negative sampler: the properties the reference guarantees (sampler.py:140-183) + distribution checks.

Stream equality with numpy is impossible by construction; what is pinned instead: support,
rejection of used ids, output layout, uniformity / popularity frequencies, error behaviour, determinism per
(seed, call counter)."""

import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def dev():
    return torch.device("cuda", 0)


def kg_dataset(z):
    return types.SimpleNamespace(head_entities=z["heads"], tail_entities=z["tails"], entity_num=int(z["entity_num"]),
                                 head_entity_field="head_id", tail_entity_field="tail_id")


def rec_datasets(z):
    out = []
    for p in ("train", "valid", "test"):
        out.append(types.SimpleNamespace(uid_field="user_id", iid_field="item_id", user_num=int(z["rec.user_num"]),
                                         item_num=int(z["rec.item_num"]),
                                         inter_feat={"user_id": z[f"rec.{p}.users"], "item_id": z[f"rec.{p}.items"]}))
    return out


def test_kg_sampler_properties_on_reference_fixture():
    from hopwise_amd.sampler import KGSampler

    z = np.load(os.path.join(GOLDEN, "sampler.npz"))
    s = KGSampler(kg_dataset(z), "uniform", device=dev(), seed=7)
    used = [set() for _ in range(s.entity_num)]
    for h, t in zip(z["heads"], z["tails"]):
        used[int(h)].add(int(t))
    keys = z["kg.uniform.keys"]
    for num in (1, 3, 64):
        v = s.sample_by_entity_ids(keys, num).cpu().numpy()
        assert v.shape == (len(keys) * num,) and v.dtype == np.int64
        assert v.min() >= 1 and v.max() < s.entity_num
        for j in range(num):
            for i, k in enumerate(keys):
                assert int(v[j * len(keys) + i]) not in used[int(k)]
    # the dense head (entity 7 has used 50 of 59 ids): only the 9 free ids may appear, all of them do, uniformly
    v = s.sample_by_entity_ids(np.full(9000, 7), 1).cpu().numpy()
    free = sorted(set(range(1, s.entity_num)) - used[7])
    cnt = np.array([(v == f).sum() for f in free])
    assert cnt.sum() == 9000
    exp = 9000 / len(free)
    assert ((cnt - exp) ** 2 / exp).sum() < 40  # chi-square, 8 dof: P(>40) ~ 3e-6
    with pytest.raises(ValueError):
        s.sample_by_entity_ids(np.array([s.entity_num + 5]), 1)


def test_sampler_determinism_and_streams():
    from hopwise_amd import ops

    ptr = torch.zeros(11, dtype=torch.int64, device=dev())
    idx = torch.zeros(0, dtype=torch.int64, device=dev())
    keys = torch.arange(1, 9, device=dev())
    a = ops.neg_sample(ptr, idx, keys, 5, 1000, seed=3, offset=1)
    b = ops.neg_sample(ptr, idx, keys, 5, 1000, seed=3, offset=1)
    c = ops.neg_sample(ptr, idx, keys, 5, 1000, seed=3, offset=2)
    assert torch.equal(a, b) and not torch.equal(a, c)
    big = ops.neg_sample(ptr, idx, torch.ones(200000, dtype=torch.int64, device=dev()), 1, 101, seed=1, offset=0).cpu().numpy()
    cnt = np.bincount(big, minlength=101)[1:]
    assert cnt.min() > 0 and ((cnt - 2000.0) ** 2 / 2000.0).sum() < 180  # 99 dof


def test_rec_sampler_phases_and_popularity():
    from hopwise_amd.sampler import Sampler

    z = np.load(os.path.join(GOLDEN, "sampler.npz"))
    ds = rec_datasets(z)
    phases = ["train", "valid", "test"]
    s = Sampler(phases, ds, "uniform", device=dev())
    cum = [set() for _ in range(ds[0].user_num)]
    for p, d in zip(phases, ds):
        for u, i in zip(d.inter_feat["user_id"], d.inter_feat["item_id"]):
            cum[int(u)].add(int(i))
        sp = s.set_phase(p)
        keys = z[f"rec.uniform.{p}.keys"]
        v = sp.sample_by_user_ids(keys, None, 40).cpu().numpy()
        for j in range(40):
            for i, k in enumerate(keys):
                x = int(v[j * len(keys) + i])
                assert 1 <= x < ds[0].item_num and x not in cum[int(k)]
    with pytest.raises(ValueError):
        s.set_phase("nope")
    with pytest.raises(ValueError):
        s.set_phase("train").sample_by_user_ids(np.array([-1]), None, 1)
    # popularity: frequencies follow count^alpha over ids that occur, restricted to the key's free ids
    alpha = 0.5
    sp = Sampler(phases, ds, "popularity", alpha=alpha, device=dev()).set_phase("train")
    items = np.concatenate([d.inter_feat["item_id"] for d in ds])
    ids, counts = np.unique(items, return_counts=True)
    w = np.power(counts / len(items), alpha)
    key = int(ds[0].inter_feat["user_id"][0])
    used = {int(i) for u, i in zip(ds[0].inter_feat["user_id"], ds[0].inter_feat["item_id"]) if int(u) == key}
    allowed = np.array([i not in used for i in ids])
    p = np.where(allowed, w, 0.0)
    p /= p.sum()
    n = 120000
    v = sp.sample_by_user_ids(np.full(n, key), None, 1).cpu().numpy()
    got = np.array([(v == i).sum() for i in ids]) / n
    assert set(np.unique(v)) <= set(ids[allowed].tolist())
    np.testing.assert_allclose(got, p, atol=4 * np.sqrt(p.max() / n) + 1e-3)


def test_sampler_rejects_saturated_key():
    from hopwise_amd.sampler import KGSampler

    n = 6
    heads = np.array([2, 2, 2, 2, 2, 3])
    tails = np.array([1, 2, 3, 4, 5, 1])
    with pytest.raises(ValueError):
        KGSampler(types.SimpleNamespace(head_entities=heads, tail_entities=tails, entity_num=n), device=dev())


def test_bitmap_fast_path_equals_generic_kernel(monkeypatch):
    from hopwise_amd.data import synthetic

    """Grouped layout, many draws per key: the LDS-bitmap kernel and the binary-search kernel draw the same ids."""
    from hopwise_amd import ops

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(1)
    n_users, n_items = 300, 2985
    iu, ii = rng.integers(1, n_users, 40000), rng.integers(1, n_items, 40000)
    ptr, idx = synthetic.used_csr(iu, ii, n_users)
    ptr, idx = torch.from_numpy(ptr).to(dev), torch.from_numpy(idx).to(dev)
    keys = torch.from_numpy(rng.integers(1, n_users, 77)).to(dev)
    outs = []
    for knob in ("1", "0"):
        monkeypatch.setenv("HW_SAMPLER_BITMAP", knob)
        buf = torch.zeros((77, 1 + 300), dtype=torch.int64, device=dev)
        ops.neg_sample(ptr, idx, keys, 300, n_items, seed=9, offset=4, out=buf[:, 1:])
        outs.append(buf.cpu().numpy())
    np.testing.assert_array_equal(outs[0], outs[1])
    used = {(int(u), int(i)) for u, i in zip(iu, ii)}
    k = keys.cpu().numpy()
    assert all((int(k[r]), int(v)) not in used for r in range(77) for v in outs[0][r, 1:])
    assert outs[0][:, 1:].min() >= 1 and outs[0][:, 1:].max() < n_items


@pytest.mark.parametrize("model,dim", [("RotatE", 200), ("TransE", 64), ("ComplEx", 100), ("DistMult", 24)])
def test_scorer_with_in_kernel_sampler_equals_sampler_then_scorer(model, dim):
    """hw_kge_score_grouped_sampled == hw_neg_sample followed by hw_kge_score_grouped: ids and scores, bit for bit."""
    from hopwise_amd import ops
    from hopwise_amd.data import synthetic
    from tests.helpers import random_tables

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    nU, nI, nE, nR, B, K = 200, 2985, 4000, 7, 93, 130
    iu, ii = rng.integers(1, nU, 30000), rng.integers(1, nI, 30000)
    ptr, idx = synthetic.used_csr(iu, ii, nU)
    ptr, idx = torch.from_numpy(ptr).to(dev), torch.from_numpy(idx).to(dev)
    tabs = {k: torch.from_numpy(v).to(dev) for k, v in random_tables(model, nU, nE, nR, dim, seed=1).items()}
    from oracle import kge as okge

    hn, rn, tn = okge.slots(model, "rec")
    desc = ops.make_desc(model, tuple(tabs[n] for n in hn), tuple(tabs[n] for n in rn), tuple(tabs[n] for n in tn),
                         margin=1.0, tail_rows=nI)
    users = torch.from_numpy(rng.integers(1, nU, B)).to(dev)
    pos = torch.from_numpy(rng.integers(1, nI, B)).to(dev)
    a = torch.zeros((B, 1 + K), dtype=torch.int64, device=dev)
    a[:, 0] = pos
    b = a.clone()
    oob_before = ops.oob_count(dev)  # cumulative per device: other tests provoke it on purpose
    ops.neg_sample(ptr, idx, users, K, nI, seed=11, offset=5, out=a[:, 1:])
    ref = ops.kge_score_grouped_raw(desc, users, None, nR - 1, a)
    got = ops.kge_score_grouped_sampled_raw(desc, users, None, nR - 1, b, ptr, idx, users, nI, 11, 5)
    np.testing.assert_array_equal(b.cpu().numpy(), a.cpu().numpy())
    np.testing.assert_array_equal(got.cpu().numpy(), ref.cpu().numpy())
    assert ops.oob_count(dev) == oob_before


def test_in_kernel_sampler_on_a_value_space_beyond_the_lds_bitmap():
    """value_num > 65 536 (Yelp2018's 90 962 entities; the HBM-resident roofline table): the fused kernels test membership by
    binary search in the key's sorted used slice instead of an LDS bitmap -- still the ids hw_neg_sample writes, bit for bit,
    with and without the fused loss."""
    from hopwise_amd import ops
    from hopwise_amd.data import synthetic
    from tests.helpers import random_tables

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    nU, nE, nR, B, K, d = 300, 90962, 7, 257, 256, 64
    # dense used sets on a narrow id range so that rejections actually happen in a 90 k space
    iu = rng.integers(1, nU, 60000)
    ii = np.where(rng.random(60000) < 0.5, rng.integers(1, 400, 60000), rng.integers(1, nE, 60000))
    ptr, idx = synthetic.used_csr(iu, ii, nU)
    ptr, idx = torch.from_numpy(ptr).to(dev), torch.from_numpy(idx).to(dev)
    tabs = {k: torch.from_numpy(v).to(dev) for k, v in random_tables("RotatE", nU, nE, nR, d, seed=2).items()}
    from oracle import kge as okge

    hn, rn, tn = okge.slots("RotatE", "rec")
    desc = ops.make_desc("RotatE", tuple(tabs[n] for n in hn), tuple(tabs[n] for n in rn), tuple(tabs[n] for n in tn), margin=1.0)
    users = torch.from_numpy(rng.integers(1, nU, B)).to(dev)
    a = torch.zeros((B, 1 + K), dtype=torch.int64, device=dev)
    a[:, 0] = torch.from_numpy(rng.integers(1, nE, B)).to(dev)
    b, c = a.clone(), a.clone()
    ops.neg_sample(ptr, idx, users, K, nE, seed=21, offset=9, out=a[:, 1:])
    ref = ops.kge_score_grouped_raw(desc, users, None, nR - 1, a)
    got = ops.kge_score_grouped_sampled_raw(desc, users, None, nR - 1, b, ptr, idx, users, nE, 21, 9)
    np.testing.assert_array_equal(b.cpu().numpy(), a.cpu().numpy())
    np.testing.assert_array_equal(got.cpu().numpy(), ref.cpu().numpy())
    got2, loss, _ = ops.kge_score_grouped_sampled_selfadv_raw(desc, users, None, nR - 1, c, ptr, idx, users, nE, 21, 9, alpha=1.0)
    np.testing.assert_array_equal(c.cpu().numpy(), a.cpu().numpy())
    np.testing.assert_array_equal(got2.cpu().numpy(), ref.cpu().numpy())
    np.testing.assert_allclose(loss.item(), ops.self_adversarial_mean(ref, 1.0).item(), rtol=1e-6)
    # the draws avoided the used sets
    used = set(zip(iu.tolist(), ii.tolist()))
    av, uv = a.cpu().numpy(), users.cpu().numpy()
    assert all((int(uv[r]), int(v)) not in used for r in range(B) for v in av[r, 1:])


@pytest.mark.parametrize("model,dim,B,K,alpha", [("RotatE", 200, 2048, 256, 1.0), ("TransE", 64, 93, 130, 0.5),
                                                 ("ComplEx", 100, 7, 511, 2.0), ("DistMult", 24, 300, 127, 1.0)])
def test_sampler_scorer_and_loss_in_one_launch(model, dim, B, K, alpha):
    """hw_kge_score_grouped_sampled_selfadv == hw_kge_score_grouped_sampled + hw_loss_self_adversarial: ids, scores and the
    loss bit for bit (the block that scores a group reduces it; the last block sums the groups in the finish kernel's order),
    twice in a row."""
    from hopwise_amd import ops
    from hopwise_amd.data import synthetic
    from oracle import kge as okge
    from tests.helpers import random_tables

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(B + K)
    nU, nI, nE, nR = 200, 2985, 4000, 7
    iu, ii = rng.integers(1, nU, 30000), rng.integers(1, nI, 30000)
    ptr, idx = synthetic.used_csr(iu, ii, nU)
    ptr, idx = torch.from_numpy(ptr).to(dev), torch.from_numpy(idx).to(dev)
    tabs = {k: torch.from_numpy(v).to(dev) for k, v in random_tables(model, nU, nE, nR, dim, seed=1).items()}
    hn, rn, tn = okge.slots(model, "rec")
    desc = ops.make_desc(model, tuple(tabs[n] for n in hn), tuple(tabs[n] for n in rn), tuple(tabs[n] for n in tn),
                         margin=1.0, tail_rows=nI)
    users = torch.from_numpy(rng.integers(1, nU, B)).to(dev)
    a = torch.zeros((B, 1 + K), dtype=torch.int64, device=dev)
    a[:, 0] = torch.from_numpy(rng.integers(1, nI, B)).to(dev)
    b = a.clone()
    for offset in (5, 6):
        ref = ops.kge_score_grouped_sampled_raw(desc, users, None, nR - 1, a, ptr, idx, users, nI, 11, offset)
        ref_loss = ops.self_adversarial_mean(ref, alpha)
        got, loss, counter = ops.kge_score_grouped_sampled_selfadv_raw(desc, users, None, nR - 1, b, ptr, idx, users, nI, 11,
                                                                        offset, alpha)
        assert torch.equal(b, a) and torch.equal(got, ref)
        assert loss.item() == ref_loss.item(), (loss.item(), ref_loss.item())
    # against the oracle as well
    np.testing.assert_allclose(loss.item(), okge.self_adversarial_loss(ref.cpu().numpy(), alpha), rtol=1e-5)


def test_chained_losses_equal_the_unchained_ones():
    """hw_kge_score_grouped_sampled_selfadv_chained: batch i's group terms are added by block 0 of batch i + 1, the last
    batch's by hw_selfadv_finish -- ids, scores and every loss equal the unchained launches' bit for bit."""
    from hopwise_amd import ops
    from hopwise_amd.data import synthetic
    from oracle import kge as okge
    from tests.helpers import random_tables

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(77)
    model, dim, B, K, alpha = "RotatE", 40, 96, 255, 1.0
    nU, nI, nE, nR = 200, 2985, 4000, 7
    iu, ii = rng.integers(1, nU, 30000), rng.integers(1, nI, 30000)
    ptr, idx = synthetic.used_csr(iu, ii, nU)
    ptr, idx = torch.from_numpy(ptr).to(dev), torch.from_numpy(idx).to(dev)
    tabs = {k: torch.from_numpy(v).to(dev) for k, v in random_tables(model, nU, nE, nR, dim, seed=1).items()}
    hn, rn, tn = okge.slots(model, "rec")
    desc = ops.make_desc(model, tuple(tabs[n] for n in hn), tuple(tabs[n] for n in rn), tuple(tabs[n] for n in tn),
                         margin=1.0, tail_rows=nI)
    batches = []
    for _ in range(5):
        users = torch.from_numpy(rng.integers(1, nU, B)).to(dev)
        t = torch.zeros((B, 1 + K), dtype=torch.int64, device=dev)
        t[:, 0] = torch.from_numpy(rng.integers(1, nI, B)).to(dev)
        batches.append((users, t))
    want = []
    for i, (users, t) in enumerate(batches):
        got, loss, _ = ops.kge_score_grouped_sampled_selfadv_raw(desc, users, None, nR - 1, t.clone(), ptr, idx, users, nI, 11,
                                                                 i, alpha)
        want.append((got, loss.item()))
    chain = ops.SelfAdvChain(B, dev)
    outs = []
    for i, (users, t) in enumerate(batches):
        tt = t.clone()
        loss = torch.full((1,), float("nan"), device=dev)
        outs.append(chain.launch(desc, users, None, nR - 1, tt, ptr, idx, users, nI, 11, i, alpha, loss=loss))
    assert np.isnan(outs[-1][1].item())  # the last batch is not finished before the flush ...
    assert outs[-2][1].item() == want[-2][1]  # ... the one before it is
    chain.flush()
    for (got, loss), (ref, ref_loss) in zip(outs, want):
        assert torch.equal(got, ref)
        assert loss.item() == ref_loss
    chain.flush()  # nothing pending: no launch
    with pytest.raises(ValueError):
        chain.launch(desc, batches[0][0][:8], None, nR - 1, batches[0][1][:8].clone(), ptr, idx, batches[0][0][:8], nI, 11, 0)
