"""TPRec golden vectors: the reference's TPRec class on the reference's PGPR fixture (tests/test_data/test) with
FABRICATED temporal weights -- the reference's own TPRecTimestampDataset raises KeyError on that fixture (its time span
is shorter than the 90-day window of _structuralWithGap), so uc_weight / timeClassifyLabel are a deterministic function
of (user, item) here.  Stages: "pretrain" (calculate_loss_transe + gradients), then "policy" from that checkpoint
(actions with the per-user temporal vector, beam search, collect_scores, one recorded REINFORCE episode).
Called from make_golden.py ("tprec")."""

import os
import types

import numpy as np
import torch
import torch.nn.functional as F
import yaml

from make_golden_pgpr import RecordingDropout, build, enc_actions, enc_paths, identity_dropout

N_CLUSTERS = 4


def fabricate_temporal(users, items, n_users):
    label = {(int(u), int(i)): int((7 * int(u) + 13 * int(i)) % N_CLUSTERS) for u, i in zip(users, items)}
    ucp = {}
    for (u, i), c in label.items():
        ucp.setdefault(u, {}).setdefault(c, []).append(i)
    ucw = {}
    for u in range(1, n_users):  # _build_interacted_matrix walks uid = 1 .. len(uc_weight): every user needs an entry
        d = ucp.get(u)
        ucw[u] = {u % N_CLUSTERS: 1.0} if d is None else {c: len(v) / sum(len(x) for x in d.values()) for c, v in d.items()}
    return types.SimpleNamespace(uc_weight=ucw, timenum=N_CLUSTERS, timeClassifyLabel=label)


def gen_tprec(out_dir):
    from hopwise.model.knowledge_aware_recommender.tprec import TPRec

    real_dropout = F.dropout
    os.makedirs("/tmp/scratch/saved", exist_ok=True)
    os.chdir("/tmp/scratch")
    with open("/root/reference/hopwise/properties/quick_start_config/knowledge_base_on_ml-100k.yaml") as f:
        constraints = yaml.safe_load(f)["path_constraint"]
    config, ds, _ = build({"max_acts": 30, "path_constraint": constraints})
    iu, ii = ds.inter_feat[ds.uid_field].numpy(), ds.inter_feat[ds.iid_field].numpy()
    tw = fabricate_temporal(iu, ii, ds.user_num)
    ds.temporal_weights = tw
    cfg = dict(config.final_config_dict)
    cfg.update(train_stage="pretrain", margin=1.0, cluster_num=N_CLUSTERS, model="TPRec", dataset="test",
               checkpoint_dir="/tmp/scratch/saved", max_acts=30, action_dropout=0.5, gamma=0.9, weight_factor=0.01)
    out = {"n_clusters": np.int64(N_CLUSTERS)}
    out["uc.users"] = np.array([u for u in tw.uc_weight for _ in tw.uc_weight[u]], np.int64)
    out["uc.clusters"] = np.array([c for u in tw.uc_weight for c in tw.uc_weight[u]], np.int64)
    out["uc.weights"] = np.array([w for u in tw.uc_weight for w in tw.uc_weight[u].values()], np.float64)
    out["label.users"], out["label.items"] = iu.astype(np.int64), ii.astype(np.int64)
    out["label.cluster"] = np.array([tw.timeClassifyLabel[(int(u), int(i))] for u, i in zip(iu, ii)], np.int64)

    # ---- stage 1: pretrain loss
    torch.manual_seed(5)
    pre = TPRec(cfg, ds)
    out["pre.ui_clust_relation_embedding"] = pre.ui_clust_relation_embedding.weight.detach().numpy().copy()
    rng = np.random.default_rng(3)
    nE, nR, nI = ds.entity_num, ds.relation_num, ds.item_num
    # a batch whose interactions all share ONE cluster crashes the reference (tprec.py:313 makes rec_r_e 1-D, :328 cats
    # it with a 2-D tensor), so only mixed batches are recorded
    for tag, sel in (("multi", rng.permutation(len(iu))[:20]), ("multi2", rng.permutation(len(iu))[:33])):
        n = len(sel)
        inter = dict(user_id=iu[sel], item_id=ii[sel], neg_item_id=rng.integers(1, nI, n), head_id=rng.integers(1, nE, 16),
                     relation_id=rng.integers(1, nR - 1, 16), tail_id=rng.integers(1, nE, 16),
                     neg_tail_id=rng.integers(1, nE, 16))
        inter = {k: np.asarray(v, np.int64) for k, v in inter.items()}
        pre.zero_grad()
        loss = pre.calculate_loss({k: torch.from_numpy(v) for k, v in inter.items()})
        loss.backward()
        for k, v in inter.items():
            out[f"pre.{tag}.{k}"] = v
        out[f"pre.{tag}.loss"] = loss.detach().numpy()
        out[f"pre.{tag}.grad"] = pre.ui_clust_relation_embedding.weight.grad.numpy().copy()
        assert all(p.grad is None for n_, p in pre.named_parameters() if "clust" not in n_)
    with torch.no_grad():
        users = torch.tensor([1, 2, 3])
        out["pre.fs"] = pre.full_sort_predict({"user_id": users}).numpy()
    torch.save({"state_dict": pre.state_dict()}, "/tmp/scratch/saved/TPRec-test-pretrained.pth")

    # ---- stage 2: policy
    cfg["train_stage"] = "policy"
    torch.manual_seed(6)
    model = TPRec(cfg, ds)
    for m in (model.l1, model.l2, model.actor, model.critic):
        torch.nn.init.normal_(m.bias, std=0.1)
    for k, v in model.state_dict().items():
        out[f"w.{k}"] = v.numpy().copy()
    out["user_emb"], out["entity_emb"], out["relation_emb"] = model.user_embedding, model.entity_embedding, model.relation_embedding
    out["clust_emb"] = model.ui_clust_relation_embedding
    rel2id = ds.field2token_id["relation_id"]
    out["patterns"] = np.array([[len(p)] + [rel2id.get(n, -2) for n in p] + [-2] * (3 - len(p)) for p in model.patterns], np.int64)
    active = np.unique(iu)
    users = torch.from_numpy(active[:12].astype(np.int64))
    out["users"] = users.numpy()
    identity_dropout()
    model.beam_search_hop = [12, 4, 2]
    paths, probs = model.beam_search(users)
    out["beam.paths"] = enc_paths(paths)
    out["beam.probs"] = np.array(probs, np.float32).reshape(len(paths), -1)
    interacted = model._build_interacted_matrix(tw)
    out["interacted_row"] = np.asarray(interacted[0]).copy()
    assert all(x is interacted[0] for x in interacted)
    scores, expl = model.collect_scores(users, paths, probs, interacted)
    out["collect.scores"] = scores.numpy()
    out["collect.expl_uis"] = np.array([[e[0], e[1]] for e in expl], np.int64).reshape(-1, 2)
    out["collect.expl_score"] = np.array([e[2] for e in expl], np.float64)
    sample = []
    for p in paths[:: max(1, len(paths) // 40)]:
        for L in (1, 2, 3):
            sample.append(p[:L])
    out["prim.paths_flat"] = np.concatenate([enc_paths([p]).reshape(-1, 3) for p in sample])
    out["prim.paths_off"] = np.cumsum([0] + [len(p) for p in sample]).astype(np.int64)
    out["prim.acts_flat"], out["prim.acts_off"] = enc_actions(model._batch_get_actions(sample, False))

    # one recorded REINFORCE episode (as make_golden_pgpr_train.py)
    model.rng = np.random.default_rng(77)
    rec = RecordingDropout(9)
    F.dropout = rec
    masks, acts, rewards, stats, parts = [], [], [], {}, {}
    real_mask, real_select, real_step, real_update = model.batch_action_mask, model.select_action, model.batch_step, model.update

    def mask_wrap(dropout):
        m = real_mask(dropout)
        masks.append(m.copy())
        return m

    def select_wrap(state, mask):
        a = real_select(state, mask)
        acts.append(np.array(a, np.int64))
        return a

    def step_wrap(idx):
        s, r, d = real_step(idx)
        rewards.append(np.array(r, np.float64))
        return s, r, d

    def update_wrap():
        res = real_update()
        parts["v"] = np.array([float(x) for x in res], np.float64)
        return res

    model.batch_action_mask, model.select_action, model.batch_step, model.update = mask_wrap, select_wrap, step_wrap, update_wrap
    torch.manual_seed(8)
    model.zero_grad()
    tr_users = torch.from_numpy(np.concatenate([active[:20], [0]]).astype(np.int64))
    loss = model.calculate_loss({model.USER_ID: tr_users})
    loss.backward()
    F.dropout = real_dropout
    out["train.users"] = tr_users.numpy()
    out["train.masks"], out["train.acts"], out["train.rewards"] = np.stack(masks).astype(np.uint8), np.stack(acts), np.stack(rewards)
    out["train.loss_parts"] = parts["v"]
    for i in range(len(acts)):
        out[f"train.drop1.{i}"], out[f"train.drop2.{i}"] = rec.masks[2 * i], rec.masks[2 * i + 1]
    for k, p in model.named_parameters():
        out[f"train.g.{k}"] = p.grad.numpy().copy()
    out["train.final_paths"] = enc_paths(model._batch_path)
    out["gamma"], out["weight_factor"], out["max_acts"] = np.float64(0.9), np.float64(0.01), np.int64(30)
    np.savez_compressed(os.path.join(out_dir, "tprec.npz"), **out)
    print("tprec", float(out["pre.multi.loss"]), float(out["pre.multi2.loss"]), out["beam.paths"].shape,
          out["collect.scores"].shape, parts["v"], "nonzero rewards", int((out["train.rewards"] > 0).sum()))
