"""PGPR golden vectors: the reference's PGPR run on the reference's own fixture (tests/test_data/test,
201 users / 301 items / 6 238 entities / 26 relations, pretrained dim-4 embeddings).  Called from
make_golden.py; see the header there.  Only inputs/outputs are stored."""

import os

import numpy as np
import torch
import torch.nn.functional as F

REL = {"self_loop": -1}
TYPE = {"user": 0, "entity": 1}


def enc_paths(paths):
    if len(paths) == 0:
        return np.zeros((0, 0, 3), np.int64)
    L = len(paths[0])
    out = np.zeros((len(paths), L, 3), np.int64)
    for i, p in enumerate(paths):
        for j, (r, t, n) in enumerate(p):
            out[i, j] = (REL.get(r, r) if isinstance(r, str) else int(r), TYPE[t], int(n))
    return out


def enc_actions(acts_pool):
    flat, off = [], [0]
    for acts in acts_pool:
        for r, n in acts:
            flat.append((REL.get(r, r) if isinstance(r, str) else int(r), int(n)))
        off.append(len(flat))
    return np.array(flat, np.int64).reshape(-1, 2), np.array(off, np.int64)


def build(config_dict):
    from hopwise.config import Config
    from hopwise.data import create_dataset, data_preparation
    from hopwise.utils import get_model, init_seed

    base = {
        "data_path": "/root/reference/tests/test_data/",
        "user_inter_num_interval": [1, 10],
        "device": "cpu",
        "checkpoint_dir": "/tmp/scratch/saved",
    }
    base.update(config_dict)
    config = Config(model="PGPR", dataset="test", config_file_list=["/root/reference/tests/model/test_model.yaml"],
                    config_dict=base)
    init_seed(config["seed"], config["reproducibility"])
    dataset = create_dataset(config)
    train_data, valid_data, test_data = data_preparation(config, dataset)
    ds = train_data._dataset
    model = get_model("PGPR")(config, ds).to("cpu")
    return config, ds, model


def identity_dropout():
    F.dropout = lambda x, p=0.5, training=True, inplace=False: x


class RecordingDropout:
    """Replacement for F.dropout that draws its own seeded keep-mask and remembers it."""

    def __init__(self, seed):
        self.gen = torch.Generator().manual_seed(seed)
        self.masks = []

    def __call__(self, x, p=0.5, training=True, inplace=False):
        keep = (torch.rand(x.shape, generator=self.gen) >= p).float() / (1.0 - p)
        self.masks.append(keep.numpy().copy())
        return x * keep


def gen_pgpr(out_dir):
    real_dropout = F.dropout
    os.makedirs("/tmp/scratch", exist_ok=True)
    os.chdir("/tmp/scratch")
    wrote_graph = False
    for tag, extra, hop_sets in (
        ("default", {}, (("h25_5_1", [25, 5, 1]), ("h4_3_2", [4, 3, 2]))),
        ("trim20", {"max_acts": 20}, (("h20_5_1", [20, 5, 1]), ("h4_3_2", [4, 3, 2]))),
        ("trim20_fix", {"max_acts": 20, "fix_scores_sorting_bug": True}, (("h20_5_1", [20, 5, 1]),)),
        ("trim70", {"max_acts": 70}, (("h64_64_64", [64, 64, 64]),)),
        ("hist2", {"state_history": 2, "max_acts": 50}, (("h25_5_1", [25, 5, 1]),)),
        ("hist0", {"state_history": 0}, (("h25_5_1", [25, 5, 1]),)),
    ):
        config, ds, model = build(extra)
        torch.manual_seed(1234)
        for m in (model.l1, model.l2, model.actor, model.critic):  # non-trivial biases
            torch.nn.init.normal_(m.bias, std=0.1)
        out = {}
        gr = {}
        gr["n_users"], gr["n_items"] = np.int64(ds.user_num), np.int64(ds.item_num)
        gr["n_entities"], gr["n_relations"] = np.int64(ds.entity_num), np.int64(ds.relation_num)
        gr["ui_relation"] = np.int64(model.ui_relation_id)
        gr["inter_users"] = ds.inter_feat[ds.uid_field].numpy().astype(np.int64)
        gr["inter_items"] = ds.inter_feat[ds.iid_field].numpy().astype(np.int64)
        gr["heads"] = np.asarray(ds.head_entities, np.int64).astype(np.int32)
        gr["rels"] = np.asarray(ds.relations, np.int64).astype(np.int32)
        gr["tails"] = np.asarray(ds.tail_entities, np.int64).astype(np.int32)
        gr["user_emb"], gr["entity_emb"], gr["relation_emb"] = (
            model.user_embedding, model.entity_embedding, model.relation_embedding)
        gr["positives"] = model.positives.numpy().astype(np.int64)
        if not wrote_graph:  # the dataset (graph, embeddings, positives) is the same for every variant
            np.savez_compressed(os.path.join(out_dir, "pgpr_graph.npz"), **gr)
            wrote_graph = True
        for k, v in model.state_dict().items():
            out[f"w.{k}"] = v.numpy().copy()
        out["max_acts"] = np.int64(model.max_acts)
        out["state_history"] = np.int64(model.state_history)
        out["fix_scores_sorting_bug"] = np.int64(bool(model.fix_scores_sorting_bug))
        out["topk"] = np.array(config["topk"], np.int64)

        active = np.unique(gr["inter_users"])
        users = torch.from_numpy(np.concatenate([active[:14], [3, 0 if False else active[0]]]).astype(np.int64))
        users = torch.unique(users, sorted=False)
        out["users"] = users.numpy()

        identity_dropout()
        for hname, hops in hop_sets:
            model.beam_search_hop = hops
            paths, probs = model.beam_search(users)
            out[f"beam.{hname}.hops"] = np.array(hops, np.int64)
            out[f"beam.{hname}.paths"] = enc_paths(paths)
            out[f"beam.{hname}.probs"] = np.array(probs, np.float32).reshape(len(paths), -1)
            scores, expl = model.collect_scores(users, paths, probs)
            out[f"collect.{hname}.scores"] = scores.numpy()
            out[f"collect.{hname}.expl_uis"] = np.array([[e[0], e[1]] for e in expl], np.int64).reshape(-1, 2)
            out[f"collect.{hname}.expl_score"] = np.array([e[2] for e in expl], np.float64)
            out[f"collect.{hname}.expl_paths"] = enc_paths([e[3] for e in expl])

        # per-path primitives on a spread of prefixes (1, 2, 3 and 4 nodes long)
        model.beam_search_hop = [6, 4, 2]
        paths, _ = model.beam_search(users)
        sample = []
        for p in paths[:: max(1, len(paths) // 60)]:
            for L in (1, 2, 3, 4):
                sample.append(p[:L])
        # highest-degree tails as well
        deg = sorted(((sum(len(v) for v in rn.values()), t, n) for t in model.graph_dict for n, rn in model.graph_dict[t].items()),
                     reverse=True)[:6]
        for _, t, n in deg:
            sample.append([("self_loop", "user", int(active[1])), (int(model.ui_relation_id), t, n)]
                          if t == "entity" else [("self_loop", "user", int(n))])
        out["prim.paths_flat"] = np.concatenate([enc_paths([p]).reshape(-1, 3) for p in sample])
        out["prim.paths_off"] = np.cumsum([0] + [len(p) for p in sample]).astype(np.int64)
        acts = model._batch_get_actions(sample, False)
        out["prim.acts_flat"], out["prim.acts_off"] = enc_actions(acts)
        out["prim.states"] = model._batch_get_state(sample)
        if tag not in ("default", "trim20"):  # the MLP forward is pinned once per act_dim; keep the others small
            np.savez_compressed(os.path.join(out_dir, f"pgpr_{tag}.npz"), **out)
            print("pgpr", tag, {k: v.shape for k, v in out.items() if k.startswith("beam.") and k.endswith("paths")})
            F.dropout = real_dropout
            continue
        masks = model._batch_acts_to_masks(acts)
        out["prim.masks"] = masks.astype(np.uint8)
        st = torch.FloatTensor(out["prim.states"])
        with torch.no_grad():
            pr, val = model.forward((st, torch.BoolTensor(masks)))
            out["prim.fwd_probs"], out["prim.fwd_values"] = pr.numpy(), val.numpy()
            rec = RecordingDropout(77)
            F.dropout = rec
            pr, val = model.forward((st, torch.BoolTensor(masks)))
            out["prim.drop1"], out["prim.drop2"] = rec.masks
            out["prim.fwd_probs_drop"], out["prim.fwd_values_drop"] = pr.numpy(), val.numpy()
        F.dropout = real_dropout
        np.savez_compressed(os.path.join(out_dir, f"pgpr_{tag}.npz"), **out)
        print("pgpr", tag, {k: v.shape for k, v in out.items() if k.startswith("beam.") and k.endswith("paths")})
