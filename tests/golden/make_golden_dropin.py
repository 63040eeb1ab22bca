"""Capture the reference's OWN caller around the hot path, in the build container.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden_dropin.py /root/repo/tests/golden

Runs hopwise's Config -> create_dataset -> data_preparation on the reference's `tests/test_data/test` fixture
with the reference's RotatE class and stores, as plain arrays:

  * dataset statistics and the id-space facts the models read (n_users, n_items, n_entities, n_relations,
    ui_relation id);
  * two RSKG training batches exactly as `KnowledgeBasedDataLoader.__next__` yields them
    (knowledge_dataloader.py:137-145; rec side through abstract_dataloader.py:192-198 with sample_num = 4),
    every key, with the reference model's `calculate_loss` on each;
  * the train-phase used-id sets of the rec sampler and the KG sampler (to check the negatives);
  * evaluation batches of `FullSortRecEvalDataLoader` (general_dataloader.py:170-281): the interaction's
    user ids, `history_index`, `positive_u`, `positive_i`, the reference model's `full_sort_predict`, the matrix
    after `scores[:, 0] = -inf; scores[history_index] = -inf` (trainer.py:731-734) and `torch.topk(.., 10)`
    (collector.py:173-183);
  * the same for link prediction (`FullSortLPEvalDataLoader`, `full_sort_predict_kg`) with a split KG.

Only arrays are stored (inputs and expected outputs); no reference source travels.
"""

import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refimport  # noqa: E402

_refimport.install()

import numpy as np  # noqa: E402
import torch  # noqa: E402

REF = "/root/reference"


def _np(t):
    return t.detach().cpu().numpy().copy()


def main(out_dir):
    from hopwise.config import Config
    from hopwise.data import create_dataset, data_preparation
    from hopwise.utils import KGDataLoaderState, get_model, get_trainer, init_seed

    base = {
        "data_path": os.path.join(REF, "tests/test_data"),
        "embedding_size": 16,
        "train_batch_size": 64,
        "eval_batch_size": 4096,
        "train_neg_sample_args": {"distribution": "uniform", "sample_num": 4, "alpha": 1.0, "dynamic": False,
                                  "candidate_num": 0},
        "topk": [10],
        "metrics": ["Recall", "NDCG"],
        "valid_metric": "NDCG@10",
    }
    yaml = os.path.join(REF, "tests/model/test_model.yaml")
    out = {}
    for tag, extra in (("rec", {}), ("lp", {"eval_lp_args": {"knowledge_split": {"RS": [0.8, 0.1, 0.1]},
                                                               "knowledge_group_by": "relation", "mode": "full"},
                                     "metrics_lp": ["Recall", "MRR"], "topk_kg": [10],
                                     "valid_metric_kg": "MRR@10"})):
        cfg = dict(base)
        cfg.update(extra)
        try:
            config = Config(model="RotatE", dataset="test", config_file_list=[yaml], config_dict=cfg)
        except Exception as e:  # the lp variant depends on optional keys; the rec capture must not be lost
            print("config failed for", tag, repr(e))
            continue
        init_seed(config["seed"], config["reproducibility"])
        dataset = create_dataset(config)
        train_data, valid_data, test_data = data_preparation(config, dataset)
        model = get_model(config["model"])(config, train_data.dataset)
        trainer_cls = get_trainer(config["MODEL_TYPE"], config["model"])
        print(tag, type(dataset).__name__, type(train_data).__name__, trainer_cls.__name__,
              [type(v).__name__ for v in (valid_data if isinstance(valid_data, list) else [valid_data])])
        ds = train_data.dataset
        out[f"{tag}.stats"] = np.array([ds.user_num, ds.item_num, ds.entity_num, ds.relation_num,
                                        ds.field2token_id["relation_id"][ds.ui_relation]], np.int64)
        out[f"{tag}.loader_classes"] = np.array([type(dataset).__name__, type(train_data).__name__,
                                                 trainer_cls.__name__])
        for k, v in model.state_dict().items():
            out[f"{tag}.table.{k}"] = _np(v)

        if tag == "rec":
            # ---- A8: training batches as the reference's loader yields them (state RSKG) ----
            train_data.set_mode(KGDataLoaderState.RSKG)
            it = iter(train_data)
            for b in range(2):
                inter = next(it)
                keys = sorted(inter.interaction.keys())
                out[f"rec.train{b}.keys"] = np.array(keys)
                for k in keys:
                    out[f"rec.train{b}.{k}"] = _np(inter[k])
                model.zero_grad()
                loss = model.calculate_loss(inter)
                out[f"rec.train{b}.loss"] = _np(loss)
            out["rec.train.len"] = np.int64(len(train_data))
            out["rec.train.inter_count"] = np.int64(len(train_data.dataset))
            out["rec.train.kg_count"] = np.int64(len(train_data.kg_dataloader._dataset.kg_feat))
            # used-id sets of the samplers (what a negative may never be): flattened CSR
            for name, used in (("rec_used", train_data.general_dataloader._sampler.used_ids),
                               ("kg_used", train_data.kg_dataloader._sampler.used_ids)):
                ptr = np.zeros(len(used) + 1, np.int64)
                flat = []
                for i, s in enumerate(used):
                    srt = sorted(int(x) for x in s)
                    flat += srt
                    ptr[i + 1] = len(flat)
                out[f"rec.{name}.ptr"] = ptr
                out[f"rec.{name}.ids"] = np.array(flat, np.int64)
            # the raw training pairs / triples the batches are drawn from
            tf = train_data.dataset.inter_feat
            out["rec.train.users"] = _np(tf[ds.uid_field])
            out["rec.train.items"] = _np(tf[ds.iid_field])
            out["rec.kg.heads"] = _np(ds.kg_feat[ds.head_entity_field])
            out["rec.kg.rels"] = _np(ds.kg_feat[ds.relation_field])
            out["rec.kg.tails"] = _np(ds.kg_feat[ds.tail_entity_field])

        # ---- A18: evaluation batches ----
        loaders = valid_data if isinstance(valid_data, list) else [valid_data]
        model.eval()
        for li, loader in enumerate(loaders):
            kind = type(loader).__name__
            is_lp = "LP" in kind
            if tag == "rec" and is_lp:
                continue
            if tag == "lp" and not is_lp:
                continue
            n_cols = ds.entity_num if is_lp else ds.item_num
            fn = model.full_sort_predict_kg if is_lp else model.full_sort_predict
            nb = 0
            for batch in loader:
                interaction, history_index, positive_u, positive_i = batch
                with torch.no_grad():
                    scores = fn(interaction).view(-1, n_cols).clone()
                raw = scores.clone()
                scores[:, 0] = -np.inf
                if history_index is not None:
                    scores[history_index] = -np.inf
                tv, ti = torch.topk(scores, 10, dim=-1)
                p = f"{tag}.eval{nb}"
                out[f"{p}.loader"] = np.array(kind)
                for k in interaction.interaction:
                    out[f"{p}.inter.{k}"] = _np(interaction[k])
                if history_index is not None:
                    out[f"{p}.history_rows"] = _np(history_index[0])
                    out[f"{p}.history_cols"] = _np(history_index[1])
                out[f"{p}.positive_u"] = _np(positive_u)
                out[f"{p}.positive_i"] = _np(positive_i)
                out[f"{p}.scores"] = _np(raw)
                out[f"{p}.masked"] = _np(scores)
                out[f"{p}.topk_values"] = _np(tv)
                out[f"{p}.topk_idx"] = _np(ti)
                nb += 1
                if nb == 2:
                    break
            out[f"{tag}.eval.batches"] = np.int64(nb)
            out[f"{tag}.eval.step"] = np.int64(loader.step)
    path = os.path.join(out_dir, "dropin_RotatE_testdata.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, f"{os.path.getsize(path) / 1e3:.0f} kB", len(out), "arrays")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else HERE)
