"""Shared test helpers: fake config/dataset objects in the shape the reference passes to a model."""

import numpy as np

FIELDS = dict(
    USER_ID_FIELD="user_id",
    ITEM_ID_FIELD="item_id",
    NEG_PREFIX="neg_",
    ENTITY_ID_FIELD="entity_id",
    RELATION_ID_FIELD="relation_id",
    HEAD_ENTITY_ID_FIELD="head_id",
    TAIL_ENTITY_ID_FIELD="tail_id",
)


class FakeDataset:
    ui_relation = "[UI-Relation]"

    def __init__(self, n_users, n_items, n_entities, n_relations):
        self._num = dict(user_id=n_users, item_id=n_items, entity_id=n_entities, relation_id=n_relations)
        self.field2token_id = {"relation_id": {self.ui_relation: n_relations - 1}}

    def num(self, field):
        return self._num[field]


def make_config(device, embedding_size, margin=1.0, **extra):
    return dict(FIELDS, embedding_size=embedding_size, margin=margin, device=device, **extra)


def random_tables(model, n_users, n_entities, n_relations, dim, seed=0):
    """float32 tables keyed by the reference state_dict names (without '.weight')."""
    from oracle import kge

    rng = np.random.default_rng(seed)
    out = {}
    for name in kge.table_names(model):
        rows = n_users if name.startswith("user") else (
            n_relations if name.startswith("relation") or name in ("norm_vec", "proj_mat_e") else n_entities)
        if model == "RotatE" and name == "relation_embedding":
            out[name] = rng.uniform(0, 2 * np.pi, (rows, dim)).astype(np.float32)
        elif (model == "RESCAL" and name == "relation_embedding") or name == "proj_mat_e":
            out[name] = (rng.standard_normal((rows, dim * dim)) / np.sqrt(dim)).astype(np.float32)
        elif model == "TorusE":  # rows well outside (-1, 1) so that frac() matters
            out[name] = (rng.standard_normal((rows, dim)) * 1.5).astype(np.float32)
        else:
            out[name] = (rng.standard_normal((rows, dim)) * np.sqrt(2.0 / (rows + dim))).astype(np.float32)
    return out


def load_tables(model_obj, tables, device):
    import torch

    sd = {k + ".weight": torch.from_numpy(np.ascontiguousarray(v)).to(device) for k, v in tables.items()}
    model_obj.load_state_dict(sd)
    return model_obj


class FakeKGDataset(FakeDataset):
    """What PGPR reads from a KnowledgeBasedDataset: id counts, the five graph arrays, positives, preloaded tables."""

    uid_field, iid_field = "user_id", "item_id"

    def __init__(self, n_users, n_items, n_entities, n_relations, ui_relation_id, inter_users, inter_items, heads, rels,
                 tails, positives, user_emb, entity_emb, relation_emb):
        super().__init__(n_users, n_items, n_entities, n_relations)
        self.field2token_id = {"relation_id": {self.ui_relation: int(ui_relation_id)}}
        self.inter_feat = {"user_id": np.asarray(inter_users), "item_id": np.asarray(inter_items)}
        self.head_entities, self.relations, self.tail_entities = np.asarray(heads), np.asarray(rels), np.asarray(tails)
        self._positives = np.asarray(positives)
        self._w = {"user_embedding_id": user_emb, "entity_embedding_id": entity_emb, "relation_embedding_id": relation_emb}

    def history_item_matrix(self):
        return self._positives, None, None

    def get_preload_weight(self, name):
        return self._w[name]


def pgpr_config(device, max_acts=250, state_history=1, hops=(25, 5, 1), fix_bug=False, topk=(10,), dropout="identity",
                hidden=(512, 256)):
    return dict(FIELDS, device=device, topk=list(topk), state_history=state_history, max_acts=max_acts,
                hidden_sizes=list(hidden), beam_search_hop=list(hops), fix_scores_sorting_bug=fix_bug,
                policy_dropout=dropout, gamma=0.99, action_dropout=0.0, max_path_len=3, weight_factor=1e-3,
                path_constraint=[])
