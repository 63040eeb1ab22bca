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
        rows = n_users if name.startswith("user") else (n_relations if name.startswith("relation") else n_entities)
        if model == "RotatE" and name == "relation_embedding":
            out[name] = rng.uniform(0, 2 * np.pi, (rows, dim)).astype(np.float32)
        else:
            out[name] = (rng.standard_normal((rows, dim)) * np.sqrt(2.0 / (rows + dim))).astype(np.float32)
    return out


def load_tables(model_obj, tables, device):
    import torch

    sd = {k + ".weight": torch.from_numpy(np.ascontiguousarray(v)).to(device) for k, v in tables.items()}
    model_obj.load_state_dict(sd)
    return model_obj
