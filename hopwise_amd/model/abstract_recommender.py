"""Model base classes: the drop-in protocol of /root/reference/hopwise/model/abstract_recommender.py
(AbstractRecommender :36-109, KnowledgeRecommender :197-223, ExplainableRecommender :226-260).

A model is constructed as ``Model(config, dataset)``; ``config`` is any mapping with the reference's
YAML keys, ``dataset`` any object offering ``num(field)``, ``field2token_id`` and ``ui_relation``.

When hopwise itself is importable the engine's models SUBCLASS hopwise's own bases (so ``model.type`` is
hopwise's ``ModelType.KNOWLEDGE`` member and the reference's Config / create_dataset / get_trainer dispatch
on it; see utils/enum_type.py).  The mirrors below serve the standalone case (GPU box, this repo's driver).
"""

from __future__ import annotations

import numpy as np
from torch import nn

from ..utils import ModelType
from ..utils.enum_type import USING_HOPWISE

_bases = None
if USING_HOPWISE:
    try:
        from hopwise.model import abstract_recommender as _bases
    except ImportError:  # enums importable but the model package is not: keep the mirrors
        _bases = None

if _bases is not None:
    AbstractRecommender = _bases.AbstractRecommender
    KnowledgeRecommender = _bases.KnowledgeRecommender
    ExplainableRecommender = _bases.ExplainableRecommender
else:

    class AbstractRecommender(nn.Module):
        def calculate_loss(self, interaction):
            raise NotImplementedError

        def predict(self, interaction):
            raise NotImplementedError

        def full_sort_predict(self, interaction):
            raise NotImplementedError

        def full_sort_predict_kg(self, interaction):
            raise NotImplementedError

        def other_parameter(self):
            if hasattr(self, "other_parameter_name"):
                return {key: getattr(self, key) for key in self.other_parameter_name}
            return dict()

        def load_other_parameter(self, para):
            if para is None:
                return
            for key, value in para.items():
                setattr(self, key, value)

        def __str__(self):
            params = sum(int(np.prod(p.size())) for p in self.parameters() if p.requires_grad)
            return super().__str__() + f"\nTrainable parameters: {params}"

    class KnowledgeRecommender(AbstractRecommender):
        type = ModelType.KNOWLEDGE

        def __init__(self, config, dataset):
            super().__init__()
            self.USER_ID = config["USER_ID_FIELD"]
            self.ITEM_ID = config["ITEM_ID_FIELD"]
            self.NEG_ITEM_ID = config["NEG_PREFIX"] + self.ITEM_ID
            self.ENTITY_ID = config["ENTITY_ID_FIELD"]
            self.RELATION_ID = config["RELATION_ID_FIELD"]
            self.HEAD_ENTITY_ID = config["HEAD_ENTITY_ID_FIELD"]
            self.TAIL_ENTITY_ID = config["TAIL_ENTITY_ID_FIELD"]
            self.NEG_TAIL_ENTITY_ID = config["NEG_PREFIX"] + self.TAIL_ENTITY_ID
            self.n_users = dataset.num(self.USER_ID)
            self.n_items = dataset.num(self.ITEM_ID)
            self.n_entities = dataset.num(self.ENTITY_ID)
            self.n_relations = dataset.num(self.RELATION_ID)
            self.device = config["device"]

    class ExplainableRecommender:
        def explain(self, interaction):
            raise NotImplementedError("explain is not implemented")

        def decode_path(self, path):
            raise NotImplementedError("decode_path is not implemented")
