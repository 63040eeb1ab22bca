from .abstract_recommender import AbstractRecommender, ExplainableRecommender, KnowledgeRecommender  # noqa: F401
from .knowledge_graph_embedding_recommender import (  # noqa: F401
    RESCAL, Analogy, ComplEx, ConvE, ConvKB, DistMult, HolE, RotatE, TorusE, TransD, TransE, TransH, TransR, TuckER,
)
from .cafe import CAFE  # noqa: F401,E402
from .knowledge_aware_recommender import PGPR  # noqa: F401,E402
from .tprec import TPRec  # noqa: F401,E402
