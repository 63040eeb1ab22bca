from .abstract_recommender import AbstractRecommender, ExplainableRecommender, KnowledgeRecommender  # noqa: F401
from .knowledge_graph_embedding_recommender import RESCAL, ComplEx, DistMult, RotatE, TransE, TuckER  # noqa: F401
from .cafe import CAFE  # noqa: F401,E402
from .knowledge_aware_recommender import PGPR  # noqa: F401,E402
