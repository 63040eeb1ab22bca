"""The enum members the hot path consults (mirror of /root/reference/hopwise/utils/enum_type.py:
ModelType.KNOWLEDGE, InputType.{POINTWISE,PAIRWISE,USERWISE}, KGDataLoaderState)."""

from enum import Enum


class ModelType(Enum):
    GENERAL = 1
    SEQUENTIAL = 2
    CONTEXT = 3
    KNOWLEDGE = 4
    TRADITIONAL = 5
    DECISIONTREE = 6
    PATH_LANGUAGE_MODELING = 7


class InputType(Enum):
    POINTWISE = 1
    PAIRWISE = 2
    LISTWISE = 3
    PATHWISE = 4
    USERWISE = 5


class KGDataLoaderState(Enum):
    RSKG = 1
    RS = 2
    KG = 3
