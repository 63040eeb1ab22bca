"""The enum members the hot path consults: ModelType.KNOWLEDGE, InputType.{POINTWISE,PAIRWISE,USERWISE},
KGDataLoaderState (/root/reference/hopwise/utils/enum_type.py).

Enum members of different classes never compare or hash equal, and the reference keys its dispatch on them
(`type2class[model_type]` data/utils.py:55, `MODEL_TYPE == ModelType.KNOWLEDGE` configurator.py:306,
`MODEL_INPUT_TYPE == InputType.PAIRWISE` in the dataloaders, `get_trainer` utils/utils.py:117-125).  A model
class handed to `Config(model=<class>)` must therefore carry hopwise's OWN members: when hopwise is importable
they are re-exported from it, and the mirrors below are used only when it is absent (the GPU box, this repo's
own driver).  `HOPWISE_AMD_STANDALONE=1` forces the mirrors.
"""

import os
from enum import Enum

USING_HOPWISE = False

if os.environ.get("HOPWISE_AMD_STANDALONE", "0") != "1":
    try:
        from hopwise.utils.enum_type import InputType, KGDataLoaderState, ModelType  # noqa: F401

        USING_HOPWISE = True
    except ImportError:
        pass

if not USING_HOPWISE:

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
