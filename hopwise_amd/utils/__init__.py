from .enum_type import InputType, KGDataLoaderState, ModelType  # noqa: F401
