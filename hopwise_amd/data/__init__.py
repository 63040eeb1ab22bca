from .interaction import Interaction  # noqa: F401
