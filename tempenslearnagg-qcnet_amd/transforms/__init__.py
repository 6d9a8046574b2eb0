from .target_builder import TargetBuilder  # noqa: F401
