from .build import build_optimizer

__all__ = ["build_optimizer"]
