"""Model API layer (drop-in for the reference's `models` package on the hot path)."""
from .models import create_model, create_optimizer  # noqa: F401
