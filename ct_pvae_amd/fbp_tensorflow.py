"""Alias so that `from <pkg>.fbp_tensorflow import iradon` keeps working (reference: ctvae/fbp_tensorflow.py)."""
from .fbp import iradon  # noqa: F401
