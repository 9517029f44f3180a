from .make_model import Signal, make_frame  # noqa: F401  (reference: modeling/__init__.py:1)
