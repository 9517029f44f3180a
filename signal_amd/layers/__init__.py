from .make_loss import make_loss  # noqa: F401
