from .make_optimizer import FusedAdam, make_optimizer, param_hyper  # noqa: F401
