from .make_optimizer import FusedAdam, make_optimizer, param_hyper  # noqa: F401
from .scheduler_factory import NoisyCosineLR, WarmupMultiStepLR, create_scheduler  # noqa: F401
