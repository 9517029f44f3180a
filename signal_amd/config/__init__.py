from .defaults import CfgNode, cfg, get_cfg_defaults  # noqa: F401
