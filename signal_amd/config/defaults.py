"""Configuration tree for the hot path: the keys of the reference's config/defaults.py that
make_frame / Signal.forward / make_loss / make_optimizer / the processor read,
with the same names and defaults, in a small yacs-compatible node (yacs itself is not a dependency):
attribute access, merge_from_file (YAML), merge_from_list (KEY VALUE pairs), freeze/defrost."""
from __future__ import annotations

import ast
import copy

import yaml


class CfgNode(dict):
    _IMMUTABLE = "__immutable__"

    def __init__(self, init=None):
        super().__init__()
        object.__setattr__(self, CfgNode._IMMUTABLE, False)
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e

    def __setattr__(self, name, value):
        if object.__getattribute__(self, CfgNode._IMMUTABLE):
            raise AttributeError(f"cannot set {name}: the config is frozen")
        self[name] = value

    def freeze(self, flag=True):
        object.__setattr__(self, CfgNode._IMMUTABLE, flag)
        for v in self.values():
            if isinstance(v, CfgNode):
                v.freeze(flag)

    def defrost(self):
        self.freeze(False)

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            out[k] = copy.deepcopy(v, memo)
        return out

    @staticmethod
    def _coerce(old, new, key):
        if isinstance(new, str) and not isinstance(old, str):
            try:
                new = ast.literal_eval(new)
            except (ValueError, SyntaxError):
                pass
        if isinstance(old, tuple) and isinstance(new, list):
            new = tuple(new)
        if isinstance(old, list) and isinstance(new, tuple):
            new = list(new)
        if isinstance(old, float) and isinstance(new, int) and not isinstance(new, bool):
            new = float(new)
        if old is not None and not isinstance(old, CfgNode) and type(old) is not type(new) and not (
                isinstance(old, (tuple, str)) and isinstance(new, (tuple, str))):
            raise TypeError(f"config key {key}: cannot replace {type(old).__name__} with {type(new).__name__}")
        return new

    def _merge(self, other: dict, path=""):
        for k, v in other.items():
            full = f"{path}.{k}" if path else k
            if k not in self:
                raise KeyError(f"non-existent config key: {full}")
            if isinstance(self[k], CfgNode):
                if not isinstance(v, dict):
                    raise TypeError(f"config key {full} is a section")
                self[k]._merge(v, full)
            else:
                self[k] = CfgNode._coerce(self[k], v, full)

    def merge_from_file(self, path: str):
        with open(path, "r") as f:
            self._merge(yaml.safe_load(f) or {})

    def merge_from_list(self, opts):
        opts = list(opts or [])
        if len(opts) % 2:
            raise ValueError("merge_from_list expects KEY VALUE pairs")
        for key, val in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                node = node[p]
            if parts[-1] not in node:
                raise KeyError(f"non-existent config key: {key}")
            node[parts[-1]] = CfgNode._coerce(node[parts[-1]], val, key)


def get_cfg_defaults() -> CfgNode:
    """Defaults of the reference's config/defaults.py for every key the hot path reads."""
    C = CfgNode
    return C(dict(
        MODEL=dict(DEVICE="cuda", DEVICE_ID="0", NAME="Signal", PRETRAIN_PATH_T="", NECK="bnneck",
                   IF_WITH_CENTER="no", ID_LOSS_TYPE="softmax", ID_LOSS_WEIGHT=1.0, TRIPLET_LOSS_WEIGHT=1.0,
                   Gram_Loss_weight=0.15, PAT_Loss_weight=0.1, MoE_Loss_weight=0.1, METRIC_LOSS_TYPE="triplet",
                   DIST_TRAIN=False, PROMPT=False, ADAPTER=False, FROZEN=False, IF_LABELSMOOTH="on", DIRECT=1,
                   DROP_PATH=0.1, DROP_OUT=0.0, ATT_DROP_RATE=0.0, TRANSFORMER_TYPE="vit_base_patch16_224",
                   STRIDE_SIZE=[16, 16], USE_A=False, USE_B=False, TOPK=64, FIXED_KEEP_RATIO=False, KEEP_RATIO=0.75,
                   stageName="CLS ", SIE_COE=3.0, SIE_CAMERA=True, SIE_VIEW=False, NO_MARGIN=True,
                   # signal_amd only (not a reference key): MFMA operand type, "bf16" or "fp16" (fp16 = the type of the
                   # reference's CUDA autocast; trained with device-side dynamic loss scaling)
                   OPERAND_DTYPE="bf16"),
        INPUT=dict(SIZE_TRAIN=[256, 128], SIZE_TEST=[256, 128], PROB=0.5, RE_PROB=0.5, PIXEL_MEAN=[0.5, 0.5, 0.5],
                   PIXEL_STD=[0.5, 0.5, 0.5], PADDING=10),
        DATASETS=dict(NAMES="RGBNT201", ROOT_DIR="./data"),
        DATALOADER=dict(NUM_WORKERS=6, SAMPLER="softmax_triplet", NUM_INSTANCE=8),
        SOLVER=dict(OPTIMIZER_NAME="SGD", MAX_EPOCHS=120, BASE_LR=0.009, LARGE_FC_LR=False, BIAS_LR_FACTOR=2,
                    MOMENTUM=0.9, MARGIN=0.3, CENTER_LR=0.5, CENTER_LOSS_WEIGHT=0.0005, WEIGHT_DECAY=0.0001,
                    WEIGHT_DECAY_BIAS=0.0001, GAMMA=0.1, STEPS=(40, 70), WARMUP_FACTOR=0.01, WARMUP_ITERS=10,
                    WARMUP_METHOD="linear", SEED=1234, CHECKPOINT_PERIOD=50, LOG_PERIOD=10, EVAL_PERIOD=5,
                    IMS_PER_BATCH=128),
        TEST=dict(EVAL=False, IMS_PER_BATCH=256, RE_RANKING="no", WEIGHT="", NECK_FEAT="before", FEAT_NORM="yes",
                  MISS="None", FEAT=0),
        OUTPUT_DIR="./test", ckpt_save_path="baseline", ckpt_test_path="test_RNT",
    ))


cfg = get_cfg_defaults()
