"""AlignmentM (modeling/AddModule/useB.py:169-190) on the HIP path: GAM and LAM losses from the token tensor."""
import torch

from .hip_engine import GamFn, LamFn


def gam_loss(model, tokens, B):
    hip = model.hip
    hip.grad_mode = torch.is_grad_enabled()
    return GamFn.apply(hip, B, tokens, hip.flat.byname["AlignM.contra_temp"])


def lam_loss(model, tokens, B):
    hip = model.hip
    hip.grad_mode = torch.is_grad_enabled()
    return LamFn.apply(hip, B, tokens, *[hip.flat.byname[n] for n in hip.das_param_names])
