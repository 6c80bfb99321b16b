"""Drop-in for ``model/marginalnet_full_dataset/model.py``.

``safe_softmax`` (reference :8-14) is the "row/col marginal softmax" of the hot path and runs as a
HIP kernel.  ``MarginalNet`` (:17-95) is the producer of the path's input in the end-to-end
configuration; its dense convolutions are MFMA-class library work, so it runs on stock
PyTorch-ROCm ops ("next" row 1) with the reference's parameter names so checkpoints
saved by the reference trainer (``{"model": state_dict}``, MN/trainer.py:660-683) load unchanged.
Its weights are the payload of the one RCCL broadcast of the multi-GPU path (``attwarp_amd.dist``).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from ._lib import call, ptr, require_gpu, stream_ptr


def safe_softmax(logits: torch.Tensor, dim: int = 1, eps: float = 1e-6) -> torch.Tensor:
    """Softmax with NaN/Inf guards and re-normalisation (reference model.py:8-14)."""
    dev = require_gpu(logits)
    if logits.dim() < 1:
        raise ValueError("safe_softmax expects at least one dimension")
    x = logits.detach().to(torch.float32)
    d = dim % x.dim()
    if d != x.dim() - 1:
        x = x.movedim(d, -1)
    shape = x.shape
    x2 = x.contiguous().reshape(-1, shape[-1])
    out = torch.empty_like(x2)
    with torch.cuda.device(dev):
        call("attwarp_safe_softmax", ptr(x2), x2.shape[0], x2.shape[1], float(eps), ptr(out), stream_ptr(dev))
    out = out.reshape(shape)
    if d != logits.dim() - 1:
        out = out.movedim(-1, d)
    return out.to(logits.dtype) if logits.dtype.is_floating_point else out


class MarginalNet(nn.Module):
    """Predicts px (B,W) and py (B,H) from a visual token map and text tokens.

    Same architecture and state_dict keys as the reference (model.py:25-53): ``proj_v`` 1x1 conv ->
    SiLU -> 3x3 conv -> SiLU; ``txt_pool`` two Linear+SiLU; ``film`` Linear -> (gamma, beta);
    ``head_x`` / ``head_y`` Conv1d(k=5) -> SiLU -> Conv1d(1).
    """

    def __init__(self, d_vis_in: int, d_txt_in: int, hidden: int = 256, eps: float = 1e-6):
        super().__init__()
        self.eps = float(eps)
        self.proj_v = nn.Sequential(nn.Conv2d(d_vis_in, hidden, 1), nn.SiLU(),
                                    nn.Conv2d(hidden, hidden, 3, padding=1), nn.SiLU())
        self.txt_pool = nn.Sequential(nn.Linear(d_txt_in, hidden), nn.SiLU(), nn.Linear(hidden, hidden), nn.SiLU())
        self.film = nn.Linear(hidden, 2 * hidden)

        def head():
            return nn.Sequential(nn.Conv1d(hidden, hidden, 5, padding=2), nn.SiLU(), nn.Conv1d(hidden, 1, 1))

        self.head_x = head()
        self.head_y = head()

    def forward_logits(self, fmap_v, H: int, W: int, txt_tok, txt_mask):
        """Everything before the two softmaxes (reference :69-91); any device."""
        v = self.proj_v(fmap_v.float())
        v = F.interpolate(v, size=(H, W), mode="bilinear", align_corners=False)
        m = txt_mask.float()
        t = (txt_tok.float() * m).sum(dim=1) / m.sum(dim=1).clamp_min(1.0)
        gamma, beta = self.film(self.txt_pool(t)).chunk(2, dim=1)
        v = gamma[:, :, None, None] * v + beta[:, :, None, None]
        logit_x = self.head_x(v.mean(dim=2)).squeeze(1)
        logit_y = self.head_y(v.mean(dim=3)).squeeze(1)
        return logit_x, logit_y

    def forward(self, fmap_v, H: int, W: int, txt_tok, txt_mask):
        lx, ly = self.forward_logits(fmap_v, H, W, txt_tok, txt_mask)
        return safe_softmax(lx, dim=1, eps=self.eps), safe_softmax(ly, dim=1, eps=self.eps)


def load_reference_checkpoint(net: MarginalNet, path_or_dict) -> MarginalNet:
    """Load a checkpoint written by the reference trainer (``torch.save({"epoch","model","opt","cfg"})``,
    MN/trainer.py:660-683) or a bare state_dict."""
    sd = torch.load(path_or_dict, map_location="cpu") if isinstance(path_or_dict, (str, bytes)) else path_or_dict
    if isinstance(sd, dict) and "model" in sd and not any(k.startswith("proj_v") for k in sd):
        sd = sd["model"]
    net.load_state_dict(sd)
    return net
