"""Drop-in for ``model/marginalnet_full_dataset/model.py``.

``safe_softmax`` (reference :8-14) is the "row/col marginal softmax" of the hot path and runs as a
HIP kernel.  ``MarginalNet`` (:17-95) is the producer of the path's input in the end-to-end
configuration ("next" row 1): its dense convolutions and linear layers are MFMA-class
library work and run on stock PyTorch-ROCm ops, the memory-bound tail between them (masked text-token
mean, FiLM + the two axis means) on HIP kernels that read each tensor once.  Parameter names are the
reference's, so checkpoints saved by the reference trainer (``{"model": state_dict}``,
MN/trainer.py:660-683) load unchanged.
Its weights are the payload of the one RCCL broadcast of the multi-GPU path (``attwarp_amd.dist``).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from ._lib import call, ptr, require_gpu, stream_ptr


def safe_softmax(logits: torch.Tensor, dim: int = 1, eps: float = 1e-6) -> torch.Tensor:
    """Softmax with NaN/Inf guards and re-normalisation (reference model.py:8-14)."""
    dev = require_gpu(logits)
    if logits.dim() < 1:
        raise ValueError("safe_softmax expects at least one dimension")
    if _lib.needs_grad(logits):           # training: the same formula on differentiable stock ops (same GPU)
        z = torch.nan_to_num(logits, nan=0.0, posinf=0.0, neginf=0.0)
        p = torch.nan_to_num(F.softmax(z - z.amax(dim=dim, keepdim=True), dim=dim), nan=0.0, posinf=0.0, neginf=0.0)
        return p / p.sum(dim=dim, keepdim=True).clamp_min(eps)
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


def masked_token_mean(txt_tok: torch.Tensor, txt_mask: torch.Tensor) -> torch.Tensor:
    """``(txt_tok * txt_mask).sum(1) / txt_mask.sum(1).clamp_min(1)`` (reference :77-78) in one pass.
    txt_tok [B,Lt,D] float32/float16/bfloat16, txt_mask [B,Lt,1] or [B,Lt] -> [B,D] float32."""
    dev = require_gpu(txt_tok, txt_mask)
    if _lib.needs_grad(txt_tok, txt_mask):
        m = txt_mask.float().reshape(txt_tok.shape[0], txt_tok.shape[1], 1)
        return (txt_tok.float() * m).sum(dim=1) / m.sum(dim=1).clamp_min(1.0)
    tok = txt_tok.detach().contiguous()
    B, Lt, D = tok.shape
    mask = txt_mask.detach().float().reshape(B, Lt).contiguous()
    out = torch.empty(B, D, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        call("attwarp_masked_token_mean", ptr(tok), _lib.dtype_id(tok), ptr(mask), B, Lt, D, ptr(out), stream_ptr(dev))
    return out


def film_axis_means(v: torch.Tensor, gamma_beta: torch.Tensor):
    """``v = gamma*v + beta; vx = v.mean(2); vy = v.mean(3)`` (reference :80-88) in one pass over ``v``.
    v [B,Ch,H,W] float32, gamma_beta [B,2*Ch] (``film`` output) -> (vx [B,Ch,W], vy [B,Ch,H])."""
    dev = require_gpu(v, gamma_beta)
    if _lib.needs_grad(v, gamma_beta):
        gamma, beta = gamma_beta.float().chunk(2, dim=1)
        w = gamma[:, :, None, None] * v.float() + beta[:, :, None, None]
        return w.mean(dim=2), w.mean(dim=3)
    x = v.detach().float().contiguous()
    gb = gamma_beta.detach().float().contiguous()
    B, Ch, H, W = x.shape
    if tuple(gb.shape) != (B, 2 * Ch):
        raise ValueError(f"gamma_beta must be [B, 2*Ch] = {(B, 2 * Ch)}; got {tuple(gb.shape)}")
    vx = torch.empty(B, Ch, W, device=dev, dtype=torch.float32)
    vy = torch.empty(B, Ch, H, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        call("attwarp_film_axis_means", ptr(x), ptr(gb), B, Ch, H, W, ptr(vx), ptr(vy), stream_ptr(dev))
    return vx, vy


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

    def forward_logits_fused(self, fmap_v, H: int, W: int, txt_tok, txt_mask):
        """GPU path of :meth:`forward`: the library GEMMs (convolutions, linear layers) on stock PyTorch-ROCm ops,
        the memory-bound tail around them on two HIP kernels that read each tensor once
        (``masked_token_mean``: reference :77-78; ``film_axis_means``: reference :80-88)."""
        v = self.proj_v(fmap_v.float())
        if tuple(v.shape[-2:]) != (H, W):       # same-size bilinear (align_corners=False) is the identity
            v = F.interpolate(v, size=(H, W), mode="bilinear", align_corners=False)
        t = masked_token_mean(txt_tok, txt_mask)
        vx, vy = film_axis_means(v, self.film(self.txt_pool(t)))
        return self.head_x(vx).squeeze(1), self.head_y(vy).squeeze(1)

    def forward(self, fmap_v, H: int, W: int, txt_tok, txt_mask):
        """Inference (no grad needed): library GEMMs + the fused HIP tail.  When autograd has to see through the call
        (training, MN/trainer.py:210-260: ``loss.backward()`` must reach the parameters) the same network runs on
        differentiable stock ops (``forward_logits``) -- the HIP kernels are forward-only."""
        if _lib.needs_grad(fmap_v, txt_tok, txt_mask, *self.parameters()):
            require_gpu(fmap_v, txt_tok, txt_mask)
            lx, ly = self.forward_logits(fmap_v, H, W, txt_tok, txt_mask)
            return safe_softmax(lx, dim=1, eps=self.eps), safe_softmax(ly, dim=1, eps=self.eps)
        lx, ly = self.forward_logits_fused(fmap_v, H, W, txt_tok, txt_mask)
        return safe_softmax(lx, dim=1, eps=self.eps), safe_softmax(ly, dim=1, eps=self.eps)


def load_reference_checkpoint(net: MarginalNet, path_or_dict) -> MarginalNet:
    """Load a checkpoint written by the reference trainer (``torch.save({"epoch","model","opt","cfg"})``,
    MN/trainer.py:660-683) or a bare state_dict."""
    sd = torch.load(path_or_dict, map_location="cpu") if isinstance(path_or_dict, (str, bytes)) else path_or_dict
    if isinstance(sd, dict) and "model" in sd and not any(k.startswith("proj_v") for k in sd):
        sd = sd["model"]
    net.load_state_dict(sd)
    return net
