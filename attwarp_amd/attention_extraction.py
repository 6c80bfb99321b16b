"""Drop-in for the hot-path half of ``Attention Guided Warping/attention_extraction/llava.py``:
``BatchMaskHookLogger`` (reference :338-448), ``MaskHookLogger`` (:37-153), ``revise_mask``
(:223-238), ``blend_mask``'s mask branch (:240-253).

The per-sample Python loops of tiny torch ops in the reference (:388-395) become one HIP launch per
generation step; the mask post-processing and the PIL LANCZOS up-sample run on the GPU and return
device tensors, so nothing has to cross PCIe before the warp.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from . import _lib, _tables
from ._lib import call, ptr, require_gpu, stream_ptr

NUM_IMAGE_TOKENS = 576  # 24 x 24 patches for LLaVA-1.5 (reference :350)


def _check_slices(starts, ntok: int, kv: int, who: str):
    """The kernels clamp an out-of-range slice start on the device (memory safety); a range that does not lie inside
    the attention row is a bookkeeping bug of the caller and must not turn into plausible-looking attention for other
    tokens.  (The reference would fail differently: a negative start wraps from the end, a truncated slice cannot be
    stacked, llava.py:390-395.)"""
    for b, st in enumerate(starts):
        if st < 0 or st + ntok > kv:
            raise ValueError(f"{who}: image-token range [{st}, {st + ntok}) of sample {b} is outside the attention row "
                             f"of {kv} keys")


def attn_reduce_step(attn_weights: torch.Tensor, starts: torch.Tensor, ntok: int,
                     out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One step of attention aggregation: [B,heads,q,kv] -> [B,ntok] (same dtype).
    ``starts``: int32 device tensor [B] of per-sample image-token offsets.  ``out``: optional caller-owned result."""
    dev = require_gpu(attn_weights, starts)
    B, heads, q, kv = attn_weights.shape
    if out is None:
        out = torch.empty(B, ntok, device=dev, dtype=attn_weights.dtype)
    elif tuple(out.shape) != (B, ntok) or out.dtype != attn_weights.dtype or not out.is_contiguous() or out.device != dev:
        raise ValueError(f"attn_reduce_step: out must be a contiguous {attn_weights.dtype} [B={B}, ntok={ntok}] tensor on {dev}")
    sb, sh, sq, skv = attn_weights.stride()
    with torch.cuda.device(dev):
        call("attwarp_attn_reduce_step", ptr(attn_weights), _lib.dtype_id(attn_weights), B, heads, q, kv, sb, sh, sq,
             skv, ptr(starts), int(ntok), ptr(out), stream_ptr(dev))
    return out


def attn_finalize(steps: torch.Tensor) -> torch.Tensor:
    """Mean over generation steps: [T,B,ntok] -> [B,ntok]."""
    dev = require_gpu(steps)
    s = steps.contiguous()
    T, B, ntok = s.shape
    out = torch.empty(B, ntok, device=dev, dtype=s.dtype)
    with torch.cuda.device(dev):
        call("attwarp_attn_finalize", ptr(s), _lib.dtype_id(s), T, B, ntok, ptr(out), stream_ptr(dev))
    return out


def attn_reduce_stack(rows: torch.Tensor, starts: torch.Tensor, ntok: int = NUM_IMAGE_TOKENS) -> torch.Tensor:
    """Fused aggregation of a captured stack of last-query rows [T,B,heads,kv] -> [B,ntok]."""
    dev = require_gpu(rows, starts)
    r = rows.contiguous()
    T, B, heads, kv = r.shape
    lib = _lib.load()
    ws = torch.empty(lib.attwarp_attn_reduce_stack_workspace_bytes(_lib.dtype_id(r), T, B, ntok), device=dev,
                     dtype=torch.uint8)
    out = torch.empty(B, ntok, device=dev, dtype=r.dtype)
    with torch.cuda.device(dev):
        call("attwarp_attn_reduce_stack", ptr(r), _lib.dtype_id(r), T, B, heads, kv, ptr(starts), int(ntok), ptr(out),
             ptr(ws), stream_ptr(dev))
    return out


def probe_last_query(query: torch.Tensor, key: torch.Tensor, starts: torch.Tensor, ntok: int = NUM_IMAGE_TOKENS,
                     kv_begin: Optional[torch.Tensor] = None, scaling: Optional[float] = None) -> torch.Tensor:
    """Hook-side capture: one generation step's [B,ntok] map computed from the post-RoPE
    query of the LAST token and the key cache, instead of from materialised [B,heads,q,kv] probabilities.

    query [B,heads,q,D] (or [B,heads,D]); key [B,kv_heads,kv,D]; starts int32 [B] on the device;
    kv_begin int32 [B] = number of left-padding keys per sample (None: none); scaling defaults to D**-0.5.
    Equals ``attn_reduce_step`` applied to HF's eager attention weights (same dtype transitions)."""
    dev = require_gpu(query, key, starts)
    q = query[:, :, -1] if query.dim() == 4 else query
    B, heads, D = q.shape
    if key.dim() != 4 or key.shape[0] != B or key.shape[3] != D:
        raise ValueError(f"key must be [B,kv_heads,kv,D] matching query {tuple(query.shape)}; got {tuple(key.shape)}")
    if key.dtype != q.dtype:
        raise ValueError(f"query/key dtype mismatch: {q.dtype} vs {key.dtype}")
    kv_heads, kv = key.shape[1], key.shape[2]
    per = 16 // q.element_size()

    def ok(t):   # 16-byte loads: unit stride on D, every other stride and the base a multiple of 16 bytes
        return t.stride(-1) == 1 and all(st % per == 0 for st in t.stride()[:-1]) and t.data_ptr() % 16 == 0

    if not ok(q):
        q = q.contiguous()
    k = key if ok(key) else key.contiguous()
    lib = _lib.load()
    dt = _lib.dtype_id(q)
    ws = torch.empty(lib.attwarp_attn_probe_workspace_bytes(dt, B, heads, int(ntok)), device=dev, dtype=torch.uint8)
    out = torch.empty(B, int(ntok), device=dev, dtype=q.dtype)
    kb = None
    if kv_begin is not None:
        require_gpu(kv_begin)
        kb = kv_begin.to(torch.int32).contiguous()
    st = starts.to(torch.int32).contiguous()
    with torch.cuda.device(dev):
        call("attwarp_attn_probe_last_query", ptr(q), ptr(k), dt, B, heads, kv_heads, D, kv, q.stride(0), q.stride(1),
             k.stride(0), k.stride(1), k.stride(2), ptr(kb) if kb is not None else None, ptr(st), int(ntok),
             float(D ** -0.5 if scaling is None else scaling), ptr(out), ptr(ws), stream_ptr(dev))
    return out


class BatchMaskHookLogger(object):
    """Same interface as the reference class (:338-448): per-sample image-token ranges, a forward
    hook on one decoder layer's attention, ``finalize_batch() -> List[Tensor[24,24]]``."""

    def __init__(self, model, device, layer_index: int = 20):
        self.device = device
        self.model = model
        self.layer_index = layer_index
        self.hook_handle = None
        self.num_image_tokens = NUM_IMAGE_TOKENS
        self.image_token_starts = None
        self.image_token_ends = None
        self.batch_size = 0
        self.step_attentions: List[torch.Tensor] = []
        self._original_forward = None
        self._starts_dev: Optional[torch.Tensor] = None

    def set_batch_image_token_ranges(self, starts: Sequence[int], ends: Sequence[int]):
        assert len(starts) == len(ends)
        self.image_token_starts = list(starts)
        self.image_token_ends = list(ends)
        self.batch_size = len(starts)
        self._starts_dev = None

    @torch.no_grad()
    def _attention_hook(self, module, input, output):
        if not isinstance(output, tuple) or len(output) < 2:
            return
        attn_weights = output[1]
        if attn_weights is None or not isinstance(attn_weights, torch.Tensor):
            return
        if len(attn_weights.shape) != 4:
            return
        self._process_attention(attn_weights)

    @torch.no_grad()
    def _process_attention(self, attn_weights: torch.Tensor):
        bsz, kv = attn_weights.shape[0], attn_weights.shape[-1]
        lens = {min(self.image_token_ends[b], kv) - self.image_token_starts[b] for b in range(bsz)}
        if len(lens) != 1:
            # the reference's torch.stack raises the same way when the slices differ in length
            raise RuntimeError(f"stack expects each tensor to be equal size, got image-token slice lengths {sorted(lens)}")
        ntok = lens.pop()
        _check_slices(self.image_token_starts[:bsz], ntok, kv, "BatchMaskHookLogger._process_attention")
        if self._starts_dev is None or self._starts_dev.device != attn_weights.device:
            self._starts_dev = torch.tensor(self.image_token_starts, dtype=torch.int32, device=attn_weights.device)
        self.step_attentions.append(attn_reduce_step(attn_weights, self._starts_dev[:bsz], ntok))

    @torch.no_grad()
    def finalize_batch(self) -> List[torch.Tensor]:
        if len(self.step_attentions) == 0:
            return [torch.ones(self.num_image_tokens, device=self.device) / self.num_image_tokens
                    for _ in range(self.batch_size)]
        avg = attn_finalize(torch.stack(self.step_attentions, dim=0))
        return [avg[i].view(24, 24) for i in range(self.batch_size)]

    def reinit(self):
        self.step_attentions = []
        self.image_token_starts = None
        self.image_token_ends = None
        self.batch_size = 0
        self._starts_dev = None

    def register_hook_and_patch(self):
        """Hook the target layer's attention module and force ``output_attentions=True`` for it only."""
        self.remove_hook_and_unpatch()
        layer_attn = self.model.model.layers[self.layer_index].self_attn
        self.hook_handle = layer_attn.register_forward_hook(self._attention_hook)
        original = layer_attn.forward
        self._original_forward = original

        def forced(*args, **kwargs):
            kwargs["output_attentions"] = True
            return original(*args, **kwargs)

        layer_attn.forward = forced

    # ---- capture without output_attentions ------------------------------------------
    @torch.no_grad()
    def _probe_attention(self, query: torch.Tensor, key: torch.Tensor, attention_mask, scaling=None):
        """Same bookkeeping as ``_process_attention``, fed with the target layer's post-RoPE query/key."""
        bsz, kv = key.shape[0], key.shape[2]
        lens = {min(self.image_token_ends[b], kv) - self.image_token_starts[b] for b in range(bsz)}
        if len(lens) != 1:
            raise RuntimeError(f"stack expects each tensor to be equal size, got image-token slice lengths {sorted(lens)}")
        ntok = lens.pop()
        _check_slices(self.image_token_starts[:bsz], ntok, kv, "BatchMaskHookLogger._probe_attention")
        if self._starts_dev is None or self._starts_dev.device != key.device:
            self._starts_dev = torch.tensor(self.image_token_starts, dtype=torch.int32, device=key.device)
        kv_begin = None
        if isinstance(attention_mask, torch.Tensor) and attention_mask.dim() == 4:
            row = attention_mask[:, 0, -1, :kv]                   # what the last query may attend to
            valid = row if row.dtype == torch.bool else row > torch.finfo(row.dtype).min / 2
            kv_begin = valid.to(torch.int32).argmax(dim=-1).to(torch.int32)   # first attended key = left padding
        self.step_attentions.append(probe_last_query(query, key, self._starts_dev[:bsz], ntok, kv_begin, scaling))

    def register_probe(self):
        """Alternative to ``register_hook_and_patch``: leave the target layer on its fast attention kernel and
        compute the one attention row the hook needs from (query, key) with ``probe_last_query``.  Uses
        transformers' ``AttentionInterface`` (>= 4.48): the target layer alone gets a private config whose
        attention implementation is a wrapper that probes and then delegates to the model's own."""
        import copy
        import sys
        try:
            from transformers.modeling_utils import ALL_ATTENTION_FUNCTIONS, AttentionInterface
        except ImportError as e:   # fail loudly: there is no silent fallback to the eager path
            raise RuntimeError("register_probe needs transformers' AttentionInterface (>= 4.48); "
                               "use register_hook_and_patch with this transformers version") from e
        self.remove_hook_and_unpatch()
        layer_attn = self.model.model.layers[self.layer_index].self_attn
        base = layer_attn.config._attn_implementation
        eager = getattr(sys.modules[type(layer_attn).__module__], "eager_attention_forward", None)
        base_fn = ALL_ATTENTION_FUNCTIONS.get_interface(base, eager)
        if base_fn is None:
            raise RuntimeError(f"register_probe: cannot resolve the attention implementation '{base}'")
        logger = self

        def probed(module, query, key, value, attention_mask, **kwargs):
            if logger.image_token_starts is not None:
                logger._probe_attention(query, key, attention_mask, kwargs.get("scaling"))
            return base_fn(module, query, key, value, attention_mask, **kwargs)

        name = f"attwarp_probe_{id(self):x}"
        AttentionInterface.register(name, probed)
        self._probe_saved = (layer_attn, layer_attn.config, name)
        private = copy.copy(layer_attn.config)
        private._attn_implementation = name
        layer_attn.config = private

    def register_probe_legacy(self):
        """``register_probe`` for transformers 4.36 - 4.47 (the reference pins 4.37.2, which predates
        ``AttentionInterface``): a forward hook (``with_kwargs``) on the target ``LlamaAttention`` that, after the layer
        ran on its own fast kernel, rebuilds the post-RoPE query of the LAST token the way that version's forward does
        (``q_proj`` -> heads -> ``rotary_emb(x, seq_len)`` tables gathered at ``position_ids`` (4.36 / 4.37; from 4.38 on
        ``rotary_emb(x, position_ids)``, told apart by the signature) -> ``q*cos +
        rotate_half(q)*sin``, modeling_llama.py of 4.37.2) and reads the layer's post-RoPE keys from
        ``past_key_value.key_cache[layer_idx]`` -- the one projection of one token instead of eager attention over the
        whole prompt.  Needs ``use_cache=True`` (what ``generate`` does)."""
        self.remove_hook_and_unpatch()
        layer_attn = self.model.model.layers[self.layer_index].self_attn
        logger = self

        @torch.no_grad()
        def hook(module, args, kwargs, output):
            if logger.image_token_starts is None:
                return
            hidden = kwargs.get("hidden_states", args[0] if args else None)
            cache = kwargs.get("past_key_value", None)
            pos = kwargs.get("position_ids", None)
            if hidden is None or cache is None:
                raise RuntimeError("register_probe_legacy: the hooked attention ran without hidden_states / past_key_value "
                                   "(use_cache=True is required)")
            key = cache.key_cache[getattr(module, "layer_idx", logger.layer_index)]          # [B, kv_heads, kv, D]
            B, kv = key.shape[0], key.shape[2]
            heads = getattr(module, "num_heads", None) or module.config.num_attention_heads
            D = key.shape[-1]
            q = module.q_proj(hidden[:, -1:, :]).view(B, 1, heads, D).transpose(1, 2)            # [B, heads, 1, D]
            if pos is None:
                pos = torch.full((B, 1), kv - 1, dtype=torch.long, device=hidden.device)
            import inspect
            if "seq_len" in inspect.signature(module.rotary_emb.forward).parameters:
                # 4.36 / 4.37: rotary_emb(x, seq_len) -> tables [seq, D], gathered at position_ids
                cos, sin = module.rotary_emb(q, seq_len=kv)
                cos, sin = cos[pos[:, -1:]].unsqueeze(1), sin[pos[:, -1:]].unsqueeze(1)       # [B, 1, 1, D]
            else:
                # 4.38 - 4.47: rotary_emb(x, position_ids) -> (cos, sin) [B, 1, D] for the given positions
                cos, sin = module.rotary_emb(q, pos[:, -1:])
                cos, sin = cos.unsqueeze(1), sin.unsqueeze(1)                                   # [B, 1, 1, D]
            half = D // 2
            rot = torch.cat((-q[..., half:], q[..., :half]), dim=-1)
            q = (q * cos) + (rot * sin)
            logger._probe_attention(q, key, kwargs.get("attention_mask", None), D ** -0.5)

        self.hook_handle = layer_attn.register_forward_hook(hook, with_kwargs=True)

    def remove_hook_and_unpatch(self):
        if self.hook_handle is not None:
            self.hook_handle.remove()
            self.hook_handle = None
        if self._original_forward is not None:
            self.model.model.layers[self.layer_index].self_attn.forward = self._original_forward
            self._original_forward = None
        saved = getattr(self, "_probe_saved", None)
        if saved is not None:
            layer_attn, config, name = saved
            layer_attn.config = config
            try:
                from transformers.modeling_utils import ALL_ATTENTION_FUNCTIONS
                ALL_ATTENTION_FUNCTIONS._global_mapping.pop(name, None)
            except Exception:
                pass
            self._probe_saved = None


def batch_image_token_ranges(unpadded_lens: Sequence[int], image_token_positions: Sequence[int],
                             num_image_tokens: int = NUM_IMAGE_TOKENS):
    """Range bookkeeping of the batched driver (reference ``functions.py:276-291``): every prompt holds ONE image
    placeholder that the multimodal expansion replaces by ``num_image_tokens`` embeddings, the batch is then
    LEFT-padded to the longest expanded prompt.  Returns ``(starts, ends, pad_offsets)`` -- what
    ``set_batch_image_token_ranges`` takes, plus the per-sample left padding (the ``kv_begin`` of
    ``probe_last_query``)."""
    if len(unpadded_lens) != len(image_token_positions):
        raise ValueError("unpadded_lens and image_token_positions differ in length")
    expanded = [int(ul) - 1 + num_image_tokens for ul in unpadded_lens]
    longest = max(expanded) if expanded else 0
    pads = [longest - e for e in expanded]
    starts = [pad + int(pos) for pad, pos in zip(pads, image_token_positions)]
    ends = [st + num_image_tokens for st in starts]
    return starts, ends, pads


def batch_hook_logger(model, device, layer_index: int = 20) -> BatchMaskHookLogger:
    """Reference :451-462."""
    prs = BatchMaskHookLogger(model, device, layer_index)
    model.config.output_attentions = False
    prs.register_hook_and_patch()
    model.batch_hooklogger = prs
    return prs


class MaskHookLogger(object):
    """Single-sample twin (reference :37-153): one image-token range for the whole batch,
    ``finalize()`` returns the flat [ntok] mean over steps and batch rows."""

    def __init__(self, model, device, layer_index: int = 20):
        self.device = device
        self.attns: List[torch.Tensor] = []
        self.model = model
        self.layer_index = layer_index
        self.hook_handle = None
        self.image_token_start = None
        self.image_token_end = None
        self.num_image_tokens = NUM_IMAGE_TOKENS

    def set_image_token_range(self, start, end):
        self.image_token_start = start
        self.image_token_end = end

    def _find_image_token_range(self, input_ids):
        """Reference :52-72 -- a fixed heuristic: the image tokens follow the BOS token."""
        return 1, 1 + self.num_image_tokens

    @torch.no_grad()
    def _attention_hook(self, module, input, output):
        """Forward hook of the target ``self_attn`` (reference :74-92): consumes ``output[1]`` when the layer
        returned 4-D attention weights, silently ignores anything else."""
        w = output[1] if isinstance(output, tuple) and len(output) >= 2 else None
        if isinstance(w, torch.Tensor) and w.dim() == 4:
            self._process_attention(w)

    def register_hook(self):
        """Reference :141-147."""
        self.remove_hook()
        self.hook_handle = self.model.model.layers[self.layer_index].self_attn.register_forward_hook(self._attention_hook)

    def remove_hook(self):
        """Reference :149-153."""
        if self.hook_handle is not None:
            self.hook_handle.remove()
            self.hook_handle = None

    @torch.no_grad()
    def _process_attention(self, attn_weights: torch.Tensor):
        kv = attn_weights.shape[-1]
        if self.image_token_start is None or self.image_token_end is None:
            st, ed = 1, min(1 + self.num_image_tokens, kv)
        else:
            st, ed = self.image_token_start, min(self.image_token_end, kv)
        _check_slices([st], ed - st, kv, "MaskHookLogger._process_attention")
        starts = torch.full((attn_weights.shape[0],), st, dtype=torch.int32, device=attn_weights.device)
        self.attns.append(attn_reduce_step(attn_weights, starts, ed - st))

    @torch.no_grad()
    def finalize(self) -> torch.Tensor:
        if len(self.attns) == 0:
            return torch.ones(self.num_image_tokens, device=self.device) / self.num_image_tokens
        rows = torch.cat(self.attns, dim=0)          # [steps*batch, ntok]
        return attn_finalize(rows.unsqueeze(1))[0]

    def reinit(self):
        self.attns = []
        self.image_token_start = None
        self.image_token_end = None


def hook_logger(model, device, layer_index: int = 20) -> MaskHookLogger:
    """Reference :156-187 (what ``main.py:38,307`` and ``new_method.py:45`` import): switches
    ``model.config.output_attentions`` on for the whole model, hooks the target layer and parks the logger and the
    previous config value on the model."""
    prs = MaskHookLogger(model, device, layer_index)
    previous = getattr(model.config, "output_attentions", False)
    model.config.output_attentions = True
    prs.register_hook()
    model.hooklogger = prs
    model._original_output_attentions = previous
    return prs


def revise_mask(patch_mask: torch.Tensor, kernel_size: int = 3, enhance_coe: float = 10) -> torch.Tensor:
    """Reference :223-238: min-max normalise -> z-score * coe -> sigmoid -> k x k mean filter with
    replicate padding.  Accepts [n,n] (returns [n,n]) or a batch [B,n,n]."""
    assert kernel_size % 2 == 1
    dev = require_gpu(patch_mask)
    m = patch_mask.detach().to(torch.float32).contiguous()
    single = m.dim() == 2
    if single:
        m = m.unsqueeze(0)
    B, n, n2 = m.shape
    if n != n2:
        raise ValueError(f"revise_mask expects square masks, got {tuple(m.shape)}")
    out = torch.empty_like(m)
    with torch.cuda.device(dev):
        call("attwarp_mask_postproc", ptr(m), B, n, int(kernel_size), float(enhance_coe), ptr(out), stream_ptr(dev))
    return out[0] if single else out


def upsample_mask_lanczos(mask: torch.Tensor, size_wh) -> torch.Tensor:
    """``toImg`` (x255, truncating uint8 cast) + ``PIL.Image.resize((W,H), LANCZOS)`` on the GPU
    (reference :192-196, :243, :253).  mask [B,h,w] float32 in [0,1] or uint8 -> uint8 [B,H,W]."""
    dev = require_gpu(mask)
    m = mask.detach().contiguous()
    if m.dim() == 2:
        m = m.unsqueeze(0)
    B, h, w = m.shape
    W, H = int(size_wh[0]), int(size_wh[1])
    is_f = m.dtype != torch.uint8
    if is_f:
        m = m.to(torch.float32)
    bx = kx = by = ky = None
    ksx = ksy = 0
    if W != w:
        bx, kx, ksx = _tables.lanczos_tables(w, W, dev)
    if H != h:
        by, ky, ksy = _tables.lanczos_tables(h, H, dev)
    tmp = torch.empty(B, h, W, device=dev, dtype=torch.uint8)
    out = torch.empty(B, H, W, device=dev, dtype=torch.uint8)
    with torch.cuda.device(dev):
        call("attwarp_mask_upsample_lanczos", ptr(m) if is_f else None, None if is_f else ptr(m), B, h, w, H, W,
             ptr(bx), ptr(kx), ksx, ptr(by), ptr(ky), ksy, ptr(tmp), ptr(out), stream_ptr(dev))
    return out


def blend_mask(image_path_or_pil_image, mask: torch.Tensor, enhance_coe, kernel_size, interpolate_method, grayscale):
    """Reference :240-270.  Returns ``(image, mask_pil)``: the mask branch (revise -> uint8 ->
    LANCZOS to the image size, mode "L") is computed on the GPU; the first element is the input
    image itself -- the Jet heat-map overlay of the reference is visualisation only and is out of
    scope (component 3)."""
    from PIL import Image
    if isinstance(image_path_or_pil_image, str):
        image = Image.open(image_path_or_pil_image)
    elif isinstance(image_path_or_pil_image, Image.Image):
        image = image_path_or_pil_image
    else:
        raise NotImplementedError
    lanczos = getattr(Image, "LANCZOS", 1)
    if interpolate_method not in (lanczos, "LANCZOS", getattr(getattr(Image, "Resampling", Image), "LANCZOS", 1)):
        raise NotImplementedError("only LANCZOS mask interpolation is implemented on the GPU path")
    if not mask.is_cuda:
        mask = mask.to(_lib.default_device())
    rev = revise_mask(mask.float().reshape(24, 24), kernel_size=kernel_size, enhance_coe=enhance_coe)
    up = upsample_mask_lanczos(rev.unsqueeze(0), image.size)[0]
    return image, Image.fromarray(up.cpu().numpy(), mode="L")
