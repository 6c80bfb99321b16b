"""attwarp_amd -- MI355X-native (gfx950) implementation of AttWarp's attention-guided
image-warping hot path behind the reference's own Python call surface.

Sub-modules mirror the reference files on the path (dwipddalal/AttWarp):

=========================================  =====================================================
reference file                              this package
=========================================  =====================================================
model/marginalnet_full_dataset/             attwarp_amd.checkpoint_utils  (warp_from_cdf_torch,
  checkpoint_utils.py                         cdf_from_density, gt_marginals, resample_cdf, ...)
model/marginalnet_full_dataset/model.py     attwarp_amd.model            (safe_softmax, MarginalNet)
Attention Guided Warping/new_method.py      attwarp_amd.new_method       (warp_image_by_attention,
                                              save_warped_image, set_transform_function)
Attention Guided Warping/                   attwarp_amd.attention_extraction (BatchMaskHookLogger,
  attention_extraction/llava.py               revise_mask, blend_mask)
=========================================  =====================================================

plus ``attwarp_amd.pipeline`` (batched device-resident launcher) and ``attwarp_amd.dist``
(one process per GPU, image shards, one RCCL broadcast of MarginalNet weights).

Compute happens in hand-written HIP kernels (``attwarp_amd/csrc``) reached through a C ABI
(``include/attwarp.h``) with ctypes.  There is no CPU fallback.
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401  (does not load the shared library until first use)
