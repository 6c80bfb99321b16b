import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def _load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return _load


def pool_input(S: int) -> np.ndarray:
    """Same recipe as tests/golden/make_golden.py::pool_input (kept in sync by
    test_oracle_golden.py::test_pool_recipe_matches_golden)."""
    rng = np.random.default_rng(1300 + S)
    A = rng.random((2, 1, S, S), dtype=np.float32)
    A[0, 0, S // 3: S // 3 + S // 8, S // 2: S // 2 + S // 8] += np.float32(5.0)
    return A


def clip_input(name: str) -> np.ndarray:
    """Same recipe as tests/golden/make_golden.py::clip_input."""
    hh, ww = {"sq500": (500, 500), "sq336": (336, 336), "land": (375, 500), "small": (64, 48)}[name]
    rng = np.random.default_rng(1800 + hh + ww)
    im = rng.integers(0, 256, (hh, ww, 3), dtype=np.uint8)
    if name == "sq500":
        yy, xx = np.mgrid[0:hh, 0:ww]
        im[..., 0] = (127 + 120 * np.sin(xx / 17.0) * np.cos(yy / 23.0)).astype(np.uint8)
    return im


def clip_digest(out: np.ndarray) -> dict:
    """Same digest as tests/golden/make_golden.py::clip_digest."""
    out = np.ascontiguousarray(out, dtype=np.float32)
    bits = out.view(np.uint32).astype(np.uint64)
    pos = (np.arange(out.size, dtype=np.uint64) % np.uint64(65521)) + np.uint64(1)
    return {"sub": out[:, ::7, ::5].copy(), "sum_bits": np.array(bits.sum(dtype=np.uint64)),
            "wsum_bits": np.array((bits.ravel() * pos).sum(dtype=np.uint64))}


def config1_inputs():
    """BASELINE configs[0] (SURVEY 8d "config 1"): same recipe as tests/golden/make_golden.py::config1_inputs."""
    img = np.random.default_rng(0).integers(0, 256, (336, 336, 3), dtype=np.uint8)
    att = np.random.default_rng(1).random((24, 24))
    att = (att / att.sum()).astype(np.float32)
    return img, att


MAIN_BATCHED_LOOP_WH = [(683, 1024), (500, 375), (333, 500), (1024, 768), (640, 427)]      # W x H as PIL reports them


def main_batched_loop_inputs():
    """Inputs of tests/golden/main_batched_loop.npz: same recipe as tests/golden/make_golden.py::main_batched_loop_inputs (five
    RGB images of different sizes and their 24 x 24 attention maps, the last one the constant 1 / 576 map)."""
    import torch
    imgs, atts = [], []
    for i, (w, h) in enumerate(MAIN_BATCHED_LOOP_WH):
        rng = np.random.default_rng(2600 + i)
        yy, xx = np.mgrid[0:h, 0:w]
        im = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        im[..., 1] = (127 + 120 * np.sin(xx / (11.0 + i)) * np.cos(yy / (17.0 + 2 * i))).astype(np.uint8)
        imgs.append(im)
        a = rng.random((24, 24)) ** 3
        a[5 + i:8 + i, 9:12] += 4.0
        atts.append((a / a.sum()).astype(np.float32))
    atts[-1] = (torch.ones(24, 24) / 576).numpy()
    return imgs, atts


def marginalnet_full_state(shapes: dict) -> dict:
    """Seeded weights of MarginalNet(1024, 4096, hidden=256) (BASELINE configs[4]), same recipe as
    tests/golden/make_golden.py: keys in sorted order, N(0,1)/sqrt(fan_in) weights, small biases.  11 MB of weights
    are regenerated from this recipe instead of being committed."""
    import torch
    g = torch.Generator().manual_seed(4096)
    sd = {}
    for k in sorted(shapes):
        shp = tuple(shapes[k])
        w = torch.randn(shp, generator=g)
        fan_in = int(np.prod(shp[1:])) if len(shp) > 1 else 0
        sd[k] = w * (2.0 / fan_in ** 0.5) if fan_in else w * 0.1       # gain 2: logits of order 1, peaked px / py
    return sd


def marginalnet_full_inputs(B: int = 4):
    """Seeded inputs at config-5 shapes: CLIP-L/14@336 token map [B,1024,24,24], 32 text tokens of width 4096 with
    ragged masks (sample 1: 5 valid tokens, sample 3: none -- the clamp_min(1) branch)."""
    import torch
    g = torch.Generator().manual_seed(2048)
    fmap = torch.randn(B, 1024, 24, 24, generator=g)
    ttok = torch.randn(B, 32, 4096, generator=g)
    tmask = torch.ones(B, 32, 1)
    tmask[1, 5:] = 0
    if B > 3:
        tmask[3] = 0
    return fmap, ttok, tmask
