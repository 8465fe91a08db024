"""Deterministic synthetic weights / frames / masks / prompts.

There are no checkpoints or datasets offline, so parity tests and `bench.py` run
on seeded synthetic data.  Everything is generated with numpy Philox streams
keyed by (seed, tensor name), so this container and the GPU box regenerate
bit-identical full-size tensors without shipping them.
"""
from __future__ import annotations

import zlib
from typing import Dict, Optional, Tuple

import numpy as np

from .config import VmaeConfig, state_dict_schema


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=[seed, zlib.crc32(name.encode())]))


def synthetic_tensor(name: str, shape: Tuple[int, ...], seed: int = 0) -> np.ndarray:
    """Xavier-uniform-like matrices (what the reference's `_init_weights` uses, vmae.py:100-107),
    but with small non-zero biases / LayerNorm affine jitter so every bias path is exercised."""
    g = _rng(seed, name)
    if name.endswith("norm.weight") or name.endswith("norm1.weight") or name.endswith("norm2.weight") or (
        "norm" in name and name.endswith(".weight")
    ):
        return (1.0 + 0.1 * (g.random(shape, dtype=np.float32) - 0.5)).astype(np.float32)
    if name.endswith("bias") or name.endswith("q_bias") or name.endswith("v_bias"):
        return (0.1 * (g.random(shape, dtype=np.float32) - 0.5)).astype(np.float32)
    if "token" in name:
        return (0.04 * (g.random(shape, dtype=np.float32) - 0.5)).astype(np.float32)
    if len(shape) >= 2:
        fan_out = shape[0]
        fan_in = int(np.prod(shape[1:]))
        a = float(np.sqrt(6.0 / (fan_in + fan_out)))
        return ((g.random(shape, dtype=np.float32) * 2.0 - 1.0) * a).astype(np.float32)
    return (0.1 * (g.random(shape, dtype=np.float32) - 0.5)).astype(np.float32)


def synthetic_state_dict(cfg: VmaeConfig, seed: int = 0, schema=None, sharp: bool = False) -> Dict[str, np.ndarray]:
    schema = schema if schema is not None else state_dict_schema(cfg)
    sd = {k: synthetic_tensor(k, shp, seed) for k, shp in schema.items()}
    return sharpen_state_dict(sd, seed) if sharp else sd


def sharpen_state_dict(sd: Dict[str, np.ndarray], seed: int = 0, qk_scale: float = 1.5, ln_range: Tuple[float, float] = (0.2, 3.0),
                       res_scale: float = 2.0) -> Dict[str, np.ndarray]:
    """Numerically hostile variant of a synthetic state dict (what trained, LayerNorm-heavy networks look like and the
    xavier-like generator does not): the q and k rows of every `attn.qkv.weight` x `qk_scale` (logits x qk_scale^2: sharp softmax),
    every LayerNorm weight ~ U(ln_range) (large dynamic range between channels), `proj` / `fc2` weights x `res_scale` (residual growth).

    The defaults are the sharpest setting at which the REFERENCE is still reproducible in fp32 (ViT-B/8: fp32 vs float64 evaluation of the
    same network 1.3e-5 max-abs; logits up to +-21, mean maximum attention weight 0.34 in the first block).  qk_scale 2 puts the reference's
    own fp32 rounding at 3.6e-4, qk_scale 4 makes the forward pass chaotic (fp32 vs float64: 6.8 max-abs on outputs of std 2): a 1e-3
    tolerance against an fp32 reference means nothing there (profiles/r3_hostile_scan.txt)."""
    out = {}
    for k, v in sd.items():
        v = v.copy()
        if k.endswith("attn.qkv.weight"):
            d = v.shape[0] // 3
            v[: 2 * d] *= qk_scale
        elif "norm" in k and k.endswith(".weight") and v.ndim == 1:
            v = (ln_range[0] + (ln_range[1] - ln_range[0]) * _rng(seed, "sharp." + k).random(v.shape, dtype=np.float32)).astype(np.float32)
        elif k.endswith("attn.proj.weight") or k.endswith("mlp.fc2.weight"):
            v *= res_scale
        out[k] = v
    return out


def synthetic_frames(batch: int, cfg: VmaeConfig, seed: int = 0) -> np.ndarray:
    """Wrapper-level input: float32 [B,T,C,H,W] in [0,1)."""
    g = np.random.Generator(np.random.PCG64(seed))
    return g.random((batch, cfg.num_frames, cfg.in_chans, cfg.img_size[0], cfg.img_size[1]), dtype=np.float32)


def synthetic_masks(batch: int, cfg: VmaeConfig, k_visible: int, seed: int = 0, clump: int = 1) -> np.ndarray:
    """bool [B,Nt]: frame 0 fully visible, frame 1 masked except `k_visible` patches per row
    (chosen as `k_visible/clump^2` clumps of clump x clump patches)."""
    g = np.random.Generator(np.random.PCG64(seed + 1))
    n = cfg.tokens_per_frame
    gh, gw = cfg.img_size[0] // cfg.patch, cfg.img_size[1] // cfg.patch
    mask = np.zeros((batch, cfg.num_frames, gh, gw), dtype=bool)
    mask[:, 1:] = True
    assert k_visible % (clump * clump) == 0
    n_clumps = k_visible // (clump * clump)
    ch, cw = gh // clump, gw // clump
    for b in range(batch):
        sel = g.permutation(ch * cw)[:n_clumps]
        for s in sel:
            i, j = divmod(int(s), cw)
            mask[b, -1, i * clump : (i + 1) * clump, j * clump : (j + 1) * clump] = False
    assert n * cfg.num_frames == mask[0].size
    return mask.reshape(batch, -1)


def synthetic_prompts(num: int, cfg: VmaeConfig, seed: int = 0, max_shift: int = 3) -> np.ndarray:
    """int32 [S,4] rows (active_h, active_w, dy, dx): one active patch and a non-zero shift in
    patch units, uniform in [-max_shift, max_shift]^2 \\ {0} (interface.py:370-377)."""
    g = np.random.Generator(np.random.PCG64(seed + 2))
    gh, gw = cfg.img_size[0] // cfg.patch, cfg.img_size[1] // cfg.patch
    out = np.zeros((num, 4), dtype=np.int32)
    for s in range(num):
        out[s, 0] = g.integers(gh)
        out[s, 1] = g.integers(gw)
        while True:
            dy, dx = g.integers(-max_shift, max_shift + 1, size=2)
            if dy != 0 or dx != 0:
                break
        out[s, 2], out[s, 3] = dy, dx
    return out
