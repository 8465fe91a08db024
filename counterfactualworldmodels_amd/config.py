"""Model shapes and state-dict schema of the VMAE predictors this package accelerates.

Mirrors the constructor arguments / factory functions of the reference
(`cwm/models/VideoMAE/vmae.py:257-384`, factories `:563-619`) so that published
checkpoints load unchanged (key names and shapes: SURVEY.md Appendix B).
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Tuple

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
LN_EPS = 1e-6


@dataclass(frozen=True)
class VmaeConfig:
    name: str = "base_8x8patch_2frames_1tube"
    img_size: Tuple[int, int] = (224, 224)
    patch: int = 8
    num_frames: int = 2
    in_chans: int = 3
    enc_dim: int = 768
    enc_depth: int = 12
    enc_heads: int = 12
    dec_dim: int = 384
    dec_depth: int = 4
    dec_heads: int = 6
    mlp_ratio: int = 4

    @property
    def tokens_per_frame(self) -> int:
        return (self.img_size[0] // self.patch) * (self.img_size[1] // self.patch)

    @property
    def num_tokens(self) -> int:
        return self.tokens_per_frame * self.num_frames

    @property
    def patch_dim(self) -> int:
        return self.in_chans * self.patch * self.patch

    @property
    def out_dim(self) -> int:
        return self.in_chans * self.patch * self.patch

    def with_image_size(self, hw) -> "VmaeConfig":
        from dataclasses import replace

        return replace(self, img_size=(int(hw[0]), int(hw[1])))


CONFIGS: Dict[str, VmaeConfig] = {
    "base_8x8patch_2frames_1tube": VmaeConfig(),  # vmae.py:605
    "base_16x16patch_2frames_1tube": VmaeConfig(name="base_16x16patch_2frames_1tube", patch=16),  # vmae.py:597
    "large_4x4patch_2frames_1tube": VmaeConfig(  # vmae.py:613
        name="large_4x4patch_2frames_1tube",
        patch=4,
        enc_dim=1024,
        enc_depth=24,
        enc_heads=16,
        dec_dim=512,
        dec_depth=12,
        dec_heads=8,
    ),
}


def _block_schema(pre: str, d: int, hidden: int, out: "OrderedDict[str, tuple]") -> None:
    out[pre + "norm1.weight"] = (d,)
    out[pre + "norm1.bias"] = (d,)
    out[pre + "attn.q_bias"] = (d,)
    out[pre + "attn.v_bias"] = (d,)
    out[pre + "attn.qkv.weight"] = (3 * d, d)
    out[pre + "attn.proj.weight"] = (d, d)
    out[pre + "attn.proj.bias"] = (d,)
    out[pre + "norm2.weight"] = (d,)
    out[pre + "norm2.bias"] = (d,)
    out[pre + "mlp.fc1.weight"] = (hidden, d)
    out[pre + "mlp.fc1.bias"] = (hidden,)
    out[pre + "mlp.fc2.weight"] = (d, hidden)
    out[pre + "mlp.fc2.bias"] = (d,)


def state_dict_schema(cfg: VmaeConfig) -> "OrderedDict[str, tuple]":
    """Parameter names -> shapes, in the reference's registration order."""
    s: "OrderedDict[str, tuple]" = OrderedDict()
    s["mask_token"] = (1, 1, cfg.dec_dim)
    s["encoder.patch_embed.proj.weight"] = (cfg.enc_dim, cfg.in_chans, 1, cfg.patch, cfg.patch)
    s["encoder.patch_embed.proj.bias"] = (cfg.enc_dim,)
    for i in range(cfg.enc_depth):
        _block_schema(f"encoder.blocks.{i}.", cfg.enc_dim, cfg.mlp_ratio * cfg.enc_dim, s)
    s["encoder.norm.weight"] = (cfg.enc_dim,)
    s["encoder.norm.bias"] = (cfg.enc_dim,)
    for i in range(cfg.dec_depth):
        _block_schema(f"decoder.blocks.{i}.", cfg.dec_dim, cfg.mlp_ratio * cfg.dec_dim, s)
    s["decoder.norm.weight"] = (cfg.dec_dim,)
    s["decoder.norm.bias"] = (cfg.dec_dim,)
    s["decoder.head.weight"] = (cfg.out_dim, cfg.dec_dim)
    s["decoder.head.bias"] = (cfg.out_dim,)
    s["encoder_to_decoder.weight"] = (cfg.dec_dim, cfg.enc_dim)
    return s


def num_parameters(cfg: VmaeConfig) -> int:
    n = 0
    for shp in state_dict_schema(cfg).values():
        k = 1
        for v in shp:
            k *= v
        n += k
    return n


def algorithmic_flops(cfg: VmaeConfig, n_vis: int) -> float:
    """GEMM + attention FLOPs (2*MAC) per frame pair, SURVEY.md §8(d)."""
    Nt, De, Dd = cfg.num_tokens, cfg.enc_dim, cfg.dec_dim
    Nm = Nt - n_vis
    f = 2.0 * Nt * cfg.patch_dim * De
    f += cfg.enc_depth * (24.0 * n_vis * De * De + 4.0 * n_vis * n_vis * De)
    f += 2.0 * n_vis * De * Dd
    f += cfg.dec_depth * (24.0 * Nt * Dd * Dd + 4.0 * Nt * Nt * Dd)
    f += 2.0 * Nm * Dd * cfg.out_dim
    return f
