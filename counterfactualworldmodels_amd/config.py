"""Model shapes and state-dict schema of the VMAE predictors this package accelerates.

Mirrors the constructor arguments / factory functions of the reference
(`cwm/models/VideoMAE/vmae.py:257-384`, factories `:563-619`) so that published
checkpoints load unchanged (key names and shapes: SURVEY.md Appendix B).
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Tuple  # noqa: F401

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


# ---------------------------------------------------------------------------------------------
# IMU-conditioned conjoined padded predictor (BASELINE configs[4])
# `imu400_base_4x4patch_2frames_1tube` (cwm/models/VideoMAE/conjoined_vmae.py:1230-1243)
# ---------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class ConjConfig:
    name: str = "imu400_base_4x4patch_2frames_1tube"
    main: VmaeConfig = VmaeConfig(name="imu400_main_4x4", patch=4)
    main_max_pad: int = 64
    ctx_in_chans: int = 6
    ctx_seq_len: int = 400
    ctx_tubelet: int = 16
    ctx_enc_dim: int = 384
    ctx_dec_dim: int = 192
    ctx_enc_heads: int = 12
    ctx_dec_heads: int = 6
    ctx_max_pad: int = 25
    enc_cross: Tuple[int, ...] = (0, 3, 6, 9)
    dec_cross: Tuple[int, ...] = (0, 1, 2, 3)
    cross_heads: int = 4
    cross_mlp_ratio: int = 2

    @property
    def ctx_tokens(self) -> int:
        return self.ctx_seq_len // self.ctx_tubelet

    @property
    def ctx_out_dim(self) -> int:
        return self.ctx_in_chans * self.ctx_tubelet


CONJ_CONFIGS: Dict[str, ConjConfig] = {"imu400_base_4x4patch_2frames_1tube": ConjConfig()}


def _cross_schema(pre: str, ci: int, cs: int, ratio: int, out: "OrderedDict[str, tuple]") -> None:
    d = ci  # inner width D = heads * (in_dim // heads) = in_dim (transformer.py:272, 300-303)
    out[pre + "norm1_cross.weight"] = (ci,)
    out[pre + "norm1_cross.bias"] = (ci,)
    out[pre + "norm1_src_cross.weight"] = (cs,)
    out[pre + "norm1_src_cross.bias"] = (cs,)
    out[pre + "norm2.weight"] = (ci,)
    out[pre + "norm2.bias"] = (ci,)
    out[pre + "norm2_src.weight"] = (cs,)
    out[pre + "norm2_src.bias"] = (cs,)
    out[pre + "cross_attention.qk.weight"] = (2 * d, ci)
    out[pre + "cross_attention.qk_src.weight"] = (2 * d, cs)
    out[pre + "cross_attention.v.weight"] = (d, ci)
    out[pre + "cross_attention.v_src.weight"] = (d, cs)
    out[pre + "cross_attention.projection.weight"] = (ci, d)
    out[pre + "cross_attention.projection.bias"] = (ci,)
    out[pre + "cross_attention.projection_src.weight"] = (cs, d)
    out[pre + "cross_attention.projection_src.bias"] = (cs,)
    out[pre + "mlp.trg.layers.0.weight"] = (ratio * ci, ci)
    out[pre + "mlp.trg.layers.0.bias"] = (ratio * ci,)
    out[pre + "mlp.trg.layers.2.weight"] = (ci, ratio * ci)
    out[pre + "mlp.trg.layers.2.bias"] = (ci,)
    out[pre + "mlp.src.layers.0.weight"] = (ratio * cs, cs)
    out[pre + "mlp.src.layers.0.bias"] = (ratio * cs,)
    out[pre + "mlp.src.layers.2.weight"] = (cs, ratio * cs)
    out[pre + "mlp.src.layers.2.bias"] = (cs,)


def conj_state_dict_schema(cfg: ConjConfig) -> "OrderedDict[str, tuple]":
    """Parameter names -> shapes in the reference's order (634 tensors)."""
    s: "OrderedDict[str, tuple]" = OrderedDict()
    m = cfg.main
    main = state_dict_schema(m)
    s["main_stream.mask_token"] = main["mask_token"]
    s["main_stream.null_token_enc"] = (1, 1, m.enc_dim)
    s["main_stream.null_token_dec"] = (1, 1, m.dec_dim)
    for k, v in main.items():
        if k != "mask_token":
            s["main_stream." + k] = v
    c = "context_stream."
    s[c + "mask_token"] = (1, 1, cfg.ctx_dec_dim)
    s[c + "null_token_enc"] = (1, 1, cfg.ctx_enc_dim)
    s[c + "null_token_dec"] = (1, 1, cfg.ctx_dec_dim)
    s[c + "encoder.patch_embed.proj.weight"] = (cfg.ctx_enc_dim, cfg.ctx_in_chans, cfg.ctx_tubelet, 1, 1)
    s[c + "encoder.patch_embed.proj.bias"] = (cfg.ctx_enc_dim,)
    for i in range(m.enc_depth):
        _block_schema(f"{c}encoder.blocks.{i}.", cfg.ctx_enc_dim, m.mlp_ratio * cfg.ctx_enc_dim, s)
    s[c + "encoder.norm.weight"] = (cfg.ctx_enc_dim,)
    s[c + "encoder.norm.bias"] = (cfg.ctx_enc_dim,)
    for i in range(m.dec_depth):
        _block_schema(f"{c}decoder.blocks.{i}.", cfg.ctx_dec_dim, m.mlp_ratio * cfg.ctx_dec_dim, s)
    s[c + "decoder.norm.weight"] = (cfg.ctx_dec_dim,)
    s[c + "decoder.norm.bias"] = (cfg.ctx_dec_dim,)
    s[c + "decoder.head.weight"] = (cfg.ctx_out_dim, cfg.ctx_dec_dim)
    s[c + "decoder.head.bias"] = (cfg.ctx_out_dim,)
    s[c + "encoder_to_decoder.weight"] = (cfg.ctx_dec_dim, cfg.ctx_enc_dim)
    # present in the checkpoints, never used on this path (spacetime_separable_pos_embed=True, vmae.py:368-369)
    s[c + "pos_embed_encoder.weight"] = (cfg.ctx_dec_dim, 2 * cfg.ctx_dec_dim)
    s[c + "pos_embed_encoder.bias"] = (cfg.ctx_dec_dim,)
    for i in cfg.enc_cross:
        _cross_schema(f"encoder_conjoining_blocks.{i}-{i}.", m.enc_dim, cfg.ctx_enc_dim, cfg.cross_mlp_ratio, s)
    for i in cfg.dec_cross:
        _cross_schema(f"decoder_conjoining_blocks.{i}-{i}.", m.dec_dim, cfg.ctx_dec_dim, cfg.cross_mlp_ratio, s)
    return s
