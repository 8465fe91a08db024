"""CPU oracle for the IMU-conditioned conjoined padded predictor (BASELINE configs[4]) -- TEST INFRASTRUCTURE ONLY.

Restates `ConjoinedPaddedVisionTransformer.forward` for the `imu400_base_4x4patch_2frames_1tube`
family (cwm/models/VideoMAE/conjoined_vmae.py:889-1011, 852-887, 1230-1243) with its parts:
`PaddedVisionTransformer` null-token padding (:49-165), `ImuEncoder` tokenisation (:1013-1147),
`BidirectionalCrossAttention` / `CrossAttentionTransformerBlock` (cwm/models/transformer.py:253-378,
442-583).  Functional, flat weight dict with the reference's state-dict keys, torch-CPU fp32.
Pinned against outputs of the reference itself (tests/golden/conj_*.npz, make_golden.py).
Same import rule as oracle/vmae_oracle.py: tests / smoke / bench cpu_baseline only.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

from . import vmae_oracle as V


@dataclass(frozen=True)
class ConjSpec:
    main: V.VmaeSpec = V.VmaeSpec(patch=4)           # rgb stream (conjoined_vmae.py:1176-1183, base scaffold :1168-1183)
    main_max_pad: int = 64                            # rgb_4x4_padded_encoder_kwargs (:1196-1197)
    ctx_in_chans: int = 6
    ctx_seq_len: int = 400                            # imu400_encoder_kwargs (:1207-1211)
    ctx_tubelet: int = 16
    ctx_enc_dim: int = 384                            # imu_encoder_kwargs (:1199-1205)
    ctx_dec_dim: int = 192
    ctx_enc_heads: int = 12                           # the scaffold's encoder_num_heads / decoder_num_heads apply to both streams
    ctx_dec_heads: int = 6
    ctx_max_pad: int = 25                             # imu400_padded_encoder_kwargs (:1213-1216)
    enc_cross: Tuple[int, ...] = (0, 3, 6, 9)         # conjoin_encoder_layers=range(0,12,3) (:1240)
    dec_cross: Tuple[int, ...] = (0, 1, 2, 3)         # conjoin_decoder_layers=True (:1241)
    cross_heads: int = 4                              # default_cross_block_kwargs (:215-220)
    cross_mlp_ratio: float = 2.0

    @property
    def ctx_tokens(self) -> int:
        return self.ctx_seq_len // self.ctx_tubelet


IMU400_BASE_4X4 = ConjSpec()


# ----------------------------------------------------------------------------------------------
# a13: null-token padding  (conjoined_vmae.py:49-116)
# ----------------------------------------------------------------------------------------------
def padding_masks(mask: torch.Tensor, max_pad: int, min_pad: int = 0):
    """Returns (full_input_mask [B, N+P], null_mask [B, N+P-vmax]) as `_set_padding_mask` builds them."""
    B, N = mask.shape
    num_vis = (~mask).sum(-1, keepdim=True)
    vmax = int(num_vis.max())
    num_pad = vmax - num_vis + min_pad
    visible_pad = torch.arange(max_pad)[None].expand(B, -1) < num_pad
    if int(num_vis.sum()) == 0:  # all-rows-empty special case: slot 0 visible (:69-85)
        visible_pad = torch.zeros(B, max_pad, dtype=torch.bool)
        visible_pad[:, 0] = True
    padding_mask = ~visible_pad
    vmax = max(vmax, 1)
    min_masked = N - vmax - min_pad
    full_input_mask = torch.cat([mask, padding_mask], -1)
    null_mask = torch.cat([torch.zeros_like(mask[:, :min_masked]), padding_mask], -1)
    return full_input_mask, null_mask


# ----------------------------------------------------------------------------------------------
# a15/a16: bidirectional cross attention block  (transformer.py:314-378, 559-583)
# ----------------------------------------------------------------------------------------------
def cross_attention(x, s, W, pre: str, heads: int):
    D = W[pre + "v.weight"].shape[0]
    hd = D // heads
    scale = hd ** -0.5
    B, N, _ = x.shape
    M = s.shape[1]
    qk = F.linear(x, W[pre + "qk.weight"]).reshape(B, N, heads, 2 * hd).permute(0, 2, 1, 3)
    qk_s = F.linear(s, W[pre + "qk_src.weight"]).reshape(B, M, heads, 2 * hd).permute(0, 2, 1, 3)
    v = F.linear(x, W[pre + "v.weight"]).reshape(B, N, heads, hd).permute(0, 2, 1, 3)
    v_s = F.linear(s, W[pre + "v_src.weight"]).reshape(B, M, heads, hd).permute(0, 2, 1, 3)
    attn = ((qk[..., :hd] * scale) @ qk_s[..., :hd].transpose(-2, -1)).softmax(-1)          # [B,H,N,M]
    attn_s = (qk_s[..., hd:] @ (qk[..., hd:] * scale).transpose(-2, -1)).softmax(-1)        # [B,H,M,N]
    y = (attn @ v_s).permute(0, 2, 1, 3).reshape(B, N, D)
    y_s = (attn_s @ v).permute(0, 2, 1, 3).reshape(B, M, D)
    y = F.linear(y, W[pre + "projection.weight"], W[pre + "projection.bias"])
    y_s = F.linear(y_s, W[pre + "projection_src.weight"], W[pre + "projection_src.bias"])
    return y, y_s


def _ln(x, W, pre):
    return F.layer_norm(x, (x.shape[-1],), W[pre + "weight"], W[pre + "bias"], V.LN_EPS)


def _cross_mlp(x, W, pre):
    h = F.gelu(F.linear(x, W[pre + "layers.0.weight"], W[pre + "layers.0.bias"]))
    return F.linear(h, W[pre + "layers.2.weight"], W[pre + "layers.2.bias"])


def cross_block(x, s, W, pre: str, heads: int):
    """`CrossAttentionTransformerBlock.forward` with with_self_attention=False (gamma_1 = 0), gammas 1."""
    y, y_s = cross_attention(_ln(x, W, pre + "norm1_cross."), _ln(s, W, pre + "norm1_src_cross."), W, pre + "cross_attention.", heads)
    x = x + y
    s = s + y_s
    x = x + _cross_mlp(_ln(x, W, pre + "norm2."), W, pre + "mlp.trg.")
    s = s + _cross_mlp(_ln(s, W, pre + "norm2_src."), W, pre + "mlp.src.")
    return x, s


# ----------------------------------------------------------------------------------------------
# a14: IMU tokenisation  (conjoined_vmae.py:1110-1125; preprocessor.py:199-206)
# ----------------------------------------------------------------------------------------------
def imu_tokenize(W, spec: ConjSpec, imu_bcl: torch.Tensor, pre: str = "context_stream.encoder.") -> torch.Tensor:
    B, Cc, L = imu_bcl.shape
    t = spec.ctx_tubelet
    tok = imu_bcl.reshape(B, Cc, L // t, t).permute(0, 2, 1, 3).reshape(B, L // t, Cc * t)  # K index = c*16 + s
    w = W[pre + "patch_embed.proj.weight"]
    tok = F.linear(tok, w.reshape(w.shape[0], -1), W[pre + "patch_embed.proj.bias"])
    return tok + V.pos_embedding_f32(L // t, spec.ctx_enc_dim)[None]  # torch-fp32 formula (transformer.py:37-52)


# ----------------------------------------------------------------------------------------------
# a17: the whole model
# ----------------------------------------------------------------------------------------------
def conj_forward(W: Dict[str, torch.Tensor], spec: ConjSpec, x_bcthw, mask, x_context, mask_context, output_context: bool = False):
    """x: [B,3,2,H,W] (what the wrapper's `_preprocess` hands the model), mask: bool [B,Nt],
    x_context: [B,6,L], mask_context: bool [B,L/16].  Returns the main-stream output
    [B, Nt + P - vmax, 3*P*P] with rows at masked pad slots zeroed (conjoined_vmae.py:998-1002); with `output_context` the tuple
    (main, context) of `forward(..., output_main=True, output_context=True)` (:990-1006): the context stream's head over ITS masked +
    pad slots, [B, L/16 + Pc - vmax_c, 6*16], pad rows zeroed the same way."""
    mask, mask_context = mask.bool(), mask_context.bool()
    ms, B = spec.main, x_bcthw.shape[0]
    P, Pc = spec.main_max_pad, spec.ctx_max_pad
    full, null = padding_masks(mask, P)
    full_c, null_c = padding_masks(mask_context, Pc)

    # tokenise + pad + gather (pad_and_mask_input :125-134)
    m = "main_stream."
    tok = V.patch_embed(x_bcthw, W[m + "encoder.patch_embed.proj.weight"], W[m + "encoder.patch_embed.proj.bias"])
    tok = tok + V.sinusoid_table(ms.num_tokens, ms.enc_dim)[None]
    tok = torch.cat([tok, W[m + "null_token_enc"].reshape(1, 1, -1).expand(B, P, -1)], 1)
    x = tok[~full].reshape(B, -1, ms.enc_dim)
    c = "context_stream."
    tok_c = imu_tokenize(W, spec, x_context)
    tok_c = torch.cat([tok_c, W[c + "null_token_enc"].reshape(1, 1, -1).expand(B, Pc, -1)], 1)
    s = tok_c[~full_c].reshape(B, -1, spec.ctx_enc_dim)

    # encoder: cross block BEFORE self-attention blocks i in enc_cross (forward_encoder_blocks :543-576)
    for i in range(ms.enc_depth):
        if i in spec.enc_cross:
            x, s = cross_block(x, s, W, f"encoder_conjoining_blocks.{i}-{i}.", spec.cross_heads)
        x = V.block(x, W, f"{m}encoder.blocks.{i}.", ms.enc_heads)
        s = V.block(s, W, f"{c}encoder.blocks.{i}.", spec.ctx_enc_heads)
    x = _ln(x, W, m + "encoder.norm.")
    s = _ln(s, W, c + "encoder.norm.")
    x = F.linear(x, W[m + "encoder_to_decoder.weight"])
    s = F.linear(s, W[c + "encoder_to_decoder.weight"])

    # decoder inputs (_pad_pos_embed :154-165; ctx table = pos_embedding(25, 192), vmae.py:443-449)
    pos = torch.cat([V.sinusoid_table(ms.num_tokens, ms.dec_dim), W[m + "null_token_dec"].reshape(1, -1).expand(P, -1)], 0)[None].expand(B, -1, -1)
    pos_c = torch.cat([V.pos_embedding_f32(spec.ctx_tokens, spec.ctx_dec_dim), W[c + "null_token_dec"].reshape(1, -1).expand(Pc, -1)], 0)[None].expand(B, -1, -1)
    x = torch.cat([x + pos[~full].reshape(B, -1, ms.dec_dim), W[m + "mask_token"].reshape(1, 1, -1) + pos[full].reshape(B, -1, ms.dec_dim)], 1)
    n_masked = int(full[0].sum())
    s = torch.cat([s + pos_c[~full_c].reshape(B, -1, spec.ctx_dec_dim), W[c + "mask_token"].reshape(1, 1, -1) + pos_c[full_c].reshape(B, -1, spec.ctx_dec_dim)], 1)

    # decoder: cross block AFTER blocks i in dec_cross (forward_decoder_blocks :688-720)
    for i in range(ms.dec_depth):
        x = V.block(x, W, f"{m}decoder.blocks.{i}.", ms.dec_heads)
        s = V.block(s, W, f"{c}decoder.blocks.{i}.", spec.ctx_dec_heads)
        if i in spec.dec_cross:
            x, s = cross_block(x, s, W, f"decoder_conjoining_blocks.{i}-{i}.", spec.cross_heads)
    y = F.linear(_ln(x[:, -n_masked:], W, m + "decoder.norm."), W[m + "decoder.head.weight"], W[m + "decoder.head.bias"])
    y = y * (~null)[..., None].to(y)
    if not output_context:
        return y
    n_masked_c = int(full_c[0].sum())
    y_c = F.linear(_ln(s[:, -n_masked_c:], W, c + "decoder.norm."), W[c + "decoder.head.weight"], W[c + "decoder.head.bias"])
    return y, y_c * (~null_c)[..., None].to(y_c)


def predict(W, spec: ConjSpec, x_btchw, mask, x_context, mask_context, normalize: bool = True, frame=-1):
    """`PredictorBasedGenerator.predict` for the padded conjoined predictor (prediction.py:406-454):
    drop the last max_padding_tokens rows, then un-embed with the raw input at visible patches."""
    y = conj_forward(W, spec, V.preprocess(x_btchw, normalize), mask, x_context, mask_context)
    y = y[:, : -spec.main_max_pad]
    out = V.pred_patches_to_video(y, x_btchw, mask, spec.main.patch)
    if frame is not None:
        f = frame % out.shape[1]
        out = out[:, f : f + 1]
    return out
