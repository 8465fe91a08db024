"""CPU oracle for the VMAE predictor forward pass  --  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch restatement (PyTorch-CPU fp32, functional, flat
weight dict) of the reference algorithm for the one hot path this repository
accelerates.  It is the *checker*: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  Nothing under
``counterfactualworldmodels_amd/`` imports it, and the product path raises if the
HIP library is missing instead of falling back to this code.

Parity pinning: the upstream reference ships no tests / golden vectors for this
path, so this oracle is pinned against outputs of the reference
itself, imported in the build container by ``tests/golden/make_golden.py`` and
committed as fixtures under ``tests/golden/*.npz`` (checked by
``tests/test_oracle_golden.py``).

Reference citations are ``path:line`` relative to ``/root/reference``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

IMAGENET_MEAN = (0.485, 0.456, 0.406)  # cwm/models/utils.py:12
IMAGENET_STD = (0.229, 0.224, 0.225)  # cwm/models/utils.py:13
LN_EPS = 1e-6  # vmae.py:575,592 (partial(nn.LayerNorm, eps=1e-6))


@dataclass(frozen=True)
class VmaeSpec:
    """Shape of a `PretrainVisionTransformer` (vmae.py:257-384)."""

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
    def out_dim(self) -> int:
        return self.in_chans * self.patch * self.patch  # vmae.py:339 (tubelet 1)


SPECS: Dict[str, VmaeSpec] = {
    # vmae.py:605-611 / :580-595
    "base_8x8patch_2frames_1tube": VmaeSpec(),
    # vmae.py:597-603
    "base_16x16patch_2frames_1tube": VmaeSpec(patch=16),
    # vmae.py:613-619 / :563-578
    "large_4x4patch_2frames_1tube": VmaeSpec(
        patch=4, enc_dim=1024, enc_depth=24, enc_heads=16, dec_dim=512, dec_depth=12, dec_heads=8
    ),
}


# ----------------------------------------------------------------------------------------------
# positional tables
# ----------------------------------------------------------------------------------------------
def sinusoid_table(n_pos: int, d: int) -> torch.Tensor:
    """`get_sinusoid_encoding_table` (VideoMAE/utils.py:251-268): numpy float64, cast to fp32."""
    pos = np.arange(n_pos, dtype=np.float64)[:, None]
    j = np.arange(d)
    denom = np.power(10000.0, 2.0 * (j // 2) / d)[None, :]
    tab = pos / denom
    tab[:, 0::2] = np.sin(tab[:, 0::2])
    tab[:, 1::2] = np.cos(tab[:, 1::2])
    return torch.from_numpy(tab.astype(np.float32))  # [n_pos, d]


def pos_embedding_f32(n_pos: int, d: int) -> torch.Tensor:
    """`pos_embedding` (transformer.py:37-52): the same formula evaluated in torch float32."""
    positions = torch.arange(n_pos).float()
    freqs = torch.arange(d).float()
    freqs = torch.pow(10000, 2 * (torch.div(freqs, 2, rounding_mode="trunc")) / d)
    out = positions[:, None] / freqs[None, :]
    out[:, 0::2] = torch.sin(out[:, 0::2])
    out[:, 1::2] = torch.cos(out[:, 1::2])
    return out  # [n_pos, d]


# ----------------------------------------------------------------------------------------------
# a1: wrapper input -> model input
# ----------------------------------------------------------------------------------------------
def preprocess(x_btchw: torch.Tensor, normalize: bool = True) -> torch.Tensor:
    """`PredictorBasedGenerator._preprocess` (prediction.py:304-312) + `imagenet_normalize`
    (utils.py:15-21): [B,T,C,H,W] in [0,1] -> [B,C,T,H,W] normalised."""
    x = x_btchw.transpose(1, 2)
    if normalize:
        mean = torch.tensor(IMAGENET_MEAN, dtype=x.dtype).view(1, 3, 1, 1, 1)
        std = torch.tensor(IMAGENET_STD, dtype=x.dtype).view(1, 3, 1, 1, 1)
        x = (x - mean) / std
    return x


# ----------------------------------------------------------------------------------------------
# a2: tubelet patch embed  (VideoMAE/utils.py:174-198; Conv3d k = stride = (1,P,P))
# ----------------------------------------------------------------------------------------------
def patchify_cthw(x_bcthw: torch.Tensor, P: int) -> torch.Tensor:
    """[B,C,T,H,W] -> [B, T*h*w, C*P*P]; token order (t,h,w), feature order (c,ph,pw)."""
    B, C, T, H, W = x_bcthw.shape
    h, w = H // P, W // P
    x = x_bcthw.reshape(B, C, T, h, P, w, P)
    x = x.permute(0, 2, 3, 5, 1, 4, 6)  # b t h w c ph pw
    return x.reshape(B, T * h * w, C * P * P)


def patch_embed(x_bcthw: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    P = w.shape[-1]
    tok = patchify_cthw(x_bcthw, P)
    return F.linear(tok, w.reshape(w.shape[0], -1), b)


# ----------------------------------------------------------------------------------------------
# operand-rounding hooks (precision budget of the HIP path; tests/precision_budget.py).  With the default (empty) table every
# product below is plain fp32, i.e. the reference arithmetic.  A scheme rounds the two operands of ONE class of matrix product
# the way an MFMA variant would see them and accumulates in fp32:
#   "bf16"     one plane each (the library's `fast` mode)           "bf16x3"   hi+lo bf16 planes, hi*hi + hi*lo + lo*hi (`parity`)
#   "fp16"     one fp16 plane each                                   "fp16_a2"  first operand hi+lo fp16, second one plane (2 MFMAs)
#   "fp16_b2"  second operand hi+lo fp16, first one plane            "fp16x3"   hi+lo fp16 planes, three products
# classes: "qk" (q k^T), "pv" (softmax . v), "qkv", "proj", "fc1", "fc2" (first operand = activations, second = weights).
# ----------------------------------------------------------------------------------------------
PRECISION: Dict[str, str] = {}


def _split(x: torch.Tensor, dt) -> Tuple[torch.Tensor, torch.Tensor]:
    hi = x.to(dt).float()
    return hi, (x - hi).to(dt).float()


def _mm(cls: str, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a @ b with the operand rounding of `PRECISION[cls]` (fp32 accumulation; exact when the class has no entry)."""
    scheme = PRECISION.get(cls)
    if scheme is None:
        return a @ b
    dt = torch.bfloat16 if scheme.startswith("bf16") else torch.float16
    ah, al = _split(a, dt)
    bh, bl = _split(b, dt)
    if scheme in ("bf16", "fp16"):
        return ah @ bh
    if scheme.endswith("x3"):
        return ah @ bh + (ah @ bl + al @ bh)
    if scheme == "fp16_a2":
        return ah @ bh + al @ bh
    if scheme == "fp16_b2":
        return ah @ bh + ah @ bl
    raise ValueError(scheme)


def _linear(cls: str, x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    if cls not in PRECISION:
        return F.linear(x, w, b)
    y = _mm(cls, x, w.t())
    return y if b is None else y + b


def attention(x: torch.Tensor, W: Dict[str, torch.Tensor], pre: str, heads: int) -> torch.Tensor:
    """`Attention.forward` (VideoMAE/utils.py:87-121): qkv bias = [q_bias | 0 | v_bias]; q scaled
    after bias; dense softmax(q k^T) v; output projection with bias."""
    B, N, C = x.shape
    q_bias = W.get(pre + "q_bias")
    bias = None
    if q_bias is not None:
        bias = torch.cat((q_bias, torch.zeros_like(q_bias), W[pre + "v_bias"]))
    qkv = _linear("qkv", x, W[pre + "qkv.weight"], bias)
    qkv = qkv.reshape(B, N, 3, heads, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    hd = q.shape[-1]
    q = q * (hd ** -0.5)
    if "pv" in PRECISION:
        # the kernels multiply the UN-normalised exp(s - max) with v and divide by the row sum afterwards
        s_ = _mm("qk", q, k.transpose(-2, -1))
        p = torch.exp(s_ - s_.amax(dim=-1, keepdim=True))
        o = _mm("pv", p, v) / p.sum(dim=-1, keepdim=True)
    else:
        attn = _mm("qk", q, k.transpose(-2, -1)).softmax(dim=-1)
        o = attn @ v
    o = o.transpose(1, 2).reshape(B, N, -1)
    return _linear("proj", o, W[pre + "proj.weight"], W[pre + "proj.bias"])


def mlp(x: torch.Tensor, W: Dict[str, torch.Tensor], pre: str) -> torch.Tensor:
    """`Mlp.forward` (VideoMAE/utils.py:47-54): fc2(GELU_erf(fc1(x)))."""
    h = F.gelu(_linear("fc1", x, W[pre + "fc1.weight"], W[pre + "fc1.bias"]))
    return _linear("fc2", h, W[pre + "fc2.weight"], W[pre + "fc2.bias"])


def layer_norm(x: torch.Tensor, W: Dict[str, torch.Tensor], pre: str) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), W[pre + "weight"], W[pre + "bias"], LN_EPS)


def block(x: torch.Tensor, W: Dict[str, torch.Tensor], pre: str, heads: int) -> torch.Tensor:
    """`Block.forward` (VideoMAE/utils.py:146-153) with gamma=None (init_values=0.), no drop-path."""
    x = x + attention(layer_norm(x, W, pre + "norm1."), W, pre + "attn.", heads)
    x = x + mlp(layer_norm(x, W, pre + "norm2."), W, pre + "mlp.")
    return x


# ----------------------------------------------------------------------------------------------
# a3,a7-a10: encoder / decoder / whole model  (vmae.py:152-182, 246-255, 539-560)
# ----------------------------------------------------------------------------------------------
def encoder_forward(W, spec: VmaeSpec, x_bcthw: torch.Tensor, mask: torch.Tensor, pre: str = "encoder.") -> torch.Tensor:
    tok = patch_embed(x_bcthw, W[pre + "patch_embed.proj.weight"], W[pre + "patch_embed.proj.bias"])
    tok = tok + sinusoid_table(tok.shape[1], spec.enc_dim)[None]  # vmae.py:162-165
    B, _, C = tok.shape
    x_vis = tok[~mask].reshape(B, -1, C)  # vmae.py:167
    for i in range(spec.enc_depth):
        x_vis = block(x_vis, W, f"{pre}blocks.{i}.", spec.enc_heads)
    return layer_norm(x_vis, W, pre + "norm.")  # vmae.py:172; head = Identity (:85)


def decoder_forward(W, spec: VmaeSpec, x_full: torch.Tensor, n_return: int, pre: str = "decoder.") -> torch.Tensor:
    for i in range(spec.dec_depth):
        x_full = block(x_full, W, f"{pre}blocks.{i}.", spec.dec_heads)
    if n_return > 0:  # vmae.py:250-253
        x_full = x_full[:, -n_return:]
    return F.linear(layer_norm(x_full, W, pre + "norm."), W[pre + "head.weight"], W[pre + "head.bias"])


def vmae_forward(W: Dict[str, torch.Tensor], spec: VmaeSpec, x_bcthw: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """`PretrainVisionTransformer.forward` (vmae.py:539-560) for main_input=None.

    x: float[B,C,T,H,W] (already normalised), mask: bool[B,Nt] (True = masked).  Returns
    float[B, Nm, C*P*P] with rows in ascending masked-token order and feature order (ph pw c)."""
    mask = mask.bool()
    x_vis = encoder_forward(W, spec, x_bcthw, mask)
    x_vis = F.linear(x_vis, W["encoder_to_decoder.weight"])  # vmae.py:355,547 (no bias)
    B, _, C = x_vis.shape
    pos = sinusoid_table(mask.shape[1], spec.dec_dim)[None].expand(B, -1, -1)  # vmae.py:366,554
    pos_vis = pos[~mask].reshape(B, -1, C)
    pos_msk = pos[mask].reshape(B, -1, C)
    x_full = torch.cat([x_vis + pos_vis, W["mask_token"].reshape(1, 1, C) + pos_msk], dim=1)  # vmae.py:557
    return decoder_forward(W, spec, x_full, pos_msk.shape[1])


# ----------------------------------------------------------------------------------------------
# a11: patch un-embed  (prediction.py:245-259; patches.py:67-109)
# ----------------------------------------------------------------------------------------------
def video_to_patches(x_btchw: torch.Tensor, P: int) -> torch.Tensor:
    """`Patchify.video_to_patches` with temporal_dim=1, pt=1 (patches.py:67-78):
    'b t c (h ph) (w pw) -> b (t h w) (ph pw c)'  (c fastest)."""
    B, T, C, H, W = x_btchw.shape
    h, w = H // P, W // P
    x = x_btchw.reshape(B, T, C, h, P, w, P).permute(0, 1, 3, 5, 4, 6, 2)  # b t h w ph pw c
    return x.reshape(B, T * h * w, P * P * C)


def patches_to_video(p: torch.Tensor, T: int, C: int, H: int, W: int, P: int) -> torch.Tensor:
    """`Patchify.patches_to_video` (patches.py:80-109) -> [B,T,C,H,W]."""
    B = p.shape[0]
    h, w = H // P, W // P
    x = p.reshape(B, T, h, w, P, P, C).permute(0, 1, 6, 2, 4, 3, 5)  # b t c h ph w pw
    return x.reshape(B, T, C, H, W)


def pred_patches_to_video(y: torch.Tensor, x_btchw: torch.Tensor, mask: torch.Tensor, P: int) -> torch.Tensor:
    """`pred_patches_to_video` (prediction.py:245-259): raw input at visible positions, predictions
    at masked positions."""
    mask = mask.bool()
    B, T, C, H, W = x_btchw.shape
    patches = video_to_patches(x_btchw, P).to(y.dtype)
    out = torch.zeros_like(patches)
    out[~mask] = patches[~mask]
    out[mask] = y.reshape(-1, y.shape[-1])
    return patches_to_video(out, T, C, H, W, P)


# ----------------------------------------------------------------------------------------------
# a12: RectangularizeMasks('min')  (masking.py:90-132)  -- bit-exact, uses torch's global RNG
# ----------------------------------------------------------------------------------------------
def rectangularize_masks_min(masks: torch.Tensor) -> torch.Tensor:
    """Equalise #masked per row to the batch minimum by un-masking random masked positions.
    Mutates (a flattened view of) the input like the reference and draws `torch.randperm` from
    the global generator in row order, so with the same seed the result is bit-identical."""
    shape = masks.shape
    m = masks.flatten(1)
    num_masked = m.float().sum(-1)
    M = torch.amin(num_masked).long()
    num_changes = num_masked.long() - M
    for b in range(m.shape[0]):
        nc = int(num_changes[b])
        if nc > 0:
            inds = torch.where(m[b])[0]
            inds = inds[torch.randperm(inds.size(0))[:nc]]
            m[b, inds] = 0
    return m.view(*shape)


# ----------------------------------------------------------------------------------------------
# wrapper-level prediction  (prediction.py:406-454, configs 1-4: non-padded predictor)
# ----------------------------------------------------------------------------------------------
def predict(
    W: Dict[str, torch.Tensor],
    spec: VmaeSpec,
    x_btchw: torch.Tensor,
    mask: torch.Tensor,
    normalize: bool = True,
    frame: Optional[int] = -1,
    return_tokens: bool = False,
):
    """`PredictorBasedGenerator.predict` for a plain VMAE predictor: preprocess, forward,
    un-embed with the raw input at visible patches, optional frame slice (prediction.py:447-449).
    `mask` must already be rectangular (the wrapper calls RectangularizeMasks when B>1)."""
    y_tok = vmae_forward(W, spec, preprocess(x_btchw, normalize), mask)
    y = pred_patches_to_video(y_tok, x_btchw, mask, spec.patch)
    if frame is not None:
        f = frame % y.shape[1]
        y = y[:, f : f + 1]
    return (y, y_tok) if return_tokens else y


# ----------------------------------------------------------------------------------------------
# algorithmic FLOPs -- used by bench.py for the roofline fraction
# ----------------------------------------------------------------------------------------------
def algorithmic_flops(spec: VmaeSpec, n_vis: int) -> float:
    Nt, De, Dd = spec.num_tokens, spec.enc_dim, spec.dec_dim
    K = spec.in_chans * spec.patch * spec.patch
    Nm = Nt - n_vis
    f = 2.0 * Nt * K * De
    f += spec.enc_depth * (24.0 * n_vis * De * De + 4.0 * n_vis * n_vis * De)
    f += 2.0 * n_vis * De * Dd
    f += spec.dec_depth * (24.0 * Nt * Dd * Dd + 4.0 * Nt * Nt * Dd)
    f += 2.0 * Nm * Dd * spec.out_dim
    return f


# ----------------------------------------------------------------------------------------------
# f-1: motion-counterfactual prompt construction (shift the active patches, keep the passive ones)
# `PatchPerturbation.forward` + `ShiftPatchesAndMask.perturb` (perturbation.py:99-113, 245-289),
# driven per sample as in `create_motion_counterfactuals` (segmentation.py:278-344).  Pure copies
# and boolean work: bit-exact.
# ----------------------------------------------------------------------------------------------
def _shift_frame(img: torch.Tensor, dy: int, dx: int, fill: float) -> torch.Tensor:
    """`CenterCrop(F.pad(img, 2*shift on one side))` == translate by (dy, dx) with `fill`
    (perturbation.py:227-243, 263-264): out[y, x] = img[y - dy, x - dx]."""
    H, W = img.shape[-2:]
    out = torch.full_like(img, fill)
    ys0, ys1 = max(0, dy), min(H, H + dy)
    xs0, xs1 = max(0, dx), min(W, W + dx)
    if ys1 > ys0 and xs1 > xs0:
        out[..., ys0:ys1, xs0:xs1] = img[..., ys0 - dy : ys1 - dy, xs0 - dx : xs1 - dx]
    return out


def shift_patches_and_mask(x_1tchw, mask_1n, active_1n, shift_patches, P: int, frame: int = 1):
    """One sample of the reference loop: `self.shifter(x, mask=min(mask, active),
    perturbation_points=~active, mask_shift=shift, frame=frame)`.

    x: [1,T,C,H,W]; mask: bool [1,Nt] (True = masked; the *passive* patches are its zeros);
    active: bool [1,Nt] (False at the active patches); shift_patches = (dy, dx) in patch units."""
    _, T, C, H, W = x_1tchw.shape
    gh, gw = H // P, W // P
    dy, dx = int(shift_patches[0]), int(shift_patches[1])
    points = ~active_1n
    m = torch.minimum(mask_1n, active_1n).clone()
    m[points] = True  # perturbation.py:106: the moved patches are re-masked at their source
    pm = active_1n.view(1, T, gh, gw)  # perturbation mask = ~points
    pm_shift = _shift_frame(pm[:, frame].float(), dy, dx, 1.0).bool()  # mask padded with 1s (:266-268)
    mask_p = pm.clone()
    mask_p[:, frame] = pm_shift
    x_out = x_1tchw.clone()
    shifted = _shift_frame(x_1tchw[:, frame], dy * P, dx * P, 0.0)  # image padded with 0s (:263-264)
    take = (~pm_shift).repeat_interleave(P, -2).repeat_interleave(P, -1).unsqueeze(1)  # [1,1,H,W]
    x_out[:, frame] = torch.where(take, shifted, x_1tchw[:, frame])  # :277-283
    mask_out = torch.minimum(m, mask_p.view(1, -1))  # perturbation.py:110-111
    return x_out, mask_out


def make_static(x_btchw: torch.Tensor, mask_bn: torch.Tensor, P: int) -> torch.Tensor:
    """`MakeStatic.perturb` (perturbation.py:120-145, called at prediction.py:802-803): the patches `mask` leaves visible (0) in every
    frame take the pixels of the same patch in frame 0; masked patches keep theirs.  Pure copies."""
    B, T, C, H, W = x_btchw.shape
    vis = (~mask_bn.bool()).view(B, T, H // P, W // P).repeat_interleave(P, -2).repeat_interleave(P, -1).unsqueeze(2)  # [B,T,1,H,W]
    return torch.where(vis, x_btchw[:, 0:1].expand(-1, T, -1, -1, -1), x_btchw)


def create_motion_counterfactuals(x_btchw, masks_bns, active_bns, shifts, P: int, frame: int = 1, fix_passive: bool = True):
    """`FlowGenerator.create_motion_counterfactuals` (segmentation.py:278-344) before the final
    `mask_rectangularizer` call: returns (x_shift [B*S,...], mask_shift [B*S,Nt]) in '(b s)' order."""
    B, N, S = masks_bns.shape
    if fix_passive:  # make_static_movie(x[:,0:1], T=2)  (prediction.py:731-739)
        x_btchw = x_btchw[:, 0:1].repeat(1, 2, 1, 1, 1)
    xs, ms = [], []
    for b in range(B):
        for s in range(S):
            xo, mo = shift_patches_and_mask(x_btchw[b : b + 1], masks_bns[b : b + 1, :, s], active_bns[b : b + 1, :, s],
                                            shifts[b * S + s], P, frame)
            xs.append(xo)
            ms.append(mo)
    return torch.cat(xs, 0), torch.cat(ms, 0)
