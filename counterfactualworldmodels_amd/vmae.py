"""Host-side mirror of the reference VMAE predictor classes, backed by libcwm_hip.so.

`PretrainVisionTransformer` keeps the reference's class surface -- constructor semantics of the
factories (`cwm/models/VideoMAE/vmae.py:563-619`), parameter names/shapes (state-dict compatible with
the published checkpoints), attributes read by the wrapper/UI (`patch_size`, `image_size`,
`num_frames`, `mask_size`, `encoder.patch_embed.proj.kernel_size`, ...) and
`forward(x[B,C,T,H,W], mask[B,Nt]) -> [B,Nm,C*P*P]` (`vmae.py:539-560`) -- but holds no compute:
the forward pass is one call into the C ABI (`include/cwm_hip.h: cwm_forward`), which runs the
hand-written HIP kernels.  There is no PyTorch fallback path.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib
from .config import CONFIGS, LN_EPS, VmaeConfig


def trunc_normal_(tensor, mean=0.0, std=1.0):
    # vmae.py:25-26 (timm's trunc_normal_ == torch.nn.init.trunc_normal_)
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=-std, b=std)


def _init_weights(m):
    # vmae.py:100-107 / :212-219
    if isinstance(m, nn.Linear):
        nn.init.xavier_uniform_(m.weight)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.LayerNorm):
        nn.init.constant_(m.bias, 0)
        nn.init.constant_(m.weight, 1.0)


class _NoForward(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError(
            "%s is a parameter container: the computation runs inside libcwm_hip.so via "
            "PretrainVisionTransformer.forward" % type(self).__name__
        )


class Attention(_NoForward):
    """Parameter layout of VideoMAE/utils.py:57-85 (fused qkv without bias, separate q/v bias)."""

    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        self.q_bias = nn.Parameter(torch.zeros(dim))
        self.v_bias = nn.Parameter(torch.zeros(dim))
        self.proj = nn.Linear(dim, dim)


class Mlp(_NoForward):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class Block(_NoForward):
    def __init__(self, dim, num_heads, mlp_ratio):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=LN_EPS)
        self.attn = Attention(dim, num_heads)
        self.norm2 = nn.LayerNorm(dim, eps=LN_EPS)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))


class PatchEmbed(_NoForward):
    def __init__(self, cfg: VmaeConfig):
        super().__init__()
        self.patch_size = (cfg.patch, cfg.patch)
        self.tubelet_size = 1
        self.num_frames = cfg.num_frames
        self.num_patches = cfg.num_tokens
        self.embed_dim = cfg.enc_dim
        self.proj = nn.Conv3d(cfg.in_chans, cfg.enc_dim, kernel_size=(1, cfg.patch, cfg.patch), stride=(1, cfg.patch, cfg.patch))


class PretrainVisionTransformerEncoder(_NoForward):
    def __init__(self, cfg: VmaeConfig):
        super().__init__()
        self.embed_dim = self.num_features = cfg.enc_dim
        self.patch_size = (1, cfg.patch, cfg.patch)
        self.pt, self.ph, self.pw = self.patch_size
        self.patch_embed = PatchEmbed(cfg)
        self.image_size = cfg.img_size[0]
        self.num_patches = cfg.num_tokens
        self.num_frames = cfg.num_frames
        self.blocks = nn.ModuleList([Block(cfg.enc_dim, cfg.enc_heads, cfg.mlp_ratio) for _ in range(cfg.enc_depth)])
        self.norm = nn.LayerNorm(cfg.enc_dim, eps=LN_EPS)
        self.head = nn.Identity()
        self.timestamps = None
        self.apply(_init_weights)


class PretrainVisionTransformerDecoder(_NoForward):
    def __init__(self, cfg: VmaeConfig):
        super().__init__()
        self.embed_dim = self.num_features = cfg.dec_dim
        self.num_classes = cfg.out_dim
        self.patch_size = (cfg.patch, cfg.patch)
        self.blocks = nn.ModuleList([Block(cfg.dec_dim, cfg.dec_heads, cfg.mlp_ratio) for _ in range(cfg.dec_depth)])
        self.norm = nn.LayerNorm(cfg.dec_dim, eps=LN_EPS)
        self.head = nn.Linear(cfg.dec_dim, cfg.out_dim)
        self.apply(_init_weights)


class WeightSync:
    """Mixin: a cheap "has any parameter changed since the last upload?" test for the modules that mirror their parameters into
    the library.  Walking `state_dict()` costs ~0.3 ms per forward for ViT-B (218 tensors) -- GPU idle time whenever the caller
    synchronises between forwards -- so the tensors are listed once, each with the module dict that owns it, and a forward only
    checks (a) that every slot still holds the listed object (`setattr` / `register_parameter` replacing a Parameter, also through
    a SUBMODULE's `.to()` when torch swaps parameter objects) and (b) the (storage pointer, version counter) pairs (`p.data = ...`,
    `encoder.double()`, `copy_`, optimiser steps, `load_state_dict`): ~60 us.  Not visible to it: in-place edits through `.data`
    that keep the storage (`p.data.mul_(2)`): call `sync_weights(force=True)` after those."""

    def _init_weight_sync(self):
        self._plist = None
        self._psig = None
        self.register_load_state_dict_post_hook(lambda module, incompatible_keys: module._forget_params())

    def _forget_params(self):
        self._plist = None

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._plist = None
        return out

    def _param_device(self):
        return self._plist[0][2].device if self._plist else next(self.parameters()).device

    def _params_unchanged(self) -> bool:
        pl = self._plist
        if pl is None:
            return False
        for slots, key, p in pl:
            if slots.get(key) is not p:
                return False
        return tuple((p.data_ptr(), p._version) for _, _, p in pl) == self._psig

    def _remember_params(self):
        pl = []
        for mod in self.modules():
            pl += [(mod._parameters, k, p) for k, p in mod._parameters.items() if p is not None]
            pl += [(mod._buffers, k, b) for k, b in mod._buffers.items() if b is not None and k not in mod._non_persistent_buffers_set]
        self._plist = pl
        self._psig = tuple((p.data_ptr(), p._version) for _, _, p in pl)


class PretrainVisionTransformer(WeightSync, nn.Module):
    """Drop-in for `cwm.models.VideoMAE.vmae.PretrainVisionTransformer` (main_input=None models)."""

    def __init__(self, cfg: VmaeConfig, mode: str = "parity", use_flash_attention: bool = True, **unused):
        super().__init__()
        self.cfg = cfg
        self.mode = mode
        _lib.mode_id(mode)
        self.get_main_input = None
        self.encoder = PretrainVisionTransformerEncoder(cfg)
        self.decoder = PretrainVisionTransformerDecoder(cfg)
        self.encoder_to_decoder = nn.Linear(cfg.enc_dim, cfg.dec_dim, bias=False)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, cfg.dec_dim))
        trunc_normal_(self.mask_token, std=0.02)
        self.timestamps = None
        self.num_frames = cfg.num_frames
        self.num_patches = cfg.num_tokens
        self.num_patches_per_frame = cfg.tokens_per_frame
        self.patch_size = self.encoder.patch_size
        self.image_size = tuple(cfg.img_size)
        self.default_cfg = {}
        self._handle: Optional[int] = None
        self._handle_device: Optional[torch.device] = None
        self._loaded: Dict[str, Tuple[int, int]] = {}
        self._init_weight_sync()

    # ---- reference attribute surface -------------------------------------------------------------
    @property
    def mask_size(self):  # vmae.py:386-390
        return (
            self.num_frames // self.patch_size[0],
            self.image_size[-2] // self.patch_size[-2],
            self.image_size[-1] // self.patch_size[-1],
        )

    def get_num_layers(self):
        return len(self.encoder.blocks)

    # ---- C-ABI plumbing --------------------------------------------------------------------------
    def _library(self):
        """The shared object this module's handle lives in: libcwm_hip.so unless `use_library` chose the development one."""
        lib = getattr(self, "_cwm", None)
        return lib if lib is not None else _lib.get_lib()

    def _check(self, rc):
        _lib.check(rc, self._library())

    def use_library(self, lib):
        """Create this model's handle in another build of the library (tools / tests: `_lib.get_dev_lib()`, whose per-shape tile overrides and
        thread-local switches a handle of the production library never sees).  Call before the first forward; an existing handle is released."""
        self._release()
        object.__setattr__(self, "_cwm", lib)

    def set_option(self, key: str, value: int):
        """One execution option of THIS model (include/cwm_hip.h cwm_model_set_option: "attn_kernel", "gemm_tile", "prune_last_block" ...): per handle, never
        process-wide.  Options set before the first forward are applied when the handle is created.  An unknown key / a refused value raises and leaves nothing behind."""
        if getattr(self, "_handle", None) is not None:  # the library validates; remembered (for a re-created handle) only once it accepted
            self._check(self._library().cwm_model_set_option(self._handle, key.encode(), int(value)))
        else:
            _lib.validate_option(key, int(value))
        self.__dict__.setdefault("_options", {})[key] = int(value)

    def _ensure_handle(self, device: torch.device) -> int:
        lib = self._library()
        if self._handle is not None and self._handle_device == device:
            return self._handle
        self._release()
        c = self.cfg
        ccfg = _lib.CwmConfig(
            c.img_size[0], c.img_size[1], c.patch, c.num_frames, c.in_chans, c.enc_dim, c.enc_depth, c.enc_heads,
            c.dec_dim, c.dec_depth, c.dec_heads, c.mlp_ratio, LN_EPS,
        )
        h = C.c_void_p()
        with torch.cuda.device(device):
            self._check(lib.cwm_model_create(C.byref(ccfg), C.byref(h)))
        self._handle = h.value
        self._handle_device = device
        self._loaded = {}
        for k, v in self.__dict__.get("_options", {}).items():
            self._check(lib.cwm_model_set_option(self._handle, k.encode(), v))
        return self._handle

    def _release(self):
        if getattr(self, "_handle", None) is not None:
            try:
                self._library().cwm_model_destroy(self._handle)
            except Exception:
                pass
            # plain attributes: nn.Module.__setattr__ can already be half torn down when __del__ runs at interpreter exit
            object.__setattr__(self, "_handle", None)
            object.__setattr__(self, "_loaded", {})

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def sync_weights(self, device: Optional[torch.device] = None, force: bool = False) -> int:
        """Push every parameter that changed since the last call into the library (packs to bf16
        hi/lo planes).  Counterpart of `load_state_dict` at the boundary (prediction.py:81-107).
        A change is detected by object identity + (storage pointer, version counter) of every parameter (`WeightSync`):
        `load_state_dict`, `copy_`, optimiser steps, `p.data = t`, `.to()` / `.double()` on the model or a submodule, a replaced
        Parameter.  In-place edits through `.data` that keep the storage (`p.data.mul_(2)`) are NOT visible -- call
        `sync_weights(force=True)` (or `invalidate_weights()`) after such an edit."""
        if device is None:
            device = self._param_device()
        if not force and self._handle is not None and self._handle_device == device and self._params_unchanged():
            return 0
        h = self._ensure_handle(device)
        lib = self._library()
        if force:
            self._loaded = {}
        n = 0
        with torch.cuda.device(device):
            for name, p in self.state_dict(keep_vars=True).items():
                tag = (p.data_ptr(), p._version)
                if self._loaded.get(name) == tag:
                    continue
                t = p.detach()
                if t.dtype != torch.float32 or not t.is_contiguous():
                    t = t.float().contiguous()
                on_dev = 1 if t.is_cuda else 0
                if t.is_cuda and t.device != device:
                    t = t.to(device)
                shape = (C.c_int64 * t.dim())(*t.shape)
                self._check(lib.cwm_model_load_weight(h, name.encode(), t.data_ptr(), on_dev, shape, t.dim()))
                self._loaded[name] = tag
                n += 1
        self._remember_params()
        return n

    def invalidate_weights(self):
        """Forget what has been uploaded: the next forward re-packs every parameter (see `sync_weights`)."""
        self._loaded = {}
        self._plist = None

    def _run(self, x, strides, normalize, mask, n_vis, want_video, xraw=None, check=True, out_tokens=None, out_video=None, weights_synced=False):
        _lib.require_gpu()
        if not x.is_cuda:
            raise RuntimeError("PretrainVisionTransformer.forward needs a CUDA/HIP tensor (no CPU fallback); got %s" % x.device)
        lib = self._library()
        dev = x.device
        if not weights_synced or self._handle is None or self._handle_device != dev:
            self.sync_weights(dev)
        c = self.cfg
        B = x.shape[0]
        Nt = c.num_tokens
        mask = mask.to(device=dev, dtype=torch.bool).reshape(B, -1).contiguous()
        if mask.shape[1] != Nt:
            raise RuntimeError("mask has %d tokens per row, model expects %d" % (mask.shape[1], Nt))
        if n_vis is None:
            n_vis = Nt - int(mask[0].sum().item())
        Nm = Nt - n_vis
        # nothing masked: the reference returns head(norm(x)) for all Nt tokens (vmae.py:250-253)
        y = self._out_buffer(out_tokens, (B, Nm if Nm > 0 else Nt, c.out_dim), dev)
        video = self._out_buffer(out_video, (B, c.num_frames, c.in_chans, c.img_size[0], c.img_size[1]), dev) if want_video else None
        args = _lib.CwmForwardArgs(
            C.sizeof(_lib.CwmForwardArgs), x.data_ptr(), strides[0], strides[1], strides[2], int(normalize), mask.data_ptr(), B, n_vis, y.data_ptr(),
            _lib.ptr(video), _lib.ptr(xraw), _lib.mode_id(self.mode), int(check), _lib.current_stream_handle(dev),
        )
        with torch.cuda.device(dev):
            self._check(lib.cwm_forward(self._handle, C.byref(args)))
        return y, video

    @staticmethod
    def _out_buffer(given, shape, dev):
        """The caller's output tensor (a row slice of a larger result: lets a chunked driver assemble its result without a
        concatenation pass) or a fresh one."""
        if given is None:
            return torch.empty(shape, device=dev, dtype=torch.float32)
        if tuple(given.shape) != tuple(shape) or given.dtype != torch.float32 or given.device != dev or not given.is_contiguous():
            raise RuntimeError("output buffer must be a contiguous float32 tensor of shape %s on %s" % (tuple(shape), dev))
        return given

    @staticmethod
    def _frame_strides(x: torch.Tensor, c_dim: int, t_dim: int):
        """(tensor, (stride_b, stride_c, stride_t)) with H,W contiguous; copies only if needed."""
        if x.dtype != torch.float32:
            x = x.float()
        H, W = x.shape[-2:]
        if x.stride(-1) != 1 or x.stride(-2) != W:
            x = x.contiguous()
        return x, (x.stride(0), x.stride(c_dim), x.stride(t_dim))

    # ---- reference forward: vmae.py:539-560 ------------------------------------------------------
    @torch.no_grad()
    def forward(self, x, mask, timestamps=None, *args, n_vis: Optional[int] = None, check: bool = True, **kwargs):
        """x: float[B,C,T,H,W] (already pre-processed by the caller, any strides), mask: bool[B,Nt]
        with equal visible counts per row.  Returns float[B,Nm,C*P*P]."""
        if x.dim() != 5 or x.shape[1] != self.cfg.in_chans or x.shape[2] != self.cfg.num_frames:
            raise RuntimeError("expected x of shape [B,%d,%d,H,W], got %s" % (self.cfg.in_chans, self.cfg.num_frames, tuple(x.shape)))
        if tuple(x.shape[-2:]) != tuple(self.cfg.img_size):
            raise RuntimeError("input image size %s does not match the model's %s" % (tuple(x.shape[-2:]), self.cfg.img_size))
        x, strides = self._frame_strides(x, 1, 2)
        y, _ = self._run(x, strides, False, mask, n_vis, False, check=check)
        return y

    @torch.no_grad()
    def predict_video(self, x_btchw, mask, normalize: bool = True, n_vis: Optional[int] = None, check: bool = True,
                      out_tokens: Optional[torch.Tensor] = None, out_video: Optional[torch.Tensor] = None, weights_synced: bool = False):
        """Fused wrapper path: raw [B,T,C,H,W] frames in [0,1] -> (tokens [B,Nm,C*P*P], video
        [B,T,C,H,W]) = `_preprocess` + forward + `pred_patches_to_video`
        (prediction.py:304-312, :419-422, :245-259) in one library call."""
        if x_btchw.dim() != 5 or x_btchw.shape[2] != self.cfg.in_chans or x_btchw.shape[1] != self.cfg.num_frames:
            raise RuntimeError("expected x of shape [B,%d,%d,H,W], got %s" % (self.cfg.num_frames, self.cfg.in_chans, tuple(x_btchw.shape)))
        x, strides = self._frame_strides(x_btchw, 2, 1)
        # (weights_synced: the caller has just run sync_weights() itself -- the wrapper does, BEFORE its mask read-back, so that the walk over the
        # parameters overlaps the previous call's kernels instead of delaying this call's first launch)
        return self._run(x, strides, normalize, mask, n_vis, True, xraw=x, check=check, out_tokens=out_tokens, out_video=out_video,
                         weights_synced=weights_synced)

    # ---- execution options ----------------------------------------------------------------------------
    def set_lanes(self, lanes: int):
        """1: every kernel on the current stream; 2 (library default): batches whose halves keep >= 3000 encoder rows (ViT-B/8: batch >= 8) run as two half batches on two HIP
        streams (forked / joined inside the library), which fills the idle time between dependent kernels."""
        if self._handle is None:
            raise RuntimeError("run a forward pass (or sync_weights) before set_lanes")
        self._check(self._library().cwm_model_set_lanes(self._handle, int(lanes)))

    # ---- kernel timing (bench.py roofline) ------------------------------------------------------------
    def timing_enable(self, kclass: int, enable: bool = True):
        if self._handle is None:
            raise RuntimeError("run a forward pass (or sync_weights) before enabling timing")
        self._check(self._library().cwm_timing_enable(self._handle, kclass, int(enable)))

    def timing_collect(self, kclass: int):
        st = _lib.CwmKernelStats()
        self._check(self._library().cwm_timing_collect(self._handle, kclass, C.byref(st)))
        return {"launches": st.launches, "total_ms": st.total_ms, "total_flops": st.total_flops}


# ---- factories (same names / defaults as vmae.py:597-619) -------------------------------------------
def base_16x16patch_2frames_1tube(**kwargs):
    return PretrainVisionTransformer(CONFIGS["base_16x16patch_2frames_1tube"], **kwargs)


def base_8x8patch_2frames_1tube(**kwargs):
    return PretrainVisionTransformer(CONFIGS["base_8x8patch_2frames_1tube"], **kwargs)


def large_4x4patch_2frames_1tube(**kwargs):
    return PretrainVisionTransformer(CONFIGS["large_4x4patch_2frames_1tube"], **kwargs)
