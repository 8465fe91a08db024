"""Host-side mirror of the IMU-conditioned conjoined padded predictor, backed by libcwm_hip.so.

`ConjoinedPaddedVisionTransformer` / `imu400_base_4x4patch_2frames_1tube` keep the reference's surface
(`cwm/models/VideoMAE/conjoined_vmae.py:889-1011, 1230-1243`): the 634 state-dict keys and shapes of
the published checkpoint, `forward(x, mask, timestamps=None, x_context=None,
mask_context=None, output_main=None, output_context=None)` returning the main-stream tokens
`[B, Nt + 64 - max_visible, 48]` with rows at masked pad slots zeroed, and the attributes the wrapper
reads (`main_stream`, `context_stream`, `max_padding_tokens`, `min_padding_tokens`, `padding_mask`,
`_reset_padding_mask`, `get_current_inputs`, `patch_size`, `image_size`, `num_frames`, `mask_size`).
The parameter tree holds no compute: the forward pass is one `cwm_conj_forward` call.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib
from .config import CONJ_CONFIGS, LN_EPS, ConjConfig, conj_state_dict_schema
from .vmae import WeightSync


class _Params(nn.Module):
    """Parameter container (one node of the reference's module tree); never called."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container: the computation runs inside libcwm_hip.so")


def _init_param(name: str, p: torch.Tensor) -> None:
    if "norm" in name and name.endswith("weight"):
        nn.init.ones_(p)
    elif name.endswith("bias"):
        nn.init.zeros_(p)
    elif "token" in name:
        nn.init.trunc_normal_(p, std=0.02, a=-0.02, b=0.02)
    elif p.dim() >= 2:
        nn.init.xavier_uniform_(p.view(p.shape[0], -1))
    else:
        nn.init.zeros_(p)


def _build_tree(root: nn.Module, schema) -> None:
    for key, shape in schema.items():
        node = root
        parts = key.split(".")
        for part in parts[:-1]:
            if part not in node._modules:
                node.add_module(part, _Params())
            node = node._modules[part]
        p = nn.Parameter(torch.empty(shape))
        _init_param(key, p.data)
        node.register_parameter(parts[-1], p)


class ConjoinedPaddedVisionTransformer(WeightSync, nn.Module):
    def __init__(self, cfg: ConjConfig, mode: str = "parity", **unused):
        super().__init__()
        self.cfg = cfg
        self.mode = mode
        _lib.mode_id(mode)
        _build_tree(self, conj_state_dict_schema(cfg))
        m = cfg.main
        ms, cs = self.main_stream, self.context_stream
        ms.max_padding_tokens, ms.min_padding_tokens = cfg.main_max_pad, 0
        cs.max_padding_tokens, cs.min_padding_tokens = cfg.ctx_max_pad, 0
        ms.patch_size = (1, m.patch, m.patch)
        ms.image_size = tuple(m.img_size)
        ms.num_frames = m.num_frames
        ms.num_patches = m.num_tokens
        ms.padding_mask = ms.full_input_mask = ms.null_mask = None
        ms._reset_padding_mask = lambda: self._reset_stream(ms)
        cs.patch_size = (cfg.ctx_tubelet, 1, 1)
        cs.image_size = (1, 1)
        cs.num_frames = 0
        cs.padding_mask = cs.full_input_mask = cs.null_mask = None
        cs._reset_padding_mask = lambda: self._reset_stream(cs)
        cs.encoder.num_tokens = cfg.ctx_tokens
        cs.encoder.sequence_length = cfg.ctx_seq_len
        self.num_frames = m.num_frames
        self.get_context_input = type("IMU", (), {"num_channels": cfg.ctx_in_chans, "num_frames": None})()
        self.get_main_input = type("RGB01", (), {"num_channels": m.in_chans, "num_frames": m.num_frames})()
        self._output_main, self._output_context = True, False
        self.default_cfg = {}
        self._handle: Optional[int] = None
        self._handle_device: Optional[torch.device] = None
        self._loaded: Dict[str, Tuple[int, int]] = {}
        self._init_weight_sync()

    # ---- reference attribute surface ---------------------------------------------------------------
    @staticmethod
    def _reset_stream(s):
        s.padding_mask = s.full_input_mask = s.null_mask = None

    def _reset_padding_mask(self):
        self._reset_stream(self.main_stream)
        self._reset_stream(self.context_stream)

    @property
    def patch_size(self):
        return self.main_stream.patch_size

    @property
    def image_size(self):
        return self.main_stream.image_size

    @image_size.setter
    def image_size(self, v):
        self.main_stream.image_size = tuple(v)

    @property
    def padding_mask(self):
        # the reference's __getattr__ falls through to the main stream and raises while it is None
        # (conjoined_vmae.py:347-354), so hasattr(model, 'padding_mask') is False before a forward
        if self.main_stream.padding_mask is None:
            raise AttributeError("no attr padding_mask in the module or the main transformer stream")
        return self.main_stream.padding_mask

    @property
    def max_padding_tokens(self):
        return self.main_stream.max_padding_tokens

    @property
    def min_padding_tokens(self):
        return self.main_stream.min_padding_tokens

    @property
    def mask_size(self):  # conjoined_vmae.py:356-360
        ps = self.main_stream.patch_size
        return (self.num_frames // ps[0], self.main_stream.image_size[-2] // ps[-2], self.main_stream.image_size[-1] // ps[-1])

    def get_current_inputs(self, x, mask, *args, **kwargs):
        """conjoined_vmae.py:722-732 with output_main only: the main stream sees (x, mask) unchanged ('rgb01')."""
        return ((x, mask, None),)

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
        """One execution option of THIS model (include/cwm_hip.h cwm_conj_set_option: "attn_kernel", "gemm_tile", "prune_last_block" ...): per handle, never
        process-wide.  Options set before the first forward are applied when the handle is created.  An unknown key / a refused value raises and leaves nothing behind."""
        if getattr(self, "_handle", None) is not None:  # the library validates; remembered (for a re-created handle) only once it accepted
            self._check(self._library().cwm_conj_set_option(self._handle, key.encode(), int(value)))
        else:
            _lib.validate_option(key, int(value))
        self.__dict__.setdefault("_options", {})[key] = int(value)

    def _ensure_handle(self, device: torch.device) -> int:
        lib = self._library()
        if self._handle is not None and self._handle_device == device:
            return self._handle
        self._release()
        c, m = self.cfg, self.cfg.main
        cc = _lib.CwmConjConfig()
        cc.main = _lib.CwmConfig(m.img_size[0], m.img_size[1], m.patch, m.num_frames, m.in_chans, m.enc_dim, m.enc_depth, m.enc_heads,
                                 m.dec_dim, m.dec_depth, m.dec_heads, m.mlp_ratio, LN_EPS)
        cc.main_max_pad = c.main_max_pad
        cc.ctx_in_chans, cc.ctx_seq_len, cc.ctx_tubelet = c.ctx_in_chans, c.ctx_seq_len, c.ctx_tubelet
        cc.ctx_enc_dim, cc.ctx_dec_dim, cc.ctx_enc_heads, cc.ctx_dec_heads = c.ctx_enc_dim, c.ctx_dec_dim, c.ctx_enc_heads, c.ctx_dec_heads
        cc.ctx_max_pad = c.ctx_max_pad
        cc.n_enc_cross, cc.n_dec_cross = len(c.enc_cross), len(c.dec_cross)
        for i, v in enumerate(c.enc_cross):
            cc.enc_cross[i] = v
        for i, v in enumerate(c.dec_cross):
            cc.dec_cross[i] = v
        cc.cross_heads, cc.cross_mlp_ratio = c.cross_heads, c.cross_mlp_ratio
        h = C.c_void_p()
        with torch.cuda.device(device):
            self._check(lib.cwm_conj_create(C.byref(cc), C.byref(h)))
        self._handle, self._handle_device, self._loaded = h.value, device, {}
        for k, v in self.__dict__.get("_options", {}).items():
            self._check(lib.cwm_conj_set_option(self._handle, k.encode(), v))
        return self._handle

    def _release(self):
        if getattr(self, "_handle", None) is not None:
            try:
                self._library().cwm_conj_destroy(self._handle)
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
        """See `vmae.PretrainVisionTransformer.sync_weights` (in-place `.data` edits need force=True)."""
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
                shape = (C.c_int64 * t.dim())(*t.shape)
                self._check(lib.cwm_conj_load_weight(h, name.encode(), t.data_ptr(), 1 if t.is_cuda else 0, shape, t.dim()))
                self._loaded[name] = tag
                n += 1
        self._remember_params()
        return n

    # ---- reference forward: conjoined_vmae.py:852-887 ----------------------------------------------
    @torch.no_grad()
    def forward(self, x, mask, timestamps=None, x_context=None, mask_context=None, output_main=None, output_context=None,
                *args, normalize: bool = False, check: bool = True, n_vis: Optional[int] = None, n_vis_context: Optional[int] = None,
                **kwargs):
        """`n_vis` / `n_vis_context`: the caller knows that every row of `mask` / `mask_context` has exactly this many visible
        tokens (a rectangularised batch; `mask_context=None` means all visible): skips the host read-back of the row counts."""
        _lib.require_gpu()
        # `_set_decoder_outputs` (conjoined_vmae.py:589-593): a flag that is given replaces the model's setting and STAYS replaced
        if output_main is not None:
            self._output_main = bool(output_main)
        if output_context is not None:
            self._output_context = bool(output_context)
        want_main, want_ctx = self._output_main, self._output_context
        if not want_main and not want_ctx:  # "return all the tokens from both streams" (:1010-1011): the same tuple in the padded model
            want_main = want_ctx = True
        if x_context is None:
            raise RuntimeError("the IMU-conditioned predictor needs x_context [B,%d,%d]" % (self.cfg.ctx_in_chans, self.cfg.ctx_seq_len))
        if not x.is_cuda:
            raise RuntimeError("ConjoinedPaddedVisionTransformer.forward needs CUDA/HIP tensors (no CPU fallback); got %s" % x.device)
        c, m = self.cfg, self.cfg.main
        if x.dim() != 5 or x.shape[1] != m.in_chans or x.shape[2] != m.num_frames or tuple(x.shape[-2:]) != tuple(m.img_size):
            raise RuntimeError("expected x of shape [B,%d,%d,%d,%d], got %s" % (m.in_chans, m.num_frames, m.img_size[0], m.img_size[1], tuple(x.shape)))
        dev, B, Nt = x.device, x.shape[0], m.num_tokens
        self.sync_weights(dev)
        if x.dtype != torch.float32:
            x = x.float()
        if x.stride(-1) != 1 or x.stride(-2) != x.shape[-1]:
            x = x.contiguous()
        mask = mask.to(device=dev, dtype=torch.bool).reshape(B, -1).contiguous()
        if mask.shape[1] != Nt:
            raise RuntimeError("mask has %d tokens per row, model expects %d" % (mask.shape[1], Nt))
        ctx = x_context.to(device=dev, dtype=torch.float32).reshape(B, c.ctx_in_chans, c.ctx_seq_len).contiguous()
        mask_context_given = mask_context is not None
        if mask_context is None:
            mask_context = torch.zeros(B, c.ctx_tokens, dtype=torch.bool, device=dev)
        mc = mask_context.to(device=dev, dtype=torch.bool).reshape(B, c.ctx_tokens).contiguous()
        vis = (~mask).sum(-1)
        vis_c = (~mc).sum(-1)
        if not mask_context_given:
            n_vis_context = c.ctx_tokens
        if n_vis is not None and n_vis_context is not None:
            vmax = vmin = int(n_vis)
            vcmax = vcmin = int(n_vis_context)
        else:
            vmax, vmin, vcmax, vcmin = (int(v) for v in torch.stack([vis.max(), vis.min(), vis_c.max(), vis_c.min()]).tolist())
        if vmax - vmin > c.main_max_pad or vcmax - vcmin > c.ctx_max_pad:
            raise RuntimeError("visible-token counts differ by more than max_padding_tokens (%d / %d)" % (c.main_max_pad, c.ctx_max_pad))
        if vmax < 1 or vcmax < 1:
            raise RuntimeError("every stream needs at least one visible token")
        n_out = Nt + c.main_max_pad - vmax
        y = torch.empty((B, n_out, m.out_dim), device=dev, dtype=torch.float32)
        # the context stream's predictions: head(norm(x_c[:, -n:])) * ~null_mask over its masked + pad slots (conjoined_vmae.py:990-1002)
        y_ctx = torch.empty((B, c.ctx_tokens + c.ctx_max_pad - vcmax, c.ctx_out_dim), device=dev, dtype=torch.float32) if want_ctx else None
        args_ = _lib.CwmConjForwardArgs(
            C.sizeof(_lib.CwmConjForwardArgs),
            x.data_ptr(), x.stride(0), x.stride(1), x.stride(2), int(normalize), mask.data_ptr(), B, vmax, ctx.data_ptr(), mc.data_ptr(),
            vcmax, y.data_ptr(), _lib.mode_id(self.mode), int(check), _lib.current_stream_handle(dev), _lib.ptr(y_ctx))
        with torch.cuda.device(dev):
            self._check(self._library().cwm_conj_forward(self._handle, C.byref(args_)))
        self._record_padding_state(mask, vis, vmax, mc, vis_c, vcmax)
        if want_main and want_ctx:
            return y, y_ctx
        return y if want_main else y_ctx

    def _record_padding_state(self, mask, vis, vmax, mask_ctx=None, vis_ctx=None, vmax_ctx=None):
        """The padding attributes the reference leaves set on BOTH streams after a forward until the wrapper resets them
        (prediction.py:451-452; conjoined_vmae.py:49-116): `padding_mask` (pad slot j of row b is masked unless
        j < max visible - visible(b)), `full_input_mask` = [mask | padding_mask], `null_mask` = [zeros(min masked) | padding_mask]."""
        def record(stream, m, v, vm, max_pad):
            B, N = m.shape
            pad = torch.arange(max_pad, device=m.device)[None] >= (vm - v)[:, None]
            stream.padding_mask = pad
            stream.full_input_mask = torch.cat([m, pad], -1)
            stream.null_mask = torch.cat([torch.zeros(B, N - vm, dtype=torch.bool, device=m.device), pad], -1)

        record(self.main_stream, mask, vis, vmax, self.cfg.main_max_pad)
        if mask_ctx is not None:
            record(self.context_stream, mask_ctx, vis_ctx, vmax_ctx, self.cfg.ctx_max_pad)

    def set_lanes(self, lanes: int):
        """See `vmae.PretrainVisionTransformer.set_lanes`."""
        if self._handle is None:
            raise RuntimeError("run a forward pass (or sync_weights) before set_lanes")
        self._check(self._library().cwm_conj_set_lanes(self._handle, int(lanes)))

    def timing_enable(self, kclass: int, enable: bool = True):
        self._check(self._library().cwm_conj_timing_enable(self._handle, kclass, int(enable)))

    def timing_collect(self, kclass: int):
        st = _lib.CwmKernelStats()
        self._check(self._library().cwm_conj_timing_collect(self._handle, kclass, C.byref(st)))
        return {"launches": st.launches, "total_ms": st.total_ms, "total_flops": st.total_flops}


def imu400_base_4x4patch_2frames_1tube(**kwargs):
    """conjoined_vmae.py:1230-1243"""
    return ConjoinedPaddedVisionTransformer(CONJ_CONFIGS["imu400_base_4x4patch_2frames_1tube"], **kwargs)
