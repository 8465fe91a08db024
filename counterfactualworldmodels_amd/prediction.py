"""The "Predictor class surface": `PredictorBasedGenerator` over the HIP predictor.

Mirrors the part of `cwm/models/prediction.py:17-540,703-739` that sits directly around the
predictor call (`predict`, `_preprocess`, `pred_patches_to_video`, mask/shape helpers, per-sample
batching), with the same method names, argument meaning and error behaviour, so notebook code
written against the reference wrapper runs unchanged on a `counterfactualworldmodels_amd.vmae`
predictor.  All floating-point work is delegated to libcwm_hip.so; when the predictor is the HIP
`PretrainVisionTransformer`, `predict` uses the fused path (normalise + forward + un-embed in one
library call).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch
from torch import nn

from . import _lib
from .config import IMAGENET_MEAN, IMAGENET_STD
from .masking import RectangularizeMasks, upsample_masks
from .vmae import PretrainVisionTransformer


def imagenet_normalize(x, temporal_dim=1):
    """cwm/models/utils.py:15-21"""
    mean = torch.as_tensor(IMAGENET_MEAN).to(x.device)[None, None, :, None, None].to(x)
    std = torch.as_tensor(IMAGENET_STD).to(x.device)[None, None, :, None, None].to(x)
    if temporal_dim == 2:
        mean = mean.transpose(1, 2)
        std = std.transpose(1, 2)
    return (x - mean) / std


class PredictorBasedGenerator(nn.Module):
    """Wrapper for factual / counterfactual predictions from a masked predictor (prediction.py:17)."""

    def __init__(
        self,
        predictor=None,
        predictor_load_path=None,
        imagenet_normalize_inputs=False,
        temporal_dim=2,
        seed=0,
        mask_generator=None,
        max_shift_fraction=0.15,
        **kwargs,
    ):
        super().__init__()
        self.set_predictor(predictor, predictor_load_path)
        self.imagenet_normalize_inputs = imagenet_normalize_inputs
        self.set_temporal_dim(temporal_dim)
        self.rng = np.random.RandomState(seed=seed)
        self.torch_rng = torch.manual_seed(seed)  # the reference seeds the global RNG here (prediction.py:45)
        self.seed = seed
        self.mask_generator = mask_generator
        self.mask_rectangularizer = RectangularizeMasks("min")
        self.max_shift_fraction = max_shift_fraction
        self._shift_rng = np.random.RandomState(seed=0)  # PatchPerturbation.__init__ default seed (perturbation.py:26-27)
        self.shifts = None
        self.x, self.mask, self.timestamps = None, None, None

    # ---- predictor management (prediction.py:75-107) ---------------------------------------------
    def set_predictor(self, net, predictor_load_path=None):
        if net is None:
            raise ValueError("There is no predictor set for this generator and no model to load to")
        self.predictor = net
        self.load_predictor(predictor_load_path)
        self.x = self.mask = self.inp_shape = None

    def load_predictor(self, load_path=None, model=None, map_location="cpu"):
        if load_path is None:
            if hasattr(self.predictor, "_predictor_load_path"):
                self._predictor_load_path = self.predictor._predictor_load_path
            return
        weights = torch.load(load_path, map_location=torch.device(map_location))
        if "model" in weights.keys():
            weights = weights["model"]
        did_load = (model or self.predictor).load_state_dict(weights)
        if model is None:
            self._predictor_load_path = load_path
        print(did_load, load_path)

    # ---- shapes / attributes (prediction.py:131-207) ---------------------------------------------------
    @property
    def patch_size(self):
        if hasattr(self.predictor, "patch_size"):
            return self.predictor.patch_size
        return self.predictor.encoder.patch_embed.proj.kernel_size

    @property
    def image_size(self):
        return self.predictor.image_size

    @property
    def sequence_length(self):
        if hasattr(self.predictor, "sequence_length"):
            return self.predictor.sequence_length
        return getattr(self.predictor, "num_frames", 2)

    @property
    def mask_shape(self):
        if hasattr(self.predictor, "mask_shape"):
            return self.predictor.mask_shape
        pt, ph, pw = self.patch_size
        return (self.sequence_length // pt, self.inp_shape[-2] // ph, self.inp_shape[-1] // pw)

    @property
    def inp_mask_shape(self):
        return (self.x.shape[0], int(np.prod(self.mask_shape)))

    def set_temporal_dim(self, t_dim=1):
        if t_dim == 1:
            self.predictor.t_dim, self.predictor.c_dim = 1, 2
        elif t_dim == 2:
            self.predictor.c_dim, self.predictor.t_dim = 1, 2
        else:
            raise ValueError("temporal_dim must be 1 or 2")

    @property
    def t_dim(self):
        return self.predictor.t_dim

    @property
    def c_dim(self):
        return self.predictor.c_dim

    def set_image_size(self, *args, **kwargs):
        if hasattr(self.predictor, "set_image_size"):
            self.predictor.set_image_size(*args, **kwargs)
        else:
            self.predictor.image_size = args[0]

    # ---- masks (prediction.py:109-119, 216-229, 357-384) ------------------------------------------------
    def generate_mask(self, x=None):
        assert self.mask_generator is not None
        if x is None:
            x = self.x
        mask = self.mask_generator(x).view(x.size(0), -1).to(x.device)
        return self.mask_rectangularizer(mask)

    def set_new_mask(self, x=None):
        self.mask = self.generate_mask(self.x if x is None else x)

    def reset_padding_masks(self):
        """prediction.py:121-129"""
        if hasattr(self.predictor, "main_stream"):
            self.predictor._reset_padding_mask()

    def get_zeros_mask(self, x=None, frame=-1):
        if x is None:
            x = self.x
        if self.inp_shape is None:
            self.inp_shape = x.shape
        mask = torch.zeros(self.mask_shape, device=x.device, dtype=torch.bool)
        if frame is not None:
            mask[frame, ...] = True
        return mask.flatten().unsqueeze(0).expand(x.shape[0], -1)

    def get_fully_visible_mask(self, x=None):
        if x is None:
            x = self.x
        return torch.zeros(self.mask_shape, device=x.device, dtype=torch.bool)

    def get_mask_image(self, mask, upsample=False, invert=False, shape=None):
        if shape is None:
            shape = self.mask_shape
        mask = mask.view(-1, *shape)
        if upsample:
            mask = upsample_masks(mask.view(mask.size(0), -1, *self.mask_shape[-2:]).float(), self.inp_shape[-2:])
        if invert:
            mask = 1 - mask
        return mask

    # ---- the path itself -----------------------------------------------------------------------------
    def _preprocess(self, x):
        """prediction.py:304-312"""
        if self.t_dim != 1:
            x = x.transpose(self.t_dim, self.c_dim)
        if self.imagenet_normalize_inputs:
            x = imagenet_normalize(x, temporal_dim=self.t_dim)
        return x

    def pred_patches_to_video(self, y, x, mask):
        """input at visible positions, preds at masked positions (prediction.py:245-259) -- HIP scatter."""
        _lib.require_gpu()
        B, T, Cc, H, W = x.shape
        P = self.patch_size[-1]
        dev = y.device
        x = x.to(device=dev, dtype=torch.float32).contiguous()
        y = y.to(torch.float32).contiguous()
        mask = mask.to(device=dev, dtype=torch.bool).reshape(B, -1).contiguous()
        n_vis = mask.shape[1] - int(mask[0].sum().item())
        out = torch.empty_like(x)
        with torch.cuda.device(dev):
            _lib.check(
                _lib.get_lib().cwm_unembed(
                    y.data_ptr(), x.data_ptr(), mask.data_ptr(), B, T, Cc, H, W, P, n_vis, out.data_ptr(),
                    _lib.current_stream_handle(dev),
                )
            )
        return out

    def predict(self, x=None, mask=None, frame=-1, reset_masks=True, *args, **kwargs):
        """prediction.py:406-454 for a non-padded predictor."""
        if x is None:
            x = self.x
        if mask is None:
            mask = self.generate_mask(x)
        self.set_image_size(x.shape[-2:])
        self.inp_shape = x.shape
        n_masked = None
        if x.size(0) > 1:
            mask = self.mask_rectangularizer(mask)
            n_masked = getattr(self.mask_rectangularizer, "last_num_masked", None)

        fused = isinstance(self.predictor, PretrainVisionTransformer) and self.t_dim == 2 and not args and not kwargs
        if fused:
            # the rectangularizer has just made every row hold `n_masked` masked tokens (read back in its one host sync), so the
            # library needs neither the n_vis round trip nor the device-side row check (which would synchronise again)
            n_tok = mask[0].numel()
            if n_masked is not None:
                _, y = self.predictor.predict_video(x, mask, normalize=self.imagenet_normalize_inputs, n_vis=n_tok - n_masked, check=False)
            else:
                _, y = self.predictor.predict_video(x, mask, normalize=self.imagenet_normalize_inputs)
        else:
            y = self.predictor(self._preprocess(x), mask, *args, **kwargs)
            if hasattr(self.predictor, "main_stream"):  # padded conjoined predictor: drop the pad rows (prediction.py:424-428)
                num_pad = self.predictor.main_stream.max_padding_tokens - self.predictor.main_stream.min_padding_tokens
                y = y[:, :-num_pad]
            if len(y.shape) != 5:
                # NB the reference un-embeds with unnormalize(normalize(x)) (prediction.py:436-446); we use the raw
                # input itself, which differs from that by at most one fp32 ulp at the visible pixels
                y = self.pred_patches_to_video(y, x, mask=mask)
        if frame is not None:
            frame = frame % y.size(1)
            y = y[:, frame : frame + 1]
        if reset_masks:
            self.reset_padding_masks()
        return y

    def predict_tokens(self, x, mask):
        """Raw predictor output [B,Nm,C*P*P] for wrapper-level input (the seam of prediction.py:419)."""
        mask = mask if (x.size(0) == 1) else self.mask_rectangularizer(mask)
        return self.predictor(self._preprocess(x), mask)

    # ---- per-sample batching (prediction.py:456-540) ----------------------------------------------------
    def predict_per_sample(self, x, masks, frame=-1, batch_size=None, split_samples=True, *args, **kwargs):
        """Run predictions in parallel for S sample masks [B,N,S]."""
        assert len(masks.shape) == 3, masks.shape
        S = masks.size(-1)
        if x is None:
            x = self.x
        B = x.size(0)
        BS = B * S
        x = x[:, None].expand(-1, S, -1, -1, -1, -1).reshape(BS, *x.shape[1:])
        masks = masks.transpose(1, 2).reshape(BS, -1)
        y = self.predict(x=x, mask=masks, frame=frame, *args, **kwargs)
        if not split_samples:
            return y
        p_dims = tuple(range(2, len(y.shape) + 1))
        return y.view(B, S, *y.shape[1:]).permute(0, *p_dims, 1)

    def sample_tile(self, z, num_samples):
        S = num_samples
        rank = len(z.shape)
        return z[:, None].expand(-1, S, *([-1] * (rank - 1))).reshape(-1, *z.shape[1:])

    def sample_tile_all_tensors(self, num_samples, **kwargs):
        return {kw: self.sample_tile(v, num_samples) if isinstance(v, torch.Tensor) else v for kw, v in kwargs.items()}

    def batch_predict_per_sample(self, x, masks, frame=-1, batch_size=None, sample_dim=None, **kwargs):
        S = masks.size(-1) if sample_dim != 0 else masks.size(0)
        if batch_size is None:
            batch_size = S
        else:
            batch_size = max(1, batch_size // x.size(0))
        ys = []
        for b in range(int(np.ceil(S / batch_size))):
            b0, b1 = b * batch_size, (b + 1) * batch_size
            if sample_dim != 0:
                ys.append(
                    self.predict_per_sample(
                        x, masks=masks[..., b0:b1], split_samples=True, frame=frame,
                        **self.sample_tile_all_tensors(masks[..., b0:b1].size(-1), **kwargs),
                    )
                )
            else:
                assert x.size(0) in (masks.size(0), masks.size(-1)), (x.shape, masks.shape)
                _masks = masks[b0:b1] if len(masks.shape) == 2 else masks[..., b0:b1].permute(0, 2, 1).reshape(-1, masks.shape[1])
                ys.append(
                    self.predict(
                        x[b0:b1], mask=_masks, frame=frame, reset_masks=True,
                        **self.sample_tile_all_tensors(x[b0:b1].size(0), **kwargs),
                    )
                )
            self.reset_padding_masks()
        return torch.cat(ys, -1 if sample_dim != 0 else 0)

    # ---- inputs (prediction.py:703-739) -------------------------------------------------------------------
    def set_input(self, x, mask=None, make_mask=False, timestamps=None):
        shape = x.shape
        if len(shape) == 4:
            x = x.unsqueeze(1)
        else:
            assert len(shape) == 5, "Input must be a movie of shape [B,T,C,H,W] or a single frame of shape [B,C,H,W]"
        self.inp_shape = x.shape
        self.x = x
        self.B, self.T, self.C = self.inp_shape[0], self.inp_shape[1], self.inp_shape[2]
        if mask is not None:
            self.mask = mask
        elif make_mask:
            assert self.mask_generator is not None, "You need to have a mask generator to set a new mask"
            self.set_new_mask(self.x)
        if timestamps is not None:
            self.timestamps = timestamps

    def get_static_input(self, x=None):
        if x is None:
            x = self.x
        return torch.tile(x[:, 0:1], (1, x.size(1), 1, 1, 1))

    def make_static_movie(self, x=None, T=None, frame=0):
        if x is None:
            x = self.x
        if T is None:
            T = getattr(self.predictor, "num_frames", 2)
        if len(x.shape) == 4:
            x = x[:, None]
        assert len(x.shape) == 5, "x must be of shape [B,C,H,W] or [B,T,C,H,W], but is %s" % (x.shape,)
        return torch.tile(x[:, frame % x.size(1), None], (1, T, 1, 1, 1))

    # ---- motion counterfactuals ---------------------------------------------------
    def reset_shifts(self):
        self.shifts = []

    def _random_mask_shift(self):
        """`ShiftPatchesAndMask.get_random_shift(is_mask_shift=True)` (perturbation.py:209-225), integer shifts."""
        max_shift = [int(self.max_shift_fraction * s) for s in self.inp_shape[-2:]]
        shift = (0, 0)
        while sum(shift) == 0:  # (sic) the reference rejects any shift whose components sum to zero
            shift = (
                int(self._shift_rng.randint(-max_shift[0], max_shift[0] + 1) // self.patch_size[-2]),
                int(self._shift_rng.randint(-max_shift[1], max_shift[1] + 1) // self.patch_size[-1]),
            )
        return shift

    def create_motion_counterfactuals(self, x, masks, active_patches=None, shifts=None, frame=1, num_samples=None,
                                      fix_passive=True, reset_shifts=False):
        """`FlowGenerator.create_motion_counterfactuals` (segmentation.py:278-344): shift the active patches,
        keep the passive ones; all B*S prompts are built by one pair of HIP kernels instead of the reference's
        per-sample Python loop.  `shifts`: S (or B*S) pairs (dy, dx) in patch units, or None for random ones.
        Returns (x_shift [B*S,T,C,H,W], mask_shift [B*S,Nt]) in '(b s)' order, masks rectangularised."""
        _lib.require_gpu()
        if (getattr(self, "shifts", None) is None) or reset_shifts:
            self.reset_shifts()
        if len(masks.shape) == 2:
            assert num_samples is not None, "Choose how many samples to shift with arg num_samples"
            masks = masks.unsqueeze(-1).expand(-1, -1, num_samples)
        else:
            num_samples = masks.size(-1)
        if active_patches is None:
            active_patches = torch.ones_like(masks)
        elif len(active_patches.shape) == 2:
            active_patches = active_patches.unsqueeze(-1).expand(-1, -1, masks.size(-1))
        B, N, S = masks.shape
        assert active_patches.size(-1) in [1, S]
        if active_patches.size(-1) == 1:
            active_patches = active_patches.expand(-1, -1, S)
        if len(x.shape) == 4:
            x = x[:, None]
        T = x.shape[1] if not fix_passive else 2
        dev = x.device
        if not x.is_cuda:
            raise RuntimeError("create_motion_counterfactuals needs CUDA/HIP tensors (no CPU fallback)")
        self.inp_shape = (B, T) + tuple(x.shape[2:])
        xs = x[:, 0:1] if fix_passive else x
        xs = xs.to(torch.float32).contiguous()
        if fix_passive:  # the kernel reads frame 0 for every output frame: give it a T-frame view without copying
            xs = xs.expand(-1, T, -1, -1, -1).contiguous() if T != 1 else xs
        if shifts is None:
            shifts = [self._random_mask_shift() for _ in range(S)]
        elif hasattr(shifts, "shape"):
            arr = shifts.detach().cpu().numpy() if isinstance(shifts, torch.Tensor) else np.asarray(shifts)
            shifts = [tuple(int(v) for v in arr[..., s]) for s in range(arr.shape[-1])] if arr.shape[0] == 2 and arr.ndim == 2 and arr.shape[-1] in (S, 1, B * S) and arr.shape[0] != arr.shape[-1] else [tuple(int(v) for v in r) for r in arr]
        shifts = [tuple(int(v) for v in sh) for sh in shifts]
        if len(shifts) == 1:
            shifts = shifts * S
        assert len(shifts) in (S, B * S), (len(shifts), S)
        if len(shifts) == S:
            shifts = shifts * B
        sh = torch.tensor(shifts, dtype=torch.int32, device=dev).contiguous()
        act = active_patches.permute(0, 2, 1).reshape(B * S, N).to(device=dev, dtype=torch.bool).contiguous()
        msk = masks.permute(0, 2, 1).reshape(B * S, N).to(device=dev, dtype=torch.bool).contiguous()
        Cc, H, W = xs.shape[2:]
        x_shift = torch.empty((B * S, T, Cc, H, W), device=dev, dtype=torch.float32)
        mask_shift = torch.empty((B * S, N), device=dev, dtype=torch.bool)
        with torch.cuda.device(dev):
            _lib.check(
                _lib.get_lib().cwm_shift_prompts(
                    xs.data_ptr(), B, T, Cc, H, W, self.patch_size[-1], frame % T, S, int(bool(fix_passive)), act.data_ptr(),
                    msk.data_ptr(), sh.data_ptr(), x_shift.data_ptr(), mask_shift.data_ptr(), _lib.current_stream_handle(dev),
                )
            )
        for dy, dx in shifts:
            self.shift = [dy, dx]
            self.shifts.append(np.array(self.shift))
        mask_shift = self.mask_rectangularizer(mask_shift)
        return (x_shift, mask_shift)

    def predict_counterfactual_videos(self, x, active_patches, passive_patches=None, shifts=None, num_samples=8,
                                      sample_batch_size=8, fix_passive=True, frame=1, **kwargs):
        """The predictor half of `predict_counterfactual_videos_and_flows` (segmentation.py:346-430): build the
        motion counterfactuals and batch-predict them; returns y_mocos [B*S,T,C,H,W] (the RAFT flow step that
        follows in the reference is outside this package's scope)."""
        if len(x.shape) == 3:
            x = x.unsqueeze(0).unsqueeze(1).expand(-1, 2, -1, -1, -1)
            fix_passive = True
        elif len(x.shape) == 4:
            x = x.unsqueeze(1).expand(-1, 2, -1, -1, -1)
            fix_passive = True
        elif len(x.shape) == 5 and x.size(1) == 1:
            x = x.expand(-1, 2, -1, -1, -1)
        assert len(x.shape) == 5, x.shape
        x = x[:, 0:2]
        self.set_input(x)
        self.reset_shifts()
        if passive_patches is None:
            passive_patches = self.get_zeros_mask().unsqueeze(-1)
        elif len(passive_patches.shape) == 2:
            passive_patches = passive_patches.unsqueeze(-1)
        if len(active_patches.shape) == 2:
            active_patches = active_patches.unsqueeze(-1)
        S = max(active_patches.size(-1), passive_patches.size(-1))
        if (S == 1) and num_samples > 1:
            S = num_samples
        if shifts is None:
            shifts = [self._random_mask_shift() for _ in range(S)]
        num_samples = len(shifts) if not hasattr(shifts, "shape") else shifts.shape[-1]
        if (active_patches.size(-1) == 1) and (num_samples > 1):
            active_patches = active_patches.expand(-1, -1, num_samples)
        if (passive_patches.size(-1) == 1) and (num_samples > 1):
            passive_patches = passive_patches.expand(-1, -1, num_samples)
        assert active_patches.size(-1) == passive_patches.size(-1) == num_samples, (active_patches.shape, passive_patches.shape, num_samples)
        x_mocos, masks_mocos = self.create_motion_counterfactuals(
            x, masks=passive_patches, active_patches=active_patches, shifts=shifts, num_samples=num_samples,
            fix_passive=fix_passive, frame=frame, reset_shifts=False)
        return self.batch_predict_per_sample(
            x_mocos, masks=masks_mocos, frame=None, batch_size=(sample_batch_size or x_mocos.size(0)), sample_dim=0, **kwargs)

    # ---- statistics over the flow samples (segmentation.py:250-276, 479-547): device kernels, see flowstats.py ----
    def compute_flow_samples_magnitude(self, flows, normalize=True, dim=-4, eps=1e-2):
        from . import flowstats

        return flowstats.compute_flow_samples_magnitude(flows, normalize=normalize, dim=dim, eps=eps)

    def compute_mean_motion_map(self, flows, normalize_per_sample=False, normalize=True, dim=-4, eps=1e-2):
        from . import flowstats

        return flowstats.compute_mean_motion_map(flows, normalize_per_sample=normalize_per_sample, normalize=normalize, dim=dim, eps=eps)

    @staticmethod
    def compute_flow_corrs(flow_samples, *args, **kwargs):
        from . import flowstats

        return flowstats.compute_flow_corrs(flow_samples, *args, **kwargs)

    def forward(self, x, mask=None, frame=None, *args, **kwargs):
        self.set_input(x, mask)
        if mask is None:
            self.mask = self.generate_mask(x)
        return self.predict(self.x, self.mask, frame=frame, *args, **kwargs)
