"""`PredictorBasedGenerator`: the wrapper surface the notebooks / UI drive (reference: cwm/models/prediction.py:17-850),
re-built around the HIP predictor.

Same method names, argument meaning and error behaviour as the reference class, so code written against it runs on a
`counterfactualworldmodels_amd.vmae` / `.conjoined_vmae` predictor.  What differs is how the work is organised:

* a prediction is ONE library call (normalise + forward + un-embed fused, `cwm_forward`); the wrapper never touches
  pixels or tokens itself, and there is no CPU path
* batches are described once -- frames, masks and the number of masked tokens per row (`_RectBatch`) -- and then cut
  into row ranges that are fed to the library back to back with no host synchronisation in between
  (`_run_rect_batch`); the only host round trip of a multi-row call is the mask rectangulariser's single read-back
* per-sample tiling (`sample_tile*`, `predict_per_sample`) is expressed as index arithmetic on the batch description
  instead of materialised `expand().reshape()` copies wherever the library can take strided rows
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, Optional, Sequence

import numpy as np
import torch
from torch import nn

from . import _lib
from .config import IMAGENET_MEAN, IMAGENET_STD
from .masking import RectangularizeMasks, upsample_masks
from .conjoined_vmae import ConjoinedPaddedVisionTransformer
from .vmae import PretrainVisionTransformer


def imagenet_normalize(x, temporal_dim=1):
    """(x - mean_c) / std_c with the channel axis at 2 (temporal_dim=1) or 1 (temporal_dim=2); cwm/models/utils.py:15-21."""
    shape = [1, 1, 1, 1, 1]
    shape[2 if temporal_dim == 1 else 1] = 3
    mean = torch.tensor(IMAGENET_MEAN, dtype=x.dtype, device=x.device).view(shape)
    std = torch.tensor(IMAGENET_STD, dtype=x.dtype, device=x.device).view(shape)
    return (x - mean) / std


@dataclass
class _RectBatch:
    """R independent prediction rows: frames [R,T,C,H,W] in [0,1], masks [R,Nt] whose rows all mask `n_masked` tokens
    (None: not known on the host; the library then counts row 0 and verifies the others), and per-row keyword tensors
    for the predictor (the IMU stream of the conjoined model)."""

    x: torch.Tensor
    mask: torch.Tensor
    n_masked: Optional[int]
    row_kwargs: Dict[str, object] = field(default_factory=dict)
    weights_synced: bool = False          # `_as_rect` ran the predictor's sync_weights() already (before the mask read-back)
    out: Optional[torch.Tensor] = None    # output video allocated there too (fused path)

    @property
    def rows(self) -> int:
        return self.x.shape[0]


class PredictorBasedGenerator(nn.Module):
    """Factual / counterfactual predictions from a masked predictor (reference class: prediction.py:17)."""

    def __init__(self, predictor=None, predictor_load_path=None, keypoint_predictor=None, keypoint_predictor_load_path=None,
                 error_func=None, imagenet_normalize_inputs=False, temporal_dim=2, seed=0, mask_generator=None, raft_iters=None,
                 max_shift_fraction=0.15, **kwargs):
        super().__init__()
        self.set_predictor(predictor, predictor_load_path)
        self.error_func = error_func if error_func is not None else nn.MSELoss(reduction="none")
        self.imagenet_normalize_inputs = bool(imagenet_normalize_inputs)
        self.set_temporal_dim(temporal_dim)
        # RNG streams in the reference's construction order (prediction.py:43-45, perturbation.py:26-28): the wrapper seeds
        # numpy + the GLOBAL torch generator, then each perturbation module seeds the global generator with its own default 0
        self.seed = seed
        self.rng = np.random.RandomState(seed=seed)
        self.torch_rng = torch.manual_seed(seed)
        self._shift_rng = np.random.RandomState(seed=0)
        torch.manual_seed(0)
        self.mask_generator = mask_generator
        self.mask_rectangularizer = RectangularizeMasks("min")
        self.max_shift_fraction = max_shift_fraction
        self.keypoint_predictor = keypoint_predictor
        if keypoint_predictor is not None:
            self.load_predictor(keypoint_predictor_load_path, model=keypoint_predictor)
        self.shifts = None
        self.x = self.mask = self.timestamps = None

    # ---- predictor management (prediction.py:75-107) ---------------------------------------------------------------
    def set_predictor(self, net, predictor_load_path=None):
        if net is None:
            raise ValueError("There is no predictor set for this generator and no model to load to")
        self.predictor = net
        self.load_predictor(predictor_load_path)
        self.x = self.mask = self.inp_shape = None

    def load_predictor(self, load_path=None, model=None, map_location="cpu"):
        target = model if model is not None else getattr(self, "predictor", None)
        if target is None:
            raise ValueError("There is no predictor set for this generator and no model to load to")
        if load_path is None:
            inherited = getattr(self.predictor, "_predictor_load_path", None)
            if inherited is not None:
                self._predictor_load_path = inherited
            return
        state = torch.load(load_path, map_location=torch.device(map_location))
        state = state["model"] if "model" in state else state
        report = target.load_state_dict(state)
        if model is None:
            self._predictor_load_path = load_path
        print(report, load_path)

    # ---- geometry (prediction.py:131-207) --------------------------------------------------------------------------
    @property
    def patch_size(self):
        p = self.predictor
        return p.patch_size if hasattr(p, "patch_size") else p.encoder.patch_embed.proj.kernel_size

    @property
    def image_size(self):
        return self.predictor.image_size

    @property
    def sequence_length(self):
        p = self.predictor
        return p.sequence_length if hasattr(p, "sequence_length") else getattr(p, "num_frames", 2)

    @property
    def mask_shape(self):
        if hasattr(self.predictor, "mask_shape"):
            return self.predictor.mask_shape
        pt, ph, pw = self.patch_size
        h, w = self.inp_shape[-2:]
        return (self.sequence_length // pt, h // ph, w // pw)

    @property
    def inp_mask_shape(self):
        return (self.x.shape[0], int(np.prod(self.mask_shape)))

    @property
    def _is_padded(self):
        return hasattr(self.predictor, "main_stream") or hasattr(self.predictor, "max_padding_tokens")

    def set_temporal_dim(self, t_dim=1):
        if t_dim not in (1, 2):
            raise ValueError("temporal_dim must be 1 or 2")
        self.predictor.t_dim, self.predictor.c_dim = t_dim, 3 - t_dim

    t_dim = property(lambda self: self.predictor.t_dim)
    c_dim = property(lambda self: self.predictor.c_dim)

    def set_image_size(self, *args, **kwargs):
        setter = getattr(self.predictor, "set_image_size", None)
        if setter is not None:
            setter(*args, **kwargs)
        else:
            self.predictor.image_size = args[0]

    # ---- masks (prediction.py:109-129, 216-229, 357-384, 600-660) ---------------------------------------------------
    def generate_mask(self, x=None):
        assert self.mask_generator is not None
        x = self.x if x is None else x
        return self.mask_rectangularizer(self.mask_generator(x).view(x.size(0), -1).to(x.device))

    def set_new_mask(self, x=None):
        self.mask = self.generate_mask(self.x if x is None else x)

    def reset_padding_masks(self):
        reset = getattr(self.predictor, "_reset_padding_mask", None)
        if self._is_padded and reset is not None:
            reset()

    def get_zeros_mask(self, x=None, frame=-1):
        """All-visible mask, with `frame` (if not None) fully masked; [B,Nt] bool (prediction.py:216-226)."""
        x = self.x if x is None else x
        if self.inp_shape is None:
            self.inp_shape = x.shape
        t, h, w = self.mask_shape
        masked_frame = torch.arange(t, device=x.device) == (frame % t if frame is not None else -1)
        return masked_frame.repeat_interleave(h * w)[None].expand(x.shape[0], -1)

    def get_fully_visible_mask(self, x=None):
        x = self.x if x is None else x
        return torch.zeros(self.mask_shape, device=x.device, dtype=torch.bool)

    def get_mask_image(self, mask, upsample=False, invert=False, shape=None):
        grid = mask.view(-1, *(self.mask_shape if shape is None else shape))
        if upsample:
            grid = upsample_masks(grid.view(grid.size(0), -1, *self.mask_shape[-2:]).float(), self.inp_shape[-2:])
        return 1 - grid if invert else grid

    @staticmethod
    def make_visible_from_patch_idx_list(mask, patch_idx_list, stride=1, b=0, t=-1):
        """Un-mask the listed patches of a [B,T,h,w] mask grid in place.  Entries are (h,w), (t,h,w) or (b,t,h,w); spatial
        indices are image coordinates divided by `stride` (prediction.py:619-638)."""
        if len(patch_idx_list) == 0:
            return mask
        idx = torch.as_tensor(np.asarray(patch_idx_list), dtype=torch.long, device=mask.device).reshape(-1, np.asarray(patch_idx_list).shape[-1])
        k = idx.shape[1]
        assert k in (2, 3, 4), k
        hh = (idx[:, -2] // stride) % mask.size(-2)
        ww = (idx[:, -1] // stride) % mask.size(-1)
        bb = idx[:, 0] if k == 4 else torch.full_like(hh, b)
        tt = idx[:, -3] if k >= 3 else torch.full_like(hh, t)
        mask[bb, tt, hh, ww] = 0
        return mask

    def generate_mask_from_patch_idx_list(self, patch_idx_list, stride=None, b=0, frame=-1):
        """Mask with `frame` hidden except the listed patches (prediction.py:640-649)."""
        assert self.x is not None
        grid = self.get_mask_image(self.get_zeros_mask(frame=frame).clone())
        if stride is None:
            stride = self.inp_shape[-1] // grid.size(-1)
        grid = self.make_visible_from_patch_idx_list(grid, patch_idx_list, stride=stride, b=b, t=frame)
        return grid.view(grid.size(0), -1)

    def get_masked_pred_patches(self, preds, mask, invert=False, fill_value=None):
        """`preds` [B,T',C,H,W] with everything outside the masked patches zeroed (or replaced by `fill_value`: a tensor of
        the same shape or one value per channel); prediction.py:261-283."""
        t_out, (h, w) = preds.shape[1], self.mask_shape[-2:]
        keep = upsample_masks(mask.view(-1, t_out, h, w), preds.shape[-2:]).to(preds)
        if invert:
            keep = 1.0 - keep
        keep = keep.unsqueeze(2)
        out = preds * keep
        if fill_value is None:
            return out
        if not isinstance(fill_value, torch.Tensor):
            fill_value = torch.tensor(fill_value, dtype=torch.float32).to(out).view(1, 1, -1, 1, 1)
        else:
            assert list(fill_value.shape) == list(out.shape)
        return out + (1 - keep) * fill_value

    # ---- the path itself -------------------------------------------------------------------------------------------
    def _preprocess(self, x):
        """What the predictor sees: [B,C,T,H,W] (t_dim=2), imagenet-normalised if configured (prediction.py:304-312)."""
        if self.t_dim != 1:
            x = x.transpose(self.t_dim, self.c_dim)
        return imagenet_normalize(x, temporal_dim=self.t_dim) if self.imagenet_normalize_inputs else x

    def pred_patches_to_video(self, y, x, mask):
        """Video with the input at visible patches and `y` at masked ones (prediction.py:245-259): one HIP scatter."""
        _lib.require_gpu()
        B, T, Cc, H, W = x.shape
        dev = y.device
        x = x.to(device=dev, dtype=torch.float32).contiguous()
        y = y.to(torch.float32).contiguous()
        mask = mask.to(device=dev, dtype=torch.bool).reshape(B, -1).contiguous()
        n_vis = mask.shape[1] - int(mask[0].sum().item())
        out = torch.empty_like(x)
        with torch.cuda.device(dev):
            _lib.check(_lib.get_lib().cwm_unembed(y.data_ptr(), x.data_ptr(), mask.data_ptr(), B, T, Cc, H, W, self.patch_size[-1], n_vis,
                                                  out.data_ptr(), _lib.current_stream_handle(dev)))
        return out

    def _uses_fused_path(self, extra_args, row_kwargs) -> bool:
        return isinstance(self.predictor, PretrainVisionTransformer) and self.t_dim == 2 and not extra_args and not row_kwargs

    def _run_rect_batch(self, batch: _RectBatch, rows_per_call: Optional[int] = None, extra_args: Sequence = ()) -> torch.Tensor:
        """All rows of `batch` -> videos [R,T,C,H,W], `rows_per_call` rows per library call, every call writing its slice
        of ONE output tensor.  With `n_masked` known nothing here reads the device back, so the calls queue up behind each
        other on the stream."""
        R = batch.rows
        step = R if not rows_per_call else max(1, int(rows_per_call))
        Nt = batch.mask.shape[1]
        n_vis = None if batch.n_masked is None else Nt - batch.n_masked
        fused = self._uses_fused_path(extra_args, batch.row_kwargs)
        out = None
        if fused:
            out = batch.out if batch.out is not None else torch.empty(batch.x.shape, device=batch.x.device, dtype=torch.float32)
        pieces = []
        for r0 in range(0, R, step):
            r1 = min(r0 + step, R)
            xs, ms = batch.x[r0:r1], batch.mask[r0:r1]
            if fused:
                self.predictor.predict_video(xs, ms, normalize=self.imagenet_normalize_inputs, n_vis=n_vis, check=n_vis is None,
                                             out_video=out[r0:r1], weights_synced=batch.weights_synced)
                continue
            kw = {k: (v[r0:r1] if isinstance(v, torch.Tensor) else v) for k, v in batch.row_kwargs.items()}
            if n_vis is not None and isinstance(self.predictor, ConjoinedPaddedVisionTransformer):
                kw.update(n_vis=n_vis, check=False)  # counts known on the host: the library need not read them back
            y = self.predictor(self._preprocess(xs), ms, *extra_args, **kw)
            if self._is_padded:  # the padded predictors append max - min pad slots to the output (prediction.py:424-433)
                stream = getattr(self.predictor, "main_stream", self.predictor)
                y = y[:, : y.shape[1] - (stream.max_padding_tokens - stream.min_padding_tokens)]
            # NB the reference un-embeds with unnormalize(normalize(x)) for the conjoined model (prediction.py:436-446); the raw
            # input used here differs from that by at most one fp32 ulp at the visible pixels
            pieces.append(y if y.dim() == 5 else self.pred_patches_to_video(y, xs, mask=ms))
            if r1 < R:  # between pieces only: the state after the LAST piece is the caller's (`predict(reset_masks=False)` keeps it,
                self.reset_padding_masks()  # like the reference, prediction.py:451-452)
        return out if fused else (pieces[0] if len(pieces) == 1 else torch.cat(pieces, 0))

    def _as_rect(self, x, mask, row_kwargs=None, extra_args=(), for_video=False) -> _RectBatch:
        """Equalise the masked count over the rows the way the reference does for every multi-row call (prediction.py:421):
        in place, on torch's global RNG, and -- the one host sync -- remember the count."""
        # Host work that does not depend on the mask goes FIRST: the read-back below waits for everything queued on the stream (the previous
        # call's forward), and whatever the host still has to do after it delays this call's first kernel launch by as much (measured:
        # 230 us between the read-back and the first launch, 90 of them the walk over the parameters in sync_weights; tools/wrap_gap.py)
        synced, out = False, None
        if for_video and x.is_cuda and x.dim() == 5 and self._uses_fused_path(extra_args, row_kwargs or {}):
            self.predictor.sync_weights(x.device)
            synced = True
            out = torch.empty(x.shape, device=x.device, dtype=torch.float32)
        n_masked = None
        if x.size(0) > 1:
            mask = self.mask_rectangularizer(mask)
            n_masked = self.mask_rectangularizer.last_num_masked
        return _RectBatch(x, mask, n_masked, dict(row_kwargs or {}), synced, out)

    @staticmethod
    def _select_frame(video, frame):
        if frame is None:
            return video
        f = frame % video.size(1)
        return video[:, f : f + 1]

    def predict(self, x=None, mask=None, frame=-1, reset_masks=True, *args, **kwargs):
        """Predicted video [B,T,C,H,W] (or its frame `frame`) for movie x [B,T,C,H,W] in [0,1] and mask [B,Nt];
        prediction.py:406-454."""
        x = self.x if x is None else x
        mask = self.generate_mask(x) if mask is None else mask
        self.set_image_size(x.shape[-2:])
        self.inp_shape = x.shape
        video = self._run_rect_batch(self._as_rect(x, mask, kwargs, args, for_video=True), extra_args=args)
        if reset_masks:
            self.reset_padding_masks()
        return self._select_frame(video, frame)

    def predict_tokens(self, x, mask):
        """Raw predictor output [B,Nm,C*P*P] for wrapper-level input (the seam of prediction.py:419)."""
        return self.predictor(self._preprocess(x), self._as_rect(x, mask).mask)

    def predict_error(self, x=None, mask=None, target=None, frame=None, dim=-3):
        """error_func(prediction, target) summed over `dim` (prediction.py:331-343)."""
        x = self.x if x is None else x
        mask = self.generate_mask(x) if mask is None else mask
        pred = self.predict(x, mask, frame=frame)
        target = x if target is None else target
        if frame is not None:
            target = target[:, frame].unsqueeze(1)
        return self.error_func(pred, target).sum(dim, True)

    # ---- per-sample batching (prediction.py:456-540) -----------------------------------------------------------------
    def sample_tile(self, z, num_samples):
        """[B,...] -> [(B num_samples),...]: every row repeated num_samples times, '(b s)' order."""
        return z.repeat_interleave(num_samples, 0, output_size=z.shape[0] * num_samples)

    def sample_tile_all_tensors(self, num_samples, **kwargs):
        return {k: (self.sample_tile(v, num_samples) if torch.is_tensor(v) else v) for k, v in kwargs.items()}

    def predict_per_sample(self, x, masks, frame=-1, batch_size=None, split_samples=True, *args, **kwargs):
        """S masks per movie, masks [B,Nt,S]: all B*S predictions as one rectangular batch.  Returns [(B S),T',C,H,W]
        ('(b s)' rows) or, with split_samples, [B,T',C,H,W,S]."""
        assert masks.dim() == 3, masks.shape
        x = self.x if x is None else x
        B, S = x.size(0), masks.size(-1)
        rows = self.sample_tile(x, S)
        row_masks = masks.permute(0, 2, 1).reshape(B * S, -1)
        y = self.predict(rows, row_masks, frame, True, *args, **kwargs)
        return y.view(B, S, *y.shape[1:]).movedim(1, -1) if split_samples else y

    def batch_predict_per_sample(self, x, masks, frame=-1, batch_size=None, sample_dim=None, **kwargs):
        """Chunked form.  sample_dim != 0: masks [B,Nt,S], chunks of `batch_size // B` samples, result [B,T',C,H,W,S].
        sample_dim == 0: x [R,...] and masks [R,Nt] are already one row per sample, chunks of `batch_size // R`... rows as
        the reference computes it (prediction.py:504-507), result [R,T',C,H,W].  Tensor keyword arguments are per movie and
        are tiled over each chunk's samples."""
        S = masks.size(0) if sample_dim == 0 else masks.size(-1)
        per_call = S if batch_size is None else max(1, batch_size // x.size(0))
        outs = []
        for s0 in range(0, S, per_call):
            s1 = min(s0 + per_call, S)
            if sample_dim == 0:
                assert x.size(0) in (masks.size(0), masks.size(-1)), (x.shape, masks.shape)
                m = masks[s0:s1] if masks.dim() == 2 else masks[..., s0:s1].permute(0, 2, 1).reshape(-1, masks.shape[1])
                outs.append(self.predict(x[s0:s1], m, frame, True, **self.sample_tile_all_tensors(s1 - s0, **kwargs)))
            else:
                outs.append(self.predict_per_sample(x, masks[..., s0:s1], frame=frame, split_samples=True,
                                                    **self.sample_tile_all_tensors(s1 - s0, **kwargs)))
            self.reset_padding_masks()
        return torch.cat(outs, 0 if sample_dim == 0 else -1)

    # ---- inputs (prediction.py:703-739) -------------------------------------------------------------------------------
    def set_input(self, x, mask=None, make_mask=False, timestamps=None):
        if x.dim() == 4:
            x = x.unsqueeze(1)
        elif x.dim() != 5:
            raise AssertionError("Input must be a movie of shape [B,T,C,H,W] or a single frame of shape [B,C,H,W]")
        self.x, self.inp_shape = x, x.shape
        self.B, self.T, self.C = x.shape[:3]
        if mask is not None:
            self.mask = mask
        elif make_mask:
            assert self.mask_generator is not None, "You need to have a mask generator to set a new mask"
            self.set_new_mask(x)
        if timestamps is not None:
            self.timestamps = timestamps

    def get_static_input(self, x=None):
        x = self.x if x is None else x
        return x[:, :1].repeat(1, x.size(1), 1, 1, 1)

    def make_static_movie(self, x=None, T=None, frame=0):
        x = self.x if x is None else x
        T = getattr(self.predictor, "num_frames", 2) if T is None else T
        if x.dim() == 4:
            x = x.unsqueeze(1)
        assert x.dim() == 5, "x must be of shape [B,C,H,W] or [B,T,C,H,W], but is %s" % (tuple(x.shape),)
        f = frame % x.size(1)
        return x[:, f : f + 1].repeat(1, T, 1, 1, 1)

    # ---- single-prompt counterfactual (the UI's click handler, interface.py:273-299) ----------------------------------
    def reset_shifts(self):
        self.shifts = []

    def _record_shift(self, dy, dx):
        if getattr(self, "shifts", None) is None:
            self.shifts = []
        self.shift = [int(dy), int(dx)]
        self.shifts.append(np.array(self.shift))

    def _random_mask_shift(self):
        """A random shift in patch units, `ShiftPatchesAndMask.get_random_shift(is_mask_shift=True)` (perturbation.py:209-225):
        pixel shifts up to max_shift_fraction of the image, floored to whole patches, redrawn while dy + dx == 0 (sic)."""
        ph, pw = self.patch_size[-2:]
        lim = [int(self.max_shift_fraction * s) for s in self.inp_shape[-2:]]
        while True:
            dy = int(self._shift_rng.randint(-lim[0], lim[0] + 1) // ph)
            dx = int(self._shift_rng.randint(-lim[1], lim[1] + 1) // pw)
            if dy + dx != 0:
                return (dy, dx)

    def _shift_rows(self, x, passive, active, shifts, frame, fix_passive, samples_per_movie=1, frames=True, masks=True, num_frames=None):
        """Device-side prompt construction for R = B * samples_per_movie rows (library: cwm_shift_prompts; reference:
        PatchPerturbation.forward + ShiftPatchesAndMask.perturb, perturbation.py:99-113, 245-289).  x [B,T,C,H,W]; passive /
        active [R,Nt] bool (0 = patch stays visible / 0 = patch is moved); shifts int32 [R,2] (dy,dx) in patch units.
        Returns (x_shift [R,T,C,H,W], mask_shift [R,Nt]) before rectangularisation; `frames=False` / `masks=False` skips that output (None): the
        sharded loop needs the masks of all prompts on rank 0 but frames only for the rows a rank predicts (dist.py)."""
        assert frames or masks
        _lib.require_gpu()
        if not x.is_cuda:
            raise RuntimeError("counterfactual prompts are built on the GPU (no CPU fallback); got a %s tensor" % x.device)
        dev = x.device
        B, T, Cc, H, W = x.shape
        if num_frames is not None and num_frames != T:
            # a static movie given as its first frame alone (`make_static_movie`, prediction.py:731-739, is frame 0 repeated): with fix_passive=True the kernel
            # reads frame 0 for every output frame, so the T-fold copy need not exist
            if not (T == 1 and fix_passive is True):
                raise RuntimeError("num_frames=%d needs a one-frame movie and fix_passive=True" % num_frames)
            T = int(num_frames)
        R, N = passive.shape
        x = x.to(torch.float32).contiguous() if frames else None  # (masks only: the frames are not read; no 1.2-MB materialisation of an expanded movie on rank 0's path)
        passive = passive.to(device=dev, dtype=torch.bool).contiguous()
        active = active.to(device=dev, dtype=torch.bool).contiguous()
        shifts = shifts.to(device=dev, dtype=torch.int32).contiguous()
        x_shift = torch.empty((R, T, Cc, H, W), device=dev, dtype=torch.float32) if frames else None
        mask_shift = torch.empty((R, N), device=dev, dtype=torch.bool) if masks else None
        with torch.cuda.device(dev):
            _lib.check(_lib.get_lib().cwm_shift_prompts(
                _lib.ptr(x), B, T, Cc, H, W, self.patch_size[-1], frame % T, samples_per_movie,
                2 if fix_passive == "make_static" else int(bool(fix_passive)),
                active.data_ptr(), passive.data_ptr(), shifts.data_ptr(), _lib.ptr(x_shift), _lib.ptr(mask_shift),
                _lib.current_stream_handle(dev)))
        return x_shift, mask_shift

    def _shift(self, x, mask, active_patches=None, shift=None, frame=1, fix_passive=False):
        """One shift applied to every movie of x (prediction.py:760-779): returns (x_shift, rectangularised mask_shift) and
        appends the shift (patch units) to `self.shifts`.  fix_passive="make_static" applies `MakeStatic` (perturbation.py:120-145)
        to the patches `mask` leaves visible first, in the same kernel."""
        if active_patches is None:
            active_patches = torch.ones_like(mask)
        self.inp_shape = x.shape
        dy, dx = self._random_mask_shift() if shift is None else (int(shift[0]), int(shift[1]))
        table = torch.tensor([[dy, dx]], dtype=torch.int32).expand(x.shape[0], 2)
        x_shift, mask_shift = self._shift_rows(x, mask.reshape(x.shape[0], -1), active_patches.reshape(x.shape[0], -1), table, frame,
                                               fix_passive=fix_passive)
        self._record_shift(dy, dx)
        return x_shift, self.mask_rectangularizer(mask_shift)

    def get_counterfactual_prediction(self, x, mask=None, active_patches=None, shift=None, fix_passive=False, **kwargs):
        """Move the active patches of frame 1 by `shift` patches, keep the passive ones (`mask`), predict the whole movie
        (prediction.py:781-812).  x may be an image [C,H,W] / [B,C,H,W] (made into a static 2-frame movie) or a movie."""
        if x.dim() == 3:
            x = x[None, None]
        elif x.dim() == 4:
            x = x[:, None]
        if x.size(1) == 1:
            x = self.make_static_movie(x, T=2)
        self.inp_shape = x.shape
        mask = self.get_zeros_mask(x) if mask is None else mask
        active_patches = self.get_zeros_mask(x) if active_patches is None else active_patches
        # fix_passive: `x, _ = self.make_static(x, mask)` (prediction.py:802-803) folded into the prompt kernel
        x_p, mask_p = self._shift(x, mask, active_patches=active_patches, shift=shift, frame=1,
                                  fix_passive="make_static" if fix_passive else False)
        return self.predict(x_p, mask_p, frame=None, **kwargs)

    def make_static(self, x, mask):
        """`MakeStatic` (perturbation.py:120-145; the generator's `make_static` attribute, prediction.py:51-55): the patches `mask`
        leaves visible in frames t > 0 take the pixels of the same patch in frame 0.  Returns (x_static, mask) like the reference."""
        B = x.shape[0]
        m = mask.reshape(B, -1)
        zero = torch.zeros((B, 2), dtype=torch.int32)
        x_static, _ = self._shift_rows(x, m, torch.ones_like(m), zero, 1, fix_passive="make_static")
        return x_static, mask

    def forward(self, x, mask=None, frame=None, *args, **kwargs):
        self.set_input(x, mask)
        if mask is None:
            self.mask = self.generate_mask(x)
        return self.predict(self.x, self.mask, frame=frame, *args, **kwargs)
