"""`FlowGenerator`: batched motion counterfactuals over the HIP predictor (reference: cwm/models/segmentation.py:62-547,
760-963 -- `create_motion_counterfactuals`, `predict_counterfactual_videos_and_flows`, the IMU-conditioned override and the
flow-sample statistics).

The reference builds the B*S prompts in a per-sample Python loop and pushes them through `batch_predict_per_sample`, which
re-rectangularises and synchronises per chunk.  Here the whole prompt set is ONE batch description: every prompt's frames and
mask come out of one pair of HIP kernels (`cwm_shift_prompts`), the masks are rectangularised once (the reference's single
`mask_rectangularizer` call at segmentation.py:342: same global-RNG consumption, one host read-back), and the predictor then
runs over row ranges of that batch with the masked count already known -- no further host round trip until the result is
used.  The optical-flow model that follows in the reference (RAFT) is outside this package; any module with the reference's
`flow_model(video, backward=...)` call signature can be plugged in.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .prediction import PredictorBasedGenerator, _RectBatch


def _shift_list(shifts, num: int) -> List[Tuple[int, int]]:
    """Normalise a shifts argument to `num` (dy, dx) pairs in patch units.  Accepted, as by the reference's
    `_preprocess_shifts_sequence` (perturbation.py:181-207): one pair, a sequence of pairs (length 1 = the same shift for
    every sample), or an array / tensor of shape [2, S] (S = 1 broadcasts)."""
    if hasattr(shifts, "shape"):
        arr = shifts.detach().cpu().numpy() if torch.is_tensor(shifts) else np.asarray(shifts)
        assert arr.ndim == 2 and arr.shape[0] == 2, arr.shape
        pairs = [(int(arr[0, s]), int(arr[1, s])) for s in range(arr.shape[1])]
    else:
        seq = list(shifts)
        if len(seq) == 2 and not isinstance(seq[0], (list, tuple, np.ndarray)):
            seq = [seq]
        assert all(len(p) == 2 for p in seq), seq
        pairs = [(int(p[0]), int(p[1])) for p in seq]
    if len(pairs) == 1:
        pairs = pairs * num
    assert len(pairs) == num, (len(pairs), num)
    return pairs


class FlowGenerator(PredictorBasedGenerator):
    """Counterfactual videos (and, with a flow model, flows) from moving patches; reference class segmentation.py:62."""

    def __init__(self, *args, flow_model=None, flow_model_load_path=None, raft_iters=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.flow_model = flow_model
        if flow_model is not None and flow_model_load_path is not None:
            self.load_predictor(flow_model_load_path, model=flow_model)
        self.raft_iters = raft_iters

    # ---- flow model hook (segmentation.py:141-153) -------------------------------------------------------------------
    def set_raft_iters(self, iters=None):
        self.raft_iters = iters
        if hasattr(self.flow_model, "set_iters"):
            self.flow_model.set_iters(iters)

    def predict_flow(self, vid, backward=False, iters=None, **kwargs):
        if self.flow_model is None:
            raise RuntimeError("this FlowGenerator has no flow_model (the reference plugs RAFT in here; any module called as "
                               "flow_model(video[B,T,C,H,W], backward=...) -> [B,T-1,2,H,W] works)")
        if iters is not None:
            self.set_raft_iters(iters)
        return self.flow_model(vid, backward=backward, **kwargs).to(vid)

    @staticmethod
    def batch_to_samples(flows, t=0, B=1):
        """[(b s),T,C,H,W] -> frame t as [b,C,H,W,s] (segmentation.py:129-132)."""
        assert flows.dim() == 5, flows.shape
        f = flows[:, t]
        return f.reshape(B, -1, *f.shape[1:]).movedim(1, -1)

    def _batch_to_samples(self, flows, t=0):
        assert self.x is not None
        if flows.dim() != 5:
            flows, t = flows.unsqueeze(1), 0
        return self.batch_to_samples(flows, t=t, B=self.x.size(0))

    # ---- prompt construction (segmentation.py:278-344) -----------------------------------------------------------------
    def _build_prompts(self, x, passive, active, pairs, frame, fix_passive) -> _RectBatch:
        """passive / active: [B,Nt,S] bool; pairs: S shifts (shared by all movies) or B*S shifts in '(b s)' order, None = S random
        ones.  Returns the '(b s)'-ordered rectangular batch of all B*S prompts."""
        B, N, S = passive.shape
        if x.dim() == 4:
            x = x[:, None]
        src = x[:, :1] if fix_passive else x
        T = 2 if fix_passive else x.shape[1]
        if fix_passive and T != 1:
            src = src.expand(-1, T, -1, -1, -1)  # every output frame of a static movie reads frame 0
        self.inp_shape = (B, T) + tuple(x.shape[2:])
        if pairs is None:
            pairs = [self._random_mask_shift() for _ in range(S)]
        assert len(pairs) in (S, B * S), (len(pairs), S, B)
        rows = list(pairs) * B if len(pairs) == S else list(pairs)
        table = torch.tensor(rows, dtype=torch.int32).reshape(B * S, 2)
        x_shift, mask_shift = self._shift_rows(src, passive.permute(0, 2, 1).reshape(B * S, N), active.permute(0, 2, 1).reshape(B * S, N),
                                               table, frame, fix_passive, samples_per_movie=S)
        for dy, dx in rows:
            self._record_shift(dy, dx)
        mask_shift = self.mask_rectangularizer(mask_shift)
        return _RectBatch(x_shift, mask_shift, self.mask_rectangularizer.last_num_masked)

    def create_motion_counterfactuals(self, x, masks, active_patches=None, shifts=None, frame=1, num_samples=None, fix_passive=True,
                                      reset_shifts=False):
        """Shift the active patches of `frame`, keep the passive ones (`masks`, 0 = visible) in place.  masks / active_patches:
        [B,Nt] (with num_samples) or [B,Nt,S]; shifts: see `_shift_list` (S pairs are used for every movie; B*S pairs address
        the '(b s)' rows one by one), None = random.  Returns (x_shift [B*S,T,C,H,W], mask_shift [B*S,Nt]), masks
        rectangularised once for all rows."""
        if getattr(self, "shifts", None) is None or reset_shifts:
            self.reset_shifts()
        if masks.dim() == 2:
            assert num_samples is not None, "Choose how many samples to shift with arg num_samples"
            masks = masks.unsqueeze(-1).expand(-1, -1, num_samples)
        S = masks.size(-1)
        if active_patches is None:
            active_patches = torch.ones_like(masks)
        elif active_patches.dim() == 2:
            active_patches = active_patches.unsqueeze(-1)
        assert active_patches.size(-1) in (1, S)
        active_patches = active_patches.expand(-1, -1, S)
        B = masks.size(0)
        if shifts is None:
            pairs = None
        elif not hasattr(shifts, "shape") and B > 1 and len(shifts) == B * S and isinstance(shifts[0], (list, tuple, np.ndarray)):
            pairs = _shift_list(shifts, B * S)
        else:
            pairs = _shift_list(shifts, S)
        batch = self._build_prompts(x, masks, active_patches, pairs, frame, fix_passive)
        return batch.x, batch.mask

    # ---- the batch driver (segmentation.py:346-432) ------------------------------------------------------------------
    def _conditioning_kwargs(self, x, kwargs):
        """Per-movie keyword tensors for the predictor (none for the plain VMAE; the IMU subclass adds its context stream)."""
        return kwargs

    @staticmethod
    def _two_frame_movie(x, fix_passive):
        if x.dim() == 3:
            return x[None, None].expand(-1, 2, -1, -1, -1), True
        if x.dim() == 4:
            return x[:, None].expand(-1, 2, -1, -1, -1), True
        assert x.dim() == 5, x.shape
        if x.size(1) == 1:
            x = x.expand(-1, 2, -1, -1, -1)
        return x[:, :2], fix_passive

    def _counterfactual_batch(self, x, active_patches, passive_patches, shifts, num_samples, fix_passive, frame, row_kwargs) -> _RectBatch:
        x, fix_passive = self._two_frame_movie(x, fix_passive)
        self.set_input(x)
        self.reset_shifts()
        passive = self.get_zeros_mask() if passive_patches is None else passive_patches
        passive = passive.unsqueeze(-1) if passive.dim() == 2 else passive
        active = active_patches.unsqueeze(-1) if active_patches.dim() == 2 else active_patches
        S = max(active.size(-1), passive.size(-1))
        if S == 1 and num_samples > 1:
            S = num_samples
        if shifts is None:
            pairs = [self._random_mask_shift() for _ in range(S)]
        else:
            n_given = shifts.shape[-1] if hasattr(shifts, "shape") else (1 if not isinstance(shifts[0], (list, tuple, np.ndarray)) else len(shifts))
            pairs = _shift_list(shifts, n_given)
        S = len(pairs)
        active = active.expand(-1, -1, S) if active.size(-1) == 1 else active
        passive = passive.expand(-1, -1, S) if passive.size(-1) == 1 else passive
        assert active.size(-1) == passive.size(-1) == S, (active.shape, passive.shape, S)
        batch = self._build_prompts(x, passive, active, pairs, frame, fix_passive)
        B = x.shape[0]
        for k, v in row_kwargs.items():  # per-movie tensors (the IMU stream) follow their movie's S prompts
            if torch.is_tensor(v) and v.shape[0] == B and B != batch.rows:
                v = self.sample_tile(v, S)
            batch.row_kwargs[k] = v
        return batch

    def predict_counterfactual_videos(self, x, active_patches, passive_patches=None, shifts=None, num_samples=8, sample_batch_size=8,
                                      fix_passive=True, frame=1, **kwargs):
        """y_mocos [B*S,T,C,H,W]: the predictor half of `predict_counterfactual_videos_and_flows`.  `sample_batch_size` rows go
        into one predictor call (None: all of them); the result does not depend on it."""
        batch = self._counterfactual_batch(x, active_patches, passive_patches, shifts, num_samples, fix_passive, frame,
                                           self._conditioning_kwargs(x, kwargs))
        y = self._run_rect_batch(batch, rows_per_call=sample_batch_size)
        self.reset_padding_masks()
        return y

    def predict_counterfactual_videos_and_flows(self, x, active_patches, passive_patches=None, shifts=None, num_samples=8,
                                                sample_batch_size=8, fix_passive=True, max_shift_fraction=None, frame=1, raft_iters=None,
                                                backward=False, **kwargs):
        """(y_mocos [B*S,T,C,H,W], flow_mocos [B*S,T-1,2,H,W]) for S motion counterfactuals per movie: active patches moved by
        the shifts, passive patches revealed in place (segmentation.py:346-432)."""
        if max_shift_fraction is not None and shifts is None:
            self.max_shift_fraction = max_shift_fraction
        y_mocos = self.predict_counterfactual_videos(x, active_patches, passive_patches, shifts, num_samples, sample_batch_size, fix_passive,
                                                     frame, **kwargs)
        return y_mocos, self.predict_flow(y_mocos, backward=backward, iters=raft_iters)

    # ---- statistics over the flow samples (segmentation.py:250-276, 479-547): device kernels, see flowstats.py ----------
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


class ImuConditionedFlowGenerator(FlowGenerator):
    """The IMU-conditioned variant (segmentation.py:760-963) for a conjoined RGB+IMU predictor.  The reference derives the
    head motion from a second (flow -> IMU) model whose preprocessing needs a RAFT checkpoint; here the head motion is an input
    (`head_motion` [B,6,400] in the predictor's layout), and it is forwarded exactly as the reference forwards it: as
    `x_context`, with an all-visible (or, with mask_head_motion, all-masked) `mask_context`, tiled over every movie's prompts."""

    @property
    def num_head_tokens(self):
        return self.predictor.context_stream.encoder.num_tokens

    @property
    def head_tubelet_size(self):
        return self.predictor.context_stream.patch_size[0]

    @property
    def head_motion_channels(self):
        return getattr(self.predictor.get_context_input, "num_channels", 6)

    def get_zeros_imu(self, x=None):
        x = self.x if x is None else x
        return torch.zeros((x.shape[0], self.head_motion_channels, self.head_tubelet_size * self.num_head_tokens), device=x.device, dtype=x.dtype)

    def _conditioning_kwargs(self, x, kwargs):
        """segmentation.py:931-963: `head_motion` -> x_context, an all-visible (mask_head_motion: all-masked) mask_context."""
        kw = dict(kwargs)
        head_motion = kw.pop("head_motion", None)
        mask_head_motion = kw.pop("mask_head_motion", False)
        kw.pop("static_head_motion", None)
        kw.pop("timestamps", None)
        if head_motion is None:
            raise RuntimeError("pass head_motion [B,%d,%d]: estimating it from the video needs the reference's flow->IMU model (RAFT), "
                               "which is outside this package" % (self.head_motion_channels, self.head_tubelet_size * self.num_head_tokens))
        h_mask = torch.zeros(head_motion.shape[0], self.num_head_tokens, dtype=torch.bool, device=head_motion.device)
        if mask_head_motion:
            h_mask = ~h_mask
        kw.update(x_context=head_motion, mask_context=h_mask, n_vis_context=0 if mask_head_motion else self.num_head_tokens)
        return kw
