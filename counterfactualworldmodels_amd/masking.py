"""Host-side mask helpers on the predictor path (bit-exact index work, tiny tensors).

`RectangularizeMasks` mirrors `cwm/models/masking.py:90-132`: it is the only cross-row operation
before the predictor call (`prediction.py:421`), it draws from torch's *global* RNG and mutates its
input in place, and both behaviours are kept so seeded runs reproduce the reference's masks.
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch


class RectangularizeMasks:
    """Make sure all masks in a batch have the same number of 1s and 0s (masking.py:90-132)."""

    def __init__(self, truncation_mode="min"):
        assert truncation_mode in ["min", "max", "mean", "full", "none", None], truncation_mode
        self._mode = truncation_mode
        self.last_num_masked = None
        self.spin_wait = os.environ.get("CWM_SPIN_WAIT", "1") != "0"  # poll for the device -> host copy of the row counts instead of blocking on it (see _counts_to_host)
        self._event = None

    def set_mode(self, mode):
        self._mode = mode

    def _counts_to_host(self, masks: torch.Tensor):
        """Masked count of every row of the device masks [B, Nt] (library: cwm_mask_row_counts) -> a persistent pinned staging buffer: 4 bytes per row come
        back instead of the masks themselves."""
        from . import _lib

        B, Nt = masks.shape
        dev = masks.device
        if getattr(self, "_stage", None) is None or self._stage.numel() < B:
            self._stage = torch.empty(max(B, 1024), dtype=torch.int32, pin_memory=True)
        if getattr(self, "_counts_dev", None) is None or self._counts_dev.numel() < B or self._counts_dev.device != dev:
            self._counts_dev = torch.empty(max(B, 1024), dtype=torch.int32, device=dev)
        stream = torch.cuda.current_stream(dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.get_lib().cwm_mask_row_counts(masks.data_ptr(), B, Nt, self._counts_dev.data_ptr(), stream.cuda_stream))
        host = self._stage[:B]
        if self.spin_wait:
            # The copy waits for everything queued on the stream (the previous call's forward: milliseconds).  A blocking wait parks the thread,
            # and the host code that follows it -- all of it between this read-back and the call's first kernel launch -- then runs on a core
            # that has just left a sleep state: measured 190 us for what takes 40 us on a busy core (tools/wrap_host_profile.py), an idle GPU
            # for as long.  Polling the event keeps the core awake; `spin_wait = False` restores the blocking copy.
            host.copy_(self._counts_dev[:B], non_blocking=True)
            if self._event is None:
                self._event = torch.cuda.Event()
            self._event.record(stream)
            while not self._event.query():
                time.sleep(0)  # (gives the GIL away between two polls -- other Python threads of the process keep running -- without parking the core)
        else:
            host.copy_(self._counts_dev[:B])
        return host.tolist()

    def _target(self, counts):
        if self._mode == "min":
            return min(counts)
        if self._mode == "max":
            return max(counts)
        # "mean": the reference takes torch.mean of the float32 counts and truncates (masking.py:108-111)
        return int(torch.tensor(counts, dtype=torch.float32).mean().long())

    def _call_device(self, masks: torch.Tensor) -> torch.Tensor:
        """Device masks [B, Nt] (contiguous, in place).  The reference edits a changed row at `torch.where(row)[0][torch.randperm(n)[:surplus]]`: the draw needs
        only n = the row's count, so the host reads back the COUNTS (4 B per row, one synchronisation), draws the same `torch.randperm(n)` from the global
        CPU generator for the same rows in the same order, and sends the picks -- "the k-th masked / visible token of row r" -- to `cwm_mask_flip_picks`, which
        applies them to the rows on the device.  (Until round 5 the whole mask tensor came to the host and went back: 400 KB each way for 256 prompts.)"""
        from . import _lib

        B, Nt = masks.shape
        counts = self._counts_to_host(masks)
        target = self._target(counts)
        self.last_num_masked = int(target)
        rows, offsets, to_value, picks = [], [0], [], []
        for b, n_b in enumerate(counts):
            surplus = n_b - target
            if surplus == 0:
                continue
            # one randperm per changed row, in row order: over the masked tokens to un-mask `surplus` of them, or over the visible ones to mask `-surplus`
            k = torch.randperm(n_b)[:surplus] if surplus > 0 else torch.randperm(Nt - n_b)[:-surplus]
            rows.append(b)
            to_value.append(0 if surplus > 0 else 1)
            picks.append(k)
            offsets.append(offsets[-1] + k.numel())
        if rows:
            head = [len(rows)] + rows + offsets + to_value
            n = len(head) + offsets[-1]
            if getattr(self, "_table_host", None) is None or self._table_host.numel() < n:
                self._table_host = torch.empty(max(n, 4096), dtype=torch.int32, pin_memory=True)
            if getattr(self, "_table_event", None) is not None:
                self._table_event.synchronize()  # the previous call's copy out of the pinned buffer has run (it may have been queued on another stream)
            # written straight into the pinned buffer: the header through numpy, the picks with ONE concatenation (32 changed rows used to cost 32 casts + a 33-way cat)
            th = self._table_host.numpy()
            th[: len(head)] = head
            self._table_host[len(head) : n].copy_(torch.cat(picks) if len(picks) > 1 else picks[0])
            table_dev = self._table_host[:n].to(masks.device, non_blocking=True)
            self._table_event = torch.cuda.Event()
            self._table_event.record(torch.cuda.current_stream(masks.device))
            with torch.cuda.device(masks.device):
                _lib.check(_lib.get_lib().cwm_mask_flip_picks(masks.data_ptr(), B, Nt, table_dev.data_ptr(), len(rows), _lib.current_stream_handle(masks.device)))
        return masks

    def __call__(self, masks: torch.Tensor) -> torch.Tensor:
        self.last_num_masked = None
        if self._mode in ["none", None]:
            return masks
        assert isinstance(masks, torch.Tensor), type(masks)
        if self._mode == "full":
            return torch.ones_like(masks)
        shape = masks.shape
        if masks.is_cuda and masks.dtype == torch.bool and masks.shape[0] > 0 and masks[0].numel() <= 16384:
            flat = masks.flatten(1)
            if flat.is_contiguous() and flat.data_ptr() == masks.data_ptr():
                self._call_device(flat)
            else:  # a strided view: the kernels work on a packed copy, the result goes back in place like the reference's row assignments
                work = flat.contiguous()
                self._call_device(work)
                masks.copy_(work.view(shape))
            return masks
        masks = masks.flatten(1)
        work = masks
        if masks.is_cuda:  # (non-bool or very long rows: through the host)
            work = masks.cpu()
        rows = work.numpy()  # (shares memory with `work`; the row edits are index arithmetic on <= Nt bytes: numpy, no thread-pool spin-up)
        counts = rows.view(np.uint8).sum(axis=1, dtype=np.int32).tolist() if rows.dtype == np.bool_ else (rows != 0).sum(axis=1).tolist()
        target = self._target(counts)
        # every row now gets exactly `target` masked tokens: callers that need the count (the predictor's n_vis) read it here
        # instead of paying a second device round trip
        self.last_num_masked = int(target)
        changed = False
        for b, n_b in enumerate(counts):
            surplus = n_b - target
            if surplus > 0:    # un-mask `surplus` random masked positions (one randperm per changed row, in row order)
                where = np.flatnonzero(rows[b])
                rows[b, where[torch.randperm(where.size)[:surplus].numpy()]] = False
                changed = True
            elif surplus < 0:  # mask random visible positions
                where = np.flatnonzero(~rows[b])
                rows[b, where[torch.randperm(where.size)[:-surplus].numpy()]] = True
                changed = True
        if changed and work is not masks:
            masks.copy_(work)  # in place, like the reference
        return masks if list(masks.shape) == list(shape) else masks.view(*shape)


def upsample_masks(masks: torch.Tensor, size) -> torch.Tensor:
    """Nearest-neighbour mask upsampling for integer ratios (masking.py:10-30)."""
    shape = masks.shape
    h, w = shape[-2:]
    H, W = size
    if (H == h) and (W == w):
        return masks
    if (H < h) and (W < w):
        s = (h // H, w // W)
        return masks[..., :: s[0], :: s[1]]
    if (H % h) or (W % w):
        raise NotImplementedError("non-integer mask upsampling is outside the predictor path")
    masks = masks.unsqueeze(-2).unsqueeze(-1)
    masks = masks.repeat(*([1] * (len(shape) - 2)), 1, H // h, 1, W // w)
    return masks.view(*shape[:-2], H, W)


# ---- mask generators that PRODUCE the prompts' masks (masking.py:267-401, 478-545) ---------------------------------------
class MaskingGenerator(torch.nn.Module):
    """Uniformly random masks, one per frame: `num_masks_per_frame = int(mask_ratio * cells)` of the (clumped) grid cells are
    masked.  Bit-compatible with the reference generator under the same seeds: one `torch.randperm(cells)` on the GLOBAL torch
    generator per frame and batch row, and -- with clumping -- two draws of the instance's numpy stream per frame for the
    placement of the leftover rows / columns (which are always masked)."""

    def __init__(self, input_size, mask_ratio, seed=0, visible_frames=0, clumping_factor=1, randomize_num_visible=False,
                 create_on_cpu=True, always_batch=False):
        super().__init__()
        dims = (input_size,) if isinstance(input_size, int) else tuple(input_size)
        self.frames = dims[0] if len(dims) == 3 else None
        self.height, self.width = (dims[-2], dims[-1]) if len(dims) >= 2 else (dims[0], dims[0])
        self.clumping_factor = clumping_factor
        ch, cw = self.c
        self.pad_h, self.pad_w = self.height % ch, self.width % cw
        self.num_patches_per_frame = (self.height // ch) * (self.width // cw)
        self.mask_ratio = mask_ratio
        self.visible_frames = visible_frames
        self.always_batch = always_batch
        self.create_on_cpu = create_on_cpu
        self.randomize_num_visible = randomize_num_visible
        self.seed = seed
        self.rng = np.random.RandomState(seed=seed)
        torch.manual_seed(seed)

    @property
    def c(self):
        f = self.clumping_factor
        return (f, f) if isinstance(f, int) else tuple(f[:2])

    @property
    def mask_ratio(self):
        return self._mask_ratio

    @mask_ratio.setter
    def mask_ratio(self, ratio):
        self._mask_ratio = ratio
        self._num_masks_per_frame = int(ratio * self.num_patches_per_frame)

    @property
    def num_masks_per_frame(self):
        return self._num_masks_per_frame

    @num_masks_per_frame.setter
    def num_masks_per_frame(self, count):
        self._num_masks_per_frame = count
        self._mask_ratio = count / self.num_patches_per_frame

    @property
    def num_visible(self):
        return self.num_patches_per_frame - self.num_masks_per_frame

    @num_visible.setter
    def num_visible(self, count):
        self.num_masks_per_frame = self.num_patches_per_frame - count

    def __repr__(self):
        return "%s(cells/frame=%d, masked=%d, visible=%d, mask_ratio=%.3f, randomize_num_visible=%s)" % (
            type(self).__name__, self.num_patches_per_frame, self.num_masks_per_frame, self.num_visible, self.mask_ratio,
            self.randomize_num_visible)

    def sample_mask_per_frame(self, *args, **kwargs):
        n_masked = self.num_masks_per_frame
        if self.randomize_num_visible:
            n_masked = self.rng.randint(low=n_masked, high=self.num_patches_per_frame + 1)
        # cell i is visible iff a random permutation sends it among the first (cells - n_masked) slots
        cells = torch.randperm(self.num_patches_per_frame) >= self.num_patches_per_frame - n_masked
        ch, cw = self.c
        if max(ch, cw) == 1:
            return cells
        grid = cells.view(self.height // ch, self.width // cw).repeat_interleave(ch, 0).repeat_interleave(cw, 1)
        bottom = int(self.rng.choice(range(self.pad_h + 1)))
        right = int(self.rng.choice(range(self.pad_w + 1)))
        full = torch.ones(self.height, self.width, dtype=torch.bool)
        top, left = self.pad_h - bottom, self.pad_w - right
        full[top : top + grid.shape[0], left : left + grid.shape[1]] = grid
        return full

    def _one_row(self, num_frames):
        return torch.cat([self.sample_mask_per_frame() for _ in range(num_frames)], 0).flatten()

    def forward(self, x=None, num_frames=None):
        num_frames = (num_frames or self.frames) or 1
        if isinstance(x, torch.Tensor):
            rows = x.size(0)
            masks = torch.stack([self._one_row(num_frames) for _ in range(rows)], 0)
            if not self.create_on_cpu:
                masks = masks.to(x.device)
            if rows == 1 and not self.always_batch:
                masks = masks[0]
        else:
            rows = 1
            masks = self._one_row(num_frames)
            if self.always_batch:
                masks = masks[None]
        if self.visible_frames > 0:
            lead = torch.zeros(masks.shape[:-1] + (self.visible_frames * self.height * self.width,), dtype=torch.bool, device=masks.device)
            masks = torch.cat([lead, masks], -1)
        return masks


class RotatedTableUniformMaskingGenerator(MaskingGenerator):
    """The demos' prompt mask: the first `visible_frames` frames (default: all but the last) fully visible (or masked at
    `context_mask_ratio`), the remaining frame(s) masked at `mask_ratio` (masking.py:478-545)."""

    def __init__(self, input_size, mask_ratio, visible_frames=None, context_mask_ratio=None, seed=0, clumping_factor=1,
                 always_batch=True, randomize_num_visible=False, full_mask_prob=0):
        assert len(input_size) == 3, input_size
        if visible_frames is None:
            visible_frames = input_size[0] - 1
        super().__init__(input_size=(input_size[0] - visible_frames,) + tuple(input_size[1:]), mask_ratio=mask_ratio,
                         visible_frames=visible_frames, seed=seed, clumping_factor=clumping_factor, always_batch=always_batch,
                         randomize_num_visible=randomize_num_visible)
        self.full_mask_prob = full_mask_prob
        self.context_mask_ratio = context_mask_ratio or 0
        self.vis_frame_sampler = None
        if context_mask_ratio is not None:
            self.vis_frame_sampler = MaskingGenerator(input_size=(1, self.height, self.width), mask_ratio=context_mask_ratio, visible_frames=0,
                                                      clumping_factor=1, create_on_cpu=self.create_on_cpu, always_batch=self.always_batch)

    def __repr__(self):
        return super().__repr__() + " visible_frames=%d context_mask_ratio=%s" % (self.visible_frames, self.context_mask_ratio)

    def forward(self, x=None, *args, **kwargs):
        masks = super().forward(x=x, *args, **kwargs)
        n_lead = self.visible_frames * self.height * self.width
        if self.full_mask_prob > 0:
            hide = (torch.rand((masks.size(0), 1)).to(masks.device) < self.full_mask_prob).expand(-1, masks.size(-1) - n_lead)
            masks = torch.cat([masks[:, :n_lead], masks[:, n_lead:] | hide], -1)
        if self.vis_frame_sampler is not None:
            context = torch.cat([self.vis_frame_sampler(x) for _ in range(self.visible_frames)], -1)
            masks = torch.cat([context, masks[:, n_lead:]], -1)
        return masks
