"""Host-side mask helpers on the predictor path (bit-exact index work, tiny tensors).

`RectangularizeMasks` mirrors `cwm/models/masking.py:90-132`: it is the only cross-row operation
before the predictor call (`prediction.py:421`), it draws from torch's *global* RNG and mutates its
input in place, and both behaviours are kept so seeded runs reproduce the reference's masks.
"""
from __future__ import annotations

import torch


class RectangularizeMasks:
    """Make sure all masks in a batch have the same number of 1s and 0s (masking.py:90-132)."""

    def __init__(self, truncation_mode="min"):
        assert truncation_mode in ["min", "max", "mean", "full", "none", None], truncation_mode
        self._mode = truncation_mode
        self.last_num_masked = None

    def set_mode(self, mode):
        self._mode = mode

    def __call__(self, masks: torch.Tensor) -> torch.Tensor:
        self.last_num_masked = None
        if self._mode in ["none", None]:
            return masks
        assert isinstance(masks, torch.Tensor), type(masks)
        if self._mode == "full":
            return torch.ones_like(masks)
        shape = masks.shape
        masks = masks.flatten(1)
        num_masked = masks.float().sum(-1)
        target = {"min": torch.amin, "max": torch.amax, "mean": torch.mean}[self._mode](num_masked).long()
        vals = torch.cat([num_masked.long() - target, target.reshape(1)]).tolist()  # one host sync for the whole batch
        num_changes = vals[:-1]
        # every row now has exactly this many masked tokens: callers that need the count (the predictor's n_vis) read it
        # here instead of paying a second device round trip
        self.last_num_masked = int(vals[-1])
        for b, nc in enumerate(num_changes):
            if nc > 0:
                inds = torch.where(masks[b])[0]
                inds = inds[torch.randperm(inds.size(0))[:nc].to(inds.device)]
                masks[b, inds] = 0
            elif nc < 0:
                inds = torch.where(~masks[b])[0]
                inds = inds[torch.randperm(inds.size(0))[:-nc].to(inds.device)]
                masks[b, inds] = 1
        if list(masks.shape) != list(shape):
            masks = masks.view(*shape)
        return masks


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
