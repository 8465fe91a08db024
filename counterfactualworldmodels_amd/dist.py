"""Multi-GPU sharding of the batched counterfactual-sampling loop (BASELINE configs[3]).

The S counterfactual prompts over one frame pair are independent forwards (the reference merely chunks them on one device:
`prediction.py:513-540`, `segmentation.py:423-430`), so they shard across ranks with no collective inside the path: one
process per GPU, weights replicated.  Around the path there are exactly two collectives:

  1. ONE broadcast from rank 0 of a packed byte buffer {header | frame pair | prompt table | rectangularised masks}
     (1.2 MB + 4 KB + S*Nt bytes).  The masks travel because `RectangularizeMasks` is the only cross-row operation before the
     path: rank 0 applies it once to all S rows -- the same rows, the same torch-RNG draws as the single-process call
     (segmentation.py:342) -- so sharding cannot change which patches are un-masked.
  2. the all-gather of the per-rank prediction blocks, issued per chunk of 32 rows: ALWAYS the plain equal-block all-gather
     (`ncclAllGather`, the tuned ring collective) -- in place in the result when every rank has one equal chunk (8 ranks x 32),
     otherwise through a [world, chunk, ...] staging block per chunk that one strided copy places into the result (2 / 4 ranks:
     4 / 2 chunks per rank; ragged shards send a padded block).

On GPUs the collectives are RCCL called directly through the C ABI (`cwm_comm_*`, `cwm_broadcast`, `cwm_allgatherv` in
include/cwm_hip.h; `RcclComm`); the launcher's `torch.distributed` group is used only to hand the 128-byte RCCL id to the
other ranks.  `TorchComm` runs the same logic on any torch.distributed backend -- it is what the world_size > 1 CPU tests
use (gloo).  Per-rank fixed cost at 8 ranks: on rank 0 the MASKS of all S prompts (S*Nt bytes; the frames of a
prompt are built only by the rank that predicts it) + one host read-back for the rectangulariser, the broadcast, one 8-byte
header read on the other ranks, the gather; the predictor calls in between queue without host synchronisation.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

_HEADER_WORDS = 16
_MAGIC = 0x43574D31  # "CWM1"


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of `total` units owned by `rank` (sizes differ by at most 1)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _equal_contiguous(offsets: Sequence[int], counts: Sequence[int]) -> bool:
    """True when the blocks are non-empty, equal and back to back in rank order (what `ncclAllGather` needs)."""
    n = counts[0]
    return n > 0 and all(c == n for c in counts) and all(offsets[r] == offsets[0] + r * n for r in range(len(counts)))


# ---- communicators --------------------------------------------------------------------------------------------------------
class LocalComm:
    """World of one: every collective is the identity."""

    rank, world = 0, 1
    last_collective = None  # which collective the last gather used (tests / bench line)

    def broadcast_bytes(self, buf: torch.Tensor, src: int = 0) -> None:
        pass

    def all_gather_blocks(self, local: torch.Tensor, out: torch.Tensor, counts: Sequence[int]) -> torch.Tensor:
        """`out` [sum(counts), ...] receives rank r's `counts[r]` rows at their offset (rank order)."""
        offsets = [sum(counts[:r]) for r in range(self.world)]
        if counts[self.rank]:
            out[offsets[self.rank] : offsets[self.rank] + counts[self.rank]].copy_(local)
        return self.all_gather_rows(out, offsets, counts)

    def all_gather_rows(self, out: torch.Tensor, offsets: Sequence[int], counts: Sequence[int]) -> torch.Tensor:
        """In place: rows [offsets[r], offsets[r] + counts[r]) of `out` are valid on rank r before the call and on every rank
        after it.  The per-chunk form of the gather (`sharded_counterfactual_predictions`) calls this once per chunk."""
        return out

    def all_gather_stage(self, stage: torch.Tensor) -> torch.Tensor:
        """In place on a contiguous staging block `stage` [world, n, ...]: `stage[rank]` is valid on this rank before the call, all of it on every
        rank after it.  Equal blocks back to back by construction: the plain all-gather on every backend (`ncclAllGather` on RCCL)."""
        assert stage.shape[0] == self.world and stage.is_contiguous()
        n = stage.shape[1]
        return self.all_gather_rows(stage.view((self.world * n,) + tuple(stage.shape[2:])), [r * n for r in range(self.world)], [n] * self.world)

    def all_reduce_sum(self, t: torch.Tensor) -> None:
        pass


class TorchComm(LocalComm):
    """The same three collectives on a torch.distributed process group (gloo in the CPU tests)."""

    def __init__(self, group=None):
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def broadcast_bytes(self, buf, src=0):
        dist.broadcast(buf, src=src, group=self.group)

    def all_gather_rows(self, out, offsets, counts):
        self.last_collective = "all_gather_into_tensor" if _equal_contiguous(offsets, counts) else "all_gather(padded)"
        if max(counts) == 0:
            return out
        me = out[offsets[self.rank] : offsets[self.rank] + counts[self.rank]]
        if _equal_contiguous(offsets, counts):  # equal blocks, back to back: the plain all-gather straight into place
            dist.all_gather_into_tensor(out[offsets[0] : offsets[0] + self.world * counts[0]], me.contiguous(), group=self.group)
            return out
        width = max(counts)  # ragged: equal-sized padded blocks through the collective, trimmed into place
        pad = out.new_zeros((width,) + tuple(out.shape[1:]))
        pad[: counts[self.rank]] = me
        parts = [torch.empty_like(pad) for _ in range(self.world)]
        dist.all_gather(parts, pad, group=self.group)
        for r, n in enumerate(counts):
            if r != self.rank and n:
                out[offsets[r] : offsets[r] + n] = parts[r][:n]
        return out

    def all_reduce_sum(self, t):
        dist.all_reduce(t, group=self.group)


class RcclComm(LocalComm):
    """RCCL over xGMI through the C ABI (include/cwm_hip.h: cwm_comm_init / cwm_broadcast / cwm_allgatherv / cwm_allreduce_sum_f32),
    enqueued on the current HIP stream of `device`."""

    def __init__(self, rank: int, world: int, unique_id: bytes, device):
        from . import _lib

        self._lib_mod = _lib
        self.rank, self.world, self.device = rank, world, torch.device(device)
        ident = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.get_lib().cwm_comm_init(rank, world, ident, C.byref(handle)))
        self._handle = handle

    @staticmethod
    def new_unique_id() -> bytes:
        from . import _lib

        buf = (C.c_uint8 * 128)()
        _lib.check(_lib.get_lib().cwm_comm_unique_id(buf))
        return bytes(buf)

    def close(self):
        if getattr(self, "_handle", None):
            self._lib_mod.get_lib().cwm_comm_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return self._lib_mod.current_stream_handle(self.device)

    def broadcast_bytes(self, buf, src=0):
        assert buf.is_cuda and buf.is_contiguous()
        with torch.cuda.device(self.device):
            self._lib_mod.check(self._lib_mod.get_lib().cwm_broadcast(self._handle, buf.data_ptr(), buf.numel() * buf.element_size(), src, self._stream()))

    @staticmethod
    def gather_args(base: int, row: int, offsets: Sequence[int], counts: Sequence[int], rank: int):
        """The pointer arithmetic of `all_gather_rows` as a pure function (so that the 8-rank layout can be checked without 8 GPUs).
        Equal blocks back to back: ("allgather", send, recv, bytes_per_rank) with NCCL's in-place contract send == recv + rank * bytes;
        otherwise ("allgatherv", send or None, base, byte offsets, byte counts)."""
        world = len(counts)
        assert len(offsets) == world and 0 <= rank < world and row >= 0
        if _equal_contiguous(offsets, counts):
            nbytes = counts[0] * row
            send, recv = base + offsets[rank] * row, base + offsets[0] * row
            assert send == recv + rank * nbytes  # ncclAllGather in place: every rank's block already sits at its slot of the receive buffer
            return "allgather", send, recv, nbytes
        sizes, offs = [n * row for n in counts], [o * row for o in offsets]
        for r in range(world):  # blocks may be empty or differ, but never overlap
            for q in range(r + 1, world):
                assert offs[r] + sizes[r] <= offs[q] or offs[q] + sizes[q] <= offs[r] or sizes[r] == 0 or sizes[q] == 0, (offs, sizes)
        return "allgatherv", (base + offs[rank] if counts[rank] else None), base, offs, sizes

    def all_gather_rows(self, out, offsets, counts):
        """Equal blocks back to back -> `ncclAllGather` in place (`cwm_allgather`, the tuned ring collective); anything else ->
        `cwm_allgatherv` (one group of per-root broadcasts, blocks may differ or be empty).  Both on the current stream."""
        assert out.is_cuda and out.is_contiguous()
        lib, row = self._lib_mod.get_lib(), (out[0].numel() * out.element_size() if out.shape[0] else 0)
        args = self.gather_args(out.data_ptr(), row, offsets, counts, self.rank)
        assert max(o + n for o, n in zip(offsets, counts)) <= out.shape[0], "a block lies outside the result"
        with torch.cuda.device(self.device):
            if args[0] == "allgather":
                self.last_collective = "ncclAllGather"
                self._lib_mod.check(lib.cwm_allgather(self._handle, args[1], args[2], args[3], self._stream()))
            else:
                self.last_collective = "grouped ncclBroadcast"
                sizes = (C.c_size_t * self.world)(*args[4])
                offs = (C.c_size_t * self.world)(*args[3])
                self._lib_mod.check(lib.cwm_allgatherv(self._handle, args[1], args[2], offs, sizes, self._stream()))
        return out

    def all_reduce_sum(self, t):
        assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float32
        with torch.cuda.device(self.device):
            self._lib_mod.check(self._lib_mod.get_lib().cwm_allreduce_sum_f32(self._handle, t.data_ptr(), t.numel(), self._stream()))


_comms: dict = {}


def get_comm(device=None) -> LocalComm:
    """The process's communicator: `LocalComm` without a launcher; with `torch.distributed` initialised, `RcclComm` for a CUDA/HIP
    device (the RCCL id is created on rank 0 and handed out through the launcher's group) and `TorchComm` otherwise
    (CWM_COMM=torch selects `TorchComm` on GPUs too: the launcher's own NCCL/RCCL group)."""
    key = torch.device(device).type if device is not None else "cpu"  # one communicator per device type: a first CPU call must not
    if key in _comms:                                                   # pin `TorchComm` for later GPU calls
        return _comms[key]
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        _default = LocalComm()
    elif key == "cuda" and os.environ.get("CWM_COMM", "rccl") != "torch":
        # rank 0 creates the RCCL id; a failure there is broadcast too, so that every rank raises instead of waiting for an id
        box = [None]
        if dist.get_rank() == 0:
            try:
                box[0] = RcclComm.new_unique_id()
            except Exception as e:  # noqa: BLE001
                box[0] = RuntimeError("rank 0 could not create an RCCL id: %s" % e)
        dist.broadcast_object_list(box, src=0)
        if isinstance(box[0], Exception):
            raise box[0]
        _default = RcclComm(dist.get_rank(), dist.get_world_size(), box[0], device)
    else:
        _default = TorchComm()
    _comms[key] = _default
    return _default


def reset_comm():
    for c in _comms.values():
        if isinstance(c, RcclComm):
            c.close()
    _comms.clear()


# ---- packed broadcast ---------------------------------------------------------------------------------------------------
def _aligned(n: int) -> int:
    return (n + 15) // 16 * 16


def _layout(x_shape: Sequence[int], table_shape: Sequence[int], n_tok: int):
    """Byte offsets of {header | frame pair fp32 | prompt table int32 | masks uint8} and the total size."""
    S = table_shape[0]
    o_x = _aligned(_HEADER_WORDS * 8)
    o_t = o_x + _aligned(4 * int(torch.Size(x_shape).numel()))
    o_m = o_t + _aligned(4 * int(torch.Size(table_shape).numel()))
    return o_x, o_t, o_m, o_m + _aligned(S * n_tok)


def pack_inputs(x: torch.Tensor, table: torch.Tensor, masks: torch.Tensor, n_masked: int, device) -> torch.Tensor:
    """Rank 0: one contiguous uint8 buffer holding everything the other ranks need."""
    assert x.dim() == 5 and x.shape[0] == 1 and table.dim() == 2 and masks.shape[0] == table.shape[0]
    o_x, o_t, o_m, total = _layout(x.shape, table.shape, masks.shape[1])
    buf = torch.zeros(total, dtype=torch.uint8, device=device)
    head = torch.tensor([_MAGIC, *x.shape, *table.shape, masks.shape[1], n_masked] + [0] * (_HEADER_WORDS - 10), dtype=torch.int64)
    if torch.device(device).type == "cuda":  # (a pageable host -> device copy synchronises; 128 bytes from pinned memory do not)
        head = head.pin_memory()
    buf[: _HEADER_WORDS * 8].copy_(head.view(torch.uint8), non_blocking=True)
    buf[o_x : o_x + 4 * x.numel()] = x.to(device=device, dtype=torch.float32).contiguous().view(-1).view(torch.uint8)
    buf[o_t : o_t + 4 * table.numel()] = table.to(device=device, dtype=torch.int32).contiguous().view(-1).view(torch.uint8)
    buf[o_m : o_m + masks.numel()] = masks.to(device=device).contiguous().view(-1).view(torch.uint8)
    return buf


def unpack_inputs(buf: torch.Tensor):
    """Every rank: (x [1,T,C,H,W] fp32, table [S,K] int32, masks [S,Nt] bool, n_masked) as views of the received buffer.  Reads the
    128-byte header back to the host (the one synchronisation a receiving rank needs)."""
    head = buf[: _HEADER_WORDS * 8].view(torch.int64).tolist()
    assert head[0] == _MAGIC, "packed prompt buffer: bad magic"
    x_shape, t_shape, n_tok, n_masked = head[1:6], head[6:8], head[8], head[9]
    o_x, o_t, o_m, _ = _layout(x_shape, t_shape, n_tok)
    n_x, n_t = int(torch.Size(x_shape).numel()), int(torch.Size(t_shape).numel())
    x = buf[o_x : o_x + 4 * n_x].view(torch.float32).view(*x_shape)
    table = buf[o_t : o_t + 4 * n_t].view(torch.int32).view(*t_shape)
    masks = buf[o_m : o_m + t_shape[0] * n_tok].view(torch.bool).view(t_shape[0], n_tok)
    return x, table, masks, n_masked


# ---- the config-4 loop --------------------------------------------------------------------------------------------------
class PhaseTimes:
    """HIP-event spans around the phases of ONE sharded call -- {build (rank 0: all prompts + rectangularise + pack), broadcast, own_prompts,
    predict, gather} -- on the stream each phase is issued on (the gathers run on the side stream).  `bench.py` puts every rank's result in its
    line, so that a bad scaling curve can be read from one run: which rank, which phase.  Costs nothing when not passed."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.spans: List[Tuple[str, object, object]] = []
        self.host: dict = {}  # {phase: milliseconds the HOST spent inside the phase's spans} (issue time; `result()` reports it as "<phase>_host")

    class _Span:
        def __init__(self, owner, phase, stream):
            self.owner, self.phase, self.stream = owner, phase, stream

        def __enter__(self):
            import time

            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record(self.stream or torch.cuda.current_stream(self.owner.device))
            self.t0 = time.perf_counter()

        def __exit__(self, *exc):
            import time

            host = time.perf_counter() - self.t0
            b = torch.cuda.Event(enable_timing=True)
            b.record(self.stream or torch.cuda.current_stream(self.owner.device))
            self.owner.spans.append((self.phase, self.a, b))
            self.owner.host[self.phase] = self.owner.host.get(self.phase, 0.0) + 1e3 * host
            return False

    def span(self, phase: str, stream=None):
        return PhaseTimes._Span(self, phase, stream)

    def result(self) -> dict:
        """{phase: summed milliseconds} (synchronises the device)."""
        torch.cuda.synchronize(self.device)
        out: dict = {}
        for phase, a, b in self.spans:
            out[phase] = out.get(phase, 0.0) + a.elapsed_time(b)
        for phase, ms in self.host.items():
            out[phase + "_host"] = ms
        return out


class _NoTimes:
    def span(self, phase, stream=None):
        import contextlib

        return contextlib.nullcontext()


class RemoteRankError(RuntimeError):
    """Rank 0 failed while preparing the prompts; raised on the other ranks instead of leaving them inside a collective."""


def _side_stream(comm, device):
    """The stream the per-chunk gathers run on (one per communicator), so that a chunk's gather rides under the next chunk."""
    st = getattr(comm, "_gather_stream", None)
    if st is None:
        st = torch.cuda.Stream(device=device)
        comm._gather_stream = st
    return st


def sharded_counterfactual_predictions(
    x: Optional[torch.Tensor],
    prompts: Optional[torch.Tensor],
    build_fn: Callable[..., Tuple[Optional[torch.Tensor], torch.Tensor]],
    rect_fn: Callable[[torch.Tensor], Tuple[torch.Tensor, int]],
    predict_fn: Callable[[torch.Tensor, torch.Tensor, int, int], torch.Tensor],
    device,
    chunk: int = 32,
    gather: bool = True,
    comm: Optional[LocalComm] = None,
    shapes: Optional[Tuple[Sequence[int], Sequence[int], int]] = None,
    times: Optional[PhaseTimes] = None,
) -> torch.Tensor:
    """S prompts over ONE frame pair, sharded over the ranks of `comm`.

    x [1,T,C,H,W], prompts [S,K] int32: valid on rank 0 (ignored elsewhere).
      build_fn(x, prompt_rows, frames=True) -> (x_rows [n,T,C,H,W] | None, mask_rows [n,Nt])
                                                                            device-side prompt construction, NOT rectangularised;
                                                                            frames=False: masks only (x_rows is None)
      rect_fn(masks [S,Nt])    -> (masks, n_masked)                         rank 0 only, once for all S rows (global RNG order kept)
      predict_fn(x_rows, mask_rows, n_masked, chunk) -> y [n, ...]          the predictor over `chunk` rows per call, no host sync
    shapes = (x.shape, prompts.shape, Nt), if known on every rank, lets the receivers size the packed buffer without a
    metadata collective.  Returns all S predictions in prompt order on every rank (only the local block with gather=False).

    With more than one rank, rank 0 builds the MASKS of all S prompts (the rectangulariser is the only cross-row step: S*Nt bytes) and, like every
    other rank, the frames of its own rows only -- the reference builds every prompt's frames in one place (segmentation.py:324-342) because it
    predicts them in one place; here 7/8 of that work would sit on every rank's critical path (the peers wait for the broadcast) to be thrown away.

    The gather is issued PER CHUNK: as soon as a rank's chunk c is queued, the all-gather of every rank's chunk c is queued on a side stream behind
    it and runs under chunk c + 1.  It is always the equal-block all-gather (`all_gather_stage` / `ncclAllGather`): with one equal chunk per rank
    (8 ranks, 256 prompts) in place in the result; otherwise every rank's chunk c lands in a [world, chunk, ...] staging block that one strided copy
    (side stream too) moves to rows `lo_r + c*chunk` of the result.  A failure on rank 0 before the broadcast reaches the other ranks as a status
    word in the packed header (they raise `RemoteRankError`)."""
    comm = comm or get_comm(device)
    rank, world = comm.rank, comm.world
    T = times if times is not None else _NoTimes()
    failure = None
    x_all = None
    if rank == 0:
        try:
            with T.span("build"):
                xb, table = x.to(device), prompts.to(device)
                if world == 1:
                    x_all, masks = build_fn(xb, table)
                else:
                    _, masks = build_fn(xb, table, frames=False)
                masks, n_masked = rect_fn(masks)
                if world > 1:
                    buf = pack_inputs(xb, table, masks, n_masked, device)
                    if shapes is not None and buf.numel() != _layout(shapes[0], shapes[1], shapes[2])[3]:
                        raise ValueError("packed prompt buffer has %d bytes, the shape hint says %d" % (buf.numel(), _layout(shapes[0], shapes[1], shapes[2])[3]))
        except Exception as e:  # noqa: BLE001 -- the peers are (about to be) inside the broadcast: tell them, then raise here
            if world == 1:
                raise
            failure = e
    if world > 1:
        if shapes is None:  # receivers learn the buffer size from a first 8-byte broadcast (0 = rank 0 failed)
            meta = torch.tensor([buf.numel() if rank == 0 and failure is None else 0], dtype=torch.int64, device=device).view(torch.uint8)
            comm.broadcast_bytes(meta, 0)
            total = int(meta.view(torch.int64).item())
            if failure is not None:
                raise failure
            if total == 0:
                raise RemoteRankError("rank 0 failed while building the prompts")
        else:
            total = _layout(shapes[0], shapes[1], shapes[2])[3]
            if failure is not None:
                buf = torch.zeros(total, dtype=torch.uint8, device=device)  # magic 0 = status word "failed"
        if rank != 0:
            buf = torch.empty(total, dtype=torch.uint8, device=device)
        with T.span("broadcast"):
            comm.broadcast_bytes(buf, 0)
        if failure is not None:
            raise failure
        if rank != 0:
            if int(buf[:8].view(torch.int64).item()) == 0:
                raise RemoteRankError("rank 0 failed while building the prompts")
            xb, table, masks, n_masked = unpack_inputs(buf)
    S = table.shape[0]
    bounds = [shard_range(S, r, world) for r in range(world)]
    lo, hi = bounds[rank]
    if hi > lo:
        with T.span("own_prompts"):
            x_own = x_all[lo:hi] if x_all is not None else build_fn(xb, table[lo:hi])[0]
    if not gather or world == 1:
        if hi > lo:
            with T.span("predict"):
                return predict_fn(x_own, masks[lo:hi], n_masked, chunk)
        # an empty slice keeps the right trailing shape: predict one row, keep none
        return predict_fn(build_fn(xb, table[:1])[0], masks[:1], n_masked, chunk)[:0]
    # ---- chunk c of every rank is gathered while chunk c + 1 is predicted
    on_gpu = torch.device(device).type == "cuda"
    sizes = [h - l for l, h in bounds]
    n_chunks = max(1, -(-max(sizes) // chunk))
    in_place = n_chunks == 1 and min(sizes) == max(sizes)  # one equal block per rank: they already sit back to back in the result
    out = stage = None
    if on_gpu:
        main, side = torch.cuda.current_stream(device), _side_stream(comm, device)
    for c in range(n_chunks):
        a, b = min(lo + c * chunk, hi), min(lo + (c + 1) * chunk, hi)
        if b > a:
            with T.span("predict"):
                y = predict_fn(x_own[a - lo : b - lo], masks[a:b], n_masked, chunk)
        elif out is None:  # this rank owns nothing at all: one row gives the trailing shape
            y = predict_fn(build_fn(xb, table[:1])[0], masks[:1], n_masked, chunk)[:0]
        if out is None:
            out = torch.empty((S,) + tuple(y.shape[1:]), dtype=y.dtype, device=y.device)
            if not in_place:  # (freed after the final join below: the side stream is done with it by then)
                stage = torch.empty((n_chunks, world * min(chunk, max(sizes))) + tuple(y.shape[1:]), dtype=y.dtype, device=y.device)
        offs = [min(l + c * chunk, h) for l, h in bounds]
        cnts = [min(l + (c + 1) * chunk, h) - o for (l, h), o in zip(bounds, offs)]
        if in_place:
            out[a:b].copy_(y)
        else:
            width = max(cnts)            # ragged shards: every rank sends `width` rows, the surplus rows are never copied out
            block = stage[c, : world * width].view((world, width) + tuple(y.shape[1:]))  # equal blocks, back to back
            if b > a:
                block[rank, : b - a].copy_(y)
        if on_gpu and n_chunks > 1:
            side.wait_stream(main)
        with (torch.cuda.stream(side) if on_gpu and n_chunks > 1 else _null()), T.span("gather", side if on_gpu and n_chunks > 1 else None):
            if in_place:
                comm.all_gather_rows(out, offs, cnts)
            else:
                comm.all_gather_stage(block)
                _place(out, block, offs, cnts, sizes, c * chunk)
    if on_gpu and n_chunks > 1:
        main.wait_stream(side)
    return out


def _null():
    import contextlib

    return contextlib.nullcontext()


def _place(out: torch.Tensor, block: torch.Tensor, offs: Sequence[int], cnts: Sequence[int], sizes: Sequence[int], start: int) -> None:
    """Rows [0, cnts[r]) of `block[r]` -> rows [offs[r], offs[r] + cnts[r]) of `out`: ONE strided copy when the shards and this chunk's blocks are
    equal (the result viewed as [world, rows per rank, ...]), one copy per rank otherwise."""
    world, n = len(cnts), cnts[0]
    if n > 0 and all(k == n for k in cnts) and all(k == sizes[0] for k in sizes):
        out.view((world, sizes[0]) + tuple(out.shape[1:]))[:, start : start + n].copy_(block[:, :n])
        return
    for r in range(world):
        if cnts[r]:
            out[offs[r] : offs[r] + cnts[r]].copy_(block[r, : cnts[r]])


def prompt_hooks(G, frame: Optional[int] = -1):
    """(build_fn, rect_fn, predict_fn) of `sharded_counterfactual_predictions` for a `segmentation.FlowGenerator` and prompt rows
    (active_h, active_w, dy, dx): one active patch of frame 1 moved by (dy, dx) patches, nothing passive (cfg 4)."""
    from .prediction import _RectBatch

    def build(xb, rows, frames=True):
        """The prompt rows -> (frames [n,T,C,H,W] | None, masks [n,Nt]): `cwm_prompt_table_expand` (table -> dense active / passive masks + shifts, one launch) and
        `cwm_shift_prompts` on the image as a static two-frame movie (only frame 0 exists in memory)."""
        from . import _lib

        n_rows = rows.shape[0]
        T = 2
        xb = xb[:, :1]
        G.inp_shape = (1, T) + tuple(xb.shape[2:])
        _, gh, gw = G.mask_shape
        dev = xb.device
        if not xb.is_cuda:
            raise RuntimeError("counterfactual prompts are built on the GPU (no CPU fallback); got a %s tensor" % dev)
        rows = rows.to(device=dev, dtype=torch.int32).contiguous()
        active = torch.empty((n_rows, T * gh * gw), device=dev, dtype=torch.bool)
        passive = torch.empty_like(active)
        shifts = torch.empty((n_rows, 2), device=dev, dtype=torch.int32)
        with torch.cuda.device(dev):
            _lib.check(_lib.get_lib().cwm_prompt_table_expand(rows.data_ptr(), n_rows, T, gh, gw, 1, active.data_ptr(), passive.data_ptr(), shifts.data_ptr(),
                                                             _lib.current_stream_handle(dev)))
        return G._shift_rows(xb, passive, active, shifts, 1, True, samples_per_movie=n_rows, frames=frames, num_frames=T)

    def rect(masks):
        masks = G.mask_rectangularizer(masks)
        return masks, G.mask_rectangularizer.last_num_masked

    def predict(xs, ms, n_masked, chunk):
        y = G._run_rect_batch(_RectBatch(xs, ms, n_masked), rows_per_call=chunk)
        G.reset_padding_masks()
        return G._select_frame(y, frame)

    return build, rect, predict


# ---- sample statistics over sharded samples -----------------------------------------------------------
def all_gather_last_axis(x_local: torch.Tensor, comm: Optional[LocalComm] = None) -> torch.Tensor:
    """Concatenate per-rank tensors along the LAST axis (rank order = sample order); sizes may differ (or be 0)."""
    comm = comm or get_comm(x_local.device)
    if comm.world == 1:
        return x_local
    n = torch.zeros(comm.world, dtype=torch.float32, device=x_local.device)
    n[comm.rank] = x_local.shape[-1]
    comm.all_reduce_sum(n)
    counts = [int(v) for v in n.tolist()]
    rows = x_local.movedim(-1, 0).contiguous()  # sample-major blocks: the gather concatenates along dim 0
    out = torch.empty((sum(counts),) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
    return comm.all_gather_blocks(rows, out, counts).movedim(0, -1).contiguous()


def sharded_flow_corrs(flow_samples_local: torch.Tensor, downsample: int = 1, use_covariance: bool = True, gather: bool = False,
                       features_fn: Optional[Callable] = None, cov_rows_fn: Optional[Callable] = None, comm: Optional[LocalComm] = None) -> torch.Tensor:
    """`FlowGenerator.compute_flow_corrs` (segmentation.py:479-547) when the S flow samples are spread over the ranks
    (`flow_samples_local` [B,C,H,W,S_rank]).  One collective on small data: the pooled features [B,P,S_rank] are all-gathered
    (P*S*4 bytes per frame pair: 12.8 MB at P = 12544, S = 256); then every rank computes rows `shard_range(P, rank, world)`
    of the [P,P] matrix -- the 629-MB result stays sharded unless `gather`.  Returns [B,nrows,P] ([B,P,P] with gather).
    `features_fn` / `cov_rows_fn` default to the HIP kernels (flowstats.py); the CPU tests inject the oracle."""
    if features_fn is None or cov_rows_fn is None:
        from . import flowstats

        features_fn = features_fn or flowstats.flow_features
        cov_rows_fn = cov_rows_fn or flowstats.feature_cov_rows
    comm = comm or get_comm(flow_samples_local.device)
    B, _, H, W, S_local = flow_samples_local.shape
    ds = int(downsample or 1)
    if S_local > 0:
        x_local = features_fn(flow_samples_local, ds)
    else:  # a rank with an empty shard still joins the collective
        x_local = torch.zeros((B, (H // ds) * (W // ds), 0), dtype=torch.float32, device=flow_samples_local.device)
    x = all_gather_last_axis(x_local, comm)
    P = x.shape[1]
    lo, hi = shard_range(P, comm.rank, comm.world)
    slab = cov_rows_fn(x, lo, hi - lo, use_covariance)
    if not gather or comm.world == 1:
        return slab
    counts = [shard_range(P, r, comm.world)[1] - shard_range(P, r, comm.world)[0] for r in range(comm.world)]
    rows = slab.transpose(0, 1).contiguous()
    out = torch.empty((P,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
    return comm.all_gather_blocks(rows, out, counts).transpose(0, 1).contiguous()


def sharded_mean_motion_map(flows_local: torch.Tensor, normalize_per_sample: bool = False, normalize: bool = True, eps: float = 1e-2,
                            sum_fn: Optional[Callable] = None, finish_fn: Optional[Callable] = None, comm: Optional[LocalComm] = None) -> torch.Tensor:
    """`FlowGenerator.compute_mean_motion_map` (segmentation.py:257-276) over sharded samples: per-rank sums of the (per-sample
    normalised) magnitudes, ONE all-reduce of [B,1,H,W] with the sample count appended, then the range normalisation on every rank."""
    if sum_fn is None or finish_fn is None:
        from . import flowstats

        sum_fn = sum_fn or flowstats.motion_map_sum
        finish_fn = finish_fn or flowstats.finish_motion_map
    comm = comm or get_comm(flows_local.device)
    B, _, H, W, S = flows_local.shape
    total = sum_fn(flows_local, normalize_per_sample, eps) if S > 0 else torch.zeros((B, 1, H, W), dtype=torch.float32, device=flows_local.device)
    packed = torch.cat([total.reshape(-1).float(), torch.tensor([float(S)], device=total.device)])
    comm.all_reduce_sum(packed)
    return finish_fn(packed[:-1].view(B, 1, H, W), int(round(packed[-1].item())), normalize, eps)
