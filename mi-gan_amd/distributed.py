"""Batch sharding of the generator forward over the GPUs of one node.

Images in a batch are independent (no batch statistics anywhere in the generator), so the only
exchange step is the gather of the output shards: one ``all_gather`` (RCCL over xGMI when the
backend is "nccl"; gloo on CPU in the tests).  One process per GPU, weights replicated.

The reference has no multi-GPU inference path (its only parallelism is training-time DDP,
lib/utils.py:41-46); the contract here is: gathered output == single-GPU output of the same batch.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import os

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [start, stop) of ``total`` images for ``rank``; sizes differ by at most 1."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def gather_outputs(y_local: torch.Tensor, total: int, group=None) -> torch.Tensor:
    """All-gather output shards [n_r,3,R,R] into [total,3,R,R] on every rank (rank order)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return y_local
    world = dist.get_world_size(group)
    sizes = [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]
    if len(set(sizes)) == 1:
        out = torch.empty((total,) + tuple(y_local.shape[1:]), dtype=y_local.dtype, device=y_local.device)
        dist.all_gather_into_tensor(out, y_local.contiguous(), group=group)
        return out
    # ragged shards: pad to the largest, gather, drop the padding
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(y_local.shape[1:]), dtype=y_local.dtype, device=y_local.device)
    pad[: y_local.shape[0]] = y_local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)


def sharded_forward(forward: Callable[..., torch.Tensor], x_global, group=None, gather: bool = True) -> torch.Tensor:
    """Run ``forward`` on this rank's slice of ``x_global`` and (optionally) gather all outputs.

    ``x_global`` is one tensor or a tuple of tensors sharing the batch dimension, all sliced alike and passed as
    positional arguments -- e.g. ``(x, z)`` for the Co-Mod-GAN generator, whose latent is per image
    (``sharded_forward(lambda x, z: model(x, z=z, noise_mode="const"), (x, z))``).  Co-Mod-GAN images interact only through
    the batch-wide style normalisation (stylegan.py:139), which the demodulation cancels up to its 1e-8 epsilon, so the
    gathered output equals the single-GPU output to fp32 rounding rather than bit for bit."""
    many = isinstance(x_global, (tuple, list))
    xs = tuple(x_global) if many else (x_global,)
    total = xs[0].shape[0]
    if any(t.shape[0] != total for t in xs):
        raise ValueError("all sharded inputs need the same batch size")
    if not dist.is_initialized():
        return forward(*xs)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_range(total, rank, world)
    y = forward(*(t[lo:hi] for t in xs))
    return gather_outputs(y, total, group) if gather else y


class OutputGather:
    """Double-buffered asynchronous all-gather of equal-size output shards, IN PLACE and per sub-batch.

    Every slot owns the receive buffers of the collective; ``shards(slot)`` hands out this rank's slices of them and the forward writes
    its images straight there (``Generator.forward(x, out=...)`` / ``forward_parts``), so ``all_gather_into_tensor`` runs in place: no
    local copy of the shard (100.7 MB per step at 32 x 3 x 512 x 512 fp32), and at world size 1 no data movement at all.

    ``chunks`` = the sub-batch sizes of the forward (``Generator.sub_batches(n)``, e.g. [16, 16]): one receive buffer and one collective
    per sub-batch ("split the shard into 2-4 micro-batches and gather chunk k while computing k+1").  ``forward_and_submit``
    runs ``Generator.forward_parts`` and enqueues the collective of sub-batch k right behind that sub-batch -- sub-batch 0's shard goes out
    over xGMI while sub-batch 1 still computes, and the gathers of step i overlap step i+1.  ``submit(y_local)`` is the copying form for
    producers that cannot write in place (uint8 I/O, Co-Mod-GAN): one copy into the slot's shard views, then the same collectives.

    ``result(slot)`` waits for that slot's collectives and returns the gathered tensor in rank order ([world * n, ...]; zero-copy with
    one chunk, assembled from the chunk buffers otherwise -- ``result_chunks`` returns the [world, n_k, ...] views without a copy), valid
    until the slot is reused ``depth`` submits later; ``drain()`` waits for everything in flight.
    """

    def __init__(self, shard_shape, dtype, device, group=None, depth: int = 2, chunks=None):
        if not dist.is_initialized():
            raise RuntimeError("OutputGather needs an initialised process group")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        shard_shape = tuple(shard_shape)
        self.n = int(shard_shape[0])
        self.chunks = [int(c) for c in (chunks or [self.n])]
        if sum(self.chunks) != self.n or any(c <= 0 for c in self.chunks):
            raise ValueError(f"chunks {self.chunks} do not partition the shard of {self.n} images")
        self.device = torch.device(device)
        rest = shard_shape[1:]
        # one receive buffer per (slot, chunk): [world * n_k, ...]; this rank's slice of it is where the forward writes
        self.bufs = [[torch.empty((self.world * c,) + rest, dtype=dtype, device=device) for c in self.chunks] for _ in range(max(1, depth))]
        self.works = [[None] * len(self.chunks) for _ in self.bufs]
        self.keep = [None] * len(self.bufs)          # a copied-from local shard must stay alive until its copy was enqueued
        self.next = 0
        self.part_streams = []
        if self.device.type == "cuda" and len(self.chunks) > 1:
            self.part_streams = [torch.cuda.Stream(self.device) for _ in self.chunks[1:]]

    # ---- slots --------------------------------------------------------------------------------------------------------------------
    def _take_slot(self) -> int:
        slot = self.next
        self.next = (slot + 1) % len(self.bufs)
        self._wait(slot)                              # the buffers are about to be overwritten
        return slot

    def _wait(self, slot: int) -> None:
        for k, w in enumerate(self.works[slot]):
            if w is not None:
                w.wait()
                self.works[slot][k] = None
        self.keep[slot] = None

    def shards(self, slot: int):
        """this rank's slice of every chunk buffer of `slot`: [n_k, ...] views the producer writes into"""
        return [b[self.rank * c:(self.rank + 1) * c] for b, c in zip(self.bufs[slot], self.chunks)]

    def _gather_chunk(self, slot: int, k: int):
        c = self.chunks[k]
        buf = self.bufs[slot][k]
        return dist.all_gather_into_tensor(buf, buf[self.rank * c:(self.rank + 1) * c], group=self.group, async_op=True)

    # ---- producers ----------------------------------------------------------------------------------------------------------------
    def submit(self, y_local: torch.Tensor) -> int:
        """copying form: y_local [n, ...] -> the slot's shard views (skipped where it already lives there), then one collective per chunk"""
        slot = self._take_slot()
        views = self.shards(slot)
        lo = 0
        for k, (v, c) in enumerate(zip(views, self.chunks)):
            src = y_local[lo:lo + c]
            if src.data_ptr() != v.data_ptr():
                v.copy_(src)
            lo += c
        self.keep[slot] = y_local
        for k in range(len(self.chunks)):
            self.works[slot][k] = self._gather_chunk(slot, k)
        return slot

    def forward_and_submit(self, model, x: torch.Tensor) -> int:
        """the in-place form for mi-gan_amd's Generator: the forward writes its images into the receive buffers, sub-batch by sub-batch, and
        each sub-batch's collective is enqueued right behind it"""
        slot = self._take_slot()
        outs = self.shards(slot)
        if len(outs) == 1:
            model(x, out=outs[0])
            self.works[slot][0] = self._gather_chunk(slot, 0)
            return slot
        on_gpu = self.device.type == "cuda"            # (a CPU device: the gloo tests drive this path with a stand-in model and no streams)
        cur = torch.cuda.current_stream(self.device) if on_gpu else None
        for s in self.part_streams:
            s.wait_stream(cur)                        # (whatever produced x and freed the workspace is behind us on the current stream)
        sizes = model.forward_parts(x, outs, self.part_streams)
        if list(sizes) != self.chunks:
            raise RuntimeError(f"the forward ran sub-batches {list(sizes)}, this gather was built for {self.chunks}")
        # the process group's communication stream picks up the CURRENT stream's position: sub-batch 0 here, sub-batch k on its own stream
        self.works[slot][0] = self._gather_chunk(slot, 0)
        for k in range(1, len(self.chunks)):
            if on_gpu:
                with torch.cuda.stream(self.part_streams[k - 1]):
                    self.works[slot][k] = self._gather_chunk(slot, k)
            else:
                self.works[slot][k] = self._gather_chunk(slot, k)
        for s in self.part_streams:
            cur.wait_stream(s)                        # the join migan_forward_parts leaves to its caller
        return slot

    # ---- consumers ----------------------------------------------------------------------------------------------------------------
    def result_chunks(self, slot: int):
        """[world, n_k, ...] view of every chunk buffer of `slot` (no copy), after its collectives completed"""
        self._wait(slot)
        return [b.view((self.world, c) + tuple(b.shape[1:])) for b, c in zip(self.bufs[slot], self.chunks)]

    def result(self, slot: int) -> torch.Tensor:
        """the gathered batch in rank order, [world * n, ...]"""
        parts = self.result_chunks(slot)
        if len(parts) == 1:
            return self.bufs[slot][0]
        return torch.cat(parts, dim=1).reshape((self.world * self.n,) + tuple(parts[0].shape[2:]))

    def drain(self) -> None:
        for slot in range(len(self.bufs)):
            self._wait(slot)


def _hip_runtime():
    """the HIP runtime torch itself has loaded (a second copy opened by its bare name would hand out stream handles torch cannot use)"""
    import ctypes
    lib_dir = os.path.join(os.path.dirname(torch.__file__), "lib")
    for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
        path = os.path.join(lib_dir, name)
        if os.path.exists(path):
            return ctypes.CDLL(path)
    return ctypes.CDLL("libamdhip64.so")      # (a system ROCm build of torch links the system runtime)


class MaskedStream:
    """torch.cuda.ExternalStream over a stream made by hipExtStreamCreateWithCUMask; destroyed with the object (or close())"""

    def __init__(self, hip, handle: int, device):
        self._hip, self._handle = hip, handle
        self.stream = torch.cuda.ExternalStream(handle, device=device)

    def close(self) -> None:
        if self._handle:
            import ctypes
            self._hip.hipStreamDestroy(ctypes.c_void_p(self._handle))
            self._handle = 0

    def __del__(self):   # pragma: no cover - interpreter shutdown order
        try:
            self.close()
        except Exception:
            pass


def cu_masked_stream(device, reserve_cus: int, total_cus: int = 0, xcds: int = 8):
    """A HIP stream restricted to total_cus - reserve_cus compute units (hipExtStreamCreateWithCUMask), wrapped as a
    torch.cuda.ExternalStream: kernels launched on it leave ``reserve_cus`` CUs (reserve_cus / xcds on every XCD) to whatever else
    is running -- here the RCCL kernels of the overlapped output all-gather, which otherwise queue behind layers that fill all
    256 CUs of an MI355X.  CU i of the mask is bit i (logical CU numbering: XCD = i % xcds, as workgroups are dealt).
    total_cus = 0: the device's multiprocessor count.  The returned stream keeps its owner alive (``stream.owner.close()`` destroys it)."""
    import ctypes
    device = torch.device(device)
    if total_cus <= 0:
        total_cus = int(torch.cuda.get_device_properties(device).multi_processor_count)
    if reserve_cus <= 0 or reserve_cus % xcds or reserve_cus >= total_cus:
        raise ValueError(f"reserve_cus must be a positive multiple of {xcds} below {total_cus}")
    per_xcd = reserve_cus // xcds
    words = [0] * ((total_cus + 31) // 32)
    for cu in range(total_cus):
        if cu // xcds >= per_xcd:                      # the first per_xcd CUs of every XCD stay free
            words[cu // 32] |= 1 << (cu % 32)
    hip = _hip_runtime()
    stream = ctypes.c_void_p()
    mask = (ctypes.c_uint32 * len(words))(*words)
    with torch.cuda.device(device):
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(stream), ctypes.c_uint32(len(words)), mask)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed with {rc}")
    owner = MaskedStream(hip, stream.value, device)
    owner.stream.owner = owner
    return owner.stream
