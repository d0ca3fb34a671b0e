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
    """Double-buffered asynchronous all-gather of equal-size output shards.

    ``submit(y_local)`` starts the gather of one batch on the backend's communication stream (RCCL: its own HIP
    stream, ordered after the kernels that produced ``y_local``) and returns a slot; the caller goes on to compute
    the next batch, so at N > 1 the collective (100 MB per GPU and batch at 32 x 3 x 512 x 512 fp32) overlaps the
    next forward instead of adding to it.  ``result(slot)`` waits for that gather and returns the gathered tensor
    (valid until the slot is reused, ``depth`` submits later); ``drain()`` waits for everything in flight.
    """

    def __init__(self, shard_shape, dtype, device, group=None, depth: int = 2):
        if not dist.is_initialized():
            raise RuntimeError("OutputGather needs an initialised process group")
        self.group = group
        self.world = dist.get_world_size(group)
        shard_shape = tuple(shard_shape)
        self.bufs = [torch.empty((self.world * shard_shape[0],) + shard_shape[1:], dtype=dtype, device=device)
                     for _ in range(max(1, depth))]
        self.works = [None] * len(self.bufs)
        self.keep = [None] * len(self.bufs)          # the local shard must stay alive until its gather completed
        self.next = 0

    def submit(self, y_local: torch.Tensor) -> int:
        slot = self.next
        self.next = (slot + 1) % len(self.bufs)
        if self.works[slot] is not None:             # the buffer is about to be overwritten
            self.works[slot].wait()
        y_local = y_local.contiguous()
        self.keep[slot] = y_local
        self.works[slot] = dist.all_gather_into_tensor(self.bufs[slot], y_local, group=self.group, async_op=True)
        return slot

    def result(self, slot: int) -> torch.Tensor:
        if self.works[slot] is not None:
            self.works[slot].wait()
            self.works[slot] = None
            self.keep[slot] = None
        return self.bufs[slot]

    def drain(self) -> None:
        for slot in range(len(self.bufs)):
            self.result(slot)


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
