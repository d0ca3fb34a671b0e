"""N > 1 path on CPU: two processes, gloo backend, batch sharding + all-gather of the output shards
(the only exchange step of the path).  The per-rank compute is the oracle here (no GPU in this
suite); on the GPU box the same helpers wrap Generator.forward with the nccl (= RCCL) backend."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, res, out_dir):
    sys.path.insert(0, ROOT)
    import importlib
    pkg = importlib.import_module("mi-gan_amd")
    from oracle import migan_torch_cpu as torc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    sd = pkg.synth.make_state_dict(res, seed=3)
    x = torch.from_numpy(pkg.synth.make_input(total, res, seed=3))
    fwd = lambda t: torc.generator(t, sd, res)
    y_all = pkg.distributed.sharded_forward(fwd, x)
    lo, hi = pkg.distributed.shard_range(total, rank, world)
    np.save(os.path.join(out_dir, f"y_{rank}.npy"), y_all.numpy())
    np.save(os.path.join(out_dir, f"range_{rank}.npy"), np.array([lo, hi]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [4, 5])          # even and ragged shards
def test_sharded_forward_world2_gloo(pkg, tmp_path, total):
    importlib = __import__("importlib")
    importlib.import_module("mi-gan_amd.distributed")
    from oracle import migan_torch_cpu as torc
    res, world = 8, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total, res, str(tmp_path)), nprocs=world, join=True)
    sd = pkg.synth.make_state_dict(res, seed=3)
    x = pkg.synth.make_input(total, res, seed=3)
    want = torc.generator(x, sd, res).numpy()
    ys = [np.load(tmp_path / f"y_{r}.npy") for r in range(world)]
    ranges = [tuple(np.load(tmp_path / f"range_{r}.npy")) for r in range(world)]
    assert ranges[0][0] == 0 and ranges[-1][1] == total and ranges[0][1] == ranges[1][0]
    for y in ys:                                    # every rank holds the full gathered batch
        assert y.shape == want.shape
        np.testing.assert_allclose(y, want, rtol=0, atol=1e-5)
    np.testing.assert_array_equal(ys[0], ys[1])


def test_shard_range_partitions(pkg):
    import importlib
    d = importlib.import_module("mi-gan_amd.distributed")
    for total in (1, 7, 32, 256):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                lo, hi = d.shard_range(total, r, world)
                cover += list(range(lo, hi))
                assert 0 <= hi - lo <= total // world + 1
            assert cover == list(range(total))
    with pytest.raises(ValueError):
        d.shard_range(4, 2, 2)


def _pipe_worker(rank, world, port, steps, out_dir):
    sys.path.insert(0, ROOT)
    import importlib
    pkg = importlib.import_module("mi-gan_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shard = (3, 3, 8, 8)
    pipe = pkg.distributed.OutputGather(shard, torch.float32, torch.device("cpu"))
    got = []
    slots = []
    for i in range(steps):                              # the bench.py loop: submit every step, never wait in the loop
        y = torch.full(shard, float(100 * i + rank))
        y[:, 0, 0, 0] = torch.arange(3, dtype=torch.float32) + 10 * rank
        slots.append(pipe.submit(y))
        if i >= 1:                                      # a consumer one step behind sees the previous batch intact
            got.append(pipe.result(slots[i - 1]).clone())
    pipe.drain()
    got.append(pipe.result(slots[-1]).clone())
    np.save(os.path.join(out_dir, f"pipe_{rank}.npy"), torch.stack(got).numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_output_gather_pipeline_world2_gloo(tmp_path):
    """bench.py's N > 1 step: asynchronous, double-buffered all-gather of every step's output shards."""
    world, steps = 2, 5
    mp.spawn(_pipe_worker, args=(world, _free_port(), steps, str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(tmp_path / "pipe_0.npy"), np.load(tmp_path / "pipe_1.npy")
    np.testing.assert_array_equal(a, b)                 # every rank holds the same gathered batches
    assert a.shape == (steps, world * 3, 3, 8, 8)
    for i in range(steps):
        for r in range(world):
            blk = a[i, 3 * r:3 * r + 3]
            assert np.all(blk[:, 1:] == 100 * i + r)    # rank r's shard of step i, in rank order
            np.testing.assert_array_equal(blk[:, 0, 0, 0], np.arange(3) + 10 * r)


class _StandInModel:
    """what OutputGather.forward_and_submit needs of mi-gan_amd's Generator, on CPU: forward(x, out=) and forward_parts(x, outs, streams) write
    a function of x into the buffers they are given and report the sub-batch sizes (the HIP module does the same through migan_forward_parts)"""

    def __init__(self, sizes):
        self.sizes = list(sizes)
        self.calls = []

    @staticmethod
    def f(x):
        return x * 2.0 + 1.0

    def __call__(self, x, out=None):
        assert out is not None and out.is_contiguous() and out.shape == x.shape
        out.copy_(self.f(x))
        self.calls.append(("forward", out.data_ptr()))
        return out

    def forward_parts(self, x, outs, streams):
        lo = 0
        for o, n in zip(outs, self.sizes):
            assert o.is_contiguous() and o.shape[0] == n
            o.copy_(self.f(x[lo:lo + n]))
            lo += n
        self.calls.append(("parts", tuple(o.data_ptr() for o in outs)))
        return list(self.sizes)


def _inplace_worker(rank, world, port, steps, chunks, out_dir):
    sys.path.insert(0, ROOT)
    import importlib
    pkg = importlib.import_module("mi-gan_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = sum(chunks)
    shard = (n, 3, 4, 4)
    pipe = pkg.distributed.OutputGather(shard, torch.float32, torch.device("cpu"), chunks=chunks)
    model = _StandInModel(chunks)
    got, slots = [], []
    for i in range(steps):
        x = torch.arange(n * 3 * 4 * 4, dtype=torch.float32).reshape(shard) + 1000.0 * i + 100000.0 * rank
        slots.append(pipe.forward_and_submit(model, x))
        # in place: the forward wrote into this rank's slices of the receive buffers of that slot, nowhere else
        want_ptrs = tuple(v.data_ptr() for v in pipe.shards(slots[-1]))
        assert model.calls[-1][1] == (want_ptrs if len(chunks) > 1 else want_ptrs[0])
        if i >= 1:
            got.append(pipe.result(slots[i - 1]).clone())
    pipe.drain()
    got.append(pipe.result(slots[-1]).clone())
    parts = pipe.result_chunks(slots[-1])
    assert [tuple(p.shape[:2]) for p in parts] == [(world, c) for c in chunks]
    np.save(os.path.join(out_dir, f"inplace_{rank}.npy"), torch.stack(got).numpy())
    # the copying form on the same object (producers that cannot write in place), chunked the same way
    y = torch.full(shard, float(rank))
    np.save(os.path.join(out_dir, f"copy_{rank}.npy"), pipe.result(pipe.submit(y)).clone().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("chunks", [[3], [2, 1], [2, 2]])
def test_output_gather_in_place_and_per_sub_batch_world2_gloo(tmp_path, chunks):
    """the forward writes its images into the collective's receive buffers (no local copy) and every sub-batch's
    shard is gathered by its own collective, enqueued right behind that sub-batch; the gathered batch is in rank order whatever the
    chunking, step after step, with the double buffering of bench.py's loop."""
    world, steps = 2, 4
    mp.spawn(_inplace_worker, args=(world, _free_port(), steps, chunks, str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(tmp_path / "inplace_0.npy"), np.load(tmp_path / "inplace_1.npy")
    np.testing.assert_array_equal(a, b)
    n = sum(chunks)
    assert a.shape == (steps, world * n, 3, 4, 4)
    base = np.arange(n * 3 * 4 * 4, dtype=np.float32).reshape(n, 3, 4, 4)
    for i in range(steps):
        for r in range(world):
            np.testing.assert_array_equal(a[i, r * n:(r + 1) * n], (base + 1000.0 * i + 100000.0 * r) * 2.0 + 1.0)
    c = np.load(tmp_path / "copy_0.npy")
    assert c.shape == (world * n, 3, 4, 4) and all(np.all(c[r * n:(r + 1) * n] == r) for r in range(world))


def _worker_comodgan(rank, world, port, total, out_dir):
    sys.path.insert(0, ROOT)
    import importlib
    pkg = importlib.import_module("mi-gan_amd")
    from oracle import comodgan_oracle as corc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    cs = pkg.comodgan_schema
    cfg = cs.Config(resolution=16, ch_base=1024, ch_max=64, num_ws=cs.default_num_ws(16))
    sd = pkg.synth.make_comodgan_state_dict(cfg, 1)
    x = torch.from_numpy(pkg.synth.make_input(total, 16, seed=2))
    z = torch.from_numpy(pkg.synth.make_latent(total, 512, seed=2))
    fwd = lambda xs, zs: torch.from_numpy(corc.generator(xs.numpy(), zs.numpy(), sd, 16, cfg.num_ws))
    y_all = pkg.distributed.sharded_forward(fwd, (x, z))          # image and latent sliced alike
    np.save(os.path.join(out_dir, f"cm_{rank}.npy"), y_all.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_comodgan_forward_world2_gloo(pkg, tmp_path):
    """Co-Mod-GAN takes (x, z) per image: both are sharded; the gathered output equals the whole-batch output up to the
    batch-wide style normalisation's epsilon (stylegan.py:139,147)."""
    from oracle import comodgan_oracle as corc
    cs = pkg.comodgan_schema
    total, world = 3, 2
    mp.spawn(_worker_comodgan, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    cfg = cs.Config(resolution=16, ch_base=1024, ch_max=64, num_ws=cs.default_num_ws(16))
    sd = pkg.synth.make_comodgan_state_dict(cfg, 1)
    want = corc.generator(pkg.synth.make_input(total, 16, seed=2), pkg.synth.make_latent(total, 512, seed=2), sd, 16, cfg.num_ws)
    for r in range(world):
        got = np.load(tmp_path / f"cm_{r}.npy")
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 1e-4 * max(1.0, np.abs(want).max())


def _worker_u8(rank, world, port, total, res, out_dir):
    sys.path.insert(0, ROOT)
    import importlib
    pkg = importlib.import_module("mi-gan_amd")
    from oracle import migan_prepost as pp
    from oracle import migan_torch_cpu as torc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    sd = pkg.synth.make_state_dict(res, seed=4)
    img, mask = pkg.synth.make_uint8_input(total, res, seed=4)

    def fwd(img_t, mask_t):                         # stands in for Generator.forward_uint8 (demo.py:56-66, forward, :135-140)
        i, m = img_t.numpy(), mask_t.numpy()
        return torch.from_numpy(pp.compose(torc.generator(pp.preprocess(i, m), sd, res).numpy(), i, m))

    out = pkg.distributed.sharded_forward(fwd, (torch.from_numpy(img), torch.from_numpy(mask)))
    assert out.dtype == torch.uint8 and tuple(out.shape) == (total, res, res, 3)
    np.save(os.path.join(out_dir, f"u8_{rank}.npy"), out.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [4, 3])
def test_sharded_uint8_forward_world2_gloo(pkg, tmp_path, total):
    """the uint8-in / uint8-out form: image + mask sharded alike, composed uint8 shards gathered (a quarter of the
    fp32 bytes per rank), every rank ends up with the single-process result"""
    from oracle import migan_prepost as pp
    from oracle import migan_torch_cpu as torc
    res, world = 8, 2
    mp.spawn(_worker_u8, args=(world, _free_port(), total, res, str(tmp_path)), nprocs=world, join=True)
    sd = pkg.synth.make_state_dict(res, seed=4)
    img, mask = pkg.synth.make_uint8_input(total, res, seed=4)
    want = pp.compose(torc.generator(pp.preprocess(img, mask), sd, res).numpy(), img, mask)
    for r in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / f"u8_{r}.npy"), want)
