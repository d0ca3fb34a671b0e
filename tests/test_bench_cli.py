"""bench.py's launch plumbing on a CPU box: `--gpus 2` creates its two ranks itself, they rendezvous (gloo here, nccl = RCCL on
the GPU box), gather through the same OutputGather as the real run, and rank 0 prints one JSON line that says how many ranks
took part; a mismatch between --gpus and the ranks is an error."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env=None):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], cwd=ROOT, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, **(env or {})))
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_gpus_2_spawns_two_ranks_and_reports_them():
    r, out = _run("--gpus", "2", "--backend", "gloo", "--dry", "--steps", "3")
    assert r.returncode == 0, r.stderr[-2000:]
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["dry"] is True
    assert out["config"]["global_batch"] == 64 and out["scaling"] == "weak"


def test_n1_line_schema_and_rank_mismatch_is_an_error():
    r, out = _run("--dry", "--steps", "2")
    assert r.returncode == 0 and out["n_gpus"] == 1 and out["rccl_ranks"] == 1
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config"):
        assert key in out
    # launched as ONE rank of a 1-rank group but told --gpus 2: the line carries the error and the exit code is non-zero
    r, out = _run("--gpus", "2", "--dry", "--steps", "1", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "error" in out and out["rccl_ranks"] == 1


def test_gpu_run_is_refused_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        return
    r, out = _run("--steps", "1")
    assert r.returncode != 0 and out is None and "MI355X" in (r.stderr + r.stdout)


def test_uint8_io_gathers_uint8_shards():
    """--io u8 (migan_forward_u8): the shards the ranks exchange are the composed uint8 images, a quarter of the fp32 bytes"""
    r, out = _run("--gpus", "2", "--backend", "gloo", "--dry", "--steps", "2", "--io", "u8")
    assert r.returncode == 0, r.stderr[-2000:]
    assert out["rccl_ranks"] == 2 and out["gather_dtype"] == "uint8"


def test_world_8_dry_run_is_one_line_with_eight_ranks_and_no_single_gpu_legs():
    """the shape of the driver's scaling run, on CPU: eight ranks rendezvous and gather, rank 0 prints exactly one JSON line, and the legs
    that only make sense at N = 1 (secondary workloads, CPU baseline, batch-1 latency) are not there"""
    r, out = _run("--gpus", "8", "--backend", "gloo", "--dry", "--steps", "2")
    assert r.returncode == 0, r.stderr[-2000:]
    assert len([l for l in r.stdout.splitlines() if l.startswith("{")]) == 1
    assert out["n_gpus"] == 8 and out["rccl_ranks"] == 8 and out["config"]["global_batch"] == 256 and out["scaling"] == "weak"
    for key in ("secondary", "cpu_baseline", "latency_b1_ms", "value_exact_f32", "rccl_world1"):
        assert key not in out
    assert out["gather_dtype"] == "float32"


def test_gather_dtype_choices():
    r, out = _run("--gpus", "2", "--backend", "gloo", "--dry", "--steps", "2", "--gather-dtype", "f16")
    assert r.returncode == 0 and out["gather_dtype"] == "float16"
    r32, out32 = _run("--gpus", "2", "--backend", "gloo", "--dry", "--steps", "2")
    assert out["gather_mb_per_rank_per_step"] * 2 == out32["gather_mb_per_rank_per_step"]
    r, out = _run("--gpus", "2", "--backend", "gloo", "--dry", "--steps", "1", "--gather-dtype", "u8")
    assert r.returncode != 0                                     # uint8 shards exist only with --io u8


def test_uint8_source_of_the_synthetic_input_matches_it():
    import importlib
    import numpy as np
    from oracle import migan_prepost as pp
    synth = importlib.import_module("mi-gan_amd").synth
    img, mask = synth.make_uint8_input(3, 32, seed=5)
    assert img.dtype == np.uint8 and img.shape == (3, 32, 32, 3) and set(np.unique(mask)) <= {0, 255}
    np.testing.assert_array_equal(pp.preprocess(img, mask), synth.make_input(3, 32, seed=5))


def test_roofline_arithmetic_on_synthetic_launches():
    """roofline_from_launches: dominant kernel = largest share of GPU time; its binding roof = the one that gives the longer time
    for its algorithmic work; achieved = algorithmic bytes (flops) of its launches / their summed duration; attach_traffic looks
    the kernel up by the symbol rocprofv3 prints and says so when it is missing."""
    import importlib.util
    import json as _json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    batch = 32
    launches = [
        dict(layer="a", kernel="k_stream", flops=1e9, mfma_flops=0.5e9, bytes=30e6),      # per image: 32 x 30 MB = 0.96 GB per launch
        dict(layer="b", kernel="k_stream", flops=1e9, mfma_flops=0.5e9, bytes=30e6),
        dict(layer="c", kernel="k_gemm", flops=8e9, mfma_flops=8e9, bytes=1e6),
    ]
    rounds = [[0.30, 0.30, 0.40], [0.32, 0.28, 0.40], [0.30, 0.30, 0.41]]                # ms per launch, three rounds (median is used)
    r = b.roofline_from_launches(launches, rounds, batch, "f16x2")
    assert r["kernel"] == "k_stream" and r["launches"] == 2 and r["bound"] == "hbm" and r["peak"] == 8000.0
    want = 2 * 32 * 30e6 / 0.60e-3 / 1e9
    assert abs(r["achieved"] - want) < 0.01 and abs(r["frac"] - want / 8000.0) < 1e-4
    assert abs(r["avg_launch_ms"] - 0.30) < 1e-9 and abs(r["share_of_gpu_time"] - 0.6) < 1e-3
    assert r["alg_per_launch"]["bytes"] == 32 * 30e6
    wf = r["whole_forward"]
    assert abs(wf["sum_kernel_ms"] - 1.0) < 1e-9 and abs(wf["hbm_gbs"] - (32 * 61e6) / 1e-3 / 1e9) < 0.1
    # a matrix-bound dominant kernel is priced against the f16x2 ceiling (2500 / 3 algorithmic TFLOP/s)
    r2 = b.roofline_from_launches(launches, [[0.1, 0.1, 0.9]], batch, "f16x2")
    assert r2["kernel"] == "k_gemm" and r2["bound"] == "mfma" and abs(r2["peak"] - 833.3) < 0.1
    assert abs(r2["achieved"] - 32 * 8e9 / 0.9e-3 / 1e12) < 0.01
    # traffic lookup
    b.attach_traffic(r, "profiles/pmc_traffic_latest.json", True)
    assert r["traffic"] is None and "not in" in r["traffic_note"]
    table = _json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")))
    real = next(k for k in table if not k.startswith("_"))
    r["kernel"] = real
    b.attach_traffic(r, "profiles/pmc_traffic_latest.json", True)
    assert r["traffic"] and r["traffic"] > 0
    # the table names the sources it was measured on; the line says whether those are the sources of the library that ran
    assert table["_meta"]["kernel_source_sha"] == r["traffic_measured_on"]
    assert r["traffic_stale"] == (r["traffic_measured_on"] != b.kernel_source_sha())


def test_line_roofline_describes_the_whole_forward():
    """`roofline.frac` on the JSON line is the time-weighted fraction over ALL launches (sum of algorithmic bytes /
    sum of launch durations), with the fraction of the guide's achievable 6.3 TB/s beside it; the dominant kernel symbol is a sub-object
    that carries its share of the GPU time."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    batch = 32
    launches = [dict(layer="a", kernel="k_stream", flops=1e9, mfma_flops=0.5e9, bytes=30e6),
                dict(layer="b", kernel="k_stream", flops=1e9, mfma_flops=0.5e9, bytes=30e6),
                dict(layer="c", kernel="k_small", flops=1e8, mfma_flops=1e8, bytes=1e6)]
    dom = b.roofline_from_launches(launches, [[0.30, 0.30, 0.40]], batch, "f16x2")
    r = b.forward_roofline(dom, ms_per_step=0.95)
    b.finish_traffic(r, dom)
    whole = 32 * 61e6 / 1.0e-3 / 1e9
    assert r["bound"] == "hbm" and abs(r["achieved"] - whole) < 0.1 and abs(r["frac"] - whole / 8000.0) < 1e-4
    assert abs(r["frac_of_achievable"] - whole / 6300.0) < 1e-4 and r["achievable_peak"] == 6300.0
    assert abs(r["frac_of_timed_step"] - 32 * 61e6 / 0.95e-3 / 8e12) < 1e-4
    d = r["dominant"]
    assert d["kernel"] == "k_stream" and abs(d["share_of_gpu_time"] - 0.6) < 1e-3 and d["frac"] > r["frac"]
    assert "kernel" not in r and r["traffic"] is None and "per_kernel" in r and r["whole_forward"]["ms_per_step"] == 0.95
    # a matrix-bound forward is priced against the GEMM variant's ceiling
    gemm = [dict(layer="g", kernel="k_gemm", flops=8e9, mfma_flops=8e9, bytes=1e6)]
    r2 = b.forward_roofline(b.roofline_from_launches(gemm, [[0.9]], batch, "f16x2"), 0.9)
    assert r2["bound"] == "mfma" and abs(r2["peak"] - 833.3) < 0.1 and abs(r2["frac"] - 32 * 8e9 / 0.9e-3 / 1e12 / 833.3) < 1e-3
