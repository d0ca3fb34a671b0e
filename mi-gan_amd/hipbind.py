"""ctypes binding of the C ABI in include/migan_hip.h (libmigan_hip.so).

This is the binding a maintainer of the reference would add next to
``lib/model_zoo/migan_inference.py``.  It deals in raw
addresses and sizes only; torch appears nowhere in this file.  There is no
fallback: if the shared library is missing or fails to load, ``load_library``
raises ``MiganError`` naming the build command.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBNAME = "libmigan_hip.so"

MIGAN_OK, MIGAN_EINVAL, MIGAN_ESTATE, MIGAN_ERUNTIME, MIGAN_EUNSUPPORTED = 0, 1, 2, 3, 4
# activation storage formats (MIGAN_DTYPE_*) and GEMM variants (MIGAN_GEMM_*) of include/migan_hip.h
DTYPES = {"f32": 0, "fp32": 0, "float32": 0, "bf16": 1, "bfloat16": 1, "f16": 2, "fp16": 2, "float16": 2}
DTYPE_NAMES = {0: "f32", 1: "bf16", 2: "f16"}
GEMMS = {"default": -1, "f32": 0, "bf16x3": 1, "f16x2": 2, "f16": 3}
GEMM_NAMES = {0: "f32", 1: "bf16x3", 2: "f16x2", 3: "f16"}


def dtype_code(dtype) -> int:
    """'f32' | 'bf16' | 'f16' | a MIGAN_DTYPE_* int | anything whose str() ends in one of those names (torch.bfloat16)"""
    if isinstance(dtype, int):
        if dtype not in DTYPE_NAMES:
            raise ValueError(f"unknown activation dtype code {dtype}")
        return dtype
    key = str(dtype).split(".")[-1].lower()
    if key not in DTYPES:
        raise ValueError(f"activation dtype must be one of f32 / bf16 / f16, got {dtype!r}")
    return DTYPES[key]


class MiganError(RuntimeError):
    def __init__(self, msg: str, code: int = MIGAN_ERUNTIME):
        super().__init__(msg)
        self.code = code


def library_path() -> str:
    """the in-tree libmigan_hip.so; MIGAN_HIP_LIBRARY may name ANOTHER BUILD OF THE SAME LIBRARY (measurement builds for same-box
    A/B runs: scripts/phase_profile.py) -- whatever is loaded must report the gfx950 backend, see MiganLib"""
    return os.environ.get("MIGAN_HIP_LIBRARY", os.path.join(_HERE, "csrc", _LIBNAME))


PRODUCT_BACKEND = "hip:gfx950"


class SepConvDesc(C.Structure):
    """struct migan_sepconv_desc"""
    _fields_ = [(n, C.c_void_p) for n in (
        "x", "y", "skip", "conv1_weight", "conv1_bias", "conv2_weight", "noise_const", "noise_strength",
        "fromrgb_weight", "fromrgb_bias", "torgb_weight", "torgb_bias", "img_prev", "img_out")] + [
        (n, C.c_int) for n in ("batch", "cin", "cout", "res_in", "down", "up")] + [
        ("scratch", C.c_void_p), ("scratch_bytes", C.c_size_t), ("wsplit", C.c_void_p), ("wsplit_bytes", C.c_size_t)] + [
        (n, C.c_int) for n in ("gemm", "dtype", "width_in")]


EXPORTS = (
    "migan_create", "migan_destroy", "migan_set_gemm", "migan_get_gemm", "migan_assume_static_weights", "migan_set_streams",
    "migan_num_weights", "migan_weight_info", "migan_set_weight",
    "migan_commit", "migan_workspace_bytes", "migan_forward", "migan_workspace_bytes_hw", "migan_forward_hw", "migan_forward_u8",
    "migan_num_launches", "migan_launch_info",
    "migan_forward_timed", "migan_set_debug", "migan_debug_tensor", "migan_sepconv_forward",
    "migan_pack_input", "migan_compose_output",
    "migan_pipeline_mask_resize", "migan_pipeline_scratch_bytes", "migan_pipeline_bbox", "migan_pipeline_pre", "migan_pipeline_post",
    "migan_forward_split", "migan_forward_parts", "migan_set_tuning", "migan_last_error", "migan_last_kernel", "migan_nan_policy", "migan_backend", "migan_gemm_variant", "migan_version",
    # include/comodgan_hip.h
    "comodgan_create", "comodgan_destroy", "comodgan_num_weights", "comodgan_weight_info", "comodgan_set_weight",
    "comodgan_commit", "comodgan_workspace_bytes", "comodgan_assume_static_weights", "comodgan_noise_floats", "comodgan_forward",
    "comodgan_num_launches",
    "comodgan_launch_info", "comodgan_forward_timed", "comodgan_set_debug", "comodgan_debug_tensor", "comodgan_set_truncation_cutoff",
)


class CoModGANConfig(C.Structure):
    """struct comodgan_config"""
    _fields_ = [(n, C.c_int) for n in ("resolution", "ch_base", "ch_max", "z_dim", "w_dim", "w0_dim", "map_layers", "num_ws")]


NOISE_MODES = {"none": 0, "const": 1, "random": 2}


class MiganLib:
    """Typed view of one loaded libmigan_hip.so."""

    def __init__(self, path: Optional[str] = None, allow_test_backend: bool = False):
        """allow_test_backend: only the CPU test-suite passes True, with an explicit path, to drive the same host code + kernel
        source through the fiber emulator (tests/emu); the package itself never does, so no environment variable can make the
        product run on anything but the HIP library."""
        self.path = path or library_path()
        self.stamp_missing = False
        if not os.path.exists(self.path):
            raise MiganError(
                f"{self.path} not found: the MI355X HIP extension is not built. "
                f"Run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
                f"There is no CPU/PyTorch fallback for this path.")
        # the in-tree product library must be the product build: build.py stamps it with a digest of its compile flags, and a measurement
        # build (-DMIGAN_PHASE_PROF) left under this name would otherwise be picked up silently
        if os.path.abspath(self.path) == os.path.abspath(os.path.join(_HERE, "csrc", _LIBNAME)):
            stamp = self.path + ".flags"
            if not os.path.exists(stamp):
                # build.py deletes the stamp exactly when the ISA lint has not passed (the packed-fp32 hazard): a library without
                # one is refused, not warned about -- hand builds opt out explicitly or go under another name
                if os.environ.get("MIGAN_ALLOW_UNSTAMPED") != "1":
                    raise MiganError(f"{self.path} has no build stamp ({stamp}): it was not produced by mi-gan_amd/build.py, or its ISA lint never "
                                     f"passed; rebuild with `python -c 'import __graft_entry__ as g; g.build()'` (MIGAN_ALLOW_UNSTAMPED=1 loads "
                                     f"a hand build anyway; measurement builds belong under another name via MIGAN_HIP_LIBRARY)")
                self.stamp_missing = True
            else:
                from . import build as _build
                if open(stamp).read().strip() != _build.flags_digest(()):
                    raise MiganError(f"{self.path} was not built with the product flags (stamp {stamp} differs): rebuild with "
                                     f"`python -c 'import __graft_entry__ as g; g.build()'`; measurement builds go under another name")
        try:
            self.lib = C.CDLL(self.path)
        except OSError as e:  # pragma: no cover - depends on the machine
            raise MiganError(f"cannot load {self.path}: {e}") from e
        L = self.lib
        for name in EXPORTS:
            if not hasattr(L, name):
                raise MiganError(f"{self.path} does not export {name}")
        vp, ci = C.c_void_p, C.c_int
        L.migan_create.argtypes = [ci, ci, ci, C.POINTER(vp)]
        L.migan_destroy.argtypes = [vp]
        L.migan_set_gemm.argtypes = [vp, ci]
        L.migan_get_gemm.argtypes = [vp, C.POINTER(ci)]
        L.migan_assume_static_weights.argtypes = [vp, ci]
        L.migan_set_streams.argtypes = [vp, ci]
        L.migan_workspace_bytes_hw.argtypes = [vp, ci, ci, ci, C.POINTER(C.c_size_t)]
        L.migan_forward_hw.argtypes = [vp, vp, vp, ci, ci, ci, vp, C.c_size_t, vp]
        L.migan_forward_u8.argtypes = [vp, vp, vp, vp, ci, vp, C.c_size_t, vp]
        L.migan_forward_split.argtypes = [vp, ci, C.POINTER(ci), C.POINTER(ci)]
        L.migan_forward_parts.argtypes = [vp, vp, C.POINTER(vp), ci, vp, C.c_size_t, vp, C.POINTER(vp), ci, C.POINTER(ci), C.POINTER(ci)]
        L.migan_pipeline_mask_resize.argtypes = [vp, ci, ci, vp, ci, ci, vp]
        L.migan_pipeline_scratch_bytes.argtypes = [ci, ci, C.POINTER(C.c_size_t)]
        L.migan_pipeline_bbox.argtypes = [vp, ci, ci, ci, ci, vp, C.POINTER(ci), vp]
        L.migan_pipeline_pre.argtypes = [vp, vp, ci, ci, C.POINTER(ci), ci, vp, vp]
        L.migan_pipeline_post.argtypes = [vp, vp, ci, ci, C.POINTER(ci), ci, vp, C.POINTER(C.c_float), vp, vp]
        L.migan_num_weights.argtypes = [vp, C.POINTER(ci)]
        L.migan_weight_info.argtypes = [vp, ci, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(ci), C.POINTER(ci)]
        L.migan_set_weight.argtypes = [vp, C.c_char_p, vp, C.POINTER(C.c_int64), ci]
        L.migan_commit.argtypes = [vp, vp]
        L.migan_workspace_bytes.argtypes = [vp, ci, C.POINTER(C.c_size_t)]
        L.migan_forward.argtypes = [vp, vp, vp, ci, vp, C.c_size_t, vp]
        L.migan_num_launches.argtypes = [vp, C.POINTER(ci)]
        L.migan_launch_info.argtypes = [vp, ci, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_double),
                                        C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(ci)]
        L.migan_forward_timed.argtypes = [vp, vp, vp, ci, vp, C.c_size_t, vp, C.POINTER(C.c_float), ci]
        L.migan_set_debug.argtypes = [vp, ci]
        L.migan_debug_tensor.argtypes = [vp, ci, C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_int64)]
        L.migan_sepconv_forward.argtypes = [C.POINTER(SepConvDesc), vp]
        L.migan_set_tuning.argtypes = [C.c_char_p, ci]
        L.migan_pack_input.argtypes = [vp, vp, vp, ci, ci, vp]
        L.migan_compose_output.argtypes = [vp, vp, vp, vp, ci, ci, vp]
        fp = C.POINTER(C.c_float)
        L.comodgan_create.argtypes = [C.POINTER(CoModGANConfig), ci, C.POINTER(vp)]
        L.comodgan_destroy.argtypes = [vp]
        L.comodgan_num_weights.argtypes = [vp, C.POINTER(ci)]
        L.comodgan_weight_info.argtypes = [vp, ci, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(ci), C.POINTER(ci)]
        L.comodgan_set_weight.argtypes = [vp, C.c_char_p, vp, C.POINTER(C.c_int64), ci]
        L.comodgan_commit.argtypes = [vp, vp]
        L.comodgan_workspace_bytes.argtypes = [vp, ci, C.POINTER(C.c_size_t)]
        L.comodgan_noise_floats.argtypes = [vp, C.POINTER(C.c_size_t)]
        L.comodgan_assume_static_weights.argtypes = [vp, ci]
        L.comodgan_forward.argtypes = [vp, vp, vp, vp, ci, C.c_float, ci, vp, vp, C.c_size_t, vp]
        L.comodgan_forward_timed.argtypes = [vp, vp, vp, vp, ci, C.c_float, ci, vp, vp, C.c_size_t, vp, fp, ci]
        L.comodgan_num_launches.argtypes = [vp, C.POINTER(ci)]
        L.comodgan_launch_info.argtypes = [vp, ci, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_double),
                                           C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.comodgan_set_debug.argtypes = [vp, ci]
        L.comodgan_set_truncation_cutoff.argtypes = [vp, ci]
        L.comodgan_debug_tensor.argtypes = [vp, ci, C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_int64), C.POINTER(ci)]
        L.migan_last_error.restype = C.c_char_p
        L.migan_last_kernel.restype = C.c_char_p
        L.migan_nan_policy.restype = C.c_char_p
        L.migan_backend.restype = C.c_char_p
        L.migan_gemm_variant.restype = C.c_char_p
        L.migan_backend.restype = C.c_char_p
        if not allow_test_backend and L.migan_backend().decode() != PRODUCT_BACKEND:
            raise MiganError(f"{self.path} reports backend {L.migan_backend().decode()!r}, not {PRODUCT_BACKEND!r}: only the gfx950 HIP "
                             f"library is a product backend (the CPU emulator build is test infrastructure)")
        for name in EXPORTS:
            if name not in ("migan_last_error", "migan_last_kernel", "migan_nan_policy", "migan_backend", "migan_gemm_variant"):
                getattr(L, name).restype = ci

    # -- error mapping: EINVAL -> ValueError-like, like the reference's constructor / load_state_dict
    def check(self, rc: int) -> None:
        if rc == MIGAN_OK:
            return
        msg = (self.lib.migan_last_error() or b"").decode()
        if rc == MIGAN_EINVAL:
            err: Exception = ValueError(msg)
        elif rc == MIGAN_EUNSUPPORTED:
            err = NotImplementedError(msg)
        else:
            err = MiganError(msg, rc)
        raise err

    def backend(self) -> str:
        return self.lib.migan_backend().decode()

    def gemm_variant(self) -> str:
        return self.lib.migan_gemm_variant().decode()

    def nan_policy(self) -> str:
        """"propagate" (default build: Tensor.clamp's behaviour, reference :21-23) or "clamp" (libmigan_hip_nanclamp.so): what lrelu_agc's clamp does with a NaN"""
        return self.lib.migan_nan_policy().decode()

    def last_kernel(self) -> str:
        """symbol of the fused-SeparableConv2d kernel this thread launched last"""
        return (self.lib.migan_last_kernel() or b"").decode()

    def set_tuning(self, key: str, value: int) -> None:
        self.check(self.lib.migan_set_tuning(key.encode(), int(value)))

    def pack_input(self, img_ptr: int, mask_ptr: int, x_ptr: int, batch: int, resolution: int, stream: int = 0) -> None:
        self.check(self.lib.migan_pack_input(C.c_void_p(img_ptr), C.c_void_p(mask_ptr), C.c_void_p(x_ptr), int(batch),
                                             int(resolution), C.c_void_p(stream)))

    def compose_output(self, y_ptr: int, img_ptr: int, mask_ptr: int, out_ptr: int, batch: int, resolution: int,
                       stream: int = 0) -> None:
        self.check(self.lib.migan_compose_output(C.c_void_p(y_ptr), C.c_void_p(img_ptr), C.c_void_p(mask_ptr),
                                                 C.c_void_p(out_ptr), int(batch), int(resolution), C.c_void_p(stream)))

    # the deployed pipeline (include/migan_hip.h, reference scripts/create_onnx_pipeline.py:118-264)
    def pipeline_mask_resize(self, mask_ptr: int, mask_height: int, mask_width: int, out_ptr: int, height: int, width: int, stream: int = 0) -> None:
        self.check(self.lib.migan_pipeline_mask_resize(C.c_void_p(mask_ptr), int(mask_height), int(mask_width), C.c_void_p(out_ptr),
                                                       int(height), int(width), C.c_void_p(stream)))

    def pipeline_scratch_bytes(self, height: int, width: int) -> int:
        n = C.c_size_t()
        self.check(self.lib.migan_pipeline_scratch_bytes(int(height), int(width), C.byref(n)))
        return int(n.value)

    def pipeline_bbox(self, mask_ptr: int, height: int, width: int, resolution: int, padding: int, scratch_ptr: int, stream: int = 0):
        box = (C.c_int * 4)()
        self.check(self.lib.migan_pipeline_bbox(C.c_void_p(mask_ptr), int(height), int(width), int(resolution), int(padding),
                                                C.c_void_p(scratch_ptr), box, C.c_void_p(stream)))
        return tuple(int(v) for v in box)

    def pipeline_pre(self, image_ptr: int, mask_ptr: int, height: int, width: int, bbox, resolution: int, x_ptr: int, stream: int = 0) -> None:
        box = (C.c_int * 4)(*[int(v) for v in bbox])
        self.check(self.lib.migan_pipeline_pre(C.c_void_p(image_ptr), C.c_void_p(mask_ptr), int(height), int(width), box, int(resolution),
                                               C.c_void_p(x_ptr), C.c_void_p(stream)))

    def pipeline_post(self, image_ptr: int, mask_ptr: int, height: int, width: int, bbox, resolution: int, y_ptr: int, scratch_ptr: int,
                      gauss25=None, stream: int = 0) -> None:
        box = (C.c_int * 4)(*[int(v) for v in bbox])
        g = None if gauss25 is None else (C.c_float * 25)(*[float(v) for v in gauss25])
        self.check(self.lib.migan_pipeline_post(C.c_void_p(image_ptr), C.c_void_p(mask_ptr), int(height), int(width), box, int(resolution),
                                                C.c_void_p(y_ptr), g, C.c_void_p(scratch_ptr), C.c_void_p(stream)))

    def sepconv_forward(self, stream: int = 0, **kw) -> None:
        d = SepConvDesc()
        for f, _ in SepConvDesc._fields_:
            setattr(d, f, kw.pop(f, None if f not in ("batch", "cin", "cout", "res_in", "down", "up", "scratch_bytes", "wsplit_bytes",
                                                      "gemm", "dtype", "width_in") else (-1 if f == "gemm" else 0)))
        if kw:
            raise TypeError(f"unknown sepconv fields: {sorted(kw)}")
        d.down = d.down or 1
        d.up = d.up or 1
        self.check(self.lib.migan_sepconv_forward(C.byref(d), C.c_void_p(stream)))


class MiganHandle:
    """RAII wrapper of ``migan_handle*`` (one Generator(resolution) instance)."""

    def __init__(self, lib: MiganLib, resolution: int, device: int = 0, dtype=0):
        self.lib = lib
        self._h = C.c_void_p()
        self.dtype = dtype_code(dtype)
        lib.check(lib.lib.migan_create(int(resolution), self.dtype, int(device), C.byref(self._h)))
        self.resolution = int(resolution)

    # -- per-handle options
    def set_gemm(self, variant) -> None:
        code = GEMMS[variant] if isinstance(variant, str) else int(variant)
        if code < 0:
            # MIGAN_GEMM_DEFAULT: what migan_create picks for this storage format (f16x2 for fp32 storage unless MIGAN_GEMM says
            # otherwise, f16 for 16-bit storage)
            code = GEMMS[self.lib.lib.migan_gemm_variant().decode()] if self.dtype == 0 else GEMMS["f16"]
        self.lib.check(self.lib.lib.migan_set_gemm(self._h, code))

    def gemm(self) -> str:
        v = C.c_int()
        self.lib.check(self.lib.lib.migan_get_gemm(self._h, C.byref(v)))
        return GEMM_NAMES[v.value]

    def assume_static_weights(self, on: bool) -> None:
        self.lib.check(self.lib.lib.migan_assume_static_weights(self._h, 1 if on else 0))

    def set_streams(self, n: int) -> None:
        self.lib.check(self.lib.lib.migan_set_streams(self._h, int(n)))

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.lib.migan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def weights(self) -> List[Tuple[str, Tuple[int, ...], bool]]:
        n = C.c_int()
        self.lib.check(self.lib.lib.migan_num_weights(self._h, C.byref(n)))
        out = []
        for i in range(n.value):
            name = C.c_char_p()
            shape = (C.c_int64 * 4)()
            nd, isb = C.c_int(), C.c_int()
            self.lib.check(self.lib.lib.migan_weight_info(self._h, i, C.byref(name), shape, C.byref(nd), C.byref(isb)))
            out.append((name.value.decode(), tuple(int(shape[k]) for k in range(nd.value)), bool(isb.value)))
        return out

    def set_weight(self, name: str, ptr: int, shape: Sequence[int]) -> None:
        arr = (C.c_int64 * max(1, len(shape)))(*[int(s) for s in shape])
        self.lib.check(self.lib.lib.migan_set_weight(self._h, name.encode(), C.c_void_p(ptr), arr, len(shape)))

    def commit(self, stream: int = 0) -> None:
        self.lib.check(self.lib.lib.migan_commit(self._h, C.c_void_p(stream)))

    def workspace_bytes(self, batch: int) -> int:
        n = C.c_size_t()
        self.lib.check(self.lib.lib.migan_workspace_bytes(self._h, int(batch), C.byref(n)))
        return int(n.value)

    def forward(self, x_ptr: int, y_ptr: int, batch: int, ws_ptr: int, ws_bytes: int, stream: int = 0) -> None:
        self.lib.check(self.lib.lib.migan_forward(self._h, C.c_void_p(x_ptr), C.c_void_p(y_ptr), int(batch),
                                                  C.c_void_p(ws_ptr), C.c_size_t(ws_bytes), C.c_void_p(stream)))

    def forward_split(self, batch: int) -> List[int]:
        """images per sub-batch of a forward of `batch` images (migan_forward_split)"""
        n, parts = (C.c_int * 4)(), C.c_int()
        self.lib.check(self.lib.lib.migan_forward_split(self._h, int(batch), n, C.byref(parts)))
        return [int(n[k]) for k in range(parts.value)]

    def forward_parts(self, x_ptr: int, y_ptrs: List[int], batch: int, ws_ptr: int, ws_bytes: int, stream: int, part_streams: List[int]) -> List[int]:
        """migan_forward_parts: sub-batch k -> y_ptrs[k]; sub-batch k >= 1 on the caller's stream part_streams[k - 1]; streams are not joined"""
        ys = (C.c_void_p * 4)(*([C.c_void_p(p) for p in y_ptrs] + [C.c_void_p(0)] * (4 - len(y_ptrs))))
        st = (C.c_void_p * 4)(*([C.c_void_p(p) for p in part_streams] + [C.c_void_p(0)] * (4 - len(part_streams))))
        n, parts = (C.c_int * 4)(), C.c_int()
        self.lib.check(self.lib.lib.migan_forward_parts(self._h, C.c_void_p(x_ptr), ys, int(batch), C.c_void_p(ws_ptr), C.c_size_t(ws_bytes),
                                                        C.c_void_p(stream), st, len(part_streams), n, C.byref(parts)))
        return [int(n[k]) for k in range(parts.value)]

    def workspace_bytes_hw(self, batch: int, height: int, width: int) -> int:
        n = C.c_size_t()
        self.lib.check(self.lib.lib.migan_workspace_bytes_hw(self._h, int(batch), int(height), int(width), C.byref(n)))
        return int(n.value)

    def forward_hw(self, x_ptr: int, y_ptr: int, batch: int, height: int, width: int, ws_ptr: int, ws_bytes: int, stream: int = 0) -> None:
        self.lib.check(self.lib.lib.migan_forward_hw(self._h, C.c_void_p(x_ptr), C.c_void_p(y_ptr), int(batch), int(height), int(width),
                                                     C.c_void_p(ws_ptr), C.c_size_t(ws_bytes), C.c_void_p(stream)))

    def forward_u8(self, img_ptr: int, mask_ptr: int, out_ptr: int, batch: int, ws_ptr: int, ws_bytes: int, stream: int = 0) -> None:
        self.lib.check(self.lib.lib.migan_forward_u8(self._h, C.c_void_p(img_ptr), C.c_void_p(mask_ptr), C.c_void_p(out_ptr), int(batch),
                                                     C.c_void_p(ws_ptr), C.c_size_t(ws_bytes), C.c_void_p(stream)))

    def forward_timed(self, x_ptr: int, y_ptr: int, batch: int, ws_ptr: int, ws_bytes: int, stream: int = 0) -> List[float]:
        n = len(self.launches())
        ms = (C.c_float * n)()
        self.lib.check(self.lib.lib.migan_forward_timed(self._h, C.c_void_p(x_ptr), C.c_void_p(y_ptr), int(batch),
                                                        C.c_void_p(ws_ptr), C.c_size_t(ws_bytes), C.c_void_p(stream), ms, n))
        return [float(v) for v in ms]

    def launches(self) -> List[Dict]:
        n = C.c_int()
        self.lib.check(self.lib.lib.migan_num_launches(self._h, C.byref(n)))
        out = []
        for i in range(n.value):
            layer, kern = C.c_char_p(), C.c_char_p()
            fl, mf, by = C.c_double(), C.c_double(), C.c_double()
            wg = C.c_int()
            self.lib.check(self.lib.lib.migan_launch_info(self._h, i, C.byref(layer), C.byref(kern), C.byref(fl),
                                                          C.byref(mf), C.byref(by), C.byref(wg)))
            out.append(dict(layer=layer.value.decode(), kernel=kern.value.decode(), flops=fl.value,
                            mfma_flops=mf.value, bytes=by.value, workgroups_batch1=wg.value))
        return out

    def set_debug(self, keep: bool) -> None:
        self.lib.check(self.lib.lib.migan_set_debug(self._h, 1 if keep else 0))

    def debug_tensor(self, batch: int, layer: str) -> Tuple[int, Tuple[int, ...]]:
        off = C.c_size_t()
        shape = (C.c_int64 * 4)()
        self.lib.check(self.lib.lib.migan_debug_tensor(self._h, int(batch), layer.encode(), C.byref(off), shape))
        return int(off.value), tuple(int(s) for s in shape)


class CoModGANHandle:
    """RAII wrapper of ``comodgan_handle*`` (one CoModGANGenerator(mapping, encoder, synthesis) instance)."""

    def __init__(self, lib: MiganLib, resolution: int, num_ws: int, ch_base: int = 32768, ch_max: int = 512, z_dim: int = 512,
                 w_dim: int = 512, w0_dim: int = 1024, map_layers: int = 8, device: int = 0):
        self.lib = lib
        self._h = C.c_void_p()
        self.cfg = CoModGANConfig(int(resolution), int(ch_base), int(ch_max), int(z_dim), int(w_dim), int(w0_dim), int(map_layers),
                                  int(num_ws))
        lib.check(lib.lib.comodgan_create(C.byref(self.cfg), int(device), C.byref(self._h)))

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.lib.comodgan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def weights(self) -> List[Tuple[str, Tuple[int, ...], bool]]:
        n = C.c_int()
        self.lib.check(self.lib.lib.comodgan_num_weights(self._h, C.byref(n)))
        out = []
        for i in range(n.value):
            name = C.c_char_p()
            shape = (C.c_int64 * 4)()
            nd, isb = C.c_int(), C.c_int()
            self.lib.check(self.lib.lib.comodgan_weight_info(self._h, i, C.byref(name), shape, C.byref(nd), C.byref(isb)))
            out.append((name.value.decode(), tuple(int(shape[k]) for k in range(nd.value)), bool(isb.value)))
        return out

    def set_weight(self, name: str, ptr: int, shape: Sequence[int]) -> None:
        arr = (C.c_int64 * max(1, len(shape)))(*[int(s) for s in shape])
        self.lib.check(self.lib.lib.comodgan_set_weight(self._h, name.encode(), C.c_void_p(ptr), arr, len(shape)))

    def commit(self, stream: int = 0) -> None:
        self.lib.check(self.lib.lib.comodgan_commit(self._h, C.c_void_p(stream)))

    def workspace_bytes(self, batch: int) -> int:
        n = C.c_size_t()
        self.lib.check(self.lib.lib.comodgan_workspace_bytes(self._h, int(batch), C.byref(n)))
        return int(n.value)

    def noise_floats(self) -> int:
        n = C.c_size_t()
        self.lib.check(self.lib.lib.comodgan_noise_floats(self._h, C.byref(n)))
        return int(n.value)

    def assume_static_weights(self, on: bool) -> None:
        self.lib.check(self.lib.lib.comodgan_assume_static_weights(self._h, 1 if on else 0))

    def forward(self, x_ptr: int, z_ptr: int, y_ptr: int, batch: int, ws_ptr: int, ws_bytes: int, truncation_psi: float = 1.0,
                noise_mode: str = "const", noise_ptr: Optional[int] = None, stream: int = 0, timed: bool = False):
        if noise_mode not in NOISE_MODES:
            raise AssertionError(noise_mode)             # stylegan.py:280
        args = [self._h, C.c_void_p(x_ptr), C.c_void_p(z_ptr), C.c_void_p(y_ptr), int(batch), C.c_float(truncation_psi),
                NOISE_MODES[noise_mode], C.c_void_p(noise_ptr), C.c_void_p(ws_ptr), C.c_size_t(ws_bytes), C.c_void_p(stream)]
        if not timed:
            self.lib.check(self.lib.lib.comodgan_forward(*args))
            return None
        n = len(self.launches())
        ms = (C.c_float * n)()
        self.lib.check(self.lib.lib.comodgan_forward_timed(*args, ms, n))
        return [float(v) for v in ms]

    def launches(self) -> List[Dict]:
        n = C.c_int()
        self.lib.check(self.lib.lib.comodgan_num_launches(self._h, C.byref(n)))
        out = []
        for i in range(n.value):
            layer, kern = C.c_char_p(), C.c_char_p()
            fl, mf, by = C.c_double(), C.c_double(), C.c_double()
            self.lib.check(self.lib.lib.comodgan_launch_info(self._h, i, C.byref(layer), C.byref(kern), C.byref(fl), C.byref(mf),
                                                             C.byref(by)))
            out.append(dict(layer=layer.value.decode(), kernel=kern.value.decode(), flops=fl.value, mfma_flops=mf.value,
                            bytes=by.value))
        return out

    def set_truncation_cutoff(self, cutoff: Optional[int]) -> None:
        """None: truncation_psi applies to every row of ws; n: to rows [0, n) only (stylegan.py:432-437)"""
        self.lib.check(self.lib.lib.comodgan_set_truncation_cutoff(self._h, -1 if cutoff is None else int(cutoff)))

    def set_debug(self, keep: bool) -> None:
        self.lib.check(self.lib.lib.comodgan_set_debug(self._h, 1 if keep else 0))

    def debug_tensor(self, batch: int, layer: str) -> Tuple[int, Tuple[int, ...]]:
        off = C.c_size_t()
        shape = (C.c_int64 * 4)()
        nd = C.c_int()
        self.lib.check(self.lib.lib.comodgan_debug_tensor(self._h, int(batch), layer.encode(), C.byref(off), shape, C.byref(nd)))
        return int(off.value), tuple(int(shape[k]) for k in range(nd.value))


_LIB: Optional[MiganLib] = None


_CLAMP_LIB: Optional["MiganLib"] = None


def load_library(path: Optional[str] = None, nan_policy: str = "propagate") -> MiganLib:
    """Process-wide libmigan_hip.so (raises MiganError when it is not built).  nan_policy="propagate" (default): a NaN activation stays a NaN
    through lrelu_agc's clamp, like Tensor.clamp in the reference module; "clamp": the build of the same library without that repair
    (libmigan_hip_nanclamp.so, -DMIGAN_NAN_CLAMP: v_med3_f32 turns a NaN into -256, ~2 % faster)."""
    global _LIB, _CLAMP_LIB
    if nan_policy not in ("clamp", "propagate"):
        raise ValueError(f"nan_policy must be 'clamp' or 'propagate', got {nan_policy!r}")
    if path is not None:
        return MiganLib(path)
    if nan_policy == "clamp":
        if _CLAMP_LIB is None:
            _CLAMP_LIB = MiganLib(os.path.join(_HERE, "csrc", "libmigan_hip_nanclamp.so"))
            if _CLAMP_LIB.nan_policy() != "clamp":
                raise MiganError("libmigan_hip_nanclamp.so was not built with -DMIGAN_NAN_CLAMP")
        return _CLAMP_LIB
    if _LIB is None:
        _LIB = MiganLib()
    return _LIB
