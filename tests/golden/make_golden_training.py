#!/usr/bin/env python3
"""Second, independent oracle: the reference's TRAINING generator + its own export step.

    lib/model_zoo/migan.py  Generator(Encoder, Synthesis) with depthwise=True, reparametrize=True, num_reparam_tensors=9
    (configs/model/migan.yaml:273-293), forward(x, noise_mode='const')
        vs
    scripts/export_inference_model.py::copy_weights (:17-85) -> lib/model_zoo/migan_inference.py Generator (:149-151 of the script)

The training model computes the same function through different code (torch_utils/ops upfirdn2d + conv2d_resample, weights
summed and L2-normalised on the fly, migan.py:108-128), so its output on the same input pins the whole chain
"training snapshot -> re-parameterisation -> inference forward".  This script

  1. fills the training generator with seeded tensors (mi-gan_amd/synth.py, one tensor per state_dict key),
  2. records its output y_train = G(x, noise_mode='const'),
  3. runs the reference's own copy_weights into the reference's inference Generator, checks the reference's own
     self-check (isclose(rtol=1e-3), export_inference_model.py:149-151) and records that output too,
  4. stores the recipe (training keys, shapes) + outputs; tests regenerate the training tensors from the recipe, convert them
     with mi-gan_amd/convert.py and run OUR forward (oracle on CPU, HIP kernels on the GPU) against y_train.

Build container only (imports /root/reference):   PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_training.py
"""
import importlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("MIGAN_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.dont_write_bytecode = True
for name in ("cv2", "torchvision", "torchvision.transforms"):         # imported at module level by the export script, unused here
    if name not in sys.modules:
        try:
            __import__(name)
        except Exception:
            sys.modules[name] = types.ModuleType(name)
if not hasattr(sys.modules["torchvision"], "transforms"):
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]

pkg = importlib.import_module("mi-gan_amd")
synth = pkg.synth
import lib.model_zoo.migan as train_mod                     # noqa: E402  (training-time model)
import lib.model_zoo.migan_inference as inf_mod             # noqa: E402
spec = importlib.util.spec_from_file_location("ref_export", os.path.join(REF, "scripts", "export_inference_model.py"))
ref_export = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_export)

torch.set_num_threads(max(1, os.cpu_count() or 1))


def leaf_scale(key: str) -> float:
    leaf = key.rsplit(".", 1)[1]
    if leaf == "bias":
        return 0.5
    if leaf == "noise_strength":
        return 0.3
    return 1.0          # w0..w8, weight, noise_const


def case(tag, resolution, batch, seed):
    kw = dict(resolution=resolution, ch_base=32768, ch_max=512, depthwise=True, reparametrize=True, num_reparam_tensors=9)
    G = train_mod.Generator(train_mod.Encoder(ic_n=4, **kw), train_mod.Synthesis(rgb_n=3, **kw)).eval()
    sd = G.state_dict()
    new, recipe = {}, []
    for k, v in sd.items():
        if k.endswith("resample_filter") or "filter" in k.rsplit(".", 1)[1]:
            new[k] = v                                        # FIR buffers keep their constructor values
            continue
        t = torch.from_numpy((synth.normal(tuple(v.shape) or (1,), seed, "train/" + k) * leaf_scale(k)).astype(np.float32)).reshape(v.shape)
        new[k] = t
        recipe.append((k, tuple(v.shape)))
    G.load_state_dict(new, strict=True)
    x = synth.make_input(batch, resolution, seed=seed)
    with torch.no_grad():
        y_train = G(torch.from_numpy(x.copy()), noise_mode="const")
        dest = inf_mod.Generator(resolution=resolution).eval()
        ref_export.copy_weights(G, dest, resolution=resolution)            # the reference's own re-parameterisation
        y_inf = dest(torch.from_numpy(x.copy()))
    mism = (1 - torch.isclose(y_train, y_inf, rtol=1e-3).float().mean()).item()   # the export script's self-check (:149-151)
    diff = float((y_train - y_inf).abs().max())
    print(tag, "y", tuple(y_train.shape), "absmax", float(y_train.abs().max()), "training vs exported inference max diff", diff,
          "isclose mismatch", mism)
    assert diff <= 1e-3
    # the reference's exported state_dict, sampled: the converter test pins to it as well
    samp = {}
    for k, v in dest.state_dict().items():
        f = v.reshape(-1).double()
        samp["inf/" + k] = np.array([float(f.sum()), float((f * f).sum())])
    np.savez_compressed(os.path.join(HERE, f"training_{tag}.npz"), resolution=resolution, batch=batch, seed=seed,
                        keys=np.array([k for k, _ in recipe]), shapes=np.array([",".join(map(str, s)) for _, s in recipe]),
                        y_train=y_train.numpy().astype(np.float32), y_inference=y_inf.numpy().astype(np.float32),
                        y_absmax=float(y_train.abs().max()), **samp)


if __name__ == "__main__":
    case("r16", 16, 2, 21)
    case("r64", 64, 2, 22)
    case("r256", 256, 1, 23)
