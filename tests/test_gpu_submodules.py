"""The reference's module-level API below Generator (missing 3): SeparableConv2d / EncoderBlock / Encoder /
SynthesisBlockFirst / SynthesisBlock / Synthesis are callable on the GPU with the reference's signatures (migan_inference.py:154-170,
:192-200, :235-246, :271-280, :302-315, :347-352), each through migan_sepconv_forward.  Compared with the torch-CPU port's taps of the
same sub-modules and with the fused Generator.forward."""
import numpy as np
import pytest
import torch

from oracle import migan_torch_cpu as torc

pytestmark = pytest.mark.gpu
TOL = 1e-3      # the north star's tolerance for the whole generator; per-module errors are ~1e-5


def _model(pkg, res, seed, dev):
    sd = pkg.synth.make_state_dict(res, seed=seed, regime="export")
    m = pkg.Generator(resolution=res)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    return m.to(dev).eval(), sd


@pytest.mark.parametrize("res,batch", [(64, 2), (256, 1)])
def test_encoder_and_synthesis_forward(pkg, res, batch):
    dev = torch.device("cuda:0")
    m, sd = _model(pkg, res, 41, dev)
    x = pkg.synth.make_input(batch, res, seed=41, kind="demo")
    taps = {}
    want = torch.as_tensor(np.asarray(torc.generator(x, sd, res, taps=taps)))
    xd = torch.from_numpy(x).to(dev)
    with torch.no_grad():
        h, feats = m.encoder(xd)                                  # Encoder.forward(img) -> (x, feats) (:235-246)
        assert sorted(feats) == sorted(4 * 2 ** i for i in range(int(np.log2(res)) - 1))
        for r, f in feats.items():
            ref = torch.as_tensor(np.asarray(taps[f"encoder.b{r}.conv1"]))
            assert tuple(f.shape) == tuple(ref.shape)
            assert float((f.cpu() - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max())), r
        ref = torch.as_tensor(np.asarray(taps["encoder.b4.conv2"]))
        assert float((h.cpu() - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))
        img = m.synthesis(h, feats)                               # Synthesis.forward(x, enc_feats) -> img (:347-352)
        fused = m(xd)
    assert float((img.cpu() - want).abs().max()) <= TOL
    assert float((img - fused).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max()))      # same kernels, module by module


def test_blocks_and_separable_conv(pkg):
    dev = torch.device("cuda:0")
    res = 32
    m, sd = _model(pkg, res, 42, dev)
    x = pkg.synth.make_input(3, res, seed=42, kind="demo")
    taps = {}
    torc.generator(x, sd, res, taps=taps)
    t = {k: torch.as_tensor(np.asarray(v)) for k, v in taps.items()}
    xd = torch.from_numpy(x).to(dev)

    def close(a, b, what):
        assert tuple(a.shape) == tuple(b.shape), what
        assert float((a.cpu() - b).abs().max()) <= 1e-4 * max(1.0, float(b.abs().max())), what

    with torch.no_grad():
        h, feat = m.encoder.b32(None, xd)                         # EncoderBlock.forward(x, img) with fromrgb (:192-200)
        close(feat, t["encoder.b32.conv1"], "b32 feat")
        close(h, t["encoder.b32.conv2"], "b32 out")
        h16, feat16 = m.encoder.b16(h, xd)                        # ... without fromrgb: img is ignored
        close(feat16, t["encoder.b16.conv1"], "b16 feat")
        close(m.encoder.b16.conv2(feat16), t["encoder.b16.conv2"], "SeparableConv2d down=2")     # SeparableConv2d.forward (:154-170)
        # synthesis blocks on the oracle's own intermediate tensors
        x4 = t["encoder.b4.conv2"].to(dev)
        f4 = t["encoder.b4.conv1"].to(dev)
        s4, img4 = m.synthesis.b4(x4, f4)                         # SynthesisBlockFirst.forward(x, enc_feat) (:271-280)
        close(s4, t["synthesis.b4.conv2"], "b4 x")
        close(img4, t["synthesis.b4.img"], "b4 img")
        s8, img8 = m.synthesis.b8(s4, t["encoder.b8.conv1"].to(dev), img4)       # SynthesisBlock.forward(x, enc_feat, img) (:302-315)
        close(s8, t["synthesis.b8.conv2"], "b8 x")
        close(img8, t["synthesis.b8.img"], "b8 img")
        close(m.synthesis.b8.conv1(s4), t["synthesis.b8.conv1"], "SeparableConv2d up=2 + noise")


def test_submodule_errors(pkg):
    dev = torch.device("cuda:0")
    m, _ = _model(pkg, 32, 43, dev)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m.encoder(torch.zeros(1, 4, 32, 32))
    with pytest.raises(RuntimeError, match="expected input"):
        m.encoder.b16.conv1(torch.zeros(1, 7, 16, 16, device=dev))
    with pytest.raises(NotImplementedError, match="fused into"):
        m.synthesis.b8.torgb(torch.zeros(1, 512, 8, 8, device=dev))          # leaves stay parameter containers
    with pytest.raises(RuntimeError, match="noise_const"):
        m.synthesis.b8.conv2(torch.zeros(1, 512, 16, 16, device=dev))         # fixed-size noise, like the reference's broadcast error
