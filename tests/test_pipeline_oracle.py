"""The CPU restatement of the deployed pre/post-processing pipeline (oracle/migan_pipeline_oracle.py) against outputs of the
reference's own MIGAN_Pipeline (tests/golden/pipeline_*.npz, tests/golden/make_golden_pipeline.py): bit-exact."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import migan_pipeline_oracle as po
from oracle import migan_torch_cpu as torc

CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "pipeline_*.npz")))


def test_goldens_exist():
    assert len(CASES) >= 4


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[9:-4] for p in CASES])
def test_oracle_reproduces_the_reference_pipeline(pkg, path):
    g = np.load(path)
    res, seed, padding = int(g["resolution"]), int(g["seed"]), int(g["padding"])
    sd = pkg.synth.make_state_dict(res, seed=seed, regime="export")
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    got, bbox, x = po.pipeline(g["image"], g["mask"], lambda t: torc.generator(t.numpy(), sd, res), res, padding)
    assert list(bbox) == [int(v) for v in g["bbox"]]
    np.testing.assert_array_equal(x[:, :, ::7, ::5], g["x_strided"])
    assert abs(float(x.astype(np.float64).sum()) - float(g["x_sum"])) <= 1e-9 * max(1.0, float(g["x_abs_sum"]))
    np.testing.assert_array_equal(got, g["result"])
