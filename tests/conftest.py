import glob
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_files(prefix):
    return sorted(glob.glob(os.path.join(GOLDEN, prefix + "_*.npz")))


def load_golden(path):
    """-> dict of torch tensors / python scalars, grouped: in, sd, grad_in, grad_sd, mid, meta, out, cot."""
    z = np.load(path)
    g = {"in": {}, "sd": {}, "grad_in": {}, "grad_sd": {}, "mid": {}, "meta": {}, "raw": {}}
    for k in z.files:
        v = z[k]
        if k.startswith("in."):
            g["in"][k[3:]] = torch.from_numpy(v)
        elif k.startswith("sd."):
            g["sd"][k[3:]] = torch.from_numpy(v)
        elif k.startswith("grad.in."):
            g["grad_in"][k[8:]] = torch.from_numpy(v)
        elif k.startswith("grad.sd."):
            g["grad_sd"][k[8:]] = torch.from_numpy(v)
        elif k.startswith("mid."):
            g["mid"][k[4:]] = torch.from_numpy(v)
        elif k.startswith("meta."):
            g["meta"][k[5:]] = int(v)
        elif k in ("out", "cot"):
            g[k] = torch.from_numpy(v)
        else:
            g["raw"][k] = v
    return g


def ids(paths):
    return [os.path.basename(p)[:-4] for p in paths]


@pytest.fixture(scope="session")
def gpu_device():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
