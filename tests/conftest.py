import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def weights16():
    """Synthetic refiner weights (563.5 M parameters, regenerated from the seed, ~25 s)."""
    import torch
    from hifidiff_amd import synth
    torch.set_grad_enabled(False)
    return synth.refiner_state_dict(16)


def rel_l2(a, b):
    import torch
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def psnr(a, b, data_range=6.0):
    import torch
    mse = float(((torch.as_tensor(a, dtype=torch.float64) - torch.as_tensor(b, dtype=torch.float64)) ** 2).mean())
    return 10.0 * np.log10(data_range ** 2 / max(mse, 1e-30))


def psnr_pp(a, b):
    """PSNR with the GOLDEN's own peak-to-peak as the data range (a fixed range of 6 hides errors on small signals)."""
    import torch
    a, b = torch.as_tensor(a, dtype=torch.float64), torch.as_tensor(b, dtype=torch.float64)
    rng = float(b.max() - b.min())
    return 10.0 * np.log10(rng ** 2 / max(float(((a - b) ** 2).mean()), 1e-30))
