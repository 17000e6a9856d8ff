import json
import os
import sys

import numpy as np
import pytest
import torch  # noqa: F401  (before libkokorox_hip.so: one shared HIP runtime, see hip_koko.load_library)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def blob_path():
    from kokorox_amd import weights as W
    return W.ensure_synthetic_blob()


@pytest.fixture(scope="session")
def oracle(blob_path):
    from oracle import kokoro_ref as R
    return R.KokoroOracle(blob_path)


@pytest.fixture(scope="session")
def ref_inputs():
    with open(os.path.join(GOLD, "inputs_reference.json"), encoding="utf-8") as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden():
    out = {}
    for name in ("hello_world", "ort_sample_row"):
        with np.load(os.path.join(GOLD, f"forward_{name}.npz")) as z:
            out[name] = {k: z[k] for k in z.files}
    return out


@pytest.fixture(scope="session")
def hip_model(blob_path):
    """One HipKoko per session; constructing it fails loudly without the .so or a gfx950 GPU."""
    from kokorox_amd import hip_koko as hk
    m = hk.HipKoko.new(blob_path)
    yield m
    m.close()
