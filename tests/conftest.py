import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
ORACLE = os.path.join(ROOT, "oracle")
if ORACLE not in sys.path:
    sys.path.insert(0, ORACLE)  # tests are allowed to use the oracle (checker only)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def grad_family(n):
    """Parameter families with their own error level (bf16 HIP vs fp32 reference): shared by the encoder-twin and joiner
    tests so that every family gets a bar of ~2 x its measured worst instead of one bar for all (tests/test_model_gpu.py has
    the pre-training model's own, finer, list)."""
    if n.startswith("feature_extractor.") or ".feature_extractor." in n:
        return "extractor_norm" if ".2." in n or n.endswith(".0.bias") else "extractor_conv"
    if "layer_norm" in n:
        return "ln"
    if n.endswith(".bias"):
        return "bias"
    return "weight"


def by_family(errs):
    out = {}
    for n, e in errs.items():
        f = grad_family(n)
        out[f] = max(out.get(f, 0.0), float(e))
    return out


def dump_parity(tag, rep):
    """Measured errors go to gpurun_out/parity_<tag>.json on the GPU box (the bars in the tests are set from these files)."""
    import json
    d = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_%s.json" % tag), "w") as f:
            json.dump(rep, f, indent=1, sort_keys=True)
    except OSError:
        pass
