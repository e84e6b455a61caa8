import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(autouse=True)
def _restore_default_tier():
    """Every test leaves the process in the default tier (bf16, bf16 residual stream, no operand split) even when it FAILS in the
    middle of a tier switch -- otherwise one failing tier-parametrized test leaks its tier into the tests that follow."""
    yield
    try:
        import torch
        from recguru_amd import hip, ops
    except Exception:
        return
    if ops.compute_tier() != "bf16" or hip.SPLIT_OPERANDS or ops.residual_dtype() != torch.bfloat16:
        ops.set_compute_dtype(torch.bfloat16)
        ops.set_residual_dtype(torch.bfloat16)
        hip.SPLIT_OPERANDS = False
