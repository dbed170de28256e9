import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _arithmetic_modes_do_not_leak():
    """libvit_hip.so keeps two process-wide settings (products per split-arithmetic launch, attention arithmetic) that the Python
    side syncs from module globals when an op runs; tests that call the C ABI directly must not inherit another test's mode."""
    yield
    vo = sys.modules.get("styl3r_amd.vit_ops")
    if vo is not None and getattr(vo, "_lib", None) is not None:
        vo._lib.vit_x6_set_products(6)
        vo._lib.vit_attention_set_arith({"f32": 0, "bf16x6": 1, "bf16x3": 2}.get(os.environ.get("VIT_ATTENTION", "bf16x6"), 1))
