import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "needs_reference: needs /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isdir("/root/reference/videochat_flash")
    skip_ref = pytest.mark.skip(reason="/root/reference not present")
    have_gpu = None
    for it in items:
        if "needs_reference" in it.keywords and not have_ref:
            it.add_marker(skip_ref)
        if "gpu" in it.keywords:
            if have_gpu is None:                     # a plain `pytest tests` on a CPU box skips the GPU tests instead of failing them
                import torch
                lib = os.path.join(ROOT, "blim_amd", "libblim_hip.so")
                have_gpu = (torch.cuda.device_count() > 0, os.path.exists(lib))
            if not have_gpu[0]:
                it.add_marker(pytest.mark.skip(reason="no HIP device visible"))
            elif not have_gpu[1]:
                pass                                 # on a GPU box a missing library must FAIL loudly, not skip
