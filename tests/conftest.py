import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# no pretrained VGG19 file exists offline: the tests opt into the seeded synthetic weights explicitly (the product
# raises without a checkpoint; tests/test_host_api.py checks that)
os.environ.setdefault("NST_SYNTHETIC_WEIGHTS", "1")


def _cap_cpu_threads():
    """The CPU oracle (torch fp32 convolutions) is what the parity tests spend their time in.  A GPU box shows every core of
    a much larger host (nproc = 256) while its share is 16 of them: torch's default of one thread per visible core then
    oversubscribes the share many times over and the suite takes 2-3x as long (measured 403 s vs 1080 s on two boxes)."""
    try:
        import torch
        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        torch.set_num_threads(int(os.environ.get("NST_CPU_THREADS", min(avail, 16))))
    except Exception:
        pass


def pytest_configure(config):
    _cap_cpu_threads()
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU test that takes more than ~20 s")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def vgg_weights():
    from oracle import cpu_ref
    return cpu_ref.synthetic_vgg19_weights(bias_std=cpu_ref.TEST_BIAS_STD)
