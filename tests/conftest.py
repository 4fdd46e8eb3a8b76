import os
import sys

import pytest

try:
    # torch bundles its own copy of the HIP runtime; it has to initialise BEFORE libpgrhip.so pulls in the system
    # runtime (/opt/rocm), or torch.cuda finds no device later in the same process (the other order is fine).
    # Tests that use torch (multi-GPU exchange) therefore need it imported first.
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "pgr-tk_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def test_seqs(oracle):
    return oracle.read_fasta(os.path.join(GOLDEN, "test_seqs.fa"))


@pytest.fixture(scope="session")
def gpu_ctx():
    """the product context; fails loudly (no skip) when the HIP library / device is missing"""
    import pgrtk_amd
    return pgrtk_amd.default_context(0)
