import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "midi-emotion_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")

# Collection order of the GPU suite: every comparison of a HIP kernel with the oracle / a golden fixture first, the
# multi-process files last, so that a `pytest -x` stop in a launcher test still leaves the whole parity record.
_ORDER = ["test_oracle_golden", "test_host_cpu", "test_data_cpu", "test_kernels_gpu", "test_model_gpu", "test_fp16_tier_gpu", "test_decode_gpu",
          "test_c_abi_example", "test_train_cli_gpu", "test_ddp_gpu"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return _ORDER.index(mod) if mod in _ORDER else len(_ORDER) - 3      # unknown files: before the launcher tests
    items.sort(key=rank)                                                    # stable: order inside a file is kept


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
