import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run on the GPU box with -m gpu")


# Order of the GPU suite: the evidence first, the soaks last.  The driver runs `pytest -m gpu -x`; a red soak
# must still make the run red, but it must not erase the parity record: the oracle-parity tests of the kernels and of the
# full-size configurations (BASELINE.json configs 2, 3 and 5) run first, then the command line and the back-end, then
# everything whose subject is time (co-tenancy, fuzzing, the error tail, the stream-K soak, RCCL).
_GPU_FILE_ORDER = ["test_gpu_full_batch_parity", "test_gpu_frames", "test_gpu_forward", "test_gpu_frontend", "test_gpu_kernels",
                   "test_gpu_edge_cases", "test_gpu_cli", "test_gpu_backend", "test_gpu_shards", "test_gpu_variants", "test_gpu_rccl",
                   "test_gpu_tail", "test_gpu_fuzz"]
_SOAK_WORDS = ("concurrent", "soak", "do_not_depend_on_what_else", "fuzz", "no_chunk_of_4096")


def pytest_collection_modifyitems(session, config, items):
    if os.environ.get("XVEC_TEST_ORDER") == "file":         # (reproducing an earlier round's run order)
        return
    def rank(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        if item.get_closest_marker("gpu") is None:
            return (0, 0)                                   # the CPU suite keeps its order, in front
        soak = any(w in item.name for w in _SOAK_WORDS)
        pos = _GPU_FILE_ORDER.index(mod) if mod in _GPU_FILE_ORDER else len(_GPU_FILE_ORDER)
        return (2 if soak else 1, pos)
    items.sort(key=rank)                                    # stable: the order inside a file is kept


@pytest.fixture(scope="session")
def pkg():
    import importlib
    return importlib.import_module("speaker-embedding-with-phonetic-information_amd")
