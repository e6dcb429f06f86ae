import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # same setting as bench.py / smoke(): GEMMs go to hipBLASLt, BatchNorm to PyTorch's native
    # kernels; MIOpen would JIT-compile one kernel per new shape on a fresh box.
    import torch
    torch.backends.cudnn.enabled = False


# Collection order of the GPU files (under `-x` the weakest test must not gate the strongest): the
# bit-exact oracle parity of the kernels first, then the fused kernels against fp32 PyTorch, the segmented launches
# against separate calls, the reference goldens, and the end-to-end step / configuration tests last.
_ORDER = ["test_oracle_cpu.py", "test_abi_cpu.py", "test_ops_gpu.py", "test_mlp_gpu.py", "test_segments_gpu.py",
          "test_golden_models.py", "test_ddp_cpu.py", "test_graph_gpu.py", "test_ddp_graph_gpu.py", "test_configs_gpu.py"]


def pytest_collection_modifyitems(session, config, items):
    rank = {name: i for i, name in enumerate(_ORDER)}
    items.sort(key=lambda it: rank.get(os.path.basename(str(it.fspath)), len(_ORDER)))      # stable: file order kept


@pytest.fixture(scope="session")
def hip_lib():
    """Build (if stale and hipcc exists) and load the C-ABI library."""
    import tpgan_amd  # noqa: F401
    from tpgan_amd import _lib
    from tpgan_amd.build import build_hip, is_stale
    if is_stale() and os.path.exists("/opt/rocm/bin/hipcc"):
        build_hip()
    return _lib.load()


@pytest.fixture()
def oracle_cpu():
    """Register the oracle as the op backend for CPU tensors for one test."""
    import tpgan_amd  # noqa: F401
    from oracle import torch_backend
    torch_backend.install()
    yield
    torch_backend.uninstall()


@pytest.fixture(autouse=True)
def _collect_graphs_between_tests(request):
    """GPU tests build hipGraph steppers that sit in reference cycles: collect them HERE, between tests,
    not whenever the cyclic collector happens to run (inside the next test's capture or replay)."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import gc

        import torch
        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
