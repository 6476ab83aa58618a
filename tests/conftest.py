import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


# Collection order of the suite the driver runs with `-x`: the oracle-parity files first (a failure anywhere later cannot hide
# them), the C-ABI / mirror / store files next, replays, soaks and property streams last.  Files not named keep their alphabetical
# place in the middle block.
_FIRST = ["test_abi_symbols", "test_oracle_golden", "test_oracle_search", "test_oracle_kmeans", "test_oracle_aps",
          "test_bench_parity_gpu", "test_scan_gpu", "test_scan_mixed_gpu", "test_kmeans_gpu", "test_assign_pf_gpu",
          "test_dense_fused_gpu", "test_dense_pf_gpu", "test_store_dynamic_gpu", "test_aps_gpu", "test_index_gpu",
          "test_bindings_gpu", "test_group_gpu", "test_workers_gpu", "test_scan_form_selection_gpu", "test_scan_feedback_gpu",
          "test_maintenance_gpu", "test_rccl_world1_gpu", "test_sharded_gpu", "test_sharded_maintenance_gpu"]
_LAST = ["test_full_size_gpu", "test_random_shapes_gpu", "test_random_index_streams_gpu", "test_dynamic_workload_10m_gpu"]


def collection_rank(path):
    name = os.path.splitext(os.path.basename(str(path)))[0]
    if name in _FIRST:
        return (0, _FIRST.index(name))
    if name in _LAST:
        return (2, _LAST.index(name))
    return (1, 0)


def pytest_collection_modifyitems(session, config, items):
    items.sort(key=lambda it: collection_rank(it.fspath))  # (stable: the order inside a file is kept)
