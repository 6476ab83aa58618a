"""Which form of the partition scan the host rule picks for which shape (qk_scan_device, qk_scan.hip) -- pinned, because the
rule is a table of measured crossovers and a silent change of it is a performance regression no parity test sees.  The
shapes are small stores; the rule looks at the row width (16-column blocks), k, nprobe, the number of (query, list) pairs and
their batch average per list -- not at the corpus size.  qk_ctx_last_scan_kernel names the form; every answer is also checked
against the oracle (query_coordinator.cpp:612-799)."""
import numpy as np
import pytest

import oracle as O
from helpers import make_ivf, make_queries

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from quake_amd.capi import Context
    c = Context(0)
    yield c
    c.close()


def _stores(ctx, ivf):
    from quake_amd.capi import Store
    s = Store(ctx, ivf["d"])
    s.build_csr(ivf["offsets"], ivf["ids"], ivf["vecs"])
    parent = Store(ctx, ivf["d"])
    nlist = ivf["nlist"]
    parent.build_csr(np.array([0, nlist], np.int64), np.arange(nlist, dtype=np.int64), ivf["centroids"])
    return parent, s


# (d, k, queries, nprobe) -> form.  nlist = 1024 throughout: pairs per list = queries * nprobe / 1024.
TABLE = [
    # one list per query: the 16 x 16 tile form (its static cut leaves the fewest records to merge)
    (128, 10, 1024, 1, "k_scan"),
    # several lists per query, under two probing queries per list: the per-wave row-per-lane walk
    (128, 10, 256, 4, "k_scan_rl"),
    (64, 10, 512, 2, "k_scan_rl"),
    # from two probing queries per list on: the mixed sequence (hot lists as dense items behind the bf16 prefilter)
    (128, 10, 1024, 2, "k_scan_rl (mixed)"),
    (128, 10, 1024, 8, "k_scan_rl (mixed)"),
    (128, 10, 1024, 32, "k_scan_rl (mixed)"),
    (100, 32, 1024, 8, "k_scan_rl (mixed)"),
    (32, 1, 2048, 4, "k_scan_rl (mixed)"),
    # the row-per-lane form holds k <= 32 and d <= 128 (and d = 128 only up to k = 24: LDS): beyond, the tile form, with
    # query-sharing workgroups when lists are shared
    (128, 32, 1024, 8, "k_scan (query-sharing)"),
    (128, 100, 1024, 8, "k_scan (query-sharing)"),
    (256, 10, 1024, 8, "k_scan (query-sharing)"),
    (256, 10, 1024, 1, "k_scan"),
    # small batches against a small flat parent: the one-launch search
    (128, 10, 1, 10, "k_search_small"),
    (128, 10, 32, 10, "k_search_small"),
    # fewer than 1024 pairs: no hot items (a wave's share is a chunk or two)
    (128, 10, 64, 10, "k_scan_rl"),
]


@pytest.mark.parametrize("d,k,nq,nprobe,form", TABLE)
def test_form_of_the_scan(ctx, d, k, nq, nprobe, form):
    ivf = make_ivf(60000, d, 1024, seed=5 + d)
    parent, s = _stores(ctx, ivf)
    q = make_queries(nq, d, seed=6, like=ivf["x"])
    gi, gd = ctx.search(parent, s, q, nprobe, k, "l2")
    assert ctx.last_scan_kernel() == form
    oi, od = O.search(q, ivf["centroids"], ivf["vecs"], ivf["ids"], ivf["offsets"], nprobe, k, "l2", batched_scan=True)
    np.testing.assert_array_equal(gi, oi)
    np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))
    s.close()
    parent.close()
