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
    c.set_form_feedback(False)  # this module pins WHICH form answers: the static rule alone (feedback: test_scan_feedback_gpu.py)
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


# (rows, lists, d, k, queries, nprobe) -> form.  Pairs per list = queries * nprobe / lists; the mixed form needs lists that
# average >= 1400 rows (round 4: and nothing else -- it used to start at three probing queries per list).
TABLE = [
    # one list per query: the 16 x 16 tile form (its static cut leaves the fewest records to merge)
    (60000, 1024, 128, 10, 1024, 1, "k_scan"),
    (100000, 64, 128, 10, 1024, 1, "k_scan (query-sharing)"),  # (16 queries per list)
    # several lists per query, under two probing queries per list: the per-wave row-per-lane walk
    (60000, 1024, 128, 10, 256, 4, "k_scan_rl"),
    (60000, 1024, 64, 10, 512, 2, "k_scan_rl"),
    # LONG lists (1400+ rows on average), nprobe > 1, 1024+ pairs: the mixed sequence whatever the sharing -- which lists are hot
    # (>= 18 probing queries) is decided per list on the device; without one the launch is the per-wave walk
    (720000, 512, 64, 10, 512, 2, "k_scan_rl (mixed)"),
    (720000, 512, 64, 10, 512, 3, "k_scan_rl (mixed)"),
    (1500000, 1024, 64, 10, 512, 2, "k_scan_rl (mixed)"),  # (one probing query per list on average)
    # LONG lists (1560 rows on average) and many probing queries per list: the mixed sequence (hot lists as dense items behind
    # the bf16 prefilter), whatever the sharing
    (100000, 64, 128, 10, 1024, 2, "k_scan_rl (mixed)"),
    (100000, 64, 128, 10, 1024, 8, "k_scan_rl (mixed)"),
    (100000, 64, 128, 10, 1024, 32, "k_scan_rl (mixed)"),
    (100000, 64, 100, 32, 1024, 8, "k_scan_rl (mixed)"),
    (100000, 64, 32, 1, 2048, 4, "k_scan_rl (mixed)"),
    # SHORT lists (58 rows on average): the round-2 rule -- per-wave walk under 6 probing queries per list, query-sharing tile
    # form from there on
    (60000, 1024, 128, 10, 1024, 2, "k_scan_rl"),
    (60000, 1024, 128, 10, 1024, 8, "k_scan (query-sharing)"),
    (60000, 1024, 128, 10, 1024, 32, "k_scan (query-sharing)"),
    # the row-per-lane form holds k <= 32 and d <= 128 (and d = 128 only up to k = 24: LDS): beyond, the tile form, with
    # query-sharing workgroups when lists are shared
    (100000, 64, 128, 32, 1024, 8, "k_scan (query-sharing)"),
    (100000, 64, 128, 100, 1024, 8, "k_scan (query-sharing)"),
    (100000, 64, 256, 10, 1024, 8, "k_scan (query-sharing)"),
    (60000, 1024, 256, 10, 1024, 1, "k_scan"),
    # small batches against a small flat parent: the one-launch search
    (60000, 1024, 128, 10, 1, 10, "k_search_small"),
    (60000, 1024, 128, 10, 32, 10, "k_search_small"),
    # fewer than 1024 pairs: no hot items (a wave's share is a chunk or two)
    (100000, 64, 128, 10, 64, 2, "k_scan_rl"),
]


@pytest.mark.parametrize("n,nlist,d,k,nq,nprobe,form", TABLE)
def test_form_of_the_scan(ctx, n, nlist, d, k, nq, nprobe, form):
    ivf = make_ivf(n, d, nlist, seed=5 + d)
    parent, s = _stores(ctx, ivf)
    q = make_queries(nq, d, seed=6, like=ivf["x"])
    gi, gd = ctx.search(parent, s, q, nprobe, k, "l2")
    assert ctx.last_scan_kernel() == form
    oi, od = O.search(q, ivf["centroids"], ivf["vecs"], ivf["ids"], ivf["offsets"], nprobe, k, "l2", batched_scan=True)
    np.testing.assert_array_equal(gi, oi)
    np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))
    s.close()
    parent.close()
