"""Shared builders for seeded IVF test corpora (used by CPU and GPU tests)."""
import numpy as np


def make_ivf(n, d, nlist, seed=0, metric="l2", empty=(), integer=False, id_base=0, shuffle_ids=True):
    """Fixed centroids + fixed assignment (no k-means): returns dict with centroids, CSR arena, partition lists.

    `empty` lists partition numbers forced to be empty.  `integer` draws SIFT-like small-integer data, where
    every fp32 partial sum is exact, so distances tie often -- the case the (key,id) tie rule exists for.
    """
    rng = np.random.default_rng(seed)
    if integer:
        cent = rng.integers(0, 40, size=(nlist, d)).astype(np.float32)
        assign = rng.integers(0, nlist, size=n)
        x = cent[assign] + rng.integers(-3, 4, size=(n, d)).astype(np.float32)
    else:
        cent = rng.standard_normal((nlist, d)).astype(np.float32)
        assign = rng.integers(0, nlist, size=n)
        x = (cent[assign] + 0.3 * rng.standard_normal((n, d))).astype(np.float32)
    if metric == "ip":
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        cent /= np.linalg.norm(cent, axis=1, keepdims=True)
    for e in empty:
        assign[assign == e] = (e + 1) % nlist
    ids = np.arange(n, dtype=np.int64) + id_base
    if shuffle_ids:
        ids = rng.permutation(ids)
    part_vecs, part_ids = [], []
    for p in range(nlist):
        m = assign == p
        part_vecs.append(np.ascontiguousarray(x[m]))
        part_ids.append(np.ascontiguousarray(ids[m]))
    sizes = np.array([len(i) for i in part_ids], np.int64)
    offsets = np.zeros(nlist + 1, np.int64)
    offsets[1:] = np.cumsum(sizes)
    vecs = np.concatenate(part_vecs, 0) if n else np.zeros((0, d), np.float32)
    aids = np.concatenate(part_ids) if n else np.zeros(0, np.int64)
    return dict(centroids=cent, vecs=np.ascontiguousarray(vecs, np.float32), ids=aids, offsets=offsets,
                part_vecs=part_vecs, part_ids=part_ids, x=x, all_ids=ids, assign=assign, d=d, nlist=nlist)


def make_queries(nq, d, seed=1, like=None, metric="l2", integer=False):
    rng = np.random.default_rng(seed)
    if like is not None:
        base = like[rng.integers(0, like.shape[0], size=nq)]
        if integer:
            q = base + rng.integers(-2, 3, size=(nq, d)).astype(np.float32)
        else:
            q = (base + 0.1 * rng.standard_normal((nq, d))).astype(np.float32)
    else:
        q = rng.standard_normal((nq, d)).astype(np.float32)
    if metric == "ip":
        q /= np.linalg.norm(q, axis=1, keepdims=True)
    return np.ascontiguousarray(q, np.float32)


def brute_force(x, ids, q, k, metric):
    """float64 ground truth with the canonical (key, id) order; returns (ids, dist, gap-to-next)."""
    x64, q64 = x.astype(np.float64), q.astype(np.float64)
    if metric == "l2":
        key = ((q64[:, None, :] - x64[None, :, :]) ** 2).sum(-1)
    else:
        key = -(q64 @ x64.T)
    order = np.lexsort((np.broadcast_to(ids, key.shape), key), axis=1)
    kk = min(k, x.shape[0])
    top = order[:, :kk]
    kv = np.take_along_axis(key, order[:, : kk + 1], 1)
    out_ids = ids[top]
    dist = np.sqrt(np.take_along_axis(key, top, 1)) if metric == "l2" else -np.take_along_axis(key, top, 1)
    gaps = np.diff(kv, axis=1)
    return out_ids, dist.astype(np.float32), gaps


def sharded_kmeans_reference(O, shards, m, metric, niter=5, seed=1234):
    """Single-process restatement of quake_amd.sharded.sharded_kmeans with the oracle's pieces: per-shard assignment and
    partial sums, partials added in rank order, one mean update + empty-cluster split per iteration.  Returns
    (centroids, [assign of shard r])."""
    world = len(shards)
    xs = [O.normalize_rows(x) if metric == "ip" else np.ascontiguousarray(x, np.float32) for x in shards]
    n = xs[0].shape[0]
    m_r = m // world
    sub = n * world > 256 * m
    ntrain = (256 * m) // world if sub else n
    perms = [O.rand_perm(n, ntrain if sub else m_r, seed + r) for r in range(world)]
    xt = [np.ascontiguousarray(x[p[:ntrain]]) if sub else x for x, p in zip(xs, perms)]
    c = np.concatenate([x[p[:m_r]] for x, p in zip(xs, perms)], 0)
    for _ in range(niter):
        s = cnt = None
        for r in range(world):
            a, _ = O.kmeans_assign(xt[r], c, metric)
            ps, pc = O.kmeans_accumulate(xt[r], a, m, blocked=True)
            s = ps if s is None else s + ps
            cnt = pc if cnt is None else cnt + pc
        c, _ = O.kmeans_update(s, cnt, c)
    if metric == "ip":
        c = O.normalize_rows(c)
    return c, [O.kmeans_assign(x, c, metric)[0] for x in xs]
