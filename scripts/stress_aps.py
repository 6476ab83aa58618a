"""Seeded random cases of the recall-target search against the oracle's walk (ids, float32 distance bits, partitions visited):
shapes, metrics, k, targets, thresholds, candidate fractions, batch sizes, lists shorter than k at the head of a query's ranking
(the first round's sample bound must not come from them), empty lists.  Run on the GPU box:
    python scripts/stress_aps.py [cases] [seed]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O
from helpers import make_ivf, make_queries
from quake_amd.capi import Context, Store


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    ctx = Context(0)
    bad = 0
    t0 = time.time()
    done = 0
    for c in range(cases):
        rng = np.random.default_rng(seed0 + c)
        d = int(rng.choice([8, 16, 32, 48, 64, 96, 128]))
        nlist = int(rng.choice([12, 40, 64, 150, 400]))
        n = int(rng.integers(nlist * 20, nlist * 400))
        if rng.random() < 0.35:  # long lists: the row-per-lane / mixed forms of the per-pair scan become admissible
            nlist = int(rng.choice([12, 40, 64]))
            n = int(rng.integers(nlist * 1500, nlist * 3000))
        metric = "l2" if rng.random() < 0.7 else "ip"
        k = int(rng.choice([1, 3, 10, 10, 32, 50]))
        empty = tuple(int(v) for v in rng.choice(nlist, size=int(rng.integers(0, 3)), replace=False))
        ivf = make_ivf(n, d, nlist, seed=seed0 + c, metric=metric, empty=empty)
        # a few lists cut down to fewer than k rows: queries that rank one of them first get no sample bound in round 0
        offs, ids, vecs = ivf["offsets"].copy(), ivf["ids"], ivf["vecs"]
        short = [int(v) for v in rng.choice(nlist, size=int(rng.integers(0, 4)), replace=False)]
        keep = np.ones(len(ids), bool)
        for p in short:
            a, b = int(offs[p]), int(offs[p + 1])
            cut = int(rng.integers(0, max(1, min(k, b - a))))
            keep[a + cut:b] = False
        sizes = np.array([keep[int(offs[p]):int(offs[p + 1])].sum() for p in range(nlist)], np.int64)
        offs2 = np.zeros(nlist + 1, np.int64); offs2[1:] = np.cumsum(sizes)
        ids2, vecs2 = np.ascontiguousarray(ids[keep]), np.ascontiguousarray(vecs[keep])
        Q = int(rng.choice([1, 5, 33, 64, 200, 700]))
        q = make_queries(Q, d, seed=seed0 + 7 * c + 1, like=ivf["x"], metric=metric)
        if short and rng.random() < 0.5:  # some queries right at a short list's centroid
            for t, p in enumerate(short[:Q]):
                q[t] = ivf["centroids"][p]
        frac = float(rng.choice([0.1, 0.25, 0.5, 1.0]))
        if int(np.float32(nlist) * np.float32(frac)) < 2:
            frac = 1.0
        rt = float(rng.choice([0.5, 0.8, 0.9, 0.99]))
        thr = float(rng.choice([0.0, 0.001, 0.01, 0.05]))
        pre = bool(rng.random() < 0.5)
        s = Store(ctx, d); s.build_csr(offs2, ids2, vecs2)
        parent = Store(ctx, d)
        parent.build_csr(np.array([0, nlist], np.int64), np.arange(nlist, dtype=np.int64), ivf["centroids"])
        ok = True
        for rep in range(2):  # (twice: the form feedback takes another form the second time)
            gi, gd, gn = ctx.search_aps(parent, s, q, k, metric, rt, recompute_threshold=thr, use_precomputed=pre, initial_search_fraction=frac)
            if rep == 0:
                oi, od, on = O.search_aps(q, ivf["centroids"], vecs2, ids2, offs2, k, metric, rt, recompute_threshold=thr,
                                          use_precomputed=pre, initial_search_fraction=frac, expanded=True, num_threads=8)
            ok = ok and np.array_equal(gn, on) and np.array_equal(gi, oi) and np.array_equal(gd.view(np.uint32), od.view(np.uint32))
        if not ok:
            bad += 1
            print(json.dumps({"mismatch": c, "d": d, "nlist": nlist, "n": int(len(ids2)), "metric": metric, "k": k, "Q": Q, "frac": frac,
                              "target": rt, "thr": thr, "short": short}), flush=True)
        done += 1
        s.close(); parent.close()
    print(json.dumps({"script": "stress_aps.py", "cases": done, "seed0": seed0, "mismatches": bad, "seconds": round(time.time() - t0, 1)}))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
