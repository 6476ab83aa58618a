"""Seeded random cases of the dense top-k (coarse step / flat search: qk_coarse over ONE list) against the oracle: rows 300 ... 300k,
d 8 ... 128, 1 ... 400 queries, k 1 ... 192, L2 / IP, duplicated rows (ties across the cut), ids in random order.  Every dense form is
reached (one-launch, fused, prefiltered, key matrix + pool / bisection selection).  python scripts/stress_dense.py [cases] [seed]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as O
from quake_amd.capi import Context, Store


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    ctx = Context(0)
    bad, forms, t0 = 0, {}, time.time()
    for c in range(cases):
        rng = np.random.default_rng(seed0 + c)
        d = int(rng.choice([8, 30, 32, 64, 100, 128]))
        n = int(rng.choice([300, 1024, 3000, 4096, 9000, 20000, 40000, 70000, 150000, 300000]))
        n = int(n * rng.uniform(0.8, 1.2))
        metric = "l2" if rng.random() < 0.6 else "ip"
        Q = int(rng.choice([1, 3, 20, 40, 63, 64, 100, 400]))
        k = int(min(n, rng.choice([1, 2, 10, 32, 64, 65, 100, 128, 150, 192])))
        cent = rng.standard_normal((n, d)).astype(np.float32)
        if rng.random() < 0.4:  # exact duplicates: equal keys, order decided by id
            m = int(min(n // 4, rng.integers(2, 400)))
            cent[n // 2:n // 2 + m] = cent[7]
        if metric == "ip":
            cent /= np.maximum(np.linalg.norm(cent, axis=1, keepdims=True), 1e-6)
        ids = rng.permutation(n).astype(np.int64)
        parent = Store(ctx, d)
        parent.build_csr(np.array([0, n], np.int64), ids, cent)
        q = (cent[rng.integers(0, n, Q)] + 0.3 * rng.standard_normal((Q, d))).astype(np.float32)
        q[0] = cent[7]
        gp, gd = ctx.coarse(parent, q, k, metric)
        forms[ctx.last_scan_kernel()] = forms.get(ctx.last_scan_kernel(), 0) + 1
        op, od = O.coarse(q, cent, ids, k, metric)
        if not (np.array_equal(gp, op) and np.array_equal(gd.view(np.uint32), od.view(np.uint32))):
            bad += 1
            print(json.dumps({"mismatch": c, "n": n, "d": d, "Q": Q, "k": k, "metric": metric, "kernel": ctx.last_scan_kernel()}), flush=True)
        parent.close()
    print(json.dumps({"script": "stress_dense.py", "cases": cases, "seed0": seed0, "mismatches": bad, "forms": forms, "seconds": round(time.time() - t0, 1)}))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
