"""Dynamic workload harness: the build's counterpart of src/python/workload_generator.py.

DynamicWorkloadGenerator writes the SAME on-disk workload the reference's generator writes --
    <dir>/runbook.json   {parameters, initialize, operations{i: {type, sample_size, n_resident[, gt_time]}}, summary}
    <dir>/operations/<i>.pt, <i>_gt_ids.pt, <i>_gt_dists.pt, initial_indices.pt, base_vectors.pt, query_vectors.pt
(workload_generator.py:273-291,333-366) -- from a seeded stream of insert / delete / query operations drawn with
np.random.choice over the three ratios (:306-308); query ground truth is the brute-force top-100 over the vectors resident
at that point (:336-345), computed here on the GPU.  WorkloadEvaluator replays a workload against an index wrapper and
returns the per-operation records {operation_number, operation_type, latency_ms, recall, n_resident, n_list, n_total, ...}
(:512-521).  The reference's matplotlib figures are not produced (plots, not data).  The cluster-skewed sampler is kept
(StratifiedClusterSampler, :61-124) with cluster assignments from the device k-means.
"""
import json
import time
from pathlib import Path

import numpy as np
import torch

from .index import compute_recall
from .wrapper import QuakeWrapper


def knn(queries, vectors, k, metric="l2", device=0, chunk=1 << 20):
    """exact top-k (src/python/utils.py knn): ids [nq, k], distances [nq, k] (L2: euclidean; IP: dot), on the GPU."""
    dev = torch.device("cuda", device)
    q = torch.as_tensor(queries, dtype=torch.float32).to(dev)
    if q.dim() == 1:
        q = q[None, :]
    n = vectors.shape[0]
    k = n if k < 0 else min(k, n)
    best_v = torch.full((q.shape[0], k), float("inf"), device=dev)
    best_i = torch.full((q.shape[0], k), -1, dtype=torch.int64, device=dev)
    qn = (q * q).sum(1, keepdim=True)
    for i0 in range(0, n, chunk):
        xc = torch.as_tensor(vectors[i0:i0 + chunk], dtype=torch.float32).to(dev)
        key = (qn + (xc * xc).sum(1)[None, :] - 2.0 * (q @ xc.T)) if metric == "l2" else -(q @ xc.T)
        v, i = torch.topk(key, min(k, xc.shape[0]), dim=1, largest=False)
        cv, ci = torch.cat([best_v, v], 1), torch.cat([best_i, i + i0], 1)
        v2, j = torch.topk(cv, k, dim=1, largest=False)
        best_v, best_i = v2, torch.gather(ci, 1, j)
    dist = best_v.clamp_min(0).sqrt() if metric == "l2" else -best_v
    return best_i.cpu(), dist.cpu()


class UniformSampler:
    def sample(self, sample_pool, size, update_ranks=True):
        perm = torch.randperm(sample_pool.shape[0])
        return sample_pool[perm[:size]]


class StratifiedClusterSampler:
    """Draw from the clusters in order of distance from a moving root cluster (workload_generator.py:61-124)."""

    def __init__(self, assignments, centroids):
        self.assignments = assignments
        self.centroids = centroids
        non_empty = torch.unique(assignments)
        self.update_ranks(int(non_empty[torch.randint(0, non_empty.shape[0], (1,))]))

    def update_ranks(self, root_cluster):
        self.root_cluster = int(root_cluster)
        ids, _ = knn(self.centroids[self.root_cluster], self.centroids, -1, "l2")
        self.cluster_ranks = ids.flatten()

    def sample(self, sample_pool, size, update_ranks=True):
        sa = self.assignments[sample_pool]
        present = set(sa.tolist())
        order = [c for c in self.cluster_ranks.tolist() if c in present]
        out, got = [], 0
        for c in order:
            mask = (sa == c).nonzero(as_tuple=True)[0]
            if mask.numel() == 0:
                continue
            take = min(size - got, mask.numel())
            out.append(sample_pool[mask[torch.randperm(mask.numel())[:take]]])
            got += take
            if got >= size:
                break
        res = torch.cat(out) if out else torch.tensor([], dtype=torch.long)
        if update_ranks and len(order) > 1:
            self.update_ranks(order[1])
        return res


class DynamicWorkloadGenerator:
    def __init__(self, workload_dir, base_vectors, metric, insert_ratio, delete_ratio, query_ratio, update_batch_size,
                 query_batch_size, number_of_operations, initial_size, cluster_size, cluster_sample_distribution, queries,
                 query_cluster_sample_distribution="uniform", seed=1738, initial_clustering_path=None, overwrite=False):
        self.workload_dir = Path(workload_dir)
        self.base_vectors = torch.as_tensor(base_vectors, dtype=torch.float32)
        self.metric = metric.lower()
        self.insert_ratio, self.delete_ratio, self.query_ratio = insert_ratio, delete_ratio, query_ratio
        self.update_batch_size, self.query_batch_size = update_batch_size, query_batch_size
        self.number_of_operations = number_of_operations
        self.initial_size = initial_size
        self.cluster_size = cluster_size
        self.cluster_sample_distribution = cluster_sample_distribution
        self.query_cluster_sample_distribution = query_cluster_sample_distribution
        self.queries = torch.as_tensor(queries, dtype=torch.float32) if queries is not None else None
        self.seed = seed
        self.initial_clustering_path = Path(initial_clustering_path) if initial_clustering_path else None
        torch.manual_seed(seed)
        np.random.seed(seed)
        self.validate_parameters()
        self.workload_dir.mkdir(parents=True, exist_ok=True)
        self.operations_dir = self.workload_dir / "operations"
        self.operations_dir.mkdir(parents=True, exist_ok=True)
        self.resident_set = torch.zeros(self.base_vectors.shape[0], dtype=torch.bool)
        self.all_ids = torch.arange(self.base_vectors.shape[0])
        self.assignments = None
        self.runbook = {}
        self.clustered_index = None
        self.sampler = None
        self.query_sampler = None

    def workload_exists(self):
        return (self.workload_dir / "runbook.json").exists()

    def validate_parameters(self):  # workload_generator.py:197-209
        assert self.metric in ["l2", "ip"]
        assert 0 <= self.insert_ratio <= 1 and 0 <= self.delete_ratio <= 1 and 0 <= self.query_ratio <= 1
        assert abs(self.insert_ratio + self.delete_ratio + self.query_ratio - 1) < 1e-9
        assert self.update_batch_size > 0 and self.query_batch_size > 0
        assert self.number_of_operations > 0 and self.initial_size > 0 and self.cluster_size > 0
        assert self.cluster_sample_distribution in ["uniform", "skewed", "skewed_fixed"]

    def initialize_clustered_index(self):  # :211-233
        index_dir = self.initial_clustering_path or (self.workload_dir / "clustered_index.bin")
        index = QuakeWrapper()
        if index_dir.exists():
            index.load(index_dir)
        else:
            n_clusters = max(self.base_vectors.shape[0] // self.cluster_size, 1)
            index.build(self.base_vectors, nc=n_clusters, metric=self.metric, ids=torch.arange(self.base_vectors.shape[0]))
            index.save(str(self.workload_dir / "clustered_index.bin"))
        if index.index.parent is not None:
            from .index import SearchParams
            sp = SearchParams()
            sp.k = 1
            sp.batched_scan = True
            self.assignments = index.index.parent.search(self.base_vectors, sp).ids.flatten()
        else:
            self.assignments = torch.zeros(self.base_vectors.shape[0], dtype=torch.int64)
        return index

    def sample(self, size, operation_type):  # :235-256
        if operation_type == "insert":
            pool = self.all_ids[~self.resident_set]
        elif operation_type == "delete":
            pool = self.all_ids[self.resident_set]
        elif operation_type == "query":
            pool = torch.arange(self.queries.shape[0]) if self.queries is not None else self.all_ids[~self.resident_set]
        else:
            raise ValueError(f"Invalid operation type {operation_type}.")
        if pool.shape[0] == 0:
            return torch.tensor([], dtype=torch.long)
        if operation_type in ["insert", "delete"]:
            return self.sampler.sample(pool, size)
        return self.query_sampler.sample(pool, size, update_ranks=True)

    def initialize_workload(self):  # :258-297
        if self.sampler is None:
            if self.cluster_sample_distribution in ["skewed", "skewed_fixed"]:
                self.sampler = StratifiedClusterSampler(self.assignments, self.clustered_index.centroids())
            else:
                self.sampler = UniformSampler()
        if self.query_sampler is None:
            if self.query_cluster_sample_distribution in ["skewed", "skewed_fixed"] and self.queries is not None:
                cent = self.clustered_index.centroids()
                qa = knn(self.queries, cent, 1, "l2")[0].flatten()
                self.query_sampler = StratifiedClusterSampler(qa, cent)
            else:
                self.query_sampler = UniformSampler()
        initial = self.sample(self.initial_size, "insert")
        self.resident_set[initial] = True
        torch.save(initial, self.workload_dir / "initial_indices.pt")
        if self.queries is not None:
            torch.save(self.queries, self.workload_dir / "query_vectors.pt")
        torch.save(self.base_vectors, self.workload_dir / "base_vectors.pt")
        self.runbook["parameters"] = {
            "sample_queries": self.queries is None,
            "n_base_vectors": self.base_vectors.shape[0],
            "vector_dimension": self.base_vectors.shape[1],
            "metric": self.metric,
            "insert_ratio": self.insert_ratio,
            "delete_ratio": self.delete_ratio,
            "query_ratio": self.query_ratio,
            "update_batch_size": self.update_batch_size,
            "query_batch_size": self.query_batch_size,
            "number_of_operations": self.number_of_operations,
            "initial_size": self.initial_size,
            "cluster_size": self.cluster_size,
            "cluster_sample_distribution": self.cluster_sample_distribution,
            "query_cluster_sample_distribution": self.query_cluster_sample_distribution,
            "seed": self.seed,
        }
        self.runbook["initialize"] = {"size": self.initial_size}
        self.runbook["operations"] = {}

    def generate_workload(self):  # :299-397
        self.clustered_index = self.initialize_clustered_index()
        self.initialize_workload()
        n_inserts = n_deletes = n_queries = n_operations = 0
        for i in range(self.number_of_operations):
            op = np.random.choice(["insert", "delete", "query"], p=[self.insert_ratio, self.delete_ratio, self.query_ratio])
            if op == "insert":
                size, resident = self.update_batch_size, True
                n_inserts += 1
            elif op == "delete":
                size, resident = self.update_batch_size, False
                n_deletes += 1
            else:
                size, resident = self.query_batch_size, False
                n_queries += 1
            ids = self.sample(size, op)
            if ids.shape[0] == 0:
                break
            n_operations = i + 1
            if op in ["insert", "delete"]:
                self.resident_set[ids] = resident
            n_resident = int(self.resident_set.sum().item())
            if n_resident < 5 * self.update_batch_size:
                break
            entry = {"type": str(op), "sample_size": int(ids.shape[0]), "n_resident": n_resident}
            torch.save(ids, self.operations_dir / f"{i}.pt")
            if op == "query":
                q = self.queries[ids] if self.queries is not None else self.base_vectors[ids]
                t0 = time.time()
                resident_ids = self.all_ids[self.resident_set]
                gi, gd = knn(q, self.base_vectors[resident_ids], 100, self.metric)
                gi = resident_ids[gi]
                entry["gt_time"] = time.time() - t0
                torch.save(gi, self.operations_dir / f"{i}_gt_ids.pt")
                torch.save(gd, self.operations_dir / f"{i}_gt_dists.pt")
            self.runbook["operations"][i] = entry
        self.runbook["summary"] = {"n_inserts": n_inserts, "n_deletes": n_deletes, "n_queries": n_queries,
                                   "n_operations": n_operations}
        with open(self.workload_dir / "runbook.json", "w") as f:
            json.dump(self.runbook, f, indent=4)
        return self.runbook


def _sync():
    """operations are enqueued on the GPU: a latency is only attributable to its operation between two synchronisations"""
    if torch.cuda.is_available():
        torch.cuda.synchronize()


class WorkloadEvaluator:
    def __init__(self, workload_dir, output_dir, base_vectors_path=None):
        self.workload_dir = Path(workload_dir)
        self.output_dir = Path(output_dir)
        self.runbook_path = self.workload_dir / "runbook.json"
        self.operations_dir = self.workload_dir / "operations"
        self.initial_indices_path = self.workload_dir / "initial_indices.pt"
        self.base_vectors_path = Path(base_vectors_path) if base_vectors_path else self.workload_dir / "base_vectors.pt"
        self.runbook = None

    def initialize_index(self, name, index, build_params, m_params):  # :407-428
        index_dir = self.workload_dir / "init_indexes"
        index_dir.mkdir(parents=True, exist_ok=True)
        index_path = index_dir / f"{name}.index"
        vectors = torch.load(self.base_vectors_path, weights_only=True).to(torch.float32)
        initial = torch.load(self.initial_indices_path, weights_only=True).to(torch.int64)
        if not index_path.exists():
            index.build(vectors[initial], ids=initial, **build_params)
            index.save(index_path)
        else:
            index.load(index_path, n_workers=build_params.get("num_workers", 0))
        if isinstance(index, QuakeWrapper) and m_params is not None:
            index.index.initialize_maintenance_policy(m_params)
        return index

    def evaluate_workload(self, name, index, build_params, search_params, do_maintenance=False, m_params=None, batch=False):
        assert "k" in search_params, "search_params must contain 'k' for number of neighbors"
        base = torch.load(self.base_vectors_path, weights_only=True).to(torch.float32)
        index = self.initialize_index(name, index, build_params, m_params)
        if do_maintenance and isinstance(index, QuakeWrapper):
            index.index.track_hits = True  # maintenance() acts on recorded hits (see quake_amd/maintenance.py)
        self.runbook = json.load(open(self.runbook_path))
        qv = (base if self.runbook["parameters"]["sample_queries"]
              else torch.load(self.workload_dir / "query_vectors.pt", weights_only=True)).to(torch.float32)
        self.runbook["initialize"]["time"] = 0.0
        results = []
        for op_id, op in self.runbook["operations"].items():
            typ = op["type"]
            ids = torch.load(self.operations_dir / f"{op_id}.pt", weights_only=True)
            mean_recall = None
            maint = None
            _sync()
            if typ == "insert":
                t0 = time.time()
                index.add(base[ids], ids=ids, num_threads=16)
                _sync()
                op_time = time.time() - t0
            elif typ == "delete":
                t0 = time.time()
                index.remove(ids)
                _sync()
                op_time = time.time() - t0
            else:
                gt = torch.load(self.operations_dir / f"{op_id}_gt_ids.pt", weights_only=True)
                q = qv[ids]
                t0 = time.time()
                if batch:
                    pred = index.search(q, **search_params).ids
                else:
                    pred = torch.cat([index.search(q[i:i + 1], **search_params).ids for i in range(q.shape[0])])
                _sync()
                op_time = time.time() - t0
                mean_recall = compute_recall(pred.cpu(), gt, search_params["k"]).mean().item()
                self.runbook["operations"][op_id]["recall"] = mean_recall
            if do_maintenance:
                t0 = time.time()
                maint = index.maintenance()
                _sync()
                maint_ms = (time.time() - t0) * 1e3
            rec = {"operation_number": int(op_id), "operation_type": typ, "latency_ms": op_time * 1000, "recall": mean_recall,
                   "n_resident": op.get("n_resident")}
            if maint is not None:
                rec.update({"maintenance_ms": maint_ms, "n_splits": getattr(maint, "n_splits", 0),
                            "n_deletes": getattr(maint, "n_deletes", 0)})
            rec.update(index.index_state())
            rec.update(search_params)
            results.append(rec)
        self.output_dir.mkdir(parents=True, exist_ok=True)
        with open(self.output_dir / f"{name}_results.json", "w") as f:
            json.dump(results, f, indent=1)
        return results
