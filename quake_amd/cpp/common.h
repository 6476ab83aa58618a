// common.h -- parameter / result records of the C++ host mirror: the names, fields and defaults of the reference's
// src/cpp/include/common.h:66-276 (IndexBuildParams, SearchParams, *TimingInfo, SearchResult, Clustering, str_to_metric_type),
// so that C++ callers and the pybind11 module see the records they know.  No arithmetic here.
#pragma once
#include <torch/torch.h>

#include <memory>
#include <string>
#include <vector>

#include "../../include/quake_hip.h"

using torch::Tensor;
using std::shared_ptr;

namespace quake_amd {

// faiss::MetricType's two values the reference uses (common.h:145-156); the numeric codes are the C ABI's
enum MetricType { METRIC_INNER_PRODUCT = QK_METRIC_IP, METRIC_L2 = QK_METRIC_L2 };

constexpr int DEFAULT_NLIST = 0, DEFAULT_NITER = 5, DEFAULT_NUM_WORKERS = 0, DEFAULT_K = 1, DEFAULT_NPROBE = 1;
constexpr const char *DEFAULT_METRIC = "l2";
constexpr float DEFAULT_RECALL_TARGET = -1.0f;

struct MaintenancePolicyParams {  // common.h:104-118
    std::string maintenance_policy = "query_cost";
    int window_size = 1000;
    int refinement_radius = 25;
    int refinement_iterations = 3;
    int min_partition_size = 32;
    float alpha = 0.9f;
    bool enable_split_rejection = true;
    bool enable_delete_rejection = true;
    float delete_threshold_ns = 10.0f;
    float split_threshold_ns = 10.0f;
    // EXTENSION (no reference field): a partition the delete branch examined and the rejection rule KEPT is then tested for a split
    // like every other kept partition.  The reference (maintenance_policies.cpp:68-131) never reaches its split test for such a
    // partition -- and its delete model is most negative for the partitions that are both larger and hotter than average, so the
    // lists a skewed insert stream grows are never split.  false = the reference's decisions.
    bool split_after_delete_rejection = false;
};

struct IndexBuildParams {  // common.h:123-143
    int dimension = 0;
    int nlist = DEFAULT_NLIST;
    int num_workers = DEFAULT_NUM_WORKERS;
    int code_size = -1;
    int num_codebooks = -1;
    std::string metric = DEFAULT_METRIC;
    int niter = DEFAULT_NITER;
    bool use_adaptive_nprobe = false;
    bool use_numa = false;
    bool use_gpu = true;  // GPU k-means is the only k-means here
    bool verify_numa = false;
    bool same_core = true;
    bool verbose = false;
    shared_ptr<IndexBuildParams> parent_params = nullptr;
};

struct SearchParams {  // common.h:171-184
    int nprobe = DEFAULT_NPROBE;
    int k = DEFAULT_K;
    float recall_target = DEFAULT_RECALL_TARGET;
    int num_threads = 1;
    float k_factor = 1.0f;
    bool use_precomputed = true;
    bool batched_scan = false;
    float recompute_threshold = 0.001f;
    float initial_search_fraction = 0.02f;
    int aps_flush_period_us = 100;
};

struct BuildTimingInfo {  // common.h:189-198
    int64_t n_vectors = 0, n_clusters = 0;
    int d = 0, num_codebooks = -1, code_size = -1;
    int train_time_us = 0, assign_time_us = 0, total_time_us = 0;
};

struct ModifyTimingInfo {  // common.h:203-209
    int64_t n_vectors = 0;
    int input_validation_time_us = 0, find_partition_time_us = 0, modify_time_us = 0, maintenance_time_us = 0;
};

struct SearchTimingInfo {  // common.h:214-228
    int64_t n_queries = 0, n_clusters = 0;
    int partitions_scanned = 0;
    shared_ptr<SearchParams> search_params = nullptr;
    shared_ptr<SearchTimingInfo> parent_info = nullptr;
    int64_t buffer_init_time_ns = 0, job_enqueue_time_ns = 0, boundary_distance_time_ns = 0, job_wait_time_ns = 0,
            result_aggregate_time_ns = 0, total_time_ns = 0;
};

struct MaintenanceTimingInfo {  // common.h:233-241
    int64_t n_splits = 0, n_deletes = 0, delete_time_us = 0, delete_refine_time_us = 0, split_time_us = 0,
            split_refine_time_us = 0, total_time_us = 0;
};

struct SearchResult {  // common.h:243-247
    Tensor ids;
    Tensor distances;
    shared_ptr<SearchTimingInfo> timing_info;
};

struct Clustering {  // common.h:249-276
    Tensor centroids;
    Tensor partition_ids;
    std::vector<Tensor> vectors;
    std::vector<Tensor> vector_ids;
    int64_t ntotal() const {
        int64_t n = 0;
        for (const auto &v : vectors)
            if (v.defined() && v.numel() > 0) n += v.size(0);
        return n;
    }
    int64_t nlist() const { return (int64_t)vectors.size(); }
    int64_t dim() const { return centroids.size(1); }
    int64_t cluster_size(int64_t i) const { return vectors[(size_t)i].size(0); }
};

int str_to_metric_type(std::string metric);  // common.h:145-156: "l2" -> 1, "ip" -> 0, else std::invalid_argument

// ---- plumbing shared by the mirror's translation units (not part of the reference surface) -------------------------------
void qk_check(int status);            // qk_status -> the exception type the reference throws (invalid_argument / runtime_error)
qk_ctx *qk_device_context(int device);  // one shared context per device
Tensor host_f32(const Tensor &t);
Tensor host_i64(const Tensor &t);

}  // namespace quake_amd
