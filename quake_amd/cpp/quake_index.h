// quake_index.h -- C++ host-side mirror of the reference's public surface for the hot path, on top of the C ABI
// (include/quake_hip.h).  Same class / field / method names as the reference (src/cpp/include/quake_index.h:18-142,
// src/cpp/include/common.h:104-247) so that C++ callers and the pybind11 module (bindings.cpp, mirroring
// src/cpp/bindings/wrap.cpp:48-368) see the interface they know; every method body is new: it marshals torch tensors
// to libquake_hip.so, where all arithmetic happens.  Bookkeeping kept here is what the reference keeps in
// PartitionManager (resident id set, next partition id).
#pragma once
#include <torch/torch.h>

#include <memory>
#include <set>
#include <string>
#include <unordered_set>
#include <vector>

#include "../../include/quake_hip.h"

using torch::Tensor;
using std::shared_ptr;

namespace quake_amd {

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

int str_to_metric_type(std::string metric);  // common.h:145-156: "l2" -> 1, "ip" -> 0, else std::invalid_argument

class QuakeIndex {  // quake_index.h:18-142
public:
    shared_ptr<QuakeIndex> parent_;
    int metric_ = QK_METRIC_L2;
    shared_ptr<IndexBuildParams> build_params_;
    shared_ptr<MaintenancePolicyParams> maintenance_policy_params_;
    int current_level_ = 0;
    bool debug_ = false;

    explicit QuakeIndex(int current_level = 0);
    ~QuakeIndex();
    QuakeIndex(const QuakeIndex &) = delete;
    QuakeIndex &operator=(const QuakeIndex &) = delete;

    shared_ptr<BuildTimingInfo> build(Tensor x, Tensor ids, shared_ptr<IndexBuildParams> build_params);
    shared_ptr<SearchResult> search(Tensor x, shared_ptr<SearchParams> search_params);
    Tensor get(Tensor ids);
    Tensor get_ids();
    shared_ptr<ModifyTimingInfo> add(Tensor x, Tensor ids);
    shared_ptr<ModifyTimingInfo> remove(Tensor ids);
    shared_ptr<ModifyTimingInfo> modify(Tensor ids, Tensor x);
    void initialize_maintenance_policy(shared_ptr<MaintenancePolicyParams> maintenance_policy_params);
    shared_ptr<MaintenanceTimingInfo> maintenance();
    void refine_partitions(Tensor partition_ids, int iterations);
    bool validate();
    void save(const std::string &path);
    void load(const std::string &path, int n_workers = 0);
    int64_t ntotal();
    int64_t nlist();
    int d();

    qk_store *store() { return store_; }

private:
    qk_ctx *ctx_ = nullptr;   // shared per-device context (not owned)
    qk_store *store_ = nullptr;
    int d_ = 0;
    std::unordered_set<int64_t> resident_;  // PartitionManager::resident_ids_
    int64_t next_pid_ = 0;                  // PartitionManager::curr_partition_id_
    void reset_store(int d);
    void require_built(const char *msg) const;
};

}  // namespace quake_amd
