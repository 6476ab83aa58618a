// quake_index.h -- QuakeIndex of the C++ host mirror: the public surface of the reference's
// src/cpp/include/quake_index.h:18-142 (same member and method names) on top of the C ABI (include/quake_hip.h).  Like the
// reference it is a facade over three collaborators it owns and exposes: partition_manager_ (the device partition store
// + its bookkeeping), query_coordinator_ (search) and maintenance_policy_.  Every method body is new: it marshals torch
// tensors to libquake_hip.so, where all arithmetic happens.
#pragma once
#include "common.h"
#include "list_scanning.h"
#include "maintenance_policies.h"
#include "partition_manager.h"
#include "query_coordinator.h"

namespace quake_amd {

class QuakeIndex : public std::enable_shared_from_this<QuakeIndex> {
public:
    shared_ptr<QuakeIndex> parent_;
    shared_ptr<PartitionManager> partition_manager_;
    shared_ptr<QueryCoordinator> query_coordinator_;
    shared_ptr<MaintenancePolicy> maintenance_policy_;
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
    // the reference never feeds its hit tracker from search(): with this switch on, search() records the
    // partitions every query probed, so maintenance() has a window to act on
    void set_track_hits(bool on);
    // scan-latency grid of the maintenance cost model from a CSV in the reference's profile format
    // (maintenance_cost_estimator.cpp:259-365): loaded if the file exists, else profiled on the device and saved there
    void set_latency_profile(const std::string &path);
    void publish();  // pending modifications visible to searches now, not inside the next query (qk_store_publish)
    bool validate();
    void save(const std::string &path);
    void load(const std::string &path, int n_workers = 0);
    int64_t ntotal();
    int64_t nlist();
    int d();

    qk_store *store() { return partition_manager_ ? partition_manager_->store() : nullptr; }

private:
    void require_built(const char *msg) const;
    void make_coordinator(int num_workers);
};

}  // namespace quake_amd
