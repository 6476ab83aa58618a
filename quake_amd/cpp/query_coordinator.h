// query_coordinator.h -- QueryCoordinator of the C++ host mirror: the public surface of the reference's
// src/cpp/include/query_coordinator.h:45-179.  search() = parent search for the nprobe nearest partitions, then
// scan_partitions(); on the device the three scan variants of the reference (serial_scan, batched_serial_scan, worker_scan:
// three ways of spreading the same work over CPU threads) are ONE pipeline -- qk_search / qk_scan of libquake_hip.so -- and
// return the same result, so all three names enqueue it.  A WORKER IS A GPU: initialize_workers(n) distributes the partitions
// over a device group of n members (PartitionManager::distribute_partitions -> qk_group, partition p on member p % n) and from
// then on search / scan_partitions / worker_scan run on all members at once (qk_group_search / qk_group_scan); results -- ids and
// distance bits -- equal the one-device pipeline's.
#pragma once
#include "common.h"

namespace quake_amd {

class QuakeIndex;
class PartitionManager;
class MaintenancePolicy;

class QueryCoordinator {
public:
    shared_ptr<PartitionManager> partition_manager_;
    shared_ptr<MaintenancePolicy> maintenance_policy_;
    shared_ptr<QuakeIndex> parent_;
    MetricType metric_;
    bool workers_initialized_ = false;
    int num_workers_ = 0;
    bool debug_ = false;
    // Device (CUDA) tensors in -> device tensors out, enqueued on torch's current stream with NO host synchronisation: the counters
    // and phase times of SearchTimingInfo need one (they are read back from the device), so for device tensors they are filled only
    // when this is set -- host-tensor searches synchronise anyway and always fill them (query_coordinator.cpp:612-657 always does: its
    // scan is host code).  Recall-target searches and hit tracking read their counters in either case.
    bool device_timing_ = false;

    QueryCoordinator(shared_ptr<QuakeIndex> parent, shared_ptr<PartitionManager> partition_manager,
                     shared_ptr<MaintenancePolicy> maintenance_policy, MetricType metric, int num_workers = 0);
    ~QueryCoordinator();

    shared_ptr<SearchResult> search(Tensor x, shared_ptr<SearchParams> search_params);
    shared_ptr<SearchResult> scan_partitions(Tensor x, Tensor partition_ids, shared_ptr<SearchParams> search_params);
    shared_ptr<SearchResult> serial_scan(Tensor x, Tensor partition_ids, shared_ptr<SearchParams> search_params);
    shared_ptr<SearchResult> batched_serial_scan(Tensor x, Tensor partition_ids, shared_ptr<SearchParams> search_params);
    shared_ptr<SearchResult> worker_scan(Tensor x, Tensor partition_ids, shared_ptr<SearchParams> search_params);
    void initialize_workers(int num_workers);
    void shutdown_workers();

private:
    shared_ptr<SearchResult> empty_result(shared_ptr<SearchParams> sp) const;
};

}  // namespace quake_amd
