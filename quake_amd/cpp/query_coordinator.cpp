// query_coordinator.cpp -- see query_coordinator.h.  No arithmetic here: tensors are marshalled to the C ABI.
#include "query_coordinator.h"

#include <c10/hip/HIPStream.h>

#include <chrono>
#include <cstring>
#include <stdexcept>

#include "maintenance_policies.h"
#include "partition_manager.h"
#include "quake_index.h"

namespace quake_amd {

namespace {
using clk = std::chrono::high_resolution_clock;
inline int64_t ns_since(clk::time_point t0) { return std::chrono::duration_cast<std::chrono::nanoseconds>(clk::now() - t0).count(); }

// Device tensors in: the library works on torch's CURRENT stream for the length of the call (the newly bound stream waits for
// the previous one through an event, qk_ctx_set_stream), so it is ordered behind whatever produced the inputs and the outputs are
// ordered in front of whatever torch enqueues next -- no device-wide synchronisation.  The context is shared (one per device):
// the binding it had on entry -- its private stream, the NULL stream, or a stream a caller bound it to -- comes back on every
// way out.
struct BoundToTorchStream {
    qk_ctx *c = nullptr;
    void *prev = nullptr;
    int prev_kind = 0;
    BoundToTorchStream(qk_ctx *ctx, const Tensor &t) {
        if (!ctx || !t.is_cuda()) return;
        qk_check(qk_ctx_get_stream(ctx, &prev, &prev_kind));
        c = ctx;
        hipStream_t st = c10::hip::getCurrentHIPStream(t.device().index()).stream();
        qk_check(st ? qk_ctx_set_stream(c, (void *)st) : qk_ctx_set_null_stream(c));
    }
    ~BoundToTorchStream() {
        if (!c) return;
        if (prev_kind == 1) (void)qk_ctx_set_null_stream(c);
        else (void)qk_ctx_set_stream(c, prev_kind == 2 ? prev : nullptr);
    }
};
}  // namespace

QueryCoordinator::QueryCoordinator(shared_ptr<QuakeIndex> parent, shared_ptr<PartitionManager> partition_manager,
                                   shared_ptr<MaintenancePolicy> maintenance_policy, MetricType metric, int num_workers)
    : partition_manager_(partition_manager), maintenance_policy_(maintenance_policy), parent_(parent), metric_(metric) {
    if (num_workers > 0) initialize_workers(num_workers);
}

QueryCoordinator::~QueryCoordinator() { shutdown_workers(); }

void QueryCoordinator::initialize_workers(int num_workers) {  // query_coordinator.cpp:50-74
    if (workers_initialized_) return;
    num_workers_ = num_workers;
    workers_initialized_ = num_workers > 0;
    // the reference starts one thread per worker and pins partition i to core i % num_workers; here the partitions move into a
    // device group (one member per worker, member j on GPU j % #GPUs) unless the manager was told the count before it was filled
    if (partition_manager_ && workers_initialized_) partition_manager_->distribute_partitions(num_workers);
}

void QueryCoordinator::shutdown_workers() {  // :77-95
    // the reference joins its threads and keeps serving scans serially; the partitions of a device group stay where they are
    // (they ARE the index) -- only the flag the reference's tests read goes down
    workers_initialized_ = false;
}

shared_ptr<SearchResult> QueryCoordinator::empty_result(shared_ptr<SearchParams> sp) const {  // :251-257, 476-482
    auto res = std::make_shared<SearchResult>();
    res->ids = torch::empty({0}, torch::kInt64);
    res->distances = torch::empty({0}, torch::kFloat32);
    res->timing_info = std::make_shared<SearchTimingInfo>();
    res->timing_info->search_params = sp;
    return res;
}

shared_ptr<SearchResult> QueryCoordinator::search(Tensor x, shared_ptr<SearchParams> sp) {  // :612-657
    if (!partition_manager_) throw std::runtime_error("[QueryCoordinator::search] partition_manager_ is null.");
    if (!x.defined() || x.size(0) == 0) return empty_result(sp);
    auto t0 = clk::now();
    qk_ctx *ctx = partition_manager_->ctx();
    qk_store *store = partition_manager_->store();
    qk_group *group = partition_manager_->group();  // partitions distributed over workers (GPUs)
    if (!store && !group) throw std::runtime_error("[QueryCoordinator::search] partitions are not initialized.");
    const bool on_dev = x.is_cuda();
    Tensor xq = on_dev ? x.to(torch::kFloat32).contiguous() : host_f32(x);
    // (workers: the parent search of the hit-tracking path runs on the shared context, the scan on the group's lead -- both are
    //  ordered on torch's stream)
    BoundToTorchStream bound(ctx, xq), bound_lead(group ? partition_manager_->search_ctx() : nullptr, xq);
    const int64_t Q = xq.size(0);
    const int k = sp->k > 0 ? sp->k : 1;  // :490
    const int nprobe = std::max(sp->nprobe, 1);
    auto res = std::make_shared<SearchResult>();
    auto ti = res->timing_info = std::make_shared<SearchTimingInfo>();
    ti->search_params = sp;
    ti->n_queries = Q;
    ti->n_clusters = partition_manager_->nlist();
    res->ids = torch::empty({Q, k}, torch::TensorOptions().dtype(torch::kInt64).device(xq.device()));
    res->distances = torch::empty({Q, k}, torch::TensorOptions().dtype(torch::kFloat32).device(xq.device()));
    qk_timing tm;
    std::memset(&tm, 0, sizeof(tm));
    const int mem = on_dev ? QK_MEM_DEVICE : QK_MEM_HOST;
    const bool track = maintenance_policy_ && maintenance_policy_->track_hits_ && parent_;
    // per-call phase timing for SearchTimingInfo (mode 1: the library reads its events behind a synchronisation of the stream -- which
    // a call that asks for the counters of SearchTimingInfo performs anyway), for host and device tensors alike: the reference always
    // fills the phase fields (query_coordinator.cpp:612-657).  The caller's mode on the (shared, one per device) context is put back
    // on every way out.
    struct TimingMode {
        qk_ctx *c = nullptr;
        int was = 0;
        TimingMode(qk_ctx *ctx, bool want) {
            if (!want || qk_ctx_get_timing(ctx, &was) != QK_OK || was == 1) return;
            if (qk_ctx_set_timing(ctx, 1) == QK_OK) c = ctx;
        }
        ~TimingMode() {
            if (c) (void)qk_ctx_set_timing(c, was);
        }
    } timing_mode(ctx, !on_dev || device_timing_);
    qk_timing *tmp = (!on_dev || device_timing_) ? &tm : nullptr;  // device tensors: asynchronous unless the caller opted in
    if (sp->recall_target > 0.0f && parent_ && !sp->batched_scan) {
        // adaptive partition scanning (:502,637-641): candidates = nlist * initial_search_fraction; with workers the rounds run on
        // the group's lead and every member scans the pairs whose partitions it holds (the APS hook of worker_scan, :364-428)
        Tensor nscan = torch::empty({Q}, torch::TensorOptions().dtype(torch::kInt32).device(xq.device()));
        if (group)
            qk_check(qk_group_search_aps(group, parent_->store(), xq.data_ptr<float>(), Q, k, (int)metric_, sp->recall_target,
                                         sp->recompute_threshold, sp->use_precomputed ? 1 : 0, sp->initial_search_fraction,
                                         res->ids.data_ptr<int64_t>(), res->distances.data_ptr<float>(), nscan.data_ptr<int32_t>(), mem, &tm));
        else
            qk_check(qk_search_aps(ctx, parent_->store(), store, xq.data_ptr<float>(), Q, k, (int)metric_, sp->recall_target,
                                   sp->recompute_threshold, sp->use_precomputed ? 1 : 0, sp->initial_search_fraction,
                                   res->ids.data_ptr<int64_t>(), res->distances.data_ptr<float>(), nscan.data_ptr<int32_t>(), mem, &tm));
        ti->partitions_scanned = (int)nscan.sum().item<int64_t>();
        ti->job_wait_time_ns = (int64_t)(tm.total_ms * 1e6);
    } else if (track) {
        // hit tracking on: the probed lists go to the policy.  One store: qk_search_tracked (the nearest-centroid step writes `pids`, the
        // scan reads it, one enqueue); a group: the coarse step on the shared context, then the members' scan
        const int kk = (int)std::min<int64_t>(nprobe, parent_->ntotal());
        Tensor pids = torch::empty({Q, std::max(kk, 1)}, torch::TensorOptions().dtype(torch::kInt64).device(xq.device()));
        if (group) {
            qk_check(qk_coarse(ctx, parent_->store(), xq.data_ptr<float>(), Q, nprobe, (int)metric_, pids.data_ptr<int64_t>(), nullptr, mem));
            qk_check(qk_group_scan(group, xq.data_ptr<float>(), Q, pids.data_ptr<int64_t>(), kk, k, (int)metric_,
                                   res->ids.data_ptr<int64_t>(), res->distances.data_ptr<float>(), mem, &tm));
        } else {
            qk_check(qk_search_tracked(ctx, parent_->store(), store, xq.data_ptr<float>(), Q, nprobe, k, (int)metric_,
                                       res->ids.data_ptr<int64_t>(), res->distances.data_ptr<float>(), pids.data_ptr<int64_t>(), mem, &tm));
        }
        if (kk < 1) pids = pids.slice(1, 0, 0);
        maintenance_policy_->record_query_batch_later(pids);  // (recorded before the next modification / maintenance: flush_hits)
        ti->partitions_scanned = (int)tm.partitions_scanned;
        ti->job_enqueue_time_ns = (int64_t)(tm.group_ms * 1e6);
        ti->job_wait_time_ns = (int64_t)(tm.scan_ms * 1e6);
        ti->result_aggregate_time_ns = (int64_t)(tm.merge_ms * 1e6);
    } else if (group) {  // workers: every member scans the partitions it holds (worker_scan, :243-469), the lead merges
        qk_check(qk_group_search(group, parent_->store(), xq.data_ptr<float>(), Q, nprobe, k, (int)metric_,
                                 res->ids.data_ptr<int64_t>(), res->distances.data_ptr<float>(), mem, tmp));
        ti->partitions_scanned = (int)tm.partitions_scanned;
        ti->job_wait_time_ns = (int64_t)(tm.scan_ms * 1e6);
        ti->result_aggregate_time_ns = (int64_t)(tm.merge_ms * 1e6);
    } else {
        qk_check(qk_search(ctx, parent_ ? parent_->store() : nullptr, store, xq.data_ptr<float>(), Q, nprobe, k, (int)metric_,
                           res->ids.data_ptr<int64_t>(), res->distances.data_ptr<float>(), mem, tmp));
        ti->partitions_scanned = (int)tm.partitions_scanned;
        ti->job_enqueue_time_ns = (int64_t)(tm.group_ms * 1e6);
        ti->job_wait_time_ns = (int64_t)(tm.scan_ms * 1e6);
        ti->result_aggregate_time_ns = (int64_t)(tm.merge_ms * 1e6);
    }
    if (parent_) {
        ti->parent_info = std::make_shared<SearchTimingInfo>();
        ti->parent_info->n_queries = Q;
        ti->parent_info->n_clusters = 1;
        ti->parent_info->total_time_ns = (int64_t)(tm.coarse_ms * 1e6);
    }
    ti->total_time_ns = ns_since(t0);
    return res;
}

shared_ptr<SearchResult> QueryCoordinator::scan_partitions(Tensor x, Tensor partition_ids, shared_ptr<SearchParams> sp) {  // :659-673
    if (!partition_manager_) throw std::runtime_error("[QueryCoordinator::scan_partitions] partition_manager_ is null.");
    if (!x.defined() || x.size(0) == 0) return empty_result(sp);
    auto t0 = clk::now();
    qk_store *store = partition_manager_->store();
    qk_group *group = partition_manager_->group();
    if (!store && !group) throw std::runtime_error("[QueryCoordinator::scan_partitions] partitions are not initialized.");
    // device queries stay on the device (the list numbers follow them there), host queries go through the staging buffers
    const bool on_dev = x.is_cuda();
    Tensor xq = on_dev ? x.to(torch::kFloat32).contiguous() : host_f32(x);
    BoundToTorchStream bound(partition_manager_->search_ctx(), xq);
    const int64_t Q = xq.size(0);
    const int k = sp->k > 0 ? sp->k : 1;
    Tensor pids = on_dev ? partition_ids.to(xq.device(), torch::kInt64).contiguous() : host_i64(partition_ids);
    if (pids.dim() == 1) pids = pids.unsqueeze(0).expand({Q, pids.size(0)}).contiguous();  // the same set for every query (:506-508)
    const int P = (int)pids.size(1);
    auto res = std::make_shared<SearchResult>();
    auto ti = res->timing_info = std::make_shared<SearchTimingInfo>();
    ti->search_params = sp;
    ti->n_queries = Q;
    ti->n_clusters = partition_manager_->nlist();
    res->ids = torch::empty({Q, k}, torch::TensorOptions().dtype(torch::kInt64).device(xq.device()));
    res->distances = torch::empty({Q, k}, torch::TensorOptions().dtype(torch::kFloat32).device(xq.device()));
    qk_timing tm;
    std::memset(&tm, 0, sizeof(tm));
    Tensor none = torch::full({Q, 1}, -1, torch::TensorOptions().dtype(torch::kInt64).device(xq.device()));  // zero partitions: padded output (:459-497)
    const Tensor &pp = P > 0 ? pids : none;
    qk_timing *tmp = (!on_dev || device_timing_) ? &tm : nullptr;  // (as in search: device tensors stay asynchronous)
    if (group)
        qk_check(qk_group_scan(group, xq.data_ptr<float>(), Q, pp.data_ptr<int64_t>(), (int)pp.size(1), k, (int)metric_,
                               res->ids.data_ptr<int64_t>(), res->distances.data_ptr<float>(), on_dev ? QK_MEM_DEVICE : QK_MEM_HOST, tmp));
    else
        qk_check(qk_scan(partition_manager_->ctx(), store, xq.data_ptr<float>(), Q, pp.data_ptr<int64_t>(), (int)pp.size(1), k, (int)metric_,
                         res->ids.data_ptr<int64_t>(), res->distances.data_ptr<float>(), on_dev ? QK_MEM_DEVICE : QK_MEM_HOST, tmp));
    ti->partitions_scanned = (int)tm.partitions_scanned;
    ti->total_time_ns = ns_since(t0);
    return res;
}

shared_ptr<SearchResult> QueryCoordinator::serial_scan(Tensor x, Tensor partition_ids, shared_ptr<SearchParams> sp) {
    return scan_partitions(x, partition_ids, sp);
}
shared_ptr<SearchResult> QueryCoordinator::batched_serial_scan(Tensor x, Tensor partition_ids, shared_ptr<SearchParams> sp) {
    return scan_partitions(x, partition_ids, sp);
}
shared_ptr<SearchResult> QueryCoordinator::worker_scan(Tensor x, Tensor partition_ids, shared_ptr<SearchParams> sp) {
    return scan_partitions(x, partition_ids, sp);
}

}  // namespace quake_amd
