// maintenance_policies.h -- the maintenance policy of the C++ host mirror: hit tracking, cost model, split / delete / local
// refinement decisions.  Counterparts (names and public methods kept) of
//   HitCountTracker            src/cpp/include/hit_count_tracker.h, src/cpp/src/hit_count_tracker.cpp:3-98
//   ListScanLatencyEstimator   src/cpp/src/maintenance_cost_estimator.cpp:21-365 (latency grid, bilinear inter/extrapolation)
//   MaintenanceCostEstimator   src/cpp/src/maintenance_cost_estimator.cpp:368-498 (split / delete deltas)
//   MaintenancePolicy          src/cpp/src/maintenance_policies.cpp:18-202
// This is scalar host bookkeeping; every data-parallel step it triggers runs on the device through PartitionManager
// (coarse step, 2-means split, reassignment, refinement).  Two deliberate differences from the reference snapshot
// (SURVEY.md 8f-4): search() can record the partitions each query scanned (`track_hits_`; the reference declares
// record_query_hits but never calls it, so its maintenance() cannot act), and the latency grid is profiled on the DEVICE
// scan in the throughput regime (time of one qk_scan over many (query, n-row partition) pairs / pairs) unless a profile
// function is injected.
#pragma once
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "common.h"

namespace quake_amd {

class QuakeIndex;
class PartitionManager;

class HitCountTracker {
public:
    HitCountTracker(int window_size, int total_vectors);
    void reset();
    void set_total_vectors(int total_vectors);
    void add_query_data(const std::vector<int64_t> &hit_partition_ids, const std::vector<int64_t> &scanned_sizes);
    float get_current_scan_fraction() const { return current_scan_fraction_; }
    const std::vector<std::vector<int64_t>> &get_per_query_hits() const { return per_query_hits_; }
    const std::vector<std::vector<int64_t>> &get_per_query_scanned_sizes() const { return per_query_scanned_sizes_; }
    int get_window_size() const { return window_size_; }
    int64_t get_num_queries_recorded() const { return num_queries_recorded_; }
    std::map<int64_t, int> aggregated_hits() const;  // partition id -> window queries that scanned it

private:
    int window_size_, total_vectors_;
    int curr_query_index_ = 0;
    int64_t num_queries_recorded_ = 0;
    float running_sum_scan_fraction_ = 0.0f, current_scan_fraction_ = 1.0f;
    std::vector<std::vector<int64_t>> per_query_hits_, per_query_scanned_sizes_;
    float fraction(const std::vector<int64_t> &sizes) const;
};

using ScanProfileFn = std::function<double(int n, int k)>;  // ns to scan one n-row partition for one query with top-k

class ListScanLatencyEstimator {
public:
    ListScanLatencyEstimator(int d, const std::vector<int> &n_values, const std::vector<int> &k_values, int n_trials = 5,
                             bool adaptive_nprobe = false, const std::string &profile_filename = "", ScanProfileFn profile_fn = nullptr);
    void profile_scan_latency(ScanProfileFn fn = nullptr);
    double estimate_scan_latency(int n, int k) const;
    // the smallest grid value n0 from which the modelled latency is nondecreasing in n at this k (along the grid in both k columns
    // that bracket k, hence for every interpolated / extrapolated n >= n0), or -1 when it never is / k lies beyond the grid
    int monotone_from(int k) const;
    void set_scan_latency(int n, int k, double latency_ns);
    bool save_latency_profile(const std::string &filename) const;
    bool load_latency_profile(const std::string &filename);
    int d_;
    std::vector<int> n_values_, k_values_;
    std::vector<std::vector<double>> scan_latency_model_;
    int n_trials_;
    std::string profile_filename_;
};

ScanProfileFn device_profile_fn(int d, int n_trials = 5);
// the default grid (common.h:97-99) with its latency model read from / written to `profile_filename`
shared_ptr<ListScanLatencyEstimator> default_latency_estimator(int d, const std::string &profile_filename);

class MaintenanceCostEstimator {
public:
    MaintenanceCostEstimator(int d, float alpha, int k, shared_ptr<ListScanLatencyEstimator> latency_estimator = nullptr,
                             ScanProfileFn profile_fn = nullptr);
    double compute_split_delta(int partition_size, float hit_rate, int total_partitions) const;
    double compute_delete_delta(int partition_size, float hit_rate, int total_partitions, float avg_partition_hit_rate,
                                float avg_partition_size) const;
    double compute_delete_delta_w_reassign(int partition_size, float hit_rate, int total_partitions, const std::vector<int64_t> &reassign_counts,
                                           const std::vector<int64_t> &reassign_sizes, const std::vector<float> &reassign_hit_rates) const;
    shared_ptr<ListScanLatencyEstimator> get_latency_estimator() const { return latency_estimator_; }
    int get_k() const { return k_; }

private:
    int d_;
    float alpha_;
    int k_;
    shared_ptr<ListScanLatencyEstimator> latency_estimator_;
};

class MaintenancePolicy {
public:
    MaintenancePolicy(shared_ptr<PartitionManager> partition_manager, shared_ptr<MaintenancePolicyParams> params,
                      shared_ptr<MaintenanceCostEstimator> cost_estimator = nullptr);
    shared_ptr<MaintenanceTimingInfo> perform_maintenance();
    void record_query_hits(std::vector<int64_t> partition_ids);
    void record_query_batch(const Tensor &partition_ids);  // [Q, P] host tensor, -1 = none: one record_query_hits per row
    // the same, LATER: a tracked search hands its [Q, P] list numbers over (host or device tensor) and returns -- no device
    // synchronisation, no window bookkeeping inside the search; flush_hits() records everything pending, in order.  Called before
    // anything that changes a partition's size or reads the window (add / remove / refine / maintenance), so every hit is credited
    // with the size its partition had when it was scanned -- as if it had been recorded inside search().
    void record_query_batch_later(const Tensor &partition_ids);
    void flush_hits();
    void reset();
    void local_refinement(const Tensor &partition_ids);
    bool track_hits_ = false;  // QueryCoordinator::search records the probed partitions when set
    shared_ptr<HitCountTracker> hit_count_tracker_;
    shared_ptr<MaintenanceCostEstimator> cost_estimator_;
    shared_ptr<MaintenancePolicyParams> params_;

private:
    shared_ptr<PartitionManager> partition_manager_;
    std::vector<Tensor> pending_hits_;
    void ensure_cost_estimator();
};

}  // namespace quake_amd
