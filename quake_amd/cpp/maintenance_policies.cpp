// maintenance_policies.cpp -- see maintenance_policies.h.
#include "maintenance_policies.h"

#include <cmath>
#include <cstdio>
#include <set>
#include <cstdlib>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <fstream>
#include <sstream>
#include <stdexcept>

#include "partition_manager.h"
#include "quake_index.h"

namespace quake_amd {

namespace {
using clk = std::chrono::high_resolution_clock;
inline int64_t us_since(clk::time_point t0) { return std::chrono::duration_cast<std::chrono::microseconds>(clk::now() - t0).count(); }
const std::vector<int> kDefaultRangeN = {1, 2, 4, 16, 64, 256, 1024, 4096, 16384, 65536};  // common.h:97
const std::vector<int> kDefaultRangeK = {1, 4, 16, 64, 256};                                // common.h:98
}  // namespace

// ---- HitCountTracker --------------------------------------------------------------------------------------------------
HitCountTracker::HitCountTracker(int window_size, int total_vectors) : window_size_(window_size), total_vectors_(total_vectors) {
    if (window_size <= 0) throw std::invalid_argument("Window size must be positive");
    if (total_vectors <= 0) throw std::invalid_argument("Total vectors must be positive");
    reset();
}

void HitCountTracker::reset() {
    curr_query_index_ = 0;
    num_queries_recorded_ = 0;
    running_sum_scan_fraction_ = 0.0f;
    current_scan_fraction_ = 1.0f;
    per_query_hits_.assign((size_t)window_size_, {});
    per_query_scanned_sizes_.assign((size_t)window_size_, {});
}

void HitCountTracker::set_total_vectors(int total_vectors) {
    if (total_vectors <= 0) throw std::invalid_argument("Total vectors must be positive");
    total_vectors_ = total_vectors;
}

float HitCountTracker::fraction(const std::vector<int64_t> &sizes) const {
    int64_t sum = 0;
    for (int64_t s : sizes) sum += s;
    return (float)sum / (float)total_vectors_;
}

void HitCountTracker::add_query_data(const std::vector<int64_t> &hits, const std::vector<int64_t> &sizes) {
    if (hits.size() != sizes.size()) throw std::invalid_argument("hit_partition_ids and scanned_sizes must be of equal length");
    const float frac = fraction(sizes);
    if (num_queries_recorded_ < window_size_) {
        per_query_hits_[(size_t)num_queries_recorded_] = hits;
        per_query_scanned_sizes_[(size_t)num_queries_recorded_] = sizes;
        running_sum_scan_fraction_ += frac;
        num_queries_recorded_++;
    } else {
        running_sum_scan_fraction_ -= fraction(per_query_scanned_sizes_[(size_t)curr_query_index_]);
        per_query_hits_[(size_t)curr_query_index_] = hits;
        per_query_scanned_sizes_[(size_t)curr_query_index_] = sizes;
        running_sum_scan_fraction_ += frac;
        curr_query_index_ = (curr_query_index_ + 1) % window_size_;
    }
    const int64_t eff = std::min<int64_t>(num_queries_recorded_, window_size_);
    current_scan_fraction_ = running_sum_scan_fraction_ / (float)eff;
}

std::map<int64_t, int> HitCountTracker::aggregated_hits() const {  // maintenance_policies.cpp:45-51
    std::map<int64_t, int> out;
    const int64_t eff = std::min<int64_t>(num_queries_recorded_, window_size_);
    for (int64_t i = 0; i < eff; i++)
        for (int64_t p : per_query_hits_[(size_t)i]) out[p]++;
    return out;
}

// ---- ListScanLatencyEstimator -----------------------------------------------------------------------------------------
ListScanLatencyEstimator::ListScanLatencyEstimator(int d, const std::vector<int> &n_values, const std::vector<int> &k_values, int n_trials,
                                                   bool, const std::string &profile_filename, ScanProfileFn profile_fn)
    : d_(d), n_values_(n_values), k_values_(k_values), n_trials_(n_trials), profile_filename_(profile_filename) {
    if (!std::is_sorted(n_values_.begin(), n_values_.end())) throw std::runtime_error("n_values must be sorted in ascending order.");
    if (!std::is_sorted(k_values_.begin(), k_values_.end())) throw std::runtime_error("k_values must be sorted in ascending order.");
    scan_latency_model_.assign(n_values_.size(), std::vector<double>(k_values_.size(), 0.0));
    const bool loaded = !profile_filename.empty() && load_latency_profile(profile_filename);
    if (!loaded) {
        profile_scan_latency(profile_fn);
        if (!profile_filename.empty()) save_latency_profile(profile_filename);
    }
}

void ListScanLatencyEstimator::profile_scan_latency(ScanProfileFn fn) {
    if (!fn) fn = device_profile_fn(d_, n_trials_);
    for (size_t i = 0; i < n_values_.size(); i++)
        for (size_t j = 0; j < k_values_.size(); j++) scan_latency_model_[i][j] = fn(n_values_[i], k_values_[j]);
}

void ListScanLatencyEstimator::set_scan_latency(int n, int k, double latency_ns) {
    const auto in = std::find(n_values_.begin(), n_values_.end(), n), ik = std::find(k_values_.begin(), k_values_.end(), k);
    if (in == n_values_.end() || ik == k_values_.end()) throw std::out_of_range("set_scan_latency: (n, k) is not a grid node");
    scan_latency_model_[(size_t)(in - n_values_.begin())][(size_t)(ik - k_values_.begin())] = latency_ns;
}

namespace {
// (lower index, upper index, fraction, inside) along one axis; beyond the grid the fraction is measured from the LAST node in
// units of the last interval (maintenance_cost_estimator.cpp:143-190)
struct AxisPos {
    size_t lo, hi;
    double t;
    bool inside;
};
AxisPos axis_pos(const std::vector<int> &v, int x) {
    const size_t n = v.size();
    if (x <= v.back()) {
        const size_t it = (size_t)(std::upper_bound(v.begin(), v.end(), x) - v.begin());
        if (it == n) return {n - 2, n - 1, 1.0, true};
        return {it - 1, it, (double)(x - v[it - 1]) / (double)(v[it] - v[it - 1]), true};
    }
    return {n - 2, n - 1, (double)(x - v[n - 1]) / (double)(v[n - 1] - v[n - 2]), false};
}
inline double extrap(double f1, double f2, double t) { return f2 + t * (f2 - f1); }  // slope of the last interval
}  // namespace

double ListScanLatencyEstimator::estimate_scan_latency(int n, int k) const {
    if (n == 0 || k == 0) return 0.0;
    if (n < n_values_.front() || k < k_values_.front()) throw std::out_of_range("n or k is below the minimum supported values.");
    const AxisPos a = axis_pos(n_values_, n), b = axis_pos(k_values_, k);
    const auto &m = scan_latency_model_;
    const double f11 = m[a.lo][b.lo], f12 = m[a.lo][b.hi], f21 = m[a.hi][b.lo], f22 = m[a.hi][b.hi];
    const double t = a.t, u = b.t;
    if (a.inside && b.inside) return (1 - t) * (1 - u) * f11 + t * (1 - u) * f21 + (1 - t) * u * f12 + t * u * f22;
    if (!a.inside && b.inside) return (1 - u) * extrap(f11, f21, t) + u * extrap(f12, f22, t);
    if (a.inside && !b.inside) return (1 - t) * extrap(f11, f12, u) + t * extrap(f21, f22, u);
    return extrap(extrap(f11, f21, t), extrap(f12, f22, t), u);
}

int ListScanLatencyEstimator::monotone_from(int k) const {
    if (k < k_values_.front() || n_values_.size() < 2) return -1;
    const AxisPos b = axis_pos(k_values_, k);
    if (!b.inside) return -1;
    const auto &m = scan_latency_model_;
    size_t i0 = n_values_.size() - 1;
    if (m[i0][b.lo] < m[i0 - 1][b.lo] || m[i0][b.hi] < m[i0 - 1][b.hi]) return -1;
    while (i0 > 0 && m[i0][b.lo] >= m[i0 - 1][b.lo] && m[i0][b.hi] >= m[i0 - 1][b.hi]) i0--;
    return n_values_[i0];
}

// the reference's CSV layout (maintenance_cost_estimator.cpp:259-365): header, "n_size,k_size", n values, k values, rows
bool ListScanLatencyEstimator::save_latency_profile(const std::string &filename) const {
    std::ofstream f(filename);
    if (!f.is_open()) return false;
    auto join = [&](const std::vector<int> &v) {
        std::ostringstream s;
        for (size_t i = 0; i < v.size(); i++) s << (i ? "," : "") << v[i];
        return s.str();
    };
    f << "n_size,k_size\n" << n_values_.size() << "," << k_values_.size() << "\n" << join(n_values_) << "\n" << join(k_values_) << "\n";
    f.precision(17);
    for (const auto &row : scan_latency_model_) {
        for (size_t j = 0; j < row.size(); j++) f << (j ? "," : "") << row[j];
        f << "\n";
    }
    return true;
}

bool ListScanLatencyEstimator::load_latency_profile(const std::string &filename) {
    std::ifstream f(filename);
    if (!f.is_open()) return false;
    std::vector<std::string> lines;
    for (std::string ln; std::getline(f, ln);) lines.push_back(ln);
    auto split = [](const std::string &s) {
        std::vector<double> out;
        std::stringstream ss(s);
        for (std::string tok; std::getline(ss, tok, ',');) {
            try {
                out.push_back(std::stod(tok));
            } catch (...) {
                return std::vector<double>();
            }
        }
        return out;
    };
    if (lines.size() < 4) return false;
    const auto dims = split(lines[1]), nv = split(lines[2]), kv = split(lines[3]);
    if (dims.size() != 2 || nv.size() != n_values_.size() || kv.size() != k_values_.size()) return false;
    if ((size_t)dims[0] != n_values_.size() || (size_t)dims[1] != k_values_.size()) return false;
    for (size_t i = 0; i < nv.size(); i++)
        if ((int)nv[i] != n_values_[i]) return false;
    for (size_t i = 0; i < kv.size(); i++)
        if ((int)kv[i] != k_values_[i]) return false;
    if (lines.size() < 4 + n_values_.size()) return false;
    std::vector<std::vector<double>> model;
    for (size_t i = 0; i < n_values_.size(); i++) {
        auto row = split(lines[4 + i]);
        if (row.size() != k_values_.size()) return false;
        model.push_back(row);
    }
    scan_latency_model_ = model;
    return true;
}

// device scan in the throughput regime: many n-row partitions scanned in one qk_scan call, one query each (up to 1024 pairs,
// at most 2^22 rows in all); the cost of one (query, partition) pair = the call's time / pairs
ScanProfileFn device_profile_fn(int d, int n_trials) {
    struct State {
        qk_store *store = nullptr;
        int n = -1, npart = 0;
        Tensor q, pids;
        ~State() {
            if (store) qk_store_destroy(store);
        }
    };
    auto st = std::make_shared<State>();
    return [st, d, n_trials](int n, int k) -> double {
        qk_ctx *ctx = qk_device_context(0);
        if (st->n != n) {
            if (st->store) qk_store_destroy(st->store);
            st->store = nullptr;
            const int npart = (int)std::max<int64_t>(16, std::min<int64_t>(1024, ((int64_t)1 << 22) / std::max(n, 1)));
            qk_check(qk_store_create(ctx, d, &st->store));
            Tensor off = torch::arange(npart + 1, torch::kInt64) * n, ids = torch::arange((int64_t)npart * n, torch::kInt64);
            Tensor v = torch::rand({(int64_t)npart * n, d}, torch::kFloat32);
            qk_check(qk_store_build_csr(st->store, npart, off.data_ptr<int64_t>(), ids.data_ptr<int64_t>(), v.data_ptr<float>(), QK_MEM_HOST));
            st->n = n;
            st->npart = npart;
            st->q = torch::rand({npart, d}, torch::kFloat32);
            st->pids = torch::arange(npart, torch::kInt64).reshape({npart, 1}).contiguous();
        }
        const int kk = std::min(k, QK_MAX_K);
        Tensor oi = torch::empty({st->npart, kk}, torch::kInt64), od = torch::empty({st->npart, kk}, torch::kFloat32);
        auto once = [&]() {
            qk_check(qk_scan(ctx, st->store, st->q.data_ptr<float>(), st->npart, st->pids.data_ptr<int64_t>(), 1, kk, QK_METRIC_L2,
                             oi.data_ptr<int64_t>(), od.data_ptr<float>(), QK_MEM_HOST, nullptr));
        };
        once();
        auto t0 = clk::now();
        for (int i = 0; i < std::max(n_trials, 1); i++) once();
        const double ns = (double)std::chrono::duration_cast<std::chrono::nanoseconds>(clk::now() - t0).count();
        return ns / std::max(n_trials, 1) / st->npart;
    };
}

shared_ptr<ListScanLatencyEstimator> default_latency_estimator(int d, const std::string &profile_filename) {
    return std::make_shared<ListScanLatencyEstimator>(d, kDefaultRangeN, kDefaultRangeK, 5, false, profile_filename);
}

// ---- MaintenanceCostEstimator -----------------------------------------------------------------------------------------
MaintenanceCostEstimator::MaintenanceCostEstimator(int d, float alpha, int k, shared_ptr<ListScanLatencyEstimator> lat, ScanProfileFn fn)
    : d_(d), alpha_(alpha), k_(k), latency_estimator_(lat) {
    if (k <= 0) throw std::invalid_argument("k must be positive");
    if (alpha <= 0.0f) throw std::invalid_argument("alpha must be positive");
    if (!latency_estimator_) latency_estimator_ = std::make_shared<ListScanLatencyEstimator>(d, kDefaultRangeN, kDefaultRangeK, 5, false, "", fn);
}

double MaintenanceCostEstimator::compute_split_delta(int partition_size, float hit_rate, int total_partitions) const {  // :384-394
    const auto &L = *latency_estimator_;
    const double delta_overhead = L.estimate_scan_latency(total_partitions + 1, k_) - L.estimate_scan_latency(total_partitions, k_);
    const double old_cost = L.estimate_scan_latency(partition_size, k_) * hit_rate;
    const double new_cost = L.estimate_scan_latency(partition_size / 2, k_) * hit_rate * (2.0 * alpha_);
    return delta_overhead + new_cost - old_cost;
}

double MaintenanceCostEstimator::compute_delete_delta(int partition_size, float hit_rate, int total_partitions, float avg_hit_rate,
                                                      float avg_size) const {  // :396-447
    if (total_partitions <= 1) return 0.0;
    const auto &L = *latency_estimator_;
    const int k = k_, T = total_partitions;
    const double delta_overhead = L.estimate_scan_latency(T - 1, k) - L.estimate_scan_latency(T, k);
    const double cost_old = (T - 1) * avg_hit_rate * L.estimate_scan_latency((int)avg_size, k) + hit_rate * L.estimate_scan_latency(partition_size, k);
    const double merged_size = avg_size + (double)partition_size / (T - 1);
    const double merged_hit_rate = avg_hit_rate + hit_rate / (double)(T - 1);
    double cost_new;
    if (partition_size < T)
        cost_new = partition_size * merged_hit_rate * L.estimate_scan_latency((int)(avg_size + 1), k) +
                   (T - partition_size - 1) * merged_hit_rate * L.estimate_scan_latency((int)avg_size, k);
    else
        cost_new = (T - 1) * merged_hit_rate * L.estimate_scan_latency((int)std::ceil(merged_size), k);
    return delta_overhead + (cost_new - cost_old);
}

double MaintenanceCostEstimator::compute_delete_delta_w_reassign(int partition_size, float hit_rate, int total_partitions,
                                                                 const std::vector<int64_t> &counts, const std::vector<int64_t> &sizes,
                                                                 const std::vector<float> &hit_rates) const {  // :449-493
    if (total_partitions <= 1) return 0.0;
    if (sizes.size() != counts.size() || sizes.size() != hit_rates.size()) throw std::invalid_argument("reassign vectors disagree in length");
    const auto &L = *latency_estimator_;
    const double delta_overhead = L.estimate_scan_latency(total_partitions - 1, k_) - L.estimate_scan_latency(total_partitions, k_);
    const double removal_delta = hit_rate * L.estimate_scan_latency(partition_size, k_);
    double reassign_delta = 0.0;
    for (size_t i = 0; i < sizes.size(); i++) {
        const double old = hit_rates[i] * L.estimate_scan_latency((int)sizes[i], k_);
        reassign_delta += (hit_rates[i] + hit_rate) * L.estimate_scan_latency((int)(sizes[i] + partition_size), k_) - old;
    }
    return delta_overhead + removal_delta + reassign_delta;
}

// ---- MaintenancePolicy ------------------------------------------------------------------------------------------------
MaintenancePolicy::MaintenancePolicy(shared_ptr<PartitionManager> pm, shared_ptr<MaintenancePolicyParams> params,
                                     shared_ptr<MaintenanceCostEstimator> cost_estimator)
    : cost_estimator_(cost_estimator), params_(params), partition_manager_(pm) {
    hit_count_tracker_ = std::make_shared<HitCountTracker>(params->window_size, (int)std::max<int64_t>(pm ? pm->ntotal() : 1, 1));
}

void MaintenancePolicy::ensure_cost_estimator() {  // built on first use: profiling the device scan takes a moment
    if (!cost_estimator_) cost_estimator_ = std::make_shared<MaintenanceCostEstimator>(partition_manager_->d(), params_->alpha, 10);
}

void MaintenancePolicy::record_query_hits(std::vector<int64_t> partition_ids) {
    hit_count_tracker_->add_query_data(partition_ids, partition_manager_->get_partition_sizes(partition_ids));
}

void MaintenancePolicy::record_query_batch(const Tensor &partition_ids) {
    Tensor p = host_i64(partition_ids);
    if (p.dim() == 1) p = p.unsqueeze(0);
    p = p.contiguous();
    const int64_t *pp = p.data_ptr<int64_t>();  // (plain reads: an element access through the tensor API is ~0.6 us, 8192 of them per batch)
    const int64_t P_ = p.size(1);
    std::map<int64_t, int64_t> size_of;
    for (int64_t i = 0; i < p.size(0); i++) {
        std::vector<int64_t> hits, sizes;
        for (int64_t j = 0; j < P_; j++) {
            const int64_t pid = pp[i * P_ + j];
            if (pid < 0) continue;
            auto it = size_of.find(pid);
            if (it == size_of.end()) it = size_of.emplace(pid, partition_manager_->get_partition_size(pid)).first;
            hits.push_back(pid);
            sizes.push_back(it->second);
        }
        hit_count_tracker_->add_query_data(hits, sizes);
    }
}

void MaintenancePolicy::record_query_batch_later(const Tensor &partition_ids) {
    pending_hits_.push_back(partition_ids);
    if (pending_hits_.size() >= 64) flush_hits();
}

void MaintenancePolicy::flush_hits() {
    std::vector<Tensor> pend;
    pend.swap(pending_hits_);
    for (const Tensor &t : pend) record_query_batch(host_i64(t));
}

void MaintenancePolicy::reset() {
    pending_hits_.clear();
    hit_count_tracker_->reset();
}

shared_ptr<MaintenanceTimingInfo> MaintenancePolicy::perform_maintenance() {  // maintenance_policies.cpp:33-177
    auto info = std::make_shared<MaintenanceTimingInfo>();
    auto &pm = *partition_manager_;
    auto &p = *params_;
    auto &tr = *hit_count_tracker_;
    flush_hits();
    if (tr.get_num_queries_recorded() < p.window_size) return info;  // :36-41 window not full yet
    if (!pm.parent_) return info;                                    // a flat index has nothing to split or delete
    ensure_cost_estimator();
    auto t_total = clk::now();
    static const bool trace = getenv("QUAKE_MAINTENANCE_TRACE") != nullptr;  // stage times of a call on stderr (scripts/dynamic_workload.py)
    auto t_stage = clk::now();
    auto stage = [&](const char *what) {
        if (trace) fprintf(stderr, "[maintenance] %-22s %8.2f ms\n", what, (double)us_since(t_stage) / 1e3);
        t_stage = clk::now();
    };
    const auto hits = tr.aggregated_hits();
    stage("aggregated hits");
    Tensor all_pids = pm.get_partition_ids();
    const int total_partitions = (int)pm.nlist();
    const float scan_fraction = tr.get_current_scan_fraction();
    const float avg_size = (float)(pm.ntotal() / std::max(total_partitions, 1));
    std::map<int64_t, int64_t> sizes;
    {
        Tensor ap = all_pids.contiguous();
        const int64_t *ip = ap.data_ptr<int64_t>();
        for (int64_t i = 0; i < ap.size(0); i++) sizes[ip[i]] = pm.get_partition_size(ip[i]);
    }
    auto hit_rate_of = [&](int64_t pid) {
        auto it = hits.find(pid);
        return (float)(it == hits.end() ? 0 : it->second) / (float)p.window_size;
    };
    std::vector<int64_t> to_delete, to_split;
    const auto &ce = *cost_estimator_;
    // where would a delete candidate's vectors go?  the nearest OTHER centroid of every vector (:79-101) -- asked for ALL the
    // candidates the rejection rule examines at once: their lists are extracted on the device, one nearest-two search per chunk of
    // rows, counted per candidate from plain arrays (two host round trips and 5000 tensor element reads per candidate before: a 50M
    // index has hundreds of candidates per call)
    // Which of the delete-branch partitions need the search at all?  The rule deletes when (overhead + hit_rate L(size)) + [a sum whose
    // every term is >= 0 where L is nondecreasing] < -threshold: a candidate whose first bracket is already >= -threshold (beyond a guard
    // band for the last-bit wobble of the interpolation) is KEPT whatever its targets are, and is not searched -- the large, hot partitions,
    // two thirds of the candidates of a 50M index.  Taken only when every partition's size lies where the grid is nondecreasing.
    std::set<int64_t> kept_early;
    bool shortcut = total_partitions > 1;
    {
        const int n0 = ce.get_latency_estimator()->monotone_from(ce.get_k());
        if (n0 < 0) shortcut = false;
        for (const auto &kv : sizes)
            if (kv.second > 0 && kv.second < n0) shortcut = false;
    }
    const auto &Lat = *ce.get_latency_estimator();
    const double d_overhead = total_partitions > 1 ? Lat.estimate_scan_latency(total_partitions - 1, ce.get_k()) - Lat.estimate_scan_latency(total_partitions, ce.get_k()) : 0.0;
    std::vector<int64_t> cand;
    for (const auto &kv : sizes) {
        const float hr = hit_rate_of(kv.first);
        const double dd = ce.compute_delete_delta((int)kv.second, hr, total_partitions, scan_fraction, avg_size);
        if (dd < -p.delete_threshold_ns && p.enable_delete_rejection && (int)kv.second > p.min_partition_size) {
            if (shortcut) {
                const double bracket = d_overhead + hr * Lat.estimate_scan_latency((int)kv.second, ce.get_k());
                const double guard = 1e-9 * (std::fabs(bracket) + std::fabs((double)p.delete_threshold_ns) + 1.0);
                if (bracket >= -(double)p.delete_threshold_ns + guard) {
                    kept_early.insert(kv.first);
                    continue;
                }
            }
            cand.push_back(kv.first);
        }
    }
    stage("sizes + candidates");
    if (trace) fprintf(stderr, "[maintenance] %zu partitions, %zu delete candidates\n", sizes.size(), cand.size());
    std::map<int64_t, std::map<int64_t, int64_t>> targets;
    std::vector<int32_t> flat;  // per-target counter of the candidate at hand, all zero between candidates
    {
        const int64_t chunk_rows = (int64_t)1 << 18;
        const int dev = 0;  // (pm.ctx() is the shared context of device 0; a group hands a list out on its lead, device 0 too)
        const auto fopt = torch::TensorOptions().dtype(torch::kFloat32).device(torch::kCUDA, dev);
        const auto iopt = torch::TensorOptions().dtype(torch::kInt64).device(torch::kCUDA, dev);
        size_t c0 = 0;
        while (c0 < cand.size()) {
            size_t c1 = c0;
            int64_t rows = 0;
            while (c1 < cand.size() && (rows == 0 || rows + sizes[cand[c1]] <= chunk_rows)) rows += sizes[cand[c1++]];
            if (rows > 0) {
                Tensor x = torch::empty({rows, (int64_t)pm.d()}, fopt), near = torch::empty({rows, 2}, iopt);
                int64_t at = 0;
                for (size_t c = c0; c < c1; c++) {
                    const int64_t n = sizes[cand[c]];
                    if (n > 0) qk_check(pm.lists().get_list(cand[c], x.data_ptr<float>() + at * pm.d(), nullptr, QK_MEM_DEVICE));
                    at += n;
                }
                qk_check(qk_coarse(pm.ctx(), pm.parent_->store(), x.data_ptr<float>(), rows, 2, pm.metric_, near.data_ptr<int64_t>(), nullptr,
                                   QK_MEM_DEVICE));
                qk_check(qk_ctx_synchronize(pm.ctx()));
                Tensor nh = near.cpu();
                const int64_t *np_ = nh.data_ptr<int64_t>();
                at = 0;
                std::vector<std::pair<int64_t, int64_t>> tab;
                // (counted in a flat array indexed by the target's partition number -- a candidate's rows name ~100 distinct targets, and a
                //  std::map increment, or a search of a small table, per row was 120 ms of a 50M index's call)
                for (size_t c = c0; c < c1; c++) {
                    const int64_t n = sizes[cand[c]], own = cand[c];
                    tab.clear();
                    for (int64_t i = at * 2; i < (at + n) * 2; i++) {
                        const int64_t t = np_[i];
                        if (t < 0 || t == own) continue;
                        if ((size_t)t >= flat.size()) flat.resize((size_t)t + 1024, 0);
                        if (flat[(size_t)t]++ == 0) tab.emplace_back(t, 0);
                    }
                    for (auto &e : tab) {
                        e.second = flat[(size_t)e.first];
                        flat[(size_t)e.first] = 0;
                    }
                    auto &counts = targets[own];
                    for (const auto &e : tab) counts[e.first] += e.second;
                    at += n;
                }
            }
            c0 = c1;
        }
    }
    stage("reassign targets");
    for (const auto &kv : sizes) {
        const int64_t pid = kv.first;
        const int size = (int)kv.second;
        const float hr = hit_rate_of(pid);
        const double dd = ce.compute_delete_delta(size, hr, total_partitions, scan_fraction, avg_size);
        if (dd < -p.delete_threshold_ns) {
            if (p.enable_delete_rejection && size > p.min_partition_size && kept_early.count(pid)) {
                // (kept by the rejection rule without the search: the same split test as any kept partition under the extension)
                if (p.split_after_delete_rejection && ce.compute_split_delta(size, hr, total_partitions) < -p.split_threshold_ns) to_split.push_back(pid);
            } else if (p.enable_delete_rejection && size > p.min_partition_size) {
                const auto &counts = targets[pid];
                std::vector<int64_t> rc, rs;
                std::vector<float> rh;
                for (const auto &c : counts) {
                    rc.push_back(c.second);
                    rs.push_back(sizes.count(c.first) ? sizes[c.first] : 0);
                    rh.push_back(hit_rate_of(c.first));
                }
                if (ce.compute_delete_delta_w_reassign(size, hr, total_partitions, rc, rs, rh) < -p.delete_threshold_ns)
                    to_delete.push_back(pid);
                else if (p.split_after_delete_rejection && ce.compute_split_delta(size, hr, total_partitions) < -p.split_threshold_ns)
                    to_split.push_back(pid);  // (extension, common.h: kept by the rejection -> split test)
            } else {
                to_delete.push_back(pid);
            }
        } else if (size > p.min_partition_size) {
            if (ce.compute_split_delta(size, hr, total_partitions) < -p.split_threshold_ns) to_split.push_back(pid);
        }
    }
    if ((int)to_delete.size() >= total_partitions && !to_delete.empty()) {
        // (safety, not in the reference: a model that wants every partition gone would leave the vectors nowhere to go -- the
        //  largest partition survives)
        auto keep = std::max_element(to_delete.begin(), to_delete.end(), [&](int64_t a, int64_t b) { return sizes[a] < sizes[b]; });
        to_delete.erase(keep);
    }
    stage("walk");
    auto t0 = clk::now();
    if (!to_delete.empty()) {
        Tensor td = torch::tensor(to_delete, torch::kInt64);
        if (!pm.delete_partitions_in_place(td)) pm.delete_partitions(td, true);  // (rows stay on the device: partition_manager.h)
    }
    info->delete_time_us = us_since(t0);
    t0 = clk::now();
    Tensor new_pids;
    if (!to_split.empty()) {
        Tensor sp = torch::tensor(to_split, torch::kInt64);
        new_pids = pm.split_partitions_in_place(sp);  // (rows stay on the device: partition_manager.h)
        if (!new_pids.defined()) {
            auto split = pm.split_partitions(sp);
            pm.delete_partitions(sp, false);
            pm.add_partitions(split);
            new_pids = split->partition_ids;
        }
    }
    info->split_time_us = us_since(t0);
    if (new_pids.defined() && new_pids.numel() > 0) {
        t0 = clk::now();
        local_refinement(new_pids);
        info->split_refine_time_us = us_since(t0);
    }
    info->n_splits = (int64_t)to_split.size();
    info->n_deletes = (int64_t)to_delete.size();
    info->total_time_us = us_since(t_total);
    tr.set_total_vectors((int)std::max<int64_t>(pm.ntotal(), 1));
    return info;
}

void MaintenancePolicy::local_refinement(const Tensor &partition_ids) {  // :187-202
    auto &pm = *partition_manager_;
    if (params_->refinement_radius == 0 || !pm.parent_) return;
    Tensor cent = pm.parent_->get(partition_ids);
    const int64_t n = cent.size(0);
    const int r = (int)std::min<int64_t>(params_->refinement_radius, pm.parent_->ntotal());
    Tensor near = torch::empty({n, r}, torch::kInt64);
    qk_check(qk_coarse(pm.ctx(), pm.parent_->store(), cent.data_ptr<float>(), n, r, pm.metric_, near.data_ptr<int64_t>(), nullptr, QK_MEM_HOST));
    Tensor flat = std::get<0>(torch::_unique(near.reshape({-1})));
    flat = flat.masked_select(flat >= 0);
    pm.refine_partitions(flat, params_->refinement_iterations);
}

}  // namespace quake_amd
