// quake_index.cpp -- method bodies of the QuakeIndex facade (see quake_index.h) and the plumbing shared by the mirror's
// translation units.  No arithmetic here.
#include "quake_index.h"

#include <algorithm>
#include <chrono>
#include <filesystem>
#include <fstream>
#include <map>
#include <mutex>
#include <stdexcept>

namespace quake_amd {

namespace {
using clk = std::chrono::high_resolution_clock;
inline int us_since(clk::time_point t0) { return (int)std::chrono::duration_cast<std::chrono::microseconds>(clk::now() - t0).count(); }
}  // namespace

void qk_check(int st) {
    if (st == QK_OK) return;
    if (st == QK_ERR_INVALID) throw std::invalid_argument(qk_last_error());
    throw std::runtime_error(qk_last_error());
}

qk_ctx *qk_device_context(int device) {
    static std::mutex mu;
    static std::map<int, qk_ctx *> ctxs;
    std::lock_guard<std::mutex> lock(mu);
    auto it = ctxs.find(device);
    if (it != ctxs.end()) return it->second;
    qk_ctx *c = nullptr;
    qk_check(qk_ctx_create(device, &c));
    ctxs[device] = c;
    return c;
}

Tensor host_f32(const Tensor &t) { return t.to(torch::kCPU, torch::kFloat32).contiguous(); }
Tensor host_i64(const Tensor &t) { return t.to(torch::kCPU, torch::kInt64).contiguous(); }

int str_to_metric_type(std::string metric) {
    std::transform(metric.begin(), metric.end(), metric.begin(), ::tolower);
    if (metric == "l2") return QK_METRIC_L2;
    if (metric == "ip") return QK_METRIC_IP;
    throw std::invalid_argument("Invalid metric type: " + metric);
}

QuakeIndex::QuakeIndex(int current_level) : current_level_(current_level) {}
QuakeIndex::~QuakeIndex() = default;

void QuakeIndex::require_built(const char *msg) const {
    if (!partition_manager_ || !partition_manager_->has_lists()) throw std::runtime_error(msg);
}

void QuakeIndex::make_coordinator(int num_workers) {
    initialize_maintenance_policy(maintenance_policy_params_ ? maintenance_policy_params_ : std::make_shared<MaintenancePolicyParams>());
    query_coordinator_ = std::make_shared<QueryCoordinator>(parent_, partition_manager_, maintenance_policy_, (MetricType)metric_, num_workers);
}

shared_ptr<BuildTimingInfo> QuakeIndex::build(Tensor x, Tensor ids, shared_ptr<IndexBuildParams> build_params) {  // quake_index.cpp:29-90
    auto t_total = clk::now();
    build_params_ = build_params;
    metric_ = str_to_metric_type(build_params_->metric);
    if (x.dim() != 2) throw std::runtime_error("[QuakeIndex::build] x must be 2-D [num_vectors, dimension]");
    if (x.size(0) != ids.size(0)) throw std::runtime_error("[QuakeIndex::build] x.size(0) != ids.size(0)");
    const int64_t n = x.size(0);
    const int d = (int)x.size(1);
    auto info = std::make_shared<BuildTimingInfo>();
    info->n_vectors = n;
    info->d = d;
    partition_manager_ = std::make_shared<PartitionManager>();
    partition_manager_->metric_ = metric_;
    // num_workers > 0: the partitions go straight to the members of a device group (initialize_workers -> distribute_partitions
    // would move them there anyway, through one device)
    partition_manager_->plan_workers(build_params_->num_workers);
    const int nlist = build_params_->nlist;
    if (nlist > 1) {
        // The build's copy of x (:33) lives on the DEVICE: one transfer of the caller's rows, then k-means, the stable bucketing by
        // assignment (torch::sort + index_select in the reference, clustering.cpp:69-72) and the store's ingest all run there.
        // (Host-side the same steps -- clone, argsort and index_select of 5 GB, a second transfer -- were 10.8 s for 10M x 128.)
        auto t0 = clk::now();
        const auto dev0 = torch::Device(torch::kCUDA, 0);
        Tensor xd = x.to(dev0, torch::kFloat32, /*non_blocking=*/false, /*copy=*/true).contiguous();
        Tensor idd = ids.to(dev0, torch::kInt64).reshape({-1}).contiguous();
        Tensor centroids_d = torch::empty({nlist, d}, torch::TensorOptions().dtype(torch::kFloat32).device(dev0));
        Tensor assign_d = torch::empty({n}, torch::TensorOptions().dtype(torch::kInt64).device(dev0));
        torch::cuda::synchronize();
        // kmeans() (clustering.cpp:13-97); IP: the copy is normalised in place, the normalised rows are what gets stored
        qk_check(qk_kmeans(qk_device_context(0), xd.data_ptr<float>(), n, d, nlist, metric_, build_params_->niter, 1234ULL,
                           centroids_d.data_ptr<float>(), assign_d.data_ptr<int64_t>(), QK_MEM_DEVICE));
        qk_check(qk_ctx_synchronize(qk_device_context(0)));
        info->train_time_us = us_since(t0);
        t0 = clk::now();
        Tensor order = torch::argsort(assign_d, /*stable=*/true);
        Tensor counts = torch::bincount(assign_d, {}, nlist).to(torch::kInt64).cpu();
        Tensor offsets = torch::zeros({nlist + 1}, torch::kInt64);
        offsets.slice(0, 1, nlist + 1).copy_(torch::cumsum(counts, 0));
        parent_ = std::make_shared<QuakeIndex>(current_level_ + 1);
        auto pp = std::make_shared<IndexBuildParams>();
        pp->metric = build_params_->metric;
        pp->num_workers = build_params_->num_workers;
        parent_->build(centroids_d.cpu(), torch::arange(nlist, torch::kInt64), pp);
        Tensor ids_sorted = idd.index_select(0, order);
        Tensor x_sorted = xd.index_select(0, order);
        xd = Tensor();  // (the unsorted copy is not needed next to the sorted one and the arena)
        partition_manager_->init_from_csr(parent_, offsets, ids_sorted, x_sorted);
        info->assign_time_us = us_since(t0);
        info->n_clusters = nlist;
    } else {  // flat index (:68-79): one partition
        Tensor xh = host_f32(x).clone();  // build clones x (:33)
        Tensor idh = host_i64(ids);
        parent_ = nullptr;
        Tensor offsets = torch::tensor({(int64_t)0, n}, torch::kInt64);
        partition_manager_->init_from_csr(nullptr, offsets, idh, xh);
        info->n_clusters = 1;
    }
    make_coordinator(build_params_->num_workers);
    info->total_time_us = us_since(t_total);
    return info;
}

shared_ptr<SearchResult> QuakeIndex::search(Tensor x, shared_ptr<SearchParams> sp) {  // :93-99
    if (!query_coordinator_) throw std::runtime_error("[QuakeIndex::search()] No query coordinator. Did you build the index?");
    return query_coordinator_->search(x, sp);
}

Tensor QuakeIndex::get(Tensor ids) {
    require_built("[QuakeIndex::get()] No partition manager. Index not built?");
    return partition_manager_->get(ids);
}

Tensor QuakeIndex::get_ids() {
    require_built("[QuakeIndex::get_ids()] No partition manager. Index not built?");
    return partition_manager_->get_ids();
}

shared_ptr<ModifyTimingInfo> QuakeIndex::add(Tensor x, Tensor ids) {
    require_built("[QuakeIndex::add()] No partition manager. Build the index first.");
    if (maintenance_policy_) maintenance_policy_->flush_hits();  // (hits are credited with the sizes of the moment they were scanned)
    auto info = partition_manager_->add(x, ids);
    publish();
    return info;
}

// what a modification left for the next search to do (list table upload, the parent's row-major copy) is done now: the queries
// after an add / remove / maintenance do not pay for it (qk_store_publish, include/quake_hip.h)
void QuakeIndex::publish() {
    if (!partition_manager_) return;
    qk_check(partition_manager_->lists().publish());
    if (parent_ && parent_->partition_manager_) qk_check(parent_->partition_manager_->lists().publish());
}

shared_ptr<ModifyTimingInfo> QuakeIndex::remove(Tensor ids) {
    require_built("[QuakeIndex::remove()] No partition manager. Build the index first.");
    if (maintenance_policy_) maintenance_policy_->flush_hits();
    auto info = partition_manager_->remove(ids);
    publish();
    return info;
}

shared_ptr<ModifyTimingInfo> QuakeIndex::modify(Tensor ids, Tensor x) {  // :147-150
    remove(ids);
    return add(x, ids);
}

void QuakeIndex::initialize_maintenance_policy(shared_ptr<MaintenancePolicyParams> p) {  // :152-155
    maintenance_policy_params_ = p;
    if (partition_manager_) {
        if (maintenance_policy_) maintenance_policy_->flush_hits();  // (hits recorded under the old policy reach it before it goes)
        const bool track = maintenance_policy_ && maintenance_policy_->track_hits_;
        maintenance_policy_ = std::make_shared<MaintenancePolicy>(partition_manager_, p);
        maintenance_policy_->track_hits_ = track;
        if (query_coordinator_) query_coordinator_->maintenance_policy_ = maintenance_policy_;
    }
}

void QuakeIndex::set_latency_profile(const std::string &path) {
    if (!maintenance_policy_ || !partition_manager_) throw std::runtime_error("[QuakeIndex::set_latency_profile()] No maintenance policy set.");
    auto lat = default_latency_estimator(partition_manager_->d(), path);
    maintenance_policy_->cost_estimator_ =
        std::make_shared<MaintenanceCostEstimator>(partition_manager_->d(), maintenance_policy_->params_->alpha, 10, lat);
}

void QuakeIndex::set_track_hits(bool on) {
    if (!maintenance_policy_) throw std::runtime_error("[QuakeIndex::set_track_hits()] No maintenance policy set.");
    maintenance_policy_->track_hits_ = on;
}

shared_ptr<MaintenanceTimingInfo> QuakeIndex::maintenance() {  // :157-163
    if (!maintenance_policy_) throw std::runtime_error("[QuakeIndex::maintenance()] No maintenance policy set.");
    auto info = maintenance_policy_->perform_maintenance();
    publish();
    return info;
}

void QuakeIndex::refine_partitions(Tensor partition_ids, int iterations) {
    require_built("[PartitionManager] refine_partitions: index not built");
    if (maintenance_policy_) maintenance_policy_->flush_hits();
    partition_manager_->refine_partitions(partition_ids, iterations);
}

bool QuakeIndex::validate() { return partition_manager_ && partition_manager_->validate(); }

int64_t QuakeIndex::ntotal() { return partition_manager_ ? partition_manager_->ntotal() : 0; }
int64_t QuakeIndex::nlist() { return partition_manager_ ? partition_manager_->nlist() : 0; }
int QuakeIndex::d() { return partition_manager_ ? partition_manager_->d() : 0; }

// on-disk layout of the reference (quake_index.cpp:170-267): <dir>/metadata.txt, <dir>/partitions, <dir>/parent/...
void QuakeIndex::save(const std::string &dir_path) {
    namespace fs = std::filesystem;
    require_built("Cannot save an index that was not built");
    if (fs::exists(dir_path) && !fs::is_directory(dir_path)) throw std::runtime_error("save path exists but is not a directory: " + dir_path);
    fs::create_directories(dir_path);
    {
        std::ofstream ofs((fs::path(dir_path) / "metadata.txt").string());
        if (!ofs.is_open()) throw std::runtime_error("Cannot open metadata file for writing");
        ofs << "metric=" << metric_ << "\n" << "level=" << current_level_ << "\n" << "ntotal=" << ntotal() << "\n" << "nlist=" << nlist() << "\n";
    }
    partition_manager_->save((fs::path(dir_path) / "partitions").string());
    if (parent_) parent_->save((fs::path(dir_path) / "parent").string());
}

void QuakeIndex::load(const std::string &dir_path, int n_workers) {
    namespace fs = std::filesystem;
    if (!fs::exists(dir_path) || !fs::is_directory(dir_path)) throw std::runtime_error("Cannot load QuakeIndex, directory does not exist: " + dir_path);
    {
        std::ifstream ifs((fs::path(dir_path) / "metadata.txt").string());
        if (!ifs.is_open()) throw std::runtime_error("Cannot open metadata file for reading");
        std::string line;
        while (std::getline(ifs, line)) {
            auto pos = line.find('=');
            if (pos == std::string::npos) continue;
            std::string key = line.substr(0, pos), val = line.substr(pos + 1);
            if (key == "metric") metric_ = std::stoi(val);
            else if (key == "level") current_level_ = std::stoi(val);
        }
    }
    const std::string pdir = (fs::path(dir_path) / "parent").string();
    if (fs::exists(pdir) && fs::is_directory(pdir)) {
        parent_ = std::make_shared<QuakeIndex>(current_level_ + 1);
        parent_->load(pdir, n_workers);
    } else {
        parent_ = nullptr;
    }
    partition_manager_ = std::make_shared<PartitionManager>();
    partition_manager_->metric_ = metric_;
    partition_manager_->parent_ = parent_;
    partition_manager_->plan_workers(n_workers);
    partition_manager_->load((fs::path(dir_path) / "partitions").string());
    maintenance_policy_params_ = nullptr;  // load resets the policy to its defaults (:262-264)
    maintenance_policy_ = nullptr;
    make_coordinator(n_workers);
}

}  // namespace quake_amd
