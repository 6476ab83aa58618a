// quake_index.cpp -- method bodies of the C++ host mirror (see quake_index.h).  No arithmetic here.
#include "quake_index.h"

#include <algorithm>
#include <chrono>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <map>
#include <mutex>
#include <stdexcept>

namespace quake_amd {

namespace {
using clk = std::chrono::high_resolution_clock;
inline int us_since(clk::time_point t0) { return (int)std::chrono::duration_cast<std::chrono::microseconds>(clk::now() - t0).count(); }

void check(int st) {  // qk_status -> the exception type the reference throws
    if (st == QK_OK) return;
    if (st == QK_ERR_INVALID) throw std::invalid_argument(qk_last_error());
    throw std::runtime_error(qk_last_error());
}

qk_ctx *device_context(int device) {
    static std::mutex mu;
    static std::map<int, qk_ctx *> ctxs;
    std::lock_guard<std::mutex> lock(mu);
    auto it = ctxs.find(device);
    if (it != ctxs.end()) return it->second;
    qk_ctx *c = nullptr;
    check(qk_ctx_create(device, &c));
    ctxs[device] = c;
    return c;
}

Tensor host_f32(const Tensor &t) { return t.to(torch::kCPU, torch::kFloat32).contiguous(); }
Tensor host_i64(const Tensor &t) { return t.to(torch::kCPU, torch::kInt64).contiguous(); }
}  // namespace

int str_to_metric_type(std::string metric) {
    std::transform(metric.begin(), metric.end(), metric.begin(), ::tolower);
    if (metric == "l2") return QK_METRIC_L2;
    if (metric == "ip") return QK_METRIC_IP;
    throw std::invalid_argument("Invalid metric type: " + metric);
}

QuakeIndex::QuakeIndex(int current_level) : current_level_(current_level) {}

QuakeIndex::~QuakeIndex() {
    if (store_) qk_store_destroy(store_);
    store_ = nullptr;
}

void QuakeIndex::require_built(const char *msg) const {
    if (!store_) throw std::runtime_error(msg);
}

void QuakeIndex::reset_store(int d) {
    ctx_ = device_context(0);
    if (store_) qk_store_destroy(store_);
    store_ = nullptr;
    check(qk_store_create(ctx_, d, &store_));
    d_ = d;
    resident_.clear();
}

shared_ptr<BuildTimingInfo> QuakeIndex::build(Tensor x, Tensor ids, shared_ptr<IndexBuildParams> build_params) {
    auto t_total = clk::now();
    build_params_ = build_params;
    metric_ = str_to_metric_type(build_params_->metric);
    if (x.dim() != 2) throw std::runtime_error("[QuakeIndex::build] x must be 2-D [num_vectors, dimension]");
    if (x.size(0) != ids.size(0)) throw std::runtime_error("[QuakeIndex::build] x.size(0) != ids.size(0)");
    Tensor xh = host_f32(x).clone();  // build clones x (quake_index.cpp:33)
    Tensor idh = host_i64(ids);
    const int64_t n = xh.size(0);
    const int d = (int)xh.size(1);
    auto info = std::make_shared<BuildTimingInfo>();
    info->n_vectors = n;
    info->d = d;
    reset_store(d);
    const int nlist = build_params_->nlist;
    if (nlist > 1) {
        auto t0 = clk::now();
        Tensor centroids = torch::empty({nlist, d}, torch::kFloat32);
        Tensor assign = torch::empty({n}, torch::kInt64);
        // kmeans() (clustering.cpp:13-97); IP: xh is normalised in place, the normalised copy is what gets stored
        check(qk_kmeans(ctx_, xh.data_ptr<float>(), n, d, nlist, metric_, build_params_->niter, 1234ULL,
                        centroids.data_ptr<float>(), assign.data_ptr<int64_t>(), QK_MEM_HOST));
        info->train_time_us = us_since(t0);
        t0 = clk::now();
        Tensor order = torch::argsort(assign, /*stable=*/true);  // torch::sort + index_select (clustering.cpp:69-72)
        Tensor counts = torch::bincount(assign, {}, nlist).to(torch::kInt64);
        Tensor offsets = torch::zeros({nlist + 1}, torch::kInt64);
        offsets.slice(0, 1, nlist + 1).copy_(torch::cumsum(counts, 0));
        Tensor xs = xh.index_select(0, order).contiguous();
        Tensor is = idh.index_select(0, order).contiguous();
        check(qk_store_build_csr(store_, nlist, offsets.data_ptr<int64_t>(), is.data_ptr<int64_t>(), xs.data_ptr<float>(), QK_MEM_HOST));
        parent_ = std::make_shared<QuakeIndex>(current_level_ + 1);
        auto pp = std::make_shared<IndexBuildParams>();
        pp->metric = build_params_->metric;
        pp->num_workers = build_params_->num_workers;
        parent_->build(centroids, torch::arange(nlist, torch::kInt64), pp);
        info->assign_time_us = us_since(t0);
        info->n_clusters = nlist;
        next_pid_ = nlist;
    } else {  // flat index (quake_index.cpp:68-79)
        int64_t offs[2] = {0, n};
        check(qk_store_build_csr(store_, 1, offs, idh.data_ptr<int64_t>(), xh.data_ptr<float>(), QK_MEM_HOST));
        parent_ = nullptr;
        info->n_clusters = 1;
        next_pid_ = 1;
    }
    const int64_t *ip = idh.data_ptr<int64_t>();
    resident_.insert(ip, ip + n);
    initialize_maintenance_policy(std::make_shared<MaintenancePolicyParams>());
    info->total_time_us = us_since(t_total);
    return info;
}

shared_ptr<SearchResult> QuakeIndex::search(Tensor x, shared_ptr<SearchParams> sp) {
    require_built("[QuakeIndex::search()] No query coordinator. Did you build the index?");
    auto res = std::make_shared<SearchResult>();
    res->timing_info = std::make_shared<SearchTimingInfo>();
    res->timing_info->search_params = sp;
    res->timing_info->n_clusters = nlist();
    if (!x.defined() || x.size(0) == 0) {  // query_coordinator.cpp:476-482
        res->ids = torch::empty({0}, torch::kInt64);
        res->distances = torch::empty({0}, torch::kFloat32);
        return res;
    }
    auto t0 = clk::now();
    const bool on_dev = x.is_cuda();
    Tensor xq = on_dev ? x.to(torch::kFloat32).contiguous() : host_f32(x);
    // the library runs on its own stream: device inputs must be complete before it starts (the calls below hand back
    // a drained stream, so the outputs are ready for torch's stream)
    if (on_dev) torch::cuda::synchronize(xq.device().index());
    const int64_t Q = xq.size(0);
    const int k = sp->k > 0 ? sp->k : 1;  // query_coordinator.cpp:490
    const int nprobe = std::max(sp->nprobe, 1);
    auto opts_i = torch::TensorOptions().dtype(torch::kInt64).device(xq.device());
    auto opts_f = torch::TensorOptions().dtype(torch::kFloat32).device(xq.device());
    res->ids = torch::empty({Q, k}, opts_i);
    res->distances = torch::empty({Q, k}, opts_f);
    qk_timing tm;
    std::memset(&tm, 0, sizeof(tm));
    if (sp->recall_target > 0.0f && parent_ && !sp->batched_scan) {
        // adaptive partition scanning (query_coordinator.cpp:502,637-641): candidates = nlist * initial_search_fraction
        Tensor nscan = torch::empty({Q}, torch::TensorOptions().dtype(torch::kInt32).device(xq.device()));
        check(qk_search_aps(ctx_, parent_->store_, store_, xq.data_ptr<float>(), Q, k, metric_, sp->recall_target,
                            sp->recompute_threshold, sp->use_precomputed ? 1 : 0, sp->initial_search_fraction,
                            res->ids.data_ptr<int64_t>(), res->distances.data_ptr<float>(), nscan.data_ptr<int32_t>(),
                            on_dev ? QK_MEM_DEVICE : QK_MEM_HOST, &tm));
        auto ti = res->timing_info;
        ti->n_queries = Q;
        ti->partitions_scanned = (int)nscan.sum().item<int64_t>();
        ti->job_wait_time_ns = (int64_t)(tm.total_ms * 1e6);
        ti->parent_info = std::make_shared<SearchTimingInfo>();
        ti->parent_info->n_queries = Q;
        ti->parent_info->n_clusters = 1;
        ti->total_time_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(clk::now() - t0).count();
        return res;
    }
    check(qk_ctx_set_timing(ctx_, 1));
    int st = qk_search(ctx_, parent_ ? parent_->store_ : nullptr, store_, xq.data_ptr<float>(), Q, nprobe, k, metric_,
                       res->ids.data_ptr<int64_t>(), res->distances.data_ptr<float>(), on_dev ? QK_MEM_DEVICE : QK_MEM_HOST, &tm);
    qk_ctx_set_timing(ctx_, 0);
    check(st);
    auto ti = res->timing_info;
    ti->n_queries = Q;
    ti->partitions_scanned = (int)tm.n_items;
    ti->job_enqueue_time_ns = (int64_t)(tm.group_ms * 1e6);
    ti->job_wait_time_ns = (int64_t)(tm.scan_ms * 1e6);
    ti->result_aggregate_time_ns = (int64_t)(tm.merge_ms * 1e6);
    if (parent_) {
        ti->parent_info = std::make_shared<SearchTimingInfo>();
        ti->parent_info->n_queries = Q;
        ti->parent_info->n_clusters = 1;
        ti->parent_info->total_time_ns = (int64_t)(tm.coarse_ms * 1e6);
    }
    ti->total_time_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(clk::now() - t0).count();
    return res;
}

Tensor QuakeIndex::get(Tensor ids) {
    require_built("[QuakeIndex::get()] No partition manager. Index not built?");
    Tensor idh = host_i64(ids).reshape({-1});
    Tensor out = torch::empty({idh.size(0), d_}, torch::kFloat32);
    for (int64_t i = 0; i < idh.size(0); i++) {
        int found = 0;
        check(qk_store_get_vector(store_, idh.data_ptr<int64_t>()[i], out.data_ptr<float>() + i * d_, &found));
        if (!found) throw std::runtime_error("ID not found in any partition");
    }
    return out;
}

Tensor QuakeIndex::get_ids() {
    require_built("[QuakeIndex::get_ids()] No partition manager. Index not built?");
    int64_t nl = 0;
    check(qk_store_list_ids(store_, nullptr, &nl));
    std::vector<int64_t> lists((size_t)nl);
    if (nl) check(qk_store_list_ids(store_, lists.data(), &nl));
    Tensor out = torch::empty({qk_store_ntotal(store_)}, torch::kInt64);
    int64_t pos = 0;
    for (int64_t p : lists) {
        int64_t sz = 0;
        check(qk_store_list_size(store_, p, &sz));
        if (sz) check(qk_store_get_list(store_, p, nullptr, out.data_ptr<int64_t>() + pos, QK_MEM_HOST));
        pos += sz;
    }
    return out;
}

shared_ptr<ModifyTimingInfo> QuakeIndex::add(Tensor x, Tensor ids) {  // partition_manager.cpp:123-262
    require_built("[QuakeIndex::add()] No partition manager. Build the index first.");
    auto info = std::make_shared<ModifyTimingInfo>();
    auto t0 = clk::now();
    if (x.size(0) != ids.size(0)) throw std::runtime_error("[PartitionManager] add: mismatch in vectors.size(0) and vector_ids.size(0).");
    const int64_t n = x.size(0);
    info->n_vectors = n;
    if (n == 0) return info;
    if (x.dim() != 2) throw std::runtime_error("[PartitionManager] add: 'vectors' must be 2D [N, dim].");
    Tensor xh = host_f32(x);
    Tensor idh = host_i64(ids).reshape({-1});
    const int64_t *ip = idh.data_ptr<int64_t>();
    std::unordered_set<int64_t> uniq(ip, ip + n);
    for (int64_t i = 0; i < n; i++)
        if (ip[i] > (int64_t)INT32_MAX) throw std::runtime_error("[PartitionManager] add: vector_ids must be less than INT_MAX.");
    if ((int64_t)uniq.size() != n) throw std::runtime_error("[PartitionManager] add: vector_ids must be unique.");
    for (int64_t i = 0; i < n; i++)
        if (resident_.count(ip[i])) throw std::runtime_error("[PartitionManager] init_partitions: vector ID already exists in the index.");
    resident_.insert(ip, ip + n);
    info->input_validation_time_us = us_since(t0);
    t0 = clk::now();
    Tensor assign = torch::zeros({n}, torch::kInt64);
    if (parent_) {  // parent_->search(x, {k = 1, ...}) (:219-230) == coarse step with nprobe 1
        check(qk_coarse(ctx_, parent_->store_, xh.data_ptr<float>(), n, 1, metric_, assign.data_ptr<int64_t>(), nullptr, QK_MEM_HOST));
    }
    info->find_partition_time_us = us_since(t0);
    t0 = clk::now();
    // per-list append order = input order (:245-258)
    check(qk_store_add_batch(store_, n, idh.data_ptr<int64_t>(), xh.data_ptr<float>(), assign.data_ptr<int64_t>(), QK_MEM_HOST));
    info->modify_time_us = us_since(t0);
    return info;
}

shared_ptr<ModifyTimingInfo> QuakeIndex::remove(Tensor ids) {  // partition_manager.cpp:264-320
    require_built("[QuakeIndex::remove()] No partition manager. Build the index first.");
    auto info = std::make_shared<ModifyTimingInfo>();
    info->n_vectors = ids.size(0);
    if (ids.size(0) == 0) return info;
    auto t0 = clk::now();
    Tensor idh = host_i64(ids).reshape({-1});
    const int64_t *ip = idh.data_ptr<int64_t>();
    for (int64_t i = 0; i < idh.size(0); i++) {
        if (!resident_.count(ip[i])) throw std::runtime_error("[PartitionManager] remove: vector ID does not exist in the index.");
        resident_.erase(ip[i]);
    }
    info->input_validation_time_us = us_since(t0);
    t0 = clk::now();
    check(qk_store_remove_ids(store_, idh.size(0), ip, nullptr));
    info->modify_time_us = us_since(t0);
    return info;
}

shared_ptr<ModifyTimingInfo> QuakeIndex::modify(Tensor ids, Tensor x) {  // quake_index.cpp:147-150
    remove(ids);
    return add(x, ids);
}

void QuakeIndex::initialize_maintenance_policy(shared_ptr<MaintenancePolicyParams> p) { maintenance_policy_params_ = p; }

shared_ptr<MaintenanceTimingInfo> QuakeIndex::maintenance() {
    // MaintenancePolicy::perform_maintenance (maintenance_policies.cpp:33-177): in the reference snapshot search() never
    // calls record_query_hits, so through the public API the policy always returns at the "window not full" guard
    // (:36-41); that observable behaviour is kept.  The policy itself is out of scope.
    if (!maintenance_policy_params_) throw std::runtime_error("[QuakeIndex::maintenance()] No maintenance policy set.");
    return std::make_shared<MaintenanceTimingInfo>();
}

void QuakeIndex::refine_partitions(Tensor partition_ids, int iterations) {  // partition_manager.cpp:446-487
    require_built("[PartitionManager] refine_partitions: index not built");
    if (!parent_) return;
    if (!partition_ids.defined()) partition_ids = parent_->get_ids();
    if (partition_ids.size(0) == 0) return;
    Tensor pids = host_i64(partition_ids).reshape({-1});
    Tensor cent = parent_->get(pids);
    check(qk_store_refine_lists(store_, pids.data_ptr<int64_t>(), pids.size(0), cent.data_ptr<float>(), metric_, iterations, QK_MEM_HOST));
    parent_->modify(pids, cent);  // :478
}

bool QuakeIndex::validate() { return store_ != nullptr; }

int64_t QuakeIndex::ntotal() { return store_ ? qk_store_ntotal(store_) : 0; }
int64_t QuakeIndex::nlist() { return store_ ? qk_store_nlist(store_) : 0; }
int QuakeIndex::d() { return d_; }

// on-disk format of the reference: metadata.txt + "partitions" (32-byte header, offsets, partition ids, [codes|ids] chunks)
// + parent/ (quake_index.cpp:170-267, dynamic_inverted_list.cpp:338-520)
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
    int64_t nl = 0;
    check(qk_store_list_ids(store_, nullptr, &nl));
    std::vector<int64_t> lists((size_t)nl);
    if (nl) check(qk_store_list_ids(store_, lists.data(), &nl));
    std::ofstream ofs((fs::path(dir_path) / "partitions").string(), std::ios::binary);
    if (!ofs.is_open()) throw std::runtime_error("Could not open file for writing: " + dir_path);
    const uint32_t magic = 0x44494E4C, version = 3;
    const uint64_t nlist64 = (uint64_t)nl, code_size = (uint64_t)d_ * 4, nparts = (uint64_t)nl;
    ofs.write((const char *)&magic, 4);
    ofs.write((const char *)&version, 4);
    ofs.write((const char *)&nlist64, 8);
    ofs.write((const char *)&code_size, 8);
    ofs.write((const char *)&nparts, 8);
    std::vector<uint64_t> offsets((size_t)nl + 1, 0);
    std::vector<int64_t> sizes((size_t)nl, 0);
    for (int64_t i = 0; i < nl; i++) {
        check(qk_store_list_size(store_, lists[i], &sizes[i]));
        offsets[i + 1] = offsets[i] + (uint64_t)sizes[i] * (code_size + 8);
    }
    ofs.write((const char *)offsets.data(), (std::streamsize)(offsets.size() * 8));
    for (int64_t i = 0; i < nl; i++) {
        uint64_t pid = (uint64_t)lists[i];
        ofs.write((const char *)&pid, 8);
    }
    for (int64_t i = 0; i < nl; i++) {
        std::vector<float> v((size_t)sizes[i] * d_);
        std::vector<int64_t> id((size_t)sizes[i]);
        if (sizes[i]) check(qk_store_get_list(store_, lists[i], v.data(), id.data(), QK_MEM_HOST));
        ofs.write((const char *)v.data(), (std::streamsize)(v.size() * 4));
        ofs.write((const char *)id.data(), (std::streamsize)(id.size() * 8));
    }
    ofs.close();
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
    std::ifstream ifs((fs::path(dir_path) / "partitions").string(), std::ios::binary);
    if (!ifs.is_open()) throw std::runtime_error("Could not open file for reading: " + dir_path);
    uint32_t magic = 0, version = 0;
    uint64_t nlist64 = 0, code_size = 0, nparts = 0;
    ifs.read((char *)&magic, 4);
    ifs.read((char *)&version, 4);
    if (magic != 0x44494E4C) throw std::runtime_error("Invalid file format (bad magic number).");
    if (version != 3) throw std::runtime_error("Unsupported file version: " + std::to_string(version));
    ifs.read((char *)&nlist64, 8);
    ifs.read((char *)&code_size, 8);
    ifs.read((char *)&nparts, 8);
    std::vector<uint64_t> offsets((size_t)nparts + 1), pids((size_t)nparts);
    ifs.read((char *)offsets.data(), (std::streamsize)(offsets.size() * 8));
    ifs.read((char *)pids.data(), (std::streamsize)(pids.size() * 8));
    const int d = (int)(code_size / 4);
    reset_store(d);
    const uint64_t rec = code_size + 8;
    int64_t max_pid = -1;
    for (uint64_t i = 0; i < nparts; i++) {
        const uint64_t chunk = offsets[i + 1] - offsets[i];
        if (chunk % rec != 0) throw std::runtime_error("Partition chunk size not divisible by (code_size+sizeof(idx_t))");
        const int64_t nv = (int64_t)(chunk / rec);
        std::vector<float> v((size_t)nv * d);
        std::vector<int64_t> id((size_t)nv);
        ifs.read((char *)v.data(), (std::streamsize)(v.size() * 4));
        ifs.read((char *)id.data(), (std::streamsize)(id.size() * 8));
        check(qk_store_add_list(store_, (int64_t)pids[i]));
        if (nv) check(qk_store_add_entries(store_, (int64_t)pids[i], nv, id.data(), v.data(), QK_MEM_HOST));
        resident_.insert(id.begin(), id.end());
        max_pid = std::max<int64_t>(max_pid, (int64_t)pids[i]);
    }
    next_pid_ = max_pid + 1;
    const std::string pdir = (fs::path(dir_path) / "parent").string();
    if (fs::exists(pdir) && fs::is_directory(pdir)) {
        parent_ = std::make_shared<QuakeIndex>(current_level_ + 1);
        parent_->load(pdir, n_workers);
    } else {
        parent_ = nullptr;
    }
    initialize_maintenance_policy(std::make_shared<MaintenancePolicyParams>());
}

}  // namespace quake_amd
