// partition_manager.cpp -- see partition_manager.h.  Every data-parallel step goes through the C ABI: assignment of new
// vectors = the coarse step (qk_coarse), 2-means split = qk_kmeans, refinement = qk_store_refine_lists.
#include "partition_manager.h"

#include <cstdlib>
#include <mutex>
#include <thread>

#include <algorithm>
#include <chrono>
#include <climits>
#include <filesystem>
#include <fstream>
#include <stdexcept>
#include <unordered_set>

#include "quake_index.h"

namespace quake_amd {

namespace {
using clk = std::chrono::high_resolution_clock;
inline int us_since(clk::time_point t0) { return (int)std::chrono::duration_cast<std::chrono::microseconds>(clk::now() - t0).count(); }
}  // namespace

size_t DynamicInvertedLists::list_size(size_t list_no) const {
    int64_t sz = 0;
    qk_check(s_->list_size((int64_t)list_no, &sz));
    return (size_t)sz;
}

PartitionManager::PartitionManager() = default;

PartitionManager::~PartitionManager() {
    partition_store_ = nullptr;
    release_lists();
}

void PartitionManager::release_lists() {
    if (lists_.store) qk_store_destroy(lists_.store);
    if (lists_.group) qk_group_destroy(lists_.group);
    lists_ = DeviceLists();
}

// num_workers members, member j on GPU j % #GPUs (more workers than GPUs: several members per device, like several worker
// threads per core in the reference)
void PartitionManager::make_group(int num_workers, int d) {
    const int ndev = std::max<int>(1, (int)torch::cuda::device_count());
    std::vector<int> devices((size_t)num_workers);
    for (int j = 0; j < num_workers; j++) devices[(size_t)j] = j % ndev;
    qk_check(qk_group_create(devices.data(), num_workers, d, &lists_.group));
    ctx_ = qk_device_context(0);  // parent searches, k-means of a split: the shared context of device 0, as without workers
}

qk_ctx *PartitionManager::search_ctx() const {
    qk_ctx *c = ctx_;
    if (lists_.group) qk_check(qk_group_member(lists_.group, 0, &c, nullptr));
    return c;
}

void PartitionManager::reset_store(int d) {
    partition_store_ = nullptr;
    release_lists();
    // workers only where there are partitions to distribute: a flat index (no parent: ONE partition) stays one store
    if (planned_workers_ > 0 && parent_) {
        make_group(planned_workers_, d);
    } else {
        ctx_ = qk_device_context(0);
        qk_check(qk_store_create(ctx_, d, &lists_.store));
    }
    partition_store_ = std::make_shared<DynamicInvertedLists>(&lists_, (size_t)d * 4);
    d_ = d;
    resident_ids_.clear();
    core_of_.clear();
}

void PartitionManager::require_store(const char *who) const {
    if (!lists_) throw std::runtime_error(std::string("[PartitionManager] ") + who + ": partitions are not initialized.");
}

void PartitionManager::init_partitions(shared_ptr<QuakeIndex> parent, shared_ptr<Clustering> c, bool check_uniques) {
    // partition_manager.cpp:33-121
    if (!c) throw std::runtime_error("[PartitionManager] init_partitions: clustering is null.");
    const int64_t nlist = c->nlist();
    if (nlist <= 0) throw std::runtime_error("[PartitionManager] init_partitions: nlist is zero.");
    if (c->partition_ids.size(0) != nlist || (int64_t)c->vector_ids.size() != nlist)
        throw std::runtime_error("[PartitionManager] init_partitions: partition_ids / vectors / vector_ids disagree.");
    parent_ = parent;
    check_uniques_ = check_uniques;
    reset_store((int)c->dim());
    Tensor pids = host_i64(c->partition_ids);
    int64_t max_pid = -1;
    for (int64_t i = 0; i < nlist; i++) {
        const int64_t pid = pids[i].item<int64_t>();
        qk_check(lists_.add_list(pid));
        Tensor v = host_f32(c->vectors[(size_t)i]), id = host_i64(c->vector_ids[(size_t)i]).reshape({-1});
        if (v.size(0) != id.size(0)) throw std::runtime_error("[PartitionManager] init_partitions: vectors and ids disagree.");
        const int64_t *ip = id.data_ptr<int64_t>();
        if (check_uniques)
            for (int64_t j = 0; j < id.size(0); j++)
                if (!resident_ids_.insert(ip[j]).second)
                    throw std::runtime_error("[PartitionManager] init_partitions: vector ID already exists in the index.");
        if (!check_uniques) resident_ids_.insert(ip, ip + id.size(0));
        if (id.size(0)) qk_check(lists_.add_entries(pid, id.size(0), ip, v.data_ptr<float>(), QK_MEM_HOST));
        max_pid = std::max(max_pid, pid);
    }
    curr_partition_id_ = max_pid + 1;
}

void PartitionManager::init_from_csr(shared_ptr<QuakeIndex> parent, const Tensor &offsets, const Tensor &ids, const Tensor &vectors) {
    parent_ = parent;
    Tensor off = host_i64(offsets);
    const int64_t nlist = off.size(0) - 1;
    if (vectors.is_cuda() && ids.is_cuda()) {
        // rows and ids already on the device (QuakeIndex::build buckets them there): the store ingests them in place
        Tensor v = vectors.to(torch::kFloat32).contiguous(), idd = ids.to(torch::kInt64).contiguous();
        reset_store((int)v.size(1));
        torch::cuda::synchronize();  // (torch's stream made them; the store's launches run on the library's)
        qk_check(lists_.build_csr(nlist, off.data_ptr<int64_t>(), idd.data_ptr<int64_t>(), v.data_ptr<float>(), QK_MEM_DEVICE));
        qk_check(qk_ctx_synchronize(qk_device_context(0)));
        Tensor id = idd.cpu();
        const int64_t *ip = id.data_ptr<int64_t>();
        resident_ids_.insert(ip, ip + id.size(0));
        curr_partition_id_ = nlist;
        return;
    }
    Tensor id = host_i64(ids), v = host_f32(vectors);
    reset_store((int)v.size(1));
    qk_check(lists_.build_csr(nlist, off.data_ptr<int64_t>(), id.data_ptr<int64_t>(), v.data_ptr<float>(), QK_MEM_HOST));
    const int64_t *ip = id.data_ptr<int64_t>();
    resident_ids_.insert(ip, ip + id.size(0));
    curr_partition_id_ = nlist;
}

shared_ptr<ModifyTimingInfo> PartitionManager::add(const Tensor &vectors, const Tensor &vector_ids, const Tensor &assignments,
                                                   bool check_uniques) {  // partition_manager.cpp:123-262
    require_store("add");
    auto info = std::make_shared<ModifyTimingInfo>();
    auto t0 = clk::now();
    if (!vectors.defined() || !vector_ids.defined()) throw std::runtime_error("[PartitionManager] add: vectors or vector_ids is undefined.");
    if (vectors.size(0) != vector_ids.size(0)) throw std::runtime_error("[PartitionManager] add: mismatch in vectors.size(0) and vector_ids.size(0).");
    const int64_t n = vectors.size(0);
    info->n_vectors = n;
    if (n == 0) return info;
    if (vectors.dim() != 2) throw std::runtime_error("[PartitionManager] add: 'vectors' must be 2D [N, dim].");
    if (vectors.size(1) != d_) throw std::runtime_error("[PartitionManager] add: dimension mismatch.");
    Tensor xh = host_f32(vectors), idh = host_i64(vector_ids).reshape({-1});
    const int64_t *ip = idh.data_ptr<int64_t>();
    for (int64_t i = 0; i < n; i++)
        if (ip[i] > (int64_t)INT32_MAX) throw std::runtime_error("[PartitionManager] add: vector_ids must be less than INT_MAX.");
    if (check_uniques) {
        // uniqueness inside the batch: a bitmap over the batch's OWN id range [min, max] when that range is dense enough (1M ids: 42 -> ~5
        // ms against std::unordered_set), sort + adjacent compare otherwise -- one vector with an id near 2^31 must not allocate and zero
        // a 256 MB bitmap indexed from 0
        int64_t lo = ip[0], hi = ip[0];
        for (int64_t i = 1; i < n; i++) {
            lo = std::min(lo, ip[i]);
            hi = std::max(hi, ip[i]);
        }
        const uint64_t span = (uint64_t)(hi - lo) + 1;
        bool dup = false;
        if (n >= 64 && span <= (uint64_t)n * 64) {
            std::vector<uint64_t> bits((size_t)((span + 63) >> 6), 0);
            for (int64_t i = 0; i < n && !dup; i++) {
                const uint64_t off = (uint64_t)(ip[i] - lo), m = 1ull << (off & 63);
                dup = (bits[(size_t)(off >> 6)] & m) != 0;
                bits[(size_t)(off >> 6)] |= m;
            }
        } else {
            std::vector<int64_t> sorted(ip, ip + n);
            std::sort(sorted.begin(), sorted.end());
            dup = std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end();
        }
        if (dup) throw std::runtime_error("[PartitionManager] add: vector_ids must be unique.");
        for (int64_t i = 0; i < n; i++)
            if (resident_ids_.count(ip[i])) throw std::runtime_error("[PartitionManager] add: vector ID already exists in the index.");
    }
    info->input_validation_time_us = us_since(t0);
    t0 = clk::now();
    Tensor assign;
    if (assignments.defined() && assignments.numel() > 0) {
        if (assignments.size(0) != n) throw std::runtime_error("[PartitionManager] add: assignments.size(0) != vectors.size(0).");
        assign = host_i64(assignments).reshape({-1});
    } else if (parent_) {  // parent_->search(x, {k = 1}) (:219-230) == the coarse step with nprobe 1
        assign = torch::empty({n}, torch::kInt64);
        qk_check(qk_coarse(ctx_, parent_->store(), xh.data_ptr<float>(), n, 1, metric_, assign.data_ptr<int64_t>(), nullptr, QK_MEM_HOST));
    } else {  // flat index: its single partition
        Tensor only = get_partition_ids();
        assign = torch::full({n}, only.numel() ? only[0].item<int64_t>() : 0, torch::kInt64);
    }
    info->find_partition_time_us = us_since(t0);
    t0 = clk::now();
    // per-list append order = input order (:245-258); the ids become resident only once the device step succeeded
    qk_check(lists_.add_batch(n, ip, xh.data_ptr<float>(), assign.data_ptr<int64_t>(), QK_MEM_HOST));
    resident_ids_.insert(ip, ip + n);
    info->modify_time_us = us_since(t0);
    return info;
}

shared_ptr<ModifyTimingInfo> PartitionManager::remove(const Tensor &ids) {  // partition_manager.cpp:264-320
    require_store("remove");
    auto info = std::make_shared<ModifyTimingInfo>();
    info->n_vectors = ids.size(0);
    if (ids.size(0) == 0) return info;
    auto t0 = clk::now();
    Tensor idh = host_i64(ids).reshape({-1});
    const int64_t *ip = idh.data_ptr<int64_t>();
    for (int64_t i = 0; i < idh.size(0); i++)
        if (!resident_ids_.count(ip[i])) throw std::runtime_error("[PartitionManager] remove: vector ID does not exist in the index.");
    info->input_validation_time_us = us_since(t0);
    t0 = clk::now();
    qk_check(lists_.remove_ids(idh.size(0), ip, nullptr));
    for (int64_t i = 0; i < idh.size(0); i++) resident_ids_.erase(ip[i]);
    info->modify_time_us = us_since(t0);
    return info;
}

Tensor PartitionManager::get(const Tensor &ids) {
    require_store("get");
    Tensor idh = host_i64(ids).reshape({-1});
    Tensor out = torch::empty({idh.size(0), d_}, torch::kFloat32);
    for (int64_t i = 0; i < idh.size(0); i++) {
        int found = 0;
        qk_check(lists_.get_vector(idh.data_ptr<int64_t>()[i], out.data_ptr<float>() + i * d_, &found));
        if (!found) throw std::runtime_error("ID not found in any partition");
    }
    return out;
}

Tensor PartitionManager::get_partition_ids() {
    require_store("get_partition_ids");
    int64_t nl = 0;
    qk_check(lists_.list_ids(nullptr, &nl));
    Tensor out = torch::empty({nl}, torch::kInt64);
    if (nl) qk_check(lists_.list_ids(out.data_ptr<int64_t>(), &nl));
    return out;
}

int64_t PartitionManager::get_partition_size(int64_t partition_id) {
    require_store("get_partition_size");
    int64_t sz = 0;
    qk_check(lists_.list_size(partition_id, &sz));
    return sz;
}

std::vector<int64_t> PartitionManager::get_partition_sizes(std::vector<int64_t> partition_ids) {
    std::vector<int64_t> out(partition_ids.size());
    for (size_t i = 0; i < partition_ids.size(); i++) out[i] = get_partition_size(partition_ids[i]);
    return out;
}

Tensor PartitionManager::get_partition_sizes(Tensor partition_ids) {
    if (!partition_ids.defined() || partition_ids.numel() == 0) partition_ids = get_partition_ids();
    Tensor p = host_i64(partition_ids).reshape({-1}).contiguous();
    Tensor out = torch::empty({p.size(0)}, torch::kInt64);
    const int64_t *ip = p.data_ptr<int64_t>();
    int64_t *op = out.data_ptr<int64_t>();
    for (int64_t i = 0; i < p.size(0); i++) op[i] = get_partition_size(ip[i]);
    return out;
}

Tensor PartitionManager::get_ids() {
    require_store("get_ids");
    Tensor lists = get_partition_ids();
    Tensor out = torch::empty({lists_.ntotal()}, torch::kInt64);
    int64_t pos = 0;
    for (int64_t i = 0; i < lists.size(0); i++) {
        const int64_t p = lists[i].item<int64_t>(), sz = get_partition_size(p);
        if (sz) qk_check(lists_.get_list(p, nullptr, out.data_ptr<int64_t>() + pos, QK_MEM_HOST));
        pos += sz;
    }
    return out;
}

shared_ptr<Clustering> PartitionManager::select_partitions(const Tensor &partition_ids, bool /*copy*/) {  // :344-390
    require_store("select_partitions");
    Tensor p = host_i64(partition_ids).reshape({-1});
    auto c = std::make_shared<Clustering>();
    c->partition_ids = p.clone();
    c->centroids = parent_ ? parent_->get(p) : torch::empty({p.size(0), d_}, torch::kFloat32);
    for (int64_t i = 0; i < p.size(0); i++) {
        const int64_t pid = p[i].item<int64_t>(), sz = get_partition_size(pid);
        Tensor v = torch::empty({sz, d_}, torch::kFloat32), id = torch::empty({sz}, torch::kInt64);
        if (sz) qk_check(lists_.get_list(pid, v.data_ptr<float>(), id.data_ptr<int64_t>(), QK_MEM_HOST));
        c->vectors.push_back(v);
        c->vector_ids.push_back(id);
    }
    return c;
}

shared_ptr<Clustering> PartitionManager::split_partitions(const Tensor &partition_ids) {  // :392-444: 2-means per partition
    require_store("split_partitions");
    auto sel = select_partitions(partition_ids);
    auto out = std::make_shared<Clustering>();
    const int64_t np = sel->nlist();
    std::vector<Tensor> cents;
    for (int64_t i = 0; i < np; i++) {
        Tensor v = sel->vectors[(size_t)i].clone(), id = sel->vector_ids[(size_t)i];
        const int64_t n = v.size(0);
        if (n < 8) {  // too small to split: one half keeps everything (:409-413 asserts instead)
            cents.push_back(sel->centroids[i].unsqueeze(0).repeat({2, 1}));
            out->vectors.push_back(v);
            out->vector_ids.push_back(id);
            out->vectors.push_back(torch::empty({0, d_}, torch::kFloat32));
            out->vector_ids.push_back(torch::empty({0}, torch::kInt64));
            continue;
        }
        Tensor c2 = torch::empty({2, d_}, torch::kFloat32), a = torch::empty({n}, torch::kInt64);
        qk_check(qk_kmeans(ctx_, v.data_ptr<float>(), n, d_, 2, metric_, DEFAULT_NITER, 1234ULL, c2.data_ptr<float>(), a.data_ptr<int64_t>(),
                           QK_MEM_HOST));
        cents.push_back(c2);
        for (int64_t h = 0; h < 2; h++) {
            Tensor rows = torch::nonzero(a == h).reshape({-1});
            out->vectors.push_back(v.index_select(0, rows).contiguous());
            out->vector_ids.push_back(id.index_select(0, rows).contiguous());
        }
    }
    out->centroids = np ? torch::cat(cents, 0) : torch::empty({0, d_}, torch::kFloat32);
    out->partition_ids = torch::arange(2 * np, torch::kInt64);  // placeholders: add_partitions hands out the real ids
    return out;
}

Tensor PartitionManager::split_partitions_in_place(const Tensor &partition_ids) {
    require_store("split_partitions");
    if (!parent_) throw std::runtime_error("[PartitionManager] split_partitions: no parent index.");
    Tensor p = host_i64(partition_ids).reshape({-1}).contiguous();
    const int64_t np = p.size(0);
    if (np == 0) return torch::empty({0}, torch::kInt64);
    std::vector<int64_t> sz((size_t)np);
    int64_t total = 0;
    for (int64_t i = 0; i < np; i++) {
        sz[(size_t)i] = get_partition_size(p[i].item<int64_t>());
        if (sz[(size_t)i] < 8) return Tensor();  // (the host path keeps such a partition whole: rare, left to it)
        total += sz[(size_t)i];
    }
    const auto fopt = torch::TensorOptions().dtype(torch::kFloat32).device(torch::kCUDA, 0);
    const auto iopt = torch::TensorOptions().dtype(torch::kInt64).device(torch::kCUDA, 0);
    // (torch only allocates here: every kernel and copy below runs on the library's stream, synchronised before a buffer changes hands)
    Tensor x = torch::empty({total, (int64_t)d_}, fopt), idd = torch::empty({total}, iopt), assign = torch::empty({total}, iopt);
    Tensor cents_dev = torch::empty({2 * np, (int64_t)d_}, fopt);
    qk_check(lists_.get_lists(p.data_ptr<int64_t>(), np, x.data_ptr<float>(), idd.data_ptr<int64_t>(), QK_MEM_DEVICE));
    // the same call split_partitions makes per partition, on device memory.  The 2-means are independent problems of ~0.5 ms of
    // launch latency each (5 Lloyd iterations on a 512-row sample, a synchronisation per iteration) -- hundreds of them when the
    // window first fills: several in flight at once on worker contexts (a private stream and k-means scratch each), the same bits
    std::vector<int64_t> start((size_t)np + 1, 0);
    for (int64_t i = 0; i < np; i++) start[(size_t)i + 1] = start[(size_t)i] + sz[(size_t)i];
    float *xp = x.data_ptr<float>(), *cp = cents_dev.data_ptr<float>();
    int64_t *ap_dev = assign.data_ptr<int64_t>();
    auto one = [&](qk_ctx *c, int64_t i) {
        return qk_kmeans(c, xp + start[(size_t)i] * d_, sz[(size_t)i], d_, 2, metric_, DEFAULT_NITER, 1234ULL, cp + 2 * i * d_,
                         ap_dev + start[(size_t)i], QK_MEM_DEVICE);
    };
    int nw = 8;
    if (const char *e = std::getenv("QUAKE_SPLIT_THREADS")) nw = std::max(1, atoi(e));
    nw = (int)std::min<int64_t>(nw, np);
    if (nw >= 2) {
        static std::mutex mu;
        static std::vector<qk_ctx *> workers;  // (kept for the life of the process, like the index's own context)
        std::vector<qk_ctx *> mine;
        {
            std::lock_guard<std::mutex> g(mu);
            while ((int)workers.size() < nw) {
                qk_ctx *c = nullptr;
                qk_check(qk_ctx_create(0, &c));
                workers.push_back(c);
            }
            mine.assign(workers.begin(), workers.begin() + nw);
            qk_check(qk_ctx_synchronize(ctx_));  // the rows were gathered on the index's stream
            std::vector<int> rc((size_t)nw, QK_OK);
            std::vector<std::string> msg((size_t)nw);
            std::vector<std::thread> th;
            for (int w = 0; w < nw; w++)
                th.emplace_back([&, w] {
                    for (int64_t i = w; i < np && rc[(size_t)w] == QK_OK; i += nw) rc[(size_t)w] = one(mine[(size_t)w], i);
                    if (rc[(size_t)w] == QK_OK) rc[(size_t)w] = qk_ctx_synchronize(mine[(size_t)w]);
                    if (rc[(size_t)w] != QK_OK) msg[(size_t)w] = qk_last_error();  // (the message is the calling thread's)
                });
            for (auto &t : th) t.join();
            for (int w = 0; w < nw; w++)
                if (rc[(size_t)w] != QK_OK) throw std::runtime_error("[PartitionManager] split_partitions: " + msg[(size_t)w]);
        }
    } else {
        for (int64_t i = 0; i < np; i++) qk_check(one(ctx_, i));
    }
    qk_check(qk_ctx_synchronize(ctx_));
    Tensor ah = assign.cpu(), cents = cents_dev.cpu();
    Tensor new_ids = torch::arange(curr_partition_id_, curr_partition_id_ + 2 * np, torch::kInt64);
    curr_partition_id_ += 2 * np;
    // list number of every row: half a[r] of partition i -> new_ids[2 i + a[r]]; rows stay in their order, so every half keeps the
    // order index_select(nonzero(a == h)) gives it
    Tensor listno = torch::empty({total}, torch::kInt64);
    {
        const int64_t *ap = ah.data_ptr<int64_t>();
        int64_t *lp = listno.data_ptr<int64_t>();
        int64_t r = 0;
        for (int64_t i = 0; i < np; i++)
            for (int64_t j = 0; j < sz[(size_t)i]; j++, r++) lp[r] = curr_partition_id_ - 2 * np + 2 * i + (ap[r] != 0 ? 1 : 0);
    }
    Tensor listno_dev = listno.to(x.device());
    torch::cuda::synchronize(0);
    parent_->remove(p);
    for (int64_t i = 0; i < np; i++) qk_check(lists_.remove_list(p[i].item<int64_t>()));
    parent_->add(cents, new_ids);
    for (int64_t j = 0; j < 2 * np; j++) qk_check(lists_.add_list(new_ids[j].item<int64_t>()));
    qk_check(lists_.add_batch(total, idd.data_ptr<int64_t>(), x.data_ptr<float>(), listno_dev.data_ptr<int64_t>(), QK_MEM_DEVICE));
    return new_ids;  // (the rows' ids never left the index: resident_ids_ is unchanged)
}

bool PartitionManager::delete_partitions_in_place(const Tensor &partition_ids) {
    require_store("delete_partitions");
    if (!parent_) throw std::runtime_error("[PartitionManager] delete_partitions: no parent index.");
    Tensor p = host_i64(partition_ids).reshape({-1}).contiguous();
    const int64_t np = p.size(0);
    const int64_t *pp = p.data_ptr<int64_t>();
    int64_t total = 0;
    for (int64_t i = 0; i < np; i++) total += get_partition_size(pp[i]);
    if (np == 0 || total == 0 || np >= nlist()) return false;
    const auto fopt = torch::TensorOptions().dtype(torch::kFloat32).device(torch::kCUDA, 0);
    const auto iopt = torch::TensorOptions().dtype(torch::kInt64).device(torch::kCUDA, 0);
    Tensor x = torch::empty({total, (int64_t)d_}, fopt), idd = torch::empty({total}, iopt), assign = torch::empty({total}, iopt);
    qk_check(lists_.get_lists(pp, np, x.data_ptr<float>(), idd.data_ptr<int64_t>(), QK_MEM_DEVICE));
    parent_->remove(p);
    for (int64_t i = 0; i < np; i++) qk_check(lists_.remove_list(pp[i]));
    // PartitionManager::add(vectors, ids, {}, check_uniques = false) (:533-552): the nearest remaining centroid of every row
    qk_check(qk_coarse(ctx_, parent_->store(), x.data_ptr<float>(), total, 1, metric_, assign.data_ptr<int64_t>(), nullptr, QK_MEM_DEVICE));
    qk_check(lists_.add_batch(total, idd.data_ptr<int64_t>(), x.data_ptr<float>(), assign.data_ptr<int64_t>(), QK_MEM_DEVICE));
    return true;  // (the rows' ids never left the index: resident_ids_ is unchanged)
}

void PartitionManager::add_partitions(shared_ptr<Clustering> c) {  // :489-520
    require_store("add_partitions");
    if (!parent_) throw std::runtime_error("[PartitionManager] add_partitions: no parent index.");
    const int64_t n = c->nlist();
    Tensor new_ids = torch::arange(curr_partition_id_, curr_partition_id_ + n, torch::kInt64);
    curr_partition_id_ += n;
    c->partition_ids = new_ids;
    parent_->add(c->centroids, new_ids);
    // ONE ingest for all the new partitions (rows in partition order: a partition's append order is its own row order, what one
    // add_entries per partition leaves behind -- each of those was a transfer, a launch and a synchronisation)
    std::vector<Tensor> vs, is_, as_;
    for (int64_t i = 0; i < n; i++) {
        const int64_t pid = new_ids[i].item<int64_t>();
        qk_check(lists_.add_list(pid));
        Tensor v = host_f32(c->vectors[(size_t)i]), id = host_i64(c->vector_ids[(size_t)i]).reshape({-1});
        if (!id.size(0)) continue;
        vs.push_back(v);
        is_.push_back(id);
        as_.push_back(torch::full({id.size(0)}, pid, torch::kInt64));
    }
    if (!vs.empty()) {
        Tensor v = torch::cat(vs, 0).contiguous(), id = torch::cat(is_, 0).contiguous(), a = torch::cat(as_, 0).contiguous();
        qk_check(lists_.add_batch(id.size(0), id.data_ptr<int64_t>(), v.data_ptr<float>(), a.data_ptr<int64_t>(), QK_MEM_HOST));
        const int64_t *ip = id.data_ptr<int64_t>();
        resident_ids_.insert(ip, ip + id.size(0));
    }
}

void PartitionManager::delete_partitions(const Tensor &partition_ids, bool reassign) {  // :522-554
    require_store("delete_partitions");
    if (!parent_) throw std::runtime_error("[PartitionManager] delete_partitions: no parent index.");
    auto sel = select_partitions(partition_ids, true);
    Tensor p = host_i64(partition_ids).reshape({-1});
    parent_->remove(p);
    for (int64_t i = 0; i < p.size(0); i++) {
        qk_check(lists_.remove_list(p[i].item<int64_t>()));
        Tensor id = sel->vector_ids[(size_t)i];
        Tensor idc = id.contiguous();
        const int64_t *ip = idc.data_ptr<int64_t>();
        for (int64_t j = 0; j < idc.size(0); j++) resident_ids_.erase(ip[j]);
    }
    if (reassign && sel->ntotal() > 0) {
        Tensor v = torch::cat(sel->vectors, 0), id = torch::cat(sel->vector_ids, 0);
        add(v, id);
    }
}

void PartitionManager::refine_partitions(Tensor partition_ids, int refinement_iterations) {  // :446-487
    require_store("refine_partitions");
    if (!parent_) return;
    if (!partition_ids.defined() || partition_ids.numel() == 0) partition_ids = get_partition_ids();
    if (partition_ids.size(0) == 0) return;
    Tensor pids = host_i64(partition_ids).reshape({-1});
    Tensor cent = parent_->get(pids);
    qk_check(lists_.refine_lists(pids.data_ptr<int64_t>(), pids.size(0), cent.data_ptr<float>(), metric_, refinement_iterations,
                                   QK_MEM_HOST));
    parent_->modify(pids, cent);  // :478
}

// :557-603: partition i -> worker i % num_workers.  Here: the partitions move into a device group of num_workers members
// (partition p -> member p % num_workers), device to device, and every later call goes to the member that holds the partition.
void PartitionManager::distribute_partitions(int num_workers) {
    if (num_workers <= 0 || !lists_) return;
    planned_workers_ = num_workers;
    if (!parent_) return;  // a flat index is ONE partition: nothing to distribute, its one store keeps serving it
    if (lists_.group) {
        if (qk_group_size(lists_.group) == num_workers) return;
        throw std::runtime_error("[PartitionManager] distribute_partitions: partitions are already distributed over " +
                                 std::to_string(qk_group_size(lists_.group)) + " workers.");
    }
    qk_store *old = lists_.store;
    lists_.store = nullptr;
    try {
        make_group(num_workers, d_);
        int64_t nl = 0;
        qk_check(qk_store_list_ids(old, nullptr, &nl));
        std::vector<int64_t> lists((size_t)nl);
        if (nl) qk_check(qk_store_list_ids(old, lists.data(), &nl));
        for (int64_t p : lists) {
            int64_t sz = 0;
            qk_check(qk_store_list_size(old, p, &sz));
            qk_check(qk_group_add_list(lists_.group, p));
            if (!sz) continue;
            Tensor v = torch::empty({sz, (int64_t)d_}, torch::TensorOptions().dtype(torch::kFloat32).device(torch::kCUDA, 0));
            Tensor id = torch::empty({sz}, torch::TensorOptions().dtype(torch::kInt64).device(torch::kCUDA, 0));
            qk_check(qk_store_get_list(old, p, v.data_ptr<float>(), id.data_ptr<int64_t>(), QK_MEM_DEVICE));
            qk_check(qk_ctx_synchronize(qk_device_context(0)));
            qk_check(qk_group_add_entries(lists_.group, p, sz, id.data_ptr<int64_t>(), v.data_ptr<float>(), QK_MEM_DEVICE));
            qk_check(qk_store_remove_list(old, p));
        }
    } catch (...) {
        if (lists_.group) qk_group_destroy(lists_.group);
        lists_.group = nullptr;
        lists_.store = old;
        throw;
    }
    qk_store_destroy(old);
}

void PartitionManager::set_partition_core_id(int64_t partition_id, int core_id) { core_of_[partition_id] = core_id; }

int PartitionManager::get_partition_core_id(int64_t partition_id) {
    auto it = core_of_.find(partition_id);
    if (it != core_of_.end()) return it->second;
    return lists_.group ? qk_group_owner(lists_.group, partition_id) : -1;
}

int64_t PartitionManager::ntotal() const { return lists_ ? lists_.ntotal() : 0; }
int64_t PartitionManager::nlist() const { return lists_ ? lists_.nlist() : 0; }
int PartitionManager::d() const { return d_; }

bool PartitionManager::validate() {
    if (!lists_) return false;
    if ((int64_t)resident_ids_.size() != ntotal()) return false;
    Tensor ids = get_ids();
    const int64_t *ip = ids.data_ptr<int64_t>();
    std::unordered_set<int64_t> seen(ip, ip + ids.size(0));
    return (int64_t)seen.size() == ids.size(0);
}

// on-disk format of the reference's "partitions" file (dynamic_inverted_list.cpp:338-520): 32-byte header
// {magic "LNID", version 3, nlist, code_size, npartitions}, offsets [n+1], partition ids [n], then [codes | ids] chunks
void PartitionManager::save(const std::string &path) {
    require_store("save");
    Tensor lists = get_partition_ids();
    const int64_t nl = lists.size(0);
    std::ofstream ofs(path, std::ios::binary);
    if (!ofs.is_open()) throw std::runtime_error("Could not open file for writing: " + path);
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
        sizes[(size_t)i] = get_partition_size(lists[i].item<int64_t>());
        offsets[(size_t)i + 1] = offsets[(size_t)i] + (uint64_t)sizes[(size_t)i] * (code_size + 8);
    }
    ofs.write((const char *)offsets.data(), (std::streamsize)(offsets.size() * 8));
    for (int64_t i = 0; i < nl; i++) {
        const uint64_t pid = (uint64_t)lists[i].item<int64_t>();
        ofs.write((const char *)&pid, 8);
    }
    for (int64_t i = 0; i < nl; i++) {  // one partition at a time: the host never holds more than a partition
        std::vector<float> v((size_t)sizes[(size_t)i] * d_);
        std::vector<int64_t> id((size_t)sizes[(size_t)i]);
        if (sizes[(size_t)i]) qk_check(lists_.get_list(lists[i].item<int64_t>(), v.data(), id.data(), QK_MEM_HOST));
        ofs.write((const char *)v.data(), (std::streamsize)(v.size() * 4));
        ofs.write((const char *)id.data(), (std::streamsize)(id.size() * 8));
    }
}

void PartitionManager::load(const std::string &path) {
    std::ifstream ifs(path, std::ios::binary);
    if (!ifs.is_open()) throw std::runtime_error("Could not open file for reading: " + path);
    uint32_t magic = 0, version = 0;
    uint64_t nlist64 = 0, code_size = 0, nparts = 0;
    ifs.read((char *)&magic, 4);
    ifs.read((char *)&version, 4);
    if (magic != 0x44494E4C) throw std::runtime_error("Invalid file format (bad magic number).");
    if (version != 3) throw std::runtime_error("Unsupported file version: " + std::to_string(version));
    ifs.read((char *)&nlist64, 8);
    ifs.read((char *)&code_size, 8);
    ifs.read((char *)&nparts, 8);
    // nothing is allocated on the word of a header field before it has been checked against the file's size
    const uint64_t file_size = (uint64_t)std::filesystem::file_size(path);
    if (!ifs || code_size == 0 || code_size % 4 != 0 || nparts > file_size / 16 || 32 + 16 * nparts + 8 > file_size)
        throw std::runtime_error("Invalid file format (truncated offset / partition id table).");
    std::vector<uint64_t> offsets((size_t)nparts + 1), pids((size_t)nparts);
    ifs.read((char *)offsets.data(), (std::streamsize)(offsets.size() * 8));
    ifs.read((char *)pids.data(), (std::streamsize)(pids.size() * 8));
    const int d = (int)(code_size / 4);
    reset_store(d);
    const uint64_t rec = code_size + 8;
    int64_t max_pid = -1;
    if (!ifs) throw std::runtime_error("Invalid file format (truncated offset / partition id table).");
    const uint64_t start_of_chunks = 32 + 8 * (nparts + 1) + 8 * nparts;
    for (uint64_t i = 0; i < nparts; i++) {
        if (offsets[i + 1] < offsets[i]) throw std::runtime_error("Invalid file format (partition offsets are not ascending).");
        const uint64_t chunk = offsets[i + 1] - offsets[i];
        if (start_of_chunks + offsets[i + 1] > file_size) throw std::runtime_error("Invalid file format (truncated partition data).");
        if (chunk % rec != 0) throw std::runtime_error("Partition chunk size not divisible by (code_size+sizeof(idx_t))");
        const int64_t nv = (int64_t)(chunk / rec);
        ifs.seekg((std::streamoff)(start_of_chunks + offsets[i]), std::ios::beg);  // dynamic_inverted_list.cpp:494
        std::vector<float> v((size_t)nv * d);
        std::vector<int64_t> id((size_t)nv);
        ifs.read((char *)v.data(), (std::streamsize)(v.size() * 4));
        ifs.read((char *)id.data(), (std::streamsize)(id.size() * 8));
        if (!ifs) throw std::runtime_error("Invalid file format (truncated partition data).");
        qk_check(lists_.add_list((int64_t)pids[i]));
        if (nv) qk_check(lists_.add_entries((int64_t)pids[i], nv, id.data(), v.data(), QK_MEM_HOST));
        resident_ids_.insert(id.begin(), id.end());
        max_pid = std::max<int64_t>(max_pid, (int64_t)pids[i]);
    }
    curr_partition_id_ = max_pid + 1;
}

}  // namespace quake_amd
