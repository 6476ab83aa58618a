// partition_manager.h -- PartitionManager of the C++ host mirror: the public surface of the reference's
// src/cpp/include/partition_manager.h:25-187, over the DEVICE partition store (qk_store, the counterpart of
// faiss::DynamicInvertedLists).  The reference keeps vectors in host partitions and exposes raw pointers into them
// (get_vectors); here the vectors live in HBM and every accessor copies out.  Host bookkeeping kept: the resident id
// set, the next partition id.  distribute_partitions(num_workers) -- the reference's partition -> core map
// (partition_manager.cpp:557-603) -- distributes the partitions over a DEVICE GROUP (qk_group: one process, num_workers members,
// partition p in member p % num_workers, member j on GPU j % #GPUs): from then on every call below goes to the member that
// holds the partition and searches run on all members at once.
#pragma once
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace quake_amd {

class QuakeIndex;

// The device lists a PartitionManager drives: ONE store, or a device group once the partitions are distributed over workers.
// Same arguments, same status codes either way (include/quake_hip.h).
struct DeviceLists {
    qk_store *store = nullptr;
    qk_group *group = nullptr;
    explicit operator bool() const { return store || group; }
    int add_list(int64_t p) const { return group ? qk_group_add_list(group, p) : qk_store_add_list(store, p); }
    int remove_list(int64_t p) const { return group ? qk_group_remove_list(group, p) : qk_store_remove_list(store, p); }
    int add_entries(int64_t p, int64_t n, const int64_t *ids, const float *v, int mem) const {
        return group ? qk_group_add_entries(group, p, n, ids, v, mem) : qk_store_add_entries(store, p, n, ids, v, mem);
    }
    int add_batch(int64_t n, const int64_t *ids, const float *v, const int64_t *assign, int mem) const {
        return group ? qk_group_add_batch(group, n, ids, v, assign, mem) : qk_store_add_batch(store, n, ids, v, assign, mem);
    }
    int build_csr(int64_t nlist, const int64_t *off, const int64_t *ids, const float *v, int mem) const {
        return group ? qk_group_build_csr(group, nlist, off, ids, v, mem) : qk_store_build_csr(store, nlist, off, ids, v, mem);
    }
    int remove_ids(int64_t n, const int64_t *ids, int64_t *removed) const {
        return group ? qk_group_remove_ids(group, n, ids, removed) : qk_store_remove_ids(store, n, ids, removed);
    }
    int list_size(int64_t p, int64_t *out) const { return group ? qk_group_list_size(group, p, out) : qk_store_list_size(store, p, out); }
    int list_ids(int64_t *out, int64_t *n) const { return group ? qk_group_list_ids(group, out, n) : qk_store_list_ids(store, out, n); }
    int get_list(int64_t p, float *v, int64_t *ids, int mem) const {
        return group ? qk_group_get_list(group, p, v, ids, mem) : qk_store_get_list(store, p, v, ids, mem);
    }
    int get_lists(const int64_t *ps, int64_t n, float *v, int64_t *ids, int mem) const {
        return group ? qk_group_get_lists(group, ps, n, v, ids, mem) : qk_store_get_lists(store, ps, n, v, ids, mem);
    }
    int get_vector(int64_t id, float *v, int *found) const {
        return group ? qk_group_get_vector(group, id, v, found) : qk_store_get_vector(store, id, v, found);
    }
    int refine_lists(const int64_t *lists, int64_t m, float *c, int metric, int iters, int mem) const {
        return group ? qk_group_refine_lists(group, lists, m, c, metric, iters, mem) : qk_store_refine_lists(store, lists, m, c, metric, iters, mem);
    }
    // pending modifications become visible to searches now (list table upload) instead of inside the next query (qk_store_publish);
    // a group's members upload theirs at their next scan
    int publish() const { return (!group && store) ? qk_store_publish(store) : 0; }
    int64_t ntotal() const { return group ? qk_group_ntotal(group) : qk_store_ntotal(store); }
    int64_t nlist() const { return group ? qk_group_nlist(group) : qk_store_nlist(store); }
};

// What the reference exposes as PartitionManager::partition_store_ (faiss::DynamicInvertedLists, dynamic_inverted_list.h:25-33;
// its tests read list sizes through it, test/cpp/partition_manager.cpp:203-248): here a read-only view of the device store.
class DynamicInvertedLists {
public:
    explicit DynamicInvertedLists(const DeviceLists *s, size_t code_size) : code_size(code_size), s_(s) {}
    size_t list_size(size_t list_no) const;  // throws like the reference when the list does not exist (:68-74)
    size_t ntotal() const { return (size_t)s_->ntotal(); }
    size_t get_nlist() const { return (size_t)s_->nlist(); }
    size_t code_size;  // bytes per vector (d * 4)

private:
    const DeviceLists *s_;  // the manager's lists (follows a store -> group change)
};

// The ids currently stored (the reference keeps a std::set<int64_t>, partition_manager.h; a red-black tree of 10M nodes was 5 of
// the 10.8 s a 10M-vector build took here): a bitmap over the non-negative ids below 2^31 -- the range add() admits -- and a
// std::set for anything else.  The subset of std::set's interface the manager uses.
class IdSet {
public:
    struct InsertResult {
        bool second;
    };
    InsertResult insert(int64_t id) {
        if (id < 0 || id > (int64_t)0x7FFFFFFF) return InsertResult{other_.insert(id).second};
        const size_t w = (size_t)(id >> 6);
        if (w >= bits_.size()) bits_.resize(std::max(w + 1, bits_.size() + bits_.size() / 2), 0);
        const uint64_t m = 1ull << (id & 63);
        if (bits_[w] & m) return InsertResult{false};
        bits_[w] |= m;
        n_++;
        return InsertResult{true};
    }
    template <class It>
    void insert(It a, It b) {
        for (; a != b; ++a) insert((int64_t)*a);
    }
    size_t count(int64_t id) const {
        if (id < 0 || id > (int64_t)0x7FFFFFFF) return other_.count(id);
        const size_t w = (size_t)(id >> 6);
        return w < bits_.size() && ((bits_[w] >> (id & 63)) & 1ull) ? 1 : 0;
    }
    size_t erase(int64_t id) {
        if (id < 0 || id > (int64_t)0x7FFFFFFF) return other_.erase(id);
        const size_t w = (size_t)(id >> 6);
        const uint64_t m = 1ull << (id & 63);
        if (w >= bits_.size() || !(bits_[w] & m)) return 0;
        bits_[w] &= ~m;
        n_--;
        return 1;
    }
    size_t size() const { return n_ + other_.size(); }
    void clear() {
        bits_.clear();
        other_.clear();
        n_ = 0;
    }

private:
    std::vector<uint64_t> bits_;
    std::set<int64_t> other_;
    size_t n_ = 0;
};

class PartitionManager {
public:
    shared_ptr<QuakeIndex> parent_ = nullptr;  // index over the centroids (partition_manager.h:27)
    shared_ptr<DynamicInvertedLists> partition_store_ = nullptr;  // view of the device store (partition_manager.h:28)
    int64_t curr_partition_id_ = 0;            // next partition id to hand out
    bool debug_ = false;
    bool check_uniques_ = false;
    IdSet resident_ids_;                       // vector ids currently stored

    PartitionManager();
    ~PartitionManager();
    PartitionManager(const PartitionManager &) = delete;
    PartitionManager &operator=(const PartitionManager &) = delete;

    void init_partitions(shared_ptr<QuakeIndex> parent, shared_ptr<Clustering> partitions, bool check_uniques = true);
    // bulk form used by QuakeIndex::build: lists 0..nlist-1 from a CSR arena (no per-cluster tensors in between)
    void init_from_csr(shared_ptr<QuakeIndex> parent, const Tensor &offsets, const Tensor &ids, const Tensor &vectors);
    shared_ptr<ModifyTimingInfo> add(const Tensor &vectors, const Tensor &vector_ids, const Tensor &assignments = Tensor(),
                                     bool check_uniques = true);
    shared_ptr<ModifyTimingInfo> remove(const Tensor &ids);
    Tensor get(const Tensor &ids);
    shared_ptr<Clustering> split_partitions(const Tensor &partition_ids);
    // split_partitions + delete_partitions(.., false) + add_partitions in one step WITHOUT the host round trip of the rows (what
    // perform_maintenance does with its splits): lists extracted on the device, the same 2-means each, one ingest for all the halves.
    // Returns the new partition ids ([2 * n]), or an undefined tensor when a partition is too small for the device path (the
    // caller then takes the three calls above).  Same partitions, same row order, same centroid bits as the three calls.
    Tensor split_partitions_in_place(const Tensor &partition_ids);
    // delete_partitions(partition_ids, reassign = true) with the rows staying on the device: extracted there, ranked against the
    // remaining centroids there, re-ingested from there (the other path carries every row to the host and back).  Same lists, same
    // row order.  false: nothing done (no rows), the caller takes delete_partitions.
    bool delete_partitions_in_place(const Tensor &partition_ids);
    void refine_partitions(Tensor partition_ids = Tensor(), int refinement_iterations = 0);
    void delete_partitions(const Tensor &partition_ids, bool reassign = false);
    void add_partitions(shared_ptr<Clustering> partitions);
    shared_ptr<Clustering> select_partitions(const Tensor &partition_ids, bool copy = false);
    void distribute_partitions(int num_workers);
    void set_partition_core_id(int64_t partition_id, int core_id);
    int get_partition_core_id(int64_t partition_id);
    int64_t ntotal() const;
    int64_t nlist() const;
    int d() const;
    Tensor get_partition_sizes(Tensor partition_ids = Tensor());
    std::vector<int64_t> get_partition_sizes(std::vector<int64_t> partition_ids);
    int64_t get_partition_size(int64_t partition_id);
    Tensor get_partition_ids();
    Tensor get_ids();
    bool validate();
    void save(const std::string &path);
    void load(const std::string &path);

    // device side
    qk_store *store() const { return lists_.store; }    // the one store (nullptr once the partitions are distributed)
    qk_group *group() const { return lists_.group; }    // the device group (nullptr before distribute_partitions)
    bool has_lists() const { return (bool)lists_; }
    const DeviceLists &lists() const { return lists_; }
    qk_ctx *ctx() const { return ctx_; }                // the shared context of device 0: parent searches, k-means of a split
    qk_ctx *search_ctx() const;                         // what a search is ordered on: ctx(), or the group's lead member
    int num_workers() const { return lists_.group ? qk_group_size(lists_.group) : 0; }
    // the number of workers the partitions will be distributed over, told BEFORE they are initialised (QuakeIndex::build / load
    // know it from IndexBuildParams::num_workers): the lists then go straight to their members instead of through one store
    void plan_workers(int num_workers) { planned_workers_ = num_workers; }
    int metric_ = QK_METRIC_L2;  // set by the owning index (the reference reads it from the parent)

private:
    qk_ctx *ctx_ = nullptr;  // shared per-device context (not owned)
    DeviceLists lists_;
    int d_ = 0;
    int planned_workers_ = 0;
    std::unordered_map<int64_t, int> core_of_;  // set_partition_core_id overrides (bookkeeping only)
    void reset_store(int d);
    void release_lists();
    void make_group(int num_workers, int d);
    void require_store(const char *who) const;
};

}  // namespace quake_amd
