// partition_manager.h -- PartitionManager of the C++ host mirror: the public surface of the reference's
// src/cpp/include/partition_manager.h:25-187, over the DEVICE partition store (qk_store, the counterpart of
// faiss::DynamicInvertedLists).  The reference keeps vectors in host partitions and exposes raw pointers into them
// (get_vectors); here the vectors live in HBM and every accessor copies out.  Host bookkeeping kept: the resident id
// set, the next partition id, the partition -> worker map of distribute_partitions.
#pragma once
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace quake_amd {

class QuakeIndex;

// What the reference exposes as PartitionManager::partition_store_ (faiss::DynamicInvertedLists, dynamic_inverted_list.h:25-33;
// its tests read list sizes through it, test/cpp/partition_manager.cpp:203-248): here a read-only view of the device store.
class DynamicInvertedLists {
public:
    explicit DynamicInvertedLists(qk_store *s, size_t code_size) : code_size(code_size), s_(s) {}
    size_t list_size(size_t list_no) const;  // throws like the reference when the list does not exist (:68-74)
    size_t ntotal() const { return (size_t)qk_store_ntotal(s_); }
    size_t get_nlist() const { return (size_t)qk_store_nlist(s_); }
    size_t code_size;  // bytes per vector (d * 4)

private:
    qk_store *s_;
};

class PartitionManager {
public:
    shared_ptr<QuakeIndex> parent_ = nullptr;  // index over the centroids (partition_manager.h:27)
    shared_ptr<DynamicInvertedLists> partition_store_ = nullptr;  // view of the device store (partition_manager.h:28)
    int64_t curr_partition_id_ = 0;            // next partition id to hand out
    bool debug_ = false;
    bool check_uniques_ = false;
    std::set<int64_t> resident_ids_;           // vector ids currently stored

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
    qk_store *store() const { return store_; }
    qk_ctx *ctx() const { return ctx_; }
    int metric_ = QK_METRIC_L2;  // set by the owning index (the reference reads it from the parent)

private:
    qk_ctx *ctx_ = nullptr;  // shared per-device context (not owned)
    qk_store *store_ = nullptr;
    int d_ = 0;
    std::unordered_map<int64_t, int> core_of_;
    void reset_store(int d);
    void require_store(const char *who) const;
};

}  // namespace quake_amd
