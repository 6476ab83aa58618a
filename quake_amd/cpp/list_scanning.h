// list_scanning.h -- the list-scanning seam of the reference (src/cpp/include/list_scanning.h:39-366) on the device:
// TypedTopKBuffer / TopkBuffer (the append-and-flush buffer search results are merged through), create_buffers,
// scan_list (one query x one list) and batched_scan_list (a block of queries x one list).  The buffer is a host container
// with the reference's public members and methods; the two scan functions take the reference's raw host pointers, run the
// list through libquake_hip.so (a temporary one-list device store + qk_scan) and feed the list's top-k into the buffers --
// the same buffer contents after flush() as adding every row, because a buffer only ever keeps its k best.
// Order inside a buffer is the total order (distance, id), a refinement of the reference's.
#pragma once
#include <algorithm>
#include <atomic>
#include <cassert>
#include <limits>
#include <memory>
#include <mutex>
#include <utility>
#include <vector>

#include "common.h"

namespace quake_amd {

#define TOP_K_BUFFER_CAPACITY (8 * 1024)

template <typename DistanceType = float, typename IdType = int>
class TypedTopKBuffer {
public:
    int k_;
    int curr_offset_ = 0;
    std::vector<std::pair<DistanceType, IdType>> topk_;
    bool is_descending_;
    std::recursive_mutex buffer_mutex_;
    std::atomic<bool> processing_query_;
    std::atomic<int> jobs_left_;
    std::atomic<int> partitions_scanned_;

    TypedTopKBuffer(int k, bool is_descending, int buffer_capacity = TOP_K_BUFFER_CAPACITY)
        : k_(k), topk_((size_t)std::max(buffer_capacity, k)), is_descending_(is_descending), processing_query_(true), jobs_left_(0),
          partitions_scanned_(0) {
        assert(k <= buffer_capacity);
        std::fill(topk_.begin(), topk_.end(), sentinel());
    }

    void set_k(int new_k) {
        std::lock_guard<std::recursive_mutex> lock(buffer_mutex_);
        assert(new_k <= (int)topk_.size());
        k_ = new_k;
        reset();
    }
    void set_processing_query(bool v) { processing_query_.store(v, std::memory_order_relaxed); }
    bool currently_processing_query() { return processing_query_.load(std::memory_order_relaxed); }
    void set_jobs_left(int n) { jobs_left_.store(n, std::memory_order_relaxed); }
    void record_skipped_jobs(int n) { jobs_left_.fetch_sub(n, std::memory_order_relaxed); }
    void record_empty_job() { jobs_left_.fetch_sub(1, std::memory_order_relaxed); }
    bool finished_all_jobs() { return jobs_left_.load(std::memory_order_relaxed) <= 0; }
    int get_num_partitions_scanned() { return partitions_scanned_.load(std::memory_order_relaxed); }

    void reset() {
        std::lock_guard<std::recursive_mutex> lock(buffer_mutex_);
        curr_offset_ = 0;
        std::fill(topk_.begin(), topk_.begin() + k_, sentinel());
        partitions_scanned_.store(0, std::memory_order_relaxed);
    }

    void add(DistanceType distance, IdType index) {
        if (curr_offset_ >= (int)topk_.size()) flush();
        topk_[(size_t)curr_offset_++] = {distance, index};
    }

    void batch_add(DistanceType *distances, const IdType *indices, int num_values) {
        if (num_values == 0 || !currently_processing_query()) {
            jobs_left_.fetch_sub(1, std::memory_order_relaxed);
            return;
        }
        std::lock_guard<std::recursive_mutex> lock(buffer_mutex_);
        int done = 0;
        while (done < num_values) {
            if (curr_offset_ >= (int)topk_.size()) flush();
            const int room = (int)topk_.size() - curr_offset_, take = std::min(room, num_values - done);
            for (int i = 0; i < take; i++) topk_[(size_t)curr_offset_ + i] = {distances[done + i], indices[done + i]};
            curr_offset_ += take;
            done += take;
        }
        partitions_scanned_.fetch_add(1, std::memory_order_relaxed);
        jobs_left_.fetch_sub(1, std::memory_order_relaxed);
    }

    // keep the k best of what was appended, sorted best first
    DistanceType flush() {
        std::lock_guard<std::recursive_mutex> lock(buffer_mutex_);
        const auto better = [this](const std::pair<DistanceType, IdType> &a, const std::pair<DistanceType, IdType> &b) {
            if (a.first != b.first) return is_descending_ ? a.first > b.first : a.first < b.first;
            return a.second < b.second;
        };
        if (curr_offset_ > k_) {
            std::partial_sort(topk_.begin(), topk_.begin() + k_, topk_.begin() + curr_offset_, better);
            curr_offset_ = k_;
        } else {
            std::sort(topk_.begin(), topk_.begin() + curr_offset_, better);
        }
        return topk_[(size_t)std::max(k_ - 1, 0)].first;
    }

    std::vector<DistanceType> get_topk() {
        flush();
        std::vector<DistanceType> out((size_t)std::min(curr_offset_, k_));
        for (size_t i = 0; i < out.size(); i++) out[i] = topk_[i].first;
        return out;
    }
    std::vector<IdType> get_topk_indices() {
        flush();
        std::vector<IdType> out((size_t)std::min(curr_offset_, k_));
        for (size_t i = 0; i < out.size(); i++) out[i] = topk_[i].second;
        return out;
    }
    DistanceType get_kth_distance() {
        flush();
        return topk_[(size_t)std::max(k_ - 1, 0)].first;
    }

private:
    std::pair<DistanceType, IdType> sentinel() const {
        return {is_descending_ ? -std::numeric_limits<DistanceType>::infinity() : std::numeric_limits<DistanceType>::max(), (IdType)-1};
    }
};

using TopkBuffer = TypedTopKBuffer<float, int64_t>;

inline std::vector<shared_ptr<TopkBuffer>> create_buffers(int n, int k, bool is_descending) {  // list_scanning.h:233-239
    std::vector<shared_ptr<TopkBuffer>> buffers((size_t)n);
    for (auto &b : buffers) b = std::make_shared<TopkBuffer>(k, is_descending, 10 * k);
    return buffers;
}

// [n, k] tensors (ids, distances) out of n buffers, padded with -1 / the sentinel distance (list_scanning.h:206-231)
std::tuple<Tensor, Tensor> buffers_to_tensor(std::vector<shared_ptr<TopkBuffer>> &buffers);

// mean |intersection of the first k ids| / k (list_scanning.h:14-37)
double calculate_recall(const Tensor &ids, const Tensor &gt_ids);

// scan_list (list_scanning.h:292-311): distances from one query to every row of one list (host pointers; list_ids may be
// nullptr = row numbers) into `buffer`.  L2 adds sqrt distances, IP dot products, like the reference.
void scan_list(const float *query_vec, const float *list_vecs, const int64_t *list_ids, int list_size, int d, TopkBuffer &buffer,
               MetricType metric = METRIC_L2);

// batched_scan_list (list_scanning.h:313-366): num_queries queries against one list, buffer i receives query i's results.
void batched_scan_list(const float *query_vecs, const float *list_vecs, const int64_t *list_ids, int num_queries, int list_size,
                       int dim, std::vector<shared_ptr<TopkBuffer>> &topk_buffers, MetricType metric = METRIC_L2);

}  // namespace quake_amd
