// list_scanning.cpp -- device-backed scan_list / batched_scan_list (see list_scanning.h).
#include "list_scanning.h"

#include <numeric>
#include <unordered_set>

namespace quake_amd {

std::tuple<Tensor, Tensor> buffers_to_tensor(std::vector<shared_ptr<TopkBuffer>> &buffers) {
    const int64_t n = (int64_t)buffers.size();
    const int k = n ? buffers[0]->k_ : 0;
    Tensor ids = torch::full({n, k}, -1, torch::kInt64);
    Tensor dist = torch::empty({n, k}, torch::kFloat32);
    for (int64_t i = 0; i < n; i++) {
        auto d = buffers[(size_t)i]->get_topk();
        auto id = buffers[(size_t)i]->get_topk_indices();
        const float pad = buffers[(size_t)i]->is_descending_ ? -std::numeric_limits<float>::infinity() : std::numeric_limits<float>::infinity();
        for (int j = 0; j < k; j++) {
            dist[i][j] = j < (int)d.size() ? d[(size_t)j] : pad;
            if (j < (int)id.size()) ids[i][j] = id[(size_t)j];
        }
    }
    return std::make_tuple(ids, dist);
}

double calculate_recall(const Tensor &ids, const Tensor &gt_ids) {
    Tensor a = host_i64(ids), g = host_i64(gt_ids);
    const int64_t n = a.size(0), k = a.size(1);
    if (n == 0 || k == 0) return 0.0;
    double hits = 0;
    for (int64_t i = 0; i < n; i++) {
        std::unordered_set<int64_t> truth(g[i].data_ptr<int64_t>(), g[i].data_ptr<int64_t>() + std::min<int64_t>(k, g.size(1)));
        for (int64_t j = 0; j < k; j++) hits += truth.count(a[i][j].item<int64_t>()) ? 1 : 0;
    }
    return hits / (double)(n * k);
}

void batched_scan_list(const float *query_vecs, const float *list_vecs, const int64_t *list_ids, int num_queries, int list_size,
                       int dim, std::vector<shared_ptr<TopkBuffer>> &topk_buffers, MetricType metric) {
    if (list_size == 0 || list_vecs == nullptr) {  // list_scanning.h:321-324: empty list, nothing to add
        for (int i = 0; i < num_queries; i++) topk_buffers[(size_t)i]->batch_add(nullptr, nullptr, 0);
        return;
    }
    if (num_queries <= 0) return;
    qk_ctx *ctx = qk_device_context(0);
    qk_store *store = nullptr;
    qk_check(qk_store_create(ctx, dim, &store));
    try {
        std::vector<int64_t> rows;
        if (!list_ids) {
            rows.resize((size_t)list_size);
            std::iota(rows.begin(), rows.end(), (int64_t)0);
            list_ids = rows.data();
        }
        const int64_t offs[2] = {0, list_size};
        qk_check(qk_store_build_csr(store, 1, offs, list_ids, list_vecs, QK_MEM_HOST));
        const int k = topk_buffers[0]->k_;
        const int k_max = std::min(k, list_size);  // :327-328
        std::vector<int64_t> pids((size_t)num_queries, 0), out_i((size_t)num_queries * k_max);
        std::vector<float> out_d((size_t)num_queries * k_max);
        qk_check(qk_scan(ctx, store, query_vecs, num_queries, pids.data(), 1, k_max, (int)metric, out_i.data(), out_d.data(), QK_MEM_HOST,
                         nullptr));
        for (int i = 0; i < num_queries; i++) {
            int m = 0;
            while (m < k_max && out_i[(size_t)i * k_max + m] >= 0) m++;
            topk_buffers[(size_t)i]->batch_add(out_d.data() + (size_t)i * k_max, out_i.data() + (size_t)i * k_max, m);
        }
    } catch (...) {
        qk_store_destroy(store);
        throw;
    }
    qk_store_destroy(store);
}

void scan_list(const float *query_vec, const float *list_vecs, const int64_t *list_ids, int list_size, int d, TopkBuffer &buffer,
               MetricType metric) {
    if (list_size <= 0 || list_vecs == nullptr) return;
    // one query: the list's k best go through add() (the reference add()s every row; the buffer keeps k of them either way)
    auto tmp = std::make_shared<TopkBuffer>(std::min(buffer.k_, list_size), buffer.is_descending_, std::max(10 * buffer.k_, 16));
    std::vector<shared_ptr<TopkBuffer>> one{tmp};
    batched_scan_list(query_vec, list_vecs, list_ids, 1, list_size, d, one, metric);
    auto dist = tmp->get_topk();
    auto ids = tmp->get_topk_indices();
    for (size_t i = 0; i < dist.size(); i++) buffer.add(dist[i], ids[i]);
}

}  // namespace quake_amd
