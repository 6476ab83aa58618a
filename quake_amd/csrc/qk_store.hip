// qk_store.hip -- device partition store: the tile-major arena the scan kernels stream.
//
// Mirrors what faiss::DynamicInvertedLists / IndexPartition provide to the reference's hot path
// (src/cpp/include/dynamic_inverted_list.h:25-33, src/cpp/include/index_partition.h:19-32): per-list
// vectors + ids with append and swap-with-last remove (index_partition.cpp:52-59,79-102).  The memory
// layout is NOT the reference's row-major malloc array: it is the MFMA-fragment tile-major layout
// described in qk_internal.h / DESIGN.md section 4, plus a per-row squared norm.
#include "qk_internal.h"

#include <unordered_map>
#include <atomic>

#include <algorithm>
#include <cstring>
#include <functional>
#include <thread>
#include <unordered_set>

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t csr_find(const int64_t *offsets, int64_t nlist, int64_t i) {
    // largest p with offsets[p] <= i  (offsets[nlist] > i guaranteed by the caller)
    int64_t lo = 0, hi = nlist;
    while (hi - lo > 1) {
        int64_t mid = (lo + hi) >> 1;
        if (offsets[mid] <= i)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

// dst row of source row i:  rowmap_off == nullptr -> row0 + i ;  else CSR: partition p = find(i), row = part_row[p] + (i - offsets[p])
struct IngestMap {
    int64_t row0;
    const int64_t *offsets;   // [nlist+1] source-row CSR (device) or nullptr
    const int64_t *part_row;  // [nlist] arena row of each partition's first NEW row
    int64_t nlist;
    int64_t src_base;         // global index of src[0] within the CSR numbering
    const int64_t *rows;      // explicit destination row of every source row (device) or nullptr
};

__device__ __forceinline__ int64_t ingest_dst_row(const IngestMap &m, int64_t i) {
    if (m.rows) return m.rows[i];
    if (!m.offsets) return m.row0 + i;
    int64_t gi = m.src_base + i;
    int64_t p = csr_find(m.offsets, m.nlist, gi);
    const int64_t base = m.part_row[p];  // < 0: a list this store does not hold (qk_store_csr_begin with mod > 1)
    return base < 0 ? -1 : base + (gi - m.offsets[p]);
}

__global__ void k_ingest_vecs(const float *__restrict__ src, int64_t n, int d, int nblk, float4 *__restrict__ vecs, IngestMap m) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t per_row = (int64_t)nblk * 4;
    if (idx >= n * per_row) return;
    int64_t i = idx / per_row;
    int rem = (int)(idx - i * per_row);
    int c = rem >> 2, g = rem & 3;
    int64_t row = ingest_dst_row(m, i);
    if (row < 0) return;
    int64_t tile = row >> 4;
    int r = (int)(row & 15);
    const float *s = src + i * d;
    float4 v;
    int col = 16 * c + g;
    v.x = col < d ? s[col] : 0.0f;
    v.y = col + 4 < d ? s[col + 4] : 0.0f;
    v.z = col + 8 < d ? s[col + 8] : 0.0f;
    v.w = col + 12 < d ? s[col + 12] : 0.0f;
    vecs[(tile * nblk + c) * 64 + g * 16 + r] = v;
}

// canonical squared norm: one k-ordered fmaf chain per row (oracle: qo_row_norms)
__global__ void k_ingest_norms_ids(const float *__restrict__ src, const int64_t *__restrict__ src_ids, int64_t n, int d,
                                   float *__restrict__ norms, int64_t *__restrict__ ids, IngestMap m) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t row = ingest_dst_row(m, i);
    if (row < 0) return;
    const float *s = src + i * d;
    float acc = 0.0f;
    for (int k = 0; k < d; k++) acc = __fmaf_rn(s[k], s[k], acc);
    norms[row] = acc;
    if (ids && src_ids) ids[row] = src_ids[i];
}

__global__ void k_extract(const float *__restrict__ vecs, int nblk, int d, int64_t row0, const int64_t *__restrict__ rows,
                          int64_t n, float *__restrict__ dst) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * d) return;
    int64_t i = idx / d;
    int col = (int)(idx - i * d);
    int64_t row = rows ? rows[i] : row0 + i;
    int64_t tile = row >> 4;
    int r = (int)(row & 15);
    int c = col >> 4, w = col & 15, g = w & 3, t = w >> 2;
    dst[idx] = vecs[(((tile * nblk + c) * 64) + g * 16 + r) * 4 + t];
}

__global__ void k_move_rows(float4 *vecs, float *norms, int64_t *ids, int nblk, const int64_t *__restrict__ dstr,
                            const int64_t *__restrict__ srcr, int64_t n) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t per_row = (int64_t)nblk * 4;
    if (idx >= n * per_row) return;
    int64_t i = idx / per_row;
    int rem = (int)(idx - i * per_row);
    int c = rem >> 2, g = rem & 3;
    int64_t dr = dstr[i], sr = srcr[i];
    vecs[((dr >> 4) * nblk + c) * 64 + g * 16 + (dr & 15)] = vecs[((sr >> 4) * nblk + c) * 64 + g * 16 + (sr & 15)];
    if (rem == 0) {
        norms[dr] = norms[sr];
        ids[dr] = ids[sr];
    }
}

static inline unsigned grid_for(int64_t n, int block) { return (unsigned)((n + block - 1) / block); }

static int launch_ingest(qk_ctx *ctx, const float *src, const int64_t *src_ids, int64_t n, int d, int nblk, float *vecs,
                         float *norms, int64_t *ids, const IngestMap &m) {
    if (n <= 0) return QK_OK;
    int64_t total = n * nblk * 4;
    hipLaunchKernelGGL(k_ingest_vecs, dim3(grid_for(total, 256)), dim3(256), 0, ctx->stream, src, n, d, nblk, (float4 *)vecs, m);
    hipLaunchKernelGGL(k_ingest_norms_ids, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, src, src_ids, n, d, norms, ids, m);
    QK_HIP(hipGetLastError());
    return QK_OK;
}

int qk_launch_ingest(qk_ctx *ctx, const float *src, const int64_t *src_ids, int64_t n, int d, int nblk, float *vecs,
                     float *norms, int64_t *ids, int64_t row0) {
    IngestMap m{row0, nullptr, nullptr, 0, 0, nullptr};
    return launch_ingest(ctx, src, src_ids, n, d, nblk, vecs, norms, ids, m);
}

int qk_launch_extract(qk_ctx *ctx, const float *vecs, int nblk, int d, int64_t row0, const int64_t *rows, int64_t n, float *dst) {
    if (n <= 0) return QK_OK;
    hipLaunchKernelGGL(k_extract, dim3(grid_for(n * d, 256)), dim3(256), 0, ctx->stream, vecs, nblk, d, row0, rows, n, dst);
    QK_HIP(hipGetLastError());
    return QK_OK;
}

int qk_launch_move_rows(qk_ctx *ctx, float *vecs, float *norms, int64_t *ids, int nblk, const int64_t *dst, const int64_t *src,
                        int64_t n) {
    if (n <= 0) return QK_OK;
    hipLaunchKernelGGL(k_move_rows, dim3(grid_for(n * nblk * 4, 256)), dim3(256), 0, ctx->stream, (float4 *)vecs, norms, ids,
                       nblk, dst, src, n);
    QK_HIP(hipGetLastError());
    return QK_OK;
}

// ------------------------------------------------------------------------------------------------
// arena management
// ------------------------------------------------------------------------------------------------
int qk_store_reserve_rows(qk_store *s, int64_t rows) {
    int64_t need = s->used_rows + rows;
    if (need <= s->cap_rows) return QK_OK;
    qk_ctx *c = s->ctx;
    // growth doubles: a re-allocation is a hipMalloc + copy + hipFree of the whole arena -- 1-2 s at 50M rows, measured inside a
    // maintenance call -- and an index that is modified at all keeps being modified (refinements re-add millions of rows at the tail)
    int64_t ncap = std::max<int64_t>(need, s->cap_rows * 2);
    ncap = qk_round_up64(std::max<int64_t>(ncap, 1024), 16);
    float *nv = nullptr, *nn = nullptr;
    int64_t *ni = nullptr;
    size_t vb = (size_t)ncap * s->dpad * sizeof(float);
    if (hipMalloc((void **)&nv, vb) != hipSuccess || hipMalloc((void **)&nn, (size_t)ncap * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&ni, (size_t)ncap * sizeof(int64_t)) != hipSuccess) {
        if (nv) hipFree(nv);
        if (nn) hipFree(nn);
        if (ni) hipFree(ni);
        QK_FAIL(QK_ERR_OOM, "store arena allocation failed for %lld rows x %d dims", (long long)ncap, s->dpad);
    }
    if (s->cap_rows > 0) {
        s->counters[0]++;
        s->counters[3] += s->used_rows;
    }
    if (s->used_rows > 0) {
        QK_HIP(hipMemcpyAsync(nv, s->vecs, (size_t)s->used_rows * s->dpad * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        QK_HIP(hipMemcpyAsync(nn, s->norms, (size_t)s->used_rows * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        QK_HIP(hipMemcpyAsync(ni, s->ids, (size_t)s->used_rows * sizeof(int64_t), hipMemcpyDeviceToDevice, c->stream));
    }
    // rows past used_rows are only ever read masked, but keep them finite and deterministic
    QK_HIP(hipMemsetAsync((char *)nv + (size_t)s->used_rows * s->dpad * sizeof(float), 0,
                          (size_t)(ncap - s->used_rows) * s->dpad * sizeof(float), c->stream));
    QK_HIP(hipMemsetAsync(nn + s->used_rows, 0, (size_t)(ncap - s->used_rows) * sizeof(float), c->stream));
    QK_HIP(hipMemsetAsync(ni + s->used_rows, 0xFF, (size_t)(ncap - s->used_rows) * sizeof(int64_t), c->stream));
    QK_HIP(hipStreamSynchronize(c->stream));
    if (s->vecs) hipFree(s->vecs);
    if (s->norms) hipFree(s->norms);
    if (s->ids) hipFree(s->ids);
    s->vecs = nv;
    s->norms = nn;
    s->ids = ni;
    s->cap_rows = ncap;
    return QK_OK;
}

int qk_store_sync_table(qk_store *s) {
    if (!s->table_dirty) return QK_OK;
    s->version++;
    s->counters[5]++;
    s->rowmajor_valid = false;
    qk_ctx *c = s->ctx;
    int64_t n = (int64_t)s->parts.size();
    if (n > s->table_cap) {
        QK_HIP(hipStreamSynchronize(c->stream));
        if (s->d_off) hipFree(s->d_off);
        if (s->d_size) hipFree(s->d_size);
        s->table_cap = std::max<int64_t>(n + n / 2, 64);
        QK_HIP(hipMalloc((void **)&s->d_off, (size_t)s->table_cap * sizeof(int64_t)));
        QK_HIP(hipMalloc((void **)&s->d_size, (size_t)s->table_cap * sizeof(int32_t)));
    }
    if (n > 0) {
        size_t bytes = (size_t)n * (sizeof(int64_t) + sizeof(int32_t));
        QK_TRY(qk_pinned_reserve(c, bytes));
        QK_HIP(hipStreamSynchronize(c->stream));  // pinned buffer may still feed an earlier async copy
        int64_t *ho = (int64_t *)c->pinned;
        int32_t *hs = (int32_t *)(c->pinned + (size_t)n * sizeof(int64_t));
        int64_t mx = 0, nonempty = 0;
        for (int64_t p = 0; p < n; p++) {
            const qk_part &pt = s->parts[p];
            ho[p] = pt.row_off;
            hs[p] = pt.present ? (int32_t)pt.size : -1;
            if (pt.present) mx = std::max(mx, pt.size);
            if (pt.present && pt.size > 0) nonempty++;
        }
        s->max_size = mx;
        s->n_nonempty = nonempty;
        QK_HIP(hipMemcpyAsync(s->d_off, ho, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
        QK_HIP(hipMemcpyAsync(s->d_size, hs, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        QK_HIP(hipStreamSynchronize(c->stream));
    } else {
        s->max_size = 0;
        s->n_nonempty = 0;
    }
    s->table_dirty = false;
    return QK_OK;
}

// id -> list number, built lazily from the id mirrors (the reference has no such index: find_id / remove_vectors scan)
void qk_store_ensure_index(qk_store *s) {
    if (s->index_valid) return;
    s->counters[6]++;
    // (first list wins for an id held twice: the order get_vector / remove meet them in)
    std::vector<const int64_t *> keys;
    std::vector<int64_t> lens;
    std::vector<int32_t> vals;
    for (size_t pi = 0; pi < s->parts.size(); pi++) {
        const qk_part &p = s->parts[pi];
        if (!p.present || p.size <= 0) continue;
        keys.push_back(p.ids.data());
        lens.push_back(p.size);
        vals.push_back((int32_t)pi);
    }
    const unsigned hw = std::thread::hardware_concurrency();
    s->id_to_list.build_from_segments(keys.data(), lens.data(), vals.data(), keys.size(), (int)std::min<unsigned>(16u, hw ? hw : 1u));
    s->index_valid = true;
}

static void note_ids(qk_store *s, const int64_t *ids, int64_t n) {
    for (int64_t i = 0; i < n; i++) {
        if (ids[i] > s->max_id_seen) s->max_id_seen = ids[i];
        if (ids[i] < s->min_id_seen) s->min_id_seen = ids[i];
    }
}

static int check_list(qk_store *s, int64_t list_no, const char *who) {
    if (list_no < 0 || list_no >= (int64_t)s->parts.size() || !s->parts[list_no].present)
        QK_FAIL(QK_ERR_NOT_FOUND, "List does not exist in %s (list %lld)", who, (long long)list_no);
    return QK_OK;
}


// ---- arena compaction: live extents copied into a new arena, abandoned ones dropped --------------------------------------
struct CompactMove {
    int64_t old_off, new_off;  // first row (multiple of 16)
    int64_t size;              // valid rows
};
// one workgroup per live partition
__global__ __launch_bounds__(256) void k_compact_copy(const CompactMove *__restrict__ mv, int dpad, const float4 *__restrict__ ovecs,
                                                      const float *__restrict__ onorms, const int64_t *__restrict__ oids,
                                                      float4 *__restrict__ nvecs, float *__restrict__ nnorms, int64_t *__restrict__ nids) {
    const CompactMove m = mv[blockIdx.x];
    const int64_t tiles = (m.size + 15) >> 4;
    const int64_t n4 = tiles * 16 * (dpad / 4);  // float4 per extent in use (tile-major: whole tiles)
    const float4 *src = ovecs + m.old_off * (dpad / 4);
    float4 *dst = nvecs + m.new_off * (dpad / 4);
    for (int64_t i = threadIdx.x; i < n4; i += blockDim.x) dst[i] = src[i];
    for (int64_t i = threadIdx.x; i < m.size; i += blockDim.x) {
        nnorms[m.new_off + i] = onorms[m.old_off + i];
        nids[m.new_off + i] = oids[m.old_off + i];
    }
}

// Rebuild the arena with every present partition (its capacity kept) laid out back to back, `last` (if >= 0) at the very
// end so that it can grow in place, and room for `extra_rows` more.  Called when a partition has to move and a quarter or
// more of the arena is abandoned extents: under skewed inserts the bump pointer would otherwise run through 10x the live
// data (every doubling of a hot partition leaves its old extent behind).
static int compact_arena_fresh(qk_store *s, int64_t extra_rows, int64_t last) {
    qk_ctx *c = s->ctx;
    std::vector<CompactMove> mv;
    std::vector<int64_t> order;
    for (size_t pi = 0; pi < s->parts.size(); pi++)
        if (s->parts[pi].present && (int64_t)pi != last) order.push_back((int64_t)pi);
    if (last >= 0 && last < (int64_t)s->parts.size() && s->parts[(size_t)last].present) order.push_back(last);
    int64_t live_cap = 0;
    for (int64_t pi : order) live_cap += s->parts[(size_t)pi].cap;
    // (never smaller than the arena it replaces: a compaction that shrank the arena to its live rows + 25 % was followed, a few
    //  refinements later, by a re-allocation -- the two alternated every ~10 maintenance calls of a 50M index)
    int64_t ncap = qk_round_up64(std::max<int64_t>(std::max<int64_t>(live_cap + extra_rows + live_cap / 4, s->cap_rows), 1024), 16);
    float *nv = nullptr, *nn = nullptr;
    int64_t *ni = nullptr;
    if (hipMalloc((void **)&nv, (size_t)ncap * s->dpad * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&nn, (size_t)ncap * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&ni, (size_t)ncap * sizeof(int64_t)) != hipSuccess) {
        if (nv) hipFree(nv);
        if (nn) hipFree(nn);
        if (ni) hipFree(ni);
        QK_FAIL(QK_ERR_OOM, "store arena compaction: allocation failed for %lld rows x %d dims", (long long)ncap, s->dpad);
    }
    QK_HIP(hipMemsetAsync(nv, 0, (size_t)ncap * s->dpad * sizeof(float), c->stream));
    QK_HIP(hipMemsetAsync(nn, 0, (size_t)ncap * sizeof(float), c->stream));
    QK_HIP(hipMemsetAsync(ni, 0xFF, (size_t)ncap * sizeof(int64_t), c->stream));
    s->counters[1]++;
    int64_t at = 0;
    std::vector<int64_t> new_off(order.size());
    for (size_t i = 0; i < order.size(); i++) {
        const qk_part &p = s->parts[(size_t)order[i]];
        new_off[i] = at;
        if (p.size > 0) mv.push_back({p.row_off, at, p.size});
        s->counters[3] += p.size;
        at += p.cap;
    }
    if (!mv.empty()) {
        const size_t bytes = mv.size() * sizeof(CompactMove);
        QK_TRY(qk_stage_reserve(c, bytes));
        QK_HIP(hipMemcpyAsync(c->stage, mv.data(), bytes, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_compact_copy, dim3((unsigned)mv.size()), dim3(256), 0, c->stream, (const CompactMove *)c->stage, s->dpad,
                           (const float4 *)s->vecs, s->norms, s->ids, (float4 *)nv, nn, ni);
        QK_HIP(hipGetLastError());
    }
    QK_HIP(hipStreamSynchronize(c->stream));  // (also keeps the pageable `mv` alive for the copy)
    if (s->vecs) hipFree(s->vecs);
    if (s->norms) hipFree(s->norms);
    if (s->ids) hipFree(s->ids);
    s->vecs = nv;
    s->norms = nn;
    s->ids = ni;
    s->cap_rows = ncap;
    for (size_t i = 0; i < order.size(); i++) s->parts[(size_t)order[i]].row_off = new_off[i];
    s->used_rows = at;
    s->dead_rows = 0;
    s->table_dirty = true;
    return QK_OK;
}

// ---- arena compaction IN PLACE ------------------------------------------------------------------------------------------------
// The form above replaces the arena: at 50M rows a hipMalloc / hipFree pair of 50 GB, 1.2 s inside one maintenance call
// (profiles/r06_dynamic_workload_hot_50M.json, the one call above 0.25 s after the first).  Here the live extents slide DOWN inside the
// arena they are in.  In ascending order of their offsets the shift of an extent (the abandoned rows below it) never decreases, so the
// extents that stay are a prefix and every later one moves to at or below where it was; its destination ends where the next extent's
// destination begins, which is at or below that extent's source -- a move never lands on rows that are still to be read, EXCEPT the
// rows of the extents moved with it.  So the moves go in batches through a bounce buffer (~1 GiB of rows): arena -> bounce, then
// bounce -> arena, one launch each, batches in stream order; an extent larger than the buffer goes in consecutive pieces (the same
// argument holds piece by piece).  Every row is read and written twice: 50M x 128 with a quarter abandoned is ~80 GB of traffic,
// ~25 ms.  No allocation, no free, the arena keeps its capacity (a compaction never shrank it anyway).
struct RowMove {
    int64_t from, to;  // first row (multiples of 16)
    int64_t rows;      // whole tiles
};
__global__ __launch_bounds__(256) void k_move_rows(const RowMove *__restrict__ mv, int dpad, const float4 *__restrict__ sv,
                                                   const float *__restrict__ sn, const int64_t *__restrict__ si, float4 *__restrict__ dv,
                                                   float *__restrict__ dn, int64_t *__restrict__ di) {
    const RowMove m = mv[blockIdx.y];
    const int64_t n4 = m.rows * (dpad / 4);
    const float4 *src = sv + m.from * (dpad / 4);
    float4 *dst = dv + m.to * (dpad / 4);
    const int64_t step = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += step) dst[i] = src[i];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m.rows; i += step) {
        dn[m.to + i] = sn[m.from + i];
        di[m.to + i] = si[m.from + i];
    }
}
// rows of a moved extent past its data: finite and deterministic, like a fresh arena's (they are only ever read masked)
struct RowFill {
    int64_t vrow0, vrows;  // vector rows to clear (from the end of the last tile in use to the end of the extent)
    int64_t nrow0, nrows;  // norm / id entries to clear (from the first unused row)
};
__global__ __launch_bounds__(256) void k_fill_rows(const RowFill *__restrict__ fl, int dpad, float4 *__restrict__ v, float *__restrict__ n,
                                                   int64_t *__restrict__ ids) {
    const RowFill f = fl[blockIdx.y];
    const int64_t n4 = f.vrows * (dpad / 4);
    float4 *dst = v + f.vrow0 * (dpad / 4);
    const int64_t step = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += step) dst[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < f.nrows; i += step) {
        n[f.nrow0 + i] = 0.0f;
        ids[f.nrow0 + i] = -1;
    }
}

static int compact_arena(qk_store *s, int64_t extra_rows, int64_t last) {
    qk_ctx *c = s->ctx;
    static const bool fresh = std::getenv("QK_COMPACT_FRESH") != nullptr;  // (A/B switch, read once: tests compare the two forms)
    if (fresh) return compact_arena_fresh(s, extra_rows, last);
    if (!s->bounce_v) {
        int64_t rows = std::max<int64_t>(16, (((int64_t)1 << 30) / ((int64_t)s->dpad * 4)) & ~(int64_t)15);
        // (test hook: a buffer of a few tiles makes a small store go through many batches and through extents cut in pieces)
        if (const char *e = std::getenv("QK_COMPACT_BOUNCE_ROWS")) rows = std::max<int64_t>(16, (int64_t)atoll(e) & ~(int64_t)15);
        float *bv = nullptr, *bn = nullptr;
        int64_t *bi = nullptr;
        if (hipMalloc((void **)&bv, (size_t)rows * s->dpad * sizeof(float)) != hipSuccess ||
            hipMalloc((void **)&bn, (size_t)rows * sizeof(float)) != hipSuccess ||
            hipMalloc((void **)&bi, (size_t)rows * sizeof(int64_t)) != hipSuccess) {
            if (bv) hipFree(bv);
            if (bn) hipFree(bn);
            if (bi) hipFree(bi);
            (void)hipGetLastError();
            return compact_arena_fresh(s, extra_rows, last);  // (no room for the bounce buffer: the replacing form says so if it fails too)
        }
        s->bounce_v = bv;
        s->bounce_n = bn;
        s->bounce_i = bi;
        s->bounce_rows = rows;
    }
    // `last` goes to the very end (it is about to grow: there it grows in place and abandons nothing): its rows wait in a buffer of
    // their own while the others slide over the place they were in.  (A `last` larger than the bounce buffer stays in line; the
    // caller then moves it like any extent that outgrew its place.)
    const int64_t B = s->bounce_rows;
    bool hold_last = last >= 0 && last < (int64_t)s->parts.size() && s->parts[(size_t)last].present && s->parts[(size_t)last].cap > 0 &&
                     ((s->parts[(size_t)last].size + 15) & ~(int64_t)15) <= B;
    const int64_t hold_rows = hold_last ? ((s->parts[(size_t)last].size + 15) & ~(int64_t)15) : 0;
    float *hv = nullptr, *hn = nullptr;
    int64_t *hi = nullptr;
    if (hold_rows > 0) {
        if (hipMalloc((void **)&hv, (size_t)hold_rows * s->dpad * sizeof(float)) != hipSuccess ||
            hipMalloc((void **)&hn, (size_t)hold_rows * sizeof(float)) != hipSuccess ||
            hipMalloc((void **)&hi, (size_t)hold_rows * sizeof(int64_t)) != hipSuccess) {
            if (hv) hipFree(hv);
            if (hn) hipFree(hn);
            if (hi) hipFree(hi);
            hv = hn = nullptr;
            hi = nullptr;
            (void)hipGetLastError();
            hold_last = false;
        }
    }
    struct HoldFree {  // (every exit below)
        float *v, *n;
        int64_t *i;
        ~HoldFree() {
            if (v) hipFree(v);
            if (n) hipFree(n);
            if (i) hipFree(i);
        }
    } hold_free{hv, hn, hi};
    std::vector<int64_t> order;
    int64_t live_cap = 0;
    for (size_t pi = 0; pi < s->parts.size(); pi++)
        if (s->parts[pi].present) {
            live_cap += s->parts[pi].cap;
            if (!(hold_last && (int64_t)pi == last)) order.push_back((int64_t)pi);
        }
    // the caller is about to take `extra_rows` at the end (with `last` there: its new capacity in place of the old one).  If the
    // arena cannot hold that even compacted, it has to be replaced anyway: the replacing form sizes the new one to what is live
    // (in place first and a re-allocation after it would double the arena: tests/test_store_dynamic_gpu.py bounds it)
    if (live_cap + extra_rows - (hold_last ? s->parts[(size_t)last].cap : 0) > s->cap_rows) return compact_arena_fresh(s, extra_rows, last);
    std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
        const qk_part &x = s->parts[(size_t)a], &y = s->parts[(size_t)b];
        return x.row_off != y.row_off ? x.row_off < y.row_off : a < b;  // (extents of capacity 0 share an offset with a neighbour)
    });
    std::vector<RowMove> out_mv, in_mv;   // arena -> bounce, bounce -> arena: entry i of both is piece i
    std::vector<size_t> batch_begin;      // first piece of every batch (+ the end)
    std::vector<RowFill> fills;
    std::vector<int64_t> new_off(order.size());
    int64_t at = 0, in_batch = 0, moved_rows = 0;
    batch_begin.push_back(0);
    for (size_t i = 0; i < order.size(); i++) {
        const qk_part &p = s->parts[(size_t)order[i]];
        new_off[i] = at;
        if (p.cap > 0 && at > p.row_off) return compact_arena_fresh(s, extra_rows, last);  // (extents that overlap: never, by construction)
        const int64_t rows = ((p.size + 15) >> 4) << 4;
        if (at != p.row_off && p.cap > 0) {
            for (int64_t o = 0; o < rows; o += B) {
                const int64_t r = std::min<int64_t>(B, rows - o);
                if (in_batch + r > B || out_mv.size() - batch_begin.back() == 65535) {
                    batch_begin.push_back(out_mv.size());
                    in_batch = 0;
                }
                out_mv.push_back({p.row_off + o, in_batch, r});
                in_mv.push_back({in_batch, at + o, r});
                in_batch += r;
            }
            moved_rows += p.size;
            if (p.cap > rows || p.cap > p.size) fills.push_back({at + rows, p.cap - rows, at + p.size, p.cap - p.size});
        }
        at += p.cap;
    }
    batch_begin.push_back(out_mv.size());
    const size_t np = out_mv.size();  // pieces of the batches; the two moves of `last` ride behind them in the same arrays
    int64_t last_new_off = -1;
    if (hold_last) {
        const qk_part &p = s->parts[(size_t)last];
        last_new_off = at;
        out_mv.push_back({p.row_off, 0, hold_rows});
        in_mv.push_back({0, at, hold_rows});
        moved_rows += p.size;
        if (p.cap > hold_rows || p.cap > p.size) fills.push_back({at + hold_rows, p.cap - hold_rows, at + p.size, p.cap - p.size});
        at += p.cap;
    }
    s->counters[1]++;
    s->counters[3] += moved_rows;
    if (!out_mv.empty()) {
        const size_t mv_bytes = out_mv.size() * sizeof(RowMove), fl_bytes = fills.size() * sizeof(RowFill);
        QK_TRY(qk_stage_reserve(c, 2 * mv_bytes + fl_bytes + 64));
        char *dst = (char *)c->stage;
        QK_HIP(hipMemcpyAsync(dst, out_mv.data(), mv_bytes, hipMemcpyHostToDevice, c->stream));
        QK_HIP(hipMemcpyAsync(dst + mv_bytes, in_mv.data(), mv_bytes, hipMemcpyHostToDevice, c->stream));
        if (fl_bytes) QK_HIP(hipMemcpyAsync(dst + 2 * mv_bytes, fills.data(), fl_bytes, hipMemcpyHostToDevice, c->stream));
        const RowMove *d_out = (const RowMove *)dst, *d_in = (const RowMove *)(dst + mv_bytes);
        const RowFill *d_fl = (const RowFill *)(dst + 2 * mv_bytes);
        if (hold_last && hold_rows > 0)  // out of the way first ...
            hipLaunchKernelGGL(k_move_rows, dim3(64, 1), dim3(256), 0, c->stream, d_out + np, s->dpad, (const float4 *)s->vecs,
                               (const float *)s->norms, (const int64_t *)s->ids, (float4 *)hv, hn, hi);
        for (size_t b = 0; b + 1 < batch_begin.size(); b++) {
            const size_t p0 = batch_begin[b], cnt = batch_begin[b + 1] - p0;
            if (cnt == 0) continue;
            // (a batch of many small extents: few workgroups each; a batch that is one piece of a huge extent: many)
            const unsigned gx = (unsigned)std::max<size_t>(1, std::min<size_t>(256, 4096 / cnt));
            hipLaunchKernelGGL(k_move_rows, dim3(gx, (unsigned)cnt), dim3(256), 0, c->stream, d_out + p0, s->dpad, (const float4 *)s->vecs,
                               (const float *)s->norms, (const int64_t *)s->ids, (float4 *)s->bounce_v, s->bounce_n, s->bounce_i);
            hipLaunchKernelGGL(k_move_rows, dim3(gx, (unsigned)cnt), dim3(256), 0, c->stream, d_in + p0, s->dpad, (const float4 *)s->bounce_v,
                               (const float *)s->bounce_n, (const int64_t *)s->bounce_i, (float4 *)s->vecs, s->norms, s->ids);
        }
        if (hold_last && hold_rows > 0)  // ... and behind everybody else at the end
            hipLaunchKernelGGL(k_move_rows, dim3(64, 1), dim3(256), 0, c->stream, d_in + np, s->dpad, (const float4 *)hv, (const float *)hn,
                               (const int64_t *)hi, (float4 *)s->vecs, s->norms, s->ids);
        for (size_t f0 = 0; f0 < fills.size(); f0 += 65535) {
            const size_t cnt = std::min<size_t>(65535, fills.size() - f0);
            hipLaunchKernelGGL(k_fill_rows, dim3(8, (unsigned)cnt), dim3(256), 0, c->stream, d_fl + f0, s->dpad, (float4 *)s->vecs, s->norms, s->ids);
        }
        QK_HIP(hipGetLastError());
    }
    // what the bump pointer hands out next: rows the moved extents left behind
    if (at < s->used_rows) {
        QK_HIP(hipMemsetAsync(s->vecs + at * s->dpad, 0, (size_t)(s->used_rows - at) * s->dpad * sizeof(float), c->stream));
        QK_HIP(hipMemsetAsync(s->norms + at, 0, (size_t)(s->used_rows - at) * sizeof(float), c->stream));
        QK_HIP(hipMemsetAsync(s->ids + at, 0xFF, (size_t)(s->used_rows - at) * sizeof(int64_t), c->stream));
    }
    QK_HIP(hipStreamSynchronize(c->stream));  // (also keeps the pageable piece lists alive for their copies)
    for (size_t i = 0; i < order.size(); i++) s->parts[(size_t)order[i]].row_off = new_off[i];
    if (hold_last) s->parts[(size_t)last].row_off = last_new_off;
    s->used_rows = at;
    s->dead_rows = 0;
    s->table_dirty = true;
    return QK_OK;
}

// make sure partition `p` can take `extra` more rows; relocates the extent (amortised doubling, like
// IndexPartition::ensure_capacity index_partition.cpp:247-255) when it does not fit
static int ensure_part_capacity(qk_store *s, qk_part &p, int64_t extra) {
    int64_t need = p.size + extra;
    if (need <= p.cap) return QK_OK;
    qk_ctx *c = s->ctx;
    int64_t ncap = qk_round_up64(std::max<int64_t>(need, p.cap * 2), 16);
    // the arena would have to grow although a quarter of it is abandoned extents: compact instead (p goes last)
    if (s->used_rows + ncap > s->cap_rows && s->dead_rows * 4 >= s->cap_rows && s->dead_rows >= 1024)
        QK_TRY(compact_arena(s, ncap, (int64_t)(&p - s->parts.data())));
    // extent at the very end of the arena can grow in place
    if (p.cap > 0 && p.row_off + p.cap == s->used_rows) {
        QK_TRY(qk_store_reserve_rows(s, ncap - p.cap));
        s->used_rows += ncap - p.cap;
        p.cap = ncap;
        return QK_OK;
    }
    QK_TRY(qk_store_reserve_rows(s, ncap));
    int64_t nrow = s->used_rows;
    s->used_rows += ncap;
    if (p.size > 0) {
        s->counters[2]++;
        s->counters[3] += p.size;
        int64_t tiles = (p.size + 15) / 16;
        QK_HIP(hipMemcpyAsync(s->vecs + nrow * s->dpad, s->vecs + p.row_off * s->dpad, (size_t)tiles * 16 * s->dpad * sizeof(float),
                              hipMemcpyDeviceToDevice, c->stream));
        QK_HIP(hipMemcpyAsync(s->norms + nrow, s->norms + p.row_off, (size_t)p.size * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        QK_HIP(hipMemcpyAsync(s->ids + nrow, s->ids + p.row_off, (size_t)p.size * sizeof(int64_t), hipMemcpyDeviceToDevice, c->stream));
    }
    s->dead_rows += p.cap;
    p.row_off = nrow;
    p.cap = ncap;
    s->table_dirty = true;
    return QK_OK;
}

// copy `bytes` from caller memory (host or device) to a device pointer usable by kernels; returns the device pointer
static int to_device(qk_ctx *c, const void *src, size_t bytes, int mem, char *stage_dst, const void **out) {
    if (mem == QK_MEM_DEVICE) {
        *out = src;
        return QK_OK;
    }
    QK_HIP(hipMemcpyAsync(stage_dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    *out = stage_dst;
    return QK_OK;
}

extern "C" {

int qk_store_create(qk_ctx *ctx, int d, qk_store **out) {
    if (!ctx || !out) QK_FAIL(QK_ERR_INVALID, "qk_store_create: null argument");
    if (d <= 0) QK_FAIL(QK_ERR_INVALID, "qk_store_create: d must be positive (got %d)", d);
    static std::atomic<uint64_t> next_uid{1};
    qk_store *s = new qk_store();
    s->ctx = ctx;
    s->uid = next_uid.fetch_add(1);
    s->d = d;
    s->dpad = qk_round_up(d, 16);
    s->nblk = s->dpad / 16;
    *out = s;
    return QK_OK;
}

// Row-major copy of rows [row_off, row_off + nrows) (one list: the centroids of a parent / flat index), made on the STORE's
// context stream and complete on return (a host synchronisation, once per change of the store): any context may read it afterwards.
// Meant for a PARENT's centroids (2 MB at 4096 rows, 33 MB at 65536): a list beyond QK_ROWMAJOR_MAX_BYTES gets no copy (*out stays
// null and the caller gathers from the tile-major arena) -- a flat index of millions of rows must not silently double its
// footprint.  One (row_off, nrows) entry is cached; it is replaced, under the store, by the next different request: callers on
// several host threads must serialise searches that go through the prefiltered dense path of the SAME store.
extern "C++" int qk_store_rowmajor(qk_store *s, int64_t row_off, int nrows, const float **out) {
    *out = nullptr;
    if (nrows <= 0 || s->table_dirty) return QK_OK;
    if ((int64_t)nrows * s->d * (int64_t)sizeof(float) > QK_ROWMAJOR_MAX_BYTES) return QK_OK;
    if (s->rowmajor_valid && s->rowmajor_row_off == row_off && s->rowmajor_rows == nrows) {
        *out = s->rowmajor;
        return QK_OK;
    }
    qk_ctx *c = s->ctx;
    const int64_t need = (int64_t)nrows * s->d;
    if (need > s->rowmajor_cap) {
        QK_HIP(hipStreamSynchronize(c->stream));
        if (s->rowmajor) QK_HIP(hipFree(s->rowmajor));
        s->rowmajor = nullptr;
        s->rowmajor_cap = 0;
        if (hipMalloc((void **)&s->rowmajor, (size_t)(need + need / 4) * sizeof(float)) != hipSuccess) {
            (void)hipGetLastError();
            return QK_OK;  // no room for the copy: the caller gathers from the arena
        }
        s->rowmajor_cap = need + need / 4;
    }
    QK_TRY(qk_launch_extract(c, s->vecs, s->nblk, s->d, row_off, nullptr, nrows, s->rowmajor));
    QK_HIP(hipStreamSynchronize(c->stream));
    s->counters[4]++;
    s->rowmajor_valid = true;
    s->rowmajor_row_off = row_off;
    s->rowmajor_rows = nrows;
    *out = s->rowmajor;
    return QK_OK;
}

int qk_store_destroy(qk_store *s) {
    if (!s) return QK_OK;
    hipSetDevice(s->ctx->device);
    hipStreamSynchronize(s->ctx->stream);
    if (s->vecs) hipFree(s->vecs);
    if (s->norms) hipFree(s->norms);
    if (s->ids) hipFree(s->ids);
    if (s->d_off) hipFree(s->d_off);
    if (s->d_size) hipFree(s->d_size);
    if (s->rowmajor) hipFree(s->rowmajor);
    if (s->bounce_v) hipFree(s->bounce_v);
    if (s->bounce_n) hipFree(s->bounce_n);
    if (s->bounce_i) hipFree(s->bounce_i);
    delete s;
    return QK_OK;
}

int qk_store_reset(qk_store *s) {
    if (!s) QK_FAIL(QK_ERR_INVALID, "qk_store_reset: null store");
    s->parts.clear();
    s->nlist = 0;
    s->ntotal = 0;
    s->used_rows = 0;
    s->dead_rows = 0;
    s->table_dirty = true;
    s->id_to_list.clear();
    s->index_valid = false;
    return QK_OK;
}

int qk_store_add_list(qk_store *s, int64_t list_no) {
    if (!s) QK_FAIL(QK_ERR_INVALID, "qk_store_add_list: null store");
    if (list_no < 0) QK_FAIL(QK_ERR_INVALID, "qk_store_add_list: negative list number");
    if (list_no < (int64_t)s->parts.size() && s->parts[list_no].present)
        QK_FAIL(QK_ERR_INVALID, "List already exists in add_list (list %lld)", (long long)list_no);
    if (list_no >= (int64_t)s->parts.size()) s->parts.resize(list_no + 1);
    qk_part &p = s->parts[list_no];
    p = qk_part();
    p.present = true;
    s->nlist++;
    s->table_dirty = true;
    return QK_OK;
}

int qk_store_remove_list(qk_store *s, int64_t list_no) { return qk_store_remove_list_ex(s, list_no, false); }

}  // extern "C"

// keep_index: the id -> list entries of the list's ids stay as they are -- for a caller that re-adds the SAME ids right away
// (qk_store_refine_lists: its rows only change lists), which then overwrites them in place instead of erasing and inserting
int qk_store_remove_list_ex(qk_store *s, int64_t list_no, bool keep_index) {
    if (!s) QK_FAIL(QK_ERR_INVALID, "qk_store_remove_list: null store");
    if (list_no < 0 || list_no >= (int64_t)s->parts.size() || !s->parts[list_no].present) return QK_OK;  // "Already doesn't exist"
    qk_part &p = s->parts[list_no];
    if (s->index_valid && !keep_index)
        for (int64_t id : p.ids) {
            // only the entries that still name THIS list: while refine_lists replaces its lists one after the other, an id of
            // this list's old contents may already live in (and be indexed under) a list replaced before it
            s->id_to_list.erase_if(id, (int32_t)list_no);
        }
    s->ntotal -= p.size;
    s->dead_rows += p.cap;
    p = qk_part();
    s->nlist--;
    s->table_dirty = true;
    return QK_OK;
}

extern "C" {

int qk_store_add_entries(qk_store *s, int64_t list_no, int64_t n, const int64_t *ids, const float *vecs, int mem) {
    if (!s) QK_FAIL(QK_ERR_INVALID, "qk_store_add_entries: null store");
    if (n == 0) return QK_OK;
    if (n < 0 || !ids || !vecs) QK_FAIL(QK_ERR_INVALID, "qk_store_add_entries: bad arguments");
    QK_TRY(check_list(s, list_no, "add_entries"));
    qk_ctx *c = s->ctx;
    QK_HIP(hipSetDevice(c->device));
    qk_part &p = s->parts[list_no];
    QK_TRY(ensure_part_capacity(s, p, n));
    size_t vb = (size_t)n * s->d * sizeof(float), ib = (size_t)n * sizeof(int64_t);
    const void *dv = vecs, *di = ids;
    size_t old = p.ids.size();
    p.ids.resize(old + n);
    if (mem == QK_MEM_HOST) {
        QK_TRY(qk_stage_reserve(c, vb + ib + 256));
        QK_TRY(to_device(c, vecs, vb, mem, c->stage, &dv));
        QK_TRY(to_device(c, ids, ib, mem, c->stage + ((vb + 255) & ~(size_t)255), &di));
        memcpy(p.ids.data() + old, ids, ib);
    } else {
        QK_HIP(hipMemcpyAsync(p.ids.data() + old, ids, ib, hipMemcpyDeviceToHost, c->stream));
    }
    QK_TRY(qk_launch_ingest(c, (const float *)dv, (const int64_t *)di, n, s->d, s->nblk, s->vecs, s->norms, s->ids, p.row_off + p.size));
    QK_HIP(hipStreamSynchronize(c->stream));  // staging buffer / caller memory reusable on return
    note_ids(s, p.ids.data() + old, n);
    if (s->index_valid)
        for (int64_t i = 0; i < n; i++) s->id_to_list.set(p.ids[old + i], (int32_t)list_no);
    p.size += n;
    s->ntotal += n;
    s->table_dirty = true;
    return QK_OK;
}

int qk_store_build_csr(qk_store *s, int64_t nlist, const int64_t *offsets, const int64_t *ids, const float *vecs, int mem) {
    if (!s || !offsets || nlist < 0) QK_FAIL(QK_ERR_INVALID, "qk_store_build_csr: bad arguments");
    qk_ctx *c = s->ctx;
    QK_HIP(hipSetDevice(c->device));
    const int64_t total = offsets[nlist];
    if (total > 0 && (!ids || !vecs)) QK_FAIL(QK_ERR_INVALID, "qk_store_build_csr: null data");
    // host mirror of ids
    std::vector<int64_t> host_ids;
    const int64_t *hid = ids;
    if (mem == QK_MEM_DEVICE && total > 0) {
        host_ids.resize(total);
        // the caller's device arrays were produced on the context's stream (possibly a non-blocking one, which the
        // synchronous copies below would not wait for)
        QK_HIP(hipStreamSynchronize(c->stream));
        QK_HIP(hipMemcpy(host_ids.data(), ids, (size_t)total * sizeof(int64_t), hipMemcpyDeviceToHost));
        hid = host_ids.data();
    }
    qk_csr_build b;
    QK_TRY(qk_store_csr_begin(s, nlist, offsets, hid, 1, 0, &b));
    int rc = QK_OK;
    if (total > 0) {
        const int64_t CH = mem == QK_MEM_HOST ? std::max<int64_t>(1, (int64_t)(128u << 20) / ((int64_t)s->d * 4 + 8)) : total;
        if (mem == QK_MEM_HOST) rc = qk_stage_reserve(c, (size_t)CH * ((size_t)s->d * 4 + 8) + 512);
        for (int64_t i0 = 0; rc == QK_OK && i0 < total; i0 += CH) {
            int64_t n = std::min(CH, total - i0);
            const float *dv = vecs + i0 * s->d;
            const int64_t *di = ids + i0;
            if (mem == QK_MEM_HOST) {
                size_t vb = (size_t)n * s->d * sizeof(float);
                char *si = c->stage + ((vb + 255) & ~(size_t)255);
                if (hipMemcpyAsync(c->stage, dv, vb, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
                    hipMemcpyAsync(si, di, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, c->stream) != hipSuccess) {
                    qk_set_error("qk_store_build_csr: H2D copy failed");
                    rc = QK_ERR_HIP;
                    break;
                }
                dv = (const float *)c->stage;
                di = (const int64_t *)si;
            }
            rc = qk_store_csr_chunk(s, &b, dv, di, n, i0);
            if (rc == QK_OK && mem == QK_MEM_HOST && hipStreamSynchronize(c->stream) != hipSuccess) rc = QK_ERR_HIP;
        }
    }
    return qk_store_csr_end(s, &b, rc);
}

}  // extern "C"

int qk_store_csr_begin(qk_store *s, int64_t nlist, const int64_t *offsets, const int64_t *hid, int mod, int rem, qk_csr_build *b) {
    qk_ctx *c = s->ctx;
    QK_HIP(hipSetDevice(c->device));
    QK_TRY(qk_store_reset(s));
    b->nlist = nlist;
    b->total = offsets[nlist];
    s->parts.resize(nlist);
    int64_t rows = 0, owned_total = 0, owned_lists = 0;
    std::vector<int64_t> part_row(nlist);
    for (int64_t p = 0; p < nlist; p++) {
        int64_t sz = offsets[p + 1] - offsets[p];
        if (sz < 0) QK_FAIL(QK_ERR_INVALID, "qk_store_build_csr: offsets not monotone at %lld", (long long)p);
        qk_part &pt = s->parts[p];
        if (mod > 1 && (int)(p % mod) != rem) {  // another member's list
            pt = qk_part();
            part_row[p] = -1;
            continue;
        }
        pt.present = true;
        pt.size = sz;
        pt.cap = qk_round_up64(sz, 16);
        pt.row_off = rows;
        part_row[p] = rows;
        rows += pt.cap;
        owned_total += sz;
        owned_lists++;
    }
    s->nlist = owned_lists;
    s->ntotal = owned_total;
    QK_TRY(qk_store_reserve_rows(s, rows));
    s->used_rows = rows;
    for (int64_t p = 0; p < nlist; p++)
        if (s->parts[p].present) {
            s->parts[p].ids.assign(hid + offsets[p], hid + offsets[p + 1]);
            note_ids(s, hid + offsets[p], offsets[p + 1] - offsets[p]);
        }
    if (b->total > 0) {
        // CSR tables on the device
        QK_HIP(hipMalloc((void **)&b->d_offsets, (size_t)(nlist + 1) * sizeof(int64_t)));
        QK_HIP(hipMalloc((void **)&b->d_part_row, (size_t)nlist * sizeof(int64_t)));
        QK_HIP(hipMemcpy(b->d_offsets, offsets, (size_t)(nlist + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
        QK_HIP(hipMemcpy(b->d_part_row, part_row.data(), (size_t)nlist * sizeof(int64_t), hipMemcpyHostToDevice));
    }
    return QK_OK;
}

int qk_store_csr_chunk(qk_store *s, const qk_csr_build *b, const float *dv, const int64_t *di, int64_t n, int64_t i0) {
    qk_ctx *c = s->ctx;
    QK_HIP(hipSetDevice(c->device));
    IngestMap m{0, b->d_offsets, b->d_part_row, b->nlist, i0, nullptr};
    return launch_ingest(c, dv, di, n, s->d, s->nblk, s->vecs, s->norms, s->ids, m);
}

int qk_store_csr_end(qk_store *s, qk_csr_build *b, int rc) {
    qk_ctx *c = s->ctx;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    if (b->d_offsets) hipFree(b->d_offsets);
    if (b->d_part_row) hipFree(b->d_part_row);
    b->d_offsets = b->d_part_row = nullptr;
    QK_TRY(rc);
    s->table_dirty = true;
    QK_TRY(qk_store_sync_table(s));
    // the id -> list index is built HERE, with the bulk build (16 host threads: ~0.3 s at 50M ids), not by the first remove / get_vector
    // that needs it: on a serving index that first call was a 0.3-1.8 s stall (qk_store_counters: id_index_rebuilds stays 0 after this)
    s->id_to_list.clear();
    s->index_valid = false;
    qk_store_ensure_index(s);
    s->counters[6]--;  // (part of the build, not a rebuild)
    return QK_OK;
}

extern "C" {

int qk_store_remove_ids(qk_store *s, int64_t n, const int64_t *ids_host, int64_t *n_removed) {
    if (!s) QK_FAIL(QK_ERR_INVALID, "qk_store_remove_ids: null store");
    if (n_removed) *n_removed = 0;
    if (n <= 0) return QK_OK;
    if (!ids_host) QK_FAIL(QK_ERR_INVALID, "qk_store_remove_ids: null ids");
    qk_ctx *c = s->ctx;
    QK_HIP(hipSetDevice(c->device));
    // membership of the ids to remove: one bit per id of the range the store has ever held when that range is small (ids are
    // row numbers of a corpus in practice: 50M ids = 6 MB, cache-resident) -- the sweep below asks once per row of every
    // touched list, and a hash probe per row (5M rows for 500k ids spread over the index) was 80 % of the call; else a hash set.
    // The bitmap lives in the store, all zero between calls (the bits set here are cleared before returning).
    const int64_t id_lo = s->min_id_seen, id_hi = s->max_id_seen;
    const bool use_bits = id_hi >= id_lo && (id_hi - id_lo) < ((int64_t)1 << 30);
    // everything that can fail for lack of memory happens BEFORE the host mirror is touched: at most one row move per removed id
    QK_TRY(qk_stage_reserve(c, 2 * (((size_t)n * sizeof(int64_t) + 255) & ~(size_t)255) + 256));
    // the bits set below are cleared on EVERY way out of this function (a bit left behind would make the next call delete a
    // row it was never asked to)
    struct BitsGuard {
        qk_store *s;
        const int64_t *ids;
        int64_t n, lo, hi;
        bool on;
        ~BitsGuard() {
            if (!on) return;
            for (int64_t i = 0; i < n; i++) {
                const int64_t v = ids[i];
                if (v >= lo && v <= hi) s->kill_bits[(size_t)((v - lo) >> 6)] = 0;
            }
        }
    } bits_guard{s, ids_host, n, id_lo, id_hi, use_bits};
    QkIdMap kill;  // (as a set)
    if (use_bits) {
        const size_t words = (size_t)((id_hi - id_lo) >> 6) + 1;
        if (s->kill_bits.size() < words) s->kill_bits.resize(words, 0);
        for (int64_t i = 0; i < n; i++) {
            const int64_t v = ids_host[i];
            if (v >= id_lo && v <= id_hi) s->kill_bits[(size_t)((v - id_lo) >> 6)] |= 1ull << ((v - id_lo) & 63);
        }
    } else {
        kill.reserve((size_t)n);
        for (int64_t i = 0; i < n; i++) {
            if (i + 16 < n) kill.prefetch(ids_host[i + 16]);
            kill.set(ids_host[i], 0);
        }
    }
    const uint64_t *bits = s->kill_bits.data();
    auto is_kill = [&](int64_t v) -> bool {
        if (use_bits) return v >= id_lo && v <= id_hi && ((bits[(size_t)((v - id_lo) >> 6)] >> ((v - id_lo) & 63)) & 1ull);
        return kill.find(v) >= 0;
    };
    std::vector<int64_t> mv_dst, mv_src;
    int64_t removed = 0;
    // the id -> list index tells which lists hold something to remove; only those are swept (the reference sweeps
    // every partition, dynamic_inverted_list.cpp:137-149 -- same result, O(touched lists) instead of O(N))
    qk_store_ensure_index(s);
    std::vector<char> touched_list(s->parts.size(), 0);
    // (the index entry of an id goes HERE, in the one pass over the request whose table accesses can be requested ahead: the list
    //  that holds the id is swept below and every row with a requested id leaves it, so the entry would be erased there anyway --
    //  by a random access per removed row in the middle of a sequential sweep)
    for (int64_t i = 0; i < n; i++) {
        if (i + 16 < n) s->id_to_list.prefetch(ids_host[i + 16]);
        const int32_t holder = s->id_to_list.take(ids_host[i]);
        if (holder >= 0) touched_list[(size_t)holder] = 1;
    }
    // The touched lists are independent of one another: swept by several threads when there is enough to sweep (12.5k ids spread
    // over a 50M-row index touch 12.5k lists = 31M ids to test: 54 ms on one thread).  Every thread takes a contiguous run of
    // the touched lists and keeps its own move list; the runs are joined in list order, so the moves are those of one thread.
    std::vector<size_t> work;
    int64_t rows_to_sweep = 0;
    for (size_t pi = 0; pi < s->parts.size(); pi++) {
        const qk_part &p = s->parts[pi];
        if (!touched_list[pi] || !p.present || p.size == 0) continue;
        work.push_back(pi);
        rows_to_sweep += p.size;
    }
    struct SweepOut {
        std::vector<int64_t> dst, src;
        int64_t removed = 0;
    };
    auto sweep = [&](size_t w0, size_t w1, SweepOut &out) {
        std::vector<int64_t> cur;
        for (size_t w = w0; w < w1; w++) {
            qk_part &p = s->parts[work[w]];
            // scan, swap-with-last on a hit, re-examine the swapped-in row (IndexPartition::remove, index_partition.cpp:79-102)
            bool touched = false;
            int64_t sz = p.size;
            for (int64_t i = 0; i < sz;) {
                if (is_kill(p.ids[i])) {
                    if (!touched) {
                        cur.resize(p.size);
                        for (int64_t t = 0; t < p.size; t++) cur[t] = t;
                        touched = true;
                    }
                    if (i != sz - 1) {
                        p.ids[i] = p.ids[sz - 1];
                        cur[i] = cur[sz - 1];
                    }
                    sz--;
                } else {
                    i++;
                }
            }
            if (!touched) continue;
            for (int64_t i = 0; i < sz; i++)
                if (cur[i] != i) {
                    out.dst.push_back(p.row_off + i);
                    out.src.push_back(p.row_off + cur[i]);
                }
            out.removed += p.size - sz;
            p.size = sz;
            p.ids.resize(sz);
        }
    };
    {
        const unsigned hw = std::thread::hardware_concurrency();
        const int T = (int)std::min<int64_t>(std::min<unsigned>(16u, hw ? hw : 1u), std::min<int64_t>((int64_t)work.size(), rows_to_sweep >> 20));
        if (T <= 1) {
            SweepOut o;
            sweep(0, work.size(), o);
            mv_dst.swap(o.dst);
            mv_src.swap(o.src);
            removed = o.removed;
        } else {
            std::vector<SweepOut> outs((size_t)T);
            qk_run_shares(T, [&](int t) { sweep(work.size() * (size_t)t / (size_t)T, work.size() * (size_t)(t + 1) / (size_t)T, outs[(size_t)t]); });
            for (auto &o : outs) {
                mv_dst.insert(mv_dst.end(), o.dst.begin(), o.dst.end());
                mv_src.insert(mv_src.end(), o.src.begin(), o.src.end());
                removed += o.removed;
            }
        }
    }
    // the host mirror is final here: counters first, so that they agree with it whatever the device step below returns
    s->ntotal -= removed;
    if (removed) s->table_dirty = true;
    if (n_removed) *n_removed = removed;
    // one row move per removed row -- which is at most one per id asked for, UNLESS a list holds an id several times (the store
    // does not enforce unique ids: add_batch / add_entries append whatever they are given).  The staging buffer was reserved for
    // n moves before the mirror was touched; a longer move list goes through it in chunks of n (the moves of a sweep are
    // independent: every source row lies beyond its list's new size, every destination row inside it).
    const size_t cap = (size_t)n, slot = ((size_t)n * sizeof(int64_t) + 255) & ~(size_t)255;
    for (size_t off = 0; off < mv_dst.size(); off += cap) {
        const size_t m = std::min(cap, mv_dst.size() - off), b = m * sizeof(int64_t);
        int64_t *dd = (int64_t *)c->stage, *ds = (int64_t *)(c->stage + slot);
        QK_HIP(hipMemcpyAsync(dd, mv_dst.data() + off, b, hipMemcpyHostToDevice, c->stream));
        QK_HIP(hipMemcpyAsync(ds, mv_src.data() + off, b, hipMemcpyHostToDevice, c->stream));
        QK_TRY(qk_launch_move_rows(c, s->vecs, s->norms, s->ids, s->nblk, dd, ds, (int64_t)m));
    }
    if (!mv_dst.empty()) QK_HIP(hipStreamSynchronize(c->stream));
    return QK_OK;
}

int qk_store_list_size(qk_store *s, int64_t list_no, int64_t *out) {
    if (!s || !out) QK_FAIL(QK_ERR_INVALID, "qk_store_list_size: null argument");
    QK_TRY(check_list(s, list_no, "list_size"));
    *out = s->parts[list_no].size;
    return QK_OK;
}

int qk_store_list_sizes(qk_store *s, const int64_t *list_nos, int64_t n, int64_t *out) {
    if (!s || (n > 0 && (!list_nos || !out))) QK_FAIL(QK_ERR_INVALID, "qk_store_list_sizes: null argument");
    for (int64_t i = 0; i < n; i++) {
        QK_TRY(check_list(s, list_nos[i], "list_sizes"));
        out[i] = s->parts[list_nos[i]].size;
    }
    return QK_OK;
}

int64_t qk_store_ntotal(qk_store *s) { return s ? s->ntotal : 0; }
int64_t qk_store_nlist(qk_store *s) { return s ? s->nlist : 0; }
int qk_store_d(qk_store *s) { return s ? s->d : 0; }

int qk_store_list_ids(qk_store *s, int64_t *out_host, int64_t *n) {
    if (!s || !n) QK_FAIL(QK_ERR_INVALID, "qk_store_list_ids: null argument");
    int64_t cnt = 0;
    for (size_t p = 0; p < s->parts.size(); p++)
        if (s->parts[p].present) {
            if (out_host) out_host[cnt] = (int64_t)p;
            cnt++;
        }
    *n = cnt;
    return QK_OK;
}

int qk_store_get_list(qk_store *s, int64_t list_no, float *vecs_out, int64_t *ids_out, int mem) {
    if (!s) QK_FAIL(QK_ERR_INVALID, "qk_store_get_list: null store");
    QK_TRY(check_list(s, list_no, "get_codes"));
    qk_ctx *c = s->ctx;
    QK_HIP(hipSetDevice(c->device));
    qk_part &p = s->parts[list_no];
    if (p.size == 0) return QK_OK;
    if (ids_out) {
        if (mem == QK_MEM_HOST)
            memcpy(ids_out, p.ids.data(), (size_t)p.size * sizeof(int64_t));
        else
            QK_HIP(hipMemcpyAsync(ids_out, s->ids + p.row_off, (size_t)p.size * sizeof(int64_t), hipMemcpyDeviceToDevice, c->stream));
    }
    if (vecs_out) {
        size_t vb = (size_t)p.size * s->d * sizeof(float);
        if (mem == QK_MEM_HOST) {
            QK_TRY(qk_stage_reserve(c, vb));
            QK_TRY(qk_launch_extract(c, s->vecs, s->nblk, s->d, p.row_off, nullptr, p.size, (float *)c->stage));
            QK_HIP(hipMemcpyAsync(vecs_out, c->stage, vb, hipMemcpyDeviceToHost, c->stream));
        } else {
            QK_TRY(qk_launch_extract(c, s->vecs, s->nblk, s->d, p.row_off, nullptr, p.size, vecs_out));
        }
    }
    if (mem == QK_MEM_HOST) QK_HIP(hipStreamSynchronize(c->stream));
    return QK_OK;
}

int qk_store_get_vectors(qk_store *s, const int64_t *ids_host, int64_t n, float *vecs_out_host, int *found) {
    if (!s || (n > 0 && (!ids_host || !vecs_out_host || !found))) QK_FAIL(QK_ERR_INVALID, "qk_store_get_vectors: null argument");
    if (n <= 0) return QK_OK;
    qk_ctx *c = s->ctx;
    QK_HIP(hipSetDevice(c->device));
    qk_store_ensure_index(s);
    // arena row of every id: the list from the id index, the place inside the list from a position table built ONCE per list the
    // request touches more than a few times (get_vector scans the list per id: 20000 centroids of a parent asked for one by one
    // were 20000 scans of a 20000-id list)
    std::vector<int64_t> rows((size_t)n, -1);
    std::vector<int32_t> holder((size_t)n);
    std::vector<int64_t> asked(s->parts.size(), 0);
    for (int64_t i = 0; i < n; i++) {
        holder[(size_t)i] = s->id_to_list.find(ids_host[i]);
        if (holder[(size_t)i] >= 0) asked[(size_t)holder[(size_t)i]]++;
    }
    std::unordered_map<int32_t, QkIdMap> pos;  // (only for the lists the request touches often: a store has tens of thousands)
    int64_t n_found = 0;
    for (int64_t i = 0; i < n; i++) {
        found[i] = 0;
        const int32_t h = holder[(size_t)i];
        if (h < 0) continue;
        const qk_part &p = s->parts[(size_t)h];
        int64_t at = -1;
        if (asked[(size_t)h] >= 8 && p.size > 64) {
            QkIdMap &m = pos[h];
            if (m.size() == 0) {
                m.reserve((size_t)p.size);
                for (int64_t r = p.size - 1; r >= 0; r--) m.set(p.ids[(size_t)r], (int32_t)r);  // (first occurrence wins: find_id's order)
            }
            at = m.find(ids_host[i]);
        } else {
            for (int64_t r = 0; r < p.size; r++)
                if (p.ids[(size_t)r] == ids_host[i]) {
                    at = r;
                    break;
                }
        }
        if (at < 0) continue;
        rows[(size_t)i] = p.row_off + at;
        found[i] = 1;
        n_found++;
    }
    if (n_found == 0) return QK_OK;
    const size_t rb = ((size_t)n * 8 + 255) & ~(size_t)255, vb = (size_t)n * s->d * sizeof(float);
    QK_TRY(qk_stage_reserve(c, rb + vb));
    int64_t *drows = (int64_t *)c->stage;
    float *dv = (float *)(c->stage + rb);
    for (auto &r : rows)
        if (r < 0) r = 0;  // (absent ids: any valid row; the caller looks at found[])
    QK_HIP(hipMemcpyAsync(drows, rows.data(), (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    QK_TRY(qk_launch_extract(c, s->vecs, s->nblk, s->d, 0, drows, n, dv));
    QK_HIP(hipMemcpyAsync(vecs_out_host, dv, vb, hipMemcpyDeviceToHost, c->stream));
    QK_HIP(hipStreamSynchronize(c->stream));
    return QK_OK;
}

int qk_store_publish(qk_store *s) {
    if (!s) QK_FAIL(QK_ERR_INVALID, "qk_store_publish: null store");
    QK_HIP(hipSetDevice(s->ctx->device));
    QK_TRY(qk_store_sync_table(s));
    // the row-major copy a one-list store (the parent: its centroids) keeps for the coarse step's exact finish: rebuilt here if the
    // store had one, so that the next search finds it
    if (s->rowmajor_rows > 0 && !s->rowmajor_valid) {
        int64_t only = -1, present = 0;
        for (size_t p = 0; p < s->parts.size(); p++)
            if (s->parts[p].present) {
                present++;
                only = (int64_t)p;
            }
        if (present == 1 && s->parts[(size_t)only].size > 0 && s->parts[(size_t)only].size <= INT32_MAX) {
            const float *rm = nullptr;
            QK_TRY(qk_store_rowmajor(s, s->parts[(size_t)only].row_off, (int)s->parts[(size_t)only].size, &rm));
        }
    }
    return QK_OK;
}

int qk_store_get_lists(qk_store *s, const int64_t *list_nos, int64_t n, float *vecs_out, int64_t *ids_out, int mem) {
    if (!s || (n > 0 && !list_nos)) QK_FAIL(QK_ERR_INVALID, "qk_store_get_lists: null argument");
    int64_t at = 0;
    for (int64_t i = 0; i < n; i++) {
        QK_TRY(check_list(s, list_nos[i], "get_codes"));
        const int64_t sz = s->parts[list_nos[i]].size;
        QK_TRY(qk_store_get_list(s, list_nos[i], vecs_out ? vecs_out + at * s->d : nullptr, ids_out ? ids_out + at : nullptr, mem));
        at += sz;
    }
    return QK_OK;
}

int qk_store_get_vector(qk_store *s, int64_t id, float *vec_out_host, int *found) {
    if (!s || !vec_out_host || !found) QK_FAIL(QK_ERR_INVALID, "qk_store_get_vector: null argument");
    qk_ctx *c = s->ctx;
    QK_HIP(hipSetDevice(c->device));
    *found = 0;
    qk_store_ensure_index(s);
    const int32_t holder = s->id_to_list.find(id);
    if (holder < 0) return QK_OK;
    qk_part &p = s->parts[(size_t)holder];
    for (int64_t i = 0; i < p.size; i++)  // find_id inside the one list that holds it (index_partition.cpp:129-145)
        if (p.ids[i] == id) {
            size_t vb = (size_t)s->d * sizeof(float);
            QK_TRY(qk_stage_reserve(c, vb));
            QK_TRY(qk_launch_extract(c, s->vecs, s->nblk, s->d, p.row_off + i, nullptr, 1, (float *)c->stage));
            QK_HIP(hipMemcpyAsync(vec_out_host, c->stage, vb, hipMemcpyDeviceToHost, c->stream));
            QK_HIP(hipStreamSynchronize(c->stream));
            *found = 1;
            return QK_OK;
        }
    return QK_OK;
}

// Batched PartitionManager::add (partition_manager.cpp:236-258): n vectors, each appended to list assign[i]; per-list
// append order = input order.  One grouping pass on the host, one capacity check per touched list, ONE ingest launch.
static int add_batch_core(qk_store *s, int64_t n, const int64_t *ids, const float *vecs, int mem, const std::vector<int64_t> &h_assign,
                          const std::vector<int64_t> &h_ids, bool ids_indexed = false);

int qk_store_add_batch(qk_store *s, int64_t n, const int64_t *ids, const float *vecs, const int64_t *assign, int mem) {
    if (!s) QK_FAIL(QK_ERR_INVALID, "qk_store_add_batch: null store");
    if (n == 0) return QK_OK;
    if (n < 0 || !ids || !vecs || !assign) QK_FAIL(QK_ERR_INVALID, "qk_store_add_batch: bad arguments");
    qk_ctx *c = s->ctx;
    QK_HIP(hipSetDevice(c->device));
    std::vector<int64_t> h_assign((size_t)n), h_ids((size_t)n);
    if (mem == QK_MEM_HOST) {
        memcpy(h_assign.data(), assign, (size_t)n * 8);
        memcpy(h_ids.data(), ids, (size_t)n * 8);
    } else {
        QK_HIP(hipMemcpyAsync(h_assign.data(), assign, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
        QK_HIP(hipMemcpyAsync(h_ids.data(), ids, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
        QK_HIP(hipStreamSynchronize(c->stream));
    }
    return add_batch_core(s, n, ids, vecs, mem, h_assign, h_ids);
}

}  // extern "C"

// rows and ids on the DEVICE, the list of every row known on the HOST (qk_store_refine_lists: its rows come out of the bucketing
// grouped by list, the counts are on the host): one ingest for all of them instead of an add_entries -- a launch, a copy of the
// ids and a synchronisation -- per list
int qk_store_add_batch_host_assign(qk_store *s, int64_t n, const int64_t *ids_dev, const float *vecs_dev, const std::vector<int64_t> &h_assign,
                                   bool ids_indexed) {
    if (n == 0) return QK_OK;
    qk_ctx *c = s->ctx;
    QK_HIP(hipSetDevice(c->device));
    std::vector<int64_t> h_ids((size_t)n);
    QK_HIP(hipMemcpyAsync(h_ids.data(), ids_dev, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    QK_HIP(hipStreamSynchronize(c->stream));
    return add_batch_core(s, n, ids_dev, vecs_dev, QK_MEM_DEVICE, h_assign, h_ids, ids_indexed);
}

static int add_batch_core(qk_store *s, int64_t n, const int64_t *ids, const float *vecs, int mem, const std::vector<int64_t> &h_assign,
                          const std::vector<int64_t> &h_ids, bool ids_indexed) {
    qk_ctx *c = s->ctx;
    std::vector<int64_t> extra(s->parts.size(), 0);
    for (int64_t i = 0; i < n; i++) {
        QK_TRY(check_list(s, h_assign[i], "add_entries"));
        extra[(size_t)h_assign[i]]++;
    }
    // what the batch's relocations will ask of the arena, asked for ONCE: a bulk-built store has exactly-sized lists, and the first
    // large add relocated 3000 of them through three arena re-allocations and a compaction (each a copy of the whole arena plus
    // multi-GB hipMalloc / hipFree: 50 ... 800 ms for 65536 vectors)
    {
        int64_t grow = 0;
        for (size_t p = 0; p < extra.size(); p++) {
            const qk_part &pt = s->parts[p];
            if (extra[p] && pt.size + extra[p] > pt.cap) grow += qk_round_up64(std::max<int64_t>(pt.size + extra[p], pt.cap * 2), 16);
        }
        if (grow > 0 && s->used_rows + grow > s->cap_rows && !(s->dead_rows * 4 >= s->cap_rows && s->dead_rows >= 1024))
            QK_TRY(qk_store_reserve_rows(s, grow));
    }
    for (size_t p = 0; p < extra.size(); p++)
        if (extra[p]) QK_TRY(ensure_part_capacity(s, s->parts[p], extra[p]));
    std::vector<int64_t> rows((size_t)n);
    // ids_indexed: every id is in the index already (under the list it just left): its entry is overwritten in place, which needs no
    // slot to change state -- shared out over threads for a large batch (2M rows of a 50M index: 80 ms of cache misses on one thread)
    const bool index_here = s->index_valid && !ids_indexed;
    if (index_here) s->id_to_list.reserve(s->id_to_list.size() + (size_t)n);  // (no rehash inside the loop)
    constexpr int64_t AHEAD = 16;
    // (runs of rows bound for the same list -- a refinement hands its rows over grouped by list, a split by half -- are appended to the
    //  list's id mirror in one piece)
    for (int64_t i = 0; i < n;) {
        int64_t j = i + 1;
        while (j < n && h_assign[j] == h_assign[i]) j++;
        qk_part &p = s->parts[(size_t)h_assign[i]];
        const int64_t at = p.row_off + p.size;
        for (int64_t t = i; t < j; t++) rows[t] = at + (t - i);
        p.ids.insert(p.ids.end(), h_ids.begin() + i, h_ids.begin() + j);
        p.size += j - i;
        for (int64_t t = i; t < j; t++) {
            if (index_here && t + AHEAD < n) s->id_to_list.prefetch(h_ids[t + AHEAD]);
            if (h_ids[t] > s->max_id_seen) s->max_id_seen = h_ids[t];
            if (h_ids[t] < s->min_id_seen) s->min_id_seen = h_ids[t];
            if (index_here) s->id_to_list.set(h_ids[t], (int32_t)h_assign[t]);
        }
        i = j;
    }
    if (s->index_valid && ids_indexed) {
        const unsigned hw = std::thread::hardware_concurrency();
        const int T = (int)std::min<int64_t>(std::min<unsigned>(16u, hw ? hw : 1u), std::max<int64_t>(1, n >> 16));
        std::vector<std::vector<int64_t>> missing((size_t)T);
        qk_run_shares(T, [&](int t) {
            const int64_t a = n * t / T, b = n * (t + 1) / T;
            for (int64_t i = a; i < b; i++) {
                if (i + AHEAD < b) s->id_to_list.prefetch(h_ids[i + AHEAD]);
                if (!s->id_to_list.overwrite_present(h_ids[i], (int32_t)h_assign[i])) missing[(size_t)t].push_back(i);
            }
        });
        for (auto &mv : missing)  // (an id the index did not hold: inserted the plain way)
            for (int64_t i : mv) s->id_to_list.set(h_ids[i], (int32_t)h_assign[i]);
    }
    s->ntotal += n;
    s->table_dirty = true;
    size_t vb = (size_t)n * s->d * 4, ib = (size_t)n * 8;
    size_t off_rows = (mem == QK_MEM_HOST) ? (((vb + 255) & ~(size_t)255) + ((ib + 255) & ~(size_t)255)) : 0;
    QK_TRY(qk_stage_reserve(c, off_rows + ib + 256));
    const float *dv = vecs;
    const int64_t *di = ids;
    if (mem == QK_MEM_HOST) {
        QK_HIP(hipMemcpyAsync(c->stage, vecs, vb, hipMemcpyHostToDevice, c->stream));
        QK_HIP(hipMemcpyAsync(c->stage + ((vb + 255) & ~(size_t)255), ids, ib, hipMemcpyHostToDevice, c->stream));
        dv = (const float *)c->stage;
        di = (const int64_t *)(c->stage + ((vb + 255) & ~(size_t)255));
    }
    int64_t *drows = (int64_t *)(c->stage + off_rows);
    QK_HIP(hipMemcpyAsync(drows, rows.data(), ib, hipMemcpyHostToDevice, c->stream));
    IngestMap m{0, nullptr, nullptr, 0, 0, drows};
    QK_TRY(launch_ingest(c, dv, di, n, s->d, s->nblk, s->vecs, s->norms, s->ids, m));
    QK_HIP(hipStreamSynchronize(c->stream));
    return QK_OK;
}

extern "C" {

int qk_store_counters(qk_store *s, int64_t *out, int n) {
    if (!s || !out || n <= 0) QK_FAIL(QK_ERR_INVALID, "qk_store_counters: bad arguments");
    for (int i = 0; i < n; i++) out[i] = i < 8 ? s->counters[i] : 0;
    if (n > 7) out[7] = s->ctx->scratch_reallocs;
    return QK_OK;
}

int64_t qk_store_device_bytes(qk_store *s) {
    if (!s) return 0;
    return s->cap_rows * ((int64_t)s->dpad * 4 + 4 + 8);
}

}  // extern "C"
