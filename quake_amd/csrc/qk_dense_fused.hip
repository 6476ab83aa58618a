// qk_dense_fused.hip -- top-k of every query over ONE list of a few thousand rows: exact keys and their selection in one launch.
//
// The coarse step of QueryCoordinator::search over a parent of 1024-4096 centroids (src/cpp/src/query_coordinator.cpp:628-644 ->
// batched_scan_list(x, centroids, ..., k = nprobe), src/cpp/include/list_scanning.h:313-366).  The other dense forms are sized
// for long lists: k_dense_ord + k_select_rows write and re-read a [Q][n] key matrix, the prefiltered form (qk_dense_pf.hip) is four
// dependent launches of 5-18 us each whatever the size.  At a few thousand rows the whole product is ~8 us of fp32 MFMA work, so:
//   k_dense_fused<DB,NQ,SR,L2>  workgroup (slice, query tile): NQ*16 queries x SR rows on v_mfma_f32_16x16x4_f32 (the canonical
//                       chain of k_scan / k_dense_ord: same bits), the keys stay in LDS; then every wave takes the k-th smallest
//                       of the SR keys of four queries side by side by bisection on the key bits (registers + ballots): the k
//                       rows at or under it are the slice's candidates, written as [Q][slices][k] (id, key) -- unsorted; only a
//                       tie on the k-th key (duplicate rows) goes through compact_pool's (key, id) order
//   k_merge_slices      (qk_dense.hip) one wave per query over slices * k <= 1024 candidates -> ids + distances
// No key ever leaves the CU; the centroid rows are read from the L2 once per query tile and slice.  Serves 1024-2048 rows (to 4096
// for k > 32), see fused_plan; beyond that the prefiltered form is faster (the merge wave's slices * k candidates grow with the rows).
#include "qk_internal.h"

#include "qk_device.h"

#include <algorithm>

struct FusedParams {
    const float4 *vecs;  // arena
    const float *norms;
    const int64_t *ids;  // arena ids + row_off
    int64_t row_off;     // first arena row of the list (multiple of 16)
    int nrows;
    int nblk;
    const float4 *xq4;   // [Q][nblk][4] fragment-ordered queries
    const float *xn;     // [Q]
    int64_t Q;
    int k;
    int slices;
    int64_t *out_ids;    // [Q][slices][k]
    uint32_t *out_ord;
};

constexpr int FU_POOL = 128;  // pool entries per wave: k <= 64 survivors + one chunk of 64 lanes

template <int DB, int NQ, int SR, bool L2>
__global__ __launch_bounds__(256) void k_dense_fused(FusedParams P) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int LDK = SR + 4;   // key row stride in words: the 16 query rows of a b128 store land in 16 different bank groups
    constexpr int R = SR / 64;    // keys per lane in the selection
    constexpr int TPW = SR / 64;  // row tiles per wave (SR / 16 tiles over 4 waves)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int nblk = P.nblk;
    constexpr bool l2 = L2;
    float4 *qs = (float4 *)smem;                                             // [NQ][nblk*64]
    float *xn_s = (float *)(smem + (size_t)NQ * nblk * 1024);                // [NQ*16]
    uint32_t *keys = (uint32_t *)(smem + (size_t)NQ * nblk * 1024 + NQ * 64);  // [NQ*16][LDK]
    unsigned char *pools = (unsigned char *)(keys + (size_t)NQ * 16 * LDK);
    int64_t *pool_id = (int64_t *)(pools + (size_t)wave * FU_POOL * 12);
    uint32_t *pool_ord = (uint32_t *)(pools + (size_t)wave * FU_POOL * 12 + FU_POOL * 8);
    const int slice = blockIdx.x;
    const int64_t q_base = (int64_t)blockIdx.y * (NQ * 16);


    // the ids of the slice's rows, lane + 64 i as in the selection: requested now, used after the products (a gather of the
    // selected rows' ids at the end was one more dependent memory round trip per wave)
    int64_t idv[R];
#pragma unroll
    for (int i = 0; i < R; i++) idv[i] = P.ids[min(slice * SR + lane + 64 * i, P.nrows - 1)];

    const int ntile_all = (P.nrows + 15) >> 4;
    const int wg_t0 = slice * (SR / 16);
    const int t0 = wg_t0 + wave * TPW, t1 = min(ntile_all, t0 + TPW);
    // tiles past the end of the list: dead keys
    for (int t = max(t0, t1); t < t0 + TPW; t++) {
        const int c0 = (t - wg_t0) * 16 + 4 * g;
#pragma unroll
        for (int nq = 0; nq < NQ; nq++)
            *(uint4 *)(keys + (size_t)(nq * 16 + j) * LDK + c0) = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    }
    const int ncd = nblk / DB;
    const int64_t tile_abs0 = (P.row_off >> 4) + t0;
    const float4 *src = P.vecs + tile_abs0 * nblk * 64 + lane;
    const float4 *nsrc = (const float4 *)(P.norms + (tile_abs0 << 4)) + g;
    const int nsteps = (t1 - t0) * ncd;
    float4 a0[DB], a1[DB];
    float4 yn_cur = make_float4(0.f, 0.f, 0.f, 0.f), yn_next = yn_cur;
    f32x4 acc[NQ];
    int dch = 0, tile = t0, ldch = 0, ltile = 0;

#define FU_LOAD(A, S)                                                 \
    {                                                                 \
        const float4 *pp_ = src + (int64_t)(S) * (DB * 64);           \
        _Pragma("unroll") for (int b_ = 0; b_ < DB; b_++) A[b_] = pp_[b_ * 64]; \
        if (ldch == 0) {                                              \
            if (l2) yn_next = nsrc[(int64_t)ltile * 4];               \
            ltile++;                                                  \
        }                                                             \
        if (++ldch == ncd) ldch = 0;                                  \
    }

#define FU_STEP(A)                                                                                           \
    {                                                                                                        \
        if (dch == 0) {                                                                                      \
            _Pragma("unroll") for (int nq_ = 0; nq_ < NQ; nq_++) acc[nq_] = (f32x4){0.f, 0.f, 0.f, 0.f};     \
        }                                                                                                    \
        _Pragma("unroll") for (int b_ = 0; b_ < DB; b_++) {                                                  \
            float4 bq_[NQ];                                                                                  \
            _Pragma("unroll") for (int nq_ = 0; nq_ < NQ; nq_++)                                             \
                bq_[nq_] = qs[(size_t)nq_ * nblk * 64 + (dch * DB + b_) * 64 + lane];                        \
            _Pragma("unroll") for (int nq_ = 0; nq_ < NQ; nq_++)                                             \
                acc[nq_] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[b_].x, bq_[nq_].x, acc[nq_], 0, 0, 0);     \
            _Pragma("unroll") for (int nq_ = 0; nq_ < NQ; nq_++)                                             \
                acc[nq_] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[b_].y, bq_[nq_].y, acc[nq_], 0, 0, 0);     \
            _Pragma("unroll") for (int nq_ = 0; nq_ < NQ; nq_++)                                             \
                acc[nq_] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[b_].z, bq_[nq_].z, acc[nq_], 0, 0, 0);     \
            _Pragma("unroll") for (int nq_ = 0; nq_ < NQ; nq_++)                                             \
                acc[nq_] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[b_].w, bq_[nq_].w, acc[nq_], 0, 0, 0);     \
        }                                                                                                    \
        if (++dch == ncd) {                                                                                  \
            dch = 0;                                                                                         \
            const float yv_[4] = {yn_cur.x, yn_cur.y, yn_cur.z, yn_cur.w};                                   \
            const int row0_ = (tile << 4) + 4 * g;                                                           \
            const int c0_ = ((tile - wg_t0) << 4) + 4 * g;                                                   \
            _Pragma("unroll") for (int nq_ = 0; nq_ < NQ; nq_++) {                                           \
                uint4 o_;                                                                                    \
                uint32_t *op_ = (uint32_t *)&o_;                                                             \
                _Pragma("unroll") for (int reg_ = 0; reg_ < 4; reg_++) {                                     \
                    const float v_ = acc[nq_][reg_];                                                         \
                    const uint32_t k_ = l2 ? ord_from_l2(l2_expanded(xnj[nq_], yv_[reg_], v_)) : ord_from_ip(v_); \
                    op_[reg_] = (row0_ + reg_ < P.nrows) ? k_ : 0xFFFFFFFFu;                                 \
                }                                                                                            \
                *(uint4 *)(keys + (size_t)(nq_ * 16 + j) * LDK + c0_) = o_;                                  \
            }                                                                                                \
            yn_cur = yn_next;                                                                                \
            tile++;                                                                                          \
        }                                                                                                    \
    }

    // the workgroup's queries -> LDS; the first row tile (and the ids above) are requested before the barrier, so the two memory
    // round trips of the prologue overlap
    for (int t = wave; t < NQ * nblk; t += 4) {
        const int nq = t / nblk, cb = t - nq * nblk;
        const int64_t row = q_base + nq * 16 + j;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < P.Q) v = P.xq4[(row * nblk + cb) * 4 + g];
        qs[(size_t)nq * nblk * 64 + cb * 64 + lane] = v;
    }
    if (tid < NQ * 16) {
        const int64_t row = q_base + tid;
        xn_s[tid] = (row < P.Q && l2) ? P.xn[row] : 0.0f;
    }
    if (t1 > t0) FU_LOAD(a0, 0);
    __syncthreads();
    if (t1 > t0) {
        float xnj[NQ];
#pragma unroll
        for (int nq = 0; nq < NQ; nq++) xnj[nq] = xn_s[nq * 16 + j];
        yn_cur = yn_next;
        int s = 0;
        while (s < nsteps) {
            if (s + 1 < nsteps) FU_LOAD(a1, s + 1);
            FU_STEP(a0);
            s++;
            if (s >= nsteps) break;
            if (s + 1 < nsteps) FU_LOAD(a0, s + 1);
            FU_STEP(a1);
            s++;
        }
#undef FU_LOAD
#undef FU_STEP
    }
    __syncthreads();

    // ---- selection: wave w takes queries w, w + 4, ... of the tile ----
    constexpr int QW = NQ * 4;  // queries per wave
    const int k = P.k;
    const int row_base = slice * SR;
    uint32_t o[QW][R];
#pragma unroll
    for (int u = 0; u < QW; u++) {
        const uint32_t *kr = keys + (size_t)(wave + 4 * u) * LDK;
#pragma unroll
        for (int i = 0; i < R; i++) o[u][i] = kr[lane + 64 * i];
    }
    // T[u] = the k-th smallest key of query u's slice (0xFFFFFFFF when the slice holds fewer than k rows): the largest value with
    // fewer than k keys below it, bit by bit -- the QW queries of the wave side by side (independent compare / count chains)
    uint32_t T[QW];
#pragma unroll
    for (int u = 0; u < QW; u++) T[u] = 0;
    for (int b = 31; b >= 0; b--) {
#pragma unroll
        for (int u = 0; u < QW; u++) {
            const uint32_t tr = T[u] | (1u << b);
            int c = 0;
#pragma unroll
            for (int i = 0; i < R; i++) c += __popcll(__ballot(o[u][i] < tr));
            if (c < k) T[u] = tr;
        }
    }
    uint32_t tie_mask = 0;
#pragma unroll
    for (int u = 0; u < QW; u++) {
        const int64_t q = q_base + wave + 4 * u;
        if (q >= P.Q) continue;
        const int64_t ob = (q * P.slices + slice) * k;
        int c_le = 0;
#pragma unroll
        for (int i = 0; i < R; i++) c_le += __popcll(__ballot(o[u][i] <= T[u] && o[u][i] != 0xFFFFFFFFu));
        if (c_le <= k) {
            // exactly the k rows at or under T (or every row of a slice shorter than k): the SET is settled without the ids; it leaves
            // unsorted -- the merge orders all candidates of the query under (key, id) anyway
            int n = 0;
#pragma unroll
            for (int i = 0; i < R; i++) {
                const bool pass = o[u][i] <= T[u] && o[u][i] != 0xFFFFFFFFu;
                const uint64_t m = __ballot(pass);
                if (pass) {
                    const int sl = n + __popcll(m & ((1ull << lane) - 1ull));
                    P.out_ids[ob + sl] = idv[i];
                    P.out_ord[ob + sl] = o[u][i];
                }
                n += __popcll(m);
            }
            for (int e = n + lane; e < k; e += 64) {
                P.out_ids[ob + e] = -1;
                P.out_ord[ob + e] = 0xFFFFFFFFu;
            }
            continue;
        }
        tie_mask |= 1u << u;
    }
    // several rows share the k-th key (duplicates): which of them stay is a matter of their ids -- through the pool, compact_pool
    // keeps the k best under (key, id).  Rare: ONE rolled copy of this path for the wave's queries (unrolled per query and chunk it
    // was 40 of the kernel's 54 KB of code), the keys re-read from LDS
    for (int u = 0; tie_mask >> u; u++) {
        if (!((tie_mask >> u) & 1u)) continue;
        const int64_t q = q_base + wave + 4 * u;
        const int64_t ob = (q * P.slices + slice) * k;
        uint32_t Tu = T[0];
#pragma unroll
        for (int v = 1; v < QW; v++) Tu = u == v ? T[v] : Tu;
        const uint32_t *kr = keys + (size_t)(wave + 4 * u) * LDK;
        int n = 0, n_ids = 0;
        for (int i = 0; i < R; i++) {
            const uint32_t ov = kr[lane + 64 * i];
            const bool pass = ov <= Tu && ov != 0xFFFFFFFFu;
            const uint64_t m = __ballot(pass);
            if (pass) {
                const int sl = n + __popcll(m & ((1ull << lane) - 1ull));
                pool_ord[sl] = ov;
                pool_id[sl] = row_base + lane + 64 * i;  // the ROW for now
            }
            n += __popcll(m);
            if (n > FU_POOL - 64 || i == R - 1) {
                for (int e = n_ids + lane; e < n; e += 64) pool_id[e] = P.ids[pool_id[e]];
                n = compact_pool<2>(pool_ord, pool_id, n, k, lane);
                n_ids = n;
            }
        }
        for (int e = lane; e < k; e += 64) {
            P.out_ids[ob + e] = e < n ? pool_id[e] : -1;
            P.out_ord[ob + e] = e < n ? pool_ord[e] : 0xFFFFFFFFu;
        }
    }
}

struct FusedPlan {
    int NQ, SR, slices, DB;
    int64_t qtiles;
    size_t lds;
};

static bool fused_plan(const qk_ctx *ctx, const qk_store *s, int64_t Q, int nrows, int k, FusedPlan *pl) {
    // where this form is the fastest of the dense ones (1024 queries, scripts/coarse_probe.py, same box, prefiltered form -> this
    // one, us per call at k = 2 / 8 / 32 / 64): 1024 rows 27 / 29 / 37 / 57 -> 22 / 22 / 24 / 28, 2048 rows 28 / 31 / 44 / 77 ->
    // 27 / 27 / 35 / 35; 3072 rows 30 / 33 / 45 / 101 against 32 / 33 / 38 / 49, 4096 rows 31 / 34 / 47 / 69 against 36 / 37 / 47 /
    // 62: up to 2048 rows, to 3072 at k = 32 and to 4096 beyond -- the prefiltered form loses where its candidate lists grow (large
    // k), this one where the merge wave's slices * k candidates do (more rows)
    if (k < 2 || k > 64 || nrows < 1024 || nrows > (k > 32 ? 4096 : k == 32 ? 3072 : 2048) || Q < 64 || Q > 65536) return false;
    const int nblk = s->nblk;
    if (nblk > 8) return false;  // d <= 128: the query tile and the keys share the LDS
    // 16 queries per workgroup (four per wave in the selection, side by side); slices of 256 rows -- 512 only beyond 3072 rows at
    // k > 32, where the merge wave would hold 1024 candidates (k = 64, 256- against 512-row slices: 1024 rows 28 against 37 us, 2048:
    // 35 / 38, 3072: 45 / 47, 4096: 59 / 56).  Measured against it at 4096 rows, k = 2 / 32: 32 queries per workgroup (half the row
    // traffic from the L2, twice the selection work per wave) 43 / 59 us against 38 / 53; 1024-row slices (a quarter of the
    // candidates, one workgroup per CU) 55 / 61
    const int NQ = 1, SR = (k <= 32 || nrows <= 3072) ? 256 : 512;
    const int slices = (nrows + SR - 1) / SR;
    if (slices * k > 1024) return false;
    pl->NQ = NQ;
    pl->SR = SR;
    pl->slices = slices;
    pl->DB = (nblk % 8 == 0) ? 8 : (nblk % 4 == 0) ? 4 : (nblk % 2 == 0) ? 2 : 1;
    pl->qtiles = (Q + NQ * 16 - 1) / (NQ * 16);
    pl->lds = (size_t)NQ * nblk * 1024 + (size_t)NQ * 64 + (size_t)NQ * 16 * (SR + 4) * 4 + (size_t)4 * FU_POOL * 12;
    return true;
}

bool qk_dense_fused_supported(const qk_ctx *ctx, const qk_store *s, int64_t Q, int nrows, int k) {
    FusedPlan pl;
    return fused_plan(ctx, s, Q, nrows, k, &pl);
}

template <int DB, int NQ, int SR, bool L2>
static int fused_launch(hipStream_t st, dim3 grid, size_t lds, const FusedParams &p) {
    QK_HIP(hipFuncSetAttribute((const void *)k_dense_fused<DB, NQ, SR, L2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_dense_fused<DB, NQ, SR, L2>), grid, dim3(256), lds, st, p);
    return QK_OK;
}
template <int DB, int NQ, int SR>
static int fused_launch_m(hipStream_t st, dim3 grid, size_t lds, const FusedParams &p, int metric) {
    return metric == QK_METRIC_L2 ? fused_launch<DB, NQ, SR, true>(st, grid, lds, p) : fused_launch<DB, NQ, SR, false>(st, grid, lds, p);
}
template <int DB>
static int fused_launch_s(hipStream_t st, dim3 grid, size_t lds, const FusedParams &p, int metric, int NQ, int SR) {
    if (SR == 512) return fused_launch_m<DB, 1, 512>(st, grid, lds, p, metric);
    return fused_launch_m<DB, 1, 256>(st, grid, lds, p, metric);
}

// top-k of every query over the rows [row_off, row_off + nrows) of one list (a.xq4 / a.xn: the prepared queries)
int qk_dense_fused_device(qk_ctx *ctx, qk_store *s, int64_t row_off, int nrows, const qk_scan_args &a) {
    const int64_t Q = a.Q;
    const int k = a.k;
    FusedPlan pl;
    if (!fused_plan(ctx, s, Q, nrows, k, &pl)) QK_FAIL(QK_ERR_UNSUPPORTED, "dense top-k (fused): unsupported shape");
    const size_t n_c = (size_t)Q * pl.slices * k;
    QK_TRY(qk_ws_reserve(ctx, n_c * 12 + 4096));
    int64_t *sl_ids = (int64_t *)qk_ws_alloc(ctx, n_c * 8);
    uint32_t *sl_ord = (uint32_t *)qk_ws_alloc(ctx, n_c * 4);
    if (!sl_ids || !sl_ord) QK_FAIL(QK_ERR_OOM, "dense top-k (fused): workspace exhausted");
    FusedParams p;
    p.vecs = (const float4 *)s->vecs;
    p.norms = s->norms;
    p.ids = s->ids + row_off;
    p.row_off = row_off;
    p.nrows = nrows;
    p.nblk = s->nblk;
    p.xq4 = a.xq4;
    p.xn = a.xn;
    p.Q = Q;
    p.k = k;
    p.slices = pl.slices;
    p.out_ids = sl_ids;
    p.out_ord = sl_ord;
    const dim3 grid((unsigned)pl.slices, (unsigned)pl.qtiles);
    hipStream_t st = ctx->stream;
    switch (pl.DB) {
        case 8: QK_TRY(fused_launch_s<8>(st, grid, pl.lds, p, a.metric, pl.NQ, pl.SR)); break;
        case 4: QK_TRY(fused_launch_s<4>(st, grid, pl.lds, p, a.metric, pl.NQ, pl.SR)); break;
        case 2: QK_TRY(fused_launch_s<2>(st, grid, pl.lds, p, a.metric, pl.NQ, pl.SR)); break;
        default: QK_TRY(fused_launch_s<1>(st, grid, pl.lds, p, a.metric, pl.NQ, pl.SR)); break;
    }
    QK_HIP(hipGetLastError());
    return qk_launch_merge_slices(ctx, sl_ids, sl_ord, Q, pl.slices, k, a.metric, a.sqrt_l2, a.out_ids, a.out_dist);
}
