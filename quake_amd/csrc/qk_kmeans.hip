// qk_kmeans.hip -- k-means assign / update kernels and the Lloyd driver.
//
// Replaces (citations relative to the reference checkout):
//   kmeans()                     src/cpp/src/clustering.cpp:13-97   (faiss::Clustering::train + IndexFlat::search(k=1))
//   kmeans_refine_partitions()   src/cpp/src/clustering.cpp:99-182  (batched_scan_list(k=1) assign :149-159,
//                                                                    scalar accumulate :162-176)
// assign  : X x C^T on v_mfma_f32_16x16x4_f32 (same canonical fmaf-chain as the scan) + fused argmin, MFMA-bound
// update  : rows bucketed stably by assignment (k_rs_*: the repo's own count / scan / scatter, one pass per 8 bits of the
//           centroid number), then one workgroup per centroid adds its rows in ascending row order (fp32) -- the sequential
//           order of the reference loop, so sums are bit-reproducible. HBM-bound.
// PARITY UNPINNED vs FAISS (RNG / init), pinned vs oracle/quake_oracle.c (qo_kmeans*).
#include "qk_internal.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t km_ord_l2(float d2) { return __float_as_uint(d2); }
__device__ __forceinline__ uint32_t km_ord_ip(float ip) {
    uint32_t b = __float_as_uint(ip);
    uint32_t asc = b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
    return ~asc;
}
__device__ __forceinline__ float km_ip_from_ord(uint32_t o) {
    uint32_t asc = ~o;
    uint32_t b = (asc & 0x80000000u) ? (asc ^ 0x80000000u) : ~asc;
    return __uint_as_float(b);
}
__device__ __forceinline__ float km_l2_expanded(float xn, float yn, float ip) {
    float r = __fmaf_rn(-2.0f, ip, xn + yn);
    return r < 0.0f ? 0.0f : r;
}

struct AssignParams {
    const float *x;        // [n][d] row-major
    int64_t n;
    int d;
    int nblk;
    const float4 *cvecs;   // centroids, tile-major, mt tiles
    const float *cnorms;   // [mt*16]
    int m;
    int metric;
    int64_t *assign;       // [n]
    float *val;            // [n] or nullptr
};

// One workgroup = NQ*16 rows of x (the MFMA "query" side, staged in LDS) against all centroids (streamed as A
// operands from L2/HBM, split across the 4 waves), running (ord, index) argmin per row.
// (L2 = the metric at compile time: as a runtime flag it left a uniform branch per result element in the epilogue)
template <int DB, int NQ, bool L2>
__global__ __launch_bounds__(256) void k_assign(AssignParams P) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int nblk = P.nblk, d = P.d;
    constexpr bool l2 = L2;
    float4 *qs = (float4 *)smem;                                            // [NQ][nblk*64]
    float *xn_s = (float *)(smem + (size_t)NQ * nblk * 1024);               // [NQ*16]
    uint32_t *red_ord = (uint32_t *)(xn_s + NQ * 16);                       // [4][NQ*16]
    int *red_idx = (int *)(red_ord + 4 * NQ * 16);                          // [4][NQ*16]
    const int64_t row_base = (int64_t)blockIdx.x * (NQ * 16);

    // stage the x rows in B-operand lane order
    for (int t = wave; t < NQ * nblk; t += 4) {
        const int nq = t / nblk, cb = t - nq * nblk;
        const int64_t row = row_base + nq * 16 + j;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < P.n) {
            const float *s = P.x + row * d;
            const int col = 16 * cb + g;
            v.x = col < d ? s[col] : 0.0f;
            v.y = col + 4 < d ? s[col + 4] : 0.0f;
            v.z = col + 8 < d ? s[col + 8] : 0.0f;
            v.w = col + 12 < d ? s[col + 12] : 0.0f;
        }
        qs[(size_t)nq * nblk * 64 + cb * 64 + lane] = v;
    }
    if (tid < NQ * 16) {
        const int64_t row = row_base + tid;
        float acc = 0.0f;
        if (row < P.n && l2) {
            const float *s = P.x + row * d;
            for (int k = 0; k < d; k++) acc = __fmaf_rn(s[k], s[k], acc);
        }
        xn_s[tid] = acc;
    }
    __syncthreads();

    float xnj[NQ];
    uint32_t best_ord[NQ];
    int best_idx[NQ];
#pragma unroll
    for (int nq = 0; nq < NQ; nq++) {
        xnj[nq] = xn_s[nq * 16 + j];
        best_ord[nq] = 0xFFFFFFFFu;
        best_idx[nq] = 0x7FFFFFFF;
    }
    const int mt = (P.m + 15) >> 4;
    const int tpw = (mt + 3) >> 2;
    const int t0 = wave * tpw, t1 = min(mt, t0 + tpw);
    const int ncd = nblk / DB;
    if (t1 > t0) {
        const float4 *src = P.cvecs + (int64_t)t0 * nblk * 64 + lane;
        const float4 *nsrc = (const float4 *)(P.cnorms + ((int64_t)t0 << 4)) + g;
        const int nsteps = (t1 - t0) * ncd;
        float4 a0[DB], a1[DB];
        float4 yn_cur = make_float4(0.f, 0.f, 0.f, 0.f), yn_next = yn_cur;
        f32x4 acc[NQ];
        int dch = 0, tile = t0, ldch = 0, ltile = 0;

#define KM_LOAD(A, S)                                                 \
    {                                                                 \
        const float4 *pp_ = src + (int64_t)(S) * (DB * 64);           \
        _Pragma("unroll") for (int b_ = 0; b_ < DB; b_++) A[b_] = pp_[b_ * 64]; \
        if (ldch == 0) {                                              \
            if (l2) yn_next = nsrc[(int64_t)ltile * 4];               \
            ltile++;                                                  \
        }                                                             \
        if (++ldch == ncd) ldch = 0;                                  \
    }

#define KM_STEP(A)                                                                                           \
    {                                                                                                        \
        if (dch == 0) {                                                                                      \
            _Pragma("unroll") for (int nq_ = 0; nq_ < NQ; nq_++) acc[nq_] = (f32x4){0.f, 0.f, 0.f, 0.f};     \
        }                                                                                                    \
        _Pragma("unroll") for (int b_ = 0; b_ < DB; b_++) {                                                  \
            _Pragma("unroll") for (int nq_ = 0; nq_ < NQ; nq_++) {                                           \
                const float4 bq_ = qs[(size_t)nq_ * nblk * 64 + (dch * DB + b_) * 64 + lane];                \
                acc[nq_] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[b_].x, bq_.x, acc[nq_], 0, 0, 0);          \
                acc[nq_] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[b_].y, bq_.y, acc[nq_], 0, 0, 0);          \
                acc[nq_] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[b_].z, bq_.z, acc[nq_], 0, 0, 0);          \
                acc[nq_] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[b_].w, bq_.w, acc[nq_], 0, 0, 0);          \
            }                                                                                                \
        }                                                                                                    \
        if (++dch == ncd) {                                                                                  \
            dch = 0;                                                                                         \
            const float yv_[4] = {yn_cur.x, yn_cur.y, yn_cur.z, yn_cur.w};                                   \
            _Pragma("unroll") for (int reg_ = 0; reg_ < 4; reg_++) {                                         \
                const int idx_ = (tile << 4) + 4 * g + reg_;                                                 \
                _Pragma("unroll") for (int nq_ = 0; nq_ < NQ; nq_++) {                                       \
                    const float v_ = acc[nq_][reg_];                                                         \
                    const uint32_t o_ = l2 ? km_ord_l2(km_l2_expanded(xnj[nq_], yv_[reg_], v_)) : km_ord_ip(v_); \
                    if (idx_ < P.m && o_ < best_ord[nq_]) {                                                  \
                        best_ord[nq_] = o_;                                                                  \
                        best_idx[nq_] = idx_;                                                                \
                    }                                                                                        \
                }                                                                                            \
            }                                                                                                \
            yn_cur = yn_next;                                                                                \
            tile++;                                                                                          \
        }                                                                                                    \
    }

        KM_LOAD(a0, 0);
        yn_cur = yn_next;
        int s = 0;
        while (s < nsteps) {
            if (s + 1 < nsteps) KM_LOAD(a1, s + 1);
            KM_STEP(a0);
            s++;
            if (s >= nsteps) break;
            if (s + 1 < nsteps) KM_LOAD(a0, s + 1);
            KM_STEP(a1);
            s++;
        }
#undef KM_LOAD
#undef KM_STEP
    }
    // reduce over the 4 row groups (lanes j, j+16, j+32, j+48), then over the 4 waves; order = (ord, index)
#pragma unroll
    for (int nq = 0; nq < NQ; nq++) {
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
            const uint32_t oo = __shfl_xor(best_ord[nq], off);
            const int oi = __shfl_xor(best_idx[nq], off);
            if (oo < best_ord[nq] || (oo == best_ord[nq] && oi < best_idx[nq])) {
                best_ord[nq] = oo;
                best_idx[nq] = oi;
            }
        }
        if (g == 0) {
            red_ord[wave * NQ * 16 + nq * 16 + j] = best_ord[nq];
            red_idx[wave * NQ * 16 + nq * 16 + j] = best_idx[nq];
        }
    }
    __syncthreads();
    if (tid < NQ * 16) {
        uint32_t bo = red_ord[tid];
        int bi = red_idx[tid];
        for (int w = 1; w < 4; w++) {
            const uint32_t oo = red_ord[w * NQ * 16 + tid];
            const int oi = red_idx[w * NQ * 16 + tid];
            if (oo < bo || (oo == bo && oi < bi)) {
                bo = oo;
                bi = oi;
            }
        }
        const int64_t row = row_base + tid;
        if (row < P.n) {
            P.assign[row] = bi == 0x7FFFFFFF ? -1 : bi;
            if (P.val) P.val[row] = l2 ? __uint_as_float(bo) : km_ip_from_ord(bo);
        }
    }
}

// ---- update -------------------------------------------------------------------------------------------------
// Stable bucketing of the rows by assignment: a least-significant-digit pass per 8 bits of the centroid number (1 pass up to
// 255 centroids, 2 up to 65535, 3 beyond), each pass = digit histogram per workgroup tile, one exclusive scan over
// [digit][tile], stable scatter.  The first pass reads the assignments themselves (key = assignment, value = row number), so
// no key / value arrays are materialised before it.  Rows whose assignment is out of range get key m: they sort behind every
// centroid and are ignored.
constexpr int RS_THREADS = 256, RS_ITEMS = 16, RS_TILE = RS_THREADS * RS_ITEMS;

template <bool FIRST>
__device__ __forceinline__ int32_t rs_key(const int64_t *__restrict__ assign, const int32_t *__restrict__ keys_in, int64_t i, int m) {
    if (FIRST) {
        const int64_t a = assign[i];
        return (a < 0 || a >= m) ? m : (int32_t)a;
    }
    return keys_in[i];
}

template <bool FIRST>
__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const int64_t *__restrict__ assign, const int32_t *__restrict__ keys_in, int64_t n,
                                                        int m, int shift, int32_t *__restrict__ hist /*[256][tiles]*/, int tiles,
                                                        int32_t *__restrict__ totals /*[256], zeroed*/) {
    __shared__ int32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t i0 = (int64_t)blockIdx.x * RS_TILE + threadIdx.x;
#pragma unroll 4
    for (int it = 0; it < RS_ITEMS; it++) {
        const int64_t i = i0 + (int64_t)it * RS_THREADS;
        if (i < n) atomicAdd(&h[(rs_key<FIRST>(assign, keys_in, i, m) >> shift) & 255], 1);
    }
    __syncthreads();
    hist[(int64_t)threadIdx.x * tiles + blockIdx.x] = h[threadIdx.x];
    if (h[threadIdx.x]) atomicAdd(&totals[threadIdx.x], h[threadIdx.x]);
}

// exclusive scan over [digit][tile] in place: workgroup d owns the row of digit d -- its base is the sum of the digit totals below
// it (k_rs_hist adds them up with one atomic per digit and tile), the row itself goes through in chunks of 256 coalesced entries
__global__ __launch_bounds__(256) void k_rs_scan(int32_t *__restrict__ hist, int tiles, const int32_t *__restrict__ totals) {
    __shared__ int32_t s_wave[4];
    __shared__ int32_t s_base;
    const int d = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        int32_t v = tid < d ? totals[tid] : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) s_wave[wave] = v;
        __syncthreads();
        if (tid == 0) s_base = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
    int32_t run = s_base;
    int32_t *row = hist + (int64_t)d * tiles;
    for (int c0 = 0; c0 < tiles; c0 += 256) {
        const int i = c0 + tid;
        const int32_t v = i < tiles ? row[i] : 0;
        int32_t inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int32_t u = __shfl_up(inc, o);
            if (lane >= o) inc += u;
        }
        __syncthreads();  // (s_wave of the previous chunk has been read)
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        int32_t wpre = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            if (w < wave) wpre += s_wave[w];
            tot += s_wave[w];
        }
        if (i < tiles) row[i] = run + wpre + inc - v;
        run += tot;
    }
}

// stable scatter of one tile: rounds of 256 keys in index order; inside a round a key's place among the equal digits is (equal
// digits in lower waves) + (equal digits in lower lanes of its wave, found with 8 ballots)
template <bool FIRST>
__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(const int64_t *__restrict__ assign, const int32_t *__restrict__ keys_in,
                                                           const int32_t *__restrict__ vals_in, int64_t n, int m, int shift,
                                                           const int32_t *__restrict__ hist, int tiles, int32_t *__restrict__ keys_out,
                                                           int32_t *__restrict__ vals_out) {
    __shared__ int32_t base[256];
    __shared__ int32_t cnt[4][256];
    base[threadIdx.x] = hist[(int64_t)threadIdx.x * tiles + blockIdx.x];
#pragma unroll
    for (int w = 0; w < 4; w++) cnt[w][threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const int64_t i0 = (int64_t)blockIdx.x * RS_TILE + threadIdx.x;
    for (int it = 0; it < RS_ITEMS; it++) {
        const int64_t i = i0 + (int64_t)it * RS_THREADS;
        const bool valid = i < n;
        int32_t key = 0, val = 0;
        if (valid) {
            key = rs_key<FIRST>(assign, keys_in, i, m);
            val = FIRST ? (int32_t)i : vals_in[i];
        }
        const int digit = (key >> shift) & 255;
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const bool bit = (digit >> b) & 1;
            const uint64_t bal = __ballot(valid && bit);
            peers &= bit ? bal : ~bal;
        }
        const int rank = __popcll(peers & lt);
        if (valid && rank == 0) cnt[wave][digit] = __popcll(peers);
        __syncthreads();
        if (valid) {
            int at = base[digit] + rank;
            for (int w = 0; w < wave; w++) at += cnt[w][digit];
            keys_out[at] = key;
            vals_out[at] = val;
        }
        __syncthreads();
        {
            const int t = threadIdx.x;
            base[t] += cnt[0][t] + cnt[1][t] + cnt[2][t] + cnt[3][t];
            cnt[0][t] = cnt[1][t] = cnt[2][t] = cnt[3][t] = 0;
        }
        __syncthreads();
    }
}

__global__ void k_segment_bounds(const int32_t *__restrict__ sorted_keys, int64_t n, int m, int64_t *seg_begin /*[m+1]*/) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    // seg_begin[c] = first position with key >= c
    int prev = i == 0 ? -1 : sorted_keys[i - 1];
    int cur = i == n ? m : min(sorted_keys[i], m);
    for (int c = prev + 1; c <= cur; c++) seg_begin[c] = i;
}

// one workgroup per centroid; thread t owns dimensions t, t + blockDim, ...; rows are added in ascending row order (one dependent
// chain of adds per (centroid, dimension): the order of the reference loop).  The kernel lasts as long as its largest cluster's
// chain, so what counts is the time per row of ONE workgroup: 32 rows per round trip, the row numbers of the next 32 requested with
// them (8 rows per trip, row numbers fetched first: 0.60 ms for 2^20 rows in 4096 skewed clusters; now 0.39).
__global__ __launch_bounds__(256) void k_accumulate(const float *__restrict__ x, int d, const int32_t *__restrict__ sorted_rows,
                                                    const int64_t *__restrict__ seg_begin, float *__restrict__ sums,
                                                    int64_t *__restrict__ counts) {
    constexpr int U = 32;
    const int c = blockIdx.x;
    const int64_t b = seg_begin[c], e = seg_begin[c + 1];
    if (threadIdx.x == 0) counts[c] = e - b;
    if (e == b) {
        for (int k = threadIdx.x; k < d; k += blockDim.x) sums[(int64_t)c * d + k] = 0.0f;
        return;
    }
    for (int k = threadIdx.x; k < d; k += blockDim.x) {
        float s = 0.0f;
        // batches of U rows: the row numbers of batch t + 1 are requested with the values of batch t (a short last batch repeats the
        // last row's address and skips its adds: adding +0 instead would turn a sum of -0 into +0)
        int32_t r[U], rn[U];
#pragma unroll
        for (int j = 0; j < U; j++) r[j] = sorted_rows[min(b + j, e - 1)];
        for (int64_t i = b; i < e; i += U) {
            const int cnt = (int)min((int64_t)U, e - i);
            float v[U];
#pragma unroll
            for (int j = 0; j < U; j++) v[j] = x[(int64_t)r[j] * d + k];
#pragma unroll
            for (int j = 0; j < U; j++) rn[j] = sorted_rows[min(i + U + j, e - 1)];
            if (cnt == U) {
#pragma unroll
                for (int j = 0; j < U; j++) s += v[j];
            } else {
#pragma unroll
                for (int j = 0; j < U; j++) s = j < cnt ? s + v[j] : s;
            }
#pragma unroll
            for (int j = 0; j < U; j++) r[j] = rn[j];
        }
        sums[(int64_t)c * d + k] = s;
    }
}

__global__ void k_finalize_centroids(const float *__restrict__ sums, const int64_t *__restrict__ counts, int64_t m, int d,
                                     int keep_empty, float *__restrict__ c) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= m * d) return;
    int64_t jn = idx / d;
    int64_t cnt = counts[jn];
    if (cnt == 0 && keep_empty) return;
    c[idx] = sums[idx] / (float)cnt;
}

__global__ void k_normalize_rows(float *x, int64_t n, int d) {
    // vectors / vectors.norm(2,1) (clustering.cpp:25-26): canonical norm = sqrt of the fmaf chain
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float *s = x + i * d;
    float acc = 0.0f;
    for (int k = 0; k < d; k++) acc = __fmaf_rn(s[k], s[k], acc);
    const float nn = sqrtf(acc);
    for (int k = 0; k < d; k++) s[k] = s[k] / nn;
}

__global__ void k_gather_rows(const float *__restrict__ x, int d, const int64_t *__restrict__ rows, int64_t n, float *__restrict__ out) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * d) return;
    int64_t i = idx / d;
    out[idx] = x[rows[i] * d + (idx - i * d)];
}

__global__ void k_gather_rows_i32(const float *__restrict__ x, int d, const int32_t *__restrict__ rows, int64_t n,
                                  float *__restrict__ out) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * d) return;
    int64_t i = idx / d;
    out[idx] = x[(int64_t)rows[i] * d + (idx - i * d)];
}

__global__ void k_gather_ids_i32(const int64_t *__restrict__ ids, const int32_t *__restrict__ rows, int64_t n,
                                 int64_t *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = ids[rows[i]];
}

// ---- host side ----------------------------------------------------------------------------------------------
static inline unsigned km_grid(int64_t n, int b) { return (unsigned)((n + b - 1) / b); }

struct KmScratch {  // device buffers owned by one qk_kmeans* call
    std::vector<void *> ptrs;
    ~KmScratch() {
        for (void *p : ptrs)
            if (p) (void)hipFree(p);
    }
    template <typename T>
    int alloc(T **out, size_t count) {
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T));
        if (e != hipSuccess) {
            qk_set_error("k-means scratch allocation of %zu bytes failed: %s", count * sizeof(T), hipGetErrorString(e));
            return QK_ERR_OOM;
        }
        ptrs.push_back(p);
        *out = (T *)p;
        return QK_OK;
    }
};

template <int DB, int NQ, bool L2>
static int launch_assign_m(hipStream_t st, unsigned grid, size_t lds, const AssignParams &ap) {
    QK_HIP(hipFuncSetAttribute((const void *)k_assign<DB, NQ, L2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_assign<DB, NQ, L2>), dim3(grid), dim3(256), lds, st, ap);
    return QK_OK;
}
template <int DB, int NQ>
static int launch_assign_t(hipStream_t st, unsigned grid, size_t lds, const AssignParams &ap) {
    return ap.metric == QK_METRIC_L2 ? launch_assign_m<DB, NQ, true>(st, grid, lds, ap) : launch_assign_m<DB, NQ, false>(st, grid, lds, ap);
}

// assign on device pointers; ctile/cnorm are scratch for the tile-major centroid copy (mt*16 rows)
static int assign_device(qk_ctx *ctx, const float *x, int64_t n, const float *c, int64_t m, int d, int metric, int64_t *assign,
                         float *val, float *ctile, float *cnorm) {
    if (n <= 0) return QK_OK;
    const int dpad = qk_round_up(d, 16), nblk = dpad / 16;
    const int64_t mt = (m + 15) / 16;
    QK_HIP(hipMemsetAsync(ctile, 0, (size_t)mt * 16 * dpad * sizeof(float), ctx->stream));
    QK_HIP(hipMemsetAsync(cnorm, 0, (size_t)mt * 16 * sizeof(float), ctx->stream));
    QK_TRY(qk_launch_ingest(ctx, c, nullptr, m, d, nblk, ctile, cnorm, nullptr, 0));
    AssignParams ap;
    ap.x = x;
    ap.n = n;
    ap.d = d;
    ap.nblk = nblk;
    ap.cvecs = (const float4 *)ctile;
    ap.cnorms = cnorm;
    ap.m = (int)m;
    ap.metric = metric;
    ap.assign = assign;
    ap.val = val;
    const int DB = (nblk % 8 == 0) ? 8 : (nblk % 4 == 0) ? 4 : (nblk % 2 == 0) ? 2 : 1;
    // query tiles per workgroup: as many as fit 64 KiB of LDS (more tiles = fewer passes over the centroids)
    int NQ = 4;
    while (NQ > 1 && (size_t)NQ * nblk * 1024 > 64 * 1024) NQ >>= 1;
    if (n <= 16) NQ = 1;
    const size_t lds = (size_t)NQ * nblk * 1024 + (size_t)NQ * 16 * 4 + (size_t)4 * NQ * 16 * 8 + 64;
    if (lds > 160 * 1024) QK_FAIL(QK_ERR_UNSUPPORTED, "qk_kmeans_assign: d=%d too large for the LDS query tile", d);
    const unsigned grid = km_grid(n, NQ * 16);
#define KM_CASE(D, N) \
    if (DB == D && NQ == N) QK_TRY((launch_assign_t<D, N>(ctx->stream, grid, lds, ap)));
    KM_CASE(8, 4) KM_CASE(8, 2) KM_CASE(8, 1) KM_CASE(4, 4) KM_CASE(4, 2) KM_CASE(4, 1)
    KM_CASE(2, 4) KM_CASE(2, 2) KM_CASE(2, 1) KM_CASE(1, 4) KM_CASE(1, 2) KM_CASE(1, 1)
#undef KM_CASE
    QK_HIP(hipGetLastError());
    return QK_OK;
}

struct AccumScratch {
    int32_t *keys = nullptr, *vals = nullptr, *keys2 = nullptr, *vals2 = nullptr;
    int64_t *seg = nullptr;
    void *tmp = nullptr;
    size_t tmp_bytes = 0;
};

static int accum_prepare(KmScratch &ks, AccumScratch &as, int64_t n, int64_t m) {
    QK_TRY(ks.alloc(&as.keys, (size_t)n));
    QK_TRY(ks.alloc(&as.vals, (size_t)n));
    QK_TRY(ks.alloc(&as.keys2, (size_t)n));
    QK_TRY(ks.alloc(&as.vals2, (size_t)n));
    QK_TRY(ks.alloc(&as.seg, (size_t)m + 2));
    const size_t tiles = (size_t)((n + RS_TILE - 1) / RS_TILE);
    int32_t *t = nullptr;
    QK_TRY(ks.alloc(&t, 256 * tiles + 256 + 64));  // [digit][tile] counters of one pass, then the 256 digit totals
    as.tmp = t;
    as.tmp_bytes = (256 * tiles + 256 + 64) * sizeof(int32_t);
    return QK_OK;
}

// rows bucketed stably by assignment into (as.keys2, as.vals2): ceil(bits(m) / 8) passes of histogram / scan / scatter
static int bucket_rows_device(hipStream_t st, AccumScratch &as, const int64_t *assign, int64_t n, int64_t m) {
    int bits = 1;
    while ((1LL << bits) <= m) bits++;
    const int passes = (bits + 7) / 8;
    const int tiles = (int)((n + RS_TILE - 1) / RS_TILE);
    int32_t *hist = (int32_t *)as.tmp;
    // the last pass must land in (keys2, vals2): with an odd number of passes the first one writes there
    int32_t *ko = (passes & 1) ? as.keys2 : as.keys, *vo = (passes & 1) ? as.vals2 : as.vals;
    const int32_t *ki = nullptr, *vi = nullptr;
    int32_t *totals = hist + (size_t)256 * tiles;
    for (int p = 0; p < passes; p++) {
        const int shift = 8 * p;
        QK_HIP(hipMemsetAsync(totals, 0, 256 * sizeof(int32_t), st));
        if (p == 0)
            hipLaunchKernelGGL(k_rs_hist<true>, dim3((unsigned)tiles), dim3(RS_THREADS), 0, st, assign, ki, n, (int)m, shift, hist, tiles, totals);
        else
            hipLaunchKernelGGL(k_rs_hist<false>, dim3((unsigned)tiles), dim3(RS_THREADS), 0, st, assign, ki, n, (int)m, shift, hist, tiles, totals);
        hipLaunchKernelGGL(k_rs_scan, dim3(256), dim3(256), 0, st, hist, tiles, totals);
        if (p == 0)
            hipLaunchKernelGGL(k_rs_scatter<true>, dim3((unsigned)tiles), dim3(RS_THREADS), 0, st, assign, ki, vi, n, (int)m, shift, hist, tiles, ko, vo);
        else
            hipLaunchKernelGGL(k_rs_scatter<false>, dim3((unsigned)tiles), dim3(RS_THREADS), 0, st, assign, ki, vi, n, (int)m, shift, hist, tiles, ko, vo);
        ki = ko;
        vi = vo;
        ko = ko == as.keys ? as.keys2 : as.keys;
        vo = vo == as.vals ? as.vals2 : as.vals;
    }
    QK_HIP(hipGetLastError());
    return QK_OK;
}

static int accumulate_device(qk_ctx *ctx, AccumScratch &as, const float *x, int64_t n, int d, const int64_t *assign, int64_t m,
                             float *sums, int64_t *counts) {
    hipStream_t st = ctx->stream;
    if (n > 0x7FFFFFF0LL) QK_FAIL(QK_ERR_UNSUPPORTED, "qk_kmeans_accumulate: n too large for 32-bit row indices");
    if (m > 0x7FFFFFF0LL) QK_FAIL(QK_ERR_UNSUPPORTED, "qk_kmeans_accumulate: too many centroids for 32-bit keys");
    if (n > 0) QK_TRY(bucket_rows_device(st, as, assign, n, m));
    hipLaunchKernelGGL(k_segment_bounds, dim3(km_grid(n + 1, 256)), dim3(256), 0, st, as.keys2, n, (int)m, as.seg);
    hipLaunchKernelGGL(k_accumulate, dim3((unsigned)m), dim3((unsigned)std::min(256, qk_round_up(d, 64))), 0, st, x, d, as.vals2, as.seg, sums, counts);
    QK_HIP(hipGetLastError());
    return QK_OK;
}

// faiss split_clusters restated deterministically -- identical to oracle split_empty()
static int split_empty_host(float *c, int64_t *counts, int64_t m, int d) {
    const float EPS = 1.0f / 1024.0f;
    int nsplit = 0;
    for (int64_t ci = 0; ci < m; ci++) {
        if (counts[ci] != 0) continue;
        int64_t cj = 0;
        for (int64_t j = 1; j < m; j++)
            if (counts[j] > counts[cj]) cj = j;
        if (counts[cj] < 2) continue;
        memcpy(c + ci * d, c + cj * d, sizeof(float) * (size_t)d);
        for (int k = 0; k < d; k++) {
            if (k % 2 == 0) {
                c[ci * d + k] *= 1 + EPS;
                c[cj * d + k] *= 1 - EPS;
            } else {
                c[ci * d + k] *= 1 - EPS;
                c[cj * d + k] *= 1 + EPS;
            }
        }
        counts[ci] = counts[cj] / 2;
        counts[cj] -= counts[ci];
        nsplit++;
    }
    return nsplit;
}

static inline uint64_t splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static void rand_perm_prefix(int64_t n, int64_t mcount, uint64_t seed, std::vector<int64_t> &out) {
    std::vector<int64_t> p((size_t)n);
    for (int64_t i = 0; i < n; i++) p[i] = i;
    uint64_t s = seed;
    for (int64_t i = 0; i < mcount && i < n - 1; i++) {
        uint64_t r = splitmix64(&s);
        int64_t j = i + (int64_t)(r % (uint64_t)(n - i));
        std::swap(p[i], p[j]);
    }
    out.assign(p.begin(), p.begin() + std::min(n, mcount));
}

extern "C" {

int qk_kmeans_assign(qk_ctx *ctx, const float *x, int64_t n, const float *c, int64_t m, int d, int metric, int64_t *assign,
                     float *val, int mem) {
    if (!ctx || !x || !c || !assign) QK_FAIL(QK_ERR_INVALID, "qk_kmeans_assign: null argument");
    if (n < 0 || m <= 0 || d <= 0) QK_FAIL(QK_ERR_INVALID, "qk_kmeans_assign: bad sizes");
    if (metric != QK_METRIC_L2 && metric != QK_METRIC_IP) QK_FAIL(QK_ERR_INVALID, "Metric type not supported");
    if (n == 0) return QK_OK;
    QK_HIP(hipSetDevice(ctx->device));
    KmScratch ks;
    const int dpad = qk_round_up(d, 16);
    const int64_t mt16 = ((m + 15) / 16) * 16;
    float *ctile, *cnorm;
    QK_TRY(ks.alloc(&ctile, (size_t)mt16 * dpad));
    QK_TRY(ks.alloc(&cnorm, (size_t)mt16));
    const float *dx = x, *dc = c;
    int64_t *da = assign;
    float *dv = val;
    float *bx = nullptr, *bc = nullptr, *bv = nullptr;
    int64_t *ba = nullptr;
    if (mem == QK_MEM_HOST) {
        QK_TRY(ks.alloc(&bx, (size_t)n * d));
        QK_TRY(ks.alloc(&bc, (size_t)m * d));
        QK_TRY(ks.alloc(&ba, (size_t)n));
        if (val) QK_TRY(ks.alloc(&bv, (size_t)n));
        QK_HIP(hipMemcpyAsync(bx, x, (size_t)n * d * 4, hipMemcpyHostToDevice, ctx->stream));
        QK_HIP(hipMemcpyAsync(bc, c, (size_t)m * d * 4, hipMemcpyHostToDevice, ctx->stream));
        dx = bx;
        dc = bc;
        da = ba;
        dv = bv;
    }
    QK_TRY(assign_device(ctx, dx, n, dc, m, d, metric, da, dv, ctile, cnorm));
    if (mem == QK_MEM_HOST) {
        QK_HIP(hipMemcpyAsync(assign, ba, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (val) QK_HIP(hipMemcpyAsync(val, bv, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    QK_HIP(hipStreamSynchronize(ctx->stream));  // scratch is freed on return
    return QK_OK;
}

int qk_kmeans_accumulate(qk_ctx *ctx, const float *x, int64_t n, int d, const int64_t *assign, int64_t m, float *sums,
                         int64_t *counts, int mem) {
    if (!ctx || !x || !assign || !sums || !counts) QK_FAIL(QK_ERR_INVALID, "qk_kmeans_accumulate: null argument");
    if (n < 0 || m <= 0 || d <= 0) QK_FAIL(QK_ERR_INVALID, "qk_kmeans_accumulate: bad sizes");
    QK_HIP(hipSetDevice(ctx->device));
    KmScratch ks;
    AccumScratch as;
    QK_TRY(accum_prepare(ks, as, std::max<int64_t>(n, 1), m));
    const float *dx = x;
    const int64_t *da = assign;
    float *ds = sums;
    int64_t *dc = counts;
    float *bx = nullptr, *bs = nullptr;
    int64_t *ba = nullptr, *bc = nullptr;
    if (mem == QK_MEM_HOST) {
        QK_TRY(ks.alloc(&bx, (size_t)n * d));
        QK_TRY(ks.alloc(&ba, (size_t)n));
        QK_TRY(ks.alloc(&bs, (size_t)m * d));
        QK_TRY(ks.alloc(&bc, (size_t)m));
        QK_HIP(hipMemcpyAsync(bx, x, (size_t)n * d * 4, hipMemcpyHostToDevice, ctx->stream));
        QK_HIP(hipMemcpyAsync(ba, assign, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
        dx = bx;
        da = ba;
        ds = bs;
        dc = bc;
    }
    QK_TRY(accumulate_device(ctx, as, dx, n, d, da, m, ds, dc));
    if (mem == QK_MEM_HOST) {
        QK_HIP(hipMemcpyAsync(sums, bs, (size_t)m * d * 4, hipMemcpyDeviceToHost, ctx->stream));
        QK_HIP(hipMemcpyAsync(counts, bc, (size_t)m * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    QK_HIP(hipStreamSynchronize(ctx->stream));
    return QK_OK;
}

// kmeans_refine_partitions (clustering.cpp:99-182) + the partition replacement of PartitionManager::refine_partitions
// (partition_manager.cpp:446-487), on the device store.  list_nos [m] (host) name the partitions, centroids [m][d]
// (in `mem`) are their centroids in the same order; on return they hold "the centroids used for the last
// assignment" (:178) and list list_nos[c] holds the vectors assigned to centroid c, in append order (:174).
int qk_store_refine_lists(qk_store *s, const int64_t *list_nos, int64_t m, float *centroids, int metric,
                          int refinement_iterations, int mem) {
    if (!s || !list_nos || !centroids || m <= 0) QK_FAIL(QK_ERR_INVALID, "qk_store_refine_lists: bad arguments");
    if (metric != QK_METRIC_L2 && metric != QK_METRIC_IP) QK_FAIL(QK_ERR_INVALID, "Metric type not supported");
    qk_ctx *ctx = s->ctx;
    QK_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int d = s->d;
    int64_t total = 0;
    for (int64_t c = 0; c < m; c++) {
        int64_t p = list_nos[c];
        if (p < 0 || p >= (int64_t)s->parts.size() || !s->parts[p].present)
            QK_FAIL(QK_ERR_NOT_FOUND, "List does not exist in refine_partitions (list %lld)", (long long)p);
        for (int64_t c2 = 0; c2 < c; c2++)
            if (list_nos[c2] == p) QK_FAIL(QK_ERR_INVALID, "qk_store_refine_lists: duplicate list %lld", (long long)p);
        total += s->parts[p].size;
    }
    const int iterations = refinement_iterations > 0 ? refinement_iterations : 1;  // clustering.cpp:110
    KmScratch ks;
    float *xa, *xb, *dc, *dsums, *ctile, *cnorm;
    int64_t *ia, *ib, *dassign, *dcounts;
    const int dpad = qk_round_up(d, 16);
    const int64_t mt16 = ((m + 15) / 16) * 16;
    QK_TRY(ks.alloc(&xa, (size_t)total * d));
    QK_TRY(ks.alloc(&xb, (size_t)total * d));
    QK_TRY(ks.alloc(&ia, (size_t)total));
    QK_TRY(ks.alloc(&ib, (size_t)total));
    QK_TRY(ks.alloc(&dassign, (size_t)total));
    QK_TRY(ks.alloc(&dc, (size_t)m * d));
    QK_TRY(ks.alloc(&dsums, (size_t)m * d));
    QK_TRY(ks.alloc(&dcounts, (size_t)m));
    QK_TRY(ks.alloc(&ctile, (size_t)mt16 * dpad));
    QK_TRY(ks.alloc(&cnorm, (size_t)mt16));
    QK_HIP(hipMemcpyAsync(dc, centroids, (size_t)m * d * 4, mem == QK_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, st));
    // concatenate the partitions in the given order (row order inside a partition = arena order)
    int64_t pos = 0;
    for (int64_t c = 0; c < m; c++) {
        const qk_part &pt = s->parts[list_nos[c]];
        if (pt.size == 0) continue;
        QK_TRY(qk_launch_extract(ctx, s->vecs, s->nblk, d, pt.row_off, nullptr, pt.size, xa + pos * d));
        QK_HIP(hipMemcpyAsync(ia + pos, s->ids + pt.row_off, (size_t)pt.size * 8, hipMemcpyDeviceToDevice, st));
        pos += pt.size;
    }
    AccumScratch as;
    QK_TRY(accum_prepare(ks, as, std::max<int64_t>(total, 1), m));
    std::vector<int64_t> hcounts((size_t)m, 0);
    for (int iter = 0; iter < iterations; iter++) {
        if (iter > 0)  // centroids = sums / counts; a count of 0 gives NaN exactly like the reference (:122-124)
            hipLaunchKernelGGL(k_finalize_centroids, dim3(km_grid(m * d, 256)), dim3(256), 0, st, dsums, dcounts, m, d, 0, dc);
        QK_TRY(assign_device(ctx, xa, total, dc, m, d, metric, dassign, nullptr, ctile, cnorm));
        QK_TRY(accumulate_device(ctx, as, xa, total, d, dassign, m, dsums, dcounts));
        // stable bucket by assignment == the per-vector append into the new partitions (:174); accumulate_device left
        // the stably sorted row list in as.vals2
        if (total > 0) {
            hipLaunchKernelGGL(k_gather_rows_i32, dim3(km_grid(total * d, 256)), dim3(256), 0, st, xa, d, as.vals2, total, xb);
            hipLaunchKernelGGL(k_gather_ids_i32, dim3(km_grid(total, 256)), dim3(256), 0, st, ia, as.vals2, total, ib);
        }
        std::swap(xa, xb);
        std::swap(ia, ib);
    }
    QK_HIP(hipMemcpyAsync(hcounts.data(), dcounts, (size_t)m * 8, hipMemcpyDeviceToHost, st));
    QK_HIP(hipMemcpyAsync(centroids, dc, (size_t)m * d * 4, mem == QK_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, st));
    QK_HIP(hipStreamSynchronize(st));
    int64_t assigned = 0;
    for (int64_t c = 0; c < m; c++) assigned += hcounts[c];
    if (assigned != total)
        QK_FAIL(QK_ERR_INVALID, "qk_store_refine_lists: %lld of %lld vectors could not be assigned (NaN centroid from an emptied cluster)",
                (long long)(total - assigned), (long long)total);
    // replace the partitions (partition_manager.cpp:481-483)
    pos = 0;
    for (int64_t c = 0; c < m; c++) {
        QK_TRY(qk_store_remove_list(s, list_nos[c]));
        QK_TRY(qk_store_add_list(s, list_nos[c]));
        if (hcounts[c] > 0) QK_TRY(qk_store_add_entries(s, list_nos[c], hcounts[c], ia + pos, xa + pos * d, QK_MEM_DEVICE));
        pos += hcounts[c];
    }
    return QK_OK;
}

int qk_kmeans(qk_ctx *ctx, float *x, int64_t n, int d, int64_t m, int metric, int niter, uint64_t seed, float *centroids,
              int64_t *assign, int mem) {
    if (!ctx || !x || !centroids || !assign) QK_FAIL(QK_ERR_INVALID, "qk_kmeans: null argument");
    if (n <= 0 || m <= 0 || d <= 0 || m > n) QK_FAIL(QK_ERR_INVALID, "qk_kmeans: bad sizes (n=%lld m=%lld d=%d)", (long long)n, (long long)m, d);
    if (metric != QK_METRIC_L2 && metric != QK_METRIC_IP) QK_FAIL(QK_ERR_INVALID, "Metric type not supported");
    QK_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    KmScratch ks;
    float *dx = x;
    if (mem == QK_MEM_HOST) {
        QK_TRY(ks.alloc(&dx, (size_t)n * d));
        QK_HIP(hipMemcpyAsync(dx, x, (size_t)n * d * 4, hipMemcpyHostToDevice, st));
    }
    if (metric == QK_METRIC_IP) hipLaunchKernelGGL(k_normalize_rows, dim3(km_grid(n, 256)), dim3(256), 0, st, dx, n, d);
    // subsample + init (oracle qo_kmeans: identical permutation)
    const int64_t max_pts = 256;
    const bool sub = n > max_pts * m;
    const int64_t ntrain = sub ? max_pts * m : n;
    std::vector<int64_t> perm;
    rand_perm_prefix(n, sub ? ntrain : m, seed, perm);
    int64_t *dperm;
    QK_TRY(ks.alloc(&dperm, perm.size()));
    QK_HIP(hipMemcpyAsync(dperm, perm.data(), perm.size() * 8, hipMemcpyHostToDevice, st));
    float *xt = dx;
    if (sub) {
        QK_TRY(ks.alloc(&xt, (size_t)ntrain * d));
        hipLaunchKernelGGL(k_gather_rows, dim3(km_grid(ntrain * d, 256)), dim3(256), 0, st, dx, d, dperm, ntrain, xt);
    }
    float *dc, *dsums, *ctile, *cnorm;
    int64_t *dcounts, *dta, *dassign = assign;
    const int dpad = qk_round_up(d, 16);
    const int64_t mt16 = ((m + 15) / 16) * 16;
    QK_TRY(ks.alloc(&dc, (size_t)m * d));
    QK_TRY(ks.alloc(&dsums, (size_t)m * d));
    QK_TRY(ks.alloc(&dcounts, (size_t)m));
    QK_TRY(ks.alloc(&ctile, (size_t)mt16 * dpad));
    QK_TRY(ks.alloc(&cnorm, (size_t)mt16));
    QK_TRY(ks.alloc(&dta, (size_t)ntrain));
    if (mem == QK_MEM_HOST) QK_TRY(ks.alloc(&dassign, (size_t)n));
    // centroids = first m rows of the permutation (of the subsample if any)
    hipLaunchKernelGGL(k_gather_rows, dim3(km_grid(m * d, 256)), dim3(256), 0, st, dx, d, dperm, m, dc);
    AccumScratch as;
    QK_TRY(accum_prepare(ks, as, ntrain, m));
    std::vector<int64_t> hcounts((size_t)m);
    std::vector<float> hc;
    struct Ev3 {  // assign start / update start / update end of the last iteration (qk_kmeans_last_timing)
        hipEvent_t e[3] = {nullptr, nullptr, nullptr};
        ~Ev3() {
            for (hipEvent_t v : e)
                if (v) (void)hipEventDestroy(v);
        }
    } ev;
    for (int it = 0; it < niter; it++) {
        const bool timed = it == niter - 1;
        if (timed) {
            for (int j = 0; j < 3; j++) QK_HIP(hipEventCreate(&ev.e[j]));
            QK_HIP(hipEventRecord(ev.e[0], st));
        }
        QK_TRY(assign_device(ctx, xt, ntrain, dc, m, d, metric, dta, nullptr, ctile, cnorm));
        if (timed) QK_HIP(hipEventRecord(ev.e[1], st));
        QK_TRY(accumulate_device(ctx, as, xt, ntrain, d, dta, m, dsums, dcounts));
        if (timed) QK_HIP(hipEventRecord(ev.e[2], st));
        hipLaunchKernelGGL(k_finalize_centroids, dim3(km_grid(m * d, 256)), dim3(256), 0, st, dsums, dcounts, m, d, 1, dc);
        QK_HIP(hipMemcpyAsync(hcounts.data(), dcounts, (size_t)m * 8, hipMemcpyDeviceToHost, st));
        QK_HIP(hipStreamSynchronize(st));
        if (timed) {
            QK_HIP(hipEventElapsedTime(&ctx->km_assign_ms, ev.e[0], ev.e[1]));
            QK_HIP(hipEventElapsedTime(&ctx->km_update_ms, ev.e[1], ev.e[2]));
            ctx->km_rows = ntrain;
            ctx->km_m = m;
        }
        bool any_empty = false;
        for (int64_t j = 0; j < m; j++)
            if (hcounts[j] == 0) {
                any_empty = true;
                break;
            }
        if (any_empty) {
            hc.resize((size_t)m * d);
            QK_HIP(hipMemcpy(hc.data(), dc, (size_t)m * d * 4, hipMemcpyDeviceToHost));
            split_empty_host(hc.data(), hcounts.data(), m, d);
            QK_HIP(hipMemcpy(dc, hc.data(), (size_t)m * d * 4, hipMemcpyHostToDevice));
        }
    }
    if (metric == QK_METRIC_IP) hipLaunchKernelGGL(k_normalize_rows, dim3(km_grid(m, 256)), dim3(256), 0, st, dc, m, d);
    QK_TRY(assign_device(ctx, dx, n, dc, m, d, metric, dassign, nullptr, ctile, cnorm));
    if (mem == QK_MEM_HOST) {
        QK_HIP(hipMemcpyAsync(centroids, dc, (size_t)m * d * 4, hipMemcpyDeviceToHost, st));
        QK_HIP(hipMemcpyAsync(assign, dassign, (size_t)n * 8, hipMemcpyDeviceToHost, st));
        if (metric == QK_METRIC_IP) QK_HIP(hipMemcpyAsync(x, dx, (size_t)n * d * 4, hipMemcpyDeviceToHost, st));
    } else {
        QK_HIP(hipMemcpyAsync(centroids, dc, (size_t)m * d * 4, hipMemcpyDeviceToDevice, st));
    }
    QK_HIP(hipGetLastError());
    QK_HIP(hipStreamSynchronize(st));
    return QK_OK;
}

int qk_normalize_rows(qk_ctx *ctx, float *x, int64_t n, int d, int mem) {
    if (!ctx || !x) QK_FAIL(QK_ERR_INVALID, "qk_normalize_rows: null argument");
    if (n < 0 || d <= 0) QK_FAIL(QK_ERR_INVALID, "qk_normalize_rows: bad sizes");
    if (n == 0) return QK_OK;
    QK_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    KmScratch ks;
    float *dx = x;
    if (mem == QK_MEM_HOST) {
        QK_TRY(ks.alloc(&dx, (size_t)n * d));
        QK_HIP(hipMemcpyAsync(dx, x, (size_t)n * d * 4, hipMemcpyHostToDevice, st));
    }
    hipLaunchKernelGGL(k_normalize_rows, dim3(km_grid(n, 256)), dim3(256), 0, st, dx, n, d);
    QK_HIP(hipGetLastError());
    if (mem == QK_MEM_HOST) {
        QK_HIP(hipMemcpyAsync(x, dx, (size_t)n * d * 4, hipMemcpyDeviceToHost, st));
        QK_HIP(hipStreamSynchronize(st));  // scratch is freed on return
    }
    return QK_OK;
}

int qk_kmeans_update(qk_ctx *ctx, const float *sums, int64_t *counts, int64_t m, int d, float *centroids, int mem) {
    if (!ctx || !sums || !counts || !centroids) QK_FAIL(QK_ERR_INVALID, "qk_kmeans_update: null argument");
    if (m <= 0 || d <= 0) QK_FAIL(QK_ERR_INVALID, "qk_kmeans_update: bad sizes");
    QK_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    std::vector<int64_t> hcounts((size_t)m);
    if (mem == QK_MEM_HOST) {
        // host data: the whole update is m*d divisions -- same expressions as the kernel
        for (int64_t j = 0; j < m; j++) {
            if (counts[j] == 0) continue;
            const float cnt = (float)counts[j];
            for (int k = 0; k < d; k++) centroids[j * d + k] = sums[j * d + k] / cnt;
        }
        split_empty_host(centroids, counts, m, d);
        return QK_OK;
    }
    hipLaunchKernelGGL(k_finalize_centroids, dim3(km_grid(m * d, 256)), dim3(256), 0, st, sums, counts, m, d, 1, centroids);
    QK_HIP(hipGetLastError());
    QK_HIP(hipMemcpyAsync(hcounts.data(), counts, (size_t)m * 8, hipMemcpyDeviceToHost, st));
    QK_HIP(hipStreamSynchronize(st));
    bool any_empty = false;
    for (int64_t j = 0; j < m && !any_empty; j++) any_empty = hcounts[j] == 0;
    if (any_empty) {
        std::vector<float> hc((size_t)m * d);
        QK_HIP(hipMemcpyAsync(hc.data(), centroids, (size_t)m * d * 4, hipMemcpyDeviceToHost, st));
        QK_HIP(hipStreamSynchronize(st));
        split_empty_host(hc.data(), hcounts.data(), m, d);
        QK_HIP(hipMemcpyAsync(centroids, hc.data(), (size_t)m * d * 4, hipMemcpyHostToDevice, st));
        QK_HIP(hipMemcpyAsync(counts, hcounts.data(), (size_t)m * 8, hipMemcpyHostToDevice, st));
        QK_HIP(hipStreamSynchronize(st));
    }
    return QK_OK;
}

int qk_kmeans_last_timing(qk_ctx *ctx, float *assign_ms, float *update_ms, int64_t *rows, int64_t *m) {
    if (!ctx || !assign_ms || !update_ms || !rows || !m) QK_FAIL(QK_ERR_INVALID, "qk_kmeans_last_timing: null argument");
    *assign_ms = ctx->km_assign_ms;
    *update_ms = ctx->km_update_ms;
    *rows = ctx->km_rows;
    *m = ctx->km_m;
    return QK_OK;
}

int qk_rand_perm(int64_t n, int64_t m, uint64_t seed, int64_t *perm_out_host) {
    if (n < 0 || m < 0 || !perm_out_host) QK_FAIL(QK_ERR_INVALID, "qk_rand_perm: bad arguments");
    std::vector<int64_t> p;
    rand_perm_prefix(n, m, seed, p);
    memcpy(perm_out_host, p.data(), p.size() * sizeof(int64_t));
    return QK_OK;
}

}  // extern "C"
