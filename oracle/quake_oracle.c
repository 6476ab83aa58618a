/*
 * quake_oracle.c -- CPU restatement of the Quake search / k-means hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under quake_amd/ may import, link or call this file.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker /
 * the timed CPU port -- never as the product path.
 *
 * What it restates (citations are relative to /root/reference):
 *   - TypedTopKBuffer<float,int64_t>        src/cpp/include/list_scanning.h:41-204
 *   - scan_list (+4 specialisations)         src/cpp/include/list_scanning.h:241-311
 *   - batched_scan_list                      src/cpp/include/list_scanning.h:313-366
 *   - QueryCoordinator::serial_scan          src/cpp/src/query_coordinator.cpp:471-611 (fixed nprobe, and the use_aps branch: qo_search_aps)
 *   - APS geometry                           src/cpp/include/geometry.h:57-211,247-295,345-407
 *   - QueryCoordinator::batched_serial_scan  src/cpp/src/query_coordinator.cpp:675-799
 *   - QueryCoordinator::search               src/cpp/src/query_coordinator.cpp:612-657 (coarse = parent batched scan)
 *   - kmeans                                 src/cpp/src/clustering.cpp:13-97
 *   - kmeans_refine_partitions               src/cpp/src/clustering.cpp:99-182
 *   - parallel_for                           src/cpp/include/parallel.h:41-63 (static chunking over threads)
 *
 * Third-party arithmetic.  The distance primitives live in facebookresearch/faiss, pinned by the
 * reference only as an un-populated git submodule (src/cpp/third_party/faiss is an empty directory;
 * gitlink SHA not recoverable; "FAISS main at or before 2025-05-23").  The published semantics are
 * restated here:  fvec_inner_product = sum x_i*y_i,  fvec_L2sqr = sum (x_i-y_i)^2,
 * knn_L2sqr(batched) = ||x||^2 + ||y||^2 - 2 x.y clamped at 0, ascending, returns SQUARED distances,
 * knn_inner_product descending.  FAISS leaves the float summation order to its SIMD/BLAS back end;
 * this restatement FIXES it (the "canonical arithmetic"):
 *     every dot product is ONE k-ordered fp32 fmaf chain  acc = fmaf(x[k], y[k], acc), k = 0..d-1, acc0 = +0
 * which is bit-for-bit what gfx950's v_mfma_f32_16x16x4_f32 computes.
 *
 * Tie rule.  The reference's comparators look at the distance only (list_scanning.h:154-171), so the
 * order of equal distances is unspecified (its own tests accept either id, test/cpp/list_scanning.cpp:52-54).
 * This restatement uses the total order (key, id): L2 ascending squared distance then ascending id;
 * IP descending inner product then ascending id.  L2 results are selected on the SQUARED distance and
 * sqrt is applied to what is returned -- a refinement of the reference's order on sqrt(d2) (sqrt is monotone).
 *
 * Pinning.  The reference cannot be built here (FAISS absent), so this oracle is pinned against
 * (a) the hand-computed vectors of test/cpp/list_scanning.cpp and test/cpp/topk_buffer.cpp transcribed
 * as data in tests/golden/, (b) torch brute force (the reference tests' own ground truth,
 * list_scanning.cpp:432-562), (c) fixtures produced by importing the reference's src/python/utils.py
 * (knn / compute_recall) in the authoring container.  k-means: parity unpinned (no reference test
 * checks centroids or assignments; FAISS's RNG stream is not reproducible here).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>
#include <immintrin.h>
#include <malloc.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define QO_METRIC_IP 0 /* faiss::METRIC_INNER_PRODUCT */
#define QO_METRIC_L2 1 /* faiss::METRIC_L2 */

#define QO_API __attribute__((visibility("default")))

/* The per-query / per-partition scratch of this file (8192-entry buffers, value blocks) is 128 KB - 1 MB: exactly glibc's
 * default mmap threshold, so every malloc/free became an mmap/munmap under the process-wide mm lock and 128 threads ran
 * 5x faster than one (BENCH r02a).  Serve those blocks from the per-thread arenas instead. */
__attribute__((constructor)) static void qo_init_allocator(void) {
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
}

/* How many cores' worth of arithmetic the host actually gives this process: a fixed register-only FMA loop, timed on 1 and on
 * `nthreads` threads; returns (aggregate rate on nthreads) / (rate on 1).  bench.py reports it next to the thread count, so
 * that a 128-thread host that is shared / sandboxed / quota-limited is not mistaken for a 128-core baseline. */
static double fma_spin(long iters) {
    __m256 a0 = _mm256_set1_ps(1.0f), a1 = a0, a2 = a0, a3 = a0;
    const __m256 m = _mm256_set1_ps(0.9999999f), c = _mm256_set1_ps(1e-7f);
    for (long i = 0; i < iters; i++) {
        a0 = _mm256_fmadd_ps(a0, m, c);
        a1 = _mm256_fmadd_ps(a1, m, c);
        a2 = _mm256_fmadd_ps(a2, m, c);
        a3 = _mm256_fmadd_ps(a3, m, c);
    }
    float out[8];
    _mm256_storeu_ps(out, _mm256_add_ps(_mm256_add_ps(a0, a1), _mm256_add_ps(a2, a3)));
    return out[0];
}
QO_API double qo_effective_cores(int nthreads, long iters) {
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    volatile double sink = 0;
    double t0 = omp_get_wtime();
    sink += fma_spin(iters);
    const double t1 = omp_get_wtime() - t0;
    t0 = omp_get_wtime();
#pragma omp parallel num_threads(nthreads)
    { sink += fma_spin(iters); }
    const double tn = omp_get_wtime() - t0;
    (void)sink;
    return tn > 0 ? (double)nthreads * t1 / tn : 1.0;
#else
    return 1.0;
#endif
}

/* ------------------------------------------------------------------------------------------------
 * Distance primitives (canonical k-ordered fmaf chains)
 * ---------------------------------------------------------------------------------------------- */

/* faiss::fvec_inner_product call sites: list_scanning.h:248,273 */
QO_API float qo_ip(const float *x, const float *y, int d) {
    float acc = 0.0f;
    for (int k = 0; k < d; k++) acc = fmaf(x[k], y[k], acc);
    return acc;
}

/* faiss::fvec_L2sqr call sites: list_scanning.h:260,286 -- direct form sum (x-y)^2 */
QO_API float qo_l2sqr_direct(const float *x, const float *y, int d) {
    float acc = 0.0f;
    for (int k = 0; k < d; k++) {
        float t = x[k] - y[k];
        acc = fmaf(t, t, acc);
    }
    return acc;
}

/* faiss::knn_L2sqr (BLAS path): ||x||^2 + ||y||^2 - 2 x.y, negative clamped to 0 */
static inline float l2sqr_expanded(float xn, float yn, float ip) {
    float s = xn + yn;
    float r = fmaf(-2.0f, ip, s); /* == s - 2*ip exactly rounded once (2*ip is exact) */
    return r < 0.0f ? 0.0f : r;
}
QO_API float qo_l2sqr_expanded(float xn, float yn, float ip) { return l2sqr_expanded(xn, yn, ip); }

QO_API void qo_row_norms(const float *x, int64_t n, int d, float *out) {
    for (int64_t i = 0; i < n; i++) out[i] = qo_ip(x + i * d, x + i * d, d);
}

/* 8 independent chains at once (8 queries vs one row): each lane is still ONE k-ordered fmaf chain,
 * so the result is bit-identical to qo_ip; this only makes the CPU port run at SIMD speed. */
static inline __m256 ip8_chain(const float *xT /* [d][8] */, const float *y, int d) {
    __m256 acc = _mm256_setzero_ps();
    for (int k = 0; k < d; k++)
        acc = _mm256_fmadd_ps(_mm256_load_ps(xT + (size_t)k * 8), _mm256_broadcast_ss(y + k), acc);
    return acc;
}

/* ------------------------------------------------------------------------------------------------
 * TopkBuffer  (list_scanning.h:41-204)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    float v;
    int64_t id;
} qo_pair;

typedef struct qo_topk {
    int k;        /* k_ */
    int cap;      /* topk_.size(), default TOP_K_BUFFER_CAPACITY 8192 (list_scanning.h:39) */
    int curr;     /* curr_offset_ */
    int desc;     /* is_descending_ */
    qo_pair *buf; /* topk_ */
} qo_topk;

static int cmp_asc(const void *a, const void *b) {
    const qo_pair *p = (const qo_pair *)a, *q = (const qo_pair *)b;
    if (p->v < q->v) return -1;
    if (p->v > q->v) return 1;
    return (p->id > q->id) - (p->id < q->id);
}
static int cmp_desc(const void *a, const void *b) {
    const qo_pair *p = (const qo_pair *)a, *q = (const qo_pair *)b;
    if (p->v > q->v) return -1;
    if (p->v < q->v) return 1;
    return (p->id > q->id) - (p->id < q->id);
}

static void topk_init(qo_topk *t, int k, int desc, int cap) {
    if (cap < k + 1) cap = k + 1; /* the reference asserts k <= capacity (list_scanning.h:55); +1 keeps add() in bounds */
    t->k = k;
    t->cap = cap;
    t->curr = 0;
    t->desc = desc;
    t->buf = (qo_pair *)malloc(sizeof(qo_pair) * (size_t)cap);
    for (int i = 0; i < cap; i++) { /* sentinels, list_scanning.h:57-63 */
        t->buf[i].v = desc ? -INFINITY : FLT_MAX;
        t->buf[i].id = -1;
    }
}

/* flush(): list_scanning.h:151-173.  std::partial_sort keeping k (heap select over the first k, then sort_heap),
 * or a full sort of curr (<= k) entries.  The comparator is the total order (key, id), so the result does not depend
 * on the algorithm; the heap form is what makes a flush of 8192 candidates cost O(n log k) like the reference's. */
static inline int pair_less(const qo_pair *p, const qo_pair *q, int desc) {
    if (p->v != q->v) return desc ? p->v > q->v : p->v < q->v;
    return p->id < q->id;
}
static inline void heap_sift_down(qo_pair *h, int n, int i, int desc) { /* max-heap under pair_less */
    qo_pair x = h[i];
    for (;;) {
        int c = 2 * i + 1;
        if (c >= n) break;
        if (c + 1 < n && pair_less(&h[c], &h[c + 1], desc)) c++;
        if (!pair_less(&x, &h[c], desc)) break;
        h[i] = h[c];
        i = c;
    }
    h[i] = x;
}
static void topk_flush(qo_topk *t) {
    const int n = t->curr, k = t->k, desc = t->desc;
    if (n > k && k > 0) {
        qo_pair *h = t->buf;
        for (int i = k / 2 - 1; i >= 0; i--) heap_sift_down(h, k, i, desc);
        for (int i = k; i < n; i++)
            if (pair_less(&h[i], &h[0], desc)) {
                h[0] = h[i];
                heap_sift_down(h, k, 0, desc);
            }
        t->curr = k;
    }
    qsort(t->buf, (size_t)t->curr, sizeof(qo_pair), t->desc ? cmp_desc : cmp_asc);
}

/* add(): list_scanning.h:117-122 */
static inline void topk_add(qo_topk *t, float v, int64_t id) {
    if (t->curr >= t->cap) topk_flush(t);
    t->buf[t->curr].v = v;
    t->buf[t->curr].id = id;
    t->curr++;
}

/* batch_add(): list_scanning.h:124-149 (job counters are threading bookkeeping, not restated) */
static void topk_batch_add(qo_topk *t, const float *v, const int64_t *ids, int n) {
    for (int i = 0; i < n; i++) topk_add(t, v[i], ids[i]);
}

QO_API qo_topk *qo_topk_create(int k, int is_descending, int capacity) {
    qo_topk *t = (qo_topk *)malloc(sizeof(qo_topk));
    topk_init(t, k, is_descending, capacity > 0 ? capacity : 8192);
    return t;
}
QO_API void qo_topk_destroy(qo_topk *t) {
    if (!t) return;
    free(t->buf);
    free(t);
}
QO_API void qo_topk_add(qo_topk *t, float v, int64_t id) { topk_add(t, v, id); }
QO_API void qo_topk_batch_add(qo_topk *t, const float *v, const int64_t *ids, int n) { topk_batch_add(t, v, ids, n); }
/* reset(): list_scanning.h:104-115 */
QO_API void qo_topk_reset(qo_topk *t) {
    t->curr = 0;
    for (int i = 0; i < t->k && i < t->cap; i++) {
        t->buf[i].v = t->desc ? -INFINITY : FLT_MAX;
        t->buf[i].id = -1;
    }
}
/* get_topk()/get_topk_indices(): list_scanning.h:175-203 -> min(curr,k) sorted entries; returns count */
QO_API int qo_topk_get(qo_topk *t, float *out_v, int64_t *out_id) {
    topk_flush(t);
    int n = t->curr < t->k ? t->curr : t->k;
    for (int i = 0; i < n; i++) {
        if (out_v) out_v[i] = t->buf[i].v;
        if (out_id) out_id[i] = t->buf[i].id;
    }
    return n;
}

/* ------------------------------------------------------------------------------------------------
 * scan_list  (list_scanning.h:241-311): one query x one partition, direct per-row distances.
 * squared_domain = 0: literal -- L2 adds sqrtf(fvec_L2sqr) like the reference (list_scanning.h:260,286)
 * squared_domain = 1: L2 adds the squared distance (search-level callers sqrt what they return)
 * list_ids == NULL: ids are row numbers (scan_list_no_ids_*)
 * ---------------------------------------------------------------------------------------------- */
QO_API void qo_scan_list(const float *q, const float *vecs, const int64_t *ids, int n, int d, qo_topk *buf, int metric,
                         int squared_domain) {
    const float *v = vecs;
    for (int l = 0; l < n; l++, v += d) {
        float val;
        if (metric == QO_METRIC_IP) {
            val = qo_ip(q, v, d);
        } else {
            val = qo_l2sqr_direct(q, v, d);
            if (!squared_domain) val = sqrtf(val);
        }
        topk_add(buf, val, ids ? ids[l] : (int64_t)l);
    }
}

/* 8-rows-at-a-time form of the L2 direct chain (8x8 register transpose): same bits as
 * qo_l2sqr_direct per row, used by the timed cpu_baseline so the port is not handicapped by
 * a scalar latency chain the reference (FAISS AVX2) does not have. */
static inline void transpose8(__m256 r[8]) {
    __m256 t0 = _mm256_unpacklo_ps(r[0], r[1]), t1 = _mm256_unpackhi_ps(r[0], r[1]);
    __m256 t2 = _mm256_unpacklo_ps(r[2], r[3]), t3 = _mm256_unpackhi_ps(r[2], r[3]);
    __m256 t4 = _mm256_unpacklo_ps(r[4], r[5]), t5 = _mm256_unpackhi_ps(r[4], r[5]);
    __m256 t6 = _mm256_unpacklo_ps(r[6], r[7]), t7 = _mm256_unpackhi_ps(r[6], r[7]);
    __m256 s0 = _mm256_shuffle_ps(t0, t2, 0x44), s1 = _mm256_shuffle_ps(t0, t2, 0xEE);
    __m256 s2 = _mm256_shuffle_ps(t1, t3, 0x44), s3 = _mm256_shuffle_ps(t1, t3, 0xEE);
    __m256 s4 = _mm256_shuffle_ps(t4, t6, 0x44), s5 = _mm256_shuffle_ps(t4, t6, 0xEE);
    __m256 s6 = _mm256_shuffle_ps(t5, t7, 0x44), s7 = _mm256_shuffle_ps(t5, t7, 0xEE);
    r[0] = _mm256_permute2f128_ps(s0, s4, 0x20);
    r[1] = _mm256_permute2f128_ps(s1, s5, 0x20);
    r[2] = _mm256_permute2f128_ps(s2, s6, 0x20);
    r[3] = _mm256_permute2f128_ps(s3, s7, 0x20);
    r[4] = _mm256_permute2f128_ps(s0, s4, 0x31);
    r[5] = _mm256_permute2f128_ps(s1, s5, 0x31);
    r[6] = _mm256_permute2f128_ps(s2, s6, 0x31);
    r[7] = _mm256_permute2f128_ps(s3, s7, 0x31);
}

/* values for rows [0,n): out[l] = chain(q, vecs[l]) ; metric L2 -> direct squared form, IP -> dot */
static void row_values_fast(const float *q, const float *vecs, int n, int d, int metric, float *out) {
    int l = 0;
    int d8 = d & ~7;
    for (; l + 8 <= n; l += 8) {
        __m256 acc = _mm256_setzero_ps();
        const float *base = vecs + (size_t)l * d;
        for (int k = 0; k < d8; k += 8) {
            __m256 r[8];
            for (int j = 0; j < 8; j++) r[j] = _mm256_loadu_ps(base + (size_t)j * d + k);
            transpose8(r); /* r[t] = element k+t of the 8 rows */
            for (int t = 0; t < 8; t++) {
                __m256 qk = _mm256_broadcast_ss(q + k + t);
                if (metric == QO_METRIC_IP) {
                    acc = _mm256_fmadd_ps(qk, r[t], acc);
                } else {
                    __m256 df = _mm256_sub_ps(qk, r[t]);
                    acc = _mm256_fmadd_ps(df, df, acc);
                }
            }
        }
        float a[8] __attribute__((aligned(32)));
        _mm256_store_ps(a, acc);
        for (int j = 0; j < 8; j++) { /* tail dims, still the same sequential chain */
            float s = a[j];
            const float *y = base + (size_t)j * d;
            for (int k = d8; k < d; k++) {
                if (metric == QO_METRIC_IP) {
                    s = fmaf(q[k], y[k], s);
                } else {
                    float t = q[k] - y[k];
                    s = fmaf(t, t, s);
                }
            }
            out[l + j] = s;
        }
    }
    for (; l < n; l++) {
        const float *y = vecs + (size_t)l * d;
        out[l] = metric == QO_METRIC_IP ? qo_ip(q, y, d) : qo_l2sqr_direct(q, y, d);
    }
}

/* fast == 2, TIMING ONLY: SIMD across the dimensions of ONE row with 8 lane-wise partial sums and a horizontal add -- what
 * FAISS's AVX2 fvec_L2sqr / fvec_inner_product do [FAISS-upstream].  Not the canonical order (last-bit differences), so this
 * form is never used as the checker; bench.py's cpu_baseline times it as "the reference's way of computing a row". */
static inline float hsum8(__m256 v) {
    __m128 lo = _mm256_castps256_ps128(v), hi = _mm256_extractf128_ps(v, 1);
    lo = _mm_add_ps(lo, hi);
    lo = _mm_hadd_ps(lo, lo);
    lo = _mm_hadd_ps(lo, lo);
    return _mm_cvtss_f32(lo);
}
static void scan_list_lanes(const float *q, const float *vecs, const int64_t *ids, int n, int d, qo_topk *buf, int metric) {
    const int d8 = d & ~7;
    for (int l = 0; l < n; l++) {
        const float *y = vecs + (size_t)l * d;
        __m256 acc = _mm256_setzero_ps();
        if (metric == QO_METRIC_IP) {
            for (int k = 0; k < d8; k += 8) acc = _mm256_fmadd_ps(_mm256_loadu_ps(q + k), _mm256_loadu_ps(y + k), acc);
        } else {
            for (int k = 0; k < d8; k += 8) {
                __m256 t = _mm256_sub_ps(_mm256_loadu_ps(q + k), _mm256_loadu_ps(y + k));
                acc = _mm256_fmadd_ps(t, t, acc);
            }
        }
        float v = hsum8(acc);
        for (int k = d8; k < d; k++) v += metric == QO_METRIC_IP ? q[k] * y[k] : (q[k] - y[k]) * (q[k] - y[k]);
        topk_add(buf, v, ids ? ids[l] : (int64_t)l);
    }
}

/* bit-identical to qo_scan_list(..., squared_domain=1) but SIMD across rows */
static void scan_list_fast(const float *q, const float *vecs, const int64_t *ids, int n, int d, qo_topk *buf, int metric) {
    enum { BLK = 1024 };
    float vals[BLK];
    for (int s = 0; s < n; s += BLK) {
        int m = n - s < BLK ? n - s : BLK;
        row_values_fast(q, vecs + (size_t)s * d, m, d, metric, vals);
        for (int l = 0; l < m; l++) topk_add(buf, vals[l], ids ? ids[s + l] : (int64_t)(s + l));
    }
}
QO_API void qo_scan_list_fast(const float *q, const float *vecs, const int64_t *ids, int n, int d, qo_topk *buf, int metric) {
    scan_list_fast(q, vecs, ids, n, d, buf, metric);
}

/* ------------------------------------------------------------------------------------------------
 * knn_L2sqr / knn_inner_product restatement + batched_scan_list (list_scanning.h:313-366)
 *   nq queries x one list -> per query local top-k_max (k_max = min(k, n)), selected on the squared
 *   distance (L2) / inner product (IP) with the (key,id) tie rule on the MAPPED id, then pushed into
 *   the per-query buffers with batch_add.  squared_domain as in qo_scan_list (the reference sqrt()s
 *   before batch_add, list_scanning.h:353-357).
 * ---------------------------------------------------------------------------------------------- */
static void batched_values(const float *queries, const float *qnorm, int nq, const float *vecs, const float *ynorm, int n, int d,
                           int metric, float *out /* [nq][n] */) {
    /* blocks of 8 queries, transposed to [d][8] so that 8 chains run in one AVX register */
    float *xT = (float *)aligned_alloc(32, sizeof(float) * 8 * (size_t)d);
    for (int q0 = 0; q0 < nq; q0 += 8) {
        int nb = nq - q0 < 8 ? nq - q0 : 8;
        for (int k = 0; k < d; k++)
            for (int j = 0; j < 8; j++) xT[(size_t)k * 8 + j] = j < nb ? queries[(size_t)(q0 + j) * d + k] : 0.0f;
        for (int l = 0; l < n; l++) {
            const float *y = vecs + (size_t)l * d;
            float ip[8] __attribute__((aligned(32)));
            _mm256_store_ps(ip, ip8_chain(xT, y, d));
            if (metric == QO_METRIC_IP) {
                for (int j = 0; j < nb; j++) out[(size_t)(q0 + j) * n + l] = ip[j];
            } else {
                const float yn = ynorm[l]; /* = qo_ip(y, y, d), computed once per list by the caller */
                for (int j = 0; j < nb; j++) out[(size_t)(q0 + j) * n + l] = l2sqr_expanded(qnorm[q0 + j], yn, ip[j]);
            }
        }
    }
    free(xT);
}

QO_API void qo_batched_scan_list(const float *queries, const float *vecs, const int64_t *ids, int nq, int n, int d,
                                 qo_topk **bufs, int metric, int squared_domain) {
    if (n == 0 || vecs == NULL) return; /* list_scanning.h:321-324 */
    int k = bufs[0]->k;
    int k_max = k < n ? k : n; /* list_scanning.h:327-328 */
    int desc = metric == QO_METRIC_IP;
    float *qnorm = (float *)malloc(sizeof(float) * (size_t)nq);
    float *ynorm = (float *)malloc(sizeof(float) * (size_t)n);
    if (metric == QO_METRIC_L2) {
        qo_row_norms(queries, nq, d, qnorm);
        qo_row_norms(vecs, n, d, ynorm);
    }
    enum { QB = 64 };
    float *vals = (float *)malloc(sizeof(float) * (size_t)QB * (size_t)n);
    qo_topk local;
    topk_init(&local, k_max, desc, k_max * 10 > 8192 ? k_max * 10 : 8192);
    float *ov = (float *)malloc(sizeof(float) * (size_t)k_max);
    int64_t *oi = (int64_t *)malloc(sizeof(int64_t) * (size_t)k_max);
    for (int q0 = 0; q0 < nq; q0 += QB) {
        int nb = nq - q0 < QB ? nq - q0 : QB;
        batched_values(queries + (size_t)q0 * d, qnorm + q0, nb, vecs, ynorm, n, d, metric, vals);
        for (int j = 0; j < nb; j++) {
            local.curr = 0;
            for (int l = 0; l < n; l++) topk_add(&local, vals[(size_t)j * n + l], ids ? ids[l] : (int64_t)l);
            topk_flush(&local);
            int m = local.curr;
            for (int i = 0; i < m; i++) {
                ov[i] = (metric == QO_METRIC_L2 && !squared_domain) ? sqrtf(local.buf[i].v) : local.buf[i].v;
                oi[i] = local.buf[i].id;
            }
            topk_batch_add(bufs[q0 + j], ov, oi, m);
        }
    }
    free(ov);
    free(oi);
    free(local.buf);
    free(vals);
    free(qnorm);
    free(ynorm);
}

/* ------------------------------------------------------------------------------------------------
 * Partition store view used by the search-level restatements: CSR arena
 *   vecs [offsets[nlist]][d], ids [offsets[nlist]], partition p = rows offsets[p] .. offsets[p+1]-1
 * (row order inside a partition = IndexPartition::codes_/ids_ order, index_partition.cpp:52-59)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const float *vecs;
    const int64_t *ids;
    const int64_t *offsets;
    int64_t nlist;
    int d;
} qo_store;

static void emit_result(qo_topk *t, int k, int metric, int64_t *out_ids, float *out_dist) {
    /* output + padding: query_coordinator.cpp:586-601 / 774-788 */
    topk_flush(t);
    int n = t->curr < k ? t->curr : k;
    for (int i = 0; i < n; i++) {
        out_ids[i] = t->buf[i].id;
        out_dist[i] = metric == QO_METRIC_L2 ? sqrtf(t->buf[i].v) : t->buf[i].v;
    }
    for (int i = n; i < k; i++) {
        out_ids[i] = -1;
        out_dist[i] = metric == QO_METRIC_IP ? -INFINITY : INFINITY;
    }
}

/* serial_scan: query_coordinator.cpp:471-611.  pids [nq][P] (-1 = skip, :540).  fast=1 uses the
 * SIMD-across-rows form (bit-identical).  num_threads as SearchParams::num_threads (parallel_for). */
QO_API void qo_serial_scan(const float *x, int64_t nq, const float *vecs, const int64_t *ids, const int64_t *offsets,
                           int64_t nlist, int d, const int64_t *pids, int P, int k, int metric, int num_threads, int fast,
                           int64_t *out_ids, float *out_dist) {
    if (k <= 0) k = 1; /* :490 */
    if (num_threads <= 0) {
#ifdef _OPENMP
        num_threads = omp_get_max_threads();
#else
        num_threads = 1;
#endif
    }
    /* one TopkBuffer per thread, re-armed per query (the reference constructs a fresh one per query, :517: same contents) */
#pragma omp parallel num_threads(num_threads)
    {
        qo_topk buf;
        topk_init(&buf, k, metric == QO_METRIC_IP, 8192 > k ? 8192 : k); /* :517 */
#pragma omp for schedule(dynamic, 1)
        for (int64_t q = 0; q < nq; q++) {
            buf.curr = 0;
            for (int p = 0; p < P; p++) {
                int64_t pi = pids[q * P + p];
                if (pi < 0 || pi >= nlist) continue;
                int64_t o = offsets[pi];
                int n = (int)(offsets[pi + 1] - o);
                if (fast == 2)
                    scan_list_lanes(x + q * d, vecs + o * d, ids + o, n, d, &buf, metric);
                else if (fast)
                    scan_list_fast(x + q * d, vecs + o * d, ids + o, n, d, &buf, metric);
                else
                    qo_scan_list(x + q * d, vecs + o * d, ids + o, n, d, &buf, metric, 1);
            }
            emit_result(&buf, k, metric, out_ids + q * k, out_dist + q * k);
        }
        free(buf.buf);
    }
}

/* batched_serial_scan: query_coordinator.cpp:675-799.  Group queries by partition (:707-721), one
 * batched_scan_list per group (:742-749), merge into per-query global buffers (:752-758). */
QO_API void qo_batched_serial_scan(const float *x, int64_t nq, const float *vecs, const int64_t *ids,
                                   const int64_t *offsets, int64_t nlist, int d, const int64_t *pids, int P, int k,
                                   int metric, int num_threads, int64_t *out_ids, float *out_dist) {
    if (k <= 0) k = 1;
    int desc = metric == QO_METRIC_IP;
    qo_topk **global = (qo_topk **)malloc(sizeof(qo_topk *) * (size_t)(nq > 0 ? nq : 1));
    for (int64_t q = 0; q < nq; q++) global[q] = qo_topk_create(k, desc, 10 * k); /* create_buffers :233-239 */
    /* counting sort of (q,p) pairs by partition id */
    int64_t *cnt = (int64_t *)calloc((size_t)nlist + 1, sizeof(int64_t));
    for (int64_t i = 0; i < nq * P; i++) {
        int64_t pi = pids[i];
        if (pi >= 0 && pi < nlist) cnt[pi + 1]++;
    }
    for (int64_t p = 0; p < nlist; p++) cnt[p + 1] += cnt[p];
    int64_t total = cnt[nlist];
    int64_t *grouped = (int64_t *)malloc(sizeof(int64_t) * (size_t)(total > 0 ? total : 1));
    int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nlist > 0 ? nlist : 1));
    memcpy(cur, cnt, sizeof(int64_t) * (size_t)nlist);
    for (int64_t q = 0; q < nq; q++)
        for (int p = 0; p < P; p++) {
            int64_t pi = pids[q * P + p];
            if (pi >= 0 && pi < nlist) grouped[cur[pi]++] = q;
        }
    if (num_threads <= 0) {
#ifdef _OPENMP
        num_threads = omp_get_max_threads();
#else
        num_threads = 1;
#endif
    }
    /* the merge into global buffers is order-independent under the (key,id) total order, so the
     * parallel loop only needs mutual exclusion per query (the reference holds a mutex per buffer, :133) */
#ifdef _OPENMP
    omp_lock_t *qlock = (omp_lock_t *)malloc(sizeof(omp_lock_t) * (size_t)(nq > 0 ? nq : 1));
    for (int64_t q = 0; q < nq; q++) omp_init_lock(&qlock[q]);
#endif
#pragma omp parallel for num_threads(num_threads) schedule(dynamic, 1)
    for (int64_t p = 0; p < nlist; p++) {
        int64_t g0 = cnt[p], g1 = cnt[p + 1];
        int nb = (int)(g1 - g0);
        if (nb == 0) continue;
        int64_t o = offsets[p];
        int n = (int)(offsets[p + 1] - o);
        if (n == 0) continue;
        float *xs = (float *)malloc(sizeof(float) * (size_t)nb * (size_t)d); /* index_select :728-729 */
        qo_topk **local = (qo_topk **)malloc(sizeof(qo_topk *) * (size_t)nb);
        for (int j = 0; j < nb; j++) {
            memcpy(xs + (size_t)j * d, x + grouped[g0 + j] * d, sizeof(float) * (size_t)d);
            local[j] = qo_topk_create(k, desc, 10 * k);
        }
        qo_batched_scan_list(xs, vecs + o * d, ids + o, nb, n, d, local, metric, 1);
        for (int j = 0; j < nb; j++) {
            topk_flush(local[j]);
            int m = local[j]->curr;
            float *v = (float *)malloc(sizeof(float) * (size_t)(m > 0 ? m : 1));
            int64_t *id = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m > 0 ? m : 1));
            for (int i = 0; i < m; i++) {
                v[i] = local[j]->buf[i].v;
                id[i] = local[j]->buf[i].id;
            }
#ifdef _OPENMP
            omp_set_lock(&qlock[grouped[g0 + j]]);
#endif
            topk_batch_add(global[grouped[g0 + j]], v, id, m);
#ifdef _OPENMP
            omp_unset_lock(&qlock[grouped[g0 + j]]);
#endif
            free(v);
            free(id);
            qo_topk_destroy(local[j]);
        }
        free(local);
        free(xs);
    }
    for (int64_t q = 0; q < nq; q++) {
        emit_result(global[q], k, metric, out_ids + q * k, out_dist + q * k);
        qo_topk_destroy(global[q]);
    }
#ifdef _OPENMP
    for (int64_t q = 0; q < nq; q++) omp_destroy_lock(&qlock[q]);
    free(qlock);
#endif
    free(global);
    free(cnt);
    free(grouped);
    free(cur);
}

/* Coarse step = parent (flat) index search, query_coordinator.cpp:628-644: batched scan of the
 * centroid list with k = min(nprobe, nlist); ids returned are the parent's ids_ = partition ids.
 * out_pids [nq][kk], kk = min(nprobe, nlist); out_cdist may be NULL. */
QO_API int qo_coarse(const float *x, int64_t nq, const float *centroids, const int64_t *centroid_ids, int64_t nlist, int d,
                     int nprobe, int metric, int num_threads, int64_t *out_pids, float *out_cdist) {
    int kk = nprobe < nlist ? nprobe : (int)nlist;
    if (kk <= 0) return 0;
    if (num_threads <= 0) {
#ifdef _OPENMP
        num_threads = omp_get_max_threads();
#else
        num_threads = 1;
#endif
    }
    /* One partition (the parent's centroid list), every query probes it: the batched scan of that single group
     * (batched_serial_scan :723-761 -> batched_scan_list -> knn_L2sqr / knn_inner_product).  FAISS's knn_* is OpenMP-parallel
     * over query blocks [FAISS-upstream]; so is this loop (blocks of 8 queries = one AVX register of chains).  Each query's
     * result is its local top-kk under the (key, id) order -- the same as pushing it through the global buffer. */
    const int n = (int)nlist;
    float *ynorm = (float *)malloc(sizeof(float) * (size_t)n);
    if (metric == QO_METRIC_L2) qo_row_norms(centroids, n, d, ynorm);
#pragma omp parallel num_threads(num_threads)
    {
        float *vals = (float *)malloc(sizeof(float) * 8 * (size_t)n);
        float *dtmp = (float *)malloc(sizeof(float) * (size_t)kk);
        qo_topk local;
        topk_init(&local, kk, metric == QO_METRIC_IP, kk * 10 > 8192 ? kk * 10 : 8192);
#pragma omp for schedule(dynamic, 1)
        for (int64_t q0 = 0; q0 < nq; q0 += 8) {
            const int nb = (int)(nq - q0 < 8 ? nq - q0 : 8);
            float qn[8];
            if (metric == QO_METRIC_L2) qo_row_norms(x + q0 * d, nb, d, qn);
            batched_values(x + q0 * d, qn, nb, centroids, ynorm, n, d, metric, vals);
            for (int j = 0; j < nb; j++) {
                local.curr = 0;
                for (int l = 0; l < n; l++) topk_add(&local, vals[(size_t)j * n + l], centroid_ids ? centroid_ids[l] : (int64_t)l);
                emit_result(&local, kk, metric, out_pids + (q0 + j) * kk, out_cdist ? out_cdist + (q0 + j) * kk : dtmp);
            }
        }
        free(local.buf);
        free(dtmp);
        free(vals);
    }
    free(ynorm);
    return kk;
}

/* QueryCoordinator::search, query_coordinator.cpp:612-657, fixed nprobe.  batched_scan selects
 * batched_serial_scan vs serial_scan (:659-673); centroids==NULL = flat index (scan every partition). */
QO_API void qo_search(const float *x, int64_t nq, const float *centroids, const int64_t *centroid_ids, const float *vecs,
                      const int64_t *ids, const int64_t *offsets, int64_t nlist, int d, int nprobe, int k, int metric,
                      int batched_scan, int num_threads, int fast, int64_t *out_ids, float *out_dist) {
    int P;
    int64_t *pids;
    if (!centroids) { /* :624-626 arange(nlist) for every query */
        P = (int)nlist;
        pids = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nq > 0 ? nq : 1) * (size_t)(P > 0 ? P : 1));
        for (int64_t q = 0; q < nq; q++)
            for (int p = 0; p < P; p++) pids[q * P + p] = p;
    } else {
        P = nprobe < nlist ? nprobe : (int)nlist;
        pids = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nq > 0 ? nq : 1) * (size_t)(P > 0 ? P : 1));
        qo_coarse(x, nq, centroids, centroid_ids, nlist, d, nprobe, metric, num_threads, pids, NULL);
    }
    if (batched_scan)
        qo_batched_serial_scan(x, nq, vecs, ids, offsets, nlist, d, pids, P, k, metric, num_threads, out_ids, out_dist);
    else
        qo_serial_scan(x, nq, vecs, ids, offsets, nlist, d, pids, P, k, metric, num_threads, fast, out_ids, out_dist);
    free(pids);
}

/* ------------------------------------------------------------------------------------------------
 * k-means (clustering.cpp:13-97 wraps faiss::Clustering::train + IndexFlat::search(k=1))
 * PARITY UNPINNED: no reference test checks centroids/assignments and FAISS's RNG is not reproducible
 * here.  Canonical choices (shared with the HIP path):
 *   assign : argmin over centroids of the expanded L2 (or argmax of the dot) with ties -> lower index
 *   update : centroid = (sum of assigned rows, added in ascending row order, fp32) * (1/count)... see below
 *   empty  : FAISS split_clusters semantics restated: an empty cluster takes a copy of a large cluster's
 *            centroid, the pair perturbed by (1 +/- 1/1024), chosen deterministically (largest count first)
 *   init   : centroids = rows perm[0..k) of a splitmix64-driven Fisher-Yates permutation (seed 1234)
 *   subsample: if n > 256*k (FAISS max_points_per_centroid), train on the first 256*k rows of that permutation
 * ---------------------------------------------------------------------------------------------- */
static inline uint64_t splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* perm[0..m): first m entries of the Fisher-Yates permutation of [0,n) */
QO_API void qo_rand_perm(int64_t n, int64_t m, uint64_t seed, int64_t *perm_out) {
    int64_t *p = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    for (int64_t i = 0; i < n; i++) p[i] = i;
    uint64_t s = seed;
    for (int64_t i = 0; i < m && i < n - 1; i++) {
        uint64_t r = splitmix64(&s);
        int64_t j = i + (int64_t)(r % (uint64_t)(n - i));
        int64_t t = p[i];
        p[i] = p[j];
        p[j] = t;
    }
    memcpy(perm_out, p, sizeof(int64_t) * (size_t)(m < n ? m : n));
    free(p);
}

/* assign: IndexFlat::search(n, x, 1) (clustering.cpp:63-66).  out_val = squared L2 / dot. */
QO_API void qo_kmeans_assign(const float *x, int64_t n, const float *c, int64_t m, int d, int metric, int num_threads,
                             int64_t *assign, float *out_val) {
    float *cn = (float *)malloc(sizeof(float) * (size_t)m);
    qo_row_norms(c, m, d, cn);
    if (num_threads <= 0) {
#ifdef _OPENMP
        num_threads = omp_get_max_threads();
#else
        num_threads = 1;
#endif
    }
#pragma omp parallel num_threads(num_threads)
    {
        float *xT = (float *)aligned_alloc(32, sizeof(float) * 8 * (size_t)d);
#pragma omp for schedule(static)
        for (int64_t b = 0; b < (n + 7) / 8; b++) {
            int64_t i0 = b * 8;
            int nb = (int)(n - i0 < 8 ? n - i0 : 8);
            float xn[8];
            for (int k = 0; k < d; k++)
                for (int j = 0; j < 8; j++) xT[(size_t)k * 8 + j] = j < nb ? x[(size_t)(i0 + j) * d + k] : 0.0f;
            for (int j = 0; j < nb; j++) xn[j] = qo_ip(x + (i0 + j) * d, x + (i0 + j) * d, d);
            float best[8];
            int64_t bi[8];
            for (int j = 0; j < 8; j++) {
                best[j] = metric == QO_METRIC_IP ? -INFINITY : INFINITY;
                bi[j] = -1;
            }
            for (int64_t cc = 0; cc < m; cc++) {
                float ip[8] __attribute__((aligned(32)));
                _mm256_store_ps(ip, ip8_chain(xT, c + cc * d, d));
                for (int j = 0; j < nb; j++) {
                    if (metric == QO_METRIC_IP) {
                        if (ip[j] > best[j] || bi[j] < 0) {
                            best[j] = ip[j];
                            bi[j] = cc;
                        }
                    } else {
                        float v = l2sqr_expanded(xn[j], cn[cc], ip[j]);
                        if (v < best[j] || bi[j] < 0) {
                            best[j] = v;
                            bi[j] = cc;
                        }
                    }
                }
            }
            for (int j = 0; j < nb; j++) {
                assign[i0 + j] = bi[j];
                if (out_val) out_val[i0 + j] = best[j];
            }
        }
        free(xT);
    }
    free(cn);
}

/* update: per-cluster sums (fp32, rows added in ascending row index order) and counts.
 * This is the accumulate loop of kmeans_refine_partitions (clustering.cpp:162-176: `centroid_sums[c][j] += vec[j]`,
 * row after row) -- the reference's OWN order, followed literally.  (The mean update of kmeans() has no such order
 * to follow: qo_kmeans_accumulate_blocked below.) */
QO_API void qo_kmeans_accumulate(const float *x, int64_t n, int d, const int64_t *assign, int64_t m, float *sums,
                                 int64_t *counts) {
    memset(sums, 0, sizeof(float) * (size_t)m * (size_t)d);
    memset(counts, 0, sizeof(int64_t) * (size_t)m);
    for (int64_t i = 0; i < n; i++) {
        int64_t a = assign[i];
        if (a < 0 || a >= m) continue;
        float *s = sums + a * d;
        const float *v = x + i * d;
        for (int k = 0; k < d; k++) s[k] += v[k];
        counts[a]++;
    }
}

/* The mean update of kmeans() (clustering.cpp:51-55 hands the iteration to faiss::Clustering, whose summation order is its
 * back end's -- unlike the refine loop above there is no reference order to follow): the repository's canonical BLOCKED order,
 * chosen so that a large cluster is many short independent chains on the GPU (k_accumulate_blocked).  With the
 * rows of a centroid in ascending row order:
 *   level 1  block  = 32 consecutive rows of the bucket: s = 0; s += x[row] row after row
 *   level 2  group  = 32 consecutive blocks (1024 rows): run = 0; run += s block after block
 *   level 3  centroid: tot = 0; tot += run group after group
 * Rows with an assignment outside [0, m) are ignored. */
#define QO_KM_L1 32
#define QO_KM_L2 32
QO_API void qo_kmeans_accumulate_blocked(const float *x, int64_t n, int d, const int64_t *assign, int64_t m, float *sums,
                                         int64_t *counts) {
    memset(sums, 0, sizeof(float) * (size_t)m * (size_t)d);
    memset(counts, 0, sizeof(int64_t) * (size_t)m);
    for (int64_t i = 0; i < n; i++)
        if (assign[i] >= 0 && assign[i] < m) counts[assign[i]]++;
    int64_t *begin = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m + 1));
    begin[0] = 0;
    for (int64_t c = 0; c < m; c++) begin[c + 1] = begin[c] + counts[c];
    int64_t *rows = (int64_t *)malloc(sizeof(int64_t) * (size_t)(begin[m] > 0 ? begin[m] : 1));
    int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * (size_t)m);
    memcpy(cur, begin, sizeof(int64_t) * (size_t)m);
    for (int64_t i = 0; i < n; i++) /* ascending row order inside every bucket */
        if (assign[i] >= 0 && assign[i] < m) rows[cur[assign[i]]++] = i;
    float *s = (float *)malloc(sizeof(float) * (size_t)d * 2);
    float *run = s + d;
    for (int64_t c = 0; c < m; c++) {
        float *tot = sums + c * d;
        const int64_t b = begin[c], e = begin[c + 1];
        for (int64_t g0 = b; g0 < e; g0 += (int64_t)QO_KM_L1 * QO_KM_L2) {
            const int64_t g1 = g0 + (int64_t)QO_KM_L1 * QO_KM_L2 < e ? g0 + (int64_t)QO_KM_L1 * QO_KM_L2 : e;
            for (int k = 0; k < d; k++) run[k] = 0.0f;
            for (int64_t b0 = g0; b0 < g1; b0 += QO_KM_L1) {
                const int64_t b1 = b0 + QO_KM_L1 < g1 ? b0 + QO_KM_L1 : g1;
                for (int k = 0; k < d; k++) s[k] = 0.0f;
                for (int64_t r = b0; r < b1; r++) {
                    const float *v = x + rows[r] * d;
                    for (int k = 0; k < d; k++) s[k] += v[k];
                }
                for (int k = 0; k < d; k++) run[k] += s[k];
            }
            for (int k = 0; k < d; k++) tot[k] += run[k];
        }
    }
    free(s);
    free(cur);
    free(rows);
    free(begin);
}

/* centroids = sums / counts (clustering.cpp:122-124: float division by the count; count 0 -> NaN there;
 * keep_empty=1 leaves the previous centroid instead, used by qo_kmeans which then splits) */
QO_API void qo_kmeans_finalize(const float *sums, const int64_t *counts, int64_t m, int d, int keep_empty, float *c) {
    for (int64_t j = 0; j < m; j++) {
        if (counts[j] == 0 && keep_empty) continue;
        float cnt = (float)counts[j];
        for (int k = 0; k < d; k++) c[j * d + k] = sums[j * d + k] / cnt;
    }
}

/* faiss split_clusters restated deterministically: for each empty cluster (ascending index) pick the
 * currently largest cluster (ties -> lowest index), copy its centroid, perturb the pair by (1 +/- eps)
 * alternating sign per dimension, split the count.  Returns number of splits. */
static int split_empty(float *c, int64_t *counts, int64_t m, int d) {
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

/* mean update + empty-cluster split of one Lloyd iteration, given (global) sums and counts -- what every rank of a
 * multi-GPU build runs after the all-reduce (quake_amd/sharded.py) and what qo_kmeans runs per iteration */
QO_API int qo_kmeans_update(const float *sums, int64_t *counts, int64_t m, int d, float *c) {
    qo_kmeans_finalize(sums, counts, m, d, 1, c);
    return split_empty(c, counts, m, d);
}

static void normalize_rows(float *x, int64_t n, int d) {
    /* vectors / vectors.norm(2,1).unsqueeze(1) (clustering.cpp:25-26,59-60): canonical norm = sqrtf(chain) */
    for (int64_t i = 0; i < n; i++) {
        float nn = sqrtf(qo_ip(x + i * d, x + i * d, d));
        for (int k = 0; k < d; k++) x[i * d + k] = x[i * d + k] / nn;
    }
}
QO_API void qo_normalize_rows(float *x, int64_t n, int d) { normalize_rows(x, n, d); }

/* kmeans(): clustering.cpp:13-97.  x is modified in place for IP (normalised copies are what get stored,
 * clustering.cpp:25-26,71).  Outputs centroids [m][d] and the final full assignment [n]. */
QO_API void qo_kmeans(float *x, int64_t n, int d, int64_t m, int metric, int niter, uint64_t seed, int num_threads,
                      float *centroids, int64_t *assign) {
    if (metric == QO_METRIC_IP) normalize_rows(x, n, d);
    int64_t ntrain = n;
    const int64_t max_pts = 256; /* faiss ClusteringParameters::max_points_per_centroid */
    int64_t *perm = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    int64_t need = n > max_pts * m ? max_pts * m : m;
    qo_rand_perm(n, need, seed, perm);
    float *xt = x;
    if (n > max_pts * m) {
        ntrain = max_pts * m;
        xt = (float *)malloc(sizeof(float) * (size_t)ntrain * (size_t)d);
        for (int64_t i = 0; i < ntrain; i++) memcpy(xt + i * d, x + perm[i] * d, sizeof(float) * (size_t)d);
        for (int64_t j = 0; j < m; j++) memcpy(centroids + j * d, xt + j * d, sizeof(float) * (size_t)d);
    } else {
        for (int64_t j = 0; j < m; j++) memcpy(centroids + j * d, x + perm[j] * d, sizeof(float) * (size_t)d);
    }
    float *sums = (float *)malloc(sizeof(float) * (size_t)m * (size_t)d);
    int64_t *counts = (int64_t *)malloc(sizeof(int64_t) * (size_t)m);
    int64_t *ta = (int64_t *)malloc(sizeof(int64_t) * (size_t)ntrain);
    for (int it = 0; it < niter; it++) {
        qo_kmeans_assign(xt, ntrain, centroids, m, d, metric, num_threads, ta, NULL);
        qo_kmeans_accumulate_blocked(xt, ntrain, d, ta, m, sums, counts);
        qo_kmeans_finalize(sums, counts, m, d, 1, centroids);
        split_empty(centroids, counts, m, d);
    }
    if (metric == QO_METRIC_IP) normalize_rows(centroids, m, d); /* clustering.cpp:59-60 */
    qo_kmeans_assign(x, n, centroids, m, d, metric, num_threads, assign, NULL); /* :63-66 */
    if (xt != x) free(xt);
    free(perm);
    free(sums);
    free(counts);
    free(ta);
}

/* kmeans_refine_partitions(): clustering.cpp:99-182 on the CSR view.  In: centroids [m][d] (updated in
 * place to "the centroids used for the last assignment", :178), vecs/ids/offsets over m partitions.
 * Out: new CSR (out_vecs [total][d], out_ids [total], out_offsets [m+1]).  Row order of each new
 * partition = order of append (:174): input partitions in order, rows in order.
 * NB create_buffers(nvec, 1, false) (:146): k=1, so the buffer's ascending flag is irrelevant;
 * batched_scan_list(metric) picks min squared L2 or max dot. */
QO_API void qo_kmeans_refine_partitions(float *centroids, int64_t m, int d, const float *vecs, const int64_t *ids,
                                        const int64_t *offsets, int metric, int refinement_iterations, int num_threads,
                                        float *out_vecs, int64_t *out_ids, int64_t *out_offsets) {
    int iterations = refinement_iterations > 0 ? refinement_iterations : 1; /* :110 */
    int64_t total = offsets[m];
    float *cur_v = (float *)malloc(sizeof(float) * (size_t)(total > 0 ? total : 1) * (size_t)d);
    int64_t *cur_i = (int64_t *)malloc(sizeof(int64_t) * (size_t)(total > 0 ? total : 1));
    int64_t *cur_o = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m + 1));
    memcpy(cur_v, vecs, sizeof(float) * (size_t)total * (size_t)d);
    memcpy(cur_i, ids, sizeof(int64_t) * (size_t)total);
    memcpy(cur_o, offsets, sizeof(int64_t) * (size_t)(m + 1));
    float *sums = (float *)calloc((size_t)m * (size_t)d, sizeof(float));
    int64_t *counts = (int64_t *)calloc((size_t)m, sizeof(int64_t));
    int64_t *assign = (int64_t *)malloc(sizeof(int64_t) * (size_t)(total > 0 ? total : 1));
    for (int iter = 0; iter < iterations; iter++) {
        if (iter > 0) qo_kmeans_finalize(sums, counts, m, d, 0, centroids); /* :122-124 (count 0 -> NaN, as there) */
        /* the arena is the concatenation of the partitions in order, so one assign over all rows ==
         * the per-partition batched_scan_list(k=1) loop (:139-159) */
        qo_kmeans_assign(cur_v, total, centroids, m, d, metric, num_threads, assign, NULL);
        qo_kmeans_accumulate(cur_v, total, d, assign, m, sums, counts); /* :168-171 */
        /* stable bucket by assignment == per-vector append (:174) */
        out_offsets[0] = 0;
        for (int64_t j = 0; j < m; j++) out_offsets[j + 1] = out_offsets[j] + counts[j];
        int64_t *cursor = (int64_t *)malloc(sizeof(int64_t) * (size_t)m);
        memcpy(cursor, out_offsets, sizeof(int64_t) * (size_t)m);
        for (int64_t i = 0; i < total; i++) {
            int64_t a = assign[i];
            if (a < 0) continue; /* NaN centroids: unassigned (cannot happen unless a cluster emptied) */
            int64_t pos = cursor[a]++;
            memcpy(out_vecs + pos * d, cur_v + i * d, sizeof(float) * (size_t)d);
            out_ids[pos] = cur_i[i];
        }
        free(cursor);
        memcpy(cur_v, out_vecs, sizeof(float) * (size_t)total * (size_t)d);
        memcpy(cur_i, out_ids, sizeof(int64_t) * (size_t)total);
        memcpy(cur_o, out_offsets, sizeof(int64_t) * (size_t)(m + 1));
    }
    free(cur_v);
    free(cur_i);
    free(cur_o);
    free(sums);
    free(counts);
    free(assign);
}

/* calculate_recall (list_scanning.h:14-37): per-query |ids ∩ gt| / k */
QO_API void qo_recall(const int64_t *ids, const int64_t *gt, int64_t nq, int k, float *out) {
    for (int64_t q = 0; q < nq; q++) {
        int c = 0;
        for (int j = 0; j < k; j++) {
            int64_t v = ids[q * k + j];
            for (int t = 0; t < k; t++)
                if (gt[q * k + t] == v) {
                    c++;
                    break;
                }
        }
        out[q] = (float)c / (float)k;
    }
}

/* compute_recall (src/python/utils.py:162-177): per-query |set(ids) & set(gt)| / k  (duplicates count once) */
QO_API void qo_recall_set(const int64_t *ids, const int64_t *gt, int64_t nq, int k, float *out) {
    for (int64_t q = 0; q < nq; q++) {
        int c = 0;
        for (int j = 0; j < k; j++) {
            int64_t v = ids[q * k + j];
            int seen = 0;
            for (int t = 0; t < j; t++)
                if (ids[q * k + t] == v) seen = 1;
            if (seen) continue;
            for (int t = 0; t < k; t++)
                if (gt[q * k + t] == v) {
                    c++;
                    break;
                }
        }
        out[q] = (float)c / (float)k;
    }
}

QO_API int qo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------------
 * Adaptive partition scanning (APS): geometry.h + the use_aps branch of serial_scan.
 * PARITY UNPINNED: the reference holds no golden vector for any of this (its tests only print the
 * recall reached, test/cpp/search_recall_tests.cpp:284-340).  The incomplete beta function is pinned
 * against scipy.special.betainc in tests/test_oracle_aps.py; the stopping rule is restated literally.
 * Canonical choices beyond the reference (all dot products are the k-ordered fmaf chain as everywhere
 * else; FAISS leaves the order open):
 *   - acos of the IP boundary angle is (float)acos((double)x) (std::acos(float) is <= 1 ulp of that)
 *   - expression contraction is off (the build flag), like the HIP path
 * ---------------------------------------------------------------------------------------------- */
#define QO_NUM_X_VALUES 1001 /* geometry.h:7 */
#define QO_STOP 1.0e-8       /* geometry.h:9 */
#define QO_TINY 1.0e-30      /* geometry.h:10 */

/* geometry.h:115-161: regularised incomplete beta I_x(a,b), Lentz continued fraction */
QO_API double qo_incomplete_beta(double a, double b, double x) {
    if (x < 0.0 || x > 1.0) return INFINITY;
    if (x > (a + 1.0) / (a + b + 2.0)) return (1.0 - qo_incomplete_beta(b, a, 1.0 - x));
    const double lbeta_ab = lgamma(a) + lgamma(b) - lgamma(a + b);
    const double front = exp(log(x) * a + log(1.0 - x) * b - lbeta_ab) / a;
    double f = 1.0, c = 1.0, d = 0.0;
    for (int i = 0; i <= 200; ++i) {
        int m = i / 2;
        double numerator;
        if (i == 0)
            numerator = 1.0;
        else if (i % 2 == 0)
            numerator = (m * (b - m) * x) / ((a + 2.0 * m - 1.0) * (a + 2.0 * m));
        else
            numerator = -((a + m) * (a + b + m) * x) / ((a + 2.0 * m) * (a + 2.0 * m + 1));
        d = 1.0 + numerator * d;
        if (fabs(d) < QO_TINY) d = QO_TINY;
        d = 1.0 / d;
        c = 1.0 + numerator / c;
        if (fabs(c) < QO_TINY) c = QO_TINY;
        const double cd = c * d;
        f *= cd;
        if (fabs(1.0 - cd) < QO_STOP) return front * (f - 1.0);
    }
    return INFINITY;
}

/* geometry.h:163-180: table of I_x((d+1)/2, 1/2) at x = i/1000.  (The reference fills ONE process-wide table
 * for the first d it sees, :184-188; this restatement takes the table of the index's own d.) */
QO_API void qo_incomplete_beta_table(int d, double *table) {
    double dx = 1.0 / (QO_NUM_X_VALUES - 1);
    double a = (d + 1.0) / 2.0, b = 0.5;
    for (int i = 0; i < QO_NUM_X_VALUES; i++) table[i] = qo_incomplete_beta(a, b, i * dx);
}

/* geometry.h:182-211: linear interpolation in the table (std::max/std::min argument order kept: NaN -> 1.0) */
QO_API double qo_incomplete_beta_lookup(const double *table, double x) {
    double t = (x < 1.0) ? x : 1.0; /* std::min(1.0, x) */
    x = (0.0 < t) ? t : 0.0;        /* std::max(0.0, .) */
    double scaled_x = x * (QO_NUM_X_VALUES - 1);
    int x_index = (int)scaled_x;
    if (x_index > QO_NUM_X_VALUES - 2) x_index = QO_NUM_X_VALUES - 2;
    if (x_index < 0) x_index = 0;
    double y1 = table[x_index], y2 = table[x_index + 1];
    double dx = 1.0 / (QO_NUM_X_VALUES - 1);
    double x1 = x_index * dx;
    return y1 + (x - x1) * (y2 - y1) / dx;
}

/* geometry.h:247-295, ratio = true */
QO_API double qo_log_cap_volume(double radius, double boundary_distance, int d, int use_precomputed, int euclidean,
                                const double *table) {
    double h = radius - boundary_distance;
    double t = (h < 2 * radius) ? h : 2 * radius; /* std::min(2 * radius, h) */
    h = (0.0 < t) ? t : 0.0;                      /* std::max(0.0, .) */
    if (euclidean) {
        double x = sqrt((2 * radius * h - h * h) / (radius * radius));
        double inc_beta = use_precomputed ? qo_incomplete_beta_lookup(table, x) : qo_incomplete_beta((d + 1.0) / 2.0, 0.5, x);
        if (inc_beta <= 0.0 || isnan(inc_beta) || isinf(inc_beta)) return -INFINITY;
        return log(0.5) + log(inc_beta);
    }
    double s1 = sin(radius / 2.0), s2 = sin(boundary_distance / 2.0);
    double log_inc_beta = log(qo_incomplete_beta((d - 1) / 2.0, 0.5, s1 * s1));
    double log_inc_beta_boundary = log(qo_incomplete_beta((d - 1) / 2.0, 0.5, s2 * s2));
    return log(0.5) + log_inc_beta - log_inc_beta_boundary;
}

/* geometry.h:345-407.  returns -1 when fewer than 2 partitions (the reference throws) */
QO_API int qo_recall_profile(const float *bd, int M, float query_radius, int d, int use_precomputed, int euclidean,
                             const double *table, float *probs) {
    if (M < 2) return -1;
    for (int j = 0; j < M; j++) probs[j] = 0.0f;
    for (int j = 1; j < M; j++) {
        float b = bd[j];
        if (b >= query_radius) {
            probs[j] = 0.0f;
            continue;
        }
        double volume_ratio = exp(qo_log_cap_volume(query_radius, b, d, use_precomputed, euclidean, table));
        probs[j] = (float)((volume_ratio > 0.0) ? volume_ratio : 0.0);
    }
    probs[0] = (float)(2.0 * probs[1]);
    double sum = 0.0;
    for (int j = 0; j < M; j++) sum += probs[j];
    if (sum > 0.0f) {
        for (int j = 0; j < M; j++) probs[j] = (float)(probs[j] / sum);
    } else {
        for (int j = 0; j < M; j++) probs[j] = (float)(1.0 / M);
    }
    return 0;
}

/* geometry.h:57-113.  cent[j] = centroid of the j-th ranked candidate partition; out[0] = -1 */
QO_API void qo_boundary_distances(const float *q, const float *const *cent, int M, int d, int euclidean, float *out) {
    float *line = (float *)malloc(sizeof(float) * (size_t)d * 3);
    float *mid = line + d, *res = line + 2 * d;
    for (int j = 0; j < M; j++) out[j] = -1.0f;
    const float *c0 = cent[0];
    if (euclidean) {
        for (int i = 0; i < d; i++) res[i] = q[i] - c0[i];
        for (int j = 1; j < M; j++) {
            for (int i = 0; i < d; i++) line[i] = cent[j][i] - c0[i];
            float A2 = qo_ip(line, line, d);
            float A = sqrtf(A2);
            float dot_val = qo_ip(res, line, d);
            out[j] = fabsf(dot_val - 0.5f * A2) / A;
        }
    } else {
        for (int j = 1; j < M; j++) {
            for (int i = 0; i < d; i++) line[i] = cent[j][i] - c0[i];
            for (int i = 0; i < d; i++) mid[i] = line[i] / 2.0f;
            for (int i = 0; i < d; i++) mid[i] = c0[i] + mid[i];
            float norm = sqrtf(qo_ip(mid, mid, d));
            for (int i = 0; i < d; i++) mid[i] = mid[i] / norm;
            float ang = qo_ip(q, mid, d);
            out[j] = (float)acos((double)ang);
        }
    }
    free(line);
}

/* QueryCoordinator::search with recall_target > 0 and batched_scan == false (query_coordinator.cpp:612-657)
 * followed by serial_scan's use_aps branch (:471-611):
 *   M = max((int)(nlist * initial_search_fraction), 1) candidate partitions from the parent (:638-641),
 *   boundary distances once per query (:529-536), then partition after partition: scan, radius = k-th distance
 *   (TopkBuffer::get_kth_distance, list_scanning.h:187-191: the sentinel when fewer than k results),
 *   recompute the profile when the radius moved by more than recompute_threshold (:562-571),
 *   stop when the probabilities of the partitions BEFORE the current one reach the target (:572-578).
 * expanded = 1: per-row distances in the expanded (norm) form of the batched path, the arithmetic of the HIP kernels;
 * expanded = 0: direct form, as scan_list.  centroid_row_of[pid] = row of that partition's centroid in `centroids`.
 * out_nscan[q] = partitions visited (p reached + 1, or M).  An empty profile (never computed because the radius test
 * produced NaN) contributes 0 to the estimate -- the reference indexes an empty vector there.
 * returns 0, or -1 when M < 2 (compute_recall_profile throws). */
QO_API int qo_search_aps(const float *x, int64_t nq, const float *centroids, const int64_t *centroid_ids, int64_t nlist_parent,
                         const float *vecs, const int64_t *ids, const int64_t *offsets, int64_t nlist, int d, int k, int metric,
                         float recall_target, float recompute_threshold, int use_precomputed, float initial_search_fraction,
                         int expanded, int num_threads, int64_t *out_ids, float *out_dist, int32_t *out_nscan) {
    if (k <= 0) k = 1;
    int M = (int)((float)nlist * initial_search_fraction);
    if (M < 1) M = 1;
    int kk = M < nlist_parent ? M : (int)nlist_parent;
    if (kk < 2) return -1;
    int64_t *pids = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nq > 0 ? nq : 1) * (size_t)kk);
    qo_coarse(x, nq, centroids, centroid_ids, nlist_parent, d, kk, metric, num_threads, pids, NULL);
    /* partition id -> centroid row */
    int64_t max_id = -1;
    for (int64_t i = 0; i < nlist_parent; i++) {
        int64_t id = centroid_ids ? centroid_ids[i] : i;
        if (id > max_id) max_id = id;
    }
    int64_t *row_of = (int64_t *)malloc(sizeof(int64_t) * (size_t)(max_id + 2));
    for (int64_t i = 0; i <= max_id; i++) row_of[i] = -1;
    for (int64_t i = 0; i < nlist_parent; i++) row_of[centroid_ids ? centroid_ids[i] : i] = i;
    double table[QO_NUM_X_VALUES];
    qo_incomplete_beta_table(d, table);
    const int euclid = metric == QO_METRIC_L2;
    if (num_threads <= 0) {
#ifdef _OPENMP
        num_threads = omp_get_max_threads();
#else
        num_threads = 1;
#endif
    }
#pragma omp parallel for num_threads(num_threads) schedule(dynamic, 4)
    for (int64_t q = 0; q < nq; q++) {
        const float *xq = x + q * d;
        qo_topk buf;
        topk_init(&buf, k, metric == QO_METRIC_IP, 8192 > k ? 8192 : k);
        qo_topk *bufp = &buf;
        const float **cent = (const float **)malloc(sizeof(float *) * (size_t)kk);
        float *bd = (float *)malloc(sizeof(float) * (size_t)kk * 2);
        float *probs = bd + kk;
        int have_probs = 0;
        for (int j = 0; j < kk; j++) {
            int64_t pi = pids[q * kk + j];
            cent[j] = centroids + (size_t)(pi >= 0 && pi <= max_id && row_of[pi] >= 0 ? row_of[pi] : 0) * d;
        }
        qo_boundary_distances(xq, cent, kk, d, euclid, bd);
        float query_radius = euclid ? 1000000.0f : -1000000.0f;
        int nscan = 0;
        for (int p = 0; p < kk; p++) {
            int64_t pi = pids[q * kk + p];
            nscan = p + 1;
            if (pi < 0 || pi >= nlist) continue; /* :540 */
            int64_t o = offsets[pi];
            int n = (int)(offsets[pi + 1] - o);
            if (expanded)
                qo_batched_scan_list(xq, vecs + o * d, ids + o, 1, n, d, &bufp, metric, 1);
            else
                scan_list_fast(xq, vecs + o * d, ids + o, n, d, &buf, metric);
            topk_flush(&buf);
            float curr_radius;
            if (buf.curr >= k)
                curr_radius = euclid ? sqrtf(buf.buf[k - 1].v) : buf.buf[k - 1].v;
            else
                curr_radius = euclid ? FLT_MAX : -INFINITY; /* sentinel slot, list_scanning.h:57-63,190 */
            float percent_change = fabsf(curr_radius - query_radius) / curr_radius;
            if (percent_change > recompute_threshold) {
                query_radius = curr_radius;
                qo_recall_profile(bd, kk, query_radius, d, use_precomputed, euclid, table, probs);
                have_probs = 1;
            }
            float recall_estimate = 0.0f;
            if (have_probs)
                for (int i = 0; i < p; i++) recall_estimate += probs[i];
            if (recall_estimate >= recall_target) break;
        }
        emit_result(&buf, k, metric, out_ids + q * k, out_dist + q * k);
        if (out_nscan) out_nscan[q] = nscan;
        free(buf.buf);
        free(cent);
        free(bd);
    }
    free(row_of);
    free(pids);
    return 0;
}
