// qk_device.h -- device-side helpers shared by the scan / dense / k-means kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <climits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- orderable keys: smaller = better --------------------------------------------------------------
__device__ __forceinline__ uint32_t ord_from_l2(float d2) { return __float_as_uint(d2); }  // d2 >= +0
__device__ __forceinline__ uint32_t ord_from_ip(float ip) {
    uint32_t b = __float_as_uint(ip);
    uint32_t asc = b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
    return ~asc;
}
__device__ __forceinline__ float ip_from_ord(uint32_t o) {
    uint32_t asc = ~o;
    uint32_t b = (asc & 0x80000000u) ? (asc ^ 0x80000000u) : ~asc;
    return __uint_as_float(b);
}
// faiss::knn_L2sqr expansion, clamped at 0 (oracle: l2sqr_expanded)
__device__ __forceinline__ float l2_expanded(float xn, float yn, float ip) {
    float r = __fmaf_rn(-2.0f, ip, xn + yn);
    return r < 0.0f ? 0.0f : r;
}

// ---- LDS pool compaction (the TopkBuffer::flush of this design) ------------------------------------
// Keeps the k best of n entries under the total order (ord, id, position) and leaves them sorted in [0,k).
template <int MAXCH>
__device__ __forceinline__ int select_core(uint32_t *ord, int64_t *id, int n, int k, int lane, uint32_t &kth);

template <int MAXCH>
__device__ __forceinline__ int compact_pool(uint32_t *ord, int64_t *id, int n, int k, int lane) {
    // wider than one wave: the rank sort below costs n x MAXCH; cut to the k survivors first (bisection select, measured
    // 2.2 us against 17 us for ranking 192 entries) -- unless a tie on the k-th key needs the full order
    if (MAXCH > 1) {
        n = __builtin_amdgcn_readfirstlane(n);
        uint32_t kth_;
        if (n > k && select_core<MAXCH>(ord, id, n, k, lane, kth_) == k) n = k;
    } else {  // one wave wide: worth it when the rank loop below would run over many more entries than survive
        n = __builtin_amdgcn_readfirstlane(n);
        uint32_t kth_;
        if (n > 2 * k + 4 && select_core<MAXCH>(ord, id, n, k, lane, kth_) == k) n = k;
    }
    uint32_t o[MAXCH];
    int64_t d[MAXCH];
    int rk[MAXCH];
#pragma unroll
    for (int i = 0; i < MAXCH; i++) {
        int e = lane + 64 * i;
        bool has = e < n;
        o[i] = has ? ord[e] : 0xFFFFFFFFu;
        d[i] = has ? id[e] : LLONG_MAX;
        rk[i] = 0;
    }
    // Fast path: rank on the key alone.  Entry t is broadcast from the register of the lane that holds it
    // (v_readlane -> SGPR operand, no LDS round trip), one compare + add per held entry.  Keys are distinct except
    // for exact fp32 distance ties; those are detected with the same pass (eq > 1) and re-ranked under the full
    // (ord, id, position) order below, so the result is always the total order of the oracle.
    n = __builtin_amdgcn_readfirstlane(n);
    int eq[MAXCH];
#pragma unroll
    for (int i = 0; i < MAXCH; i++) eq[i] = 0;
#pragma unroll
    for (int ci = 0; ci < MAXCH; ci++) {
        const int nt = min(64, n - 64 * ci);
        for (int t = 0; t < nt; t++) {
            const uint32_t ot = __builtin_amdgcn_readlane(o[ci], t);
#pragma unroll
            for (int i = 0; i < MAXCH; i++) {
                rk[i] += ot < o[i] ? 1 : 0;
                eq[i] += ot == o[i] ? 1 : 0;
            }
        }
    }
    bool tie = false;
#pragma unroll
    for (int i = 0; i < MAXCH; i++) tie |= (lane + 64 * i < n) && eq[i] > 1;
    if (__ballot(tie) != 0) {
#pragma unroll
        for (int i = 0; i < MAXCH; i++) rk[i] = 0;
        for (int t = 0; t < n; t++) {
            uint32_t ot = ord[t];
            int64_t it = id[t];
#pragma unroll
            for (int i = 0; i < MAXCH; i++) {
                int e = lane + 64 * i;
                bool less = (ot < o[i]) || (ot == o[i] && (it < d[i] || (it == d[i] && t < e)));
                rk[i] += less ? 1 : 0;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MAXCH; i++) {
        int e = lane + 64 * i;
        if (e < n && rk[i] < k) {
            ord[rk[i]] = o[i];
            id[rk[i]] = d[i];
        }
    }
    return n < k ? n : k;
}

// In-loop variant for pools wider than one wave (k > 36): keeps the k best of n entries WITHOUT sorting them and
// returns the k-th key.  The k-th smallest key is found by bisection on the 32 key bits (one ballot + popcount per held
// entry and bit: ~30x fewer instructions than ranking 160+ entries), the survivors are packed with a ballot prefix.
// A tie on the key that straddles the cut needs the (id, position) order: that case falls back to compact_pool.  The
// segment-end compaction always goes through compact_pool, so records leave the kernel sorted under the total order.
template <int MAXCH>
__device__ __forceinline__ int select_core(uint32_t *ord, int64_t *id, int n, int k, int lane, uint32_t &kth) {
    // (n > k; returns k with the survivors packed into [0, k), or -1 with the pool untouched when several entries share the
    //  k-th key -- that cut needs the (id, position) order)
    uint32_t o[MAXCH];
    int64_t d[MAXCH];
#pragma unroll
    for (int i = 0; i < MAXCH; i++) {
        const int e = lane + 64 * i;
        const bool has = e < n;
        o[i] = has ? ord[e] : 0xFFFFFFFFu;  // (a live key is never 0xFFFFFFFF: the append path rejects it)
        d[i] = has ? id[e] : LLONG_MAX;
    }
    // T = the largest value with fewer than k keys below it = the k-th smallest key
    uint32_t T = 0;
    for (int b = 31; b >= 0; b--) {
        const uint32_t tr = T | (1u << b);
        int c = 0;
#pragma unroll
        for (int i = 0; i < MAXCH; i++) c += __popcll(__ballot(o[i] < tr));
        if (c < k) T = tr;
    }
    int c_le = 0;
#pragma unroll
    for (int i = 0; i < MAXCH; i++) c_le += __popcll(__ballot(o[i] <= T));
    if (c_le != k) return -1;  // several entries share the k-th key
    int base = 0;
#pragma unroll
    for (int i = 0; i < MAXCH; i++) {
        const bool keep = o[i] <= T;
        const uint64_t m = __ballot(keep);
        if (keep) {
            const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
            ord[pos] = o[i];
            id[pos] = d[i];
        }
        base += __popcll(m);
    }
    kth = T;
    return k;
}

template <int MAXCH>
__device__ __forceinline__ int select_pool(uint32_t *ord, int64_t *id, int n, int k, int lane, uint32_t &kth) {
    n = __builtin_amdgcn_readfirstlane(n);
    if (n > k && select_core<MAXCH>(ord, id, n, k, lane, kth) == k) return k;
    const int nn = compact_pool<MAXCH>(ord, id, n, k, lane);  // n <= k, or a tie on the k-th key
    kth = nn >= k ? ord[k - 1] : 0xFFFFFFFFu;
    return nn;
}
