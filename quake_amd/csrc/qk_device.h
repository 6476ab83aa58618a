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
__device__ __forceinline__ int compact_pool(uint32_t *ord, int64_t *id, int n, int k, int lane) {
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
    if (MAXCH == 1 || n <= 64) {  // (n is wave-uniform)
        // n <= 64: every entry lives in one lane; broadcast entry t with v_readlane (SGPR operands, no LDS round trip
        // per iteration) -- ~3x faster than re-reading the pool from LDS, and this is the cold-start cost of every segment
        const uint32_t dlo = (uint32_t)(uint64_t)d[0], dhi = (uint32_t)((uint64_t)d[0] >> 32);
        for (int t = 0; t < n; t++) {
            const uint32_t ot = __builtin_amdgcn_readlane(o[0], t);
            const uint32_t tlo = __builtin_amdgcn_readlane(dlo, t);
            const uint32_t thi = __builtin_amdgcn_readlane(dhi, t);
            const int64_t it = (int64_t)(((uint64_t)thi << 32) | tlo);
            const bool less = (ot < o[0]) || (ot == o[0] && (it < d[0] || (it == d[0] && t < lane)));
            rk[0] += less ? 1 : 0;
        }
    } else {
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
