// qk_dense_pf.hip -- top-k of every query against ONE list (the coarse step with nprobe > 1, the flat index) WITHOUT a key matrix.
//
// Replaces the parent search of QueryCoordinator::search (src/cpp/src/query_coordinator.cpp:628-644 -> batched_scan_list,
// src/cpp/include/list_scanning.h:313-366: knn_L2sqr / knn_inner_product + heap) for 2 <= k <= 64 and d <= 128.
//
// k_dense_ord + k_select_rows (qk_dense.hip) compute every key on fp32 MFMA, write the [Q][n] key matrix and read it back twice:
// 0.14-0.30 of the fp32 MFMA peak, 367 us for 1024 queries x 65536 centroids.  Only k of a query's n keys matter, and WHICH ones
// can be settled at bf16 precision with a one-sided error bound (the prefilter of the scan's hot items, qk_scan_rl.hip):
//
//   k_pf_gemm<MINIMA>  approximate keys on v_mfma_f32_16x16x32_bf16 (operands converted in registers / at staging, 16x the fp32
//                      rate); each lane keeps, per query, the minimum of an UPPER bound of the exact key over its share of the rows.
//                      The rows are thereby cut into G disjoint groups per query, and the k-th smallest of the G group minima has k
//                      rows at or under it: a valid bound of the k-th best key, tight to k/G.
//   k_pf_tau           one wave per query: k-th smallest of its G group minima (bisection on the key bits)
//   k_pf_gemm<FILTER>  the same products again; a row whose LOWER bound is at or under the query's bound is a candidate: its row
//                      number goes to the query's candidate list (one atomic per candidate; ~1.5 k candidates in all per query
//                      would overflow the list: then the query falls back to all rows, exactly)
//   k_pf_finish        one wave per query: the EXACT key of every candidate -- one k-ordered fmaf chain per lane, the arithmetic
//                      of the MFMA path -- and the top k under (key, id).
// Every returned key is exact and every row that could be in the answer is a candidate, so ids and distance bits equal those of
// the key-matrix path.  Error bound: qk_scan_rl.hip (QK_PF_C): |x~.y~ - x.y| <= c (|x|^2 + |y|^2) / 2.
#include "qk_internal.h"
#include "qk_device.h"

#include <algorithm>

typedef __bf16 pf_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 pf_bf16x4 __attribute__((ext_vector_type(4)));
constexpr float PF_C = 0.0078741f;   // >= 2^-7 * 129/128 + 2^-21 (see qk_scan_rl.hip: bf16 keeps 8 significant bits)
constexpr int PF_CAP = 1536;         // candidate rows per query (a multiple of 64)
constexpr int PF_NQ_MAX = 16;        // query tiles per workgroup: 256 queries share one pass over the rows
constexpr int PF_WBUF = 128;         // candidates a wave parks in LDS before it hands them over

struct PfParams {
    const float4 *vecs;
    const float *norms;
    int64_t row_off;
    int nrows;
    const float4 *xq4;  // [Q][nblk][4] fragment-ordered queries
    const float *xn;
    int64_t Q;
    int nq_tiles;       // query tiles per workgroup (<= PF_NQ_MAX)
    int tiles_per_wg;   // row tiles per workgroup (a multiple of 4)
    int groups;         // G = 4 x workgroups along the rows
    uint32_t *gmin;     // [Q][G] group minima of the upper-bound key (MINIMA)
    const uint32_t *tau;  // [Q] (FILTER)
    int32_t *cand;      // [Q][PF_CAP] candidate rows (FILTER)
    int32_t *ccnt;      // [Q] candidates found (may exceed PF_CAP: overflow)
};

// MODE 0: group minima; MODE 1: filter
template <int NB, bool L2, int MODE>
__global__ __launch_bounds__(256) void k_pf_gemm(PfParams P) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int NM = (NB + 1) / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int NQ = P.nq_tiles;
    uint4 *sBh = (uint4 *)smem;                                       // [NQ][NM][64]
    float *xn_s = (float *)(smem + (size_t)NQ * NM * 1024);           // [NQ*16]
    uint32_t *tau_s = (uint32_t *)(xn_s + NQ * 16);                   // [NQ*16]
    // FILTER: candidates are parked in LDS first and handed to the queries' lists together, one atomic round trip for up to
    // 64 of them (an atomic per candidate inside the products was ~45 dependent round trips per wave: +20 us)
    int *cbuf_n = (int *)(tau_s + NQ * 16);                           // [4] entries per wave
    int2 *cbuf = (int2 *)(cbuf_n + 4) + (size_t)wave * PF_WBUF;      // [4][PF_WBUF] (query, row)
    if (MODE == 1 && lane == 0) cbuf_n[wave] = 0;
    const int64_t q_base = (int64_t)blockIdx.x * (NQ * 16);

    // stage the workgroup's queries in bf16, B-operand lane order: step m of a tile = blocks 2m, 2m+1 of the fp32 fragments
    // (eight pieces requested per round trip: a loop of single dependent loads was most of the kernel at 4096 rows)
    for (int t0 = wave; t0 < NQ * NB; t0 += 32) {
        float4 f[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int t = t0 + 4 * u;
            const int nq = t / NB, cb = t - nq * NB;
            const int64_t row = q_base + nq * 16 + j;
            f[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < NQ * NB && row < P.Q) f[u] = P.xq4[(row * NB + cb) * 4 + g];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int t = t0 + 4 * u;
            if (t < NQ * NB) {
                const int nq = t / NB, cb = t - nq * NB;
                pf_bf16x4 hv = {(__bf16)f[u].x, (__bf16)f[u].y, (__bf16)f[u].z, (__bf16)f[u].w};
                *((uint2 *)(sBh + ((size_t)nq * NM + (cb >> 1)) * 64 + lane) + (cb & 1)) = __builtin_bit_cast(uint2, hv);
                if ((NB & 1) && cb == NB - 1) *((uint2 *)(sBh + ((size_t)nq * NM + (cb >> 1)) * 64 + lane) + 1) = make_uint2(0u, 0u);
            }
        }
    }
    for (int t = tid; t < NQ * 16; t += 256) {
        const int64_t row = q_base + t;
        xn_s[t] = row < P.Q ? P.xn[row] : 0.0f;
        if (MODE == 1) {
            // the bound in the float domain: "no bound" (all ones) = +inf / -inf, so that the test below is one compare per product
            const uint32_t tj = row < P.Q ? P.tau[row] : 0u;
            const float tf = L2 ? (tj == 0xFFFFFFFFu ? __builtin_inff() : __uint_as_float(tj))
                                : (tj == 0xFFFFFFFFu ? -__builtin_inff() : ip_from_ord(tj));
            tau_s[t] = __float_as_uint(tf);
        }
    }
    __syncthreads();

    const int ntile_all = (P.nrows + 15) >> 4;
    const int wg_t0 = blockIdx.y * P.tiles_per_wg;
    const int wg_t1 = min(ntile_all, wg_t0 + P.tiles_per_wg);
    // MINIMA: per query the smallest UPPER bound of a key over the lane's rows, kept as a float (L2: the bound itself, minimum;
    // IP: the lower bound of the dot product, maximum) and turned into a key once, at the end
    float mn[PF_NQ_MAX];
#pragma unroll
    for (int i = 0; i < PF_NQ_MAX; i++) mn[i] = L2 ? __builtin_inff() : -__builtin_inff();

    // wave w takes row tiles wg_t0 + w, + 4, ...; the next tile's fragments are requested before this one's products
    float4 a_cur[NB], a_nxt[NB];
    float4 y_cur = make_float4(0.f, 0.f, 0.f, 0.f), y_nxt = y_cur;
    auto load_tile = [&](int tl, float4 *A, float4 &Y) {
        const int64_t ta = (P.row_off >> 4) + min(tl, ntile_all - 1);
        const float4 *src = P.vecs + ta * (NB * 64) + lane;
#pragma unroll
        for (int c = 0; c < NB; c++) A[c] = src[c * 64];
        Y = ((const float4 *)(P.norms + (ta << 4)))[g];
    };
    int tl = wg_t0 + wave;
    if (tl < wg_t1) load_tile(tl, a_cur, y_cur);
    for (; tl < wg_t1; tl += 4) {
        if (tl + 4 < wg_t1) load_tile(tl + 4, a_nxt, y_nxt);
        pf_bf16x8 ah[NM];
#pragma unroll
        for (int m = 0; m < NM; m++) {
            const float4 f0 = a_cur[2 * m];
            const float4 f1 = 2 * m + 1 < NB ? a_cur[2 * m + 1 < NB ? 2 * m + 1 : 0] : make_float4(0.f, 0.f, 0.f, 0.f);
            ah[m] = (pf_bf16x8){(__bf16)f0.x, (__bf16)f0.y, (__bf16)f0.z, (__bf16)f0.w, (__bf16)f1.x, (__bf16)f1.y, (__bf16)f1.z, (__bf16)f1.w};
        }
        const float yv[4] = {y_cur.x, y_cur.y, y_cur.z, y_cur.w};
        const int row0 = (tl << 4) + 4 * g;
        bool rv[4];
        float yk[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            rv[r] = row0 + r < P.nrows;
            // L2 key bounds: (|x|^2 + |y|^2)(1 +- c) - 2 x~.y~ ; IP: the dot product -+ c (|x|^2 + |y|^2) / 2 (key order is descending dot)
            yk[r] = L2 ? yv[r] * (MODE == 0 ? 1.0f + PF_C : 1.0f - PF_C) : yv[r] * (0.5f * PF_C);
            // MINIMA: a row beyond the list never wins (its bound is +inf / its dot product -inf)
            if (MODE == 0 && !rv[r]) yk[r] = __builtin_inff();
        }
#pragma unroll
        for (int nq = 0; nq < PF_NQ_MAX; nq++) {
            if ((nq & 3) == 0 && nq >= NQ) break;  // (NQ is a multiple of 4: four independent products per uniform test)
            const uint4 *bq = sBh + (size_t)nq * NM * 64 + lane;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < NM; m++) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], __builtin_bit_cast(pf_bf16x8, bq[m * 64]), acc, 0, 0, 0);
            const float xnj = xn_s[nq * 16 + j];
            const float xk = L2 ? xnj * (MODE == 0 ? 1.0f + PF_C : 1.0f - PF_C) : xnj * (0.5f * PF_C);
            if (MODE == 0) {
                // (two VALU per product: fma + min; clamping at 0 and the key conversion wait for the end)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    if (L2) mn[nq] = fminf(mn[nq], __fmaf_rn(-2.0f, acc[r], xk + yk[r]));
                    else mn[nq] = fmaxf(mn[nq], acc[r] - (xk + yk[r]));
                }
            } else {
                // one compare per product in the float domain (negated, so that a NaN stays a candidate), one uniform branch per
                // query tile: candidates are rare (tens per query out of thousands of rows)
                const float tauf = __uint_as_float(tau_s[nq * 16 + j]);
                bool p[4];
#pragma unroll
                for (int r = 0; r < 4; r++)
                    p[r] = L2 ? !(__fmaf_rn(-2.0f, acc[r], xk + yk[r]) > tauf) : !(acc[r] + (xk + yk[r]) < tauf);
                const int64_t q = q_base + nq * 16 + j;
                if (__ballot((p[0] | p[1] | p[2] | p[3]) && q < P.Q)) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        if (rv[r] && q < P.Q && p[r]) {
                            const int pos = atomicAdd(&cbuf_n[wave], 1);
                            if (pos < PF_WBUF) {
                                cbuf[pos] = make_int2((int)(q - q_base), row0 + r);
                            } else {  // (buffer full: straight to the list)
                                const int slot = atomicAdd(&P.ccnt[q], 1);
                                if (slot < PF_CAP) P.cand[q * PF_CAP + slot] = row0 + r;
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < NB; c++) a_cur[c] = a_nxt[c];
        y_cur = y_nxt;
    }
    if (MODE == 1) {
        const int n = min(cbuf_n[wave], PF_WBUF);
        for (int e = lane; e < n; e += 64) {
            const int2 c = cbuf[e];
            const int64_t q = q_base + c.x;
            const int slot = atomicAdd(&P.ccnt[q], 1);
            if (slot < PF_CAP) P.cand[q * PF_CAP + slot] = c.y;
        }
    }
    if (MODE == 0) {
        // one group per (workgroup along the rows, wave): the lanes g = 0..3 of a query are folded first
#pragma unroll
        for (int nq = 0; nq < PF_NQ_MAX; nq++) {
            if ((nq & 3) == 0 && nq >= NQ) break;
            // the float bound as a key: L2 clamped at +0 (a negative bound of a squared distance), "no row seen" = all ones
            const float f = mn[nq];
            uint32_t v = L2 ? (f == __builtin_inff() ? 0xFFFFFFFFu : ord_from_l2(f > 0.0f ? f : 0.0f))
                            : (f == -__builtin_inff() ? 0xFFFFFFFFu : ord_from_ip(f));
            v = min(v, (uint32_t)__shfl_xor((int)v, 16));
            v = min(v, (uint32_t)__shfl_xor((int)v, 32));
            const int64_t q = q_base + nq * 16 + j;
            if (g == 0 && q < P.Q) P.gmin[q * P.groups + (int64_t)blockIdx.y * 4 + wave] = v;
        }
    }
}

// one wave per query: tau[q] = k-th smallest of gmin[q][0..G) (0xFFFFFFFF when fewer than k groups saw a row: everything passes);
// also clears the query's candidate counter
template <int PER>
__global__ __launch_bounds__(64) void k_pf_tau(const uint32_t *__restrict__ gmin, int G, int k, uint32_t *__restrict__ tau, int32_t *__restrict__ ccnt) {
    const int lane = threadIdx.x;
    const int64_t q = blockIdx.x;
    const uint32_t *row = gmin + q * G;
    uint32_t v[PER];  // G <= 64 PER
#pragma unroll
    for (int i = 0; i < PER; i++) v[i] = lane + 64 * i < G ? row[lane + 64 * i] : 0xFFFFFFFFu;
    uint32_t T = 0;
    for (int b = 31; b >= 0; b--) {
        const uint32_t tr = T | (1u << b);
        int c = 0;
#pragma unroll
        for (int i = 0; i < PER; i++) c += __popcll(__ballot(v[i] < tr));
        if (c < k) T = tr;
    }
    if (lane == 0) {
        tau[q] = T;  // the largest value with fewer than k minima below it = the k-th smallest (all-ones if there are not k)
        ccnt[q] = 0;
    }
}

struct PfFinish {
    const float4 *vecs;
    const float *norms;
    const int64_t *ids;   // arena ids + row_off
    int64_t row_off;
    int nrows;
    const float *x;       // [Q][d] row-major queries
    const float *xn;
    int d;
    const int32_t *cand;
    const int32_t *ccnt;
    const uint32_t *tau;  // [Q] the bound of the filter pass: at least k rows have an exact key at or under it
    const float *rm;      // [nrows][d] row-major copy of the list (RM form), or nullptr
    int k, Cm;
    int sqrt_l2;
    int64_t *out_ids;
    float *out_dist;
};

// RM: the candidate rows are read from the store's row-major copy (512 contiguous bytes at d = 128, four cache lines) instead of
// the tile-major arena (32 pieces of 16 bytes in 32 lines: ~100 candidates x 1024 queries pulled ~400 MB through the L2 for 52 MB of
// operands); the chain runs over the columns in the same natural order, so the key is the same bits
template <int NB, bool L2, int MAXCH, bool RM>
__global__ __launch_bounds__(64) void k_pf_finish(PfFinish F) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    const int64_t q = blockIdx.x;
    const int k = F.k, Cm = F.Cm;
    int64_t *pool_id = (int64_t *)smem;
    uint32_t *pool_ord = (uint32_t *)(smem + (size_t)Cm * 8);
    float *xs = (float *)(smem + (size_t)Cm * 12);  // [NB*16] the query, zero-padded
    // the first batch of candidates is requested together with the query and the count (one round trip, not three)
    const int cand0 = F.cand[q * PF_CAP + lane];
    const int found = F.ccnt[q];
    const float xnq = F.xn[q];
    for (int c = lane; c < NB * 16; c += 64) xs[c] = c < F.d ? F.x[q * F.d + c] : 0.0f;
    __syncthreads();
    const bool all_rows = found > PF_CAP;  // the list overflowed: every row is a candidate (exact, slow, rare)
    const int n = all_rows ? F.nrows : found;
    // the pool starts behind the filter's bound -- k rows are known to have exact keys at or under it -- so the slack of the
    // lower-bound test (rows listed because their BOUND was small enough) never reaches the pool: one selection at the end
    // instead of one per 64 candidates
    uint32_t tau = F.tau[q];
    int cnt = 0;
    for (int base = 0; base < n; base += 64) {
        const int e = base + lane;
        const bool has = e < n;
        int row = has ? (all_rows ? e : (base == 0 ? cand0 : F.cand[q * PF_CAP + e])) : 0;
        row = min(max(row, 0), F.nrows - 1);  // (lanes beyond the count hold whatever the list held: keep their loads in range)
        const int64_t idr = F.ids[row];        // (with the row data, not behind it)
        const int64_t arow = F.row_off + row;
        const int64_t tile = arow >> 4;
        const int r = (int)(arow & 15);
        float acc = 0.0f;
        // the lane's row, one 16-column block at a time: float4 g of block c holds columns 16c + {g, 4+g, 8+g, 12+g}; the chain
        // runs over the columns in natural order (the canonical arithmetic)
        if (RM) {
            const float4 *rp = (const float4 *)(F.rm + (int64_t)row * F.d);  // (d is a multiple of 4 in this form)
            const int nv = F.d >> 2;
#pragma unroll
            for (int c = 0; c < NB; c++) {
                float4 v[4];
#pragma unroll
                for (int i = 0; i < 4; i++) v[i] = 4 * c + i < nv ? rp[4 * c + i] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    acc = __fmaf_rn(v[i].x, xs[16 * c + 4 * i], acc);
                    acc = __fmaf_rn(v[i].y, xs[16 * c + 4 * i + 1], acc);
                    acc = __fmaf_rn(v[i].z, xs[16 * c + 4 * i + 2], acc);
                    acc = __fmaf_rn(v[i].w, xs[16 * c + 4 * i + 3], acc);
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < NB; c++) {
                const float4 *blk = F.vecs + (tile * NB + c) * 64 + r;
                const float4 v0 = blk[0], v1 = blk[16], v2 = blk[32], v3 = blk[48];
                const float e16[16] = {v0.x, v1.x, v2.x, v3.x, v0.y, v1.y, v2.y, v3.y, v0.z, v1.z, v2.z, v3.z, v0.w, v1.w, v2.w, v3.w};
#pragma unroll
                for (int t = 0; t < 16; t++) acc = __fmaf_rn(e16[t], xs[16 * c + t], acc);
            }
        }
        const float yn = F.norms[arow];
        uint32_t o = L2 ? ord_from_l2(l2_expanded(xnq, yn, acc)) : ord_from_ip(acc);
        if (!has) o = 0xFFFFFFFFu;
        const bool pass = has && o <= tau;
        const uint64_t m = __ballot(pass);
        if (m) {
            if (pass) {
                const int slot = cnt + __popcll(m & ((1ull << lane) - 1ull));
                pool_ord[slot] = o;
                pool_id[slot] = idr;
            }
            cnt += __popcll(m);
            if (cnt > Cm - 64) {
                uint32_t kth;
                cnt = select_pool<MAXCH>(pool_ord, pool_id, cnt, k, lane, kth);
                if (cnt >= k) tau = min(tau, kth);
            }
        }
    }
    cnt = compact_pool<MAXCH>(pool_ord, pool_id, cnt, k, lane);
    for (int e = lane; e < k; e += 64) {
        const bool has = e < cnt;
        F.out_ids[q * k + e] = has ? pool_id[e] : -1;
        if (F.out_dist) {
            float dv;
            if (L2) {
                const float d2 = has ? __uint_as_float(pool_ord[e]) : __builtin_inff();
                dv = (has && F.sqrt_l2) ? sqrtf(d2) : d2;
            } else {
                dv = has ? ip_from_ord(pool_ord[e]) : -__builtin_inff();
            }
            F.out_dist[q * k + e] = dv;
        }
    }
}

// ---- host ---------------------------------------------------------------------------------------------------------------------
template <int NB, bool L2>
static int pf_launch(qk_ctx *ctx, const PfParams &p0, const PfFinish &f0, dim3 grid, size_t lds, int G, int Cm) {
    hipStream_t st = ctx->stream;
    PfParams p = p0;
    QK_HIP(hipFuncSetAttribute((const void *)k_pf_gemm<NB, L2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    QK_HIP(hipFuncSetAttribute((const void *)k_pf_gemm<NB, L2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_pf_gemm<NB, L2, 0>), grid, dim3(256), lds, st, p);
#define PF_TAU(PER_) hipLaunchKernelGGL((k_pf_tau<PER_>), dim3((unsigned)p.Q), dim3(64), 0, st, p.gmin, G, f0.k, const_cast<uint32_t *>(p.tau), p.ccnt)
    if (G <= 256) PF_TAU(4);
    else if (G <= 512) PF_TAU(8);
    else if (G <= 1024) PF_TAU(16);
    else PF_TAU(32);
#undef PF_TAU
    hipLaunchKernelGGL((k_pf_gemm<NB, L2, 1>), grid, dim3(256), lds, st, p);
    const size_t lds_f = (size_t)Cm * 12 + (size_t)NB * 16 * 4;
    if (f0.rm) {
        if (Cm <= 128) hipLaunchKernelGGL((k_pf_finish<NB, L2, 2, true>), dim3((unsigned)p.Q), dim3(64), lds_f, st, f0);
        else hipLaunchKernelGGL((k_pf_finish<NB, L2, 4, true>), dim3((unsigned)p.Q), dim3(64), lds_f, st, f0);
    } else {
        if (Cm <= 128) hipLaunchKernelGGL((k_pf_finish<NB, L2, 2, false>), dim3((unsigned)p.Q), dim3(64), lds_f, st, f0);
        else hipLaunchKernelGGL((k_pf_finish<NB, L2, 4, false>), dim3((unsigned)p.Q), dim3(64), lds_f, st, f0);
    }
    QK_HIP(hipGetLastError());
    return QK_OK;
}
template <int NB>
static int pf_launch_m(qk_ctx *ctx, int metric, const PfParams &p, const PfFinish &f, dim3 grid, size_t lds, int G, int Cm) {
    return metric == QK_METRIC_L2 ? pf_launch<NB, true>(ctx, p, f, grid, lds, G, Cm) : pf_launch<NB, false>(ctx, p, f, grid, lds, G, Cm);
}

struct PfPlan {
    int NQ, tiles_per_wg, rchunks, G;
    int64_t qgroups;
};
static bool pf_plan(const qk_ctx *ctx, const qk_store *s, int64_t Q, int nrows, int k, PfPlan *pl) {
    // (k = 1: the caller asks from 32768 rows on.  The workspace is ~14 KB per query -- group minima + a 1536-slot candidate list --:
    //  batches beyond 65536 queries stay on the sliced key-matrix path / the fused argmin, whose state is a few bytes per query)
    // (64 < k <= 192 on long rows: the key-matrix path selects with one wave per query over the whole row -- 1M rows, k = 100: 19 ms,
    //  4M rows: 268 ms; the filter's bound, candidate lists and finish are the same code up to Cm = 256)
    const int k_max = nrows > 8192 ? 192 : 64;
    // (few queries against a long list: the key-matrix path's selection is one wave per query -- 1M rows, 1 query: k = 10 206 us,
    //  k = 100 3.0 ms; the filter's launches do not care how many queries there are)
    const int64_t q_min = nrows >= 32768 ? 1 : 64;
    if (!(s->nblk <= 8 && k >= 1 && k <= k_max && nrows >= 1024 && Q >= q_min && Q <= 65536)) return false;
    const int num_cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    const int ntile = (nrows + 15) / 16;
    // query tiles per workgroup: up to 256 queries share a pass over the rows (each workgroup streams its rows from L2 / the
    // Infinity Cache once: 1024 queries x 65536 centroids = 4 passes over 33 MB instead of 16)
    // (staging a query tile costs about what two row tiles of products cost: with few rows fewer queries per workgroup, so that
    //  a workgroup still gets >= 2 row tiles per staged query tile; always a multiple of 4 tiles)
#ifdef QK_PF_NQ_CAP
    constexpr int nq_cap_ = QK_PF_NQ_CAP;
#else
    constexpr int nq_cap_ = PF_NQ_MAX;
#endif
    int nq = (int)std::min<int64_t>(nq_cap_, ((Q + 63) / 64) * 4);
    if (nq > 8 && nq < 16) nq = 8;  // (4, 8 or 16 tiles: 12 measured 30 % slower than either neighbour at 32768-65536 rows)
    if (nq > 4 && nq < 8) nq = 4;
    while (nq > 4 && (int64_t)ntile * ((Q + nq * 16 - 1) / (nq * 16)) < (int64_t)2 * num_cus * 2 * nq) nq /= 2;
    pl->NQ = nq;
    pl->qgroups = (Q + pl->NQ * 16 - 1) / (pl->NQ * 16);
    // workgroups along the rows: fill the chip twice over, at least 4 row tiles (one per wave) each, at most 2048 groups per query
    const int64_t want = std::max<int64_t>(1, ((int64_t)2 * num_cus + pl->qgroups - 1) / pl->qgroups);
    pl->tiles_per_wg = qk_round_up(std::max<int64_t>(4, (ntile + want - 1) / want), 4);
    pl->rchunks = (ntile + pl->tiles_per_wg - 1) / pl->tiles_per_wg;
    while (pl->rchunks * 4 > 2048) {
        pl->tiles_per_wg *= 2;
        pl->rchunks = (ntile + pl->tiles_per_wg - 1) / pl->tiles_per_wg;
    }
    pl->G = pl->rchunks * 4;
    while (pl->G < 4 * k && pl->tiles_per_wg > 4) {  // more, smaller groups when k asks for them
        pl->tiles_per_wg = qk_round_up(pl->tiles_per_wg / 2, 4);
        pl->rchunks = (ntile + pl->tiles_per_wg - 1) / pl->tiles_per_wg;
        pl->G = pl->rchunks * 4;
    }
    return pl->G >= 4 * k && pl->G <= 2048;  // (fewer groups: the bound is loose and the candidate lists overflow -- the key-matrix path stays)
}
bool qk_dense_pf_supported(const qk_ctx *ctx, const qk_store *s, int64_t Q, int nrows, int k) {
    PfPlan pl;
    return pf_plan(ctx, s, Q, nrows, k, &pl);
}

// top-k of every query over the rows [row_off, row_off + nrows) of one list; x: [Q][d] row-major queries (device)
int qk_dense_pf_device(qk_ctx *ctx, qk_store *s, int64_t row_off, int nrows, const qk_scan_args &a) {
    const int64_t Q = a.Q;
    const int k = a.k, nblk = s->nblk;
    PfPlan pl;
    if (!pf_plan(ctx, s, Q, nrows, k, &pl)) QK_FAIL(QK_ERR_UNSUPPORTED, "dense top-k (prefiltered): unsupported shape");
    const int NQ = pl.NQ, tiles_per_wg = pl.tiles_per_wg, rchunks = pl.rchunks, G = pl.G;
    const int64_t qgroups = pl.qgroups;
    const int NM = (nblk + 1) / 2;
    const size_t lds = (size_t)NQ * NM * 1024 + (size_t)NQ * 16 * 8 + 16 + (size_t)4 * PF_WBUF * 8 + 64;

    const int Cm = qk_round_up(k + 64, 64);
    size_t need = 0;
    auto add = [&](size_t b) { need += (b + 255) & ~(size_t)255; };
    add((size_t)Q * G * 4);
    add((size_t)Q * 4);
    add((size_t)Q * 4);
    add((size_t)Q * PF_CAP * 4);
    QK_TRY(qk_ws_reserve(ctx, need + 4096));
    uint32_t *gmin = (uint32_t *)qk_ws_alloc(ctx, (size_t)Q * G * 4);
    uint32_t *tau = (uint32_t *)qk_ws_alloc(ctx, (size_t)Q * 4);
    int32_t *ccnt = (int32_t *)qk_ws_alloc(ctx, (size_t)Q * 4);
    int32_t *cand = (int32_t *)qk_ws_alloc(ctx, (size_t)Q * PF_CAP * 4);
    if (!gmin || !tau || !ccnt || !cand) QK_FAIL(QK_ERR_OOM, "dense top-k: workspace exhausted");
    PfParams p;
    p.vecs = (const float4 *)s->vecs;
    p.norms = s->norms;
    p.row_off = row_off;
    p.nrows = nrows;
    p.xq4 = a.xq4;
    p.xn = a.xn;
    p.Q = Q;
    p.nq_tiles = NQ;
    p.tiles_per_wg = tiles_per_wg;
    p.groups = G;
    p.gmin = gmin;
    p.tau = tau;
    p.cand = cand;
    p.ccnt = ccnt;
    PfFinish f;
    f.vecs = (const float4 *)s->vecs;
    f.norms = s->norms;
    f.ids = s->ids + row_off;
    f.row_off = row_off;
    f.nrows = nrows;
    f.x = a.x;
    f.xn = a.xn;
    f.d = s->d;
    f.cand = cand;
    f.ccnt = ccnt;
    f.tau = p.tau;
    f.rm = nullptr;
    if (s->d % 4 == 0) QK_TRY(qk_store_rowmajor(s, row_off, nrows, &f.rm));
    f.k = k;
    f.Cm = Cm;
    f.sqrt_l2 = a.sqrt_l2 ? 1 : 0;
    f.out_ids = a.out_ids;
    f.out_dist = a.out_dist;
    const dim3 grid((unsigned)qgroups, (unsigned)rchunks);
    switch (nblk) {
        case 1: return pf_launch_m<1>(ctx, a.metric, p, f, grid, lds, G, Cm);
        case 2: return pf_launch_m<2>(ctx, a.metric, p, f, grid, lds, G, Cm);
        case 3: return pf_launch_m<3>(ctx, a.metric, p, f, grid, lds, G, Cm);
        case 4: return pf_launch_m<4>(ctx, a.metric, p, f, grid, lds, G, Cm);
        case 5: return pf_launch_m<5>(ctx, a.metric, p, f, grid, lds, G, Cm);
        case 6: return pf_launch_m<6>(ctx, a.metric, p, f, grid, lds, G, Cm);
        case 7: return pf_launch_m<7>(ctx, a.metric, p, f, grid, lds, G, Cm);
        case 8: return pf_launch_m<8>(ctx, a.metric, p, f, grid, lds, G, Cm);
    }
    return QK_ERR_UNSUPPORTED;
}
