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
// Stable bucketing of the rows by assignment: least-significant-digit passes of RS_BITS bits of the centroid number (12 bits per pass:
// ONE pass up to 4094 centroids, two up to 2^24; 8 bits when the numbers fit), each pass = digit histogram per chunk of rows, one
// exclusive scan over [chunk][digit], stable scatter.  The unit of work is a WAVE: it owns RS_WAVE_ROWS(n) consecutive rows, counts
// and places them with wave-private LDS counters -- a key's place among the equal digits of its 64-row round comes from RS_BITS
// ballots -- so the passes hold no workgroup barrier in their loops (the first version of round 3 ranked 256 keys per round across
// the four waves of a workgroup: three barriers per round, 35-48 us per pass for 2^20 keys).  The four waves of a workgroup share
// one row of the table: the scatter re-counts the chunk (the keys come back from L2) to find each wave's offset inside it, which
// keeps the table at 2^RS_BITS * n / (4 * wave rows) entries.  The first pass reads the assignments themselves (key = assignment,
// value = row number), so no key / value arrays are materialised before it.  Rows whose assignment is out of range are dropped by the
// first pass (no key of their own: 4096 centroids stay ONE 12-bit pass); the number of rows kept travels on the device (n_dev).
constexpr int RS_THREADS = 256, RS_WAVES = 4;

template <bool FIRST>
__device__ __forceinline__ int32_t rs_key(const int64_t *__restrict__ assign, const int32_t *__restrict__ keys_in, int64_t i, int m) {
    if (FIRST) {
        const int64_t a = assign[i];
        return (a < 0 || a >= m) ? -1 : (int32_t)a;  // out of range: the row is dropped by the first pass
    }
    return keys_in[i];
}

struct RsPass {
    const int64_t *assign;    // FIRST pass: the assignments
    const int32_t *keys_in;   // later passes: keys / values of the previous pass
    const int32_t *vals_in;
    int64_t n;                // rows of the first pass (host value); later passes read the rows kept from n_dev
    int32_t *n_dev;           // [1] rows kept = sum of the digits' totals, written by every k_rs_scan
    int m, shift;
    int wave_rows;            // rows per wave (multiple of 64); a workgroup's chunk = 4 * wave_rows consecutive rows
    int nchunks;              // workgroups
    int32_t *table;           // [nchunks][1 << BITS]: counts, then (after k_rs_scan) the first output position of (chunk, digit)
    int32_t *totals;          // [1 << BITS] words, zero before k_rs_scan: word g = 1 + the rows of the digits of scan workgroup g
    int32_t *keys_out;        // may be null on the last pass
    int32_t *vals_out;
    int64_t *dbase_out;       // k_rs_scan: first output position of every digit ([(1 << BITS) + 1], may be null) -- with a single pass
                              // these ARE the segment bounds of the centroids
};

// rows in flight per lane: a workgroup's chunk is 4 waves x 1024 rows on 256 CUs -- ONE wave per SIMD, nothing to hide a load behind --
// so every loop below asks for RS_U rows before it touches the first (round 5 walked them one dependent round trip at a time:
// 16 trips of ~0.7 us per wave in the count and again in the placement)
constexpr int RS_U = 16;

// digit counts of this wave's rows into its private LDS row hw[1 << BITS] (zeroed here)
template <int BITS, bool FIRST>
__device__ __forceinline__ void rs_count_wave(const RsPass &P, int32_t *hw, int64_t r0, int64_t r1, int lane) {
    constexpr int NB = 1 << BITS;
    for (int i = lane; i < NB; i += 64) hw[i] = 0;
    __builtin_amdgcn_wave_barrier();
    for (int64_t i0 = r0; i0 < r1; i0 += 64 * RS_U) {
        int32_t kk[RS_U];
#pragma unroll
        for (int u = 0; u < RS_U; u++) {
            const int64_t i = i0 + u * 64 + lane;
            const int32_t key = rs_key<FIRST>(P.assign, P.keys_in, min(i, r1 - 1), P.m);  // (unconditional: the loads leave together)
            kk[u] = i < r1 ? key : -1;
        }
#pragma unroll
        for (int u = 0; u < RS_U; u++)
            if (kk[u] >= 0) atomicAdd(&hw[(kk[u] >> P.shift) & (NB - 1)], 1);
    }
    __builtin_amdgcn_wave_barrier();
}

template <int BITS, bool FIRST>
__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const RsPass P) {
    constexpr int NB = 1 << BITS;
    extern __shared__ int32_t rs_lds[];  // [NB]: the counts of the whole chunk (the scatter finds the waves' shares itself)
    for (int i = threadIdx.x; i < NB; i += RS_THREADS) rs_lds[i] = 0;
    __syncthreads();
    const int64_t n = FIRST ? P.n : (int64_t)*P.n_dev;
    const int64_t r0 = (int64_t)blockIdx.x * RS_WAVES * P.wave_rows, r1 = min(n, r0 + (int64_t)RS_WAVES * P.wave_rows);
    for (int64_t i0 = r0; i0 < r1; i0 += RS_THREADS * RS_U) {
        int32_t kk[RS_U];
#pragma unroll
        for (int u = 0; u < RS_U; u++) {
            const int64_t i = i0 + u * RS_THREADS + threadIdx.x;
            const int32_t key = rs_key<FIRST>(P.assign, P.keys_in, min(i, r1 - 1), P.m);  // (unconditional: the loads leave together)
            kk[u] = i < r1 ? key : -1;
        }
#pragma unroll
        for (int u = 0; u < RS_U; u++)
            if (kk[u] >= 0) atomicAdd(&rs_lds[(kk[u] >> P.shift) & (NB - 1)], 1);
    }
    __syncthreads();
    // (no digit totals from here: 256 chunks x ~2600 non-empty digits were 665 000 device-scope atomics on 4096 words; the scan adds
    // the columns up anyway)
    for (int i = threadIdx.x; i < NB; i += RS_THREADS) P.table[(int64_t)blockIdx.x * NB + i] = rs_lds[i];
}

// exclusive scan over [chunk][digit] in place, digit-major order (all chunks of digit 0, then digit 1, ...): workgroup g owns the
// 64 digits 64g .. 64g+63 and walks the chunks in RS_SCAN_WAVES contiguous parts (thread = (part, digit): 64 consecutive digits of
// one chunk are one coalesced 256-byte read; 16 parts of 16 chunks at 2^20 rows -- one batch of loads per part and pass).  The
// digits' totals are the column sums of the first walk; the base of the workgroup's first digit is the sum of the totals of the
// workgroups below it, which they publish (total + 1: zero = not yet) in P.totals[g] -- at most 64 workgroups, dispatched in index
// order, every one publishing before it waits.
constexpr int RS_SCAN_WAVES = 16;

template <int BITS>
__global__ __launch_bounds__(RS_SCAN_WAVES * 64) void k_rs_scan(const RsPass P) {
    constexpr int NB = 1 << BITS;
    __shared__ int32_t s_part[RS_SCAN_WAVES];
    __shared__ int32_t s_q[RS_SCAN_WAVES][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int dg0 = blockIdx.x * 64;
    // part `wave` of the chunks
    const int per = (P.nchunks + RS_SCAN_WAVES - 1) / RS_SCAN_WAVES, c0 = min(P.nchunks, wave * per), c1 = min(P.nchunks, c0 + per);
    int32_t *col = P.table + dg0 + lane;
    int32_t qs = 0;
#pragma unroll 16
    for (int c = c0; c < c1; c++) qs += col[(int64_t)c * NB];
    s_q[wave][lane] = qs;
    __syncthreads();
    // exclusive prefix of this workgroup's 64 digit totals (every wave computes it: lane = digit)
    int32_t tot = 0, before = 0;  // the digit's total; its rows in the parts under this wave's
#pragma unroll
    for (int w = 0; w < RS_SCAN_WAVES; w++) {
        const int32_t v = s_q[w][lane];
        before += w < wave ? v : 0;
        tot += v;
    }
    int32_t inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int32_t u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
    }
    // (relaxed on both sides: the word IS the message -- nothing else written by the publisher is read through it, so no cache
    // write-back / invalidate rides on the exchange)
    if (wave == 0 && lane == 63) __hip_atomic_store(&P.totals[blockIdx.x], inc + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int32_t below = 0;  // totals of the digits under dg0
    for (int g = tid; g < (int)blockIdx.x; g += RS_SCAN_WAVES * 64) {
        int32_t v;
        while ((v = __hip_atomic_load(&P.totals[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) __builtin_amdgcn_s_sleep(1);
        below += v - 1;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) below += __shfl_xor(below, o);
    if (lane == 0) s_part[wave] = below;
    __syncthreads();
    int32_t base0 = 0;
#pragma unroll
    for (int w = 0; w < RS_SCAN_WAVES; w++) base0 += s_part[w];
    const int32_t dbase = base0 + inc - tot;
    if (wave == 0 && P.dbase_out) P.dbase_out[dg0 + lane] = dbase;
    if (wave == 0 && lane == 63 && blockIdx.x == gridDim.x - 1) {  // rows kept (all digits) = the end of the last segment
        *P.n_dev = base0 + inc;
        if (P.dbase_out) P.dbase_out[NB] = base0 + inc;
    }
    // second walk: 16 counts asked for together, their running positions written together (a store between two loads would make the
    // wait for the later load a wait for the store: one counter, vmcnt, counts both on gfx950)
    int32_t run = dbase + before;
    for (int cb = c0; cb < c1; cb += 16) {
        int32_t v[16];
#pragma unroll
        for (int t = 0; t < 16; t++) v[t] = col[(int64_t)min(cb + t, c1 - 1) * NB];
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const int32_t cnt = v[t];
            v[t] = run;
            run += cnt;
        }
#pragma unroll
        for (int t = 0; t < 16; t++)
            if (cb + t < c1) col[(int64_t)(cb + t) * NB] = v[t];
    }
}

// stable scatter: the workgroup re-counts its chunk per wave (the waves' offsets inside the chunk's table row), then every wave
// places its rows in rounds of 64 in index order
template <int BITS, bool FIRST>
__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(const RsPass P) {
    constexpr int NB = 1 << BITS;
    extern __shared__ int32_t rs_lds[];  // [4][NB]: counts, then running output positions per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n = FIRST ? P.n : (int64_t)*P.n_dev;
    const int64_t r0 = ((int64_t)blockIdx.x * RS_WAVES + wave) * P.wave_rows, r1 = min(n, r0 + P.wave_rows);
    rs_count_wave<BITS, FIRST>(P, rs_lds + wave * NB, r0, r1, lane);
    __syncthreads();
    for (int i = threadIdx.x; i < NB; i += RS_THREADS) {
        int32_t at = P.table[(int64_t)blockIdx.x * NB + i];
#pragma unroll
        for (int w = 0; w < RS_WAVES; w++) {
            const int32_t c = rs_lds[w * NB + i];
            rs_lds[w * NB + i] = at;
            at += c;
        }
    }
    __syncthreads();
    // the scan's published totals go back to zero for the next pass / call (their only reader, k_rs_scan, ran before this launch)
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < NB; i += RS_THREADS) P.totals[i] = 0;
    int32_t *pos = rs_lds + wave * NB;
    const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int64_t b0 = r0; b0 < r1; b0 += 64 * RS_U) {
        int32_t kk[RS_U], vv[RS_U];
#pragma unroll
        for (int u = 0; u < RS_U; u++) {
            const int64_t i = b0 + u * 64 + lane;
            const int32_t key = rs_key<FIRST>(P.assign, P.keys_in, min(i, r1 - 1), P.m);  // (unconditional: the loads leave together)
            kk[u] = i < r1 ? key : -1;
            vv[u] = FIRST ? (int32_t)i : P.vals_in[min(i, r1 - 1)];
        }
        // (the store of a round leaves inside it: holding the 16 positions back and storing at the end measured 2.4 us slower)
#pragma unroll
        for (int u = 0; u < RS_U; u++) {
            if (b0 + u * 64 >= r1) break;
            const int32_t key = kk[u];
            const bool valid = key >= 0;
            const int digit = (key >> P.shift) & (NB - 1);
            uint64_t peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < BITS; b++) {
                const bool bit = (digit >> b) & 1;
                const uint64_t bal = __ballot(valid && bit);
                peers &= bit ? bal : ~bal;
            }
            const int rank = __popcll(peers & lt);
            int at = 0;
            if (valid) at = pos[digit] + rank;
            __builtin_amdgcn_wave_barrier();
            if (valid && rank == 0) pos[digit] += __popcll(peers);  // (one lane per distinct digit of the round)
            __builtin_amdgcn_wave_barrier();
            if (valid) {
                if (P.keys_out) P.keys_out[at] = key;
                P.vals_out[at] = vv[u];
            }
        }
    }
}

__global__ void k_segment_bounds(const int32_t *__restrict__ sorted_keys, const int32_t *__restrict__ n_dev, int m,
                                 int64_t *seg_begin /*[m+1]*/) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = *n_dev;
    if (i > n) return;
    // seg_begin[c] = first position with key >= c
    int prev = i == 0 ? -1 : sorted_keys[i - 1];
    int cur = i == n ? m : min(sorted_keys[i], m);
    for (int c = prev + 1; c <= cur; c++) seg_begin[c] = i;
}

// REFERENCE ORDER (kmeans_refine_partitions' own accumulate loop, clustering.cpp:162-176: `centroid_sums[c][j] += vec[j]` row after
// row): one workgroup per centroid; thread t owns dimensions t, t + blockDim, ...; rows are added in ascending row order -- one
// dependent chain of adds per (centroid, dimension).  The kernel lasts as long as its largest cluster's chain, so what counts is the
// time per row of ONE workgroup: 32 rows per round trip, the row numbers of the next 32 requested with them.  Used where the
// reference fixes the order (qk_kmeans_accumulate, qk_store_refine_lists); the Lloyd driver, whose order FAISS leaves open, takes
// k_accumulate_blocked below.
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

// BLOCKED ORDER -- the canonical summation order of the Lloyd driver's mean update (faiss::Clustering leaves it to its back end:
// clustering.cpp:51-55 hands the whole iteration to FAISS; qo_kmeans_accumulate_blocked is the same order on the host).  With the
// rows of a centroid in ascending row order r_0 < r_1 < ...:
//   level 1  a BLOCK is KM_L1 = 32 consecutive rows of the bucket; its partial is the sequential fp32 sum  ((0 + x[r_0]) + x[r_1]) + ...
//   level 2  a GROUP is KM_L2 = 32 consecutive blocks (1024 rows); its partial is the sequential sum of its block partials, from 0
//   level 3  the centroid's sum is the sequential sum of its group partials, from 0
// so the longest dependent chain is 32 + 32 + n_c / 1024 adds instead of n_c, and the blocks of one centroid are independent work.
// Grid (m, KM_GLANES): workgroup (c, j) takes the groups j, j + KM_GLANES, ... of centroid c (most centroids have one); inside a
// group `NCH = 256 / TPC` blocks run side by side (TPC threads per block chain, one float4 -- or one float when d % 4 != 0 -- of the
// row per thread), and after each such round the chain-0 threads fold the round's partials in block order.  A centroid with several
// groups leaves the group partials in `gpart` and its last workgroup to arrive (ticket) folds them in group order.
constexpr int KM_L1 = 32, KM_L2 = 32, KM_GROUP = KM_L1 * KM_L2, KM_GLANES = 8;

template <typename V>
__device__ __forceinline__ V km_zero();
template <>
__device__ __forceinline__ float km_zero<float>() { return 0.0f; }
template <>
__device__ __forceinline__ float4 km_zero<float4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float km_add(float a, float b) { return a + b; }
__device__ __forceinline__ float4 km_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float km_ld_agent(const float *p) {
    return __uint_as_float(__hip_atomic_load((const uint32_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ float4 km_ld_agent(const float4 *p) {
    const float *f = (const float *)p;
    return make_float4(km_ld_agent(f), km_ld_agent(f + 1), km_ld_agent(f + 2), km_ld_agent(f + 3));
}
// a training row is read ONCE per update: non-temporal (the partition scan's loads gained 12 % that way)
__device__ __forceinline__ float km_ld_once(const float *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ float4 km_ld_once(const float4 *p) {
    const f32x4 t = __builtin_nontemporal_load((const f32x4 *)p);
    return make_float4(t[0], t[1], t[2], t[3]);
}
// gpart slot of group g of the centroid whose bucket starts at row b: distinct for all (centroid, group) pairs
__device__ __host__ __forceinline__ int64_t km_gslot(int64_t b, int64_t c, int64_t g) { return b / KM_GROUP + c + g; }

template <typename V>  // V = float4 (d % 4 == 0) or float; `units` = d / 4 or d elements of V per row
__global__ __launch_bounds__(256) void k_accumulate_blocked(const V *__restrict__ x, int units, const int32_t *__restrict__ sorted_rows,
                                                            const int64_t *__restrict__ seg_begin, V *__restrict__ sums,
                                                            int64_t *__restrict__ counts, V *__restrict__ gpart,
                                                            unsigned int *__restrict__ tickets) {
    __shared__ V s_part[256];
    __shared__ int s_last;
    const int c = blockIdx.x, gl = blockIdx.y, tid = threadIdx.x;
    const int64_t b = seg_begin[c], e = seg_begin[c + 1], nc = e - b;
    const int64_t G = (nc + KM_GROUP - 1) / KM_GROUP;  // groups of this centroid
    if (gl == 0 && tid == 0) counts[c] = nc;
    if (nc == 0) {
        if (gl == 0)
            for (int u = tid; u < units; u += 256) sums[(int64_t)c * units + u] = km_zero<V>();
        return;
    }
    if (gl >= G) return;
    const int TPC = min(units, 256), NCH = 256 / TPC;  // threads per chain, chains side by side
    const int cs = tid / TPC, j = tid - cs * TPC;
    const bool chain = cs < NCH;
    for (int u0 = 0; u0 < units; u0 += TPC) {  // (one trip unless a row is wider than 256 units)
        const int u = u0 + j;
        const bool live = chain && u < units;
        for (int64_t g = gl; g < G; g += gridDim.y) {
            const int64_t gb = b + g * KM_GROUP, ge = min(e, gb + KM_GROUP);
            const int nb = (int)((ge - gb + KM_L1 - 1) / KM_L1);
            V run = km_zero<V>();
            for (int b0 = 0; b0 < nb; b0 += NCH) {
                const int blk = b0 + cs;
                if (live && blk < nb) {
                    const int64_t rb = gb + (int64_t)blk * KM_L1;
                    const int cnt = (int)min((int64_t)KM_L1, ge - rb);
                    int32_t r[KM_L1];
#pragma unroll
                    for (int t = 0; t < KM_L1; t++) r[t] = sorted_rows[min(rb + t, ge - 1)];
                    V v[KM_L1];
#pragma unroll
#ifdef KM_PLAIN_LOADS
                    for (int t = 0; t < KM_L1; t++) v[t] = x[(int64_t)r[t] * units + u];
#else
                    for (int t = 0; t < KM_L1; t++) v[t] = km_ld_once(x + (int64_t)r[t] * units + u);
#endif
                    V s = km_zero<V>();
                    if (cnt == KM_L1) {
#pragma unroll
                        for (int t = 0; t < KM_L1; t++) s = km_add(s, v[t]);
                    } else {
#pragma unroll
                        for (int t = 0; t < KM_L1; t++) s = t < cnt ? km_add(s, v[t]) : s;
                    }
                    s_part[tid] = s;
                }
                __syncthreads();
                if (cs == 0 && u < units) {
                    const int lim = min(NCH, nb - b0);
                    for (int t = 0; t < lim; t++) run = km_add(run, s_part[t * TPC + j]);
                }
                __syncthreads();
            }
            if (cs == 0 && u < units) {
                if (G == 1) sums[(int64_t)c * units + u] = run;
                else gpart[km_gslot(b, c, g) * units + u] = run;
            }
        }
    }
    if (G == 1) return;
    // several groups: the last workgroup of the centroid to arrive adds the group partials in group order
    __threadfence();
    __syncthreads();
    if (tid == 0) {
        const unsigned int lanes = (unsigned int)min((int64_t)gridDim.y, G);
        const unsigned int t = atomicAdd(&tickets[c], 1u);
        s_last = t == lanes - 1 ? 1 : 0;
        if (s_last) tickets[c] = 0;  // (zero between launches)
    }
    __syncthreads();
    if (!s_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    for (int u = tid; u < units; u += 256) {
        V tot = km_zero<V>();
        for (int64_t g = 0; g < G; g++) tot = km_add(tot, km_ld_agent(&gpart[km_gslot(b, c, g) * units + u]));
        sums[(int64_t)c * units + u] = tot;
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

// Device buffers owned by one qk_kmeans* call.  They come out of a per-context pool (qk_ctx::km_pool) by bump allocation: a
// maintenance call splits ~50 partitions (a 2-means each: ten buffers) and refines ~600 (two copies of 1.5M rows) -- with a hipMalloc
// per buffer and a hipFree per buffer on return (each hipFree drains the device) the allocator was most of a 0.5 ms split and of a
// 130 ms refine.  A request the pool cannot serve falls back to hipMalloc (pointers handed out never move during a call); what the
// call asked for in total is remembered, and the pool is regrown to it when the call returns.  One user at a time (km_pool_busy):
// a nested call allocates the old way.
struct KmScratch {
    static constexpr size_t POOL_MAX = (size_t)8 << 30;
    qk_ctx *ctx;
    bool own_pool = false;
    size_t used = 0, wanted = 0;
    std::vector<void *> ptrs;
    explicit KmScratch(qk_ctx *c) : ctx(c) {
        if (ctx && !ctx->km_pool_busy) {
            ctx->km_pool_busy = true;
            own_pool = true;
        }
    }
    KmScratch(const KmScratch &) = delete;
    KmScratch &operator=(const KmScratch &) = delete;
    ~KmScratch() {
        for (void *p : ptrs)
            if (p) (void)hipFree(p);
        if (!own_pool) return;
        if (wanted > ctx->km_pool_cap && wanted <= POOL_MAX) {
            if (hipStreamSynchronize(ctx->stream) != hipSuccess) (void)hipGetLastError();
            if (ctx->km_pool) (void)hipFree(ctx->km_pool);
            ctx->km_pool = nullptr;
            ctx->km_pool_cap = 0;
            const size_t want = wanted + wanted / 4 + 4096;
            if (hipMalloc((void **)&ctx->km_pool, want) == hipSuccess) ctx->km_pool_cap = want;
            else (void)hipGetLastError();
        }
        ctx->km_pool_busy = false;
    }
    template <typename T>
    int alloc(T **out, size_t count) {
        const size_t bytes = (std::max<size_t>(count, 1) * sizeof(T) + 255) & ~(size_t)255;
        wanted += bytes;
        if (own_pool && used + bytes <= ctx->km_pool_cap) {
            *out = (T *)(ctx->km_pool + used);
            used += bytes;
            return QK_OK;
        }
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            qk_set_error("k-means scratch allocation of %zu bytes failed: %s", bytes, hipGetErrorString(e));
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
    // many rows: the bf16-prefiltered form (same bits, ~6x the rate: qk_assign_pf.hip)
    // (its float4 loads want 16-byte aligned rows: d % 8 == 0 is part of `supported`, the base pointers are checked here)
    if (qk_assign_pf_supported(n, m, d, metric) && ((uintptr_t)x & 15) == 0 && ((uintptr_t)c & 15) == 0)
        return qk_assign_pf_device(ctx, x, n, c, m, d, metric, cnorm, assign, val);
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
    int32_t *keys = nullptr, *vals = nullptr, *keys2 = nullptr, *vals2 = nullptr;  // (keys2, vals2) hold the bucketed rows
    int64_t *seg = nullptr;        // [max(m + 2, 4097)] segment bounds of the centroids in vals2
    int32_t *table = nullptr;      // [nchunks][1 << bits] + [1 << bits] digit totals
    float *gpart = nullptr;        // group partials of the blocked order: [n / KM_GROUP + m + 2][d]
    unsigned int *tickets = nullptr;  // [m], zero between launches
    int wave_rows = 0, nchunks = 0, bits = 0, passes = 0;
};

static void rs_plan(int64_t n, int64_t m, AccumScratch &as) {
    int kb = 1;
    while ((1LL << kb) < m) kb++;  // the keys are 0 .. m - 1 (rows out of range are dropped by the first pass)
    as.bits = kb <= 8 ? 8 : 12;
    as.passes = (kb + as.bits - 1) / as.bits;
    // a wave's share: 1024 rows, more once the table would pass ~64 MB
#ifndef QK_RS_WAVE_ROWS
#define QK_RS_WAVE_ROWS 1024
#endif
    int64_t wr = QK_RS_WAVE_ROWS;
    while (((n + 4 * wr - 1) / (4 * wr)) * ((int64_t)4 << as.bits) > ((int64_t)64 << 20)) wr *= 2;
    as.wave_rows = (int)wr;
    as.nchunks = (int)std::max<int64_t>(1, (n + 4 * wr - 1) / (4 * wr));
}

static int accum_prepare(hipStream_t st, KmScratch &ks, AccumScratch &as, int64_t n, int64_t m, int d) {
    rs_plan(n, m, as);
    QK_TRY(ks.alloc(&as.vals2, (size_t)n));
    if (as.passes > 1) {
        QK_TRY(ks.alloc(&as.keys, (size_t)n));
        QK_TRY(ks.alloc(&as.vals, (size_t)n));
        QK_TRY(ks.alloc(&as.keys2, (size_t)n));
    }
    QK_TRY(ks.alloc(&as.seg, (size_t)std::max<int64_t>(m + 2, ((int64_t)1 << as.bits) + 1)));
    QK_TRY(ks.alloc(&as.table, (((size_t)as.nchunks + 1) << as.bits) + 16));  // + digit totals + n_dev
    QK_TRY(ks.alloc(&as.gpart, (size_t)(n / KM_GROUP + m + 2) * d));
    QK_TRY(ks.alloc(&as.tickets, (size_t)m));
    // (on the stream the kernels run on: it may be a non-blocking one, which a null-stream memset would not order with)
    QK_HIP(hipMemsetAsync(as.tickets, 0, (size_t)m * sizeof(unsigned int), st));
    QK_HIP(hipMemsetAsync(as.table + ((size_t)as.nchunks << as.bits), 0, (sizeof(int32_t) << as.bits) + 16 * sizeof(int32_t), st));  // digit totals: zero between passes
    return QK_OK;
}

template <int BITS>
static int rs_pass_launch(hipStream_t st, const RsPass &P, bool first) {
    constexpr size_t lds = (size_t)RS_WAVES * sizeof(int32_t) << BITS, lds_hist = sizeof(int32_t) << BITS;
    static bool attr_set = false;
    if (!attr_set) {
        QK_HIP(hipFuncSetAttribute((const void *)k_rs_hist<BITS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        QK_HIP(hipFuncSetAttribute((const void *)k_rs_hist<BITS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        QK_HIP(hipFuncSetAttribute((const void *)k_rs_scatter<BITS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        QK_HIP(hipFuncSetAttribute((const void *)k_rs_scatter<BITS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const dim3 grid((unsigned)P.nchunks), block(RS_THREADS);
    if (first) hipLaunchKernelGGL((k_rs_hist<BITS, true>), grid, block, lds_hist, st, P);
    else hipLaunchKernelGGL((k_rs_hist<BITS, false>), grid, block, lds_hist, st, P);
    hipLaunchKernelGGL((k_rs_scan<BITS>), dim3((1u << BITS) / 64), dim3(RS_SCAN_WAVES * 64), 0, st, P);
    if (first) hipLaunchKernelGGL((k_rs_scatter<BITS, true>), grid, block, lds, st, P);
    else hipLaunchKernelGGL((k_rs_scatter<BITS, false>), grid, block, lds, st, P);
    QK_HIP(hipGetLastError());
    return QK_OK;
}

// rows bucketed stably by assignment into as.vals2 (row numbers; as.keys2 holds their keys when there are several passes) and the
// segment bounds of the centroids into as.seg
static int bucket_rows_device(hipStream_t st, AccumScratch &as, const int64_t *assign, int64_t n, int64_t m) {
    RsPass P;
    P.assign = assign;
    P.n = n;
    P.m = (int)m;
    P.wave_rows = as.wave_rows;
    P.nchunks = as.nchunks;
    P.table = as.table;
    P.totals = as.table + ((size_t)as.nchunks << as.bits);
    P.n_dev = P.totals + ((size_t)1 << as.bits);
    // the last pass must land in (keys2, vals2): with an odd number of passes the first one writes there
    int32_t *ko = (as.passes & 1) ? as.keys2 : as.keys, *vo = (as.passes & 1) ? as.vals2 : as.vals;
    const int32_t *ki = nullptr, *vi = nullptr;
    for (int p = 0; p < as.passes; p++) {
        P.keys_in = ki;
        P.vals_in = vi;
        P.shift = as.bits * p;
        P.keys_out = as.passes > 1 ? ko : nullptr;
        P.vals_out = vo;
        P.dbase_out = as.passes == 1 ? as.seg : nullptr;  // one pass: the digit bases are the segment bounds
        if (as.bits == 8) QK_TRY(rs_pass_launch<8>(st, P, p == 0));
        else QK_TRY(rs_pass_launch<12>(st, P, p == 0));
        ki = ko;
        vi = vo;
        ko = ko == as.keys ? as.keys2 : as.keys;
        vo = vo == as.vals ? as.vals2 : as.vals;
    }
    if (as.passes > 1) hipLaunchKernelGGL(k_segment_bounds, dim3(km_grid(n + 1, 256)), dim3(256), 0, st, as.keys2, P.n_dev, (int)m, as.seg);
    QK_HIP(hipGetLastError());
    return QK_OK;
}

// blocked = false: the reference's row-after-row order (k_accumulate); true: the blocked canonical order of the Lloyd driver
static int accumulate_device(qk_ctx *ctx, AccumScratch &as, const float *x, int64_t n, int d, const int64_t *assign, int64_t m,
                             float *sums, int64_t *counts, bool blocked) {
    hipStream_t st = ctx->stream;
    if (n > 0x7FFFFFF0LL) QK_FAIL(QK_ERR_UNSUPPORTED, "qk_kmeans_accumulate: n too large for 32-bit row indices");
    if (m > 0x7FFFFFF0LL) QK_FAIL(QK_ERR_UNSUPPORTED, "qk_kmeans_accumulate: too many centroids for 32-bit keys");
    if (n <= 0) {
        QK_HIP(hipMemsetAsync(sums, 0, (size_t)m * d * sizeof(float), st));
        QK_HIP(hipMemsetAsync(counts, 0, (size_t)m * sizeof(int64_t), st));
        return QK_OK;
    }
    QK_TRY(bucket_rows_device(st, as, assign, n, m));
    if (!blocked) {
        hipLaunchKernelGGL(k_accumulate, dim3((unsigned)m), dim3((unsigned)std::min(256, qk_round_up(d, 64))), 0, st, x, d, as.vals2, as.seg, sums, counts);
    } else {
        // group lanes: workgroups per centroid (most exit at once: a centroid with <= 1024 rows has one group): two at a mean
        // cluster of 256 rows (a skewed training sample then holds clusters of a few groups), eight from ~1000 rows per cluster on;
        // empty workgroups are not free (4096 x 8 of them: ~10 us of a 100 us launch)
        const unsigned lanes = (unsigned)std::min<int64_t>(KM_GLANES, std::max<int64_t>(2, (n + 128 * m - 1) / (128 * m)));
        if (d % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)sums & 15) == 0)
            hipLaunchKernelGGL((k_accumulate_blocked<float4>), dim3((unsigned)m, lanes), dim3(256), 0, st, (const float4 *)x, d / 4, as.vals2, as.seg,
                               (float4 *)sums, counts, (float4 *)as.gpart, as.tickets);
        else
            hipLaunchKernelGGL((k_accumulate_blocked<float>), dim3((unsigned)m, lanes), dim3(256), 0, st, x, d, as.vals2, as.seg, sums, counts,
                               as.gpart, as.tickets);
    }
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
    KmScratch ks(ctx);
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

static int kmeans_accumulate_api(qk_ctx *ctx, const float *x, int64_t n, int d, const int64_t *assign, int64_t m, float *sums,
                                 int64_t *counts, int mem, bool blocked) {
    if (!ctx || !x || !assign || !sums || !counts) QK_FAIL(QK_ERR_INVALID, "qk_kmeans_accumulate: null argument");
    if (n < 0 || m <= 0 || d <= 0) QK_FAIL(QK_ERR_INVALID, "qk_kmeans_accumulate: bad sizes");
    QK_HIP(hipSetDevice(ctx->device));
    KmScratch ks(ctx);
    AccumScratch as;
    QK_TRY(accum_prepare(ctx->stream, ks, as, std::max<int64_t>(n, 1), m, d));
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
    QK_TRY(accumulate_device(ctx, as, dx, n, d, da, m, ds, dc, blocked));
    if (mem == QK_MEM_HOST) {
        QK_HIP(hipMemcpyAsync(sums, bs, (size_t)m * d * 4, hipMemcpyDeviceToHost, ctx->stream));
        QK_HIP(hipMemcpyAsync(counts, bc, (size_t)m * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    QK_HIP(hipStreamSynchronize(ctx->stream));
    return QK_OK;
}

int qk_kmeans_accumulate(qk_ctx *ctx, const float *x, int64_t n, int d, const int64_t *assign, int64_t m, float *sums,
                         int64_t *counts, int mem) {
    return kmeans_accumulate_api(ctx, x, n, d, assign, m, sums, counts, mem, false);
}

int qk_kmeans_accumulate_blocked(qk_ctx *ctx, const float *x, int64_t n, int d, const int64_t *assign, int64_t m, float *sums,
                                 int64_t *counts, int mem) {
    return kmeans_accumulate_api(ctx, x, n, d, assign, m, sums, counts, mem, true);
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
    std::vector<char> seen(s->parts.size(), 0);
    for (int64_t c = 0; c < m; c++) {
        int64_t p = list_nos[c];
        if (p < 0 || p >= (int64_t)s->parts.size() || !s->parts[p].present)
            QK_FAIL(QK_ERR_NOT_FOUND, "List does not exist in refine_partitions (list %lld)", (long long)p);
        if (seen[(size_t)p]) QK_FAIL(QK_ERR_INVALID, "qk_store_refine_lists: duplicate list %lld", (long long)p);
        seen[(size_t)p] = 1;
        total += s->parts[p].size;
    }
    const int iterations = refinement_iterations > 0 ? refinement_iterations : 1;  // clustering.cpp:110
    KmScratch ks(ctx);
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
    QK_TRY(accum_prepare(st, ks, as, std::max<int64_t>(total, 1), m, d));
    std::vector<int64_t> hcounts((size_t)m, 0);
    for (int iter = 0; iter < iterations; iter++) {
        if (iter > 0)  // centroids = sums / counts; a count of 0 gives NaN exactly like the reference (:122-124)
            hipLaunchKernelGGL(k_finalize_centroids, dim3(km_grid(m * d, 256)), dim3(256), 0, st, dsums, dcounts, m, d, 0, dc);
        QK_TRY(assign_device(ctx, xa, total, dc, m, d, metric, dassign, nullptr, ctile, cnorm));
        // (a row without an assignment -- NaN centroid -- is dropped by the bucketing: the tail of the row list it leaves must not
        //  hold garbage for the gathers below; the call fails after the loop in that case)
        if (total > 0) QK_HIP(hipMemsetAsync(as.vals2, 0, (size_t)total * sizeof(int32_t), st));
        QK_TRY(accumulate_device(ctx, as, xa, total, d, dassign, m, dsums, dcounts, false));  // the reference's own order (:162-176)
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
    // (the rows lie grouped by list in list_nos order: ONE ingest for all of them -- an add_entries per list was a launch, a copy of
    //  the ids and a synchronisation each, 0.26 s for the ~3000 lists a maintenance call of a 50M index refines)
    std::vector<int64_t> h_assign((size_t)total);
    pos = 0;
    for (int64_t c = 0; c < m; c++) {
        QK_TRY(qk_store_remove_list_ex(s, list_nos[c], true));  // (the same ids come back below: their index entries are overwritten)
        QK_TRY(qk_store_add_list(s, list_nos[c]));
        std::fill(h_assign.begin() + pos, h_assign.begin() + pos + hcounts[c], list_nos[c]);
        pos += hcounts[c];
    }
    return qk_store_add_batch_host_assign(s, total, ia, xa, h_assign, true);
}

int qk_kmeans(qk_ctx *ctx, float *x, int64_t n, int d, int64_t m, int metric, int niter, uint64_t seed, float *centroids,
              int64_t *assign, int mem) {
    if (!ctx || !x || !centroids || !assign) QK_FAIL(QK_ERR_INVALID, "qk_kmeans: null argument");
    if (n <= 0 || m <= 0 || d <= 0 || m > n) QK_FAIL(QK_ERR_INVALID, "qk_kmeans: bad sizes (n=%lld m=%lld d=%d)", (long long)n, (long long)m, d);
    if (metric != QK_METRIC_L2 && metric != QK_METRIC_IP) QK_FAIL(QK_ERR_INVALID, "Metric type not supported");
    QK_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    KmScratch ks(ctx);
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
    QK_TRY(accum_prepare(st, ks, as, ntrain, m, d));
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
        QK_TRY(accumulate_device(ctx, as, xt, ntrain, d, dta, m, dsums, dcounts, true));  // FAISS leaves the order open: blocked
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
    KmScratch ks(ctx);
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
