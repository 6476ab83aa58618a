// qk_scan_types.h -- structures shared by the partition-scan kernels (qk_scan.hip, qk_scan_rl.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

constexpr int QK_SLOTS = 32;  // ints per pair slot line: count + 31 record ids

// ---- grouping ---------------------------------------------------------------------------------------
// everything a wave needs to start on an active partition, in one 32-byte load
struct __align__(16) ActiveInfo {
    long long toff;     // first tile of this partition's items in the global tile sequence
    long long row_off;  // first arena row
    int p;              // list number
    int size;           // rows
    int cnt;            // queries probing it
    int qoff;           // offset of its group in grouped_q / grouped_pair
};

// ---- hot lists (qk_scan_rl.hip, HOT form) -----------------------------------------------------------------------------------
// A list probed by cnt >= min queries of the batch is MFMA-bound whatever form scans it: it leaves the
// per-wave sequence and becomes nqblk x nrr items -- query block b of <= hq queries (hq / 16 query tiles of
// v_mfma_f32_16x16x4_f32 staged in LDS in B-operand order, shared by the four waves of a workgroup) x row range r -- sized to
// about `unit` (row tile x query tile) products each, so that the queue of items stays fine-grained next to a ~0.4 ms launch.
constexpr int QK_HOT_NRR_MAX = 16;  // most row ranges a (list, query block) is cut into: bounds the records of a hot pair
struct HotCost {
    int min;    // lists with cnt >= min are hot (0: none)
    int hq;     // queries per block: multiple of 16, <= 128, what the workgroup's LDS holds (queries + pools)
    int unit;   // cost of an item the cut aims at, in units of the per-wave sequence (summed over the four waves)
    int w10;    // tenths of a unit per (row tile x query tile) product: the bf16 prefilter + its test
    int ht10;   // tenths of a unit per row tile: streaming it (8 KB at d = 128) -- an item costs the larger of the two
    int ovh;    // cost of starting an item (staging the block's queries, emitting its records)
    int C;      // pool capacity per query of a block (k + slack; the block's pools sit next to fp32 AND bf16 query tiles)
    int min_rows;  // ... and only lists of at least this many rows: an item has ~10 us of fixed cost (staging its query block,
                   // claiming, emitting records), a shorter list is cheaper in passes of the per-wave walk
};
__host__ __device__ inline bool hot_list(int cnt, int size, const HotCost &h) { return h.min > 0 && cnt >= h.min && size >= h.min_rows; }
struct HotShape {
    int nqblk, qpb, nrr;  // query blocks, queries per block (multiple of 16), row ranges
};
__host__ __device__ inline long long hot_block_cost(int ntl, int qt, const HotCost &h) {
    const long long a = (long long)ntl * h.ht10, b = (long long)ntl * qt * h.w10;
    return ((a > b ? a : b) + 9) / 10;
}
__host__ __device__ inline HotShape hot_shape(int cnt, int size, const HotCost &h) {
    HotShape s;
    const int ntl = (size + 15) >> 4;
    s.nqblk = (cnt + h.hq - 1) / h.hq;
    s.qpb = (((cnt + s.nqblk - 1) / s.nqblk) + 15) & ~15;
    const long long cost = hot_block_cost(ntl, s.qpb >> 4, h);
    long long nrr = (cost + h.unit / 2) / h.unit;
    const long long cap = (ntl + 7) >> 3;  // ranges are whole groups of 8 row tiles (hot_range)
    if (nrr > cap) nrr = cap;
    if (nrr < 1) nrr = 1;
    if (nrr > QK_HOT_NRR_MAX) nrr = QK_HOT_NRR_MAX;
    s.nrr = (int)nrr;
    return s;
}
// row tiles [lo, hi) of range r: the list is cut in groups of 8 row tiles -- 4 waves x one pair of tiles -- so that every wave of
// the workgroup gets the same number of pairs (a ragged cut left one wave with a pair more than the others in most items:
// 6 % of the item spent waiting at its closing barrier)
__host__ __device__ inline void hot_range(int size, int r, int nrr, int *lo, int *hi) {
    const int ntl = (size + 15) >> 4;
    const long long ng = (ntl + 7) >> 3;
    const long long a = (ng * r) / nrr * 8, b = (ng * (r + 1)) / nrr * 8;
    *lo = (int)(a < ntl ? a : ntl);
    *hi = (int)(b < ntl ? b : ntl);
}
__host__ __device__ inline long long hot_units_of(int cnt, int size, const HotCost &h) {
    const HotShape s = hot_shape(cnt, size, h);
    const int ntl = (size + 15) >> 4;
    long long u = 0;
    for (int b = 0; b < s.nqblk; b++) {
        const int nq = cnt - b * s.qpb < s.qpb ? cnt - b * s.qpb : s.qpb;
        if (nq <= 0) break;
        u += (long long)s.nrr * h.ovh + hot_block_cost(ntl, (nq + 15) >> 4, h);
    }
    return u;
}

struct ScanParams {
    const float4 *vecs;
    const float *norms;
    const int64_t *ids;
    const int64_t *pt_off;
    const int32_t *pt_size;
    int nblk;
    const float4 *xq4;
    const float *xn;
    const int32_t *grouped_q;
    const int32_t *grouped_pair;
    const int32_t *n_active;
    const ActiveInfo *active;
    const int64_t *n_tiles;
    uint32_t *gtau;  // [Q] shared running bound per query stored as ~bound (0 = none, so one memset clears it), or nullptr
    int tau_refresh;  // re-read gtau every 8 tiles (only useful when a query probes several partitions)
    int tau_publish;  // waves publish their bound into gtau (0: gtau is a read-only initial bound, qk_scan_args::tau_init)
    int k;
    int C;  // pool capacity per query; k <= C - 4
    int metric;
    int32_t *pair_head;
    int32_t *pair_slots;
    int32_t *rec_counter;
    int32_t max_recs;
    int *overflow;       // host-mapped flag of the context: set when a record does not fit (max_recs is meant to be unreachable)
    int2 *rec_hdr;       // [max_recs] {next record of the pair (-1 = end), entry count}
    uint32_t *rec_ord;   // [max_recs][k]
    int64_t *rec_id;     // [max_recs][k]
    // dynamic tail (one wave per workgroup only): the last dyn_tiles_pct % of the tile sequence is handed out in chunks of
    // dyn_chunk tiles through this counter once a wave has finished its static share (nullptr: all static)
    unsigned long long *dyn_counter;
    int dyn_chunk, dyn_pct;
    int pack;      // > 1: independent one-wave workgroups bundled per hardware workgroup (see k_scan)
    int pack_lds;  // LDS bytes of each bundled wave
    // XCD balance: hardware workgroup u belongs to class u % 8 (workgroups go round-robin over the XCDs); its share of the cut
    // is proportional to xcd_w[class] (1024 = average); xcd_stat [16] collects ticks and waves per class (nullptr: not sampled)
    int xcd_on;
    int xcd_w[8];
    unsigned long long *xcd_stat;
    // query-sharing workgroups (narrow rows, many queries per partition): the nw waves of a workgroup walk the SAME tiles
    // of a partition at the same time, each with its own 16-query tile and pools in LDS, so that a partition probed by up
    // to 16*nw queries is fetched from HBM once (the later waves hit in L2).  0: waves split the tiles instead.
    int qshare;
    int seg_ovh;  // as GroupParams::seg_ovh
    // key emission (MODE 4, k > QK_MAX_K): no top-k at all, every (pair, row) key goes to key_out[pair_base[pair] + row]
    uint32_t *key_out;
    const int64_t *pair_base;
    long long *wave_clock;  // probe (QK_SCAN_WAVE_CLOCK): [waves][2] start / end of every wave in wall_clock64 ticks, or nullptr
    // row-per-lane scan (qk_scan_rl.hip)
    const float4 *xp4;  // [Q][dpad/4] row-major zero-padded queries
    int rl_h0, rl_h1, rl_e, rl_m;  // cost model of the work sequence (RlCost; ovh = seg_ovh)
    int rl_qb, rl_app;       // queries per pass; lanes per append round (32: pools of k + 32 entries, 16: k + 16)
    int rl_probe;            // probe (QK_SCAN_RL_PROBE): bit 0 = no top-k epilogue, bit 1 = no MFMA chains (attribution only; uniform branches OUTSIDE the chain)
    // hot lists of the mixed work sequence (k_scan_rl<.., HOT = true>): lists probed by >= hot.min queries are not part of the
    // per-wave sequence; they are cut into ITEMS (query block x row range) that whole workgroups claim from a second queue
    HotCost hot;
    const int32_t *act_hoff;      // [n_active + 1] hot items in front of every active list (sentinel: the total)
    const int32_t *n_hot;         // [1] hot items of the launch
    const long long *hot_units;   // [1] their cost in units of the per-wave sequence
    int32_t *hot_counter;         // [1] next hot item to hand out (zeroed per call)
    int hot_first_pct;            // the hot-first workgroups' share of the grid = hot / (hot + walk) cost with the items' cost taken at
                                  // this percentage: > 100 puts more workgroups on items at the start, so that the ITEMS (47 us
                                  // apiece) run out first and the launch ends in the fine-grained dynamic tail of the per-wave walk
};

// ---- merge stage (qk_merge.hip) ------------------------------------------------------------------------------------------
struct MergeParams {
    int P;
    const int32_t *pair_head;
    const int32_t *pair_slots;
    const int2 *rec_hdr;
    const uint32_t *rec_ord;
    const int64_t *rec_id;
    int32_t max_recs;
    int k;
    int Cm;  // pool capacity, k <= Cm - 64
    int metric;
    int64_t *out_ids;   // [Q][k]
    float *out_dist;    // [Q][k] or nullptr
    int sqrt_l2;        // 1: output sqrt(d2) (search results); 0: squared (merge key of the sharded path)
    long long *clock;   // probe (QK_MERGE_CLOCK): [Q][8] wall_clock64 ticks of the phases of every wave, or nullptr
};

// ---- row-per-lane scan (qk_scan_rl.hip): cost model of the work sequence ----------------------------------------------------
// A partition probed by cnt queries is scanned in passes of up to RlCost::qb queries (qb / 4 groups of 4: one
// v_mfma_f32_4x4x1_16b_f32 serves 64 rows x 4 queries).  A pass walks the partition in chunks of 64 rows; a chunk of a pass
// with g groups weighs max(h, m * ceil(g / 2) + e) units -- h stands for streaming the chunk (h0: from HBM, first pass; h1:
// later passes find it in L2 / Infinity Cache), m for one step of the MFMA loop (two groups' chains interleaved: 256
// instructions + two top-k epilogues; measured 1.44 us against ~4.3 us for a chunk when every wave streams, hence m = 4,
// h0 = 12) -- and every pass starts with `ovh` units (query staging, record emission).  The grouping stage lays the
// partitions end to end in these units; k_scan_rl cuts the head of the sequence statically and hands the tail out dynamically.
constexpr int QK_RL_QB_MAX = 64;  // widest pass: the slot of a query is owned by the lane of its number
constexpr int QK_RL_DYN_MAX = 4096;  // most ranges the dynamic tail is cut into (bounds the records a launch can emit)
struct RlCost {
    int h0, h1, e, ovh, m;
    int qb;  // queries per pass (multiple of 4, <= QK_RL_QB_MAX): what fits in the wave's share of LDS (queries + pools)
};
__host__ __device__ inline int rl_w(int g, bool first, const RlCost &c) {
    const int h = first ? c.h0 : c.h1, v = c.m * ((g + 1) >> 1) + c.e;
    return v > h ? v : h;
}
__host__ __device__ inline long long rl_part_len(int cnt, int size, const RlCost &c) {
    const long long nch = (size + 63) >> 6;
    const int nqb = (cnt + c.qb - 1) / c.qb;
    const int g_last = (cnt - c.qb * (nqb - 1) + 3) >> 2;
    if (nqb <= 1) return c.ovh + nch * rl_w(g_last, true, c);
    return (long long)nqb * c.ovh +
           nch * ((long long)rl_w(c.qb / 4, true, c) + (long long)(nqb - 2) * rl_w(c.qb / 4, false, c) + rl_w(g_last, false, c));
}

// row-per-lane kernel (qk_scan_rl.hip): LDS of one wave's slice / of a hot item's block, and the launch
size_t qk_scan_rl_lds_per_wave(int nblk, int C, int qb);
size_t qk_scan_hot_lds(int nblk, int C, int hq);
int qk_launch_scan_rl(int nblk, dim3 grid, size_t lds, hipStream_t st, const ScanParams &sp);

// ---- the host's plan for one scan call (qk_scan_plan.hip) ---------------------------------------------------------------------
inline bool have_scan_qs(int db, int maxch) { return (db == 8 || db == 4 || db == 2) && (maxch == 1 || maxch == 2 || maxch == 4); }
inline int pick_maxch(int cap) { return cap <= 64 ? 1 : cap <= 128 ? 2 : cap <= 256 ? 4 : 8; }
struct ScanPlan {
    int DB = 1;              // 16-column blocks per load step of k_scan
    int C = 0;               // pool capacity per query
    int nw = 1, qshare = 0;  // waves per workgroup of the tile form; query-sharing workgroups
    bool use_rl = false;     // row-per-lane kernel (forms 1 and 2)
    int form = 0;            // 0 tile form, 1 per-wave walk, 2 mixed sequence
    int64_t rl_per_list = 0; // batch average of probing queries per list (what the static rule looks at)
    RlCost rlc{12, 8, 1, 16, 4, 32};
    int rl_app = 32, rl_waves = 4;
    HotCost hot{0, 0, 0, 0, 0, 0, 0, 0};
    qk_ctx::form_stat *measure = nullptr;  // form feedback: this call is timed (e0 is already on the stream; the caller records e1)
};
int qk_scan_plan(qk_ctx *ctx, qk_store *s, const qk_scan_args &a, bool emit, int k, int P, int64_t npairs, ScanPlan *pl);
