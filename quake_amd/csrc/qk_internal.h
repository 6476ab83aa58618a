// qk_internal.h -- shared declarations of libquake_hip.so (not part of the C ABI).
#pragma once
#include "qk_idmap.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <functional>
#include <string>
#include <vector>
#include <unordered_map>

#include "../../include/quake_hip.h"

// ---- error plumbing -------------------------------------------------------------------------
void qk_set_error(const char *fmt, ...);

#define QK_HIP(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            qk_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return QK_ERR_HIP;                                                                \
        }                                                                                     \
    } while (0)

#define QK_TRY(call)               \
    do {                           \
        int s_ = (call);           \
        if (s_ != QK_OK) return s_; \
    } while (0)

#define QK_FAIL(code, ...)        \
    do {                          \
        qk_set_error(__VA_ARGS__); \
        return (code);            \
    } while (0)

// ---- probe switches -------------------------------------------------------------------------------------------------------
// The product library has no environment switches: every QK_* knob the kernels were tuned with is a compile-time constant
// (its measured default).  A probe build (`QK_BUILD_PROBES=1 python -m quake_amd.build --force`, -DQK_PROBES) turns them
// back into environment lookups -- each read ONCE per process -- and compiles the wave-clock / merge-clock printouts in;
// scripts/ and the gpu_*.sh sweeps need that build.
#ifdef QK_PROBES
#include <cstdlib>
static inline int qk_env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}
static inline bool qk_env_set(const char *name) { return getenv(name) != nullptr; }
#else
static inline constexpr int qk_env_int(const char *, int dflt) { return dflt; }
static inline constexpr bool qk_env_set(const char *) { return false; }
#endif

// ---- geometry of the tile-major arena ----------------------------------
// A "tile" is 16 rows.  Within a tile the d dimension is cut into 16-column blocks; block c of a tile is
// 64 float4 stored in LANE ORDER of v_mfma_f32_16x16x4_f32's A operand:
//     float4 index (tile*nblk + c)*64 + (g*16 + r)  holds row r, columns 16c + {g, 4+g, 8+g, 12+g}
// so one global_load_dwordx4 per lane reads a contiguous 1 KiB and element t of the float4 is the operand
// of MFMA step 4c+t, whose four k-slices (g = 0..3) are columns 16c+4t+g: the dot product accumulates
// in natural column order k = 0..d-1.
constexpr int QK_TILE = 16;
constexpr int QK_WAVES = 4;            // waves per scan workgroup
constexpr int QK_QT = 16;              // queries per query tile (MFMA N)

static inline int qk_round_up(int64_t v, int m) { return (int)(((v + m - 1) / m) * m); }
static inline int64_t qk_round_up64(int64_t v, int64_t m) { return ((v + m - 1) / m) * m; }

// ---- context ---------------------------------------------------------------------------------
struct qk_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t stream_ev = nullptr;  // orders a newly bound stream behind the previous one (qk_ctx_set_stream)
    bool timing = false;
    hipDeviceProp_t prop{};
    // bump-allocated scratch, reset at the start of every API call
    char *ws = nullptr;
    size_t ws_cap = 0;
    size_t ws_off = 0;
    // pinned staging for host<->device result/query traffic
    char *pinned = nullptr;
    size_t pinned_cap = 0;
    // pinned staging of the HOST-buffer entry points (qk_search / qk_scan / qk_coarse with QK_MEM_HOST): the caller's pageable
    // queries are copied here and go to the device in one asynchronous transfer, the answers come back here in asynchronous
    // transfers behind the kernels and are copied out after the call's one synchronisation (a transfer from / to pageable memory
    // blocks the host once per buffer).  Its own buffer: `pinned` above is rewritten inside a call (list table, scalars).
    char *pin_io = nullptr;
    size_t pin_io_cap = 0;
    // device staging for ingest of host data
    char *stage = nullptr;
    size_t stage_cap = 0;
    // fragment-ordered copy of the current query batch + norms (shared by the coarse and the scan stage)
    char *qprep = nullptr;
    size_t qprep_cap = 0;
    unsigned long long *qprep_best64 = nullptr;  // [Q] set to ~0 by the prep kernel; valid while qprep_best64_n == Q
    int64_t qprep_best64_n = 0;
    // The "nothing yet" array of the nearest-centroid kernel exists twice and batches take them in turn: the kernel that folds the
    // query preparation into its own staging (k_dense_argmin<.., FUSE>) cannot clear the array it is about to atomicMin into, so it
    // clears the OTHER one -- the next batch's -- whose last readers (the grouping of the previous batch) are behind it in the stream.
    unsigned long long *best64_buf[2] = {nullptr, nullptr};
    int64_t best64_clean[2] = {0, 0};  // leading entries known to hold ~0 at this point of the stream
    int best64_cur = 0;
    int64_t qprep_layout_Q = -1;
    int qprep_layout_d = -1;
    const char *qprep_layout_base = nullptr;
    // query preparation left to the consumer (qk_prep_queries(.., defer)): launched by qk_prep_flush, or folded into the
    // nearest-centroid kernel
    bool prep_pending = false;
    const float *prep_x = nullptr;
    int64_t prep_Q = 0, prep_zero16 = 0;
    int prep_d = 0;
    char *km_pool = nullptr;         // scratch of the k-means calls (qk_kmeans.hip, KmScratch): grown to the largest call's need
    size_t km_pool_cap = 0;
    bool km_pool_busy = false;
    const char *last_scan_kernel = "";  // form of the last scan launch (qk_ctx_last_scan_kernel)
    // Which form of the partition scan serves a batch shape best is measured, not only modelled (qk_scan.hip, "form feedback"):
    // per (store, Q / 64, nprobe, k, metric) the context keeps the mean device time of a whole scan call -- grouping, scan
    // kernel, merge: HIP events around it, read back without synchronising -- under each of the forms the shape admits.
    struct form_stat {
        uint64_t key = 0;
        float ms[3] = {0.f, 0.f, 0.f};  // 0: 16 x 16 tile form (k_scan, plain or query-sharing), 1: per-wave walk (k_scan_rl), 2: mixed
        int n[3] = {0, 0, 0};           // measurements taken
        int64_t calls = 0, last_use = 0;
        int pending = -1;               // form whose event pair is in flight
        int rr = 0;
        hipEvent_t e0 = nullptr, e1 = nullptr;
    };
    std::vector<form_stat> form_stats;
    int64_t form_clock = 0;
    bool form_feedback = true;          // qk_ctx_set_form_feedback (QK_FORM_FEEDBACK=0 in the environment: off from the start)
    bool form_times_set = false;        // qk_ctx_set_form_times: a harvested measurement reads form_times[form], not the event pair
    float form_times[3] = {0.f, 0.f, 0.f};
    int *overflow_host = nullptr;    // pinned, device-visible: a scan kernel sets it when its record buffer overflows
    int *overflow_dev = nullptr;
    char *qprep_zero = nullptr;      // region cleared by the prep kernel for the scan of the same batch
    size_t qprep_zero_bytes = 0;     // its size; 0 once consumed
    const float4 *qprep_xp4 = nullptr;  // [Q][dpad/4] row-major zero-padded copy of the batch at the head of qprep (qk_scan_rl.hip)
    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // deferred timing (qk_ctx_set_timing(ctx, 2)): per-call event quads, read back by qk_ctx_read_timing
    int timing_mode = 0;  // 0 off, 1 per call (sync), 2 deferred, 3 deferred + scan kernel only
    std::vector<hipEvent_t> ev_free;
    std::vector<hipEvent_t> ev_pending;  // groups of 4: group start, scan start, scan end, merge end
    std::vector<hipEvent_t> ev_pending_coarse;  // groups of 2: coarse start, coarse end
    bool squared_l2 = false;  // L2 entry points return squared distances (sharded path, before the final merge)
    float km_assign_ms = 0.0f, km_update_ms = 0.0f;  // last Lloyd iteration of the last qk_kmeans (qk_kmeans_last_timing)
    int64_t km_rows = 0, km_m = 0;
    std::unordered_map<size_t, int> small_occ;  // k_search_small: resident workgroups per CU by LDS bytes (qk_small.hip)
    char *small_ws = nullptr;  // records + tickets of the one-launch small-batch search (qk_small.hip); tickets stay zero between calls
    // state of an adaptive (recall-target) search: survives the scan calls of its rounds, which recycle `ws`
    char *aps = nullptr;
    size_t aps_cap = 0;
    int64_t scratch_reallocs = 0;      // re-allocations of ws / stage / pinned / qprep / aps so far (qk_store_counters [7])
    int32_t *aps_flags = nullptr;      // host-mapped: one word per round, written by the round's last workgroup (qk_aps.hip)
    int32_t *aps_flags_dev = nullptr;
    // what a recall-target search needs and does not change from call to call (qk_aps.hip): the cap-volume table of this dimension
    // and the parent's id -> arena row map, kept on the device until the dimension / the parent's (uid, version) changes
    double *aps_table = nullptr;
    int aps_table_d = -1;
    int32_t *aps_rowof = nullptr;
    size_t aps_rowof_cap = 0;
    int64_t aps_rowof_n = 0;
    uint64_t aps_rowof_uid = 0, aps_rowof_version = 0;
    // XCD balance of the partition scan (qk_scan.hip): relative speed of the 8 workgroup classes blockIdx % 8 per store,
    // learned from the wave times of sampled launches (the physical placement of an arena makes some XCDs stream it up to
    // 25 % slower than others); one sample in flight at a time
    struct xcd_state {
        double w[8] = {1, 1, 1, 1, 1, 1, 1, 1};
        long long launches = 0;
        int samples = 0;  // samples folded in so far
    };
    std::unordered_map<uint64_t, xcd_state> xcd;  // by store uid
    unsigned long long *xcd_host = nullptr;       // pinned [16]: ticks per class, waves per class
    hipEvent_t xcd_ev = nullptr;
    bool xcd_pending = false;
    uint64_t xcd_key = 0;
    double xcd_wsnap[8] = {1, 1, 1, 1, 1, 1, 1, 1};
};

int qk_ws_reserve(qk_ctx *ctx, size_t bytes);          // make sure the workspace can hold `bytes` (may sync+realloc)
void *qk_ws_alloc(qk_ctx *ctx, size_t bytes);          // 256-B aligned slice; nullptr if exhausted
int qk_pinned_reserve(qk_ctx *ctx, size_t bytes);
int qk_stage_reserve(qk_ctx *ctx, size_t bytes);
int qk_aps_reserve(qk_ctx *ctx, size_t bytes);

// ---- store -----------------------------------------------------------------------------------
struct qk_part {
    int64_t row_off = 0;  // first arena row (multiple of 16)
    int64_t size = 0;     // valid rows
    int64_t cap = 0;      // reserved rows (multiple of 16)
    bool present = false;
    std::vector<int64_t> ids;  // host mirror of the ids, in row order
};

struct qk_store {
    qk_ctx *ctx = nullptr;
    uint64_t uid = 0;  // unique per store object of the process
    int d = 0;
    int dpad = 0;  // d rounded up to 16
    int nblk = 0;  // dpad / 16
    float *vecs = nullptr;     // tile-major arena, cap_rows * dpad floats
    float *norms = nullptr;    // [cap_rows] squared norms (canonical chain)
    int64_t *ids = nullptr;    // [cap_rows]
    int64_t cap_rows = 0;
    int64_t used_rows = 0;     // bump pointer
    int64_t dead_rows = 0;     // rows in abandoned extents
    std::vector<qk_part> parts;  // indexed by list number
    int64_t nlist = 0;
    int64_t ntotal = 0;
    int64_t max_size = 0;      // upper bound of the largest partition size (monotone; refreshed on table sync)
    int64_t n_nonempty = 0;    // lists that hold rows (as of the last table sync): mean list length for the scan-form rule
    int64_t max_id_seen = -1;  // upper bound of every id ever stored (monotone): gates the 32-bit id packing of k_dense_argmin
    int64_t min_id_seen = 0;   // lower bound (negative ids disable the packing)
    // device partition table, indexed by list number
    int64_t *d_off = nullptr;
    int32_t *d_size = nullptr;
    int64_t table_cap = 0;
    bool table_dirty = true;
    // what mutations cost beyond the rows they were asked to write (qk_store_counters): a harness attributes a slow add / remove /
    // maintenance step to these -- [0] arena re-allocations (grow: new arena + copy of everything + free), [1] compactions,
    // [2] list relocations (a list outgrew its extent), [3] rows copied by [0]-[2], [4] row-major copy rebuilds, [5] table uploads,
    // [6] rebuilds of the id -> list index (lazy: the first remove / get after a bulk build walks every id); [7] is the context's:
    // re-allocations of its scratch buffers (workspace, staging, pinned, query prep: hipFree + hipMalloc behind a synchronisation)
    int64_t counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t version = 0;  // bumped by every table sync that found the store changed: replicas (qk_group.hip) compare it
    // row-major copy of ONE list's vectors ([rows][d] floats), built on demand for the exact finish of the coarse step without key
    // matrix (qk_dense_pf.hip: a candidate row of the tile-major arena is 32 pieces of 16 bytes in 32 cache lines); dropped whenever
    // the table changes
    // bounce buffers of the in-place arena compaction (compact_arena, qk_store.hip): ~1 GiB of rows, allocated by the first
    // compaction and kept
    float *bounce_v = nullptr, *bounce_n = nullptr;
    int64_t *bounce_i = nullptr;
    int64_t bounce_rows = 0;
    float *rowmajor = nullptr;
    int64_t rowmajor_cap = 0;      // floats
    int64_t rowmajor_row_off = -1, rowmajor_rows = 0;
    bool rowmajor_valid = false;
    // id -> list number (lazy; makes get_vector / remove_ids proportional to the touched lists)
    QkIdMap id_to_list;
    std::vector<uint64_t> kill_bits;  // scratch of remove_ids: one bit per id of [min_id_seen, max_id_seen], zero between calls
    bool index_valid = false;
};
void qk_store_ensure_index(qk_store *s);
int qk_store_add_batch_host_assign(qk_store *s, int64_t n, const int64_t *ids_dev, const float *vecs_dev, const std::vector<int64_t> &h_assign,
                                   bool ids_indexed = false);
int qk_store_remove_list_ex(qk_store *s, int64_t list_no, bool keep_index);

int qk_store_sync_table(qk_store *s);                  // upload (row_off, size) if dirty
// qk_store_build_csr in three steps, with an ownership filter (qk_group.hip: member `rem` of `mod` members takes the lists
// p % mod == rem; the others are not created and their rows are skipped): begin lays the lists out and uploads the tables,
// chunk ingests source rows [i0, i0 + n) of the CSR numbering from DEVICE-readable memory (no synchronisation), end waits and
// publishes the partition table.
struct qk_csr_build {
    int64_t nlist = 0, total = 0;
    int64_t *d_offsets = nullptr, *d_part_row = nullptr;
};
int qk_store_csr_begin(qk_store *s, int64_t nlist, const int64_t *offsets_host, const int64_t *ids_host, int mod, int rem,
                       qk_csr_build *b);
int qk_store_csr_chunk(qk_store *s, const qk_csr_build *b, const float *vecs_dev, const int64_t *ids_dev, int64_t n, int64_t i0);
int qk_store_csr_end(qk_store *s, qk_csr_build *b, int rc);
constexpr int64_t QK_ROWMAJOR_MAX_BYTES = (int64_t)128 << 20;  // 262144 rows x 128: every parent the coarse forms are built for
int qk_store_rowmajor(qk_store *s, int64_t row_off, int nrows, const float **out);  // the copy above for rows [row_off, +nrows)
int qk_store_reserve_rows(qk_store *s, int64_t rows);  // grow the arena so that used_rows + rows fits

// ---- kernels exposed across translation units --------------------------------------------------
// ingest: row-major src [n][d] (device) -> arena rows [row0, row0+n) (tile-major) + norms (+ids if given)
int qk_launch_ingest(qk_ctx *ctx, const float *src, const int64_t *src_ids, int64_t n, int d, int nblk, float *vecs,
                     float *norms, int64_t *ids, int64_t row0);
// gather rows back: arena rows listed in rows[] (device, nullptr = row0..row0+n) -> row-major dst [n][d]
int qk_launch_extract(qk_ctx *ctx, const float *vecs, int nblk, int d, int64_t row0, const int64_t *rows, int64_t n,
                      float *dst);
// in-place row moves inside the arena: row dst[i] <- row src[i] (disjoint sets)
int qk_launch_move_rows(qk_ctx *ctx, float *vecs, float *norms, int64_t *ids, int nblk, const int64_t *dst,
                        const int64_t *src, int64_t n);

struct qk_scan_args {
    const float *x = nullptr;     // [Q][d] device, row-major
    int64_t Q = 0;
    const int64_t *pids = nullptr;  // [Q][P] device; nullptr with all_lists => every present list for every query
    int P = 0;
    int k = 0;
    int metric = QK_METRIC_L2;
    int64_t *out_ids = nullptr;   // [Q][k] device
    float *out_dist = nullptr;    // [Q][k] device (may be nullptr)
    bool all_lists = false;
    bool share_tau = true;
    bool sqrt_l2 = true;
    bool record_events = false;  // record the per-call phase events even when no qk_timing is passed
    // per_pair only: learn every query's bound from a sample of its FIRST list (pids[q][0]) -- valid for each of its pairs when the
    // pairs are consumed in the order given (qk_aps.hip: nothing worse than the first list's k-th best of a sample can be in a
    // running top-k from the first list on); replaces tau_init for the call
    bool seed_first = false;
    int form_salt = 0;           // form feedback: calls of one shape but different work (the rounds of a recall-target search) keep their figures apart
    bool per_pair = false;       // keep the P results of a query apart: out_* are [Q*P][k], one top-k per (query, list)
    // per_pair only: [Q] initial bound per query as ~ord (0 = none): entries worse than it are dropped in every list
    const uint32_t *tau_init = nullptr;
    // key emission (internal, the k > QK_MAX_K path): no top-k, every (pair, row) key is written to key_out[pair_base[pair] + row]
    uint32_t *key_out = nullptr;
    const int64_t *pair_base = nullptr;
    // nearest-list shortcut (nprobe = 1): the coarse stage hands over k_dense_argmin's result array as it is -- key << 32 | list
    // number per query, ~0 = none -- and the grouping / seeding kernels read the list number from it (P = 1, pids unused);
    // the conversion kernel and its launch boundary are skipped.  packed_out: filled by the coarse stage when it can do so.
    const unsigned long long *pids_packed = nullptr;
    const unsigned long long **packed_out = nullptr;
    const float4 *xq4 = nullptr;  // [Q][nblk][4] fragment-ordered queries (qk_prep_queries), required
    const float *xn = nullptr;    // [Q] squared norms, required
};
// x[Q][d] -> ctx->qprep (xq4 then xn); returns the two device pointers
// zero_bytes > 0: the kernel also clears that many bytes for the scan of this batch (qk_scan_zero_bytes)
int qk_prep_queries(qk_ctx *ctx, const float *x, int64_t Q, int d, const float4 **xq4, const float **xn, size_t zero_bytes = 0,
                    bool defer = false);
int qk_prep_flush(qk_ctx *ctx, bool force = false);
size_t qk_scan_zero_bytes(int64_t npids, int64_t Q);
// fails (once) if a scan launched earlier on this context dropped records; called at API entry and after synchronising calls
int qk_check_overflow(qk_ctx *ctx);
int qk_merge_topk_device(qk_ctx *ctx, const int64_t *in_ids, const float *in_key, int G, int64_t Q, int k, int metric,
                         int64_t *out_ids, float *out_dist, bool sqrt_l2);
size_t qk_topk_block_bytes_(int64_t per, int k);
int qk_pack_topk_device(qk_ctx *ctx, const int64_t *ids, const float *key, int G, int64_t per, int k, void *packed);
int qk_merge_topk_packed_device(qk_ctx *ctx, const void *packed, int G, int64_t per, int k, int metric, int64_t *out_ids,
                                float *out_dist, bool sqrt_l2);
int qk_scan_device(qk_ctx *ctx, qk_store *s, const qk_scan_args &a, qk_timing *timing, int ev_base);
// the body of qk_coarse / qk_scan / qk_search (qk_api.hip) and the read-back of its scalars; see there
int qk_run_search(qk_ctx *ctx, qk_store *parent, qk_store *s, const float *x, int64_t Q, const int64_t *pids, int P, int nprobe,
                  int k, int metric, int64_t *out_ids, float *out_dist, int mem, qk_timing *timing, bool coarse_only,
                  bool defer_finish, int64_t *probed_out = nullptr);
int qk_finish_timing(qk_ctx *ctx, qk_store *s, qk_timing *t, bool have_coarse, int scan_ev_base);
// adaptive (recall-target) search: the rounds run on `ctx`; a round's (query, list) pairs are scanned by `scan` (qk_aps.hip)
struct qk_aps_round {
    const float *x = nullptr;           // [Q][d] queries (device, on the context that runs the rounds)
    const float4 *xq4 = nullptr;        // that context's prepared copies
    const float *xn = nullptr;
    int64_t Q = 0;
    const int64_t *round_pids = nullptr;  // [Q][CH] partitions of the round, -1 = none
    int CH = 0, k = 0, metric = 0;        // CH = THIS round's row length (the first round's is shorter than the later ones')
    int CH_max = 0;                       // the longest row any round of this call has (buffers are sized once)
    int64_t *pr_ids = nullptr;          // out [Q*CH][k] per-pair top-k ids (-1 padding)
    float *pr_key = nullptr;            // out [Q*CH][k] squared L2 / inner product
    const uint32_t *run_tau = nullptr;  // [Q] initial bound per query (~ord, 0 = none)
    int round = 0;
};
typedef std::function<int(const qk_aps_round &)> qk_aps_scan_fn;
int qk_aps_run(qk_ctx *ctx, qk_store *parent, int64_t nlist, int d, const float *x, int64_t Q, int k, int metric, float recall_target,
               float recompute_threshold, int use_precomputed, float initial_search_fraction, int64_t *out_ids, float *out_dist,
               int32_t *out_nscanned, int mem, qk_timing *timing, const qk_aps_scan_fn &scan);
// one-launch search of a small batch (qk_small.hip)
bool qk_small_supported(qk_ctx *ctx, qk_store *parent, qk_store *s, int64_t Q, int nprobe, int k);
// coarse step of a mid-sized batch in one launch (qk_small.hip: k_coarse_small)
bool qk_coarse_small_supported(const qk_store *s, int64_t Q, int nrows, int k);
int qk_launch_coarse_small(qk_ctx *ctx, qk_store *s, int64_t row_off, int nrows, const float *x, int64_t Q, int k, int metric,
                           bool sqrt_l2, int64_t *out_ids, float *out_dist);
int qk_search_small_device(qk_ctx *ctx, qk_store *parent, qk_store *s, const float *x, int64_t Q, int nprobe, int k, int metric,
                           int64_t *out_ids, float *out_dist, bool sqrt_l2);
// dense form (every query x one list, Q large): distance matrix on MFMA + per-query select.  qk_dense.hip
int qk_dense_device(qk_ctx *ctx, qk_store *s, int64_t list_no, const qk_scan_args &a, qk_timing *timing, int ev_base);
// top-k of every query over one list without a key matrix: bf16 prefilter + exact finish (qk_dense_pf.hip; 2 <= k <= 64, d <= 128)
bool qk_dense_pf_supported(const qk_ctx *ctx, const qk_store *s, int64_t Q, int nrows, int k);
int qk_dense_pf_device(qk_ctx *ctx, qk_store *s, int64_t row_off, int nrows, const qk_scan_args &a);
// nearest centroid of many rows behind a bf16 prefilter (qk_assign_pf.hip); cnorm = the canonical norms of c's rows
bool qk_assign_pf_supported(int64_t n, int64_t m, int d, int metric);
int qk_assign_pf_device(qk_ctx *ctx, const float *x, int64_t n, const float *c, int64_t m, int d, int metric, const float *cnorm,
                        int64_t *assign, float *val);
// the same launches on caller-provided scratch (qk_assign_pf_scratch_bytes), optionally with ids that order ties and are returned,
// and k_dense_argmin's packed word (key << 32 | id) as the output: the nearest-list search of many rows (qk_dense_device, k = 1)
size_t qk_assign_pf_scratch_bytes(int64_t m, int d);
int qk_assign_pf_launch(qk_ctx *ctx, const float *x, int64_t n, const float *c, int64_t m, int d, int metric, const float *cnorm,
                        const int64_t *ids, int64_t *assign, float *val, unsigned long long *packed_out, bool want_keys, void *scratch);
// top-k of every query over one list of a few thousand rows, keys and selection in one launch (qk_dense_fused.hip; 2 <= k <= 64, d <= 128)
bool qk_dense_fused_supported(const qk_ctx *ctx, const qk_store *s, int64_t Q, int nrows, int k);
int qk_dense_fused_device(qk_ctx *ctx, qk_store *s, int64_t row_off, int nrows, const qk_scan_args &a);
int qk_launch_merge_slices(qk_ctx *ctx, const int64_t *sl_ids, const uint32_t *sl_ord, int64_t nq, int slices, int k, int metric,
                           bool sqrt_l2, int64_t *out_ids, float *out_dist);
// k > QK_MAX_K over several lists: emit every key (qk_scan_device in emission mode), then exact selection per query.  qk_dense.hip
int qk_widek_device(qk_ctx *ctx, qk_store *s, const qk_scan_args &a, qk_timing *timing, int ev_base);
constexpr int QK_MAX_WIDE_K = 8192;  // = the reference's TOP_K_BUFFER_CAPACITY (list_scanning.h:39)

// phase events of one pipeline run: per-call mode (ctx->ev[ev_base..]) and/or deferred mode (parked in the ctx)
struct qk_phase_events {
    qk_ctx *ctx = nullptr;
    bool tm = false, dtm = false;
    bool scan_only = false;  // timing mode 3: one event pair around the scan kernel, nothing else (every event record costs
                             // the stream a few microseconds: 8 per search are 10 % of a 0.35 ms step)
    int ev_base = 0;
    hipEvent_t dev[4] = {nullptr, nullptr, nullptr, nullptr};
    int begin(qk_ctx *c, bool per_call, int base) {
        ctx = c;
        tm = per_call;
        dtm = c->timing_mode == 2 || c->timing_mode == 3;
        scan_only = c->timing_mode == 3;
        ev_base = base;
        if (scan_only && base != 4) dtm = false;
        if (dtm) {
            for (int i = 0; i < 4; i++) {
                if (scan_only && (i == 0 || i == 3)) continue;
                if (!c->ev_free.empty()) {
                    dev[i] = c->ev_free.back();
                    c->ev_free.pop_back();
                } else {
                    QK_HIP(hipEventCreate(&dev[i]));
                }
            }
        }
        return QK_OK;
    }
    int mark(int i) {
        if (tm) QK_HIP(hipEventRecord(ctx->ev[ev_base + i], ctx->stream));
        if (dtm && dev[i]) QK_HIP(hipEventRecord(dev[i], ctx->stream));
        if (dtm && i == 3) {
            if (ev_base == 4) {
                for (int t = 0; t < 4; t++) ctx->ev_pending.push_back(dev[t]);
            } else {  // coarse stage: keep (start, end)
                ctx->ev_pending_coarse.push_back(dev[0]);
                ctx->ev_pending_coarse.push_back(dev[3]);
                ctx->ev_free.push_back(dev[1]);
                ctx->ev_free.push_back(dev[2]);
            }
        }
        return QK_OK;
    }
};
