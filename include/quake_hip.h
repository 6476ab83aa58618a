/*
 * quake_hip.h -- C ABI of libquake_hip.so: the MI355X (gfx950) implementation of Quake's
 * search / k-means hot path.
 *
 * The reference (marius-team/quake @ 2025-05-23) has NO C ABI or plugin registry: its boundary is the
 * C++ class API + pybind11.  These entry points are what the bodies of the
 * reference's C++ methods would bind when the hot path is delegated to the GPU; each one cites the
 * reference interface it replaces (paths relative to the reference checkout).  The host-side mirror of
 * the C++/Python surface (quake_amd/) is written against exactly this header.
 *
 * Conventions
 *   - plain pointers + sizes, no torch / C++ types; every function returns a qk_status (0 = ok) and
 *     records a message retrievable with qk_last_error() (the C++ side turns it into
 *     std::runtime_error / std::invalid_argument like the reference's throws).
 *   - `mem` says where the caller's data pointers live: QK_MEM_HOST (pageable or pinned host memory,
 *     what the reference's CPU tensors are) or QK_MEM_DEVICE (HBM of the context's device).
 *   - metric codes are faiss::MetricType's: 0 = inner product, 1 = L2 (common.h:145-156).
 *   - L2 results are sqrt distances, like the reference (list_scanning.h:260,286,353-357).
 *   - fewer than k results: ids -1, distances +inf (L2) / -inf (IP) (query_coordinator.cpp:589-601,774-788).
 *   - ordering is the total order (key, id) -- DESIGN.md section 3.
 *   - all work is enqueued on the context's HIP stream; host-memory outputs are complete on return,
 *     device-memory outputs are complete after qk_ctx_synchronize() (or stream order).
 */
#ifndef QUAKE_HIP_H
#define QUAKE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QK_API __attribute__((visibility("default")))

typedef enum {
    QK_OK = 0,
    QK_ERR_INVALID = 1,     /* bad argument (std::invalid_argument in the reference) */
    QK_ERR_NOT_FOUND = 2,   /* "List does not exist" (dynamic_inverted_list.cpp:71,79,87) */
    QK_ERR_HIP = 3,         /* HIP runtime error; also: a scan launched earlier on the context had to drop result records
                             * (its record buffer is sized by a host-side upper bound; the kernels raise a host-visible
                             * flag if that bound is ever wrong) -- reported by the call that synchronises, or the next call */
    QK_ERR_UNSUPPORTED = 4, /* outside the implemented envelope (e.g. k > QK_MAX_K) */
    QK_ERR_OOM = 5
} qk_status;

#define QK_METRIC_IP 0
#define QK_METRIC_L2 1
#define QK_MEM_HOST 0
#define QK_MEM_DEVICE 1
#define QK_MAX_K 448 /* largest k of the fused LDS top-k (pool capacity k+64 <= 512).  Larger k, up to 8192 -- the capacity
                      * of the reference's buffer (list_scanning.h:39) -- is served by emitting every key and selecting afterwards
                      * (slower, exact) */
#define QK_MAX_NPROBE 8192 /* largest nprobe / number of APS candidate partitions: the coarse step selects them with a
                              * bisection select + sort beyond QK_MAX_K (flat parent index) */

typedef struct qk_ctx qk_ctx;     /* device + stream + scratch workspace                     */
typedef struct qk_store qk_store; /* device mirror of faiss::DynamicInvertedLists (one level) */

/* Per-call timing, filled from HIP events; mirrors the fields of SearchTimingInfo (common.h:214-228)
 * that still mean something on a GPU. */
typedef struct {
    float coarse_ms;  /* parent search (query_coordinator.cpp:644)            */
    float group_ms;   /* partition -> query grouping (query_coordinator.cpp:707-721) */
    float scan_ms;    /* partition scan kernel(s)                               */
    float merge_ms;   /* per-query merge + output                               */
    float total_ms;
    int64_t n_items;          /* work items scanned                              */
    int64_t scan_bytes;       /* algorithmic bytes of the scan: sum over unique probed partitions n_p*d*4 */
    int64_t partitions_scanned; /* (query, partition) pairs scanned              */
} qk_timing;

/* ---- errors ------------------------------------------------------------------------------------ */
QK_API const char *qk_last_error(void);
QK_API const char *qk_version(void);

/* ---- context ----------------------------------------------------------------------------------- */
/* device: HIP ordinal.  Creates a private non-blocking stream. */
QK_API int qk_ctx_create(int device, qk_ctx **out);
QK_API int qk_ctx_destroy(qk_ctx *ctx);
/* Run on a caller-owned hipStream_t instead (e.g. torch's current stream); NULL restores the private one. */
QK_API int qk_ctx_set_stream(qk_ctx *ctx, void *hip_stream);
/* Run on the device's NULL (legacy default) stream -- torch's default stream has the handle 0, which qk_ctx_set_stream reads
 * as "restore the private stream". */
QK_API int qk_ctx_set_null_stream(qk_ctx *ctx);
/* The stream the context is bound to now, so that a caller that rebinds it for one call can put the binding back:
 * kind 0 = the private stream (restore with qk_ctx_set_stream(ctx, NULL)), 1 = the NULL stream (qk_ctx_set_null_stream),
 * 2 = a caller-owned stream (*hip_stream; qk_ctx_set_stream(ctx, *hip_stream)). */
QK_API int qk_ctx_get_stream(qk_ctx *ctx, void **hip_stream, int *kind);
/* Form feedback (default on): the partition scan has several forms with identical results (16 x 16 tiles, per-wave row-per-lane
 * walk, mixed sequence with dense hot items); which is fastest depends on how the batch's queries concentrate on lists, which
 * the host cannot see.  With feedback on, a context times whole scan calls per (store, batch shape) with HIP events it reads
 * back later without synchronising, tries every admissible form twice, then uses the fastest and re-checks the others every
 * few hundred calls.  Off: the static rule alone (what the first call of a shape always uses).  No reference counterpart:
 * the reference picks serial / batched / worker scans by SearchParams (query_coordinator.cpp:612-673). */
QK_API int qk_ctx_set_form_feedback(qk_ctx *ctx, int enabled);
/* The feedback RULE on injected figures: with ms3 = {tile form, per-wave walk, mixed sequence} (all > 0) every measurement the
 * context harvests reads ms3[form] in place of the elapsed time of its event pair, so which form answers which call is a pure
 * function of the call sequence (tests/test_scan_feedback_gpu.py asserts the sequence).  NULL: measured times again.  The
 * environment variable QK_FORM_FEEDBACK=0 creates every context with feedback off. */
QK_API int qk_ctx_set_form_times(qk_ctx *ctx, const float *ms3);
QK_API int qk_ctx_synchronize(qk_ctx *ctx);
/* hipEvent timing of the phases, recorded on the context's stream around the kernels:
 *   0 off; 1 per call (the qk_timing* passed to qk_scan/qk_search is filled, which synchronises the stream);
 *   2 deferred (no synchronisation inside the calls; qk_ctx_read_timing sums everything recorded since the last read);
 *   3 deferred, scan kernel only: one event pair per call around the partition-scan kernel (an event record costs the stream
 *     a few microseconds; the 8 of mode 2 add ~10 % to a 0.35 ms search). */
QK_API int qk_ctx_set_timing(qk_ctx *ctx, int mode);
/* The mode set last (0 at creation): a caller that switches the mode for one call restores what it found
 * (QueryCoordinator::search fills SearchTimingInfo, query_coordinator.cpp:612-657, on a context others may be timing with). */
QK_API int qk_ctx_get_timing(qk_ctx *ctx, int *mode);
/* Synchronises, then returns the SUM of the phase durations over the calls recorded in deferred mode and their count. */
QK_API int qk_ctx_read_timing(qk_ctx *ctx, qk_timing *sum, int64_t *calls);
/* Device properties the harness prints: CU count, clock (kHz), total HBM bytes, gcnArchName. */
QK_API int qk_ctx_device_info(qk_ctx *ctx, int *num_cus, int *clock_khz, int64_t *hbm_bytes, char *arch, int arch_len);
/* Name of the partition-scan kernel the last qk_scan / qk_search on this context launched ("k_scan", "k_scan (query-sharing)",
 * "k_scan_rl", "k_scan_rl (mixed)", "k_search_small", "k_dense"; "" before the first call): what a harness labels its kernel timings with. */
QK_API int qk_ctx_last_scan_kernel(qk_ctx *ctx, char *name, int name_len);

/* ---- partition store ---------------------------------------------------------------------------
 * Replaces faiss::DynamicInvertedLists / IndexPartition as the thing the scan reads
 * (dynamic_inverted_list.h:25-33, index_partition.h:19-32, accessors dynamic_inverted_list.cpp:68-90).
 * Observable behaviour kept: append order, swap-with-last remove (index_partition.cpp:79-102).      */
/* A store belongs to the context it was created with (mutations run on that context's stream), but it may be SEARCHED through
 * any context of the same device -- several at once, each on its own stream -- as long as nobody mutates it meanwhile
 * (bench.py's `batches_in_flight` measurement does that: two contexts, one index). */
QK_API int qk_store_create(qk_ctx *ctx, int d, qk_store **out);                 /* DynamicInvertedLists(0, d*4) */
QK_API int qk_store_destroy(qk_store *s);
QK_API int qk_store_reset(qk_store *s);                                         /* reset() :300-304 */
QK_API int qk_store_add_list(qk_store *s, int64_t list_no);                     /* add_list :262-270 */
QK_API int qk_store_remove_list(qk_store *s, int64_t list_no);                  /* remove_list :251-260 */
/* add_entries :152-173 -> IndexPartition::append (index_partition.cpp:52-59).  vecs [n][d] row-major f32. */
QK_API int qk_store_add_entries(qk_store *s, int64_t list_no, int64_t n, const int64_t *ids, const float *vecs, int mem);
/* Batched form of the add loop of PartitionManager::add (partition_manager.cpp:236-258): vector i is appended to list
 * assign[i]; the append order inside a list is the input order.  ids/vecs/assign all live in `mem`. */
QK_API int qk_store_add_batch(qk_store *s, int64_t n, const int64_t *ids, const float *vecs, const int64_t *assign, int mem);
/* Bulk form of init_partitions (partition_manager.cpp:33-121): lists 0..nlist-1 created and filled from a CSR
 * arena (vecs [offsets[nlist]][d], ids, offsets [nlist+1] on the HOST always; vecs/ids in `mem`). */
QK_API int qk_store_build_csr(qk_store *s, int64_t nlist, const int64_t *offsets_host, const int64_t *ids,
                              const float *vecs, int mem);
/* remove_vectors :137-149: remove every id in `ids` from every list, swap-with-last per removal.
 * n_removed (may be NULL) receives the number of rows removed. */
QK_API int qk_store_remove_ids(qk_store *s, int64_t n, const int64_t *ids_host, int64_t *n_removed);
QK_API int qk_store_list_size(qk_store *s, int64_t list_no, int64_t *out);      /* list_size :68-74 */
/* PartitionManager::get_partition_sizes(Tensor) (partition_manager.cpp:296-306): the sizes of n lists in one call (host arrays);
 * an absent list is QK_ERR_NOT_FOUND like list_size.  The maintenance policy asks for every partition's size on every call. */
QK_API int qk_store_list_sizes(qk_store *s, const int64_t *list_nos, int64_t n, int64_t *out);
QK_API int64_t qk_store_ntotal(qk_store *s);                                    /* ntotal :60-66 */
QK_API int64_t qk_store_nlist(qk_store *s);
QK_API int qk_store_d(qk_store *s);
/* list numbers currently present, ascending; out may be NULL to query the count (return value via *n). */
QK_API int qk_store_list_ids(qk_store *s, int64_t *out_host, int64_t *n);
/* get_codes / get_ids :76-90 as a copy-out: rows in partition order, row-major [n][d]. */
QK_API int qk_store_get_list(qk_store *s, int64_t list_no, float *vecs_out, int64_t *ids_out, int mem);
/* The same for n lists at once, rows (and ids) laid one list after the other in list_nos order -- the caller sizes the buffers from
 * qk_store_list_sizes.  PartitionManager::select_partitions (partition_manager.cpp:344-390) / the rejection rule of the maintenance
 * policy (maintenance_policies.cpp:79-101), which read hundreds of lists per call. */
QK_API int qk_store_get_lists(qk_store *s, const int64_t *list_nos, int64_t n, float *vecs_out, int64_t *ids_out, int mem);
/* get_vector_for_id :280-293 (first match in ascending list order); *found = 0 if absent. */
QK_API int qk_store_get_vector(qk_store *s, int64_t id, float *vec_out_host, int *found);
/* PartitionManager::get(ids) (partition_manager.cpp:264-283): n vectors by id in ONE call -- found[i] = 0 and row i undefined for an
 * id the store does not hold.  (The maintenance policy reads hundreds of centroids of the parent per call.) */
QK_API int qk_store_get_vectors(qk_store *s, const int64_t *ids_host, int64_t n, float *vecs_out_host, int *found);
/* What the store's mutations have cost so far beyond the rows they were asked to write (no reference counterpart: IndexPartition
 * reallocs one partition at a time, index_partition.cpp:247-255): out[0] arena re-allocations (new arena, copy of everything, free),
 * [1] arena compactions, [2] list relocations (a list outgrew its extent), [3] rows copied by [0]-[2], [4] rebuilds of the row-major
 * copy of a parent's centroids, [5] uploads of the partition table, [6] rebuilds of the id -> list index (lazy: the first remove /
 * get after a bulk build walks every id), [7] re-allocations of the scratch buffers of the store's CONTEXT (hipFree + hipMalloc
 * behind a synchronisation).  A harness takes the difference around an operation
 * to attribute a slow add / remove / maintenance step. */
/* Make pending changes visible to searches NOW: a store that was modified uploads its list table (and, a one-list store, rebuilds
 * the row-major copy of its rows) at the next search -- two or three stream synchronisations and a copy, ~0.2 ms, inside a query.
 * A caller that has just finished a batch of modifications (add / remove / maintenance) calls this so that the queries after it do
 * not pay.  No reference counterpart (its lists are host vectors). */
QK_API int qk_store_publish(qk_store *s);
QK_API int qk_store_counters(qk_store *s, int64_t *out, int n);
/* bytes of HBM held by the arena (vectors+norms+ids), for capacity planning */
QK_API int64_t qk_store_device_bytes(qk_store *s);

/* ---- search ------------------------------------------------------------------------------------ */
/* Coarse step = parent_->search(x, {k = min(nprobe, nlist), batched_scan = true})
 * (query_coordinator.cpp:628-644 -> batched_scan_list over the centroid list, list_scanning.h:313-366).
 * `parent` is the store of the parent (flat) index: its lists hold the centroids, ids = partition ids.
 * out_pids [Q][kk], out_dist [Q][kk] (may be NULL), kk = min(nprobe, parent ntotal); rows padded with -1. */
QK_API int qk_coarse(qk_ctx *ctx, qk_store *parent, const float *x, int64_t Q, int nprobe, int metric, int64_t *out_pids,
                     float *out_dist, int mem);

/* QueryCoordinator::scan_partitions (query_coordinator.cpp:659-673; serial_scan :471-611 and
 * batched_serial_scan :675-799 give the same result here).  x [Q][d]; pids [Q][P] partition numbers to scan
 * per query, -1 = skip (:540); out_ids/out_dist [Q][k].  timing may be NULL. */
QK_API int qk_scan(qk_ctx *ctx, qk_store *s, const float *x, int64_t Q, const int64_t *pids, int P, int k, int metric,
                   int64_t *out_ids, float *out_dist, int mem, qk_timing *timing);

/* QueryCoordinator::search (query_coordinator.cpp:612-657) at fixed nprobe: coarse + scan in one enqueue,
 * no host round trip between the two.  parent == NULL: flat index, every list of `s` is scanned (:624-626). */
QK_API int qk_search(qk_ctx *ctx, qk_store *parent, qk_store *s, const float *x, int64_t Q, int nprobe, int k, int metric,
                     int64_t *out_ids, float *out_dist, int mem, qk_timing *timing);

/* qk_search that also hands out WHICH lists every query scanned -- out_probed [Q][min(nprobe, parent lists)] list numbers in rank
 * order (host or device like the other buffers) -- in the same enqueue: what QuakeIndex::search passes to
 * MaintenancePolicy::record_query_hits (maintenance_policies.cpp:179-182; the reference has the list on the host anyway).  The
 * nearest-centroid step writes the caller's buffer and the scan reads it: no second call, no copy. */
QK_API int qk_search_tracked(qk_ctx *ctx, qk_store *parent, qk_store *s, const float *x, int64_t Q, int nprobe, int k, int metric,
                             int64_t *out_ids, float *out_dist, int64_t *out_probed, int mem, qk_timing *timing);

/* QueryCoordinator::search with SearchParams::recall_target > 0 and batched_scan == false: adaptive partition
 * scanning (query_coordinator.cpp:612-657 picks M = max((int)(nlist * initial_search_fraction), 1) candidate partitions
 * from the parent; the use_aps branch of serial_scan, :471-611, scans them in rank order and stops a query once the
 * recall estimate of include/geometry.h:57-113,247-295,345-407 reaches recall_target).  The batch advances in rounds on
 * the device; per query the result and the count of partitions scanned are those of the sequential walk.
 * out_ids/out_dist [Q][k]; out_nscanned [Q] (may be NULL) = partitions the walk visited; timing (may be NULL):
 * total_ms and n_items = rounds.  Errors: parent == NULL or fewer than 2 candidates -> QK_ERR_INVALID
 * ("Boundary distances must have at least 2 partitions to create an estimate.", geometry.h:350). */
QK_API int qk_search_aps(qk_ctx *ctx, qk_store *parent, qk_store *s, const float *x, int64_t Q, int k, int metric,
                         float recall_target, float recompute_threshold, int use_precomputed, float initial_search_fraction,
                         int64_t *out_ids, float *out_dist, int32_t *out_nscanned, int mem, qk_timing *timing);

/* Multi-GPU merge step (the cross-worker batch_add of worker_scan, query_coordinator.cpp:167-173,231-235):
 * merge G per-rank results [G][Q][k] (already all-gathered by the caller, e.g. torch.distributed over RCCL) into
 * [Q][k] under the same (key,id) order.  in_key are SQUARED L2 distances / inner products, i.e. what qk_search
 * returns after qk_ctx_set_squared_l2(ctx, 1); the output distances are sqrt'd.  Device pointers only. */
QK_API int qk_merge_topk(qk_ctx *ctx, const int64_t *in_ids, const float *in_key, int G, int64_t Q, int k, int metric,
                         int64_t *out_ids, float *out_dist);
/* The same exchange as ONE collective (the ids + keys of an entry travel together, 12 bytes): qk_pack_topk turns a rank's local
 * result ids/key [G*per][k] into G blocks of qk_topk_block_bytes(per, k) bytes -- block j = queries [j*per, (j+1)*per): per*k
 * int64 ids then per*k float keys, padded to 16 bytes -- which is the send buffer of one all_to_all_single with equal splits;
 * qk_merge_topk_packed merges the receive buffer (block r = rank r's results for this rank's `per` queries) into [per][k]
 * under the same (key,id) order.  Device pointers only; 16-byte aligned buffers. */
QK_API size_t qk_topk_block_bytes(int64_t per, int k);
QK_API int qk_pack_topk(qk_ctx *ctx, const int64_t *ids, const float *key, int G, int64_t per, int k, void *packed);
QK_API int qk_merge_topk_packed(qk_ctx *ctx, const void *packed, int G, int64_t per, int k, int metric, int64_t *out_ids,
                                float *out_dist);
/* When enabled, qk_scan/qk_search return squared L2 distances (the merge key) instead of sqrt distances. */
QK_API int qk_ctx_set_squared_l2(qk_ctx *ctx, int enabled);

/* ---- device group: IndexBuildParams::num_workers as GPUs ---------------------------------------------------------------
 * The reference spreads a search over cores with num_workers (common.h:73,127): QueryCoordinator::initialize_workers
 * (query_coordinator.cpp:50-74) starts one thread per core, PartitionManager::distribute_partitions pins partition i to core
 * i % num_workers (partition_manager.cpp:557-603), worker_scan (query_coordinator.cpp:243-469) hands every core the jobs of its
 * partitions and batch_adds the per-core buffers into the global one (:167-173,231-235).  Here a worker is a device: a group is
 * ONE process driving G members -- one context + one shard store each; members may share a physical device (num_workers larger
 * than the node, and one-GPU test boxes) -- and list p lives in member p % G.  A search: the batch reaches the lead (member 0),
 * the others pull it over xGMI; the coarse step is split by queries and every member writes its slice of the [Q][nprobe] list
 * numbers straight into every other member's copy (peer stores); every member scans the whole batch over ITS lists and writes
 * its packed [Q][k] (ids, merge keys) block -- qk_pack_topk's layout -- into the lead's receive buffer; the lead merges the G
 * blocks under the (key, id) order.  Events order the devices; there is no host thread per device and no host synchronisation
 * inside a call on device buffers.  Results (ids and distance bits) equal the one-store search on the same lists.
 * Requires peer access between all distinct devices of the group (QK_ERR_UNSUPPORTED otherwise). */
typedef struct qk_group qk_group;
QK_API int qk_group_create(const int *devices, int G, int d, qk_group **out);   /* initialize_workers :50-74 */
QK_API int qk_group_destroy(qk_group *g);                                       /* shutdown_workers :77-95 */
QK_API int qk_group_size(qk_group *g);
/* member i's context and shard store (borrowed; the store holds the lists p with p % G == i under their global numbers) */
QK_API int qk_group_member(qk_group *g, int i, qk_ctx **ctx, qk_store **store);
QK_API int qk_group_owner(qk_group *g, int64_t list_no);                        /* get_partition_core_id: list_no % G */
/* The lead's stream: outputs in device memory are complete in its order (qk_ctx_set_stream / _set_null_stream / _get_stream of
 * the lead's context); inputs in device memory are read behind whatever that stream holds at the time of the call. */
QK_API int qk_group_set_stream(qk_group *g, void *hip_stream);
QK_API int qk_group_set_null_stream(qk_group *g);
QK_API int qk_group_get_stream(qk_group *g, void **hip_stream, int *kind);
QK_API int qk_group_synchronize(qk_group *g);                                   /* every member's stream */
/* Submit threads (default on, G >= 2): the per-member pieces of a search -- pull the batch, rank a slice of it, scan, pack -- are
 * enqueued by one persistent host thread per member instead of the caller's thread one member after the other (0.39 ms of host
 * time per call at 8 members -> one member's share plus two fork-joins).  Same streams, same events, same results; 0 = the
 * caller's thread does everything.  Reference: one scan thread per worker, query_coordinator.cpp:50-74,98-240. */
QK_API int qk_group_set_submit_threads(qk_group *g, int enabled);
QK_API int qk_group_set_form_feedback(qk_group *g, int enabled);                /* qk_ctx_set_form_feedback on every member */
/* The store surface over the members (same arguments, same errors as the qk_store_* call each one routes to). */
QK_API int qk_group_reset(qk_group *g);
QK_API int qk_group_add_list(qk_group *g, int64_t list_no);
QK_API int qk_group_remove_list(qk_group *g, int64_t list_no);
QK_API int qk_group_add_entries(qk_group *g, int64_t list_no, int64_t n, const int64_t *ids, const float *vecs, int mem);
QK_API int qk_group_add_batch(qk_group *g, int64_t n, const int64_t *ids, const float *vecs, const int64_t *assign, int mem);
QK_API int qk_group_build_csr(qk_group *g, int64_t nlist, const int64_t *offsets_host, const int64_t *ids, const float *vecs,
                              int mem);                                         /* init_partitions + distribute_partitions */
QK_API int qk_group_remove_ids(qk_group *g, int64_t n, const int64_t *ids_host, int64_t *n_removed);
QK_API int qk_group_list_size(qk_group *g, int64_t list_no, int64_t *out);
QK_API int qk_group_list_sizes(qk_group *g, const int64_t *list_nos, int64_t n, int64_t *out);
QK_API int64_t qk_group_ntotal(qk_group *g);
QK_API int64_t qk_group_nlist(qk_group *g);
QK_API int qk_group_d(qk_group *g);
QK_API int qk_group_list_ids(qk_group *g, int64_t *out_host, int64_t *n);
QK_API int qk_group_get_list(qk_group *g, int64_t list_no, float *vecs_out, int64_t *ids_out, int mem); /* complete on return */
QK_API int qk_group_get_lists(qk_group *g, const int64_t *list_nos, int64_t n, float *vecs_out, int64_t *ids_out, int mem);
QK_API int qk_group_get_vector(qk_group *g, int64_t id, float *vec_out_host, int *found);
QK_API int64_t qk_group_device_bytes(qk_group *g);
/* qk_store_refine_lists over lists of several members: they meet in a temporary store on the member holding the first one,
 * are refined there (same order, same arithmetic) and go back to their owners, device to device. */
QK_API int qk_group_refine_lists(qk_group *g, const int64_t *list_nos, int64_t m, float *centroids, int metric,
                                 int refinement_iterations, int mem);
/* worker_scan (query_coordinator.cpp:243-469) through scan_partitions (:659-673): qk_scan over the members. */
QK_API int qk_group_scan(qk_group *g, const float *x, int64_t Q, const int64_t *pids, int P, int k, int metric, int64_t *out_ids,
                         float *out_dist, int mem, qk_timing *timing);
/* QueryCoordinator::search (:612-657) with workers: qk_search over the members.  `parent` is the parent's ordinary store (any
 * device); the group keeps a replica of it per member and refreshes the replicas when the parent has changed.  timing:
 * coarse_ms / scan_ms / merge_ms / total_ms between events on the lead, the counters summed over the members. */
QK_API int qk_group_search(qk_group *g, qk_store *parent, const float *x, int64_t Q, int nprobe, int k, int metric,
                           int64_t *out_ids, float *out_dist, int mem, qk_timing *timing);
/* qk_search_aps over the members (the APS hook of worker_scan, query_coordinator.cpp:364-428 -- whose outcome depends on thread
 * timing upstream; here the deterministic walk): the rounds run on the lead, every member scans the pairs of a round whose lists it
 * holds, the lead takes each pair's top-k from its owner.  Answers and partitions visited equal qk_search_aps on one store. */
QK_API int qk_group_search_aps(qk_group *g, qk_store *parent, const float *x, int64_t Q, int k, int metric, float recall_target,
                               float recompute_threshold, int use_precomputed, float initial_search_fraction, int64_t *out_ids,
                               float *out_dist, int32_t *out_nscanned, int mem, qk_timing *timing);

/* ---- k-means ----------------------------------------------------------------------------------- */
/* Nearest-centroid assignment: IndexFlat::search(n, x, 1) (clustering.cpp:63-66) and the
 * batched_scan_list(k=1) of kmeans_refine_partitions (clustering.cpp:149-159).
 * x [n][d], c [m][d]; assign [n] (row index into c, ties -> lower index); val [n] squared L2 / dot (may be NULL). */
QK_API int qk_kmeans_assign(qk_ctx *ctx, const float *x, int64_t n, const float *c, int64_t m, int d, int metric,
                            int64_t *assign, float *val, int mem);
/* Update: per-centroid fp32 sums and counts; sums [m][d], counts [m]; rows with an assignment outside [0, m) are ignored.
 *   qk_kmeans_accumulate          rows added one after the other in ascending row order -- the order of the reference's own
 *                                 accumulate loop in kmeans_refine_partitions (clustering.cpp:162-176: centroid_sums[c][j] += vec[j]),
 *                                 what qk_store_refine_lists uses
 *   qk_kmeans_accumulate_blocked  the mean update of kmeans() (clustering.cpp:51-55 hands it to faiss::Clustering, whose summation
 *                                 order is its back end's): this library's canonical BLOCKED order -- a centroid's rows in ascending
 *                                 row order, sequential fp32 sums over consecutive blocks of 32 rows, the block partials summed
 *                                 sequentially per group of 32 blocks, the group partials summed sequentially (all from +0) -- so
 *                                 that a large cluster is many independent chains; what qk_kmeans uses */
QK_API int qk_kmeans_accumulate(qk_ctx *ctx, const float *x, int64_t n, int d, const int64_t *assign, int64_t m,
                                float *sums, int64_t *counts, int mem);
QK_API int qk_kmeans_accumulate_blocked(qk_ctx *ctx, const float *x, int64_t n, int d, const int64_t *assign, int64_t m,
                                        float *sums, int64_t *counts, int mem);
/* kmeans_refine_partitions() (clustering.cpp:99-182) together with the partition replacement of
 * PartitionManager::refine_partitions (partition_manager.cpp:446-487), applied to the device store: the vectors of the m
 * lists `list_nos` (host array) are re-assigned to the nearest of the m centroids [m][d] (in `mem`; row c = centroid of
 * list_nos[c]) for max(refinement_iterations, 1) passes, centroids recomputed between passes; on return list_nos[c] holds
 * the vectors assigned to centroid c (append order of the reference) and `centroids` the ones used for the last pass. */
QK_API int qk_store_refine_lists(qk_store *s, const int64_t *list_nos, int64_t m, float *centroids, int metric,
                                 int refinement_iterations, int mem);
/* kmeans() (clustering.cpp:13-97): Lloyd iterations on the GPU.  x [n][d] (IP: normalised IN PLACE, as the
 * reference stores the normalised copy, clustering.cpp:25-26,71); centroids [m][d] out; assign [n] out =
 * final full assignment.  seed drives the documented splitmix64 initialisation. */
QK_API int qk_kmeans(qk_ctx *ctx, float *x, int64_t n, int d, int64_t m, int metric, int niter, uint64_t seed,
                     float *centroids, int64_t *assign, int mem);

/* The pieces of kmeans() a multi-GPU build needs between its collectives (local assign + local partial sums,
 * all-reduce of [m][d] sums + [m] counts per iteration; quake_amd/sharded.py):
 *   qk_normalize_rows   x /= ||x|| row by row, canonical norm (clustering.cpp:25-26,59-60), in place
 *   qk_kmeans_update    centroids = sums / counts for non-empty clusters, previous centroid kept for empty ones, then the
 *                       empty-cluster split of faiss::Clustering restated deterministically (largest cluster first, pair
 *                       perturbed by 1 +/- 1/1024); counts are updated by the split
 *   qk_rand_perm        first m entries of the splitmix64 Fisher-Yates permutation of [0, n) (host array out): the
 *                       subsample / initial centroids of qk_kmeans */
QK_API int qk_normalize_rows(qk_ctx *ctx, float *x, int64_t n, int d, int mem);
QK_API int qk_kmeans_update(qk_ctx *ctx, const float *sums, int64_t *counts, int64_t m, int d, float *centroids, int mem);
QK_API int qk_rand_perm(int64_t n, int64_t m, uint64_t seed, int64_t *perm_out_host);
/* Kernel-side durations (HIP events on the context's stream) of the LAST Lloyd iteration of the last qk_kmeans on this context: the
 * assign step (k_assign + its centroid re-tiling) and the update step (bucketing + k_accumulate) over `rows` training rows and `m`
 * centroids -- what a harness prices against the MFMA / HBM roofs (no reference counterpart: faiss::Clustering prints its own
 * per-iteration times under `verbose`, clustering.cpp:41).  Zeros before the first qk_kmeans. */
QK_API int qk_kmeans_last_timing(qk_ctx *ctx, float *assign_ms, float *update_ms, int64_t *rows, int64_t *m);

#ifdef __cplusplus
}
#endif
#endif /* QUAKE_HIP_H */
