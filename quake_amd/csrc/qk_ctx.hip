// qk_ctx.hip -- error state, context, scratch memory.
#include "qk_internal.h"

#include <cstdlib>

#include <cstdarg>
#include <cstdio>
#include <cstring>

static thread_local char g_err[1024] = "";

void qk_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int qk_check_overflow(qk_ctx *ctx) {
    if (ctx && ctx->overflow_host && *(volatile int *)ctx->overflow_host) {
        *(volatile int *)ctx->overflow_host = 0;
        QK_FAIL(QK_ERR_HIP, "a scan launched on this context overflowed its record buffer: the results of that call are incomplete");
    }
    return QK_OK;
}

extern "C" {

const char *qk_last_error(void) { return g_err; }
const char *qk_version(void) { return "quake_hip 0.1 (gfx950)"; }

int qk_ctx_create(int device, qk_ctx **out) {
    if (!out) QK_FAIL(QK_ERR_INVALID, "qk_ctx_create: out is null");
    int ndev = 0;
    QK_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) QK_FAIL(QK_ERR_INVALID, "qk_ctx_create: device %d out of range (%d devices)", device, ndev);
    QK_HIP(hipSetDevice(device));
    qk_ctx *c = new qk_ctx();
    c->device = device;
    // (the one environment switch of the product library, read at context creation: a whole test suite or a deployment that wants
    //  the static rule alone for every context, including the ones the mirrors create; include/quake_hip.h)
    if (const char *e = getenv("QK_FORM_FEEDBACK")) c->form_feedback = atoi(e) != 0;
    QK_HIP(hipGetDeviceProperties(&c->prop, device));
    QK_HIP(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    for (auto &e : c->ev) QK_HIP(hipEventCreate(&e));
    QK_HIP(hipHostMalloc((void **)&c->overflow_host, 64, hipHostMallocMapped));
    *c->overflow_host = 0;
    QK_HIP(hipHostGetDevicePointer((void **)&c->overflow_dev, c->overflow_host, 0));
    *out = c;
    return QK_OK;
}

int qk_ctx_destroy(qk_ctx *c) {
    if (!c) return QK_OK;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    if (c->ws) hipFree(c->ws);
    if (c->stage) hipFree(c->stage);
    if (c->qprep) hipFree(c->qprep);
    if (c->aps) hipFree(c->aps);
    if (c->small_ws) hipFree(c->small_ws);
    if (c->km_pool) hipFree(c->km_pool);
    if (c->pinned) hipHostFree(c->pinned);
    if (c->pin_io) hipHostFree(c->pin_io);
    if (c->overflow_host) hipHostFree(c->overflow_host);
    if (c->aps_flags) hipHostFree(c->aps_flags);
    if (c->aps_table) hipFree(c->aps_table);
    if (c->aps_rowof) hipFree(c->aps_rowof);
    if (c->xcd_host) hipHostFree(c->xcd_host);
    if (c->xcd_ev) hipEventDestroy(c->xcd_ev);
    for (auto &f : c->form_stats) {
        if (f.e0) hipEventDestroy(f.e0);
        if (f.e1) hipEventDestroy(f.e1);
    }
    if (c->stream_ev) hipEventDestroy(c->stream_ev);
    for (auto &e : c->ev)
        if (e) hipEventDestroy(e);
    for (auto e : c->ev_free) hipEventDestroy(e);
    for (auto e : c->ev_pending) hipEventDestroy(e);
    for (auto e : c->ev_pending_coarse) hipEventDestroy(e);
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    delete c;
    return QK_OK;
}

// The context's scratch (workspace, staging, query-prep and pinned buffers) is reused by every call.  When the stream
// changes, work enqueued on the old stream may still be using it: the new stream waits for the old one (an event, no host
// synchronisation), so two searches issued back to back on different streams cannot overwrite each other's scratch.
static int switch_stream(qk_ctx *c, hipStream_t next) {
    if (next == c->stream) return QK_OK;
    QK_HIP(hipSetDevice(c->device));
    if (!c->stream_ev) QK_HIP(hipEventCreateWithFlags(&c->stream_ev, hipEventDisableTiming));
    QK_HIP(hipEventRecord(c->stream_ev, c->stream));
    QK_HIP(hipStreamWaitEvent(next, c->stream_ev, 0));
    c->stream = next;
    return QK_OK;
}

int qk_ctx_set_stream(qk_ctx *c, void *hip_stream) {
    if (!c) QK_FAIL(QK_ERR_INVALID, "qk_ctx_set_stream: ctx is null");
    return switch_stream(c, hip_stream ? (hipStream_t)hip_stream : c->own_stream);
}

int qk_ctx_set_null_stream(qk_ctx *c) {
    if (!c) QK_FAIL(QK_ERR_INVALID, "qk_ctx_set_null_stream: ctx is null");
    return switch_stream(c, nullptr);  // hipStream_t 0: ordered with every blocking stream of the device
}

int qk_ctx_get_stream(qk_ctx *c, void **hip_stream, int *kind) {
    if (!c || !hip_stream || !kind) QK_FAIL(QK_ERR_INVALID, "qk_ctx_get_stream: null argument");
    *hip_stream = (void *)c->stream;
    *kind = c->stream == c->own_stream ? 0 : (c->stream == nullptr ? 1 : 2);
    return QK_OK;
}

int qk_ctx_set_form_feedback(qk_ctx *c, int enabled) {
    if (!c) QK_FAIL(QK_ERR_INVALID, "qk_ctx_set_form_feedback: ctx is null");
    c->form_feedback = enabled != 0;
    return QK_OK;
}

int qk_ctx_set_form_times(qk_ctx *c, const float *ms3) {
    if (!c) QK_FAIL(QK_ERR_INVALID, "qk_ctx_set_form_times: ctx is null");
    c->form_times_set = ms3 != nullptr;
    for (int f = 0; f < 3; f++) {
        if (ms3 && !(ms3[f] > 0.f)) QK_FAIL(QK_ERR_INVALID, "qk_ctx_set_form_times: ms3[%d] must be > 0", f);
        c->form_times[f] = ms3 ? ms3[f] : 0.f;
    }
    return QK_OK;
}

int qk_ctx_synchronize(qk_ctx *c) {
    if (!c) QK_FAIL(QK_ERR_INVALID, "qk_ctx_synchronize: ctx is null");
    QK_HIP(hipStreamSynchronize(c->stream));
    return qk_check_overflow(c);
}

int qk_ctx_set_timing(qk_ctx *c, int enabled) {
    if (!c) QK_FAIL(QK_ERR_INVALID, "qk_ctx_set_timing: ctx is null");
    c->timing = enabled == 1;
    c->timing_mode = enabled;
    return QK_OK;
}

int qk_ctx_get_timing(qk_ctx *c, int *mode) {
    if (!c || !mode) QK_FAIL(QK_ERR_INVALID, "qk_ctx_get_timing: null argument");
    *mode = c->timing_mode;
    return QK_OK;
}

static float elapsed_or_zero(hipEvent_t a, hipEvent_t b) {
    float ms = 0.f;
    if (!a || !b) return 0.f;
    if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return 0.f;
    return ms;
}

int qk_ctx_read_timing(qk_ctx *c, qk_timing *sum, int64_t *calls) {
    if (!c || !sum || !calls) QK_FAIL(QK_ERR_INVALID, "qk_ctx_read_timing: null argument");
    QK_HIP(hipStreamSynchronize(c->stream));
    memset(sum, 0, sizeof(*sum));
    *calls = (int64_t)(c->ev_pending.size() / 4);
    for (size_t i = 0; i + 3 < c->ev_pending.size(); i += 4) {
        hipEvent_t *e = &c->ev_pending[i];
        sum->group_ms += elapsed_or_zero(e[0], e[1]);
        sum->scan_ms += elapsed_or_zero(e[1], e[2]);
        sum->merge_ms += elapsed_or_zero(e[2], e[3]);
        sum->total_ms += elapsed_or_zero(e[0], e[3]);
    }
    for (size_t i = 0; i + 1 < c->ev_pending_coarse.size(); i += 2)
        sum->coarse_ms += elapsed_or_zero(c->ev_pending_coarse[i], c->ev_pending_coarse[i + 1]);
    for (auto e : c->ev_pending)
        if (e) c->ev_free.push_back(e);
    for (auto e : c->ev_pending_coarse) c->ev_free.push_back(e);
    c->ev_pending.clear();
    c->ev_pending_coarse.clear();
    return QK_OK;
}

int qk_ctx_last_scan_kernel(qk_ctx *c, char *name, int name_len) {
    if (!c || !name || name_len <= 0) QK_FAIL(QK_ERR_INVALID, "qk_ctx_last_scan_kernel: bad arguments");
    snprintf(name, (size_t)name_len, "%s", c->last_scan_kernel);
    return QK_OK;
}

int qk_ctx_device_info(qk_ctx *c, int *num_cus, int *clock_khz, int64_t *hbm_bytes, char *arch, int arch_len) {
    if (!c) QK_FAIL(QK_ERR_INVALID, "qk_ctx_device_info: ctx is null");
    if (num_cus) *num_cus = c->prop.multiProcessorCount;
    if (clock_khz) *clock_khz = c->prop.clockRate;
    if (hbm_bytes) *hbm_bytes = (int64_t)c->prop.totalGlobalMem;
    if (arch && arch_len > 0) {
        strncpy(arch, c->prop.gcnArchName, (size_t)arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return QK_OK;
}

}  // extern "C"

int qk_ws_reserve(qk_ctx *c, size_t bytes) {
    c->ws_off = 0;
    if (bytes <= c->ws_cap) return QK_OK;
    QK_HIP(hipStreamSynchronize(c->stream));
    if (c->ws) QK_HIP(hipFree(c->ws));
    c->ws = nullptr;
    c->ws_cap = 0;
    c->scratch_reallocs++;
    // powers of two from 64 MB: a few regrowths in the life of a context instead of one per slightly larger request
    size_t want = (size_t)64 << 20;
    while (want < bytes + (1u << 20)) want <<= 1;
    hipError_t e = hipMalloc((void **)&c->ws, want);
    if (e != hipSuccess) {
        qk_set_error("workspace allocation of %zu bytes failed: %s", want, hipGetErrorString(e));
        return QK_ERR_OOM;
    }
    c->ws_cap = want;
    return QK_OK;
}

void *qk_ws_alloc(qk_ctx *c, size_t bytes) {
    size_t off = (c->ws_off + 255) & ~(size_t)255;
    if (off + bytes > c->ws_cap) return nullptr;
    c->ws_off = off + bytes;
    return c->ws + off;
}

int qk_aps_reserve(qk_ctx *c, size_t bytes) {
    if (bytes <= c->aps_cap) return QK_OK;
    QK_HIP(hipStreamSynchronize(c->stream));
    if (c->aps) QK_HIP(hipFree(c->aps));
    c->aps = nullptr;
    c->aps_cap = 0;
    c->scratch_reallocs++;
    size_t want = bytes + bytes / 4 + (1u << 16);
    hipError_t e = hipMalloc((void **)&c->aps, want);
    if (e != hipSuccess) {
        qk_set_error("adaptive-search state allocation of %zu bytes failed: %s", want, hipGetErrorString(e));
        return QK_ERR_OOM;
    }
    c->aps_cap = want;
    return QK_OK;
}

int qk_pinned_reserve(qk_ctx *c, size_t bytes) {
    if (bytes <= c->pinned_cap) return QK_OK;
    QK_HIP(hipStreamSynchronize(c->stream));
    if (c->pinned) QK_HIP(hipHostFree(c->pinned));
    c->pinned = nullptr;
    c->pinned_cap = 0;
    c->scratch_reallocs++;
    size_t want = bytes + bytes / 4 + 4096;
    QK_HIP(hipHostMalloc((void **)&c->pinned, want, hipHostMallocDefault));
    c->pinned_cap = want;
    return QK_OK;
}

int qk_stage_reserve(qk_ctx *c, size_t bytes) {
    if (bytes <= c->stage_cap) return QK_OK;
    QK_HIP(hipStreamSynchronize(c->stream));
    if (c->stage) QK_HIP(hipFree(c->stage));
    c->stage = nullptr;
    c->stage_cap = 0;
    c->scratch_reallocs++;
    hipError_t e = hipMalloc((void **)&c->stage, bytes);
    if (e != hipSuccess) {
        qk_set_error("staging allocation of %zu bytes failed: %s", bytes, hipGetErrorString(e));
        return QK_ERR_OOM;
    }
    c->stage_cap = bytes;
    return QK_OK;
}
