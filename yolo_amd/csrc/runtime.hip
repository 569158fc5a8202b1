// Error reporting and the launch-list executor of libydet.so.
#include <pthread.h>
#include <string.h>

#include <new>

#include "ydet_internal.h"

int ydet_launch_slice(int op_kind, const ydet_slice_desc* d, int mode, hipStream_t stream);
int ydet_launch_nchw_to_nhwc(const void* x, void* y, int N, int C, int H, int W, int y_ld, int y_coff, int c_pad,
                             int x_dtype, int y_dtype, int focus, hipStream_t stream);
int ydet_launch_nhwc_to_nchw(const void* x, void* y, int N, int C, int H, int W, int x_ld, int x_coff, int x_dtype,
                             int y_dtype, hipStream_t stream);

namespace ydet {

static thread_local char g_err[512] = {0};

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
void clear_error() { g_err[0] = 0; }

DeviceInfo current_device_info() {
    static DeviceInfo info[kMaxDevices];       // cus == 0: not asked yet (racing first calls write the same values)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return DeviceInfo{256, 160 * 1024};
    if (info[dev].cus == 0) {
        DeviceInfo d{256, 160 * 1024};
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) d.cus = v;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) == hipSuccess && v > 0) d.lds_per_cu = v;
        info[dev] = d;
    }
    return info[dev];
}

int raise_dynamic_lds_limit(const void* kernel, int bytes, unsigned char (&done)[kMaxDevices], const char* what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return fail(YDET_ERR_LAUNCH, "%s: no current device", what);
    if (done[dev]) return YDET_OK;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return fail(YDET_ERR_LAUNCH, "%s: cannot raise the dynamic LDS limit to %d bytes: %s", what, bytes, hipGetErrorString(e));
    done[dev] = 1;
    return YDET_OK;
}

}  // namespace ydet

extern "C" const char* ydet_last_error(void) { return ydet::g_err; }
extern "C" int ydet_abi_version(void) { return YDET_ABI_VERSION; }

struct ydet_timer {
    int n_marks;
    hipEvent_t* ev;
};

extern "C" ydet_timer* ydet_timer_create(int n_marks) {
    ydet::clear_error();
    if (n_marks <= 0) { ydet::fail(YDET_ERR_ARG, "timer_create: n_marks=%d", n_marks); return nullptr; }
    ydet_timer* t = new (std::nothrow) ydet_timer;
    if (!t) { ydet::fail(YDET_ERR_LAUNCH, "timer_create: out of host memory"); return nullptr; }
    t->n_marks = n_marks;
    t->ev = new (std::nothrow) hipEvent_t[n_marks];
    if (!t->ev) { delete t; ydet::fail(YDET_ERR_LAUNCH, "timer_create: out of host memory"); return nullptr; }
    for (int i = 0; i < n_marks; ++i) {
        if (hipEventCreate(&t->ev[i]) != hipSuccess) {
            for (int j = 0; j < i; ++j) (void)hipEventDestroy(t->ev[j]);
            delete[] t->ev;
            delete t;
            ydet::fail(YDET_ERR_LAUNCH, "timer_create: hipEventCreate failed");
            return nullptr;
        }
    }
    return t;
}

extern "C" void ydet_timer_destroy(ydet_timer* t) {
    if (!t) return;
    for (int i = 0; i < t->n_marks; ++i) (void)hipEventDestroy(t->ev[i]);
    delete[] t->ev;
    delete t;
}

extern "C" int ydet_timer_elapsed(ydet_timer* t, int first_mark, int n_intervals, float* ms) {
    ydet::clear_error();
    if (!t || !ms || first_mark < 0 || n_intervals < 0 || first_mark + n_intervals > t->n_marks - 1)
        return ydet::fail(YDET_ERR_ARG, "timer_elapsed: bad range");
    for (int i = 0; i < n_intervals; ++i) {
        hipError_t e = hipEventElapsedTime(&ms[i], t->ev[first_mark + i], t->ev[first_mark + i + 1]);
        if (e != hipSuccess) return ydet::fail(YDET_ERR_LAUNCH, "timer_elapsed: %s", hipGetErrorString(e));
    }
    return YDET_OK;
}

static int run_ops_impl(const ydet_op* ops, int n_ops, hipStream_t s, ydet_timer* timer, int first_mark);

// Runs a pre-planned forward pass: every op is validated and enqueued on `stream` in order; the
// first failure stops the walk and is reported with the op index.
extern "C" int ydet_run_ops(const ydet_op* ops, int n_ops, void* stream) {
    ydet::clear_error();
    return run_ops_impl(ops, n_ops, ydet::as_stream(stream), nullptr, 0);
}

extern "C" int ydet_run_ops_timed(const ydet_op* ops, int n_ops, void* stream, ydet_timer* timer, int first_mark) {
    ydet::clear_error();
    if (!timer || first_mark < 0 || first_mark + n_ops + 1 > timer->n_marks)
        return ydet::fail(YDET_ERR_ARG, "run_ops_timed: timer has %d marks, need %d..%d", timer ? timer->n_marks : 0,
                          first_mark, first_mark + n_ops);
    return run_ops_impl(ops, n_ops, ydet::as_stream(stream), timer, first_mark);
}

static int launch_op(const ydet_op& op, hipStream_t s) {
    switch (op.kind) {
        case YDET_OP_CONV: return ydet_launch_conv(&op.conv, s);
        case YDET_OP_FOCUS_CONV: return ydet_launch_focus_conv(&op.conv, op.iarg3, s);
        case YDET_OP_FOCUS_PAIR: return ydet_launch_focus_pair(&op.conv, &op.conv2, op.iarg3, s, false);
        case YDET_OP_RGB_PAIR: return ydet_launch_focus_pair(&op.conv, &op.conv2, op.iarg3, s, true);
        case YDET_OP_RGB_CONV: return ydet_launch_rgb_conv(&op.conv, op.iarg3, s);
        case YDET_OP_HEAD_DECODE: return ydet_launch_head_decode(&op.conv, &op.head, s);
        case YDET_OP_BOTTLENECK: return ydet_launch_bottleneck(&op.conv, &op.conv2, nullptr, s);
        case YDET_OP_C3_TAIL: return ydet_launch_bottleneck(&op.conv, &op.conv2, &op.conv3, s);
        case YDET_OP_MAXPOOL2:
        case YDET_OP_MAXPOOL2S1:
        case YDET_OP_SPP:
        case YDET_OP_UPSAMPLE:
        case YDET_OP_SPACE2DEPTH:
        case YDET_OP_COPY: return ydet_launch_slice(op.kind, &op.slice, op.iarg, s);
        case YDET_OP_NCHW2NHWC:
            return ydet_launch_nchw_to_nhwc(op.slice.x, op.slice.y, op.slice.N, op.slice.C, op.slice.H, op.slice.W,
                                            op.slice.y_ld, op.slice.y_coff, op.iarg2, op.iarg3, op.slice.dtype, op.iarg, s);
        case YDET_OP_NHWC2NCHW:
            return ydet_launch_nhwc_to_nchw(op.slice.x, op.slice.y, op.slice.N, op.slice.C, op.slice.H, op.slice.W,
                                            op.slice.x_ld, op.slice.x_coff, op.slice.dtype, op.iarg3, s);
        default: return ydet::fail(YDET_ERR_ARG, "unknown op kind %d", op.kind);
    }
}

static int op_failed(int rc, int i, int n_ops) {
    char msg[400];
    strncpy(msg, ydet::g_err, sizeof(msg) - 1);
    msg[sizeof(msg) - 1] = 0;
    return ydet::fail(rc, "run_ops: op %d of %d: %s", i, n_ops, msg);
}

static int run_ops_impl(const ydet_op* ops, int n_ops, hipStream_t s, ydet_timer* timer, int first_mark) {
    if (n_ops < 0 || (n_ops > 0 && !ops)) return ydet::fail(YDET_ERR_ARG, "run_ops: bad op list");
    if (timer && hipEventRecord(timer->ev[first_mark], s) != hipSuccess)
        return ydet::fail(YDET_ERR_LAUNCH, "run_ops: hipEventRecord failed");
    for (int i = 0; i < n_ops; ++i) {
        const int rc = launch_op(ops[i], s);
        if (rc != YDET_OK) return op_failed(rc, i, n_ops);
        if (timer && hipEventRecord(timer->ev[first_mark + i + 1], s) != hipSuccess)
            return ydet::fail(YDET_ERR_LAUNCH, "run_ops: hipEventRecord failed");
    }
    return YDET_OK;
}

// ---- independent launch lists side by side -------------------------------------------------------------------------
// Per device: a few non-blocking side streams and the events that fork them from / join them to the caller's
// stream.  Created on first use, kept for the life of the process (streams and events own no device memory).  One
// pool per device: a call holds the pool's lock from its fork to its join (host-side enqueue only, nothing waits for
// the GPU), so that two host threads on different caller streams cannot interleave their fork / wait / join on the
// shared events; their lists still share the side streams, i.e. run one call after the other there.
namespace {
constexpr int kMaxLists = 4;
struct SidePool {
    bool ready = false;
    hipStream_t side[kMaxLists - 1];
    hipEvent_t fork, join[kMaxLists - 1];
    pthread_mutex_t busy = PTHREAD_MUTEX_INITIALIZER;
};
struct PoolGuard {
    pthread_mutex_t* m;
    explicit PoolGuard(pthread_mutex_t* mm) : m(mm) { pthread_mutex_lock(m); }
    ~PoolGuard() { pthread_mutex_unlock(m); }
};
SidePool g_pool[ydet::kMaxDevices];
pthread_mutex_t g_pool_lock = PTHREAD_MUTEX_INITIALIZER;

SidePool* side_pool() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= ydet::kMaxDevices) return nullptr;
    pthread_mutex_lock(&g_pool_lock);
    SidePool* p = &g_pool[dev];
    if (!p->ready) {
        bool ok = hipEventCreateWithFlags(&p->fork, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; ok && i < kMaxLists - 1; ++i)
            ok = hipStreamCreateWithFlags(&p->side[i], hipStreamNonBlocking) == hipSuccess &&
                 hipEventCreateWithFlags(&p->join[i], hipEventDisableTiming) == hipSuccess;
        p->ready = ok;
    }
    pthread_mutex_unlock(&g_pool_lock);
    return p->ready ? p : nullptr;
}
}  // namespace

extern "C" int ydet_run_op_lists(const ydet_op* const* lists, const int* counts, int n_lists, void* stream) {
    ydet::clear_error();
    if (n_lists < 1 || n_lists > kMaxLists || !lists || !counts)
        return ydet::fail(YDET_ERR_ARG, "run_op_lists: n_lists=%d (1..%d)", n_lists, kMaxLists);
    hipStream_t main_s = ydet::as_stream(stream);
    if (n_lists == 1) return run_ops_impl(lists[0], counts[0], main_s, nullptr, 0);
    int longest = 0;
    for (int r = 0; r < n_lists; ++r) {
        if (counts[r] < 0 || (counts[r] > 0 && !lists[r])) return ydet::fail(YDET_ERR_ARG, "run_op_lists: bad list %d", r);
        if (counts[r] > longest) longest = counts[r];
    }
    SidePool* pool = side_pool();
    if (!pool) return ydet::fail(YDET_ERR_LAUNCH, "run_op_lists: could not create the side streams");
    PoolGuard guard(&pool->busy);
    // fork: nothing of the side lists may start before the work already queued on the caller's stream
    if (hipEventRecord(pool->fork, main_s) != hipSuccess) return ydet::fail(YDET_ERR_LAUNCH, "run_op_lists: fork failed");
    for (int r = 1; r < n_lists; ++r)
        if (hipStreamWaitEvent(pool->side[r - 1], pool->fork, 0) != hipSuccess)
            return ydet::fail(YDET_ERR_LAUNCH, "run_op_lists: fork failed");
    // round-robin submission keeps every queue fed
    int rc = YDET_OK;
    for (int i = 0; i < longest && rc == YDET_OK; ++i)
        for (int r = 0; r < n_lists && rc == YDET_OK; ++r)
            if (i < counts[r]) {
                rc = launch_op(lists[r][i], r == 0 ? main_s : pool->side[r - 1]);
                if (rc != YDET_OK) rc = op_failed(rc, i, counts[r]);
            }
    // join (also after a failure: the caller's stream must not run ahead of what was queued)
    for (int r = 1; r < n_lists; ++r)
        if (hipEventRecord(pool->join[r - 1], pool->side[r - 1]) != hipSuccess ||
            hipStreamWaitEvent(main_s, pool->join[r - 1], 0) != hipSuccess)
            return rc != YDET_OK ? rc : ydet::fail(YDET_ERR_LAUNCH, "run_op_lists: join failed");
    return rc;
}
