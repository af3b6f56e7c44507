// Launch plans: the static launch sequence of one backbone call (forward, or the reverse tape) recorded ONCE by the host executor
// (adamml_amd/plan.py) as an array of fixed-size records and replayed by ONE call into this file -- no Python, no ctypes marshalling,
// no tensor bookkeeping per launch.  The shapes of a backbone call are static per (batch, segments, mode), so the sequence of C-ABI
// entry points, their descriptors and their pointers is the same every step; what varies (the call's input tensor, the incoming output
// gradient) is patched through POINTER SLOTS.  This is not graph capture: the records call the very extern "C" entry points of
// include/adamml_hip.h (each still chooses its kernel, grid and workspace itself), the streams are the caller's, and the events that
// order the weight-gradient stream against the main one are plain hipEventRecord / hipStreamWaitEvent calls.
//
// Why: at the per-GPU share of the reference's own recipe (global batch 72 over 8 GPUs = 9 videos per GPU, train_adamml.py:122) one
// step issues ~1300 launches whose device time is ~20 ms; issued from Python (~13 us per launch) the step is host-bound.
#include "common.h"
#include "../../include/adamml_hip.h"
#include <string.h>

static inline double plan_f64(uint64_t bits) {
    double d;
    memcpy(&d, &bits, sizeof(d));
    return d;
}

static int plan_call(int fn, const uint64_t* a, hipStream_t s) {
    switch (fn) {
#include "plan_thunks.inc"
        default: return adamml_set_error(ADAMML_EINVAL, "plan: unknown entry point id %d", fn);
    }
}

extern "C" int adamml_plan_num_entry_points(void) { return ADAMML_PLAN_NFN; }

// ops[i]: kind CALL  -> entry point `fn` with a[0..nargs) on streams[stream];
//         kind WAIT  -> streams[stream] waits for everything enqueued so far on streams[a[0]] (event events[a[1]]);
//         kind ZERO  -> hipMemsetAsync(a[0], 0, a[1] bytes) on streams[stream].
// Arguments flagged in `slot_mask` (bit j: argument j) are indices into `slots` (the caller's per-replay pointers) instead of values.
extern "C" int adamml_plan_run(const adamml_plan_op_t* ops, int n_ops, const hipStream_t* streams, int n_streams, const hipEvent_t* events,
                               int n_events, const uint64_t* slots, int n_slots) {
    if (!ops || n_ops < 0 || !streams) return adamml_set_error(ADAMML_EINVAL, "plan_run: null argument");
    uint64_t a[ADAMML_PLAN_MAX_ARGS];
    for (int i = 0; i < n_ops; ++i) {
        const adamml_plan_op_t& op = ops[i];
        if (op.stream < 0 || op.stream >= n_streams) return adamml_set_error(ADAMML_EINVAL, "plan_run: op %d: stream slot %d", i, op.stream);
        const hipStream_t s = streams[op.stream];
        if (op.kind == ADAMML_PLAN_CALL) {
            if (op.nargs < 0 || op.nargs > ADAMML_PLAN_MAX_ARGS) return adamml_set_error(ADAMML_EINVAL, "plan_run: op %d: %d arguments", i, op.nargs);
            for (int j = 0; j < op.nargs; ++j) {
                if ((op.slot_mask >> j) & 1u) {
                    if (op.a[j] >= (uint64_t)n_slots) return adamml_set_error(ADAMML_EINVAL, "plan_run: op %d: pointer slot %llu", i, (unsigned long long)op.a[j]);
                    a[j] = slots[op.a[j]];
                } else a[j] = op.a[j];
            }
            const int rc = plan_call(op.fn, a, s);
            if (rc) return rc;
        } else if (op.kind == ADAMML_PLAN_WAIT) {
            if (op.a[0] >= (uint64_t)n_streams || op.a[1] >= (uint64_t)n_events) return adamml_set_error(ADAMML_EINVAL, "plan_run: op %d: bad wait", i);
            hipError_t e = hipEventRecord(events[op.a[1]], streams[op.a[0]]);
            if (e == hipSuccess) e = hipStreamWaitEvent(s, events[op.a[1]], 0);
            if (e != hipSuccess) return adamml_set_error(ADAMML_ELAUNCH, "plan_run: op %d: stream wait failed: %s", i, hipGetErrorString(e));
        } else if (op.kind == ADAMML_PLAN_ZERO) {
            const hipError_t e = hipMemsetAsync((void*)(uintptr_t)op.a[0], 0, (size_t)op.a[1], s);
            if (e != hipSuccess) return adamml_set_error(ADAMML_ELAUNCH, "plan_run: op %d: memset failed: %s", i, hipGetErrorString(e));
        } else return adamml_set_error(ADAMML_EINVAL, "plan_run: op %d: unknown kind %d", i, op.kind);
    }
    return ADAMML_OK;
}

extern "C" int adamml_plan_events_create(hipEvent_t* events, int n) {
    for (int i = 0; i < n; ++i) {
        const hipError_t e = hipEventCreateWithFlags(&events[i], hipEventDisableTiming);
        if (e != hipSuccess) return adamml_set_error(ADAMML_ELAUNCH, "plan_events_create: %s", hipGetErrorString(e));
    }
    return ADAMML_OK;
}

extern "C" int adamml_plan_events_destroy(hipEvent_t* events, int n) {
    for (int i = 0; i < n; ++i)
        if (events[i]) hipEventDestroy(events[i]);
    return ADAMML_OK;
}

// dst[r][0..width) = src[r][0..width) for r < rows (byte pitches): the one strided device copy of the reverse tape (sum(g') columns of
// a BatchNorm-backward accumulator shared by conv3 and the downsample branch) as a stream-ordered C-ABI call, so that a launch plan
// records it like any other launch.
extern "C" int adamml_copy2d(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width_bytes, size_t rows, hipStream_t stream) {
    if (!dst || !src) return adamml_set_error(ADAMML_EINVAL, "copy2d: null argument");
    if (!width_bytes || !rows) return ADAMML_OK;
    const hipError_t e = hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, width_bytes, rows, hipMemcpyDeviceToDevice, stream);
    if (e != hipSuccess) return adamml_set_error(ADAMML_ELAUNCH, "copy2d: %s", hipGetErrorString(e));
    return ADAMML_OK;
}
