// Launch plans: a training step as ONE library call (include/mcamd.h, "Launch plans").
//
// The reference's step is one Python call per torch module (src/nets.py:720-774); round 2's engine issued its 241 kernel
// launches one ctypes call at a time and spent 5.4-5.8 ms of host time per dense step -- next to a 6.2 ms GPU step at the
// per-GPU batch of BASELINE configs[3] (B = 32).  A plan records the launch-type entry points (arguments copied by value,
// descriptors included) while the engine walks its layer list ONCE, and replays them from C: the per-launch cost drops
// to the argument checks + hipLaunchKernel.  A plan also carries the cross-stream ordering of the two-stream backward
// (event record on one stream, wait on the other: events owned by the plan) and segment marks, so that the data-parallel
// reducer can be called between segments (gradient slices become final segment by segment).
//
// Nothing here allocates device memory or synchronises; a plan holds raw device pointers and is only valid while the
// caller keeps those buffers in place (engine.py re-records when a pointer or the channel compaction changes).
#include <functional>
#include <vector>

#include "common.h"

struct McamdPlanOp {
    int kind;                    // 0: launch-type call, 1: cross-stream wait (waiter slot `slot`, signal slot `slot2`)
    int slot, slot2;
    hipEvent_t ev;
    std::function<int(void*)> fn;
};

struct mcamd_plan {
    std::vector<McamdPlanOp> ops;
    std::vector<int> seg_start;        // first op of every segment; seg_start.size() == segments (the last one may be empty)
    std::vector<void*> rec_streams;    // the streams the recording saw, by slot
    bool failed = false;
};

thread_local mcamd_plan* g_mcamd_rec = nullptr;

static int slot_of(mcamd_plan* p, void* stream) {
    for (size_t i = 0; i < p->rec_streams.size(); ++i)
        if (p->rec_streams[i] == stream) return (int)i;
    return -1;
}

int mcamd_rec_push(void* stream, std::function<int(void*)> fn) {
    mcamd_plan* p = g_mcamd_rec;
    const int s = slot_of(p, stream);
    if (s < 0) {
        p->failed = true;
        mcamd_set_error("plan: a call was recorded on a stream that mcamd_plan_begin was not given");
        return MCAMD_EINVAL;
    }
    McamdPlanOp op;
    op.kind = 0, op.slot = s, op.slot2 = 0, op.ev = nullptr;
    op.fn = std::move(fn);
    p->ops.push_back(std::move(op));
    return MCAMD_OK;
}

extern "C" int mcamd_plan_begin(void* const* streams, int32_t nstreams) {
    MCAMD_REQUIRE(!g_mcamd_rec, "plan_begin: this host thread is already recording");
    MCAMD_REQUIRE(streams && nstreams > 0 && nstreams <= 8, "plan_begin: 1..8 streams");
    mcamd_plan* p = new mcamd_plan();
    for (int i = 0; i < nstreams; ++i) p->rec_streams.push_back(streams[i]);
    p->seg_start.push_back(0);
    g_mcamd_rec = p;
    return MCAMD_OK;
}

extern "C" int32_t mcamd_plan_mark(void) {
    if (!g_mcamd_rec) {
        mcamd_set_error("plan_mark: not recording");
        return MCAMD_EINVAL;
    }
    g_mcamd_rec->seg_start.push_back((int)g_mcamd_rec->ops.size());
    return (int32_t)g_mcamd_rec->seg_start.size() - 1;      // index of the segment that starts here
}

extern "C" void mcamd_plan_destroy(mcamd_plan* p) {
    if (!p) return;
    for (auto& op : p->ops)
        if (op.ev) (void)hipEventDestroy(op.ev);
    delete p;
}

extern "C" mcamd_plan* mcamd_plan_end(void) {
    mcamd_plan* p = g_mcamd_rec;
    g_mcamd_rec = nullptr;
    if (!p) {
        mcamd_set_error("plan_end: not recording");
        return nullptr;
    }
    if (p->failed) {
        mcamd_plan_destroy(p);
        return nullptr;
    }
    return p;
}

extern "C" int32_t mcamd_plan_segments(const mcamd_plan* p) { return p ? (int32_t)p->seg_start.size() : 0; }
extern "C" int32_t mcamd_plan_launches(const mcamd_plan* p) { return p ? (int32_t)p->ops.size() : 0; }

extern "C" int mcamd_plan_run(mcamd_plan* p, int32_t seg_lo, int32_t seg_hi, void* const* streams, int32_t nstreams) {
    MCAMD_REQUIRE(p && !g_mcamd_rec, "plan_run: null plan, or called while recording");
    const int nseg = (int)p->seg_start.size();
    MCAMD_REQUIRE(seg_lo >= 0 && seg_lo <= seg_hi && seg_hi <= nseg, "plan_run: segments [%d, %d) of %d", seg_lo, seg_hi, nseg);
    MCAMD_REQUIRE(streams && nstreams == (int)p->rec_streams.size(), "plan_run: the plan was recorded with %d streams",
                  (int)p->rec_streams.size());
    if (seg_lo == seg_hi) return MCAMD_OK;
    const int lo = p->seg_start[seg_lo], hi = seg_hi == nseg ? (int)p->ops.size() : p->seg_start[seg_hi];
    for (int i = lo; i < hi; ++i) {
        McamdPlanOp& op = p->ops[i];
        if (op.kind == 0) {
            const int rc = op.fn(streams[op.slot]);
            if (rc) return rc;
        } else {
            if (hipEventRecord(op.ev, (hipStream_t)streams[op.slot2]) != hipSuccess ||
                hipStreamWaitEvent((hipStream_t)streams[op.slot], op.ev, 0) != hipSuccess) {
                mcamd_set_error("plan_run: event record / wait failed: %s", hipGetErrorString(hipGetLastError()));
                return MCAMD_ELAUNCH;
            }
        }
    }
    return MCAMD_OK;
}

// `waiter` waits for everything enqueued so far on `signal` (an event record + a stream wait).
extern "C" int mcamd_stream_wait(void* waiter, void* signal) {
    if (waiter == signal) return MCAMD_OK;
    hipEvent_t ev;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
        mcamd_set_error("stream_wait: hipEventCreate failed");
        return MCAMD_ELAUNCH;
    }
    if (g_mcamd_rec) {
        mcamd_plan* p = g_mcamd_rec;
        const int sw = slot_of(p, waiter), ss = slot_of(p, signal);
        if (sw < 0 || ss < 0) {
            (void)hipEventDestroy(ev);
            p->failed = true;
            mcamd_set_error("plan: stream_wait on a stream that mcamd_plan_begin was not given");
            return MCAMD_EINVAL;
        }
        McamdPlanOp op;
        op.kind = 1, op.slot = sw, op.slot2 = ss, op.ev = ev;
        p->ops.push_back(std::move(op));
        return MCAMD_OK;
    }
    hipError_t e = hipEventRecord(ev, (hipStream_t)signal);
    if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)waiter, ev, 0);
    (void)hipEventDestroy(ev);      // released by the runtime once the recorded work has completed
    if (e != hipSuccess) {
        mcamd_set_error("stream_wait: %s", hipGetErrorString(e));
        return MCAMD_ELAUNCH;
    }
    return MCAMD_OK;
}

static int memset_impl(void* dst, size_t bytes, void* stream) {
    if (hipMemsetAsync(dst, 0, bytes, (hipStream_t)stream) != hipSuccess) {
        mcamd_set_error("memset_zero: %s", hipGetErrorString(hipGetLastError()));
        return MCAMD_ELAUNCH;
    }
    return MCAMD_OK;
}

extern "C" int mcamd_memset_zero(void* dst, size_t bytes, void* stream) {
    MCAMD_REQUIRE(dst || bytes == 0, "memset_zero: null destination");
    if (bytes == 0) return MCAMD_OK;
    if (mcamd_recording()) return mcamd_rec_push(stream, [=](void* s) { return memset_impl(dst, bytes, s); });
    return memset_impl(dst, bytes, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// step flags (train.py StepGuard): one launch instead of a dozen torch micro-kernels between backward and the optimizer
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct StepFlagArgs {
    int32_t* eng[8];
    int n;
    const float* loss;
    int32_t* tflag;
    float* flags;
    float* found;
};
__global__ void step_flags_kernel(StepFlagArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (a.n > 0 || a.loss || a.tflag) {
        float over = 0.f, bad = 0.f, tover = 0.f;
        for (int i = 0; i < a.n; ++i) {
            if (*a.eng[i] != 0) over = 1.f;
            *a.eng[i] = 0;
        }
        if (a.loss) {
            const float l = *a.loss;
            bad = (l == l && fabsf(l) <= 3.402823466e38f) ? 0.f : 1.f;
        }
        if (a.tflag) {
            tover = *a.tflag != 0 ? 1.f : 0.f;
            *a.tflag = 0;
        }
        a.flags[0] = over, a.flags[1] = bad, a.flags[2] = tover;
    }
    // torch's fused optimizers skip the update only when *found_inf == 1.0 EXACTLY (fused_adam_utils.cuh / fused SGD:
    // `if (found_inf_ptr && *found_inf_ptr == 1) return;`): two flags in one step must not add up to 2
    if (a.found) *a.found = (a.flags[0] != 0.f || a.flags[1] != 0.f || a.flags[2] != 0.f) ? 1.f : 0.f;
}
}  // namespace

extern "C" int mcamd_step_flags(const int32_t* const* engine_overflow, int32_t n_engine, const float* loss,
                                int32_t* transport_overflow, float* flags, float* found, void* stream) {
    MCAMD_REQUIRE(flags, "step_flags: null flags");
    MCAMD_REQUIRE(n_engine >= 0 && n_engine <= 8 && (n_engine == 0 || engine_overflow), "step_flags: 0..8 engine flags");
    StepFlagArgs a = {};
    for (int i = 0; i < n_engine; ++i) {
        MCAMD_REQUIRE(engine_overflow[i], "step_flags: null engine flag");
        a.eng[i] = (int32_t*)engine_overflow[i];
    }
    a.n = n_engine, a.loss = loss, a.tflag = transport_overflow, a.flags = flags, a.found = found;
    if (mcamd_recording())
        return mcamd_rec_push(stream, [=](void* s) {
            hipLaunchKernelGGL(step_flags_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, a);
            return MCAMD_OK;
        });
    hipLaunchKernelGGL(step_flags_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
    MCAMD_LAUNCH_CHECK("step_flags");
    return MCAMD_OK;
}
