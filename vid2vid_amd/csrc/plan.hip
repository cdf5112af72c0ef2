// Plan executor: a recorded sequence of launches that is replayed as one native call or as
// one hipGraph.  The Python layer builds the network once per (model, resolution) against
// preallocated buffers; per frame it only refreshes the input buffers and replays.  This is
// the MI355X replacement for the reference's per-op Python dispatch (test.py:41 ->
// Vid2VidModelG.inference -> ~240 ATen launches per frame).
#include "v2v_internal.h"
#include <cstdarg>
#include <cstdio>
#include <cstdlib>

namespace v2v {

static thread_local char g_err[512] = "";
static thread_local v2v_plan* g_recording = nullptr;
static thread_local int g_lane = 0;
static int g_dry_run = 0;        // v2v_set_dry_run: validate arguments, never launch (CPU-host drop-in tests)

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace v2v

struct v2v_plan {
    std::vector<std::unique_ptr<v2v::Op>> ops;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    // segmented form (v2v_plan_instantiate_graph, mode 1): one LINEAR graph per run of ops of a lane between cross-lane
    // edges, launched on one persistent stream per lane with events for the edges
    struct Step { int kind; int a, b; };                  // kind 0: launch segment a on lane b; 1: lane a waits for lane b
    std::vector<hipGraph_t> seg_graph;
    std::vector<hipGraphExec_t> seg_exec;
    std::vector<Step> program;
    hipStream_t lane_stream[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> edge_events;                   // one per kind-1 step, in program order
    bool segmented = false;
};

namespace v2v {

int submit(std::unique_ptr<Op> op, void* stream) {
    if (g_recording) {
        op->lane = g_lane;
        g_recording->ops.push_back(std::move(op));
        return 0;
    }
    if (g_dry_run) return 0;     // every entry point has validated its arguments by now; nothing may execute
    return op->launch(reinterpret_cast<hipStream_t>(stream));
}

}  // namespace v2v

namespace v2v {
// Lanes: independent branches of the per-frame computation (the label / image / foreground towers and the image / flow
// branches of CompositeGenerator, models/networks.py:203-232) are recorded on different lanes; when the plan becomes a
// hipGraph each lane is captured on its own stream, so the branches are parallel graph paths and the tail of one
// kernel overlaps the next kernels of the other branches.  A WaitOp is the only cross-lane edge: lane `waiter`
// continues after everything recorded so far on lane `signal`.  Eager replays (v2v_plan_run / v2v_plan_profile) run
// the ops in recording order on one stream, which is a valid topological order.
struct WaitOp : Op {
    int waiter = 0, signal = 0;
    int launch(hipStream_t) override { return 0; }
    const char* name() const override { return "lane_wait"; }
};
}  // namespace v2v

using namespace v2v;

extern "C" int v2v_plan_set_lane(int32_t lane) {
    if (lane < 0 || lane >= 8) { set_error("plan: lane out of range"); return V2V_EINVAL; }
    g_lane = lane;
    return 0;
}

extern "C" int v2v_plan_lane_wait(int32_t waiter, int32_t signal) {
    if (waiter < 0 || waiter >= 8 || signal < 0 || signal >= 8) { set_error("plan: lane out of range"); return V2V_EINVAL; }
    if (!g_recording || waiter == signal) return 0;            // eager execution is ordered already
    auto op = std::make_unique<WaitOp>();
    op->waiter = waiter; op->signal = signal; op->lane = waiter;
    g_recording->ops.push_back(std::move(op));
    return 0;
}

extern "C" v2v_plan* v2v_plan_create(void) { return new v2v_plan(); }

extern "C" void v2v_plan_destroy(v2v_plan* p) {
    if (!p) return;
    if (g_recording == p) g_recording = nullptr;
    if (p->exec) hipGraphExecDestroy(p->exec);
    if (p->graph) hipGraphDestroy(p->graph);
    for (auto e : p->seg_exec) if (e) hipGraphExecDestroy(e);
    for (auto g : p->seg_graph) if (g) hipGraphDestroy(g);
    for (auto ev : p->edge_events) hipEventDestroy(ev);
    delete p;                                       // (lane streams belong to the process-wide pool below)
}

// Lane streams are a PROCESS-WIDE pool (round 3): the runtime maps streams onto a small number of hardware queues round-robin in
// creation order, so a plan that created its own lane streams late in a process (after other plans, copy streams, capture streams)
// could find two of its lanes -- or a lane and the caller's stream -- on ONE hardware queue and lose their overlap: the 2048x1024
// frame measured 17.2 ms as the fourth plan of a process and 15.1 ms alone (profiles/r03_c1 vs r03_b9).  Created once, on first use,
// the lanes keep distinct queues for every plan of the process.  Plans of one process run one after the other on the caller's
// stream, so sharing the lane streams only adds the ordering they already had.
static hipStream_t pool_lane_stream(int k) {
    static hipStream_t pool[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (k <= 0 || k >= 8) return nullptr;
    if (!pool[k]) {
        for (int j = 1; j <= k; ++j)                 // in lane order: the mapping does not depend on which lane a plan touches first
            if (!pool[j] && hipStreamCreateWithFlags(&pool[j], hipStreamNonBlocking) != hipSuccess) { pool[j] = nullptr; return nullptr; }
    }
    return pool[k];
}

extern "C" int v2v_plan_begin_record(v2v_plan* p) {
    if (!p || g_recording) { set_error("plan: already recording"); return V2V_EINVAL; }
    g_recording = p;
    return 0;
}

extern "C" int v2v_plan_end_record(v2v_plan* p) {
    if (!p || g_recording != p) { set_error("plan: not recording this plan"); return V2V_EINVAL; }
    g_recording = nullptr;
    return 0;
}

extern "C" int v2v_plan_num_ops(const v2v_plan* p) { return p ? (int)p->ops.size() : 0; }

extern "C" int v2v_plan_run(v2v_plan* p, void* stream) {
    if (!p) return V2V_EINVAL;
    if (g_dry_run) return 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    for (auto& op : p->ops) {
        int rc = op->launch(s);
        if (rc != 0) return rc;
    }
    return 0;
}

// Segmented instantiation.  Measured with v2v_plan_timeline_graph (profiles/r02_a44_lane_timeline.txt): a single hipGraph
// with three parallel branches never ran more than TWO of them at a time -- the image tower's head, ready at 60 us, started
// at 700 us when the label tower's chain had finished (queue-count knobs of the runtime change nothing).  Here every lane is
// a real HIP stream, every run of ops of a lane between two cross-lane edges is a linear graph launched on it, and the edges
// are events: the same dependencies, scheduled by the hardware queues.
__global__ void stamp_kernel(unsigned long long* dst) { *dst = wall_clock64(); }

static void plan_drop_segments(v2v_plan* p) {
    for (auto e : p->seg_exec) if (e) hipGraphExecDestroy(e);
    for (auto g : p->seg_graph) if (g) hipGraphDestroy(g);
    for (auto ev : p->edge_events) hipEventDestroy(ev);
    p->seg_exec.clear(); p->seg_graph.clear(); p->edge_events.clear(); p->program.clear();
    p->segmented = false;
}

// The launch program of the segmented form, without touching the device: `segments[k]` = indices of the ops of segment k in
// recording order, `program` = the steps {0, segment, lane} / {1, waiter, signal} in issue order.  Per lane the ops keep their
// recording order; an edge closes the open segment of both lanes it touches; every lane that was forked joins lane 0 at the end.
static int plan_segment_program(const v2v_plan* p, std::vector<std::vector<size_t>>* segments, std::vector<v2v_plan::Step>* program) {
    bool joined[8] = {true, false, false, false, false, false, false, false};
    std::vector<size_t> open[8];
    auto close = [&](int lane) {
        if (open[lane].empty()) return;
        segments->push_back(open[lane]);
        program->push_back({0, (int)segments->size() - 1, lane});
        open[lane].clear();
    };
    for (size_t i = 0; i < p->ops.size(); ++i) {
        Op* op = p->ops[i].get();
        if (WaitOp* w = dynamic_cast<WaitOp*>(op)) {
            if (!joined[w->signal]) { set_error("plan: lane %d waits for lane %d, which has no work yet", w->waiter, w->signal); return V2V_EINVAL; }
            close(w->signal);                                // the event is recorded behind everything the signalling lane has so far
            close(w->waiter);                                // and the waiter's later ops start a new segment behind the wait
            joined[w->waiter] = true;
            program->push_back({1, w->waiter, w->signal});
        } else if (op->lane < 0 || op->lane >= 8 || !joined[op->lane]) {
            set_error("plan: op '%s' recorded on lane %d before the lane was forked", op->name(), op->lane); return V2V_EINVAL;
        } else open[op->lane].push_back(i);
    }
    for (int k = 1; k < 8; ++k) close(k);
    close(0);
    for (int k = 1; k < 8; ++k)
        if (joined[k]) program->push_back({1, 0, k});       // every lane joins lane 0 at the end
    return 0;
}

static int plan_instantiate_segments(v2v_plan* p, unsigned long long* stamps = nullptr) {
    plan_drop_segments(p);
    std::vector<std::vector<size_t>> segments;
    int rc = plan_segment_program(p, &segments, &p->program);
    if (rc != 0) { p->program.clear(); return rc; }
    hipStream_t cs = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
    if (e != hipSuccess) { set_error("plan: capture stream: %s", hipGetErrorString(e)); p->program.clear(); return (int)e; }
    for (const auto& seg : segments) {                       // one linear graph per segment
        hipError_t q = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
        if (q != hipSuccess) { set_error("plan: begin capture: %s", hipGetErrorString(q)); rc = (int)q; break; }
        for (size_t m : seg) {
            if (stamps) hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, cs, stamps + 2 * m);
            rc = p->ops[m]->launch(cs);
            if (stamps) hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, cs, stamps + 2 * m + 1);
            if (rc != 0) break;
        }
        hipGraph_t g = nullptr;
        hipError_t q2 = hipStreamEndCapture(cs, &g);
        if (rc != 0) { if (g) hipGraphDestroy(g); break; }
        if (q2 != hipSuccess) { set_error("plan: end capture: %s", hipGetErrorString(q2)); rc = (int)q2; break; }
        hipGraphExec_t ex = nullptr;
        q = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
        if (q != hipSuccess) { hipGraphDestroy(g); set_error("plan: instantiate: %s", hipGetErrorString(q)); rc = (int)q; break; }
        p->seg_graph.push_back(g); p->seg_exec.push_back(ex);
    }
    hipStreamDestroy(cs);
    if (rc != 0) { plan_drop_segments(p); return rc; }
    for (auto& st : p->program) {
        if (st.kind == 1) {
            hipEvent_t ev;
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { set_error("plan: lane event"); plan_drop_segments(p); return V2V_EINVAL; }
            p->edge_events.push_back(ev);
        }
        const int lanes_of_step[2] = {st.kind == 0 ? st.b : st.a, st.kind == 0 ? st.b : st.b};
        for (int k : lanes_of_step)
            if (k != 0 && !p->lane_stream[k] && !(p->lane_stream[k] = pool_lane_stream(k))) {
                set_error("plan: lane stream"); plan_drop_segments(p); return V2V_EINVAL;
            }
    }
    p->segmented = true;
    return 0;
}

static int plan_launch_segments(v2v_plan* p, hipStream_t s) {
    size_t ev = 0;
    for (const auto& st : p->program) {
        if (st.kind == 0) {
            hipStream_t ls = st.b == 0 ? s : p->lane_stream[st.b];
            hipError_t e = hipGraphLaunch(p->seg_exec[st.a], ls);
            if (e != hipSuccess) { set_error("plan: segment launch: %s", hipGetErrorString(e)); return (int)e; }
        } else {
            hipStream_t sw = st.a == 0 ? s : p->lane_stream[st.a], sg = st.b == 0 ? s : p->lane_stream[st.b];
            hipEvent_t e_ = p->edge_events[ev++];
            hipError_t e = hipEventRecord(e_, sg);
            if (e == hipSuccess) e = hipStreamWaitEvent(sw, e_, 0);
            if (e != hipSuccess) { set_error("plan: lane edge: %s", hipGetErrorString(e)); return (int)e; }
        }
    }
    return 0;
}

static int graph_mode() {        // V2V_GRAPH_MODE: "single" = one hipGraph with parallel branches, default = one linear graph per lane segment
    static const int m = [] { const char* e = getenv("V2V_GRAPH_MODE"); return (e && e[0] == 's' && e[1] == 'i') ? 0 : 1; }();
    return m;
}

extern "C" int v2v_plan_instantiate_graph(v2v_plan* p, void* stream) {
    if (!p) return V2V_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (p->exec) { hipGraphExecDestroy(p->exec); p->exec = nullptr; }
    if (p->graph) { hipGraphDestroy(p->graph); p->graph = nullptr; }
    plan_drop_segments(p);
    {
        bool lanes_used = false;
        for (auto& op : p->ops) if (op->lane != 0) { lanes_used = true; break; }
        if (lanes_used && graph_mode() == 1) {
            int rc0 = v2v_plan_run(p, stream);               // one eager pass first (kernel attributes, code objects)
            if (rc0 != 0) return rc0;
            hipError_t e0 = hipStreamSynchronize(s);
            if (e0 != hipSuccess) { set_error("plan: warm-up failed: %s", hipGetErrorString(e0)); return (int)e0; }
            return plan_instantiate_segments(p);
        }
    }
    // one eager pass first: sets per-kernel attributes and faults in code objects outside capture
    int rc = v2v_plan_run(p, stream);
    if (rc != 0) return rc;
    hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) { set_error("plan: warm-up failed: %s", hipGetErrorString(e)); return (int)e; }
    // capture on a private stream: the caller's stream may be the legacy null stream, which
    // cannot be captured (hipErrorStreamCaptureUnsupported)
    hipStream_t cs = nullptr;
    e = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
    if (e != hipSuccess) { set_error("plan: capture stream: %s", hipGetErrorString(e)); return (int)e; }
    e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { set_error("plan: begin capture: %s", hipGetErrorString(e)); hipStreamDestroy(cs); return (int)e; }
    // lanes -> streams of the same capture: a lane joins when its first WaitOp makes it wait for an event of a lane
    // that is already capturing; every lane is joined back into lane 0 before the capture ends
    hipStream_t ls[8] = {cs, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool joined[8] = {true, false, false, false, false, false, false, false};
    std::vector<hipEvent_t> evs;
    auto edge = [&](int waiter, int signal) -> int {
        if (!joined[signal]) { set_error("plan: lane %d waits for lane %d, which has no work yet", waiter, signal); return V2V_EINVAL; }
        if (!ls[waiter] && hipStreamCreateWithFlags(&ls[waiter], hipStreamNonBlocking) != hipSuccess) { set_error("plan: lane stream"); return V2V_EINVAL; }
        hipEvent_t ev;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { set_error("plan: lane event"); return V2V_EINVAL; }
        evs.push_back(ev);
        hipError_t q = hipEventRecord(ev, ls[signal]);
        if (q == hipSuccess) q = hipStreamWaitEvent(ls[waiter], ev, 0);
        if (q != hipSuccess) { set_error("plan: lane edge: %s", hipGetErrorString(q)); return (int)q; }
        joined[waiter] = true;
        return 0;
    };
    rc = 0;
    for (auto& op : p->ops) {
        if (WaitOp* w = dynamic_cast<WaitOp*>(op.get())) { rc = edge(w->waiter, w->signal); }
        else if (!joined[op->lane]) { set_error("plan: op '%s' recorded on lane %d before the lane was forked", op->name(), op->lane); rc = V2V_EINVAL; }
        else rc = op->launch(ls[op->lane]);
        if (rc != 0) break;
    }
    for (int k = 1; k < 8 && rc == 0; ++k)
        if (joined[k]) rc = edge(0, k);
    hipError_t e2 = hipStreamEndCapture(cs, &p->graph);
    for (int k = 0; k < 8; ++k) if (ls[k]) hipStreamDestroy(ls[k]);
    for (auto ev : evs) hipEventDestroy(ev);
    if (rc != 0) return rc;
    if (e2 != hipSuccess) { set_error("plan: end capture: %s", hipGetErrorString(e2)); return (int)e2; }
    e = hipGraphInstantiate(&p->exec, p->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) { set_error("plan: instantiate: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}

extern "C" int v2v_plan_segment_program(const v2v_plan* p, int32_t* steps, int32_t max_steps, int32_t* seg_of_op, int32_t n_ops) {
    if (!p || !steps || max_steps < 0) { set_error("plan: segment_program arguments"); return V2V_EINVAL; }
    std::vector<std::vector<size_t>> segments;
    std::vector<v2v_plan::Step> program;
    const int rc = plan_segment_program(p, &segments, &program);
    if (rc != 0) return rc;
    if ((int)program.size() > max_steps || (seg_of_op && n_ops < (int)p->ops.size())) { set_error("plan: segment_program buffers too small"); return V2V_EINVAL; }
    for (size_t i = 0; i < program.size(); ++i) { steps[3 * i] = program[i].kind; steps[3 * i + 1] = program[i].a; steps[3 * i + 2] = program[i].b; }
    if (seg_of_op) {
        for (size_t i = 0; i < p->ops.size(); ++i) seg_of_op[i] = -1;       // lane_wait ops belong to no segment
        for (size_t k = 0; k < segments.size(); ++k) for (size_t m : segments[k]) seg_of_op[m] = (int32_t)k;
    }
    return (int)program.size();
}

extern "C" int v2v_plan_launch_graph(v2v_plan* p, void* stream) {
    if (p && p->segmented) return plan_launch_segments(p, reinterpret_cast<hipStream_t>(stream));
    if (!p || !p->exec) { set_error("plan: graph not instantiated"); return V2V_EINVAL; }
    hipError_t e = hipGraphLaunch(p->exec, reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) { set_error("plan: graph launch: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}

extern "C" int v2v_plan_profile(v2v_plan* p, void* stream, float* ms, int32_t n) {
    if (!p || !ms || n < (int)p->ops.size()) { set_error("plan: profile buffer too small"); return V2V_EINVAL; }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t nops = p->ops.size();
    std::vector<hipEvent_t> ev(nops + 1);
    for (auto& e : ev) hipEventCreate(&e);
    hipEventRecord(ev[0], s);
    int rc = 0;
    for (size_t i = 0; i < nops && rc == 0; ++i) {
        rc = p->ops[i]->launch(s);
        hipEventRecord(ev[i + 1], s);
    }
    hipStreamSynchronize(s);
    if (rc == 0)
        for (size_t i = 0; i < nops; ++i) hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]);
    for (auto& e : ev) hipEventDestroy(e);
    return rc;
}

// Concurrent timeline: the plan replayed eagerly with every lane on its own stream (the edges the graph capture uses) and a
// one-thread kernel that stores the device's constant-rate wall clock before and after every op on ITS stream (HIP timing
// events of different streams gave inconsistent, even negative, intervals); t0 / t1 = start / end of op i in ms since the
// replay began.  The stamps cost a few us per op, so the replay is slower than the graph, but it shows which lane waits for
// which -- rocprofv3's kernel trace serialises kernels and shows every kernel alone.

extern "C" int v2v_plan_timeline(v2v_plan* p, void* stream, float* t0, float* t1, int32_t* lanes, int32_t n) {
    if (!p || !t0 || !t1 || n < (int)p->ops.size()) { set_error("plan: timeline buffers too small"); return V2V_EINVAL; }
    if (g_dry_run) return 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t nops = p->ops.size();
    hipStream_t ls[8] = {s, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool joined[8] = {true, false, false, false, false, false, false, false};
    std::vector<hipEvent_t> edges;
    unsigned long long* dclk = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&dclk), (2 * nops + 1) * sizeof(unsigned long long)) != hipSuccess) { set_error("plan: timeline buffer"); return V2V_EINVAL; }
    hipMemsetAsync(dclk, 0, (2 * nops + 1) * sizeof(unsigned long long), s);
    auto edge = [&](int waiter, int signal) -> int {
        if (!joined[signal]) { set_error("plan: lane %d waits for lane %d, which has no work yet", waiter, signal); return V2V_EINVAL; }
        if (!ls[waiter] && hipStreamCreateWithFlags(&ls[waiter], hipStreamNonBlocking) != hipSuccess) { set_error("plan: lane stream"); return V2V_EINVAL; }
        hipEvent_t ev;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { set_error("plan: lane event"); return V2V_EINVAL; }
        edges.push_back(ev);
        hipEventRecord(ev, ls[signal]);
        hipStreamWaitEvent(ls[waiter], ev, 0);
        joined[waiter] = true;
        return 0;
    };
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, s, dclk + 2 * nops);
    int rc = 0;
    for (size_t i = 0; i < nops && rc == 0; ++i) {
        Op* op = p->ops[i].get();
        if (lanes) lanes[i] = op->lane;
        if (WaitOp* w = dynamic_cast<WaitOp*>(op)) {
            rc = edge(w->waiter, w->signal);
            if (rc == 0) { hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, ls[w->waiter], dclk + 2 * i); hipMemcpyAsync(dclk + 2 * i + 1, dclk + 2 * i, 8, hipMemcpyDeviceToDevice, ls[w->waiter]); }
        } else if (!joined[op->lane]) { set_error("plan: op on lane %d before the lane was forked", op->lane); rc = V2V_EINVAL; }
        else {
            hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, ls[op->lane], dclk + 2 * i);
            rc = op->launch(ls[op->lane]);
            hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, ls[op->lane], dclk + 2 * i + 1);
        }
    }
    for (int k = 1; k < 8 && rc == 0; ++k)
        if (joined[k]) rc = edge(0, k);
    hipStreamSynchronize(s);
    for (int k = 1; k < 8; ++k) if (ls[k]) { hipStreamSynchronize(ls[k]); hipStreamDestroy(ls[k]); }
    if (rc == 0) {
        std::vector<unsigned long long> h(2 * nops + 1);
        hipMemcpy(h.data(), dclk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        int khz = 100000, dev = 0;
        hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) { (void)hipGetLastError(); khz = 100000; }
        const unsigned long long base = h[2 * nops];
        for (size_t i = 0; i < nops; ++i) {
            t0[i] = (float)((double)(long long)(h[2 * i] - base) / (double)khz);
            t1[i] = (float)((double)(long long)(h[2 * i + 1] - base) / (double)khz);
        }
    }
    for (auto ev : edges) hipEventDestroy(ev);
    hipFree(dclk);
    return rc;
}

// The same stamps captured INTO a hipGraph (one stream per lane, as v2v_plan_instantiate_graph) and launched as a graph: the
// schedule of the real frame replay, without the host's issue order of an eager replay.  The graph is launched twice, the
// second launch is reported.
extern "C" int v2v_plan_timeline_graph(v2v_plan* p, void* stream, float* t0, float* t1, int32_t* lanes, int32_t n) {
    if (!p || !t0 || !t1 || n < (int)p->ops.size()) { set_error("plan: timeline buffers too small"); return V2V_EINVAL; }
    if (g_dry_run) return 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t nops = p->ops.size();
    unsigned long long* dclk = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&dclk), (2 * nops + 1) * sizeof(unsigned long long)) != hipSuccess) { set_error("plan: timeline buffer"); return V2V_EINVAL; }
    hipMemset(dclk, 0, (2 * nops + 1) * sizeof(unsigned long long));
    if (p->segmented) {                                        // the plan as it is replayed: per-lane segment graphs on real streams
        int rc = plan_instantiate_segments(p, dclk);
        for (int rep = 0; rep < 2 && rc == 0; ++rep) {
            hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, s, dclk + 2 * nops);
            rc = plan_launch_segments(p, s);
        }
        hipStreamSynchronize(s);
        if (rc == 0) {
            std::vector<unsigned long long> h(2 * nops + 1);
            hipMemcpy(h.data(), dclk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            int khz = 100000, dev = 0;
            hipGetDevice(&dev);
            if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) { (void)hipGetLastError(); khz = 100000; }
            const unsigned long long base = h[2 * nops];
            for (size_t i = 0; i < nops; ++i) {
                if (lanes) lanes[i] = p->ops[i]->lane;
                t0[i] = (float)((double)(long long)(h[2 * i] - base) / (double)khz);
                t1[i] = (float)((double)(long long)(h[2 * i + 1] - base) / (double)khz);
            }
        }
        const int rc2 = plan_instantiate_segments(p);           // back to the stamp-free program
        hipFree(dclk);
        return rc != 0 ? rc : rc2;
    }
    hipStream_t cs = nullptr;
    hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
    hipError_t e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { set_error("plan: begin capture: %s", hipGetErrorString(e)); hipStreamDestroy(cs); hipFree(dclk); return (int)e; }
    hipStream_t ls[8] = {cs, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool joined[8] = {true, false, false, false, false, false, false, false};
    std::vector<hipEvent_t> evs;
    auto edge = [&](int waiter, int signal) -> int {
        if (!joined[signal]) { set_error("plan: lane %d waits for lane %d, which has no work yet", waiter, signal); return V2V_EINVAL; }
        if (!ls[waiter] && hipStreamCreateWithFlags(&ls[waiter], hipStreamNonBlocking) != hipSuccess) { set_error("plan: lane stream"); return V2V_EINVAL; }
        hipEvent_t ev;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { set_error("plan: lane event"); return V2V_EINVAL; }
        evs.push_back(ev);
        hipError_t q = hipEventRecord(ev, ls[signal]);
        if (q == hipSuccess) q = hipStreamWaitEvent(ls[waiter], ev, 0);
        if (q != hipSuccess) { set_error("plan: lane edge: %s", hipGetErrorString(q)); return (int)q; }
        joined[waiter] = true;
        return 0;
    };
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, cs, dclk + 2 * nops);
    int rc = 0;
    for (size_t i = 0; i < nops && rc == 0; ++i) {
        Op* op = p->ops[i].get();
        if (lanes) lanes[i] = op->lane;
        if (WaitOp* w = dynamic_cast<WaitOp*>(op)) rc = edge(w->waiter, w->signal);
        else if (!joined[op->lane]) { set_error("plan: op on lane %d before the lane was forked", op->lane); rc = V2V_EINVAL; }
        else {
            hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, ls[op->lane], dclk + 2 * i);
            rc = op->launch(ls[op->lane]);
            hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, ls[op->lane], dclk + 2 * i + 1);
        }
    }
    for (int k = 1; k < 8 && rc == 0; ++k)
        if (joined[k]) rc = edge(0, k);
    hipGraph_t g = nullptr;
    hipError_t e2 = hipStreamEndCapture(cs, &g);
    for (int k = 0; k < 8; ++k) if (ls[k]) hipStreamDestroy(ls[k]);
    for (auto ev : evs) hipEventDestroy(ev);
    hipGraphExec_t ex = nullptr;
    if (rc == 0 && e2 != hipSuccess) { set_error("plan: end capture: %s", hipGetErrorString(e2)); rc = (int)e2; }
    if (rc == 0 && (e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0)) != hipSuccess) { set_error("plan: instantiate: %s", hipGetErrorString(e)); rc = (int)e; }
    if (rc == 0) {
        hipGraphLaunch(ex, s); hipGraphLaunch(ex, s);
        hipStreamSynchronize(s);
        std::vector<unsigned long long> h(2 * nops + 1);
        hipMemcpy(h.data(), dclk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        int khz = 100000, dev = 0;
        hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) { (void)hipGetLastError(); khz = 100000; }
        const unsigned long long base = h[2 * nops];
        for (size_t i = 0; i < nops; ++i) {
            t0[i] = (float)((double)(long long)(h[2 * i] - base) / (double)khz);
            t1[i] = (float)((double)(long long)(h[2 * i + 1] - base) / (double)khz);
        }
    }
    if (ex) hipGraphExecDestroy(ex);
    if (g) hipGraphDestroy(g);
    hipFree(dclk);
    return rc;
}

extern "C" const char* v2v_plan_op_name(const v2v_plan* p, int32_t i) {
    if (!p || i < 0 || i >= (int)p->ops.size()) return "";
    return p->ops[i]->name();
}

extern "C" int v2v_plan_op_lane(const v2v_plan* p, int32_t i) {
    if (!p || i < 0 || i >= (int)p->ops.size()) return V2V_EINVAL;
    if (const WaitOp* w = dynamic_cast<const WaitOp*>(p->ops[i].get())) return w->waiter | (w->signal << 8);     // lane_wait: waiter | signal << 8
    return p->ops[i]->lane;
}

extern "C" int v2v_plan_set_label(v2v_plan* p, const char* label) {
    // labels the most recently recorded op (layer name for per-op reports)
    if (!p || p->ops.empty() || !label) return V2V_EINVAL;
    p->ops.back()->label = label;
    return 0;
}

extern "C" const char* v2v_plan_op_label(const v2v_plan* p, int32_t i) {
    if (!p || i < 0 || i >= (int)p->ops.size()) return "";
    return p->ops[i]->label.c_str();
}

extern "C" int v2v_set_dry_run(int32_t on) { const int prev = g_dry_run; g_dry_run = on ? 1 : 0; return prev; }
extern "C" int v2v_get_dry_run(void) { return g_dry_run; }
extern "C" int v2v_version(void) { return 101; }
extern "C" const char* v2v_last_error(void) { return g_err; }

extern "C" int v2v_device_info(int32_t* cus, int32_t* lds_per_cu, int64_t* hbm_bytes, char* arch, int32_t arch_len) {
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        set_error("no HIP device");
        return V2V_EINVAL;
    }
    if (cus) *cus = prop.multiProcessorCount;
    if (lds_per_cu) *lds_per_cu = (int32_t)prop.maxSharedMemoryPerMultiProcessor;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    if (arch && arch_len > 0) snprintf(arch, arch_len, "%s", prop.gcnArchName);
    return 0;
}

// Device-to-device copy as a recordable op (rolling fake_B_prev window,
// models/vid2vid_model_G.py:228, and input staging inside a plan).
namespace v2v {
struct CopyOp : Op {
    void* dst; const void* src; size_t bytes;
    int launch(hipStream_t s) override {
        hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) { set_error("memcpy_d2d: %s", hipGetErrorString(e)); return (int)e; }
        return 0;
    }
    const char* name() const override { return "memcpy_d2d"; }
};
}  // namespace v2v

extern "C" int v2v_memcpy_d2d(void* dst, const void* src, int64_t bytes, void* stream) {
    if (!dst || !src || bytes < 0) { set_error("memcpy_d2d: bad argument"); return V2V_EINVAL; }
    auto op = std::make_unique<CopyOp>();
    op->dst = dst; op->src = src; op->bytes = (size_t)bytes;
    return submit(std::move(op), stream);
}
