// plan.hip -- launch plans: a whole step (hundreds of dependent kernel launches on two lanes) as ONE call over the C ABI.
//
// What the reference gets from `theano.function` -- train_model(index, lr) is a single compiled device function,
// /root/reference/src/trainer/poseregnettrainer.py:146-170, invoked once per minibatch at
// /root/reference/src/trainer/nettrainer.py:840 -- is a recorded list of launches here.  The bs128 ResNet step is 400+
// launches of 5-25 us kernels: issued one by one from Python (ctypes call + descriptor marshalling per launch) the host is
// slower than the GPU and every dependent kernel waits ~5 us for its successor to arrive.  A plan re-issues the same
// launches from a tight C++ loop (one hipLaunchKernel each, arguments already packed), or replays them as an explicit
// hipGraph whose two lanes are parallel branches.
#include "dpp_common.h"
#ifndef DPP_HIP_EMU
#include <hip/hip_ext.h>
#endif
// (host code only; the CPU test emulator spells these two built-ins as macros, which would clash with hipKernelNodeParams)
#undef gridDim
#undef blockDim

thread_local dpp_plan* dpp_tls_plan = nullptr;
unsigned long long* dpp_prof_buffer = nullptr;

// tools/phase_profile.py (profiling build only): device buffer of 16 uint64 per workgroup that instrumented kernels stamp
extern "C" int dpp_prof_set(void* buf) {
    dpp_prof_buffer = static_cast<unsigned long long*>(buf);
    return DPP_OK;
}

struct dpp_plan {
    std::vector<dpp_plan_node> nodes;
    int lane = 0;
    bool recording = false;
    int launches = 0, forks = 0, joins = 0;
    std::vector<hipEvent_t> events;          // one per fork / join, created lazily by the first run
    std::vector<char> last_before_join;      // per node: lane-1 kernel with a join (and no other lane-1 node) ahead of it
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

void dpp_plan_append(dpp_plan* plan, dpp_plan_node&& node) {
    node.lane = plan->lane;
    if (node.kind == 0 || node.kind == 1) plan->launches++;
    plan->nodes.push_back(std::move(node));
}

static void plan_drop_graph(dpp_plan* p) {
    if (p->exec) (void)hipGraphExecDestroy(p->exec);
    if (p->graph) (void)hipGraphDestroy(p->graph);
    p->exec = nullptr;
    p->graph = nullptr;
}

extern "C" int dpp_plan_create(dpp_plan** out) {
    if (!out) return DPP_E_BADARG;
    *out = new dpp_plan();
    return DPP_OK;
}

extern "C" int dpp_plan_destroy(dpp_plan* plan) {
    if (!plan) return DPP_E_BADARG;
    if (dpp_tls_plan == plan) dpp_tls_plan = nullptr;
    plan_drop_graph(plan);
    for (hipEvent_t e : plan->events) (void)hipEventDestroy(e);
    delete plan;
    return DPP_OK;
}

extern "C" int dpp_plan_record_begin(dpp_plan* plan) {
    if (!plan || dpp_tls_plan != nullptr) return DPP_E_BADARG;      // one recording per thread
    plan->recording = true;
    plan->lane = 0;
    dpp_tls_plan = plan;
    return DPP_OK;
}

extern "C" int dpp_plan_record_lane(dpp_plan* plan, int lane) {
    if (!plan || dpp_tls_plan != plan || lane < 0 || lane > 1) return DPP_E_BADARG;
    plan->lane = lane;
    return DPP_OK;
}

extern "C" int dpp_plan_record_end(dpp_plan* plan) {
    if (!plan || dpp_tls_plan != plan) return DPP_E_BADARG;
    dpp_tls_plan = nullptr;
    plan->recording = false;
    plan->lane = 0;
    return DPP_OK;
}

static int plan_marker(dpp_plan* plan, int kind) {
    if (!plan || dpp_tls_plan != plan) return DPP_E_BADARG;
    dpp_plan_node n;
    n.kind = kind;
    const int lane = plan->lane;
    plan->lane = 0;
    dpp_plan_append(plan, std::move(n));
    plan->lane = lane;
    (kind == 2 ? plan->forks : plan->joins)++;
    return DPP_OK;
}

extern "C" int dpp_plan_fork(dpp_plan* plan) { return plan_marker(plan, 2); }
extern "C" int dpp_plan_join(dpp_plan* plan) { return plan_marker(plan, 3); }

extern "C" int dpp_plan_count(const dpp_plan* plan, int* launches, int* forks, int* joins) {
    if (!plan) return DPP_E_BADARG;
    if (launches) *launches = plan->launches;
    if (forks) *forks = plan->forks;
    if (joins) *joins = plan->joins;
    return DPP_OK;
}

// Eager issue.  A fork is (record event on main, side waits for it), a join the mirror image; markers whose waiting lane has
// nothing to wait for (no launch on the other lane since the previous marker of that kind) are skipped.
//
// Lane 1 can be spread over SEVERAL side streams (DPP_SIDE_STREAMS = S > 1): the launches between two forks are one group
// (a layer's bias / filter gradient: they only read what the main chain has produced and write buffers of their own), groups
// are dealt round-robin to the S streams, so filter gradients of different layers overlap each other as well as the main
// chain.  Seen from outside nothing changes: a join makes main wait for every side stream, and before the call returns the
// extra streams are folded into `side_stream`, so whatever the caller issues there next (a collective) is behind all of lane 1.
static int plan_side_streams() {
    static const int n = []() {
        const char* e = getenv("DPP_SIDE_STREAMS");
        int v = e ? atoi(e) : 1;
        return v < 1 ? 1 : (v > 8 ? 8 : v);
    }();
    return n;
}

static hipStream_t plan_extra_stream(int i) {      // i >= 1; created on first use, live for the process
    static hipStream_t pool[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (!pool[i]) {
        if (hipStreamCreateWithFlags(&pool[i], hipStreamNonBlocking) != hipSuccess) pool[i] = nullptr;
    }
    return pool[i];
}

extern "C" int dpp_plan_run(dpp_plan* plan, dpp_stream_t main_stream, dpp_stream_t side_stream) {
    if (!plan || plan->recording) return DPP_E_BADARG;
    hipStream_t ms = static_cast<hipStream_t>(main_stream);
    hipStream_t ss = static_cast<hipStream_t>(side_stream);
    const bool two = side_stream != nullptr && ss != ms;
    int S = two ? plan_side_streams() : 1;
    hipStream_t sides[8] = {ss, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    for (int i = 1; i < S; ++i) {
        sides[i] = plan_extra_stream(i);
        if (!sides[i]) { S = i; break; }
    }
    size_t ev = 0;
    auto next_event = [&](hipEvent_t* e) -> hipError_t {
        if (ev == plan->events.size()) {
            hipEvent_t n;
            // device-scope release: the events only order streams of this GPU against each other (a default event releases to
            // system scope, i.e. to the host as well, when it is recorded)
            static const unsigned flags = []() {
                const char* e = getenv("DPP_EVENT_RELEASE");
                return (unsigned)hipEventDisableTiming | ((e && e[0] == 's') ? 0u : (unsigned)hipEventReleaseToDevice);
            }();
            hipError_t err = hipEventCreateWithFlags(&n, flags);
            if (err != hipSuccess) return err;
            plan->events.push_back(n);
        }
        *e = plan->events[ev++];
        return hipSuccess;
    };
    // work issued on a lane since the other lane last synchronised with it.  Main and the caller's side stream start TRUE: a plan
    // may be one segment of a longer sequence (collectives run between segments), so the lane may carry work from before this
    // call; the extra streams were folded into the side stream at the end of the previous call.
    bool main_dirty = true;
    bool side_dirty[8] = {true, false, false, false, false, false, false, false};
    hipEvent_t fork_event = nullptr;         // latest event recorded on main; covers all main work iff !main_dirty
    // A fork that directly follows a main-lane kernel takes that kernel's own completion signal as its event (hipExtLaunchKernel's
    // stop event) instead of a hipEventRecord: a recorded event is a marker packet of its own in the main queue, and the next
    // kernel of the chain waits for it -- measured 4.5-5.3 us per fork on the critical path (tools/join_probe.py), 60 forks per step.
    // (A kernel that carries a signal still costs the chain ~1.5 us.  Thinning the forks -- only every k-th one real, the groups in
    // between issued behind the next real fork -- was measured and is worse: 4.09 / 4.10 / 4.14 / 4.19 ms for k = 2 / 3 / 4 / 6 against
    // 4.07: the gradient branch ends together with the main chain, so every delay of its work shows at the final join.)
    static const bool stop_events = []() { const char* e = getenv("DPP_FORK_STOP_EVENT"); return !(e && e[0] == '0'); }();
    const size_t nn = plan->nodes.size();
    // the same for a join: the last lane-1 kernel in front of it is launched with a stop event and main waits for that
    if (plan->last_before_join.size() != nn) {
        plan->last_before_join.assign(nn, 0);
        bool join_next = false;              // walking backwards: a join lies ahead and no lane-1 node in between
        for (size_t i = nn; i-- > 0;) {
            const dpp_plan_node& q = plan->nodes[i];
            if (q.kind == 3) join_next = true;
            else if ((q.kind == 0 || q.kind == 1) && q.lane == 1) {
                plan->last_before_join[i] = join_next && q.kind == 0 && q.func != nullptr;
                join_next = false;
            }
        }
    }
    hipEvent_t side_stop[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // stop event of the stream's LAST launch
    bool waited[8] = {false, false, false, false, false, false, false, false};      // side stream i already waits for fork_event
    int cur = 0, groups = 0;
    for (size_t ni = 0; ni < nn; ++ni) {
        dpp_plan_node& n = plan->nodes[ni];
        if (n.kind == 2) {                   // fork: the next lane-1 group starts here
            if (!two) continue;
            if (S > 1) cur = groups++ % S;
            if (main_dirty) {
                hipError_t err = next_event(&fork_event);
                if (err == hipSuccess) err = hipEventRecord(fork_event, ms);
                if (err != hipSuccess) return (int)err;
                main_dirty = false;
                for (int i = 0; i < S; ++i) waited[i] = false;
            }
            if (fork_event && !waited[cur]) {
                hipError_t err = hipStreamWaitEvent(sides[cur], fork_event, 0);
                if (err != hipSuccess) return (int)err;
                waited[cur] = true;
            }
            continue;
        }
        if (n.kind == 3) {                   // join: main waits for every side stream that has issued work
            if (!two) continue;
            for (int i = 0; i < S; ++i) {
                if (!side_dirty[i]) continue;
                hipEvent_t e = side_stop[i];
                hipError_t err = hipSuccess;
                if (e == nullptr) {
                    err = next_event(&e);
                    if (err == hipSuccess) err = hipEventRecord(e, sides[i]);
                }
                if (err == hipSuccess) err = hipStreamWaitEvent(ms, e, 0);
                if (err != hipSuccess) return (int)err;
                side_dirty[i] = false;
            }
            continue;
        }
        const bool on_side = two && n.lane == 1;
        hipStream_t s = on_side ? sides[cur] : ms;
        if (!on_side && two && stop_events && n.kind == 0 && n.func != nullptr && ni + 1 < nn && plan->nodes[ni + 1].kind == 2) {
            hipError_t err = next_event(&fork_event);
            if (err == hipSuccess)
                err = hipExtLaunchKernel(n.func, n.grid, n.block, n.argptrs.data(), n.shmem, ms, nullptr, fork_event, 0);
            if (err != hipSuccess) return (int)err;
            main_dirty = false;              // the event covers everything issued on main so far
            for (int i = 0; i < S; ++i) waited[i] = false;
            continue;
        }
        if (on_side && stop_events && plan->last_before_join[ni]) {
            hipError_t err = next_event(&side_stop[cur]);
            if (err == hipSuccess)
                err = hipExtLaunchKernel(n.func, n.grid, n.block, n.argptrs.data(), n.shmem, s, nullptr, side_stop[cur], 0);
            if (err != hipSuccess) return (int)err;
            side_dirty[cur] = true;
            continue;
        }
        hipError_t err = n.kind == 0 ? n.issue(s) : hipMemsetAsync(n.ptr, 0, n.nbytes, s);
        if (err != hipSuccess) return (int)err;
        if (on_side) { side_dirty[cur] = true; side_stop[cur] = nullptr; } else main_dirty = true;
    }
    for (int i = 1; i < S; ++i) {            // fold the extra streams into the caller's side stream
        if (!side_dirty[i]) continue;
        hipEvent_t e;
        hipError_t err = next_event(&e);
        if (err == hipSuccess) err = hipEventRecord(e, sides[i]);
        if (err == hipSuccess) err = hipStreamWaitEvent(ss, e, 0);
        if (err != hipSuccess) return (int)err;
    }
    return DPP_OK;
}

// Explicit graph: node i depends on the previous node of its lane, a side node additionally on the main node that was last
// when the most recent fork was recorded, a main node on the side node that was last at the most recent join.
extern "C" int dpp_plan_graph_build(dpp_plan* plan, int two_lanes) {
    if (!plan || plan->recording) return DPP_E_BADARG;
    plan_drop_graph(plan);
    hipError_t err = hipGraphCreate(&plan->graph, 0);
    if (err != hipSuccess) return (int)err;
    hipGraphNode_t last[2] = {nullptr, nullptr};
    hipGraphNode_t fork_dep = nullptr, join_dep = nullptr;      // pending cross-lane edges
    for (dpp_plan_node& n : plan->nodes) {
        if (n.kind == 2) { if (two_lanes) fork_dep = last[0]; continue; }
        if (n.kind == 3) { if (two_lanes) join_dep = last[1]; continue; }
        const int lane = two_lanes ? n.lane : 0;
        hipGraphNode_t deps[2];
        size_t nd = 0;
        if (last[lane]) deps[nd++] = last[lane];
        if (lane == 1 && fork_dep) { if (fork_dep != last[1]) deps[nd++] = fork_dep; fork_dep = nullptr; }
        if (lane == 0 && join_dep) { if (join_dep != last[0]) deps[nd++] = join_dep; join_dep = nullptr; }
        hipGraphNode_t node = nullptr;
        if (n.kind == 0) {
            hipKernelNodeParams kp = {};
            kp.func = const_cast<void*>(n.func);
            kp.gridDim = n.grid;
            kp.blockDim = n.block;
            kp.sharedMemBytes = (unsigned)n.shmem;
            kp.kernelParams = n.argptrs.data();
            kp.extra = nullptr;
            err = hipGraphAddKernelNode(&node, plan->graph, deps, nd, &kp);
        } else {
            hipMemsetParams mp = {};
            mp.dst = n.ptr;
            mp.value = 0;
            mp.elementSize = 1;
            mp.width = n.nbytes;
            mp.height = 1;
            mp.pitch = n.nbytes;
            err = hipGraphAddMemsetNode(&node, plan->graph, deps, nd, &mp);
        }
        if (err != hipSuccess) { plan_drop_graph(plan); return (int)err; }
        last[lane] = node;
    }
    err = hipGraphInstantiate(&plan->exec, plan->graph, nullptr, nullptr, 0);
    if (err != hipSuccess) { plan_drop_graph(plan); return (int)err; }
    return DPP_OK;
}

extern "C" int dpp_plan_graph_launch(dpp_plan* plan, dpp_stream_t stream) {
    if (!plan || !plan->exec) return DPP_E_BADARG;
    hipError_t err = hipGraphLaunch(plan->exec, static_cast<hipStream_t>(stream));
    return err == hipSuccess ? DPP_OK : (int)err;
}
