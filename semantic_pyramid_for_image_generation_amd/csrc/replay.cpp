// Replay of a captured loop body (train_model.py:300-349) with plain stream launches.
//
// The loop body is ~570 kernels per iteration, most of them 5-60 us long; issued from Python the host cannot keep the GPU
// fed.  Capturing the body into a HIP graph fixes the recording, but hipGraphLaunch of this runtime (ROCm 7.2) executes the
// branches of the captured DAG one after the other and adds a gap per node: measured 63.6 ms per 5-iteration cycle against
// 56.3 ms for the same kernels issued eagerly on four streams.  This file therefore walks the captured hipGraph_t ONCE
// (kernel / memset / 1-D memcpy nodes with their frozen arguments, and the dependency edges), lays the DAG out on a few
// HIP streams -- a node continues the stream of a predecessor whose last node it follows, a fork takes another stream,
// cross-stream edges become event record / wait pairs, redundant waits are dropped -- and `sempyr_replay_launch` then
// re-issues the whole body with back-to-back hipLaunchKernel calls from native code: a few microseconds per kernel, with
// the branch concurrency of the eager multi-stream schedule.
//
// Nothing is allocated at launch time; streams and events are created when the plan is built.  The graph must stay alive
// (its nodes own the argument copies the launches read).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <mutex>
#include <vector>
#include "../../include/sempyr.h"

void sempyr_set_error(const char* fmt, ...);

namespace {

enum Kind { K_KERNEL, K_MEMSET, K_MEMCPY, K_EMPTY };

struct Step {
    Kind kind;
    hipKernelNodeParams kp;
    hipMemsetParams ms;
    void* cp_dst; const void* cp_src; size_t cp_bytes;
    int stream;                       // logical stream (0 = the caller's)
    int record_event;                 // event recorded right after this step (-1: none)
    std::vector<int> wait_events;     // events this step's stream waits for before the step
    bool use_module_launch;           // func is a hipFunction_t (not seen so far; decided at the first launch)
};

}  // namespace

struct sempyr_replay {
    std::vector<Step> steps;          // in issue order
    std::vector<hipStream_t> streams; // [0] unused (the caller's stream), others from the device's shared set
    std::vector<hipEvent_t> events;
    hipEvent_t start = nullptr;
    std::vector<int> first_on_stream; // step index of the first step of every logical stream (-1: stream unused)
    std::vector<int> tail_event;      // event recorded behind the last step of streams 1.. (joined into stream 0 at the end)
    int n_kernels = 0, n_memsets = 0, n_memcpys = 0, n_empty = 0, n_waits = 0;
};

#define HIP_TRY(call, what) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    sempyr_set_error("replay: %s: %s", what, hipGetErrorString(e_)); return (int)e_; } } while (0)

// The extra streams are shared by every plan of a device (plans whose replays overlap are ordered by the events they record,
// so sharing is always correct): HIP maps streams onto a handful of hardware queues, and a private set per plan -- 16 plans
// x 3 streams in loop.py -- makes two streams that are busy at the same time share a queue far more often.
namespace {
constexpr int REPLAY_MAX_DEVICES = 64;
std::mutex g_shared_lock;
std::vector<hipStream_t> g_shared_streams[REPLAY_MAX_DEVICES];

int shared_stream(int idx, hipStream_t* out)
{
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= REPLAY_MAX_DEVICES) { sempyr_set_error("replay_create: hipGetDevice"); return SEMPYR_ERR_INVALID; }
    std::lock_guard<std::mutex> lock(g_shared_lock);
    auto& v = g_shared_streams[dev];
    while ((int)v.size() <= idx) {
        hipStream_t st = nullptr;
        e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        if (e != hipSuccess) { sempyr_set_error("replay_create: hipStreamCreate: %s", hipGetErrorString(e)); return (int)e; }
        v.push_back(st);
    }
    *out = v[idx];
    return 0;
}
}  // namespace

extern "C" int sempyr_replay_destroy(sempyr_replay* r)
{
    if (!r) return 0;
    for (hipEvent_t e : r->events) if (e) (void)hipEventDestroy(e);
    if (r->start) (void)hipEventDestroy(r->start);
    delete r;
    return 0;
}

extern "C" int sempyr_replay_create(void* hip_graph, int max_streams, sempyr_replay** out, int32_t* stats)
{
    return sempyr_replay_create_on(hip_graph, max_streams, nullptr, 0, out, stats);
}

extern "C" int sempyr_replay_create_on(void* hip_graph, int max_streams, void* const* side_streams, int n_side_streams,
                                       sempyr_replay** out, int32_t* stats)
{
    if (!hip_graph || !out || n_side_streams < 0 || (n_side_streams > 0 && !side_streams)) {
        sempyr_set_error("replay_create: null argument"); return SEMPYR_ERR_INVALID;
    }
    if (max_streams < 1) max_streams = 1;
    if (max_streams > 16) max_streams = 16;
    hipGraph_t graph = (hipGraph_t)hip_graph;
    size_t n = 0, ne = 0;
    HIP_TRY(hipGraphGetNodes(graph, nullptr, &n), "hipGraphGetNodes");
    if (n == 0) { sempyr_set_error("replay_create: the graph is empty"); return SEMPYR_ERR_INVALID; }
    std::vector<hipGraphNode_t> nodes(n);
    HIP_TRY(hipGraphGetNodes(graph, nodes.data(), &n), "hipGraphGetNodes");
    HIP_TRY(hipGraphGetEdges(graph, nullptr, nullptr, &ne), "hipGraphGetEdges");
    std::vector<hipGraphNode_t> from(ne), to(ne);
    if (ne) HIP_TRY(hipGraphGetEdges(graph, from.data(), to.data(), &ne), "hipGraphGetEdges");
    std::map<hipGraphNode_t, int> index;
    for (size_t i = 0; i < n; ++i) index[nodes[i]] = (int)i;
    std::vector<std::vector<int>> deps(n), succ(n);
    for (size_t e = 0; e < ne; ++e) {
        auto a = index.find(from[e]), b = index.find(to[e]);
        if (a == index.end() || b == index.end()) { sempyr_set_error("replay_create: edge to an unknown node"); return SEMPYR_ERR_INVALID; }
        deps[b->second].push_back(a->second); succ[a->second].push_back(b->second);
    }
    // topological order, ties broken by creation order (= the order the body issued the launches in)
    std::vector<int> indeg(n), order; order.reserve(n);
    for (size_t i = 0; i < n; ++i) indeg[i] = (int)deps[i].size();
    {
        std::vector<int> ready;
        for (size_t i = 0; i < n; ++i) if (!indeg[i]) ready.push_back((int)i);
        std::make_heap(ready.begin(), ready.end(), std::greater<int>());
        while (!ready.empty()) {
            std::pop_heap(ready.begin(), ready.end(), std::greater<int>());
            const int u = ready.back(); ready.pop_back();
            order.push_back(u);
            for (int v : succ[u]) if (--indeg[v] == 0) { ready.push_back(v); std::push_heap(ready.begin(), ready.end(), std::greater<int>()); }
        }
        if (order.size() != n) { sempyr_set_error("replay_create: the graph has a cycle"); return SEMPYR_ERR_INVALID; }
    }
    std::vector<int> pos(n);                         // node index -> position in `order`
    for (size_t i = 0; i < n; ++i) pos[order[i]] = (int)i;

    if (const char* dump = getenv("SEMPYR_REPLAY_DUMP")) {          // debugging aid: the captured DAG as text
        if (FILE* f = fopen(dump, "a")) {
            fprintf(f, "# graph with %zu nodes, %zu edges\n", n, ne);
            for (size_t i = 0; i < n; ++i) {
                hipGraphNodeType t2 = hipGraphNodeTypeEmpty; hipKernelNodeParams kp2;
                (void)hipGraphNodeGetType(nodes[order[i]], &t2);
                const char* nm = "-";
                if (t2 == hipGraphNodeTypeKernel && hipGraphKernelNodeGetParams(nodes[order[i]], &kp2) == hipSuccess) {
                    nm = hipKernelNameRefByPtr(kp2.func, nullptr); if (!nm) nm = "?";
                }
                fprintf(f, "%zu type %d %.90s deps", i, (int)t2, nm);
                for (int d : deps[order[i]]) fprintf(f, " %d", pos[d]);
                fprintf(f, "\n");
            }
            fclose(f);
        }
    }
    sempyr_replay* r = new sempyr_replay();
    r->steps.resize(n);
    for (size_t i = 0; i < n; ++i) {
        Step& s = r->steps[i];
        hipGraphNode_t node = nodes[order[i]];
        hipGraphNodeType type;
        hipError_t e = hipGraphNodeGetType(node, &type);
        if (e != hipSuccess) { sempyr_set_error("replay_create: hipGraphNodeGetType: %s", hipGetErrorString(e)); sempyr_replay_destroy(r); return (int)e; }
        s.stream = -1; s.record_event = -1; s.use_module_launch = false;
        if (type == hipGraphNodeTypeKernel) {
            s.kind = K_KERNEL;
            e = hipGraphKernelNodeGetParams(node, &s.kp);
            if (e != hipSuccess) { sempyr_set_error("replay_create: hipGraphKernelNodeGetParams: %s", hipGetErrorString(e)); sempyr_replay_destroy(r); return (int)e; }
            if (s.kp.extra && !s.kp.kernelParams) {
                sempyr_set_error("replay_create: a kernel node passes its arguments through `extra` (not supported)");
                sempyr_replay_destroy(r); return SEMPYR_ERR_INVALID;
            }
            ++r->n_kernels;
        } else if (type == hipGraphNodeTypeMemset) {
            s.kind = K_MEMSET;
            e = hipGraphMemsetNodeGetParams(node, &s.ms);
            if (e != hipSuccess) { sempyr_set_error("replay_create: hipGraphMemsetNodeGetParams: %s", hipGetErrorString(e)); sempyr_replay_destroy(r); return (int)e; }
            if (s.ms.height > 1) { sempyr_set_error("replay_create: 2-D memset nodes are not supported"); sempyr_replay_destroy(r); return SEMPYR_ERR_INVALID; }
            ++r->n_memsets;
        } else if (type == hipGraphNodeTypeMemcpy) {
            s.kind = K_MEMCPY;
            hipMemcpy3DParms p;
            e = hipGraphMemcpyNodeGetParams(node, &p);
            if (e != hipSuccess || p.srcArray || p.dstArray || p.extent.height > 1 || p.extent.depth > 1 ||
                !(p.kind == hipMemcpyDeviceToDevice || p.kind == hipMemcpyDefault) || !p.srcPtr.ptr || !p.dstPtr.ptr) {
                char near[600] = ""; size_t off = 0;
                for (int d : deps[order[i]]) {                 // name the kernels it follows, to find the copy in the body
                    hipGraphNodeType t2; hipKernelNodeParams kp2;
                    if (hipGraphNodeGetType(nodes[d], &t2) == hipSuccess && t2 == hipGraphNodeTypeKernel &&
                        hipGraphKernelNodeGetParams(nodes[d], &kp2) == hipSuccess) {
                        const char* nm = hipKernelNameRefByPtr(kp2.func, nullptr);
                        off += snprintf(near + off, sizeof(near) - off, " after[%s]", nm ? nm : "?");
                        if (off >= sizeof(near) - 1) break;
                    }
                }
                sempyr_set_error("replay_create: only plain 1-D device-to-device memcpy nodes are supported (node %d of %d: rc %d kind %d "
                                 "w %zu h %zu d %zu)%.400s", (int)i, (int)n, (int)e, (int)p.kind, (size_t)p.extent.width, (size_t)p.extent.height,
                                 (size_t)p.extent.depth, near);
                sempyr_replay_destroy(r); return SEMPYR_ERR_INVALID;
            }
            s.cp_src = (const char*)p.srcPtr.ptr + p.srcPos.x; s.cp_dst = (char*)p.dstPtr.ptr + p.dstPos.x; s.cp_bytes = p.extent.width;
            // the copy is re-issued as device-to-device: a node captured with hipMemcpyDefault must really be one
            for (const void* ptr : {s.cp_src, (const void*)s.cp_dst}) {
                hipPointerAttribute_t at;
                hipError_t pe = hipPointerGetAttributes(&at, ptr);
                if (pe != hipSuccess || at.type != hipMemoryTypeDevice) {
                    (void)hipGetLastError();
                    sempyr_set_error("replay_create: memcpy node %d touches memory that is not device memory (only device-to-device "
                                     "copies are replayed)", (int)i);
                    sempyr_replay_destroy(r); return SEMPYR_ERR_INVALID;
                }
            }
            ++r->n_memcpys;
        } else if (type == hipGraphNodeTypeEmpty) {
            s.kind = K_EMPTY; ++r->n_empty;
        } else {
            sempyr_set_error("replay_create: unsupported graph node type %d", (int)type);
            sempyr_replay_destroy(r); return SEMPYR_ERR_INVALID;
        }
    }

    // lay the DAG out on streams
    std::vector<int> tail;                            // per logical stream: position of its last step
    std::vector<std::vector<int>> waited;             // waited[s][t] = last position on stream t that stream s has waited for
    auto new_stream = [&]() { tail.push_back(-1); for (auto& w : waited) w.push_back(-1); waited.push_back(std::vector<int>(tail.size(), -1)); return (int)tail.size() - 1; };
    new_stream();
    for (size_t i = 0; i < n; ++i) {
        Step& s = r->steps[i];
        std::vector<int> dpos;
        for (int d : deps[order[i]]) dpos.push_back(pos[d]);
        int chosen = -1;
        for (int st = 0; st < (int)tail.size() && chosen < 0; ++st)
            for (int d : dpos) if (tail[st] == d) { chosen = st; break; }
        if (chosen < 0) {
            if (dpos.empty() && tail[0] < 0) chosen = 0;
            else {
                // a fork (or a root): an unused stream, a new one, or the one whose last step is oldest (never the caller's:
                // stream 0 carries the main chain)
                for (int st = 1; st < (int)tail.size() && chosen < 0; ++st) if (tail[st] < 0) chosen = st;
                if (chosen < 0 && (int)tail.size() < max_streams) chosen = new_stream();
                if (chosen < 0) {
                    chosen = tail.size() > 1 ? 1 : 0;
                    for (int st = 2; st < (int)tail.size(); ++st) if (tail[st] < tail[chosen]) chosen = st;
                }
            }
        }
        s.stream = chosen;
        for (int d : dpos) {
            const int ds = r->steps[d].stream;
            if (ds == chosen || waited[chosen][ds] >= d) continue;
            if (r->steps[d].record_event < 0) { r->steps[d].record_event = (int)r->events.size(); r->events.push_back(nullptr); }
            s.wait_events.push_back(r->steps[d].record_event);
            waited[chosen][ds] = d;
            ++r->n_waits;
        }
        tail[chosen] = (int)i;
    }
    r->first_on_stream.assign(tail.size(), -1);
    for (size_t i = 0; i < n; ++i) if (r->first_on_stream[r->steps[i].stream] < 0) r->first_on_stream[r->steps[i].stream] = (int)i;
    r->tail_event.assign(tail.size(), -1);
    for (int st = 1; st < (int)tail.size(); ++st) {
        if (tail[st] < 0) continue;
        Step& t = r->steps[tail[st]];
        if (t.record_event < 0) { t.record_event = (int)r->events.size(); r->events.push_back(nullptr); }
        r->tail_event[st] = t.record_event;
    }
    r->streams.assign(tail.size(), nullptr);
    for (size_t st = 1; st < tail.size(); ++st) {
        if ((int)st - 1 < n_side_streams) { r->streams[st] = (hipStream_t)side_streams[st - 1]; continue; }
        const int rc = shared_stream((int)st - 1 - n_side_streams, &r->streams[st]);
        if (rc != 0) { sempyr_replay_destroy(r); return rc; }
    }
    for (hipEvent_t& ev : r->events) {
        hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e != hipSuccess) { sempyr_set_error("replay_create: hipEventCreate: %s", hipGetErrorString(e)); sempyr_replay_destroy(r); return (int)e; }
    }
    {
        hipError_t e = hipEventCreateWithFlags(&r->start, hipEventDisableTiming);
        if (e != hipSuccess) { sempyr_set_error("replay_create: hipEventCreate: %s", hipGetErrorString(e)); sempyr_replay_destroy(r); return (int)e; }
    }
    if (stats) {
        stats[0] = r->n_kernels; stats[1] = r->n_memsets; stats[2] = r->n_memcpys; stats[3] = r->n_empty;
        stats[4] = (int)tail.size(); stats[5] = r->n_waits; stats[6] = (int)r->events.size(); stats[7] = (int)ne;
    }
    *out = r;
    return 0;
}

static int replay_issue(sempyr_replay* r, hipStream_t main);

extern "C" int sempyr_replay_launch(sempyr_replay* r, void* stream)
{
    if (!r) { sempyr_set_error("replay_launch: null plan"); return SEMPYR_ERR_INVALID; }
    const int rc = replay_issue(r, (hipStream_t)stream);
    if (rc != 0) {
        // a launch failed midway: whatever was already issued on the owned streams must not race with what the caller queues
        // on its stream next (the join at the end was not reached): drain them before returning the error
        for (size_t st = 1; st < r->streams.size(); ++st) if (r->streams[st]) (void)hipStreamSynchronize(r->streams[st]);
    }
    return rc;
}

static int replay_issue(sempyr_replay* r, hipStream_t main)
{
    const int ns = (int)r->streams.size();
    if (ns > 1) {
        HIP_TRY(hipEventRecord(r->start, main), "hipEventRecord(start)");
        for (int st = 1; st < ns; ++st)
            if (r->first_on_stream[st] >= 0) HIP_TRY(hipStreamWaitEvent(r->streams[st], r->start, 0), "hipStreamWaitEvent(start)");
    }
    for (Step& s : r->steps) {
        hipStream_t st = s.stream == 0 ? main : r->streams[s.stream];
        for (int ev : s.wait_events) HIP_TRY(hipStreamWaitEvent(st, r->events[ev], 0), "hipStreamWaitEvent");
        switch (s.kind) {
        case K_KERNEL: {
            hipError_t e;
            if (!s.use_module_launch) {
                e = hipLaunchKernel(s.kp.func, s.kp.gridDim, s.kp.blockDim, s.kp.kernelParams, s.kp.sharedMemBytes, st);
                if (e == hipErrorInvalidDeviceFunction) { (void)hipGetLastError(); s.use_module_launch = true; }
            }
            if (s.use_module_launch)
                e = hipModuleLaunchKernel((hipFunction_t)s.kp.func, s.kp.gridDim.x, s.kp.gridDim.y, s.kp.gridDim.z, s.kp.blockDim.x,
                                          s.kp.blockDim.y, s.kp.blockDim.z, s.kp.sharedMemBytes, st, s.kp.kernelParams, nullptr);
            if (e != hipSuccess) { sempyr_set_error("replay_launch: kernel launch failed: %s", hipGetErrorString(e)); return (int)e; }
            break;
        }
        case K_MEMSET:
            if (s.ms.elementSize == 4) HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)s.ms.dst, (int)s.ms.value, s.ms.width, st), "hipMemsetD32Async");
            else if (s.ms.elementSize == 2) HIP_TRY(hipMemsetD16Async((hipDeviceptr_t)s.ms.dst, (unsigned short)s.ms.value, s.ms.width, st), "hipMemsetD16Async");
            else HIP_TRY(hipMemsetAsync(s.ms.dst, (int)s.ms.value, s.ms.width, st), "hipMemsetAsync");
            break;
        case K_MEMCPY:
            HIP_TRY(hipMemcpyAsync(s.cp_dst, s.cp_src, s.cp_bytes, hipMemcpyDeviceToDevice, st), "hipMemcpyAsync");
            break;
        case K_EMPTY:
            break;
        }
        if (s.record_event >= 0) HIP_TRY(hipEventRecord(r->events[s.record_event], st), "hipEventRecord");
    }
    for (int st = 1; st < ns; ++st)
        if (r->tail_event[st] >= 0) HIP_TRY(hipStreamWaitEvent(main, r->events[r->tail_event[st]], 0), "hipStreamWaitEvent(join)");
    return 0;
}
