// HIP-runtime half of the C ABI (gfx950 product library only): per-op HIP-event timing and hipGraph
// capture/replay.  The bs=1 forward is ~400 short launches; replaying them as one hipGraph removes the
// host launch cost from the latency path (MI355X_MICROARCH.md "graph-replay-floor").
#include <vector>

#include "launch.h"

extern "C" const char* i2i_backend(void) { return "gfx950"; }

#define I2I_HIP(call)                                                                                   \
    do {                                                                                                \
        hipError_t e__ = (call);                                                                        \
        if (e__ != hipSuccess) return i2i::fail(I2I_ERR_RUNTIME, "%s: %s", #call, hipGetErrorString(e__)); \
    } while (0)

extern "C" int i2i_run_timed(const i2i_op* ops, int n_ops, void* stream, float* ms) {
    if (!ops || !ms) return i2i::fail(I2I_ERR_BAD_ARG, "run_timed: null pointer");
    hipStream_t s = (hipStream_t)stream;
    std::vector<hipEvent_t> ev((size_t)n_ops + 1);
    for (auto& e : ev) I2I_HIP(hipEventCreate(&e));
    int rc = I2I_OK;
    I2I_HIP(hipEventRecord(ev[0], s));
    for (int i = 0; i < n_ops && rc == I2I_OK; ++i) {
        rc = i2i_run(ops + i, 1, stream);
        I2I_HIP(hipEventRecord(ev[i + 1], s));
    }
    I2I_HIP(hipStreamSynchronize(s));
    if (rc == I2I_OK)
        for (int i = 0; i < n_ops; ++i) I2I_HIP(hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]));
    for (auto& e : ev) (void)hipEventDestroy(e);
    return rc;
}

namespace {
struct Graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};
}  // namespace

extern "C" int i2i_graph_create(const i2i_op* ops, int n_ops, void** graph_out) {
    if (!ops || !graph_out) return i2i::fail(I2I_ERR_BAD_ARG, "graph_create: null pointer");
    hipStream_t cap;
    I2I_HIP(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
    Graph* g = new Graph();
    hipError_t e = hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { delete g; (void)hipStreamDestroy(cap); return i2i::fail(I2I_ERR_RUNTIME, "hipStreamBeginCapture: %s", hipGetErrorString(e)); }
    const int rc = i2i_run(ops, n_ops, cap);
    e = hipStreamEndCapture(cap, &g->graph);
    (void)hipStreamDestroy(cap);
    if (rc != I2I_OK) { if (g->graph) (void)hipGraphDestroy(g->graph); delete g; return rc; }
    if (e != hipSuccess) { delete g; return i2i::fail(I2I_ERR_RUNTIME, "hipStreamEndCapture: %s", hipGetErrorString(e)); }
    e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) { (void)hipGraphDestroy(g->graph); delete g; return i2i::fail(I2I_ERR_RUNTIME, "hipGraphInstantiate: %s", hipGetErrorString(e)); }
    *graph_out = g;
    return I2I_OK;
}

extern "C" int i2i_graph_launch(void* graph, void* stream) {
    if (!graph) return i2i::fail(I2I_ERR_BAD_ARG, "graph_launch: null graph");
    I2I_HIP(hipGraphLaunch(((Graph*)graph)->exec, (hipStream_t)stream));
    return I2I_OK;
}

extern "C" int i2i_graph_destroy(void* graph) {
    if (!graph) return I2I_OK;
    Graph* g = (Graph*)graph;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    return I2I_OK;
}

// ---- device memory for plan files (plan_file.hip): synchronous on purpose -- load / write / read are not on the latency path
namespace i2i {
void* rt_alloc(size_t bytes) {
    void* p = nullptr;
    return hipMalloc(&p, bytes) == hipSuccess ? p : nullptr;
}
void rt_free(void* p) { (void)hipFree(p); }
int rt_upload(void* dst, const void* src, size_t bytes) {
    I2I_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return I2I_OK;
}
int rt_download(void* dst, const void* src, size_t bytes) {
    I2I_HIP(hipDeviceSynchronize());      // (the program may still be running on a non-blocking stream)
    I2I_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return I2I_OK;
}
int rt_zero(void* dst, size_t bytes) {
    I2I_HIP(hipMemset(dst, 0, bytes));
    return I2I_OK;
}
int rt_sync() {
    I2I_HIP(hipDeviceSynchronize());
    return I2I_OK;
}
}  // namespace i2i
