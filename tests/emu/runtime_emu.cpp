// Emulator half of the C ABI (TEST INFRASTRUCTURE ONLY): timing and graph entry points degrade to plain
// program execution so the Python plan/graph code paths can be exercised on the CPU.
#include <vector>
#include <cstring>
#include "launch.h"

extern "C" const char* i2i_backend(void) { return "emu"; }

extern "C" int i2i_run_timed(const i2i_op* ops, int n_ops, void* stream, float* ms) {
    for (int i = 0; i < n_ops; ++i) ms[i] = 0.f;
    return i2i_run(ops, n_ops, stream);
}
namespace { struct G { std::vector<i2i_op> ops; }; }
extern "C" int i2i_graph_create(const i2i_op* ops, int n_ops, void** out) {
    G* g = new G(); g->ops.assign(ops, ops + n_ops); *out = g; return 0;
}
extern "C" int i2i_graph_launch(void* g, void* stream) { G* gg = (G*)g; return i2i_run(gg->ops.data(), (int)gg->ops.size(), stream); }
extern "C" int i2i_graph_destroy(void* g) { delete (G*)g; return 0; }

// "device" memory of plan files (csrc/plan_file.hip) is host memory here
#include <cstdlib>
namespace i2i {
void* rt_alloc(size_t bytes) { return aligned_alloc(256, (bytes + 255) / 256 * 256); }
void rt_free(void* p) { free(p); }
int rt_upload(void* dst, const void* src, size_t bytes) { memcpy(dst, src, bytes); return 0; }
int rt_download(void* dst, const void* src, size_t bytes) { memcpy(dst, src, bytes); return 0; }
int rt_zero(void* dst, size_t bytes) { memset(dst, 0, bytes); return 0; }
int rt_sync() { return 0; }
}  // namespace i2i
