// Plan files: a planned forward (img2img_turbo_amd.plan.ForwardPlan: a flat i2i_op program + the buffers its pointers refer to) saved by
// the Python planner (img2img_turbo_amd/plan_file.py) and run from C / C++ without Python:
//
//     void* plan;  i2i_plan_load("pix2pix_bs8_512.i2iplan", &plan);          // allocates + uploads every buffer, patches the op pointers
//     i2i_plan_write(plan, "x", host_images, bytes);  ...  "ctx", "eps"       // or i2i_plan_io() + your own device copies
//     i2i_plan_run(plan, stream);                                             // = i2i_run() over the program (i2i_plan_ops() + i2i_graph_create() for a hipGraph)
//     i2i_plan_read(plan, "out", host_out, bytes);  i2i_plan_destroy(plan);
//
// It is the whole-forward entry of the C ABI for callers that are not Python: planning (route queries, tile choices, buffer recycling,
// LoRA merge, weight packing) stays in plan.py / packer.py and happens once, at export.
//
// File layout (little endian, 8-byte aligned sections):
//   header   : "I2IPLAN1", u32 abi, u32 sizeof(i2i_op), u32 n_ops, u32 n_bufs, u32 n_relocs, u32 n_io
//   bufs     : n_bufs  x { u64 bytes; u32 kind (0 = scratch, zero-filled; 1 = contents follow in the data section); u32 pad }
//   io       : n_io    x { char name[24]; u32 buf; u32 pad; u64 offset; u64 bytes }
//   relocs   : n_relocs x { u32 op; u32 field_offset (bytes inside i2i_op); u32 buf; u32 pad; u64 offset }   -> *(void**)(op + field_offset) = base[buf] + offset
//   ops      : n_ops x sizeof(i2i_op) bytes (pointer fields are meaningless until patched; non-pointer fields verbatim)
//   data     : the contents of every kind-1 buffer, in buffer order
//
// What the loader checks (a plan file is an EXECUTABLE artefact -- it names kernels, shapes and strides -- so it deserves the trust of the
// library itself; the checks below keep a damaged or stale file from becoming a wild write, they do not make a hostile one safe):
// table sizes, every io / relocation record inside its buffer (overflow-safe), a relocation may only patch a field that IS a pointer
// of that op's parameter struct (ptr_fields below), and every pointer field must be null in the file -- relocations are the only
// source of addresses, an exporter-process address can never be used.  Allocation failures and exceptions are reported as errors.
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <exception>
#include <memory>
#include <string>
#include <vector>

#include "launch.h"

namespace i2i {      // runtime_hip.hip (hipMalloc / hipMemcpy) or tests/emu/runtime_emu.cpp (malloc / memcpy)
void* rt_alloc(size_t bytes);
void rt_free(void* p);
int rt_upload(void* dst, const void* src, size_t bytes);
int rt_download(void* dst, const void* src, size_t bytes);
int rt_zero(void* dst, size_t bytes);
int rt_sync();
}  // namespace i2i

namespace {
struct BufRec { uint64_t bytes; uint32_t kind, pad; };
struct IoRec { char name[24]; uint32_t buf, pad; uint64_t offset, bytes; };
struct RelocRec { uint32_t op, field_offset, buf, pad; uint64_t offset; };
struct Header { char magic[8]; uint32_t abi, sizeof_op, n_ops, n_bufs, n_relocs, n_io; };

struct Plan {
    std::vector<void*> base;
    std::vector<BufRec> bufs;
    std::vector<IoRec> io;
    std::vector<i2i_op> ops;
    ~Plan() { for (void* p : base) if (p) i2i::rt_free(p); }
};

bool read_exact(FILE* f, void* dst, size_t n) { return n == 0 || fread(dst, 1, n, f) == n; }

// byte offsets (inside i2i_op) of the pointer-typed members of an opcode's parameter struct: the only places a relocation may write
#define PF(member, field) (offsetof(i2i_op, u) + offsetof(decltype(i2i_op::u), member) + offsetof(decltype(decltype(i2i_op::u)::member), field))
const std::vector<size_t>& ptr_fields(int opcode) {
    static const std::vector<size_t> none;
    static const std::vector<size_t> igemm = {PF(igemm, a0), PF(igemm, a1), PF(igemm, b), PF(igemm, gn_ss), PF(igemm, bias), PF(igemm, res), PF(igemm, c),
                                              PF(igemm, ws), PF(igemm, gn_part), PF(igemm, k2_a), PF(igemm, k2_b), PF(igemm, ln_cs), PF(igemm, c2)};
    static const std::vector<size_t> gn_stats = {PF(gn_stats, x0), PF(gn_stats, x1), PF(gn_stats, gamma), PF(gn_stats, beta), PF(gn_stats, partial),
                                                 PF(gn_stats, ss), PF(gn_stats, counters)};
    static const std::vector<size_t> gn_apply = {PF(gn_apply, x), PF(gn_apply, y), PF(gn_apply, ss), PF(gn_apply, x1)};
    static const std::vector<size_t> layernorm = {PF(layernorm, x), PF(layernorm, y), PF(layernorm, gamma), PF(layernorm, beta)};
    static const std::vector<size_t> softmax = {PF(softmax, s), PF(softmax, p)};
    static const std::vector<size_t> attention = {PF(attention, q), PF(attention, k), PF(attention, vt), PF(attention, o), PF(attention, ws)};
    static const std::vector<size_t> to_nhwc = {PF(to_nhwc, x), PF(to_nhwc, y)};
    static const std::vector<size_t> to_nchw = {PF(to_nchw, x), PF(to_nchw, y)};
    static const std::vector<size_t> posterior = {PF(posterior, moments), PF(posterior, eps), PF(posterior, noise), PF(posterior, u), PF(posterior, r_dev), PF(posterior, u_f32)};
    static const std::vector<size_t> ddpm = {PF(ddpm, u), PF(ddpm, e), PF(ddpm, y), PF(ddpm, wpq), PF(ddpm, bpq)};
    static const std::vector<size_t> embed = {PF(embed, ids), PF(embed, tok), PF(embed, pos), PF(embed, y)};
    static const std::vector<size_t> lora = {PF(lora_merge, dst), PF(lora_merge, w0), PF(lora_merge, a), PF(lora_merge, b), PF(lora_merge, rg),
                                             PF(lora_merge, kscale), PF(lora_merge, kshift), PF(lora_merge, bias0), PF(lora_merge, colsum), PF(lora_merge, bias_out)};
    static const std::vector<size_t> resize = {PF(resize_u8, src), PF(resize_u8, dst), PF(resize_u8, bounds), PF(resize_u8, coeffs)};
    switch (opcode) {
        case I2I_OP_IGEMM: return igemm;
        case I2I_OP_GN_STATS: return gn_stats;
        case I2I_OP_GN_APPLY: return gn_apply;
        case I2I_OP_LAYERNORM: return layernorm;
        case I2I_OP_SOFTMAX: return softmax;
        case I2I_OP_ATTENTION: return attention;
        case I2I_OP_NCHW_TO_NHWC: return to_nhwc;
        case I2I_OP_NHWC_TO_NCHW: return to_nchw;
        case I2I_OP_POSTERIOR: return posterior;
        case I2I_OP_DDPM_POSTQUANT: return ddpm;
        case I2I_OP_EMBED: return embed;
        case I2I_OP_LORA_MERGE: return lora;
        case I2I_OP_RESIZE_U8: return resize;
        default: return none;
    }
}
#undef PF

const IoRec* find_io(const Plan* pl, const char* name) {
    for (const IoRec& r : pl->io)
        if (strncmp(r.name, name, sizeof(r.name)) == 0) return &r;
    return nullptr;
}
}  // namespace

static int plan_load_impl(const char* path, void** plan_out);
extern "C" int i2i_plan_load(const char* path, void** plan_out) {
    if (!path || !plan_out) return i2i::fail(I2I_ERR_BAD_ARG, "plan_load: null argument");
    *plan_out = nullptr;
    try {
        return plan_load_impl(path, plan_out);
    } catch (const std::exception& e) {      // std::bad_alloc of the tables / the staging buffer: no exception crosses the C ABI
        return i2i::fail(I2I_ERR_RUNTIME, "plan_load(%s): %s", path, e.what());
    } catch (...) {
        return i2i::fail(I2I_ERR_RUNTIME, "plan_load(%s): unknown exception", path);
    }
}
static int plan_load_impl(const char* path, void** plan_out) {
    FILE* f = fopen(path, "rb");
    if (!f) return i2i::fail(I2I_ERR_BAD_ARG, "plan_load: cannot open %s", path);
    struct Closer { FILE* f; ~Closer() { if (f) fclose(f); } } closer{f};      // (also on the exception path)
    std::unique_ptr<Plan> pl(new Plan());
    auto bail = [&](int code, const char* what) { return i2i::fail(code, "plan_load(%s): %s", path, what); };
    Header h;
    if (!read_exact(f, &h, sizeof(h)) || memcmp(h.magic, "I2IPLAN1", 8) != 0) return bail(I2I_ERR_BAD_ARG, "not a plan file");
    if (h.abi != (uint32_t)I2I_ABI_VERSION || h.sizeof_op != (uint32_t)sizeof(i2i_op))
        return bail(I2I_ERR_BAD_ARG, "written for another ABI version / i2i_op layout: export it again with this library");
    if (h.n_ops > (1u << 20) || h.n_bufs > (1u << 20) || h.n_relocs > (1u << 24) || h.n_io > 64) return bail(I2I_ERR_BAD_ARG, "implausible header");
    pl->bufs.resize(h.n_bufs);
    pl->io.resize(h.n_io);
    std::vector<RelocRec> rel(h.n_relocs);
    pl->ops.resize(h.n_ops);
    if (!read_exact(f, pl->bufs.data(), h.n_bufs * sizeof(BufRec)) || !read_exact(f, pl->io.data(), h.n_io * sizeof(IoRec)) ||
        !read_exact(f, rel.data(), h.n_relocs * sizeof(RelocRec)) || !read_exact(f, pl->ops.data(), (size_t)h.n_ops * sizeof(i2i_op)))
        return bail(I2I_ERR_BAD_ARG, "truncated tables");
    for (const IoRec& r : pl->io) {
        if (r.buf >= h.n_bufs || memchr(r.name, 0, sizeof(r.name)) == nullptr) return bail(I2I_ERR_BAD_ARG, "bad io record");
        const uint64_t cap = pl->bufs[r.buf].bytes;
        if (r.bytes > cap || r.offset > cap - r.bytes) return bail(I2I_ERR_BAD_ARG, "io record outside its buffer");      // (no u64 wrap)
    }
    // every pointer field of every op is null in a well-formed file (the exporter blanks them): relocations are the only source of addresses
    for (uint32_t i = 0; i < h.n_ops; ++i) {
        const std::vector<size_t>& pf = ptr_fields(pl->ops[i].opcode);
        if (pf.empty() && pl->ops[i].opcode != I2I_OP_NOP) return bail(I2I_ERR_BAD_ARG, "unknown opcode in the program");
        for (size_t off : pf) {
            void* v;
            memcpy(&v, (const char*)&pl->ops[i] + off, sizeof(void*));
            if (v) return bail(I2I_ERR_BAD_ARG, "an op carries a raw address (pointer fields must be null in the file)");
        }
    }
    for (const RelocRec& r : rel) {
        if (r.op >= h.n_ops || r.buf >= h.n_bufs || r.offset > pl->bufs[r.buf].bytes) return bail(I2I_ERR_BAD_ARG, "bad relocation record");
        const std::vector<size_t>& pf = ptr_fields(pl->ops[r.op].opcode);
        bool is_ptr = false;
        for (size_t off : pf) is_ptr |= off == (size_t)r.field_offset;
        if (!is_ptr) return bail(I2I_ERR_BAD_ARG, "a relocation targets a field that is not a pointer of its op");
    }
    // buffers: allocate all, zero the scratch ones (ticket counters and the like must start at zero), stream the data ones in
    pl->base.assign(h.n_bufs, nullptr);
    std::vector<char> stage(64u << 20);
    for (uint32_t b = 0; b < h.n_bufs; ++b) {
        const size_t n = (size_t)pl->bufs[b].bytes;
        pl->base[b] = i2i::rt_alloc(n ? n : 16);
        if (!pl->base[b]) return bail(I2I_ERR_RUNTIME, "out of device memory");
        if (pl->bufs[b].kind == 0) {
            if (i2i::rt_zero(pl->base[b], n) != I2I_OK) return bail(I2I_ERR_RUNTIME, "memset failed");
        }
    }
    for (uint32_t b = 0; b < h.n_bufs; ++b) {
        if (pl->bufs[b].kind != 1) continue;
        size_t left = (size_t)pl->bufs[b].bytes, off = 0;
        while (left) {
            const size_t n = left < stage.size() ? left : stage.size();
            if (!read_exact(f, stage.data(), n)) return bail(I2I_ERR_BAD_ARG, "truncated data section");
            if (i2i::rt_upload((char*)pl->base[b] + off, stage.data(), n) != I2I_OK) return bail(I2I_ERR_RUNTIME, "upload failed");
            off += n;
            left -= n;
        }
    }
    for (const RelocRec& r : rel) {
        void* v = (char*)pl->base[r.buf] + r.offset;
        memcpy((char*)&pl->ops[r.op] + r.field_offset, &v, sizeof(void*));
    }
    // the zero-fills above are asynchronous with respect to a NON-BLOCKING stream the caller may run the plan on right away
    if (i2i::rt_sync() != I2I_OK) return bail(I2I_ERR_RUNTIME, "device synchronisation failed");
    *plan_out = pl.release();
    return I2I_OK;
}

extern "C" int i2i_plan_io(void* plan, const char* name, void** dev_ptr, size_t* bytes) {
    if (!plan || !name) return i2i::fail(I2I_ERR_BAD_ARG, "plan_io: null argument");
    const Plan* pl = (const Plan*)plan;
    const IoRec* r = find_io(pl, name);
    if (!r) return i2i::fail(I2I_ERR_BAD_ARG, "plan_io: the plan has no buffer named '%s'", name);
    if (dev_ptr) *dev_ptr = (char*)pl->base[r->buf] + r->offset;
    if (bytes) *bytes = (size_t)r->bytes;
    return I2I_OK;
}

extern "C" int i2i_plan_write(void* plan, const char* name, const void* host_src, size_t bytes) {
    void* d;
    size_t n;
    const int rc = i2i_plan_io(plan, name, &d, &n);
    if (rc != I2I_OK) return rc;
    if (!host_src || bytes != n) return i2i::fail(I2I_ERR_BAD_ARG, "plan_write('%s'): %zu bytes given, the buffer holds %zu", name, bytes, n);
    return i2i::rt_upload(d, host_src, n);
}

extern "C" int i2i_plan_read(void* plan, const char* name, void* host_dst, size_t bytes) {
    void* d;
    size_t n;
    const int rc = i2i_plan_io(plan, name, &d, &n);
    if (rc != I2I_OK) return rc;
    if (!host_dst || bytes != n) return i2i::fail(I2I_ERR_BAD_ARG, "plan_read('%s'): %zu bytes asked, the buffer holds %zu", name, bytes, n);
    return i2i::rt_download(host_dst, d, n);
}

extern "C" int i2i_plan_ops(void* plan, const i2i_op** ops, int* n_ops) {
    if (!plan || !ops || !n_ops) return i2i::fail(I2I_ERR_BAD_ARG, "plan_ops: null argument");
    const Plan* pl = (const Plan*)plan;
    *ops = pl->ops.data();
    *n_ops = (int)pl->ops.size();
    return I2I_OK;
}

extern "C" int i2i_plan_run(void* plan, void* stream) {
    if (!plan) return i2i::fail(I2I_ERR_BAD_ARG, "plan_run: null plan");
    const Plan* pl = (const Plan*)plan;
    return i2i_run(pl->ops.data(), (int)pl->ops.size(), stream);
}

extern "C" int i2i_plan_destroy(void* plan) {
    delete (Plan*)plan;
    return I2I_OK;
}
