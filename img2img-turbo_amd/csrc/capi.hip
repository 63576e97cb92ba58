// C-ABI runtime: error strings, program execution (a forward pass = a flat array of i2i_op run in order
// on one stream), per-op timing with HIP events, and hipGraph capture/replay of a program.
#include <string.h>

#include <vector>

#include "i2i_dev.h"
#include "launch.h"

namespace i2i {
char* error_buffer() {
    static thread_local char buf[512];
    return buf;
}
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}
int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(I2I_ERR_LAUNCH, "%s: launch failed: %s", what, hipGetErrorString(e));
    return I2I_OK;
}
}  // namespace i2i

namespace {
int dispatch(const i2i_op& op, void* stream) {
    switch (op.opcode) {
        case I2I_OP_IGEMM: return i2i_igemm(&op.u.igemm, op.dtype, stream);
        case I2I_OP_GN_STATS: return i2i_gn_stats(&op.u.gn_stats, op.dtype, stream);
        case I2I_OP_GN_APPLY: return i2i_gn_apply(&op.u.gn_apply, op.dtype, stream);
        case I2I_OP_LAYERNORM: return i2i_layernorm(&op.u.layernorm, op.dtype, stream);
        case I2I_OP_SOFTMAX: return i2i_softmax(&op.u.softmax, op.dtype, stream);
        case I2I_OP_ATTENTION: return i2i_attention(&op.u.attention, op.dtype, stream);
        case I2I_OP_NCHW_TO_NHWC: return i2i_nchw_to_nhwc(&op.u.to_nhwc, op.dtype, stream);
        case I2I_OP_NHWC_TO_NCHW: return i2i_nhwc_to_nchw(&op.u.to_nchw, op.dtype, stream);
        case I2I_OP_POSTERIOR: return i2i_posterior(&op.u.posterior, op.dtype, stream);
        case I2I_OP_DDPM_POSTQUANT: return i2i_ddpm_postquant(&op.u.ddpm, op.dtype, stream);
        case I2I_OP_EMBED: return i2i_embed(&op.u.embed, op.dtype, stream);
        case I2I_OP_LORA_MERGE: return i2i_lora_merge(&op.u.lora_merge, op.dtype, stream);
        case I2I_OP_RESIZE_U8: return i2i_resize_u8(&op.u.resize_u8, op.dtype, stream);
        case I2I_OP_NOP: return i2i_nop(stream);
        default: return i2i::fail(I2I_ERR_BAD_ARG, "run: unknown opcode %d", op.opcode);
    }
}
}  // namespace

extern "C" int i2i_abi_version(void) { return I2I_ABI_VERSION; }
extern "C" const char* i2i_last_error(void) { return i2i::error_buffer(); }
extern "C" size_t i2i_sizeof_op(void) { return sizeof(i2i_op); }

extern "C" int i2i_run(const i2i_op* ops, int n_ops, void* stream) {
    if (!ops && n_ops > 0) return i2i::fail(I2I_ERR_BAD_ARG, "run: null program");
    for (int i = 0; i < n_ops; ++i) {
        const int rc = dispatch(ops[i], stream);
        if (rc != I2I_OK) {
            char tmp[400];
            strncpy(tmp, i2i::error_buffer(), sizeof(tmp) - 1);
            tmp[sizeof(tmp) - 1] = 0;
            return i2i::fail(rc, "op %d (opcode %d): %s", i, ops[i].opcode, tmp);
        }
    }
    return I2I_OK;
}
