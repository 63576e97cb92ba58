/* A host that is not Python: loads a plan file written by img2img_turbo_amd.plan_file.export_plan and runs the forward.
 *
 *     cc -O2 -I include examples/plan_host.c -o plan_host -L img2img-turbo_amd/csrc -li2i_turbo -Wl,-rpath,img2img-turbo_amd/csrc
 *     ./plan_host pix2pix_bs8_512.i2iplan x.bin ctx.bin eps.bin out.bin [noise.bin]
 *
 * The .bin files are the raw contents of the boundary buffers (include/i2i_turbo.h, "plan files": "x" fp32 NCHW in [-1, 1] -- or uint8
 * NHWC for a u8 plan --, "ctx" the text states in the plan's dtype, "eps" fp32 posterior noise, "out" the images).  What
 * src/inference_paired.py does around `model(c_t, prompt)` (src/inference_paired.py:44-62) -- tokenising, resizing, file formats --
 * stays with the host.  tests/test_e2e_emu.py (_check_plan_file_round_trip) builds this file with gcc against the CPU emulator library
 * and checks its output against the Python replay bit for bit. */
#include <stdio.h>
#include <stdlib.h>

#include "i2i_turbo.h"

static int fail(const char* what) {
    fprintf(stderr, "plan_host: %s: %s\n", what, i2i_last_error());
    return 1;
}

static int feed(void* plan, const char* name, const char* path) {
    void* dev;
    size_t bytes;
    if (i2i_plan_io(plan, name, &dev, &bytes) != I2I_OK) return fail(name);
    void* host = malloc(bytes);
    FILE* f = fopen(path, "rb");
    if (!host || !f || fread(host, 1, bytes, f) != bytes) { fprintf(stderr, "plan_host: %s: cannot read %zu bytes from %s\n", name, bytes, path); return 1; }
    fclose(f);
    const int rc = i2i_plan_write(plan, name, host, bytes);
    free(host);
    return rc == I2I_OK ? 0 : fail(name);
}

int main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage: %s plan x.bin ctx.bin eps.bin out.bin [noise.bin]\n", argv[0]); return 2; }
    if (i2i_abi_version() != I2I_ABI_VERSION) { fprintf(stderr, "plan_host: header / library ABI mismatch\n"); return 2; }
    void* plan = NULL;
    if (i2i_plan_load(argv[1], &plan) != I2I_OK) return fail("load");
    if (feed(plan, "x", argv[2]) || feed(plan, "ctx", argv[3]) || feed(plan, "eps", argv[4])) return 1;
    if (argc > 6 && feed(plan, "noise", argv[6])) return 1;
    if (i2i_plan_run(plan, NULL) != I2I_OK) return fail("run");      /* NULL = the default stream; i2i_plan_ops() + i2i_graph_create() for a hipGraph */
    void* dev;
    size_t bytes;
    if (i2i_plan_io(plan, "out", &dev, &bytes) != I2I_OK) return fail("out");
    void* host = malloc(bytes);
    if (!host || i2i_plan_read(plan, "out", host, bytes) != I2I_OK) return fail("read");
    FILE* f = fopen(argv[5], "wb");
    if (!f || fwrite(host, 1, bytes, f) != bytes) { fprintf(stderr, "plan_host: cannot write %s\n", argv[5]); return 1; }
    fclose(f);
    free(host);
    i2i_plan_destroy(plan);
    printf("plan_host: %s on %s, %zu output bytes\n", argv[1], i2i_backend(), bytes);
    return 0;
}
