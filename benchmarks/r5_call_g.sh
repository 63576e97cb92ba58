#!/bin/bash
# Round-5 call G: flash attention with the DMA address arithmetic hoisted out of the tile loop: parity (op tests incl. the causal text
# tower), then bench_attention old library vs new, interleaved, and the step A/B.
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_text_encoder.py -m gpu -x -q -k "attention or text or clip or encoder" > $O/r5g_tests_att.log 2>&1
tail -3 $O/r5g_tests_att.log
OLD=img2img-turbo_amd/csrc/libi2i_turbo_attold.so; NEW=img2img-turbo_amd/csrc/libi2i_turbo.so
for rep in 1 2 3; do for lib in $OLD $NEW; do echo "== $lib rep $rep"; I2I_LIB=$lib python benchmarks/bench_attention.py; done; done 2>&1 | grep -v amdgpu.ids | tee $O/r5g_bench_attention_ab.log
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
    --output-format csv -d $O/r5g_pmc_att -o att -- python benchmarks/bench_attention.py > $O/r5g_pmc_att.log 2>&1
python tools/pmc_summary.py $(find $O/r5g_pmc_att -name "*counter_collection.csv" | head -1) attention_dma > $O/r5g_pmc_attention_summary.txt 2>&1
cat $O/r5g_pmc_attention_summary.txt
# the wide GEMM's 3x3 gather with the nine taps of a slab back to back (L2 reuse) against tap-major stage order (old library)
SH="unet 320->320@64 gn,unet 960->320@64 gn,unet 640->640@32 gn,unet 1280->640@32 gn,vae down 128@512 s2,vae down 256@256 s2,vae down 512@128 s2"
for rep in 1 2; do for lib in $OLD $NEW; do echo "== $lib rep $rep"; python benchmarks/bench_ops.py --lib $lib --nogn --tiles 51,53 --iters 5 --only "$SH" --out $O/r5g_ops.json; done; done 2>&1 | grep -v "amdgpu.ids\|n/a" | tee $O/r5g_gather_order_ab.log
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/r5g_fetch -o fetch -- python benchmarks/bench_ops.py --nogn --tiles 51 --iters 3 --only "unet 960->320@64 gn" --out $O/r5g_ops.json > /dev/null 2>&1
python tools/pmc_summary.py $(find $O/r5g_fetch -name "*counter_collection.csv" | head -1) gemm_w32_kernel 2>&1 | tee $O/r5g_pmc_gather_fetch.txt
I2I_LIB=$OLD python bench.py --no-cpu-baseline --no-f32 --no-latency > $O/r5g_bench_bs8_oldlib.json 2> $O/r5g_bench_bs8.err
python bench.py --no-cpu-baseline --no-f32 --per-op $O/r5g_per_op_bs8.txt > $O/r5g_bench_bs8.json 2>> $O/r5g_bench_bs8.err
python - <<'PY'
import json
r=json.load(open('gpurun_out/r5g_bench_bs8.json'))
print({k:r.get(k) for k in ('value','ms_per_step','latency_bs1_ms_p50')}); print('attention_dma', r['kernel_breakdown_ms'].get('attention_dma_kernel')); print('gemm_w32', r['kernel_breakdown_ms'].get('gemm_w32_kernel'))
o=json.load(open('gpurun_out/r5g_bench_bs8_oldlib.json')); print('old library:', {k:o.get(k) for k in ('value','ms_per_step')}, o['kernel_breakdown_ms'].get('attention_dma_kernel'), o['kernel_breakdown_ms'].get('gemm_w32_kernel'))
PY
timeout 300 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q -k "cfg2 or tiny_pix2pix or odd" > $O/r5g_tests_e2e.log 2>&1
tail -3 $O/r5g_tests_e2e.log
