#!/bin/bash
# Round-6 call D: tree after the GroupNorm one-op experiment was removed.  New-op tests, A/B of routing the K-sliced small linears
# (split-K launch + reduce launch) to ONE un-sliced wide-GEMM launch (I2I_W32_UNSPLIT_ROWS), bench lines, then the whole GPU suite.
O=gpurun_out; T=r6d; export TMPDIR=/tmp; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "layernorm_folded or flag_boundaries" 2>&1 | tail -6 > $O/${T}_new_op_tests.log
cat $O/${T}_new_op_tests.log
ARMS='I2I_W32_UNSPLIT_ROWS=0 I2I_W32_UNSPLIT_ROWS=64 I2I_W32_UNSPLIT_ROWS=256 I2I_W32_UNSPLIT_ROWS=1024 I2I_W32_UNSPLIT_ROWS=4096'
python benchmarks/ab.py --arms $ARMS --repeats 6 --steps 30 --batch 1 --out $O/${T}_ab_bs1.json 2>&1 | grep -v amdgpu.ids | tee $O/${T}_ab_bs1_unsplit.log
python benchmarks/ab.py --arms I2I_W32_UNSPLIT_ROWS=0 I2I_W32_UNSPLIT_ROWS=512 I2I_W32_UNSPLIT_ROWS=2048 --repeats 6 --steps 10 --batch 8 --out $O/${T}_ab_bs8.json 2>&1 | grep -v amdgpu.ids | tee $O/${T}_ab_bs8_unsplit.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32 --no-modes --per-op $O/${T}_per_op_bs8.txt > $O/${T}_bench_bs8.json 2> $O/${T}_bench.err
I2I_W32_UNSPLIT_ROWS=1024 python bench.py --batch 1 --no-cpu-baseline --no-f32 --no-modes --per-op $O/${T}_per_op_bs1_unsplit1024.txt > $O/${T}_bench_bs1_unsplit1024.json 2>> $O/${T}_bench.err
python bench.py --batch 1 --no-cpu-baseline --no-f32 --no-modes --per-op $O/${T}_per_op_bs1.txt > $O/${T}_bench_bs1.json 2>> $O/${T}_bench.err
tail -3 $O/${T}_bench.err
for f in $O/${T}_bench_bs8.json $O/${T}_bench_bs1.json $O/${T}_bench_bs1_unsplit1024.json; do python - "$f" <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
print(sys.argv[1].split("/")[-1], r["value"], "img/s", r["ms_per_step"], "ms/step", "bs1 p50", r.get("latency_bs1_ms_p50"), "frac", (r.get("roofline") or {}).get("frac"))
print({k: (v["ms"], v["launches"]) for k, v in r["kernel_breakdown_ms"].items()})
print("calib", {k: v for k, v in (r.get("calib") or {}).items() if k in ("mfma_tflops", "hbm_tbytes_per_s", "graph_node_us", "value_normalised")})
PY
done
timeout 1800 python -m pytest tests -x -q -m gpu --durations=8 2>&1 | tail -30 > $O/${T}_gputests.log
cat $O/${T}_gputests.log
