#!/bin/bash
# Round-6 call B: the new ops on hardware (gn_norm one-launch GroupNorm, LayerNorm folded into the wide GEMM incl. to_q|k|v^T), the
# end-to-end gates, then interleaved same-box A/Bs of the two planner switches at batch 8 and batch 1, per-op tables of the new plan.
O=gpurun_out; T=r6b; export TMPDIR=/tmp; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gn_norm or layernorm_folded" 2>&1 | tail -15 > $O/${T}_new_op_tests.log
cat $O/${T}_new_op_tests.log
timeout 1500 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu 2>&1 | tail -25 > $O/${T}_e2e_tests.log
cat $O/${T}_e2e_tests.log
python benchmarks/ab.py --arms "I2I_LN_FOLD=0,I2I_GN_NORM=0" "I2I_LN_FOLD=1,I2I_GN_NORM=0" "I2I_LN_FOLD=0,I2I_GN_NORM=1" "I2I_LN_FOLD=1,I2I_GN_NORM=1" --repeats 6 --steps 10 --batch 8 --out $O/${T}_ab_bs8.json 2>&1 | grep -v amdgpu.ids | tee $O/${T}_ab_bs8.log
python benchmarks/ab.py --arms "I2I_LN_FOLD=0,I2I_GN_NORM=0" "I2I_LN_FOLD=1,I2I_GN_NORM=0" "I2I_LN_FOLD=0,I2I_GN_NORM=1" "I2I_LN_FOLD=1,I2I_GN_NORM=1" --repeats 6 --steps 30 --batch 1 --out $O/${T}_ab_bs1.json 2>&1 | grep -v amdgpu.ids | tee $O/${T}_ab_bs1.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32 --per-op $O/${T}_per_op_bs8.txt > $O/${T}_bench_bs8.json 2> $O/${T}_bench.err
python bench.py --batch 1 --no-cpu-baseline --no-f32 --per-op $O/${T}_per_op_bs1.txt > $O/${T}_bench_bs1.json 2>> $O/${T}_bench.err
tail -5 $O/${T}_bench.err
for f in $O/${T}_bench_bs8.json $O/${T}_bench_bs1.json; do python - "$f" <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
print(sys.argv[1].split("/")[-1], r["value"], "img/s", r["ms_per_step"], "ms/step", "bs1 p50", r.get("latency_bs1_ms_p50"), "frac", (r.get("roofline") or {}).get("frac"))
print({k: (v["ms"], v["launches"]) for k, v in r["kernel_breakdown_ms"].items()})
PY
done
