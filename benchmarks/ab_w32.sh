#!/bin/bash
# A/B of the wide-tile conv (tiles 41, 42) against the halo conv (tile 0 = auto) on the VAE shapes; one GPU call.
O=gpurun_out; mkdir -p $O
python -c "import diffusers, peft; print('diffusers', diffusers.__version__, 'peft', peft.__version__)" > $O/probe_diffusers.txt 2>&1
timeout 600 python benchmarks/check_w32_gpu.py > $O/w32_parity.log 2>&1; echo "parity rc $?" >> $O/w32_parity.log
S="vae 128->128@512,vae 256->256@256,vae 512->512@128,vae 512->512@64,vae 256->128@512"
timeout 900 python benchmarks/bench_ops.py --only "$S" --tiles 0,41,42 --iters 7 --out $O/w32_ab_gn.json > $O/w32_ab_gn.log 2>&1
timeout 900 python benchmarks/bench_ops.py --only "$S" --nogn --tiles 0,41,42 --iters 7 --out $O/w32_ab_nogn.json > $O/w32_ab_nogn.log 2>&1
tail -5 $O/probe_diffusers.txt; tail -3 $O/w32_parity.log; cat $O/w32_ab_gn.log; cat $O/w32_ab_nogn.log
