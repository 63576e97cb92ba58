#!/bin/bash
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
S="vae 128->128@512 gn,vae 512->512@128 gn,vae 256->256@256 gn"
for L in 0 16 64 1000; do
  echo "== I2I_W32_LGROUPS=$L" >> $O/r3e_lgroups.log
  I2I_W32_LGROUPS=$L timeout 300 python benchmarks/bench_ops.py --only "$S" --tiles 0 --iters 7 --out $O/r3e_tmp.json >> $O/r3e_lgroups.log 2>&1
done
echo "== nogn" >> $O/r3e_lgroups.log
timeout 300 python benchmarks/bench_ops.py --only "$S" --nogn --tiles 0,13 --iters 7 --out $O/r3e_tmp.json >> $O/r3e_lgroups.log 2>&1
timeout 700 bash benchmarks/pmc_conv.sh $O/r3e_pmc "vae 128->128@512 gn,vae 512->512@128 gn" > /dev/null 2>&1
python tools/pmc_summary.py $(find $O/r3e_pmc/sq1 -name "*counter_collection.csv" | head -1) $(find $O/r3e_pmc/sq2 -name "*counter_collection.csv" | head -1) conv3x3_w32_kernel > $O/r3e_pmc_conv3x3_w32_summary.txt 2>&1
grep -v amdgpu.ids $O/r3e_lgroups.log; cat $O/r3e_pmc_conv3x3_w32_summary.txt
