#!/bin/bash
# usage: tools/regs.sh <file.hip> [grep-filter]   -- per-kernel register / spill / occupancy report (hipcc remarks)
cd /root/repo/img2img-turbo_amd/csrc && mkdir -p /tmp/k && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Rpass-analysis=kernel-resource-usage -c $1 -o /tmp/k/$1.o 2>&1 | grep -E "error|Function Name|VGPRs:|Spill|Occupancy|ScratchSize" | sed 's/.*remark: *//; s/ \[-Rpass.*//; s/Function Name: _ZN12_GLOBAL__N_1//; s/i2i_[a-z_]*params//' | paste - - - - - - | grep "${2:-.}"
