#!/bin/bash
# usage: scripts/kernel_resources.sh <file.hip> [grep-pattern] [extra hipcc flags...]   - compact VGPR / AGPR / scratch / LDS table per kernel
f=$1; pat=${2:-.}; shift; shift
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -Wno-unused-value "$@" -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/_kr.o 2>&1 \
 | grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|Occupancy|LDS Size" | sed -E 's/.*remark: [^ ]+ +//; s/ \[-Rpass.*//' | paste - - - - - - \
 | sed -E 's/Function Name: //' | grep -E "$pat" | awk '{n=$1; $1=""; printf "%-70s %s\n", substr(n,1,70), $0}'
