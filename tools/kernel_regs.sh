#!/bin/bash
# kernel_regs.sh <unit.hip> [extra hipcc flags]: VGPRs, scratch, occupancy and LDS of every kernel of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage)
cd "$(dirname "$0")/../mulls_amd/csrc" || exit 1
u="$1"; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fopenmp -Wno-unused-function -c "$u" -o /tmp/kernel_regs.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
	awk '/error/ {print} /Function Name:/ {n=$NF; sub(/.*Function Name: /,""); name=$1} /VGPRs:/ {v=$(NF-1)} /AGPRs:/ {a=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /Occupancy/ {o=$(NF-1)} /LDS Size/ {l=$(NF-1); printf "%-28s VGPR %3s AGPR %3s scratch %4s B/lane  occupancy %s  LDS %6s\n", substr(name,1,28), v, a, s, o, l}'
