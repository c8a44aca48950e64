#!/usr/bin/env bash
# An option's values on the bench workload at several batch sizes, same box, alternating: tools/gpu_sizes_opt.sh <tag> <ENVNAME> "<values>" <sizes ...>
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=$1; E=$2; VALS=$3; shift 3; O=gpurun_out/$T; mkdir -p $O
for r in 1 2; do for V in $VALS; do echo "== $E=$V"; env $E=$V python tools/gpu_modes.py "$@" 2>&1 | grep pairs | sed -e 's/mode 4:.*mode 0:/auto:/'; done; done | tee $O/sizes.txt
