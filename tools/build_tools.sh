#!/usr/bin/env bash
# build_tools.sh — diagnostics binaries of tools/*.hip for gfx950 into tools/_bin/ (git-ignored; travels to the GPU box with the snapshot)
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
mkdir -p "$HERE/_bin"
for src in "$HERE"/*.hip; do
	out="$HERE/_bin/$(basename "${src%.hip}" | sed 's/^gpu_//')"
	/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-function -o "$out" "$src"
	echo "built $out"
done
