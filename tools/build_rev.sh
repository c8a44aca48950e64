#!/usr/bin/env bash
# The library as a revision of this repository builds it, for same-box A/B timing: tools/build_rev.sh <name> [rev = HEAD]  -> tools/_bin/libmulls_<name>.so
# (a clean export of the revision under /tmp, its own python -m mulls_amd.build; select the result with MULLS_HIP_LIB)
set -eu
cd "$(dirname "$0")/.."; mkdir -p tools/_bin
N=$1; R=${2:-HEAD}; D=/tmp/mulls_rev_$N
rm -rf "$D"; mkdir -p "$D"; git archive "$R" | tar -x -C "$D"
(cd "$D" && python -m mulls_amd.build > /dev/null)
cp "$D/mulls_amd/libmulls_hip.so" tools/_bin/libmulls_$N.so
echo tools/_bin/libmulls_$N.so
