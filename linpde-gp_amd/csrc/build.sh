#!/bin/bash
# Build liblpgp.so for gfx950 (MI355X) in-tree.  Usage: build.sh [extra hipcc flags]
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="$HERE/../linpde_gp_amd/_lib"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$HERE -Wall -Wno-unused-function"
pids=()
for f in api assemble gemm potrf; do
  "$HIPCC" $FLAGS "$@" -c "$HERE/$f.hip" -o "$OUT/$f.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT/liblpgp.so" "$OUT/api.o" "$OUT/assemble.o" "$OUT/gemm.o" "$OUT/potrf.o" \
  -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo "built $OUT/liblpgp.so"
