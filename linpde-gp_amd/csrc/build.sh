#!/bin/bash
# Build liblpgp.so for gfx950 (MI355X) in-tree.  Usage: build.sh [extra hipcc flags]
#        build.sh --host-asan   host-only TEST library of the descriptor lowering, g++ -fsanitize=address
#                               (csrc/hosttest/liblpgp_hosttest_asan.so; never loaded by the product)
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="$HERE/../linpde_gp_amd/_lib"
if [[ "${1:-}" == "--host-asan" ]]; then
  CXX="${CXX:-g++}"
  "$CXX" -std=c++17 -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined \
    -shared -fPIC -Wall -Wno-unknown-pragmas -I"$ROOT/include" -I"$HERE" \
    "$HERE/lower.cpp" "$HERE/hosttest/host_check.cpp" -o "$HERE/hosttest/liblpgp_hosttest_asan.so"
  echo "built $HERE/hosttest/liblpgp_hosttest_asan.so"
  exit 0
fi
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$HERE -Wall -Wno-unused-function"
pids=()
for f in api assemble gemm solve solve4 solve4p potrf chain trsv pcg dist testhooks; do
  "$HIPCC" $FLAGS "$@" -c "$HERE/$f.hip" -o "$OUT/$f.o" &
  pids+=($!)
done
"$HIPCC" -O2 -std=c++17 -fPIC -I"$ROOT/include" -I"$HERE" -Wall -c "$HERE/lower.cpp" -o "$OUT/lower.o" &
pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT/liblpgp.so" "$OUT/api.o" "$OUT/assemble.o" "$OUT/gemm.o" "$OUT/solve.o" "$OUT/solve4.o" "$OUT/solve4p.o" "$OUT/potrf.o" "$OUT/chain.o" "$OUT/trsv.o" "$OUT/pcg.o" "$OUT/dist.o" "$OUT/lower.o" \
  -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo "built $OUT/liblpgp.so"
# test hooks (include/lpgp_test.h): a library of their own, loaded by tests/ and scratch/ only
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT/liblpgp_testhooks.so" "$OUT/testhooks.o" -L"$OUT" -llpgp -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib
echo "built $OUT/liblpgp_testhooks.so"
