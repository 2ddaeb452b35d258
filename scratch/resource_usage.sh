#!/bin/bash
# Register / LDS / spill table of every kernel of the library: hipcc -Rpass-analysis=kernel-resource-usage over csrc/*.hip (no GPU
# needed: cross-compilation), formatted by scratch/resource_table.py.   usage: scratch/resource_usage.sh r04  -> profiles/r04_resource_usage.txt
set -euo pipefail
ROUND=${1:-r04}
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
CS="$ROOT/linpde-gp_amd/csrc"
TMP=$(mktemp -d)
pids=()
for f in api assemble gemm solve solve4 solve4p potrf chain trsv pcg dist; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$ROOT/include" -I"$CS" -Wall -Wno-unused-function \
    -Rpass-analysis=kernel-resource-usage -c "$CS/$f.hip" -o "$TMP/$f.o" 2> "$TMP/$f.txt" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
OUT="$ROOT/profiles/${ROUND}_resource_usage.txt"
{
  echo "# hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage over linpde-gp_amd/csrc/*.hip (scratch/resource_usage.sh $ROUND); $(/opt/rocm/bin/hipcc --version | grep -m1 'HIP version')"
  echo "# VGPR + AGPR = registers per lane (unified file, 512 per SIMD lane: waves per SIMD = floor(512 / allocation), granule 8); LDS(static) excludes the dynamic"
  echo "# shared memory of the launch: gemm_f64 73 728 B, gemm3_f64 36 864 B, gemm64_f64 73 728 / 36 864 B (ring of 4 / 2 stages), tile_solve 81 920 B,"
  echo "# panel_solve 69 632 B, potrf_tile 155 648 B.  Kernels launched in a c3 step: potrf_tile, tile_solve<false>, panel_solve<*, 1, *, *>, gemm_f64<false, *, *>,"
  echo "# gemm3_f64<true, 0>, gemm64_f64<false, *, *, *>, assemble_fast<2, 3, 3, 2>, assemble_fast<1, 3, 1, 2>, kron2<4, false>, add_diag, col_reduce2."
  python3 "$ROOT/scratch/resource_table.py" "$TMP"
} > "$OUT"
rm -rf "$TMP"
echo "wrote $OUT"
