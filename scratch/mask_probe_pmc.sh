#!/bin/bash
# FETCH_SIZE / busy cycles of the stand-alone SYRK under a CU mask (scratch/mask_probe.py), per setting of LPGP_RESERVE_CUS
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=gpurun_out/r4d; rm -rf $D; mkdir -p $D
export LPGP_TEST_GEMM_STREAM=1
for r in -1 8 32; do
  export LPGP_RESERVE_CUS=$r
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/fetch_$r -- python3 scratch/mask_probe.py > $D/fetch_$r.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $D/mfma_$r -- python3 scratch/mask_probe.py > $D/mfma_$r.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for r in (-1, 8, 32):
    for kind in ("fetch", "mfma"):
        for f in glob.glob(f"gpurun_out/r4d/{kind}_{r}/*/*counter_collection.csv"):
            acc = collections.defaultdict(list)
            for row in csv.DictReader(open(f)):
                if "gemm_f64_kernel" in row["Kernel_Name"]:
                    acc[(row["Grid_Size"], row["Counter_Name"])].append(float(row["Counter_Value"]))
            for k, v in sorted(acc.items()):
                print(f"reserve {r:3d} grid {k[0]:>9s} {k[1]:28s} n={len(v):3d} mean {sum(v)/len(v):.4g}")
PY
