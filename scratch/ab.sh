#!/bin/bash
# A/B: HEAD library vs working-tree library on the GEMM sweep
L=linpde-gp_amd/linpde_gp_amd/_lib/liblpgp.so
cp $L /tmp/new.so
for r in 1 2; do
cp scratch/liblpgp_head.so $L; echo "== HEAD"; python scratch/gemm_sweep.py
cp /tmp/new.so $L; echo "== NEW"; python scratch/gemm_sweep.py
done
