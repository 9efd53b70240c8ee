#!/bin/bash
# GPU box: time the scatter kernel of each ablation build (results are wrong by construction)
for f in ablation_libs/libpengk_abl*.so; do
  cp peng-motif_amd/libpengk.so /tmp/keep.so
  cp $f peng-motif_amd/libpengk.so
  echo "== $f"; HEADN=4 tools/kstats.sh abl_$(basename $f .so) 2>&1 | grep -E "scatter" | cut -c1-110
  cp /tmp/keep.so peng-motif_amd/libpengk.so
done
for b in 1 2 3; do echo "== blocks per CU $b"; PENGK_SCATTER_BLOCKS_PER_CU=$b HEADN=4 tools/kstats.sh occ_$b 2>&1 | grep -E "scatter" | cut -c1-110; done
