#!/bin/bash
# build count.hip variants with -DPENGK_ABLATE=<n> into separate .so copies (run in the build container)
set -e
for n in "$@"; do
  PENGK_EXTRA_FLAGS="-DPENGK_ABLATE=$n" python peng-motif_amd/build.py --force >/dev/null
  cp peng-motif_amd/libpengk.so ablation_libs/libpengk_abl$n.so
done
python peng-motif_amd/build.py --force >/dev/null
