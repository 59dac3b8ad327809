#!/bin/bash
# tools/mode_breakdown.py for each library variant given (paths relative to the repo root), on the interior at 4K and the atrium at 1080p.
mkdir -p gpurun_out
for lib in "$@"; do
  echo "== $lib"
  RT_HIP_LIB=$PWD/$lib timeout -k 10 200 python -u tools/mode_breakdown.py interior 3840 2160 4 2>&1 | grep -v exact_order || exit 1
  RT_HIP_LIB=$PWD/$lib timeout -k 10 100 python -u tools/mode_breakdown.py atrium 1920 1080 16 2>&1 | grep -v exact_order || exit 1
done
