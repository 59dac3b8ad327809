#!/bin/bash
# Per library variant: the driver-size bench line and one 1080p / one 640x360 sample per call (per-path kernel at its defaults).
mkdir -p gpurun_out
for lib in "$@"; do
  name=$(basename $lib .so)
  bash tools/variant_bench.sh $lib || exit 1
  for res in "1920 1080" "640 360"; do
    ONE_SAMPLE_SETTINGS="0:3" RT_HIP_LIB=$PWD/$lib timeout -k 10 120 python -u tools/one_sample.py atrium $res 8 2> gpurun_out/vos.err | sed "s/^/$name  /" | cut -c1-160 || { tail -3 gpurun_out/vos.err; exit 1; }
  done
done
