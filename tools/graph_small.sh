#!/bin/bash
# GPU box: one sample per call on small images, short passes replayed from HIP graphs (RT_HIP_GRAPH=1) or launched directly.
export ONE_SAMPLE_SETTINGS="131072:2"
for rep in 1 2; do for g in 0 1; do
  for a in "cornell 256 256 4" "cornell 512 512 4" "atrium 320 180 8" "atrium 640 360 8" "car 640 360 8"; do
    RT_HIP_GRAPH=$g timeout -k 10 200 python -u tools/one_sample.py $a 2> gpurun_out/gs.err | sed "s/^/graph=$g  /" | cut -c1-150 || { tail -3 gpurun_out/gs.err; exit 1; }
  done
done; done
