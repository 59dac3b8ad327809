#!/bin/bash
# One sample per call at several resolutions, per-path kernel off / at its default gate / wider: is the gate still right?
for res in "1920 1080" "1280 720" "640 360" "320 180"; do
  ONE_SAMPLE_SETTINGS="0:2 131072:2 131072:3 65536:2 32768:2 262144:2" timeout -k 10 200 python -u tools/one_sample.py atrium $res 8 2> gpurun_out/tgs.err | cut -c1-200 || { tail -3 gpurun_out/tgs.err; exit 1; }
done
ONE_SAMPLE_SETTINGS="0:2 131072:2 131072:1" timeout -k 10 200 python -u tools/one_sample.py cornell 256 256 4 2> gpurun_out/tgs.err | cut -c1-200
ONE_SAMPLE_SETTINGS="0:2 131072:2" timeout -k 10 200 python -u tools/one_sample.py car 640 360 8 2> gpurun_out/tgs.err | cut -c1-200
