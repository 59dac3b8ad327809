#!/bin/bash
# The per-path kernel off / on (default gate) for one sample per call, over the scenes and image sizes where it is offered.
export ONE_SAMPLE_SETTINGS="0:2 131072:2"
run() { timeout -k 10 200 python -u tools/one_sample.py "$@" 2> gpurun_out/tt.err | cut -c1-220 || { tail -3 gpurun_out/tt.err; exit 1; }; }
run atrium 320 180 8 && run atrium 640 360 8 && run car 640 360 8 && run interior 480 270 8 && run glass_blob 640 360 8 && run cornell 256 256 4 && run cornell 512 512 4
