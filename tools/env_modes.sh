#!/bin/bash
# tools/mode_breakdown.py (default mode rows) under each environment setting given as VAR=value arguments ("-" = none).
for kv in "$@"; do
  echo "== $kv"
  if [ "$kv" = "-" ]; then kv="RT_NOTHING=1"; fi
  env "$kv" timeout -k 10 200 python -u tools/mode_breakdown.py interior 3840 2160 4 2>&1 | grep -v exact_order || exit 1
  env "$kv" timeout -k 10 100 python -u tools/mode_breakdown.py atrium 1920 1080 16 2>&1 | grep -v exact_order || exit 1
done
