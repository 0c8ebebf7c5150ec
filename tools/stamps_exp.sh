#!/bin/bash
# tools/stamps_exp.sh "<-D flags>" ...  : rebuild the stamps library with each flag set in turn and print its timeline
for flags in "$@"; do
  python tools/diag_stamps.py --build-only $flags > /dev/null 2>&1
  echo "=== $flags"
  python tools/diag_stamps.py 1024 mfma 2>&1 | grep -E "span|staged|p1 done|exit|weights issued|stage_ld" | head -14
done
