#!/bin/bash
# A/B of libnempc variants with tools/quick_bench.py: tools/ab_quick.sh <tag|main> ...
for tag in "$@"; do
  if [ "$tag" = main ]; then lib=""; else lib="$PWD/pyneuralempc_amd/build_$tag/libnempc_$tag.so"; fi
  echo -n "$tag: "; NEMPC_LIB=$lib python tools/quick_bench.py 2>/dev/null
done
