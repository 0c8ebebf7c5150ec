#!/bin/bash
# configs[2] under every variant library given (tools/c3_ab.py --one): time and output hashes
for lib in "$@"; do
  echo "== $lib"
  timeout -k 10 120 python tools/c3_ab.py --one "$lib" 2>&1 | tail -1
done
