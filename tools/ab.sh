#!/bin/bash
# A/B of libnempc variants on the headline launch: tools/ab.sh <tag|main> ...   (prints rows-kernel us, whole-eval us)
for tag in "$@"; do
  if [ "$tag" = main ]; then lib=""; else lib="$PWD/pyneuralempc_amd/build_$tag/libnempc_$tag.so"; fi
  NEMPC_LIB=$lib python bench.py --only-eval --steps 400 --warmup 50 ${AB_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline'] if d['roofline']['bound']=='mfma' else d['roofline_mfma_row_kernel']
print('$tag', 'rows_us=%.2f' % r['kernel_us'], 'eval_us=%.2f' % d['eval_us']['event_loop'], 'timed_us=%.2f' % d['eval_us']['timed_loop'], 'jac_err=%.1e' % d['jacobian_max_abs_err_vs_cpu'])"
done
