#!/bin/bash
# A/B of the mode-0 contraction forms at BASELINE configs[4] inside bench.py (same box, same run conditions):
# cr0 = column-tile form only (-DCMTFPLS_CONTRACT_ROWS_MINSEG=0), base = the library as built.  Prints the contraction
# Usage: tools/ab_contract_cfg5.sh [rows]   (default 262144; 65536 / 32768 = the shard one GPU sees at N = 4 / 8).
# and score rates next to the plain read ceilings of the box (column-owner map vs contiguous chunks).
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
ROWS="${1:-262144}"
for v in ${VARIANTS:-cr0 base cr0 base}; do
  if [ $v = base ]; then L=$ROOT/cmtf_pls_amd/lib/libcmtfpls.so; else L=$ROOT/cmtf_pls_amd/lib/variants/libcmtfpls_$v.so; fi
  CMTFPLS_LIB=$L timeout -k 10 300 python $ROOT/bench.py --shape $ROWS 256 256 --responses 32 --steps 6 --warmup 2 --no-cpu --no-fit 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
c=d['roofline']['ceilings']['variants']
print('$v', 'it/s %.2f' % d['value'], 'contract %.0f' % d['kernels']['mode0_contract']['GBps'], 'score %.0f' % d['kernels']['score']['GBps'], '| ceilings colowner:512 %.0f chunk:1024 %.0f flat:4096 %.0f' % (c['read:colowner:512'], c['read:chunk:1024'], c['read:flat:4096']))" || exit 1
done
