#!/bin/bash
# A/B of library variants inside bench.py at any shape: VARIANTS="base name1 name2" tools/ab_bench.sh [bench.py flags]
# (variants are cmtf_pls_amd/lib/variants/libcmtfpls_<name>.so, built by hand with -D flags; "base" = the library as built)
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
for v in ${VARIANTS:-base}; do
  if [ $v = base ]; then L=$ROOT/cmtf_pls_amd/lib/libcmtfpls.so; else L=$ROOT/cmtf_pls_amd/lib/variants/libcmtfpls_$v.so; fi
  CMTFPLS_LIB=$L timeout -k 10 300 python $ROOT/bench.py --no-cpu --no-fit --no-ceilings "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('%-10s' % '$v', 'it/s %.2f' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'contract %.0f' % d['kernels']['mode0_contract']['GBps'], 'score %.0f' % d['kernels']['score']['GBps'], 'rank1 %.1f us' % (1e3 * d['kernels']['rank1']['ms']))" || exit 1
done
