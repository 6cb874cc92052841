#!/bin/bash
# headline A/B of library builds: LIBS="a.so b.so" bash tools/dev/ab_bench.sh   (value, median, spread of bench.py's default run; unchanged-trainer legs skipped)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; res=gpurun_out/ab_bench.txt; : > $res
for rep in ${REPS:-1 2}; do
for lib in ${LIBS:-build_ab/libegs_base.so egogaussian_amd/libegs_raster.so}; do
  echo "== lib=$lib rep=$rep" >> $res
  EGS_RASTER_LIB=$PWD/$lib EGS_BENCH_SKIP_EXTRA_LEGS=1 timeout 600 python bench.py ${BENCH_ARGS:---steps 200 --warmup 20 --no-sh3-leg --no-fine-all-leg --no-cpu-baseline --no-config-legs} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('value',d['value'],'median',d.get('value_median'),'spread',d.get('value_spread'),'ms',d['ms_per_step'])
st=d.get('stages') or d.get('roofline',{}).get('stages')
print({k:(round(v.get('ms_per_launch',0)*1e3,1) if isinstance(v,dict) else v) for k,v in (st or {}).items()})
" >> $res 2>&1
done; done
cat $res
