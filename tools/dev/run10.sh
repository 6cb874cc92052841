timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -6
for c in "C 500000 540 960 20" "D 1000000 1080 1920 8"; do
    set -- $c
    echo "=== $1: $(timeout 100 tools/prof_stages.sh $1_n $2 $3 $4 $5 2>&1 | grep -E "k_preprocess|k_render_forward" | tr '\n' ' ')"
done
timeout 900 python bench.py --no-config-legs --no-sh3-leg --no-cpu-baseline --no-fine-all-leg 2> gpurun_out/b10.err | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', j['value'], j['ms_per_step'], j['value_median'], j['value_spread']); print({k: v['ms_per_launch'] for k, v in j['stages'].items()})"
