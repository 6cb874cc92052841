# Round 3: everything under profiles/ that depends on the kernels, in one gpurun call:   gpurun -- bash tools/final_profiles_r3.sh
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/collect_counters.sh > gpurun_out/collect.log 2>&1; tail -3 gpurun_out/collect.log
cp gpurun_out/pmc_traffic.json gpurun_out/sq_counters.json profiles/      # the bench line below reports them (same kernel-source hash)
python bench.py > gpurun_out/r3_bench_n1.json 2> gpurun_out/r3_bench_n1.err; tail -c 400 gpurun_out/r3_bench_n1.json
tools/prof_bench.sh r3_bench --no-cpu-baseline --no-sh3-leg --no-config-legs 2>&1 | tail -16
tools/prof_bench.sh r3_graph --steps 600 --warmup 20 --no-cpu-baseline --no-sh3-leg --no-fine-all-leg --no-config-legs 2>&1 | tail -3
head -3 gpurun_out/r3_graph_step_budget.txt
python bench.py --footprints --steps 20 --warmup 5 --no-cpu-baseline --no-sh3-leg --no-fine-all-leg --no-config-legs > gpurun_out/r3_footprints.json 2> gpurun_out/r3_footprints.err
python tools/footprint_md.py gpurun_out/r3_footprints.json gpurun_out/r3_bench_n1.json > gpurun_out/r3_footprint_sweep.md; head -20 gpurun_out/r3_footprint_sweep.md
for c in "B 100000 540 960 30" "D 1000000 1080 1920 12"; do set -- $c; tools/prof_stages.sh r3_$1 $2 $3 $4 $5 > gpurun_out/r3_config_$1_kernels.txt 2>&1; cat gpurun_out/r3_config_$1_kernels.txt; done
