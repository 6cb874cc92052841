# Round 4: everything under profiles/ that depends on the kernels, in one gpurun call:   gpurun -- bash tools/final_profiles_r4.sh
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# parity log first: the -s output of the parity tests of configs A-D and the trained scene, from this same call
python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_mode.py tests/test_gpu_trained_scene.py tests/test_gpu_sweep.py -m gpu -q -s -p no:cacheprovider > gpurun_out/r4_parity_log.txt 2>&1; tail -3 gpurun_out/r4_parity_log.txt
tools/collect_counters.sh > gpurun_out/collect.log 2>&1; tail -3 gpurun_out/collect.log
cp gpurun_out/pmc_traffic.json gpurun_out/sq_counters.json profiles/      # the bench line below reports them (same kernel-source hash)
python bench.py > gpurun_out/r4_bench_n1.json 2> gpurun_out/r4_bench_n1.err; tail -c 400 gpurun_out/r4_bench_n1.json
tools/prof_bench.sh r4_bench --no-cpu-baseline --no-sh3-leg --no-config-legs 2>&1 | tail -16
tools/prof_bench.sh r4_graph --steps 600 --warmup 20 --no-cpu-baseline --no-sh3-leg --no-fine-all-leg --no-config-legs 2>&1 | tail -3
head -3 gpurun_out/r4_graph_step_budget.txt
for c in "B 100000 540 960 30" "D 1000000 1080 1920 12"; do set -- $c; tools/prof_stages.sh r4_$1 $2 $3 $4 $5 > gpurun_out/r4_config_$1_kernels.txt 2>&1; cat gpurun_out/r4_config_$1_kernels.txt; done
SCENE=$GRAFT_REPO_ROOT/bench_data/trained_scene.npz tools/prof_stages.sh r4_trained 253202 540 960 30 > gpurun_out/r4_trained_scene_kernels.txt 2>&1; cat gpurun_out/r4_trained_scene_kernels.txt
tools/pmc_stages.sh r4_D 1000000 1080 1920 8 > gpurun_out/r4_config_D_traffic.txt 2>&1; cat gpurun_out/r4_config_D_traffic.txt
SCENE=$GRAFT_REPO_ROOT/bench_data/trained_scene.npz tools/pmc_stages.sh r4_trained 253202 540 960 12 > gpurun_out/r4_trained_scene_traffic.txt 2>&1; cat gpurun_out/r4_trained_scene_traffic.txt
