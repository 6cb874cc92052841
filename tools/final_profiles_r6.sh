# Round 6: everything under profiles/ that depends on the kernels, in one gpurun call:   gpurun -- bash tools/final_profiles_r6.sh
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# the whole GPU suite as the driver runs it, then the -s output of the parity files (per-config flip counts and worst errors)
python -m pytest tests -m gpu -q -p no:cacheprovider --durations=25 > gpurun_out/r6_gpu_tests.txt 2>&1; tail -3 gpurun_out/r6_gpu_tests.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_mode.py tests/test_gpu_trained_scene.py tests/test_gpu_sweep.py tests/test_gpu_label.py tests/test_gpu_offscreen.py tests/test_gpu_provenance.py tests/test_gpu_reentrant.py tests/test_gpu_densify.py -m gpu -q -s -p no:cacheprovider -k "not ranks" > gpurun_out/r6_parity_log.txt 2>&1; tail -3 gpurun_out/r6_parity_log.txt
tools/collect_counters.sh > gpurun_out/collect.log 2>&1; tail -3 gpurun_out/collect.log
cp gpurun_out/pmc_traffic.json gpurun_out/sq_counters.json profiles/      # the bench line below reports them (same kernel-source hash)
tools/prof_bench.sh r6_graph --steps 600 --warmup 20 --no-cpu-baseline --no-sh3-leg --no-fine-all-leg --no-config-legs 2>&1 | tail -3
head -3 gpurun_out/r6_graph_step_budget.txt
cp gpurun_out/r6_graph_step_budget.json profiles/graph_step_budget.json      # the roofline of the bench line below times its kernel in the replayed step
python bench.py > gpurun_out/r6_bench_n1.json 2> gpurun_out/r6_bench_n1.err; tail -c 400 gpurun_out/r6_bench_n1.json
python bench.py --steps 20 --warmup 5 > gpurun_out/r6_bench_n1_driver_cmd.json 2> gpurun_out/r6_bench_n1_driver_cmd.err; tail -c 300 gpurun_out/r6_bench_n1_driver_cmd.json
tools/prof_bench.sh r6_bench --no-cpu-baseline --no-sh3-leg --no-config-legs 2>&1 | tail -16
for c in "B 100000 540 960 30" "D 1000000 1080 1920 12"; do set -- $c; tools/prof_stages.sh r6_$1 $2 $3 $4 $5 > gpurun_out/r6_config_$1_kernels.txt 2>&1; cat gpurun_out/r6_config_$1_kernels.txt; done
SCENE=$GRAFT_REPO_ROOT/bench_data/trained_scene.npz tools/prof_stages.sh r6_trained 253202 540 960 30 > gpurun_out/r6_trained_scene_kernels.txt 2>&1; cat gpurun_out/r6_trained_scene_kernels.txt
tools/pmc_stages.sh r6_D 1000000 1080 1920 8 > gpurun_out/r6_config_D_traffic.txt 2>&1; cat gpurun_out/r6_config_D_traffic.txt
