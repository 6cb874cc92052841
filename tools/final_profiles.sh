# Everything under profiles/ that depends on the kernels, in one gpurun call:  gpurun -- bash tools/final_profiles.sh
# (needs build_ab/libegs_m3.so / libegs_m4.so: make OBJDIR=/tmp/m3 LIB=$PWD/build_ab/libegs_m3.so EXTRA=-DEGS_MEASURE=3, likewise 4)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( EGS_RASTER_LIB=$PWD/build_ab/libegs_m3.so TIMELINE=gpurun_out/fwd_timeline.npy python tools/lane_use.py 2>&1 | grep -v amdgpu.ids; python tools/timeline_analysis.py gpurun_out/fwd_timeline.npy fwd ) > gpurun_out/timeline_fwd.txt 2>&1
( EGS_RASTER_LIB=$PWD/build_ab/libegs_m4.so BACKWARD=1 TIMELINE=gpurun_out/bwd_timeline.npy python tools/lane_use.py 2>&1 | grep -v amdgpu.ids; python tools/timeline_analysis.py gpurun_out/bwd_timeline.npy bwd ) > gpurun_out/timeline_bwd.txt 2>&1
rm -f gpurun_out/*.npy
python examples/train_synth.py --gaussians 100000 --height 540 --width 960 --iters 6000 --frames 60 --densify-from 500 --densify-until 4000 --densify-interval 100 --opacity-reset-interval 3000 --capacity-factor 4 --report-every 500 --log gpurun_out/r2_train_synth_100k_capacity.log > gpurun_out/train_synth.out 2>&1
tail -3 gpurun_out/r2_train_synth_100k_capacity.log
tools/collect_counters.sh > gpurun_out/collect.log 2>&1; tail -3 gpurun_out/collect.log
cp gpurun_out/pmc_traffic.json gpurun_out/sq_counters.json profiles/      # the bench line below reports them (same kernel-source hash)
python bench.py > gpurun_out/r2_bench_n1.json 2> gpurun_out/r2_bench_n1.err; tail -c 600 gpurun_out/r2_bench_n1.json
tools/prof_bench.sh r2_bench --no-cpu-baseline --no-sh3-leg 2>&1 | tail -16
tools/prof_bench.sh r2_graph --steps 600 --warmup 20 --no-cpu-baseline --no-sh3-leg --no-fine-all-leg 2>&1 | tail -3
head -3 gpurun_out/r2_graph_step_budget.txt
