#!/bin/bash
# rocprofv3 kernel averages of the blend kernels for several library builds: LIBS="a.so b.so" SCRIPT=tools/dev/time_bwd_modes.py bash tools/dev/ab_kernels.sh
cd "$GRAFT_REPO_ROOT" || exit 1
root=$PWD; export TMPDIR=/tmp
mkdir -p gpurun_out; res=gpurun_out/ab_kernels.txt; : > $res
for lib in ${LIBS:-build_ab/libegs_base.so egogaussian_amd/libegs_raster.so}; do
  for scene in "" bench_data/trained_scene.npz; do
    for script in ${SCRIPTS:-tools/time_stages.py tools/dev/time_bwd_modes.py}; do
      out=/tmp/prof_ab; rm -rf $out; mkdir -p $out
      (cd /tmp && EGS_RASTER_LIB=$root/$lib SCENE=${scene:+$root/$scene} rocprofv3 --kernel-trace --stats -d $out -o run -- python $root/$script > $out/stdout.txt 2> $out/err.txt) || { tail -5 $out/err.txt; }
      db=$(find $out -name "*.db" | head -1)
      echo "== lib=$lib scene=${scene:-configC} script=$script" >> $res
      python tools/kstats.py "$db" k_render | sort >> $res
    done
  done
done
cat $res
