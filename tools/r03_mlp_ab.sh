# usage: bash tools/r03_mlp_ab.sh NAME [variant.so ...]   -- per-kernel stats under rocprofv3 for the default library and variants
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
NAME=$1; shift
for lib in default "$@"; do
  tag=$(basename $lib .so)
  if [ "$lib" != default ]; then export PNR_LIB_PATH=$R/pioneer_amd/csrc/$lib; else unset PNR_LIB_PATH; fi
  python3 $R/tools/mlp_step_bench.py >> $O/mlp_ab_$NAME.jsonl
  rm -rf /tmp/prof_$tag
  rocprofv3 --kernel-trace -d /tmp/prof_$tag -o s -- python3 $R/tools/mlp_step_bench.py 32768 100 > /dev/null 2>&1
  db=$(find /tmp/prof_$tag -name '*.db' | head -1)
  echo "== $tag" >> $O/mlp_ab_${NAME}_kernels.txt
  python3 $R/tools/rocpd_stats.py $db --top 8 >> $O/mlp_ab_${NAME}_kernels.txt
done
cat $O/mlp_ab_$NAME.jsonl
