# GPU box: the learner's update and the sampler's forward at the three operand precisions (HIP events + per-kernel stats)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for P in 1 2 3; do
  PLANES=$P python3 $R/tools/mlp_step_bench.py 32768 200 >> $O/mlp_step_planes.jsonl
  rm -rf /tmp/prof_p$P
  PLANES=$P rocprofv3 --kernel-trace -d /tmp/prof_p$P -o s -- python3 $R/tools/mlp_step_bench.py 32768 100 > /dev/null 2>&1
  db=$(find /tmp/prof_p$P -name '*.db' | head -1)
  echo "== planes $P" >> $O/mlp_step_planes_kernels.txt
  python3 $R/tools/rocpd_stats.py $db --top 6 >> $O/mlp_step_planes_kernels.txt
done
cat $O/mlp_step_planes.jsonl
cat $O/mlp_step_planes_kernels.txt
