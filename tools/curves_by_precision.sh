# GPU box: learning curves per operand precision (bf16 / f32 / bf16x3 kernels and the torch float32 learner), 16 384 envs, T = 32, 32 768-sample
# minibatches.  Usage: bash tools/curves_by_precision.sh ROUND [ITERATIONS=600] [LR=2e-5: the reference's, pioneer_knm_train.py:64]
set -e
R=$GRAFT_REPO_ROOT; RND=${1:-r04}; IT=${2:-600}; LR=${3:-2e-5}; O=$R/gpurun_out/$RND; mkdir -p $O
for P in 1 f32 bf16x3 0; do
  python3 $R/tools/train_curve.py $IT kinematic 32768 4000 400 0 $P $LR > $O/train_curve_lr${LR}_$P.jsonl 2> $O/train_curve_lr${LR}_$P.err || { tail -5 $O/train_curve_lr${LR}_$P.err; exit 1; }
  tail -1 $O/train_curve_lr${LR}_$P.jsonl
done
