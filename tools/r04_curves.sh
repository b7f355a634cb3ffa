# GPU box: learning curves at the reference's lr 2e-5 (pioneer_knm_train.py:64) for each operand precision, 16 384 envs, T = 32, 32 768-sample minibatches
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
for P in 1 f32 bf16x2 0; do
  python3 $R/tools/train_curve.py ${1:-600} kinematic 32768 4000 400 0 $P 2e-5 > $O/train_curve_lr2e-5_$P.jsonl 2> $O/train_curve_lr2e-5_$P.err || { tail -5 $O/train_curve_lr2e-5_$P.err; exit 1; }
  tail -1 $O/train_curve_lr2e-5_$P.jsonl
done
