# usage: bash tools/r03_train_ab.sh NAME   -- mlp_train_kernel (default library) against mlp_forward_kernel<true> (libpioneer_amd_stream.so) + its phase stamps
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
bash $R/tools/r03_mlp_ab.sh $1 libpioneer_amd_stream.so > /dev/null
cat $O/mlp_ab_$1.jsonl
grep -E "^==|mlp_" $O/mlp_ab_$1_kernels.txt || true
grep -E "^==|mlp_" $O/mlp_ab_${1}_kernels.txt
PNR_LIB_PATH=$R/pioneer_amd/csrc/libpioneer_amd_stamps.so python3 $R/tools/mlp_train_stamps.py $O/train_stamps_$1.json
