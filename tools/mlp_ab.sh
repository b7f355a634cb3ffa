# A/B of MLP-kernel build variants (GPU box).  Usage: bash tools/mlp_ab.sh ROUND NAME name1:-DFLAG=1,-DX=2 [name2:...]
# Builds each variant of pnr_learn.hip next to the default library, then for default + variants: tools/mlp_step_bench.py (HIP events)
# and per-kernel stats under rocprofv3 --kernel-trace.  Output: gpurun_out/ROUND/mlp_ab_NAME.jsonl / _kernels.txt
set -e
R=$GRAFT_REPO_ROOT; RND=$1; NAME=$2; shift 2
O=$R/gpurun_out/$RND; mkdir -p $O
LIBS="default"
for spec in "$@"; do
  n=${spec%%:*}; flags=${spec#*:}
  python3 - "$n" "$flags" <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from pioneer_amd import _lib
n, flags = sys.argv[1], [f for f in sys.argv[2].split(",") if f]
_lib.build_library(extra_flags=flags, out_path=os.path.join(_lib.CSRC, f"libpioneer_amd_{n}.so"), units=("pnr_learn.hip",))
PY
  LIBS="$LIBS libpioneer_amd_$n.so"
done
cd /tmp && export TMPDIR=/tmp
for lib in $LIBS; do
  tag=$(basename $lib .so)
  if [ "$lib" != default ]; then export PNR_LIB_PATH=$R/pioneer_amd/csrc/$lib; else unset PNR_LIB_PATH; fi
  python3 $R/tools/mlp_step_bench.py >> $O/mlp_ab_$NAME.jsonl
  rm -rf /tmp/prof_$tag
  rocprofv3 --kernel-trace -d /tmp/prof_$tag -o s -- python3 $R/tools/mlp_step_bench.py 32768 100 > /dev/null 2>&1
  db=$(find /tmp/prof_$tag -name '*.db' | head -1)
  echo "== $tag" >> $O/mlp_ab_${NAME}_kernels.txt
  python3 $R/tools/rocpd_stats.py $db --top 5 >> $O/mlp_ab_${NAME}_kernels.txt
done
cat $O/mlp_ab_$NAME.jsonl
grep -A7 "^==" $O/mlp_ab_${NAME}_kernels.txt | grep "==\|mlp_forward_kernelILb1\|mlp_wgrad\|mlp_adam\|mlp_train"
