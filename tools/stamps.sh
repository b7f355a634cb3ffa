# GPU box: build the -DPNR_MLP_STAMPS=1 variant of pnr_learn.hip and run a stamps tool on it.  Usage: bash tools/stamps.sh TOOL.py OUT.json [extra -D flags]
set -e
R=$GRAFT_REPO_ROOT; TOOL=$1; OUT=$2; shift 2
python3 - "$@" <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from pioneer_amd import _lib
_lib.build_library(extra_flags=["-DPNR_MLP_STAMPS=1"] + sys.argv[1:], out_path=os.path.join(_lib.CSRC, "libpioneer_amd_stamps.so"), units=("pnr_learn.hip",))
PY
mkdir -p $(dirname $OUT)
PNR_LIB_PATH=$R/pioneer_amd/csrc/libpioneer_amd_stamps.so python3 $R/tools/$TOOL $OUT
