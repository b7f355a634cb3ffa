# GPU-box wrapper: bash tools/g.sh ROUND LOGNAME command...   -> runs the command with stdout+stderr in gpurun_out/ROUND/LOGNAME.log, prints its tail
RND=$1; LOG=$2; shift 2
mkdir -p gpurun_out/$RND
"$@" > gpurun_out/$RND/$LOG.log 2>&1; rc=$?
tail -40 gpurun_out/$RND/$LOG.log
exit $rc
