set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r03
for cfg in "1 0" "2 0" "2 1"; do
  set -- $cfg
  export PROBE_PARTS=$1 PROBE_GRAPH=$2
  rm -rf /tmp/prof_$1_$2
  rocprofv3 --kernel-trace -d /tmp/prof_$1_$2 -o t -- python3 $R/tools/two_stream_probe.py kinematic 65536 > $R/gpurun_out/r03/overlap_probe_$1_$2.log 2>&1
  db=$(find /tmp/prof_$1_$2 -name '*.db' | head -1)
  python3 $R/tools/rocpd_overlap.py $db --last 64 --json $R/gpurun_out/r03/overlap_$1_$2.json > $R/gpurun_out/r03/overlap_$1_$2.txt
done
