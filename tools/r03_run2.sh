set -e
mkdir -p gpurun_out/r03
hipcc -O2 --offload-arch=gfx950 -o /tmp/valu_issue_probe tools/valu_issue_probe.hip
/tmp/valu_issue_probe > gpurun_out/r03/valu_issue_probe.jsonl
for n in 8192 65536 131072 262144; do
  python tools/dyn_fixed_cost.py $n >> gpurun_out/r03/dyn_fixed_cost.jsonl
done
