set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for P in 3 2; do
  export PLANES=$P
  rm -rf /tmp/pp$P
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d /tmp/pp$P -o t -- python3 $R/tools/mlp_step_bench.py 32768 100 > $O/pmc_planes$P.log 2>&1
  python3 $R/tools/rocpd_pmc.py $(find /tmp/pp$P -name '*.db' | head -1) $O/learner_planes${P}_pmc_sq.json > /dev/null
done
echo done
