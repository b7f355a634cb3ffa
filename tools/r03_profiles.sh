# Round-3 profiles in one call (GPU box).  Usage: bash tools/r03_profiles.sh TAG   -> gpurun_out/r03/prof_TAG_*
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp
MAIN="$R/bench.py --steps 2000 --warmup 100 --fused-leg 0 --large-envs 0 --dynamic-leg 0 --split-leg 0 --ppo-iters 0 --no-cpu-baseline"
prof() { # name, rocprof args..., -- program args
  name=$1; shift
  rm -rf /tmp/p_$name
  rocprofv3 "$@" > $O/prof_${TAG}_$name.log 2>&1 || { tail -5 $O/prof_${TAG}_$name.log; exit 1; }
  find /tmp/p_$name -name '*.db' | head -1
}
# 1) headline kernel: kernel trace, then the two traffic passes
db=$(prof main --kernel-trace -d /tmp/p_main -o t -- python3 $MAIN)
python3 $R/tools/rocpd_stats.py $db --csv $O/prof_${TAG}_step_65536_kernel_stats.csv > $O/prof_${TAG}_step_65536_kernel_stats.txt
grep '^{' $O/prof_${TAG}_main.log | tail -1 > $O/prof_${TAG}_bench_main_leg_under_rocprof.json
db=$(prof fetch --kernel-trace --pmc FETCH_SIZE -d /tmp/p_fetch -o t -- python3 $MAIN)
python3 $R/tools/rocpd_pmc.py $db $O/prof_${TAG}_pmc_fetch.json > /dev/null
db=$(prof write --kernel-trace --pmc WRITE_SIZE -d /tmp/p_write -o t -- python3 $MAIN)
python3 $R/tools/rocpd_pmc.py $db $O/prof_${TAG}_pmc_write.json > /dev/null
python3 $R/tools/pmc_traffic.py $O/prof_${TAG}_pmc_fetch.json $O/prof_${TAG}_pmc_write.json $O/prof_${TAG}_pmc_traffic.json "r03 $TAG" > $O/prof_${TAG}_pmc_traffic.txt
# 2) dynamics kernels: SQ counters (single step, then the 32-step rollout)
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
DYN="$R/bench.py --mode dynamic --randomize --gravity 9.81 --warmup 20 --no-cpu-baseline --ppo-iters 0 --large-envs 0"
db=$(prof dyn1 --kernel-trace --pmc $SQ -d /tmp/p_dyn1 -o t -- python3 $DYN --steps 200)
python3 $R/tools/rocpd_pmc.py $db $O/prof_${TAG}_dyn_step_pmc.json > /dev/null
db=$(prof dyn32 --kernel-trace --pmc $SQ -d /tmp/p_dyn32 -o t -- python3 $DYN --fused 32 --steps 320)
python3 $R/tools/rocpd_pmc.py $db $O/prof_${TAG}_dyn_rollout_pmc.json > /dev/null
python3 $R/tools/dyn_counters_summary.py $O/prof_${TAG}_dyn_step_pmc.json $O/prof_${TAG}_dyn_rollout_pmc.json $O/prof_${TAG}_dyn_sq_counters.json > $O/prof_${TAG}_dyn_sq_counters.txt
# 3) the PPO loop: per-kernel stats, then the learner's SQ / MFMA counters
db=$(prof ppo --kernel-trace -d /tmp/p_ppo -o t -- python3 $R/tools/ppo_profile.py)
python3 $R/tools/rocpd_stats.py $db --skip-frac 0.6 --csv $O/prof_${TAG}_ppo_loop_kernel_stats.csv > $O/prof_${TAG}_ppo_loop_kernel_stats.txt
SQM="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
db=$(prof ppopmc --kernel-trace --pmc $SQM -d /tmp/p_ppopmc -o t -- python3 $R/tools/ppo_profile.py)
python3 $R/tools/rocpd_pmc.py $db $O/prof_${TAG}_learner_pmc_sq.json > /dev/null
echo profiles done
