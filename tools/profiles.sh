# Profiles in one call (GPU box).  Usage: bash tools/profiles.sh ROUND TAG [what ...]   -> gpurun_out/ROUND/prof_TAG_*
#   what: main (headline kernel stats + FETCH/WRITE traffic), dyn (SQ counters), dyntraffic, ppo (kernel stats + SQ/MFMA counters),
#         learner (FETCH/WRITE traffic of the learner kernels + per-kernel stats of tools/mlp_step_bench.py); default: all
set -e
R=$GRAFT_REPO_ROOT; RND=${1:-r04}; TAG=${2:-x}; shift 2 || true
WHAT="${*:-main dyn dyntraffic ppo learner}"
O=$R/gpurun_out/$RND; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
MAIN="$R/bench.py --steps 2000 --warmup 100 --fused-leg 0 --large-envs 0 --dynamic-leg 0 --split-leg 0 --ppo-iters 0 --no-cpu-baseline"
DYN="$R/bench.py --mode dynamic --randomize --gravity 9.81 --warmup 20 --no-cpu-baseline --ppo-iters 0 --large-envs 0"
prof() { # name, rocprof args..., -- program args
  name=$1; shift
  rm -rf /tmp/p_$name
  rocprofv3 "$@" > $O/prof_${TAG}_$name.log 2>&1 || { tail -5 $O/prof_${TAG}_$name.log; exit 1; }
  find /tmp/p_$name -name '*.db' | head -1
}
has() { case " $WHAT " in *" $1 "*) return 0;; *) return 1;; esac; }
if has main; then
  db=$(prof main --kernel-trace -d /tmp/p_main -o t -- python3 $MAIN)
  python3 $R/tools/rocpd_stats.py $db --csv $O/prof_${TAG}_step_65536_kernel_stats.csv > $O/prof_${TAG}_step_65536_kernel_stats.txt
  grep '^{' $O/prof_${TAG}_main.log | tail -1 > $O/prof_${TAG}_bench_main_leg_under_rocprof.json
  db=$(prof fetch --kernel-trace --pmc FETCH_SIZE -d /tmp/p_fetch -o t -- python3 $MAIN)
  python3 $R/tools/rocpd_pmc.py $db $O/prof_${TAG}_pmc_fetch.json > /dev/null
  db=$(prof write --kernel-trace --pmc WRITE_SIZE -d /tmp/p_write -o t -- python3 $MAIN)
  python3 $R/tools/rocpd_pmc.py $db $O/prof_${TAG}_pmc_write.json > /dev/null
  cp $R/profiles/pmc_traffic.json $O/prof_${TAG}_pmc_traffic.json
  python3 $R/tools/pmc_traffic.py $O/prof_${TAG}_pmc_fetch.json $O/prof_${TAG}_pmc_write.json $O/prof_${TAG}_pmc_traffic.json "$RND $TAG" kinematic > $O/prof_${TAG}_pmc_traffic.txt
fi
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
if has dyn; then
  db=$(prof dyn1 --kernel-trace --pmc $SQ -d /tmp/p_dyn1 -o t -- python3 $DYN --steps 200)
  python3 $R/tools/rocpd_pmc.py $db $O/prof_${TAG}_dyn_step_pmc.json > /dev/null
  db=$(prof dyn32 --kernel-trace --pmc $SQ -d /tmp/p_dyn32 -o t -- python3 $DYN --fused 32 --steps 320)
  python3 $R/tools/rocpd_pmc.py $db $O/prof_${TAG}_dyn_rollout_pmc.json > /dev/null
  python3 $R/tools/dyn_counters_summary.py $O/prof_${TAG}_dyn_step_pmc.json $O/prof_${TAG}_dyn_rollout_pmc.json $O/prof_${TAG}_dyn_sq_counters.json > $O/prof_${TAG}_dyn_sq_counters.txt
fi
if has dyntraffic; then
  db=$(prof dynf --kernel-trace --pmc FETCH_SIZE -d /tmp/p_dynf -o t -- python3 $DYN --steps 200)
  python3 $R/tools/rocpd_pmc.py $db $O/prof_${TAG}_dyn_pmc_fetch.json > /dev/null
  db=$(prof dynw --kernel-trace --pmc WRITE_SIZE -d /tmp/p_dynw -o t -- python3 $DYN --steps 200)
  python3 $R/tools/rocpd_pmc.py $db $O/prof_${TAG}_dyn_pmc_write.json > /dev/null
  [ -f $O/prof_${TAG}_pmc_traffic.json ] || cp $R/profiles/pmc_traffic.json $O/prof_${TAG}_pmc_traffic.json
  python3 $R/tools/pmc_traffic.py $O/prof_${TAG}_dyn_pmc_fetch.json $O/prof_${TAG}_dyn_pmc_write.json $O/prof_${TAG}_pmc_traffic.json "$RND $TAG" dynamic > $O/prof_${TAG}_dyn_pmc_traffic.txt
fi
if has ppo; then
  SQM="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
  for PREC in f32 bf16; do
    export PPO_PROFILE_PRECISION=$PREC
    db=$(prof ppo$PREC --kernel-trace -d /tmp/p_ppo$PREC -o t -- python3 $R/tools/ppo_profile.py)
    python3 $R/tools/rocpd_stats.py $db --skip-frac 0.6 --csv $O/prof_${TAG}_ppo_loop_${PREC}_kernel_stats.csv > $O/prof_${TAG}_ppo_loop_${PREC}_kernel_stats.txt
    db=$(prof ppopmc$PREC --kernel-trace --pmc $SQM -d /tmp/p_ppopmc$PREC -o t -- python3 $R/tools/ppo_profile.py)
    python3 $R/tools/rocpd_pmc.py $db $O/prof_${TAG}_learner_${PREC}_pmc_sq.json > /dev/null
  done
  unset PPO_PROFILE_PRECISION
fi
if has learner; then
  cp $R/profiles/learner_pmc_traffic.json $O/prof_${TAG}_learner_pmc_traffic.json 2>/dev/null || true
  for P in 1 2 3; do
    export PLANES=$P
    STEP="$R/tools/mlp_step_bench.py 32768 100"
    db=$(prof lstat$P --kernel-trace -d /tmp/p_lstat$P -o t -- python3 $STEP)
    python3 $R/tools/rocpd_stats.py $db --top 8 --csv $O/prof_${TAG}_learner_planes${P}_kernel_stats.csv > $O/prof_${TAG}_learner_planes${P}_kernel_stats.txt
    db=$(prof lfetch$P --kernel-trace --pmc FETCH_SIZE -d /tmp/p_lfetch$P -o t -- python3 $STEP)
    python3 $R/tools/rocpd_pmc.py $db $O/prof_${TAG}_learner_planes${P}_pmc_fetch.json > /dev/null
    db=$(prof lwrite$P --kernel-trace --pmc WRITE_SIZE -d /tmp/p_lwrite$P -o t -- python3 $STEP)
    python3 $R/tools/rocpd_pmc.py $db $O/prof_${TAG}_learner_planes${P}_pmc_write.json > /dev/null
    python3 $R/tools/learner_traffic.py $O/prof_${TAG}_learner_planes${P}_pmc_fetch.json $O/prof_${TAG}_learner_planes${P}_pmc_write.json $O/prof_${TAG}_learner_pmc_traffic.json "$RND $TAG" 32768 $P > $O/prof_${TAG}_learner_planes${P}_pmc_traffic.txt
  done
  unset PLANES
fi
echo profiles done
