#!/usr/bin/env python3
"""Turn the per-kernel counter JSONs of tools/rocpd_pmc.py (one rocprofv3 --pmc pass over `bench.py --mode dynamic
--randomize --gravity 9.81 ...` for the single-step kernel, one with `--fused 32` for the rollout kernel) into the summary
bench.py reads for the dynamics leg's VALU roofline (profiles/r02_*_dyn_sq_counters.json).
Usage: python tools/dyn_counters_summary.py STEP_PMC.json ROLLOUT_PMC.json OUT.json"""
import json
import sys

HOW = ("rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU "
       "SQ_WAIT_ANY SQ_WAIT_INST_ANY over `python bench.py --mode dynamic --randomize --gravity 9.81 --steps 200 --warmup 20 "
       "--no-cpu-baseline --ppo-iters 0 --large-envs 0` (and `--fused 32 --steps 320` for the rollout kernel); per-dispatch sums over "
       "all counter instances, averaged over the dispatches (tools/rocpd_pmc.py); 65 536 envs, randomised, gravity 9.81; "
       "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles")


def one(path, needle, steps, waves_per_wg=1):
    ks = json.load(open(path))["kernels"]
    name = [k for k in ks if needle in k and ("ILb1ELb1ELb1ELi0E" in k or "ILb1ELb1ELb1ELb0E" in k)][0]   # <env-major obs, env-major actions, randomised, PHYS 0>
    k = ks[name]
    c = k["counters_per_dispatch"]
    # dyn_step_kernel's workgroups are two waves since r03: wave 1 only waits during phase A and finishes one tile in phase B, so the
    # per-wave figures are quoted per PHASE-A wave (= per 64 envs), with the helper waves' cycles and instructions folded in
    waves = c["SQ_WAVES"] / waves_per_wg
    life = c["SQ_WAVE_CYCLES"] / c["SQ_WAVES"] / steps
    dur = k["avg_duration_us"]
    instr = c["SQ_INSTS_VALU"]
    simd_cycles = dur * 1e-6 * 2.4e9 * 1024
    return {"avg_duration_us_under_profiler": dur, "steps_per_dispatch": steps, "per_dispatch": c,
            "per_wave_per_step": {"valu_instructions": instr / waves / steps, "wave_lifetime_quad_cycles": life,
                                  "valu_issue_frac_of_lifetime": c["SQ_ACTIVE_INST_VALU"] / (c["SQ_WAVE_CYCLES"] / waves_per_wg),
                                  "issuing_frac": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                                  "issue_stalled_frac": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                                  "waitcnt_frac": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]},
            "valu_roofline": {"note": "one wave per SIMD (1 024 waves on 1 024 SIMDs): a wave alone issues one VALU instruction per 4 "
                                      "cycles, the SIMD could retire one per 2 with a second wave; frac_of_simd_peak = instructions x 2 "
                                      "cycles / (duration x clock x SIMDs) at 2.4 GHz",
                              "valu_wave_instructions_per_dispatch": instr,
                              # (two-wave workgroups: both waves live for the whole launch, nearly all VALU work is the phase-A wave's)
                              "frac_of_single_wave_issue": c["SQ_ACTIVE_INST_VALU"] / (c["SQ_WAVE_CYCLES"] / waves_per_wg),
                              "frac_of_simd_peak": instr * 2 / simd_cycles}}


def main():
    step, roll, out = sys.argv[1:4]
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from pioneer_amd import _lib
    res = {"_how": HOW, "csrc_sha16": _lib.source_fingerprint(), "kernels": {"dyn_step_kernel<1,1,1,0>": one(step, "dyn_step_kernel", 1, 2),
                                    "dyn_rollout_kernel<1,1,1,0>": one(roll, "dyn_rollout_kernel", 32)}}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res["kernels"].items():
        print(k, round(v["avg_duration_us_under_profiler"], 2), {a: round(b, 4) for a, b in v["per_wave_per_step"].items()})


if __name__ == "__main__":
    sys.exit(main())
