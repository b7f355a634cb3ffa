"""The C oracle under UndefinedBehaviorSanitizer + AddressSanitizer-free bounds (CPU only; GPU ASan is
not available on the pool).  Builds a -fsanitize=undefined,bounds -fno-sanitize-recover copy of the
oracle and drives every entry point; any UB aborts the child process."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_under_ubsan(tmp_path):
    so = tmp_path / "libpnr_oracle_ubsan.so"
    src = [os.path.join(ROOT, "oracle", f) for f in ("pnr_oracle.c", "pnr_dyn_oracle.c")]
    subprocess.run(["gcc", "-O1", "-g", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off",
                    "-fsanitize=undefined,bounds,float-divide-by-zero", "-fno-sanitize-recover=all",
                    "-o", str(so), *src, "-lm"], check=True)
    code = textwrap.dedent(f"""
        import sys, numpy as np
        sys.path.insert(0, {ROOT!r})
        import oracle.binding as B
        B.oracle_lib_path = lambda: {str(so)!r}
        B.build_oracle = lambda force=False: {str(so)!r}
        o = B.COracle(513, seed=1, precision=B.ORC_DEV, auto_reset=True, max_episode_steps=7, nthreads=3)
        o.reset()
        rng = np.random.RandomState(0)
        for t in range(30):
            o.step((rng.uniform(-3, 3, (513, 6)) * o.a_max).astype(np.float32), want_info=True)
        o.load_state_words(o.state_words()); o.observe(); o.fk([o.r_hi]); o.philox([1, 2, 3, 4], [5, 6])
        d = B.DynOracle(65, seed=2, auto_reset=True, max_episode_steps=5, nthreads=2,
                        dyn=dict(gravity=9.81, randomize=1, ground_z=3.0, torque_limit=900.0))
        d.reset()
        for t in range(12):
            d.step((rng.uniform(-0.5, 0.5, (65, 6)) * d.a_max).astype(np.float32))
        d.energy(9.81); d.tip(); d.load_dyn_words(d.dyn_words())
        print("ubsan-clean")
    """)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ubsan-clean" in out.stdout, out.stderr[-2000:]
    assert "runtime error" not in out.stderr
