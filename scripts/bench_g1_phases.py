"""Where the G1 kernel's time goes: step time with parts of the forward evaluation switched off (DMG1_SKIP bits)."""
import os
import subprocess
import sys

n = sys.argv[1] if len(sys.argv) > 1 else "4096"
for name, bits in (("full", 0), ("no MPR pairs", 64), ("no MPR, no plane-mesh", 64 | 128), ("broadphase only", 32), ("no PGS sweeps", 1), ("no A matrix, no sweeps", 5), ("no constraint solve at all", 16 | 5),
                   ("no rows, no solve", 8 | 16 | 5), ("no collision either", 2 | 8 | 16 | 5)):
    env = dict(os.environ, DMG1_SKIP=str(bits))
    r = subprocess.run([sys.executable, "scripts/bench_g1.py", n, "20"], env=env, capture_output=True, text=True)
    print("%-28s %s" % (name, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]))
