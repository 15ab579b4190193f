#!/usr/bin/env python3
"""A/B of scan-kernel build variants (tools/_variants/*.so) in interleaved rounds on one box:
each variant runs in its own subprocess (one library per process), several rounds, median reported."""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sorted(glob.glob(os.path.join(ROOT, "tools", "_variants", "*.so")))
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
shape = sys.argv[2:4] if len(sys.argv) > 3 else ["256", "22950458"]
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, PSK_LIB=l)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "matrix", "--samples", shape[0],
                              "--rows", shape[1], "--steps", "200", "--warmup", "20", "--no-cpu-baseline"], env=env,
                             capture_output=True, text=True).stdout.strip().splitlines()[-1]
        d = json.loads(out)
        res[l].append((d["roofline"]["kernel_ms"], d["ms_per_step"]))
for l in libs:
    ks = sorted(k for k, _ in res[l]); ss = sorted(s for _, s in res[l])
    print("%-28s kernel ms median %.4f min %.4f | step ms median %.4f" % (os.path.basename(l), ks[len(ks)//2], ks[0], ss[len(ss)//2]))
