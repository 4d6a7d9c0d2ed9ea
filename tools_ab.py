"""Same-box A/B: runs bench.py variants back to back, several rounds, prints medians."""
import subprocess, sys, json, os, statistics
variants = [("base", {}), ("no_short", {"GPSLC_EXP": "1"}), ("no_diagskip", {"GPSLC_EXP": "2"}), ("gram_nonpersist", {"GPSLC_EXP": "4"}), ("all_off", {"GPSLC_EXP": "7"})]
res = {k: [] for k, _ in variants}
mf = {k: [] for k, _ in variants}
for rnd in range(3):
    for name, env in variants:
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--samples-per-step", "512", "--no-cpu-baseline", "--streams", "1"], env=e, capture_output=True, text=True).stdout
        for l in out.splitlines():
            if l.startswith("{"):
                d = json.loads(l); res[name].append(d["value"]); mf[name].append(d["roofline"]["achieved"])
for name, _ in variants:
    print(f"{name:16s} samples/s median {statistics.median(res[name]):8.1f}  all {['%.0f' % v for v in res[name]]}  mfma TF {statistics.median(mf[name]):.1f}")
