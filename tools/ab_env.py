"""A/B the whole step between environment settings on ONE box (boxes differ by a few per cent), interleaved.

    python tools/ab_env.py [rounds] [--config cfg2] [--steps 100] NAME=VAL[,NAME2=VAL2] ...

`default` (the environment as it is) against each setting; prints every run and min / median per setting.  GPU box.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(argv):
    rounds, config, steps, specs = 3, "cfg2", "100", []
    i = 0
    while i < len(argv):
        a = argv[i]
        if a.isdigit():
            rounds = int(a)
        elif a == "--config":
            i += 1
            config = argv[i]
        elif a == "--steps":
            i += 1
            steps = argv[i]
        else:
            specs.append(a)
        i += 1
    res = {t: [] for t in ["default"] + specs}
    for _ in range(rounds):
        for t in res:
            env = dict(os.environ)
            if t != "default":
                for kv in t.split(","):
                    k, _, v = kv.partition("=")
                    env[k] = v
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--steps", steps,
                                  "--warmup", "3", "--no-extra"], env=env, capture_output=True, text=True)
            if out.returncode != 0:
                print(t, "FAILED", out.stderr[-2000:], flush=True)
                res[t].append(float("nan"))
                continue
            res[t].append(json.loads(out.stdout.strip().splitlines()[-1])["ms_per_step"])
            print(t, "%.3f ms" % res[t][-1], flush=True)
    for t, v in res.items():
        v = [x for x in v if x == x]
        if v:
            print("%-40s min %.3f  median %.3f ms/step" % (t, min(v), sorted(v)[len(v) // 2]))


if __name__ == "__main__":
    main(sys.argv[1:])
