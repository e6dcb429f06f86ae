"""A/B the whole step between builds of the library on ONE box (boxes differ by ~1 %).

    python tools/ab_step.py build tag:-DTPG_X=1,-DTPG_Y=2 ...     # here: csrc/variants/libtpgan_hip_<tag>.so
    python tools/ab_step.py run [rounds] tag ...                   # GPU box: default build vs each tag, interleaved

Kernel variants that win alone can lose in the step (its three branches share the chip), so launch
shapes are decided on this measurement, not on tools/tune_*.py alone.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "temporal-pointcloud-upsampling-gan_amd", "csrc", "variants")


def lib(tag):
    return os.path.join(VDIR, f"libtpgan_hip_{tag}.so")


def build(specs):
    import importlib.util
    spec = importlib.util.spec_from_file_location("tb", os.path.join(ROOT, "temporal-pointcloud-upsampling-gan_amd", "build.py"))
    tb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tb)
    os.makedirs(VDIR, exist_ok=True)
    for s in specs:
        tag, _, defs = s.partition(":")
        cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + tb.HIPCC_FLAGS + [d for d in defs.split(",") if d] + \
              ["-I", tb.INCLUDE, "-o", lib(tag)] + tb.sources()
        subprocess.check_call(cmd)
        print("built", lib(tag))


def run(rounds, tags):
    res = {t: [] for t in ["default"] + tags}
    for _ in range(rounds):
        for t in res:
            env = dict(os.environ)
            if t != "default":
                env["TPGAN_HIP_LIBRARY"] = lib(t)
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "3", "--no-extra"],
                                 env=env, capture_output=True, text=True, check=True).stdout
            res[t].append(json.loads(out.strip().splitlines()[-1])["ms_per_step"])
            print(t, "%.3f ms" % res[t][-1], flush=True)
    for t, v in res.items():
        print("%-16s min %.3f  median %.3f ms/step" % (t, min(v), sorted(v)[len(v) // 2]))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        n = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 3
        run(n, [a for a in sys.argv[2:] if not a.isdigit()])
