#!/usr/bin/env python3
"""K1 A/B experiments (profiling aid): builds libgomoku_hip variants that differ in eval_kernel.hip's -DGMK_K1_* switches only
(the other objects are the production ones) and times them back to back on the same box.
usage: k1_variants.py build name=FLAG,FLAG ...   (in the container; writes gomokuai_amd/var/libgomoku_hip_<name>.so)
       k1_variants.py run [name ...]             (on the GPU box; prints ms per launch of the 65 536-board batch, 3 rounds interleaved)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VAR = os.path.join(ROOT, "gomokuai_amd", "var")


def build(specs):
    from gomokuai_amd import build as B
    B.build_lib()
    os.makedirs(VAR, exist_ok=True)
    source = os.environ.get("GMK_VARIANT_SOURCE", "eval_kernel.hip")       # (the same harness for another kernel file's -D switches)
    others = [os.path.join(B.CSRC, os.path.splitext(s)[0] + ".o") for s in B.LIB_SOURCES if s != source]
    procs = []
    for spec in specs:
        name, _, flags = spec.partition("=")
        obj = os.path.join(VAR, "%s_%s.o" % (os.path.splitext(source)[0], name))
        cmd = [B.HIPCC] + B.FLAGS + ["-D" + f for f in flags.split(",") if f] + ["-x", "hip", "-c", os.path.join(B.CSRC, source), "-o", obj]
        procs.append((name, obj, subprocess.Popen(cmd)))
    for name, obj, p in procs:
        if p.wait() != 0:
            raise SystemExit("variant %s failed to compile" % name)
        subprocess.check_call([B.HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", os.path.join(VAR, "libgomoku_hip_%s.so" % name), obj] + others)
        print("built", name)


def run(names):
    names = names or sorted(f[len("libgomoku_hip_"):-3] for f in os.listdir(VAR) if f.endswith(".so"))
    results = {n: [] for n in names}
    sums = {}
    for rnd in range(3):
        for n in names:
            env = dict(os.environ, GMK_HIP_LIB=os.path.join(VAR, "libgomoku_hip_%s.so" % n), GMK_EVAL_REPS="200")
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "eval_time.py"), "all", "check"], env=env, capture_output=True, text=True, timeout=300)
            line = [l for l in out.stdout.splitlines() if l.startswith("all")]
            if not line:
                print(n, "FAILED", out.stdout[-300:], out.stderr[-600:]); results[n].append(float("nan")); continue
            results[n].append(float(line[0].split()[1]))
            sums.setdefault(n, set()).update(l.split()[1] for l in out.stdout.splitlines() if l.startswith("checksum"))
    ref = sums.get("base")
    for n in names:
        print("%-24s %s   min %.4f   %s" % (n, "  ".join("%.4f" % x for x in results[n]), min(results[n]),
                                            "" if ref is None else ("outputs = base" if sums.get(n) == ref else "OUTPUTS DIFFER FROM base: %s" % sums.get(n))))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run(sys.argv[2:])
