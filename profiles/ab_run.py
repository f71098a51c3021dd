#!/usr/bin/env python3
"""A/B on ONE GPU box: bench.py for each (label, library build, environment) of a spec, alternating, `rounds` times.

usage: profiles/ab_run.py <out.jsonl> <rounds> <steps> <warmup> label=LIB[,VAR=val,...] ...
  LIB = "default" (round 6 on: tfrec_amd/libtfrec_amd_exp.so, the build that reads the TFREC_AMD_* knobs -- the product library
  has none, csrc/knobs.h; "product" = tfrec_amd/libtfrec_amd.so, no VAR allowed) or the name of a build under tfrec_amd/ab/
  (profiles/build_variant.sh, which compiles with -DTFREC_AMD_EXPERIMENTS).
One line per run on stdout (ms_per_step, steady, min/median, parity, the longest kernels), the full bench lines in out.jsonl.
"""
import json
import os
import subprocess
import sys

R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out_path, rounds, steps, warmup = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
specs = []
for a in sys.argv[5:]:
    label, rest = a.split("=", 1)
    parts = rest.split(",")
    lib = (None if parts[0] == "product" else os.path.join(R, "tfrec_amd", "libtfrec_amd_exp.so") if parts[0] == "default"
           else os.path.join(R, "tfrec_amd", "ab", parts[0] + ".so"))
    env = dict(p.split("=", 1) for p in parts[1:])
    specs.append((label, lib, env))
extra = os.environ.get("AB_BENCH_ARGS", "--cpu-budget 0 --h2d-steps 0 --parity-streams 8 --no-extra-configs").split()
with open(out_path, "a") as fo:
    for r in range(rounds):
        for label, lib, env in specs:
            e = dict(os.environ, **env) if lib is None else dict(os.environ, TFREC_AMD_LIB=lib, **env)
            p = subprocess.run([sys.executable, os.path.join(R, "bench.py"), "--steps", str(steps), "--warmup", str(warmup)] + extra
                               + ([] if lib is None else ["--experiments"]),
                               env=e, capture_output=True, text=True)
            try:
                j = json.loads(p.stdout.strip().splitlines()[-1])
            except Exception:
                print("%-28s FAILED rc=%d %s" % (label, p.returncode, (p.stderr or "")[-300:].replace("\n", " | ")), flush=True)
                continue
            j["_label"] = label
            fo.write(json.dumps(j) + "\n")
            fo.flush()
            k = j["roofline"]["kernels_ms"]
            top = sorted(k.items(), key=lambda kv: -kv[1])[:8]
            print("%-28s %7.3f ms/step steady %s (min %.2f med %.2f) parity %s/%s  %s" % (
                label, j["ms_per_step"], j.get("ms_per_step_steady"), j["ms_min"], j["ms_median"], j["config"]["parity_ok"],
                j["config"]["parity_after_timed"], " ".join("%s=%.2f" % (a.replace("_kernel", ""), b) for a, b in top)), flush=True)
