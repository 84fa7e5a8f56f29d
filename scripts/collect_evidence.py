#!/usr/bin/env python3
"""gpurun_out/evidence_<tag>/ (scripts/evidence_round.sh on the GPU box) -> profiles/<tag>_* : every figure of a round from ONE
build and ONE session, with the hashes of that build beside them.

    python scripts/collect_evidence.py r5
"""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r5"
src = os.path.join(ROOT, "gpurun_out", f"evidence_{tag}")
dst = os.path.join(ROOT, "profiles")
copied = []
for name in sorted(os.listdir(src)):
    path = os.path.join(src, name)
    if not os.path.isfile(path) or name.endswith(".err") or os.path.getsize(path) == 0:
        continue
    if name.endswith(".json"):
        try:
            json.load(open(path))
        except Exception as e:                      # a line that did not parse is not evidence
            print(f"skipped {name}: {e}")
            continue
    shutil.copy(path, os.path.join(dst, f"{tag}_{name}"))
    copied.append(name)
print("copied:", ", ".join(copied))
for script in ("summarize_profile.py", "summarize_g3.py"):
    if os.path.isdir(os.path.join(ROOT, "gpurun_out", f"prof_{tag}" if script == "summarize_profile.py" else f"prof_g3_{tag}")):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script), tag], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        print(script, "ok" if r.returncode == 0 else "FAILED\n" + r.stdout[-800:])
