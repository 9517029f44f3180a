#!/usr/bin/env python3
"""Print VGPR/AGPR/spill/LDS/occupancy per kernel of signal_amd/csrc/*.hip (hipcc -Rpass-analysis)."""
import os, re, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "signal_amd", "csrc")
sys.path.insert(0, CSRC)
import build as B  # noqa
pat = re.compile(r"remark: (?:[^:]+:\d+:\d+: )?\s*(.+?) \[-Rpass-analysis")
for src in (sys.argv[1:] or B.sources()):
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run([B._hipcc(), *B.FLAGS, "-c", os.path.join(CSRC, src), "-o", os.path.join(d, "o.o"),
                            "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    cur = {}
    for line in r.stderr.splitlines():
        m = pat.search(line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            if cur: print(cur)
            cur = {"k": subprocess.run(["c++filt", t.split(": ")[1]], capture_output=True, text=True).stdout.strip()[:60]}
        elif ":" in t:
            k, v = t.split(":", 1)
            if k.strip() in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "VGPRs Spill", "LDS Size [bytes/block]", "Occupancy [waves/SIMD]", "SGPRs"):
                cur[k.strip().replace(" ", "_").split("_[")[0]] = v.strip()
    if cur: print(cur)
