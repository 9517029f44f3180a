#!/bin/bash
# Diagnostic (GPU box): shader clock and socket power as rocm-smi reports them while (a) the register-only MFMA probe, (b) one GEMM
# kernel in a loop, (c) the train step run.  usage: tools/clock_under_load.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
one() { rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed -e 's/.*sclk clock level: [0-9S]*: //' -e 's/.*Power (W): / W=/' | tr '\n' ' '; echo; }
sample() { for i in 1 2 3 4 5 6 7 8; do sleep 0.8; one; done; }
echo "== idle"; one
echo "== register-only MFMA probe"; (for k in 1 2 3 4 5 6 7 8; do $R/tools/micro/mfma_rate > /dev/null 2>&1; done) & P=$!; sleep 1.5; sample; wait $P
for shape in "2304 768 qkv(persistent)" "768 3072 dgrad_c_fc(320-row)"; do
  set -- $shape
  echo "== one GEMM in a loop: N=$1 K=$2 $3"
  (timeout -k 10 120 python3 - $1 $2 <<'PY' > /dev/null 2>&1
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from signal_amd import ops
n, k = int(sys.argv[1]), int(sys.argv[2]); M = 24768; Mp = ops.pad_rows(M); dev = torch.device("cuda:0")
a = torch.randn(Mp, k, device=dev).to(torch.bfloat16); w = (torch.randn(n, k, device=dev) * 0.02).to(torch.bfloat16)
out = torch.zeros(Mp, n, device=dev, dtype=torch.bfloat16)
t0 = time.time()
while time.time() - t0 < 14:
    for _ in range(200): ops.gemm_nt(a, w, M, ops.BF16, out)
    torch.cuda.synchronize()
PY
  ) & P=$!; sleep 7; sample; wait $P
done
echo "== train step (bench.py, 1500 steps)"; (timeout -k 10 300 python3 $R/bench.py --steps 1500 --warmup 3 --no-cpu-baseline --no-fwd-sim --no-other-dtype --no-h2d > /dev/null 2>&1) & P=$!; sleep 14; sample; wait $P
