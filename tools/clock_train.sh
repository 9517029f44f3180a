R=${GRAFT_REPO_ROOT:-/root/repo}
one() { rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed -e 's/.*sclk clock level: [0-9S]*: //' -e 's/.*Power (W): / W=/' | tr '\n' ' '; echo; }
(timeout -k 10 300 python3 $R/bench.py --steps 1500 --warmup 3 --no-cpu-baseline --no-fwd-sim --no-other-dtype --no-h2d > $R/gpurun_out/clk_bench.log 2>&1) & P=$!
for i in $(seq 1 40); do sleep 1; echo -n "t=$i "; one; done
wait $P; tail -c 600 $R/gpurun_out/clk_bench.log
