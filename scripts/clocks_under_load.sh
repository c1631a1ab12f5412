#!/bin/bash
# What the GPU's clocks and power are WHILE the headline frames render (rocm-smi sampled beside a 60-step bench.py run):
# usage (GPU box): bash scripts/clocks_under_load.sh [workload]
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
W=${1:-dragon}
(python bench.py --workload $W --steps 60 --warmup 3 --no-pmc --cpu-tiles 0 --no-e2e > /tmp/clk_bench.json 2>/dev/null) &
BP=$!
sleep ${2:-25}
for i in $(seq 1 ${3:-12}); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk|fclk" | sed 's/^GPU\[0\]\s*: //' | tr '\n' '|'; echo; sleep 0.4; done
wait $BP
python -c "import json;d=json.loads(open('/tmp/clk_bench.json').read().strip().splitlines()[-1]);print('ms_per_step', d['ms_per_step'])"
echo "idle:"; rocm-smi --showclocks --showpower --showmaxpower 2>/dev/null | grep -E "sclk|Power" | head -6
