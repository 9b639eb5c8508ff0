#!/bin/bash
# Developer loop on the GPU box: parity tests, then bench lines.  usage: tools/quick_gpu.sh <tag> [pytest -k expr]
set -o pipefail
TAG=$1; K=${2:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
if [ -n "$K" ]; then
  timeout -k 10 500 python -m pytest tests -x -q -m gpu -k "$K" > gpurun_out/q_${TAG}_pytest.log 2>&1 || { tail -30 gpurun_out/q_${TAG}_pytest.log; exit 1; }
else
  timeout -k 10 500 python -m pytest tests -x -q -m gpu > gpurun_out/q_${TAG}_pytest.log 2>&1 || { tail -30 gpurun_out/q_${TAG}_pytest.log; exit 1; }
fi
tail -2 gpurun_out/q_${TAG}_pytest.log
: > gpurun_out/q_${TAG}_bench.jsonl
for args in "--steps 30" "--steps 30 --workload hd1080" "--steps 300 --batch 1" "--steps 100 --batch 64"; do
  timeout -k 10 200 python bench.py --no-cpu-baseline $args 2>/dev/null | tail -1 >> gpurun_out/q_${TAG}_bench.jsonl || exit 1
done
python - <<PY
import json
for l in open("gpurun_out/q_${TAG}_bench.jsonl"):
    d = json.loads(l)
    print(d["config"]["workload"].split(":")[0], d["config"]["frames_per_gpu_per_step"], d["value"], d["ms_per_step"],
          {k: round(v * 1e3) for k, v in d["roofline"]["kernel_ms_per_step"].items()})
PY
