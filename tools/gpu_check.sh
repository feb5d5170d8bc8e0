#!/bin/bash
# On the GPU box: the -m gpu suite, a kernel-trace summary of the bench, and two bench lines.  usage: bash tools/gpu_check.sh <tag>
tag=${1:-chk}
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/${tag}_pytest.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_stats.log 2>&1
python tools/kstats.py gpurun_out/${tag}_stats | head -${2:-14}
for i in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench.log 2> gpurun_out/${tag}_bench.err
  python -c "
import json; d=json.load(open('gpurun_out/${tag}_bench.log')); print(round(d['ms_per_step'],4), round(d['device_ms_per_step'],4), {k[3:]: round(v,4) for k,v in d['stage_ms'].items()})"
done
