#!/bin/bash
# round-3 probe B (GPU box, repo root): whole GPU suite + the new bench line
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/probe_b
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee $O/rc.txt
tail -25 $O/pytest_gpu.log
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
tail -3 $O/bench.err; cat $O/bench.json
