#!/bin/bash
# Runs bench.py (HBM-resident leg only) over a list of argument sets / environment settings and prints one short line each:
#   scripts/bench_matrix.sh "ARGS" ["ENV=VAL ... -- ARGS" ...]
mkdir -p gpurun_out/matrix
for spec in "$@"; do
    envs=""; args="$spec"
    if [[ "$spec" == *" -- "* ]]; then envs="${spec%% -- *}"; args="${spec#* -- }"; fi
    env $envs python bench.py $args --no-cpu-baseline --no-end-to-end > gpurun_out/matrix/last.log 2>&1
    tail -1 gpurun_out/matrix/last.log | python -c "
import json, sys
try:
    d = json.loads(sys.stdin.read())
    print('$spec |', d['value'], 'MiB/s', d['ms_per_step'], 'ms ratio', d['ratio'], d['stages_ms_last_step'])
except Exception as e:
    print('$spec | FAILED', e)
"
done
