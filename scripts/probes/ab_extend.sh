#!/bin/bash
# A/B of the two forms of the extension kernel on the bench workload, same box: scripts/probes/ab_extend.sh
for v in records queries records; do
    if [ $v = queries ]; then export CDM_EXTEND=queries; else unset CDM_EXTEND; fi
    python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v', round(d['ms_per_step'],1), d['config']['stage_kernel_ms']['extend'])"
done
