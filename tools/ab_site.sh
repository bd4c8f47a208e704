#!/bin/bash
# usage: tools/ab_site.sh ENVVAR "v1 v2 ..." site  -- bench the default workload for each value of an env knob
var=$1; vals=$2; site=${3:-conv2}
for v in $vals; do
  env $var=$v python bench.py --no-cpu --steps 8 --warmup 2 --also-per-chunk 0 --site $site 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$var=$v', d['ms_per_step'], r['avg_launch_us'], r['achieved'])"
done
