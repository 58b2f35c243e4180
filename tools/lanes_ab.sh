#!/bin/bash
# A/B of pipeline.STREAM_LANES (1 = every network pass on the main stream, 2 = two lanes) and LANE_PIPELINES on the bench's drivers; prints value / ms per step / clock.
# usage: tools/lanes_ab.sh "<lanes>:<lane-pipelines> ..." tag [bench args]   e.g.  tools/lanes_ab.sh "1:0 2:0 2:1" iter --mode iter
set -e
out=gpurun_out/lanes_ab.txt
variants="$1"; tag="$2"; shift; shift
for rep in 1 2; do
  for v in $variants; do
    l=${v%%:*}; lp=${v##*:}
    python bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline --lanes $l --lane-pipelines $lp "$@" > gpurun_out/_lanes.json 2> gpurun_out/_lanes.err
    python - "$tag" $v >> $out <<'PY'
import json, sys
d = json.loads(open("gpurun_out/_lanes.json").read().strip().splitlines()[-1])
print(sys.argv[1], "lanes:pipelines", sys.argv[2], d["value"], d["unit"], d["ms_per_step"], "ms/step", d.get("gfx_clock", {}).get("hwmon_mhz"), "MHz")
PY
  done
done
