# Same-call A/B of library builds on bench.py's cfg 3 / cfg 5 / cfg 2 lines (profiles/r04_tile_rows_ab.txt).  The libraries compared are
# builds of EARLIER commits kept beside this script (not in git: tools/probe/*.so is ignored): check the commit out, run
# `python -m yond_public_amd.build`, copy yond_public_amd/libyond_hip.so to tools/probe/libyond_hip_prev.so (the commit before the tile-height
# rule) / libyond_hip_prev2.so (the commit with the 8- / 16-row rule only), come back to HEAD and rebuild.
cd $GRAFT_REPO_ROOT
one() { # lib cfg
  YOND_HIP_LIB=$1 python bench.py --cfg $2 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/ab_tmp.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_tmp.json").read().strip().split("\n")[-1])
print("$1" or "new", "cfg$2", d["config"]["ms_per_frame"], d["conv_stack"]["ms_per_frame"], d["gfx_clock"]["in_kernel_mhz"])
PY
}
for r in 1 2; do
one yond_public_amd/libyond_hip.so 3; one tools/probe/libyond_hip_prev2.so 3; one tools/probe/libyond_hip_prev.so 3
one yond_public_amd/libyond_hip.so 5; one tools/probe/libyond_hip_prev2.so 5
done
one yond_public_amd/libyond_hip.so 2; one tools/probe/libyond_hip_prev2.so 2
