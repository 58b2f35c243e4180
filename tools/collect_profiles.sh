set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd $R
python bench.py --steps 10 --warmup 3 2>$O/once.err | tail -1 > $O/r01_bench_once.json
python bench.py --steps 6 --warmup 2 --mode iter --no-cpu-baseline 2>/dev/null | tail -1 > $O/r01_bench_iter.json
python bench.py --steps 6 --warmup 2 --arch UNetSeeInDark --no-cpu-baseline 2>/dev/null | tail -1 > $O/r01_bench_unetseeindark.json
python bench.py --steps 6 --warmup 2 --precision fp16 --height 4000 --width 6000 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r01_bench_cfg5_fp16.json
python bench.py --steps 6 --warmup 2 --precision fp16 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r01_bench_fp16_3000x4000.json
python bench.py --steps 6 --warmup 2 --precision fp32-mfma --no-cpu-baseline 2>/dev/null | tail -1 > $O/r01_bench_fp32_mfma.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --precision fp32 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $R
python tools/kmedians.py $O/trace > $O/r01_bench_once_kernel_medians.txt 2>&1
cp $O/trace/t_kernel_stats.csv $O/r01_bench_once_kernel_stats.csv
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/r01_pmc_traffic.json | head -14
rm -rf $O/pmc_fetch/*trace* $O/pmc_write/*trace* 2>/dev/null
ls -la $O
