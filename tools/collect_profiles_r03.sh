# Round-3 evidence: bench lines, rocprofv3 kernel trace + stats, PMC passes (HBM traffic, MFMA / LDS counters).
# Run on the GPU box:  bash tools/collect_profiles_r03.sh      (writes gpurun_out/r03/, copy what is judged into profiles/)
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 2>$O/once.err | tail -1 > $O/r03_bench_once.json
python bench.py --steps 10 --warmup 3 --mode iter --no-cpu-baseline 2>/dev/null | tail -1 > $O/r03_bench_iter.json
python bench.py --steps 6 --warmup 2 --cfg 4 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r03_bench_cfg4_unet_batch8.json
python bench.py --steps 6 --warmup 2 --cfg 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r03_bench_cfg5_fp16.json
python bench.py --steps 6 --warmup 2 --precision fp32-mfma --no-cpu-baseline 2>/dev/null | tail -1 > $O/r03_bench_fp32_mfma.json
python tools/stage_bench.py > $O/r03_stage_kernels.txt 2>&1
python tools/ab_flow.py > $O/r03_ab_dataflow.txt 2>&1
python tools/layer_times.py > $O/r03_layer_times.txt 2>&1
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 2 --warmup 1 --frames-per-step 4 --no-cpu-baseline --no-extras --min-warmup-s 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES --output-format csv -d $O/pmc_mfma -o m -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_lds -o l -- $B > /dev/null 2>&1
cd $R
python tools/kmedians.py $O/trace > $O/r03_bench_once_kernel_medians.txt 2>&1
cp $O/trace/t_kernel_stats.csv $O/r03_bench_once_kernel_stats.csv
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/r03_pmc_traffic.json | head -40 > $O/r03_pmc_traffic_top.txt
python tools/pmc_kernel_means.py $O/pmc_mfma > $O/r03_pmc_mfma.txt 2>&1
python tools/pmc_kernel_means.py $O/pmc_lds > $O/r03_pmc_lds.txt 2>&1
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_mfma $O/pmc_lds $O/trace/*trace* 2>/dev/null
ls -la $O
