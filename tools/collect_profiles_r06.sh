# Round-6 evidence: bench lines (headline, cfg 3 grouped, cfg 4 / 5, iter streamed, fp32-MFMA), rocprofv3 kernel trace + stats of the headline and of cfg 5,
# PMC passes (HBM traffic, MFMA counters) for both, stage / layer timings (fp32 flow, h-only flow, cfg 3's batch), the training step, the evaluation driver
# with groups of images.
# Run on the GPU box:  bash tools/collect_profiles_r06.sh      (writes gpurun_out/r06/, copy what is judged into profiles/)
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 2>$O/once.err | tail -1 > $O/r06_bench_once.json
python bench.py --cfg 3 --steps 6 --warmup 2 2>/dev/null | tail -1 > $O/r06_bench_cfg3_sidd.json
python bench.py --steps 10 --warmup 3 --mode iter --no-cpu-baseline 2>/dev/null | tail -1 > $O/r06_bench_iter.json
python bench.py --steps 6 --warmup 2 --cfg 4 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r06_bench_cfg4_unet_batch8.json
python bench.py --steps 6 --warmup 2 --cfg 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r06_bench_cfg5_fp16.json
python bench.py --steps 6 --warmup 2 --precision fp32-mfma --no-cpu-baseline 2>/dev/null | tail -1 > $O/r06_bench_fp32_mfma.json
python tools/stage_bench.py > $O/r06_stage_kernels.txt 2>&1
python tools/layer_times.py > $O/r06_layer_times.txt 2>&1
rm -f gpurun_out/lanes_ab.txt; bash tools/lanes_ab.sh "1:0 2:0" once; bash tools/lanes_ab.sh "1:0 2:0" cfg5 --cfg 5; cp gpurun_out/lanes_ab.txt $O/r06_lanes_ab.txt
python tools/layer_times.py 32 128 128 > $O/r06_layer_times_cfg3_batch32.txt 2>&1
python tools/layer_times.py 128 128 128 > $O/r06_layer_times_cfg3_group4_batch128.txt 2>&1
PRECISION=fp16 python tools/layer_times.py 1 2016 3008 > $O/r06_layer_times_cfg5_fp16.txt 2>&1
python tools/train_bench.py --steps 30 > $O/r06_train_bench.txt 2>&1
python YOND_SIDD.py --synthetic 40 2>&1 | grep -E "images on|steady|Iter" > $O/r06_eval_driver_synthetic40_group4.txt
python YOND_SIDD.py --synthetic 40 --group 1 2>&1 | grep -E "images on|steady|Iter" > $O/r06_eval_driver_synthetic40_group1.txt
cd /tmp && export TMPDIR=/tmp
# (second session: the stream drivers run the network passes of consecutive frames on two lanes; a launch's wall duration then includes its wait for the other
#  lane's workgroups, so the trace is taken twice -- the default command, and --lanes 1 for the kernels' own durations, which the roofline objects quote -- and the
#  counter passes, which serialise the launches anyway, run on one lane)
B="python3 $R/bench.py --steps 2 --warmup 1 --frames-per-step 4 --no-cpu-baseline --no-extras --min-warmup-s 0"
B5="python3 $R/bench.py --cfg 5 --steps 2 --warmup 1 --frames-per-step 4 --no-cpu-baseline --no-extras --min-warmup-s 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace1 -o t -- $B --lanes 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- $B --lanes 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- $B --lanes 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES --output-format csv -d $O/pmc_mfma -o m -- $B --lanes 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace5 -o t -- $B5 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace51 -o t -- $B5 --lanes 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch5 -o f -- $B5 --lanes 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write5 -o w -- $B5 --lanes 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES --output-format csv -d $O/pmc_mfma5 -o m -- $B5 --lanes 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_trace -o t -- python3 $R/tools/train_bench.py --steps 4 > $O/train_trace_bench.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/frame_trace -o f -- python3 $R/tools/frame_trace.py > /dev/null 2>&1
cd $R
python tools/kmedians.py $O/trace1 > $O/r06_bench_once_kernel_medians.txt 2>&1
python tools/kmedians.py $O/trace51 > $O/r06_bench_cfg5_kernel_medians.txt 2>&1
python tools/frame_trace_summary.py $O/frame_trace > $O/r06_frame_launches.txt 2>&1
cp $O/trace/t_kernel_stats.csv $O/r06_bench_once_kernel_stats.csv
cp $O/trace5/t_kernel_stats.csv $O/r06_bench_cfg5_kernel_stats.csv
cp $O/trace1/t_kernel_stats.csv $O/r06_bench_once_one_lane_kernel_stats.csv
cp $O/trace51/t_kernel_stats.csv $O/r06_bench_cfg5_one_lane_kernel_stats.csv
cp $O/train_trace/t_kernel_stats.csv $O/r06_train_step_kernel_stats.csv
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/r06_pmc_traffic.json | head -40 > $O/r06_pmc_traffic_top.txt
python tools/pmc_traffic.py $O/pmc_fetch5 $O/pmc_write5 $O/r06_pmc_traffic_cfg5.json | head -40 > $O/r06_pmc_traffic_cfg5_top.txt
python tools/pmc_kernel_means.py $O/pmc_mfma > $O/r06_pmc_mfma.txt 2>&1
python tools/pmc_kernel_means.py $O/pmc_mfma5 > $O/r06_pmc_mfma_cfg5.txt 2>&1
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_mfma $O/pmc_fetch5 $O/pmc_write5 $O/pmc_mfma5 $O/frame_trace $O/trace/*trace* $O/trace5/*trace* $O/trace1/*trace* $O/trace51/*trace* $O/train_trace/*trace* 2>/dev/null
ls -la $O
