# counter passes over the weight-gradient kernel alone (run on the GPU box): bash tools/wgrad_pmc.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wgpmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/tools/wgrad_bench.py"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/p1 -o a -- $B > /dev/null 2>&1
echo pass1 done
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD --output-format csv -d $O/p2 -o b -- $B > /dev/null 2>&1
echo pass2 done
cd $R
for p in p1 p2; do python3 tools/pmc_kernel_means.py $O/$p; done > $O/summary.txt 2>&1
rm -rf $O/p1 $O/p2
cat $O/summary.txt
