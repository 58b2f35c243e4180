# counter passes over the box-statistics kernels alone (run on the GPU box): bash tools/box_pmc.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/boxpmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/tools/${1:-box_only.py} 4"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/p1 -o a -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $O/p2 -o b -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/p3 -o c -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/p4 -o d -- $B > /dev/null 2>&1
cd $R
for p in p1 p2 p3 p4; do python tools/pmc_kernel_means.py $O/$p; done > $O/summary.txt 2>&1
rm -rf $O/p1 $O/p2 $O/p3 $O/p4
cat $O/summary.txt
