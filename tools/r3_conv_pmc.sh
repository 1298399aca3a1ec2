#!/bin/bash
# PMC passes over the forward's convolution kernels (run ON the GPU box from the repo root): bash tools/r3_conv_pmc.sh <out dir under gpurun_out>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r3convpmc}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify"
rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d $O/p1 -o c -- $B > /dev/null 2> $O/p1.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA --output-format csv -d $O/p2 -o c -- $B > /dev/null 2> $O/p2.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/p3 -o c -- $B > /dev/null 2> $O/p3.err
python3 $R/tools/pmc_summary.py $O/r03_pmc_conv_kernels.json "k_conv3x3_halo<128, 0, 7, -1, 1>;k_conv3x3_halo<128, 0, 6, -1, 1>;k_conv3x3_halo<128, 0, 6, 0, 1>;k_conv3x3_halo<128, 0, 7, -1, 4>;k_pw<8, 2, 0, false>;k_pw<4, 2, 1, false>" $O/p1 $O/p2 $O/p3 > $O/summary.out 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
grep "^k_\|_derived" $O/summary.out | cut -c1-260
