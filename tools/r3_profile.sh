#!/bin/bash
# Round-3 profile set, run ON the GPU box from the repo root: bash tools/r3_profile.sh <out dir under gpurun_out>
# Every rocprofv3 run profiles `python3 bench.py ...` directly (no wrapper), CPU baseline and the independent re-computation off.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r3prof}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PP="python3 $R/bench.py --postproc-only --batch 128 --steps 20 --warmup 3 --no-cpu-baseline --no-verify"
echo "== postproc kernel stats"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/pp_stats -o pp -- $PP > $O/bench_postproc_b128_under_rocprof.json 2> $O/pp_stats.err
echo "== pmc passes"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pp -- $PP > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o pp -- $PP > /dev/null 2> $O/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d $O/pmc_sq1 -o pp -- $PP > /dev/null 2> $O/pmc_sq1.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_sq2 -o pp -- $PP > /dev/null 2> $O/pmc_sq2.err
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_sq3 -o pp -- $PP > /dev/null 2> $O/pmc_sq3.err
python3 $R/tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write 128 r03_pmc_traffic.json > $O/pmc_traffic.out 2>&1; cat $O/pmc_traffic.out
python3 $R/tools/pmc_summary.py $O/r03_pmc_postproc_kernels.json "k_heat_peaks,k_limb_connect<" $O/pmc_sq1 $O/pmc_sq2 $O/pmc_sq3 > $O/pmc_summary.out 2>&1; tail -3 $O/pmc_summary.out
cp $R/profiles/r03_pmc_traffic.json $O/ 2>/dev/null
echo "== e2e steady step"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/e2e -o e2e -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-verify > $O/bench_default_b128_under_rocprof.json 2> $O/e2e.err
python3 $R/tools/trace_summary.py $O/e2e $O/r03_e2e_b128_steady_step_kernels.csv
cp $(find $O/e2e -name "*kernel_stats.csv" | head -1) $O/r03_e2e_b128_kernel_stats.csv 2>/dev/null
cp $(find $O/pp_stats -name "*kernel_stats.csv" | head -1) $O/r03_postproc_b128_kernel_stats.csv 2>/dev/null
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
echo "== multiscale steady step"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/ms -o ms -- python3 $R/bench.py --multiscale --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_multiscale_b32_under_rocprof.json 2> $O/ms.err
python3 $R/tools/trace_summary.py $O/ms $O/r03_multiscale_b32_steady_step_kernels.csv --anchor=k_fullres_peaks --delete-trace
cp $(find $O/ms -name "*kernel_stats.csv" | head -1) $O/r03_multiscale_b32_kernel_stats.csv 2>/dev/null
cd $R
echo "== bench default (cpu baseline on)"; python3 bench.py --steps 20 --warmup 3 > $O/bench_default_b128.json 2> $O/bench_default.err; echo rc=$?
echo "== bench multiscale"; python3 bench.py --multiscale --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_multiscale_b32.json 2> $O/bench_ms.err; echo rc=$?
ls $O | head -40
