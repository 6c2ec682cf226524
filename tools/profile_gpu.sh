#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel trace + stats of bench.py, then the PMC passes
# (FETCH_SIZE and WRITE_SIZE in SEPARATE runs, never combined with trace domains other than
# --kernel-trace) on tools/pmc_target.py.  Everything lands under gpurun_out/prof/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_bench -o bench -- \
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace_bench.err
# HBM traffic of the SAME launches (the score_mask_k launches inside bench.py's steps): separate PMC passes
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_bench_fetch -o fetch -- \
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/pmc_bench_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_bench_write -o write -- \
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/pmc_bench_write.err
# fp64-VALU fraction / cull / dense kernel figures (stand-alone launches, not under the profiler)
python bench.py --kernel-detail --no-cpu-baseline > $OUT/bench_kernel_detail.json 2> $OUT/bench_kernel_detail.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_target -o target -- \
    python tools/pmc_target.py > $OUT/target.out 2> $OUT/trace_target.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- \
    python tools/pmc_target.py > /dev/null 2> $OUT/pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- \
    python tools/pmc_target.py > /dev/null 2> $OUT/pmc_write.err
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -o sq -- \
    python tools/pmc_target.py > /dev/null 2> $OUT/pmc_sq.err
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $OUT/pmc_grbm -o grbm -- \
    python tools/pmc_target.py > /dev/null 2> $OUT/pmc_grbm.err
find $OUT -name "*.csv" | head -50
du -sh $OUT
