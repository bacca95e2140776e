#!/bin/bash
# Profiling session behind profiles/r02_*: the bench's timed workload under rocprofv3, one pass per counter group
# (kernel trace and PMC are never combined).  Run on the GPU box from the repo root:
#     bash tools/profile_round.sh
# Summaries land in gpurun_out/prof_r02/; copy the ones to be kept into profiles/.
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/prof_r02
mkdir -p $OUT
CASE=/tmp/case8.npz
python bench.py --steps 1 --warmup 1 --no-cpu --no-extras --case-cache $CASE > /dev/null 2> $OUT/gen.log   # writes the case cache
# the same timed workload as the default `python bench.py` (8 distinct batches streamed, double-buffered), without the CPU
# baseline and the secondary measurements, which would mix other launches of the same kernels into the averages
B="python bench.py --steps 16 --warmup 4 --no-cpu --no-extras --roofline-pass-only --case-cache $CASE"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/kt -o runc -- $B > $OUT/bench_under_rocprof.json 2> /tmp/kt.err
python tools/prof_summary.py /tmp/prof/kt k_ > $OUT/rocprofv3_kernel_stats.txt
pass() { # name, counters...
    local name=$1; shift
    rm -rf /tmp/prof/$name
    timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d /tmp/prof/$name -- $B > /tmp/$name.out 2>&1
    echo "$name rc=$?"
    python tools/prof_summary.py /tmp/prof/$name k_s2m k_scan > $OUT/pmc_$name.txt
}
pass FETCH_SIZE FETCH_SIZE
pass WRITE_SIZE WRITE_SIZE
pass SQ SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
pass TCC TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
pass SQ2 SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE GRBM_TA_BUSY
# (single registrations: tools/latency_sweep.py, tools/register_trace.py, tools/persist_clock.py, tools/small_batch_sweep.py)
grep -h "k_s2m_iterate" $OUT/rocprofv3_kernel_stats.txt $OUT/pmc_*.txt | cut -c1-200
