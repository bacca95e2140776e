#!/bin/bash
# PMC passes (one counter group per run, never combined with tracing) over the split pipeline's kernels.
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
CASE=/tmp/c512.npz
[ -f $CASE ] || python bench.py --steps 1 --warmup 0 --no-cpu --case-cache $CASE > /dev/null 2>&1
B="python bench.py --steps 2 --warmup 1 --no-cpu --pipeline ${PIPE:-2} --graph 0 --case-cache $CASE"
pass() { local name=$1; shift
    rm -rf /tmp/prof/$name
    timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d /tmp/prof/$name -- $B > /tmp/$name.out 2>&1
    echo "## $name rc=$?"
    python tools/prof_summary.py /tmp/prof/$name k_s2m_cert k_s2m_scan k_s2m_fit k_s2m_iterate | grep -v "^#" | sort
}
pass SQ SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
pass SQ2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE GRBM_TA_BUSY
pass TCC TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass TCP TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
