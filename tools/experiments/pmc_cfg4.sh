#!/bin/bash
# Memory-side counters of the path kernel on the 2048^3 path-trace workload (one pass per group).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_cfg4
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CHILD="python $ROOT/bench.py --pmc-child --workload cfg4_4k_2048c_b8_sparse --variant ${1:-0} --pmc-frames 1"
i=0
for PMC in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_UTCL2_BUSY TCC_EA0_RDREQ_32B_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC -d $OUT/p$i -o pmc -- $CHILD > $OUT/p$i.log 2>&1
done
cd $ROOT
python - <<PY
import glob, sqlite3, os
for db in sorted(glob.glob("$OUT/p*/**/*.db", recursive=True)):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    pmc = [t for t in tabs if t.startswith("rocpd_pmc_event")]
    inf = [t for t in tabs if t.startswith("rocpd_info_pmc")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")]
    if not pmc: print(db, "no pmc"); continue
    q = f"""select s.kernel_name, i.name, sum(e.value), count(distinct d.id) from {pmc[0]} e join {inf[0]} i on e.pmc_id=i.id
            join {kd[0]} d on e.event_id=d.event_id join {ks[0]} s on d.kernel_id=s.id group by s.kernel_name, i.name"""
    for name, ctr, val, n in c.execute(q):
        if "path_kernel" in name: print(f"{ctr:40s} per dispatch {val/n:.6g}  ({n} dispatches)")
PY
rm -rf $OUT/p[0-9]
