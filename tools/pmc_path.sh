#!/bin/bash
# Counters of the kernel that renders a workload's bounce frames, one rocprofv3 --pmc pass per group.
# usage: tools/pmc_path.sh <tuning_flags> [workload] [variant] [tag]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
FLAGS=${1:-0}; WL=${2:-cfg4_4k_2048c_b8_sparse}; VAR=${3:-0}; TAG=${4:-flags$FLAGS}
OUT=$ROOT/gpurun_out/pmc_path_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CHILD="python $ROOT/tools/pmc_child.py $WL $VAR $FLAGS 2 V0"
i=0
for PMC in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC -d $OUT/p$i -o pmc -- $CHILD > $OUT/p$i.log 2>&1
done
cd $ROOT
python - <<PY | tee $OUT/summary.txt
import glob, sqlite3, os
print("# tools/pmc_path.sh $FLAGS $WL $VAR: per dispatch")
for db in sorted(glob.glob("$OUT/p*/**/*.db", recursive=True)):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    pmc = [t for t in tabs if t.startswith("rocpd_pmc_event")]
    inf = [t for t in tabs if t.startswith("rocpd_info_pmc")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")]
    if not pmc: print(db, "no pmc"); continue
    q = f"""select s.kernel_name, i.name, sum(e.value), count(distinct d.id), avg(d.end - d.start) from {pmc[0]} e join {inf[0]} i on e.pmc_id=i.id
            join {kd[0]} d on e.event_id=d.event_id join {ks[0]} s on d.kernel_id=s.id group by s.kernel_name, i.name"""
    for name, ctr, val, n, dur in c.execute(q):
        if "path_kernel" in name or "trace_kernel" in name or "pool_kernel" in name: print(f"{ctr:40s} {val/n:.6g}  ({n} dispatches, {dur/1e6:.2f} ms)  {name[:60]}")
PY
rm -rf $OUT/p[0-9]
