#!/bin/bash
# Fabric traffic of a workload's bounce frames by kernel (the traversal kernel and vrt_pool_resolve_kernel apart): FETCH_SIZE and WRITE_SIZE
# in separate rocprofv3 --pmc passes (they do not fit one), L2 requests / hits / misses in a third.
# usage: tools/pmc_traffic.sh <tuning_flags> [workload] [view] [tag]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
FLAGS=${1:-0}; WL=${2:-cfg4_4k_2048c_b8_sparse}; VIEW=${3:-V0}; TAG=${4:-flags$FLAGS}
OUT=$ROOT/gpurun_out/pmc_traffic_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CHILD="python $ROOT/tools/pmc_child.py $WL 0 $FLAGS 3 $VIEW"
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC -d $OUT/p$i -o pmc -- $CHILD > $OUT/p$i.log 2>&1
done
cd $ROOT
python - <<PY | tee $OUT/summary.txt
import glob, sqlite3
print("# tools/pmc_traffic.sh $FLAGS $WL $VIEW: per dispatch of the LAST launch of each kernel (the frame behind the known box)")
rows = {}
for db in sorted(glob.glob("$OUT/p*/**/*.db", recursive=True)):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    pmc = [t for t in tabs if t.startswith("rocpd_pmc_event")]
    inf = [t for t in tabs if t.startswith("rocpd_info_pmc")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")]
    if not pmc: print(db, "no pmc"); continue
    q = f"""select s.kernel_name, i.name, d.id, sum(e.value), max(d.end - d.start) from {pmc[0]} e join {inf[0]} i on e.pmc_id=i.id
            join {kd[0]} d on e.event_id=d.event_id join {ks[0]} s on d.kernel_id=s.id group by s.kernel_name, i.name, d.id order by d.id"""
    for name, ctr, did, val, dur in c.execute(q):
        if any(k in name for k in ("path_kernel", "trace_kernel", "pool_kernel", "pool_resolve")):
            short = name.split("(")[0].replace("void vrt::", "")[:64]
            rows[(short, ctr)] = (val, dur)   # (ordered by dispatch id: the last one stays)
for (k, ctr), (val, dur) in sorted(rows.items()):
    extra = ""
    if ctr == "FETCH_SIZE": extra = f"  = {2.0 * val * 1024 / 1e9:.2f} GB fetched (KiB, 64 B tallied per 128-byte line: doubled)"
    if ctr == "WRITE_SIZE": extra = f"  = {val * 1024 / 1e9:.2f} GB written"
    print(f"{k:66s} {ctr:32s} {val:.6g}  ({dur / 1e6:.2f} ms){extra}")
PY
rm -rf $OUT/p[0-9]
