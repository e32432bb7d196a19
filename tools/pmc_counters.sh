#!/bin/bash
# Any hardware counters of a workload's traversal kernel: one rocprofv3 --pmc pass per quoted set of counter names (a set must fit the
# hardware's counter slots; a name the device does not know fails that pass only).  Per dispatch of the LAST launch of each kernel.
# usage: tools/pmc_counters.sh <workload> <view> <tag> "<set 1>" ["<set 2>" ...]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
WL=${1:-cfg2_1080p_512c_b8}; VIEW=${2:-V0}; TAG=${3:-counters}; FLAGS=${PMC_FLAGS:-0}; shift 3
OUT=$ROOT/gpurun_out/pmc_counters_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CHILD="python $ROOT/tools/pmc_child.py $WL ${PMC_VARIANT:-0} $FLAGS 40 $VIEW"
i=0
for PMC in "$@"; do
  i=$((i+1))
  # (a set the device rejects can leave rocprofv3 waiting for ever: every pass has its own limit)
  timeout -k 5 ${PMC_PASS_LIMIT:-150} rocprofv3 --kernel-trace --pmc $PMC -d $OUT/p$i -o pmc -- $CHILD > $OUT/p$i.log 2>&1 || echo "# pass $i ($PMC): failed or timed out"
done
cd $ROOT
python - <<PY | tee $OUT/summary.txt
import glob, sqlite3
print("# tools/pmc_counters.sh $WL $VIEW: per dispatch of the LAST launch of each kernel")
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
    print(f"{k:66s} {ctr:36s} {val:.6g}  ({dur / 1e3:.1f} us){extra}")
PY
rm -rf $OUT/p[0-9]
