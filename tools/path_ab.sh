set -u
cd ${GRAFT_REPO_ROOT:-.}
for f in 1 2 3 4; do python bench.py --no-cpu-baseline --pmc off --frames-in-flight $f 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('in flight', $f, 'value', round(d['value']), 'ms/step', round(d['ms_per_step'],5))"; done
