cd $GRAFT_REPO_ROOT
for p in 1 0 1 0; do CMTTS_SIDE_PRIO=$p python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.load(sys.stdin);print('side_prio=$p', d['ms_per_step'], d['value'])"; done
CMTTS_SIDE_PRIO=1 python tools/latency_bench.py 2>/dev/null | head -12
CMTTS_SIDE_PRIO=1 python tools/ragged_bench.py 2>/dev/null | tail -3
CMTTS_SIDE_PRIO=0 python tools/ragged_bench.py 2>/dev/null | tail -3
