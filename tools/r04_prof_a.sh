set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O/bstats -o b --output-format csv -- python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
python tools/kernel_stats_md.py $O/bstats --steps 8 > $O/bench_kernel_stats.md
python tools/step_timeline.py $O/bstats > $O/bench_timeline.txt 2>&1
MODE=ragged N=3 timeout 300 rocprofv3 --kernel-trace -d $O/rag -o r --output-format csv -- python tools/ragged_bench.py > $O/ragged.txt 2>&1
python tools/shard_timeline.py $O/rag > $O/ragged_timeline.txt 2>&1
python tools/ragged_bench.py > $O/ragged_modes.txt 2>&1
python tools/latency_bench.py > $O/latency.txt 2>&1
ls $O
