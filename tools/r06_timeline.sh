# round 6: launch-order timeline of one bench step (text side + T = 4 evaluations) -> gpurun_out/<tag>/bench_step_timeline.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06}; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O/bstats -o b --output-format csv -- python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
python tools/kernel_stats_md.py $O/bstats --steps 12 > $O/bench_kernel_stats.md
python tools/step_timeline.py $O/bstats > $O/bench_step_timeline.txt 2>&1
rm -rf $O/bstats
