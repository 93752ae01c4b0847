# round 6: the tables profiles/r06_* are made from (run on the GPU box: bash tools/r06_prof.sh <tag> <commit>); raw directories are removed, the
# summaries land in gpurun_out/<tag>/ and are copied into profiles/ by hand.  Counter passes first: bench.py copies roofline.traffic /
# mfma_busy from profiles/pmc_traffic.json, which is refreshed from them before the default bench line is taken.
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06}; mkdir -p $O
COMMIT=${2:-unknown}
timeout 300 rocprofv3 --kernel-trace --stats -d $O/bstats -o b --output-format csv -- python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
python tools/kernel_stats_md.py $O/bstats --steps 12 > $O/bench_kernel_stats.md
python tools/step_timeline.py $O/bstats > $O/bench_step_timeline.txt 2>&1
B="python bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/bf -o b --output-format csv -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/bw -o b --output-format csv -- $B > /dev/null 2>&1
python tools/pmc_hbm_md.py $O/bf $O/bw > $O/bench_hbm.md 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d $O/bm -o b --output-format csv -- $B > /dev/null 2>&1
python tools/pmc_mfma_md.py $O/bm > $O/bench_mfma.md 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $O/bc -o b --output-format csv -- $B > /dev/null 2>&1
python tools/pmc_clock_md.py $O/bc > $O/bench_clock.md 2>&1
python tools/pmc_update_json.py denoiser_persist_kernel_wino43 "denoiser_persist_kernel<false, false, true, 2>" $O/bench_hbm.md $O/bench_mfma.md $O/bench_clock.md $COMMIT \
  "round 6: FACT + F(4,3) 8-wave instance (WINO == 2), the default fp32 stack (publish-phase gathers as buffer loads)" > $O/pmc_entry.json 2>&1
cp profiles/pmc_traffic.json $O/pmc_traffic.json
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
FACT=1 WINO=3 timeout 100 python tools/persist_timing.py > $O/persist_timing_wino43.txt 2>&1
FACT=1 WINO=1 timeout 100 python tools/persist_timing.py > $O/persist_timing_wino.txt 2>&1
timeout 900 python tools/config_bench.py > $O/configs.txt 2>&1
rm -rf $O/bstats $O/bf $O/bw $O/bm $O/bc
ls $O
