# round 4: every table profiles/r04_* is made from (run on the GPU box: bash tools/r04_final.sh)
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04f; mkdir -p $O
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/bstats -o b --output-format csv -- python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
python tools/kernel_stats_md.py $O/bstats --steps 12 > $O/bench_kernel_stats.md
python tools/step_timeline.py $O/bstats > $O/bench_step_timeline.txt 2>&1
B="python bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/bf -o b --output-format csv -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/bw -o b --output-format csv -- $B > /dev/null 2>&1
python tools/pmc_hbm_md.py $O/bf $O/bw > $O/bench_hbm.md 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d $O/bm -o b --output-format csv -- $B > /dev/null 2>&1
python tools/pmc_mfma_md.py $O/bm > $O/bench_mfma.md 2>&1
# fp32 HiFi-GAN generator (VERDICT r03 missing #4)
V="env VP=fp32 VSTREAMS=0 VPAIR=1 VN=3 python tools/voc_prof.py"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/vs -o v --output-format csv -- $V > /dev/null 2>&1
python tools/kernel_stats_md.py $O/vs > $O/voc_fp32_kernel_stats.md
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/vf -o v --output-format csv -- $V > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/vw -o v --output-format csv -- $V > /dev/null 2>&1
python tools/pmc_hbm_md.py $O/vf $O/vw > $O/voc_fp32_hbm.md 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d $O/vm -o v --output-format csv -- $V > /dev/null 2>&1
python tools/pmc_mfma_md.py $O/vm > $O/voc_fp32_mfma.md 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $O/vc -o v --output-format csv -- $V > /dev/null 2>&1
python tools/pmc_clock_md.py $O/vc > $O/voc_fp32_clock.md 2>&1
VP=fp32,bf16,fp16 timeout 300 python tools/voc_ab.py > $O/voc_ab.txt 2>&1
# text side
timeout 200 python tools/text_side_bench.py > $O/text_side.txt 2>&1
PB=32 PL=85 timeout 300 rocprofv3 --kernel-trace -d $O/ts32 -o ts --output-format csv -- python tools/text_side_probe.py > /dev/null 2>&1
python tools/text_side_summary.py $O/ts32 > $O/ts32_timeline.txt 2>&1
PB=1 PL=25 timeout 300 rocprofv3 --kernel-trace -d $O/ts1 -o ts --output-format csv -- python tools/text_side_probe.py > /dev/null 2>&1
python tools/text_side_summary.py $O/ts1 > $O/ts1_timeline.txt 2>&1
timeout 300 python tools/latency_bench.py > $O/latency.txt 2>&1
timeout 300 python tools/ragged_bench.py > $O/ragged.txt 2>&1
timeout 900 python tools/config_bench.py > $O/configs.txt 2>&1
timeout 100 python tools/lp_phases.py > $O/lp_phases.txt 2>&1
timeout 60 tools/bin/l2_lockstep_probe > $O/l2_lockstep.txt 2>&1
CMTTS_FORCE_COLLECTIVE=1 CMTTS_MULTI_EXTRAS=1 timeout 900 python bench.py --steps 10 --no-cpu-baseline > $O/bench_collective.json 2> $O/bench_collective.err
rm -rf $O/bstats $O/bf $O/bw $O/bm $O/vs $O/vf $O/vw $O/vm $O/vc $O/ts32 $O/ts1
ls $O
