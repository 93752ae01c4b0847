# round 4, final tree (Winograd stack + Winograd wide-stage generator convs): profiles/r04w2_* (run on the GPU box: bash tools/r04w2_prof.sh)
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04w2; mkdir -p $O
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
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
ROUNDS=3 timeout 300 python tools/voc_wino_ab.py > $O/voc_wino_ab.txt 2>&1
VB=64 ROUNDS=2 timeout 300 python tools/voc_wino_ab.py >> $O/voc_wino_ab.txt 2>&1
VB=8 ROUNDS=2 timeout 300 python tools/voc_wino_ab.py >> $O/voc_wino_ab.txt 2>&1
VB=1 VT=150 ROUNDS=2 timeout 300 python tools/voc_wino_ab.py >> $O/voc_wino_ab.txt 2>&1
timeout 900 python tools/config_bench.py > $O/configs.txt 2>&1
rm -rf $O/vs $O/vf $O/vw $O/vm $O/vc
ls $O
