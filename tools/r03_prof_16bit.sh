# round 3: kernel stats, calibrated HBM-side traffic, MFMA-busy and clock tables of the 16-bit paths (bf16 HiFi-GAN generator, lp persistent denoiser)
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03p; mkdir -p $O
if [ -z "$LPONLY" ]; then
V="env VP=bf16 VSTREAMS=0 VN=3 python tools/voc_prof.py"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/vs -o v --output-format csv -- $V > /dev/null 2>&1
python tools/kernel_stats_md.py $O/vs > $O/voc_bf16_kernel_stats.md
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/vf -o v --output-format csv -- $V > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/vw -o v --output-format csv -- $V > /dev/null 2>&1
python tools/pmc_hbm_md.py $O/vf $O/vw > $O/voc_bf16_hbm.md 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d $O/vm -o v --output-format csv -- $V > /dev/null 2>&1
python tools/pmc_mfma_md.py $O/vm > $O/voc_bf16_mfma.md 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $O/vc -o v --output-format csv -- $V > /dev/null 2>&1
python tools/pmc_clock_md.py $O/vc > $O/voc_bf16_clock.md 2>&1
fi
for LP in bf16 fp16x3; do
L="env LP=$LP VN=3 python tools/lp_prof.py"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/ls$LP -o l --output-format csv -- $L > /dev/null 2>&1
python tools/kernel_stats_md.py $O/ls$LP > $O/lp_${LP}_kernel_stats.md
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/lf$LP -o l --output-format csv -- $L > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/lw$LP -o l --output-format csv -- $L > /dev/null 2>&1
python tools/pmc_hbm_md.py $O/lf$LP $O/lw$LP > $O/lp_${LP}_hbm.md 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d $O/lm$LP -o l --output-format csv -- $L > /dev/null 2>&1
python tools/pmc_mfma_md.py $O/lm$LP > $O/lp_${LP}_mfma.md 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $O/lc$LP -o l --output-format csv -- $L > /dev/null 2>&1
python tools/pmc_clock_md.py $O/lc$LP > $O/lp_${LP}_clock.md 2>&1
done
rm -rf $O/vs $O/vf $O/vw $O/vm $O/vc $O/ls* $O/lf* $O/lw* $O/lm* $O/lc*
ls $O; cat $O/lp_bf16_hbm.md $O/lp_bf16_mfma.md $O/lp_bf16_clock.md | head -40
