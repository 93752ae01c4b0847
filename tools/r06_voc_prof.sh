# round 6: the generator's counters with the fused k = 3 pairs and the XCD-aware tile order (profiles/r06_voc_{fp32,bf16}_*; run on the GPU box: bash tools/r06_voc_prof.sh)
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06v; mkdir -p $O
for P in fp32 bf16; do
V="env VP=$P VSTREAMS=0 VPAIR=1 VN=3 python tools/voc_prof.py"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/vs -o v --output-format csv -- $V > /dev/null 2>&1
python tools/kernel_stats_md.py $O/vs --steps 3 > $O/voc_${P}_kernel_stats.md
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/vf -o v --output-format csv -- $V > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/vw -o v --output-format csv -- $V > /dev/null 2>&1
python tools/pmc_hbm_md.py $O/vf $O/vw > $O/voc_${P}_hbm.md 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d $O/vm -o v --output-format csv -- $V > /dev/null 2>&1
python tools/pmc_mfma_md.py $O/vm > $O/voc_${P}_mfma.md 2>&1
rm -rf $O/vs $O/vf $O/vw $O/vm
done
ls $O
