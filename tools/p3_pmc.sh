#!/bin/bash
# fp16x3 HiFi-GAN generator: fabric traffic (FETCH_SIZE / WRITE_SIZE, separate passes), MFMA-busy and clock tables -> gpurun_out/p3pmc/*.md
cd /tmp; export TMPDIR=/tmp
O=/root/repo/gpurun_out/p3pmc; rm -rf $O; mkdir -p $O
V="env VSTREAMS=0 VP=fp16x3 VN=3 python /root/repo/tools/voc_prof.py"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/vf -o v --output-format csv -- $V > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/vw -o v --output-format csv -- $V > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d $O/vm -o v --output-format csv -- $V > /dev/null 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $O/vc -o v --output-format csv -- $V > /dev/null 2>&1
cd /root/repo
python tools/pmc_hbm_md.py $O/vf $O/vw > $O/hbm.md 2>&1
python tools/pmc_mfma_md.py $O/vm > $O/mfma.md 2>&1
python tools/pmc_clock_md.py $O/vc > $O/clock.md 2>&1
rm -rf $O/vf $O/vw $O/vm $O/vc
cat $O/hbm.md
