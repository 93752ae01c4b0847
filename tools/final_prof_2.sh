set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
for P in fp32 bf16; do
  VP=$P VSTREAMS=0 VPAIR=1 VN=3 timeout 300 rocprofv3 --kernel-trace --stats -d $O/v$P -o v --output-format csv -- python tools/voc_prof.py > /dev/null 2>&1
  python tools/kernel_stats_md.py $O/v$P > $O/voc_${P}_stats.md
  VP=$P VSTREAMS=0 VPAIR=1 VN=3 timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $O/vc$P -o v --output-format csv -- python tools/voc_prof.py > /dev/null 2>&1
  python tools/pmc_clock_md.py $O/vc$P > $O/voc_${P}_clock.md
done
VP=fp32 VSTREAMS=0 VPAIR=1 VN=3 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/vf -o v --output-format csv -- python tools/voc_prof.py > /dev/null 2>&1
VP=fp32 VSTREAMS=0 VPAIR=1 VN=3 timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/vw -o v --output-format csv -- python tools/voc_prof.py > /dev/null 2>&1
python tools/pmc_hbm_md.py $O/vf $O/vw > $O/voc_fp32_hbm.md 2>&1
VP=fp32,bf16,fp16 timeout 300 python tools/voc_ab.py > $O/voc_ab.txt 2>&1
timeout 600 python tools/config_bench.py > $O/configs.txt 2>&1
timeout 300 python tools/latency_bench.py > $O/latency.txt 2>&1
timeout 200 python tools/text_side_bench.py > $O/text_side.txt 2>&1
PB=32 PL=85 timeout 300 rocprofv3 --kernel-trace -d $O/ts32 -o ts --output-format csv -- python tools/text_side_probe.py > /dev/null 2>&1
python tools/text_side_summary.py $O/ts32 > $O/ts32_timeline.txt
PB=1 PL=25 timeout 300 rocprofv3 --kernel-trace -d $O/ts1 -o ts --output-format csv -- python tools/text_side_probe.py > /dev/null 2>&1
python tools/text_side_summary.py $O/ts1 > $O/ts1_timeline.txt
timeout 100 python tools/xres_phases.py > $O/xres_phases.txt 2>&1
ls $O
