# Profiling battery of a round (run on the MI355X box through gpurun): rocprofv3 kernel stats of bench.py, calibrated FETCH/WRITE
# traffic and matrix-pipe utilisation passes (separate --pmc runs), summarised into gpurun_out/prof_<round>/ (argument: round tag, default r4).
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
RT=${1:-r4}
O=$R/gpurun_out/prof_$RT
rm -rf $O; mkdir -p $O
echo stats; rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
# the same run with the weight gradients on the main stream: every launch has the device to itself (what bench.py's roofline times)
export ECM_WGRAD_OVERLAP=0
echo stats unshared; rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_unshared -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_under_rocprof_unshared.json 2> $O/bench_under_rocprof_unshared.err
unset ECM_WGRAD_OVERLAP
P="rocprofv3 --kernel-trace --output-format csv"
echo calib; $P --pmc FETCH_SIZE -d $O/calib_fetch -- $R/tools/micro/fetch_calib > /dev/null 2>&1
$P --pmc WRITE_SIZE -d $O/calib_write -- $R/tools/micro/fetch_calib > /dev/null 2>&1
for w in wino:wino wino_wgrad:ww costvol:cv ecmw_bwd:ew conv:conv c1gn:c1; do
  k=${w%%:*}; t=${w##*:}
  echo $k; $P --pmc FETCH_SIZE -d $O/${t}_fetch -- python3 $R/tools/conv_only.py $k 4 > /dev/null 2>&1
  $P --pmc WRITE_SIZE -d $O/${t}_write -- python3 $R/tools/conv_only.py $k 4 > /dev/null 2>&1
done
M="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
for k in wino wino_wgrad ecmw_bwd; do
  echo mfma $k; $P --pmc $M -d $O/mfma_$k -- python3 $R/tools/conv_only.py $k 4 > /dev/null 2>&1
done
# round 4: the 2-D Winograd instantiation (8 images) and the instruction mix per launch of both
for k in wino2d wino2d64 wino2d128; do
  echo mfma $k; $P --pmc $M -d $O/mfma_$k -- python3 $R/tools/conv_only.py $k 8 > /dev/null 2>&1
done
for w in wino:4 wino2d:8; do
  k=${w%%:*}; b=${w##*:}
  echo insts $k; $P --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES -d $O/insts_$k -- python3 $R/tools/conv_only.py $k $b > /dev/null 2>&1
done
cd $R
python3 tools/pmc_traffic.py $O > $O/pmc_traffic.json
python3 tools/pmc_mfma_summary.py wino=$O/mfma_wino wino_wgrad=$O/mfma_wino_wgrad ecmw_bwd=$O/mfma_ecmw_bwd wino2d_32=$O/mfma_wino2d wino2d_64=$O/mfma_wino2d64 wino2d_128=$O/mfma_wino2d128 > $O/pmc_mfma_util.json
python3 tools/pmc_insts_summary.py wino=$O/insts_wino wino2d_32=$O/insts_wino2d > $O/pmc_instruction_mix.json
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
cp $(ls $O/stats_unshared/*/*kernel_stats.csv | head -1) $O/kernel_stats_unshared.csv
# keep the merged output small: drop the raw traces
find $O -name "*kernel_trace.csv" -size +2M -delete
find $O -name "*.db" -delete
du -sh $O
