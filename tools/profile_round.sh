# One profiling session of the bench command on the GPU box: kernel statistics + timeline, HBM traffic counters,
# matrix-pipe counters.  Usage (from the repo root on the box): bash tools/profile_round.sh r02
TAG=${1:-rXX}
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --cpu-sample 0 --extra-steps 0 --landmark-method device"
O=gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O profiles
rocprofv3 --kernel-trace --stats -d $O/stats -o bench -- $B --steps 3 --warmup 1 > $O/stats.out 2> $O/stats.log
grep '^{' $O/stats.out > profiles/${TAG}_bench_c3_1gpu_under_rocprof.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o bench -- $B --steps 1 --warmup 0 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o bench -- $B --steps 1 --warmup 0 > $O/write.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace -d $O/sq -o bench -- $B --steps 1 --warmup 0 > $O/sq.log 2>&1
S=$(find $O/stats -name "*.db" | head -1); F=$(find $O/fetch -name "*.db" | head -1); W=$(find $O/write -name "*.db" | head -1); Q=$(find $O/sq -name "*.db" | head -1)
python tools/rocpd_summary.py $S $F $W > profiles/${TAG}_bench_c3_summary.txt 2>&1
python tools/rocpd_timeline.py $S 20 > profiles/${TAG}_bench_c3_timeline.txt 2>&1
python tools/pmc_summary.py $Q > profiles/${TAG}_pmc_sq.txt 2>&1
python tools/make_profiles.py $S $F $W $Q 1000000 5000 5008 $TAG
cp profiles/${TAG}_* profiles/objective_traffic.json profiles/mfma_util.json gpurun_out/ 2>/dev/null
find gpurun_out -name "*.db" -size +30M -delete
