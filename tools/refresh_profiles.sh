cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r01b; rm -rf $O; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
for w in fft4096 fmchain rotate direct255 decim8 interp4 fir255_i16; do python bench.py --workload $w 2>/dev/null | tail -1 >> $O/bench_other_workloads.jsonl; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_kt -- python3 bench.py --steps 2000 --warmup 50 > $O/bench_kt.log 2>&1
bash tools/prof.sh fir255 $O/fir255 ols4096 > /dev/null 2>&1
bash tools/prof.sh fft4096 $O/fft4096 fft4096 > /dev/null 2>&1
bash tools/prof.sh fmchain $O/fmchain fmchain > /dev/null 2>&1
find $O -name "*.csv" -size +2M -delete
find $O -name "*agent_info*" -delete
du -sh $O
