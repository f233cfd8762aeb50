cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c; rm -rf $O; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> /dev/null
for w in fft4096 fmchain rotate direct255 decim8 interp4 fir255_i16; do python bench.py --workload $w 2>/dev/null | tail -1 >> $O/bench_other_workloads.jsonl; done
timeout 300 tools/floor_lab 1.0 sweep > $O/floor_lab_sweep.txt 2>&1
cat $O/floor_lab_sweep.txt
