# Round 6's judged evidence set (a trimmed tools/refresh_profiles.sh: what CHANGED this round plus the lines the driver's run is compared
# with), produced on the GPU box:  gpurun -- bash tools/refresh_profiles_r06.sh [outdir]
# then, here:  python tools/collect_profiles.py <outdir> profiles/r06
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r06z}; rm -rf $O; mkdir -p $O
ulimit -c 0
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> /dev/null
for w in fft4096 fmchain rotate abs freq_demod direct255 decim8 interp4 fir255_i16 fir4097 fir8193 fir4097_real; do python bench.py --workload $w 2>/dev/null | tail -1 >> $O/bench_other_workloads.jsonl; done
PCX_BENCH_BACKEND=gloo python bench.py --gpus 2 --shard 33554432 --steps 50 --warmup 10 --no-cpu > $O/bench_two_ranks_one_gpu_gloo.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_kt -- python3 bench.py --steps 2000 --warmup 50 --no-cpu --no-secondary > $O/bench_kt.log 2>&1
bash tools/prof.sh fir255 $O/fir255 ols4096 > /dev/null 2>&1
bash tools/prof.sh fir255_i16 $O/fir255_i16 fir_cf64_ip > /dev/null 2>&1
# (fft4096.hpp changed late in the round -- a hook in dif_rest, the other kernels' objects are bit-identical -- so their stamps are renewed as well)
bash tools/prof.sh fft4096 $O/fft4096 fft_r16 > /dev/null 2>&1
bash tools/prof.sh fmchain $O/fmchain fmchain > /dev/null 2>&1
bash tools/prof.sh decim8 $O/decim8 decim > /dev/null 2>&1
bash tools/prof.sh interp4 $O/interp4 interp > /dev/null 2>&1
bash tools/prof.sh fir4097 $O/fir4097 upols > /dev/null 2>&1
bash tools/prof.sh fir8193 $O/fir8193 upols > /dev/null 2>&1
bash tools/ab_upols.sh > $O/ab_upols.txt 2>&1
./tools/f64_lab > $O/f64_lab.txt 2>&1
bash tools/ip64_parts.sh > $O/ip64_parts.txt 2>&1
bash tools/ab_ip64_sched.sh > $O/ab_ip64_sched.txt 2>&1
timeout 200 ./tools/rccl_group_lab 2>&1 | grep "G = " > $O/rccl_group_lab.txt
python -m pytest tests -m gpu -q -n 4 2>&1 | tail -3 > $O/pytest_gpu.txt
PCX_FUZZ_SEEDS=1500 python -m pytest tests/test_fuzz_gpu.py tests/test_parity_gpu.py -m gpu -q -n 8 -k "random or fuzz or chunk or chain" 2>&1 | tail -2 > $O/soak.txt
find $O -name "*.csv" -size +2M -delete
find $O -name "*agent_info*" -delete
du -sh $O
