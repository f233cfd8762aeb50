# The judged evidence set, produced on the GPU box:  gpurun -- bash tools/refresh_profiles.sh [outdir]
# then, here:  python tools/collect_profiles.py <outdir> profiles/<round>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r02b}; rm -rf $O; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> /dev/null
for w in fft4096 fmchain rotate direct255 decim8 interp4 fir255_i16; do python bench.py --workload $w 2>/dev/null | tail -1 >> $O/bench_other_workloads.jsonl; done
PCX_BENCH_BACKEND=gloo python bench.py --gpus 2 --shard 33554432 --steps 50 --warmup 10 --no-cpu > $O/bench_two_ranks_one_gpu_gloo.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_kt -- python3 bench.py --steps 2000 --warmup 50 --no-cpu > $O/bench_kt.log 2>&1
bash tools/prof.sh fir255 $O/fir255 ols4096 > /dev/null 2>&1
bash tools/prof.sh fft4096 $O/fft4096 fft_r16 > /dev/null 2>&1
bash tools/prof.sh fmchain $O/fmchain fmchain > /dev/null 2>&1
bash tools/ab_sched.sh > $O/ab_sched.txt 2>/dev/null
bash tools/ab_oversub.sh > $O/ab_oversub.txt 2>/dev/null
bash tools/ab_fft4096_family.sh 2>/dev/null | grep -v amdgpu.ids > $O/ab_fft4096_family.txt
bash tools/ab_fft_family_oversub.sh > $O/ab_fft_family_rounds.txt 2>/dev/null
python tools/shard_probe.py > $O/shard_probe.txt 2>/dev/null
python tools/host_path.py > $O/host_path.txt 2>/dev/null
python tools/two_blocks.py > $O/two_blocks.txt 2>/dev/null
python tools/chain_path.py > $O/chain_path.txt 2>/dev/null
rocprofv3 --kernel-trace --output-format csv -d $O/two -- python3 tools/two_blocks.py trace > /dev/null 2>&1
python tools/two_blocks.py summarize $O/two >> $O/two_blocks.txt 2>/dev/null
tools/pcie_lab > $O/pcie_lab.txt 2>&1
tools/ubench > $O/ubench_roofs.txt 2>&1
timeout 300 tools/ols_lab 4 > $O/ols_lab_summary.txt 2>&1
python tools/sweep_fir.py > $O/sweep_fir_taps.txt 2>/dev/null
python tools/sweep_map.py > $O/sweep_elementwise.txt 2>/dev/null
python tools/sweep_fft.py 16 64 256 1024 2048 4096 8192 16384 > $O/sweep_fft_sizes.txt 2>/dev/null
python tools/sweep_fft_f64.py > $O/sweep_fft_f64.txt 2>/dev/null
python tools/sweep_fft_mixed.py > $O/sweep_fft_mixed.txt 2>/dev/null
python tools/sweep_fir_f64.py > $O/sweep_fir_f64.txt 2>/dev/null
python tools/real_probe.py > $O/real_f32_fir.txt 2>/dev/null
find $O -name "*.csv" -size +2M -delete
find $O -name "*agent_info*" -delete
du -sh $O
