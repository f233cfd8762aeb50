cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { PCX_HIP_LIBRARY=$1 python bench.py --no-cpu --no-cold --sustain 0 --workload $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-8s %-10s %.4f ms  frac %.4f' % ('$3', '$2', d['roofline']['avg_launch_ms'], d['roofline']['frac']))"; }
B=$PWD/tools/_ab/libpcx_hip_r04base.so; N=$PWD/tools/_ab/libpcx_hip_new.so
for rep in 1 2 3; do run $N new fmchain; run $B base fmchain; run $N new fir255; run $B base fir255; done
