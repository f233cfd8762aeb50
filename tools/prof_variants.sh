# PMC passes of the headline kernel, dynamic dealing + priority (product) against the static grid stride (round 1 walk), diagnostic library
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r02v}; mkdir -p $O
export PCX_HIP_LIBRARY=$GRAFT_REPO_ROOT/pothoscomms_amd/libpcx_hip_diag.so
bash tools/prof.sh fir255 $O/fir255_dynamic ols4096 > /dev/null 2>&1
PCX_SCHED_STATIC=1 bash tools/prof.sh fir255 $O/fir255_static ols4096 > /dev/null 2>&1
PCX_SCHED_STATIC=1 bash tools/prof.sh fft4096 $O/fft4096_static fft4096 > /dev/null 2>&1
PCX_SCHED_STATIC=1 bash tools/prof.sh fmchain $O/fmchain_static fmchain > /dev/null 2>&1
for v in fir255_dynamic fir255_static fft4096_static fmchain_static; do echo "== $v"; cat $O/$v/summary.txt; done
find $O -name "*.csv" -size +1M -delete; find $O -name "*agent_info*" -delete
