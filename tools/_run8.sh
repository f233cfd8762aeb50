cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
D=$PWD/pothoscomms_amd/libpcx_hip_diag.so
for rep in 1 2; do
echo "product (occ3 G=2 hreg tw3)"; PCX_HIP_LIBRARY=$D python tools/resampler_sweep.py decim 8 2>/dev/null | tail -1
echo "occ4 tw3=0 G=2";  PCX_HIP_LIBRARY=$D PCX_DECIM_OCC=4 PCX_DECIM_TW3=0 PCX_DECIM_G=1 python tools/resampler_sweep.py decim 8 2>/dev/null | tail -1
echo "occ4 tw3=1 G=2";  PCX_HIP_LIBRARY=$D PCX_DECIM_OCC=4 PCX_DECIM_TW3=1 PCX_DECIM_G=1 python tools/resampler_sweep.py decim 8 2>/dev/null | tail -1
echo "occ4 tw3=0 G=4";  PCX_HIP_LIBRARY=$D PCX_DECIM_OCC=4 PCX_DECIM_TW3=0 PCX_DECIM_G=2 python tools/resampler_sweep.py decim 8 2>/dev/null | tail -1
echo "occ3 hreg=0 G=2"; PCX_HIP_LIBRARY=$D PCX_DECIM_HREG=0 PCX_DECIM_G=1 python tools/resampler_sweep.py decim 8 2>/dev/null | tail -1
echo "unbatched occ4";  PCX_HIP_LIBRARY=$D PCX_DECIM_UNBATCHED=1 PCX_DECIM_OCC=4 python tools/resampler_sweep.py decim 8 2>/dev/null | tail -1
echo "unbatched occ3";  PCX_HIP_LIBRARY=$D PCX_DECIM_UNBATCHED=1 python tools/resampler_sweep.py decim 8 2>/dev/null | tail -1
done
