cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04f
( time timeout 3000 python -m pytest tests/ -x -q -m gpu -n 4 ) > gpurun_out/r04f/pytest_gpu_n4.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r04f/pytest_gpu_n4.txt
tail -4 gpurun_out/r04f/pytest_gpu_n4.txt
bash tools/refresh_profiles.sh gpurun_out/r04g
