cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a; mkdir -p $O
timeout 1200 python -m pytest tests/test_qformat_gpu.py -m gpu -x -q -n 4 > $O/pytest_qformat.txt 2>&1; echo "pytest rc $?" >> $O/pytest_qformat.txt
tail -15 $O/pytest_qformat.txt
timeout 2400 python -m pytest tests -m gpu -q -n 4 > $O/pytest_gpu_full.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu_full.txt
tail -15 $O/pytest_gpu_full.txt
