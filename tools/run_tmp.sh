cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | grep -v Warning | tail -12 > gpurun_out/r02c_tests.log
tail -12 gpurun_out/r02c_tests.log
