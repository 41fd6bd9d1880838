cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | grep -v Warning | tail -8 > gpurun_out/w2_tests.log
tail -8 gpurun_out/w2_tests.log
rm -f gpurun_out/ab.log
bash tools/ab2.sh "v1 w2" 2 "C2 C3"
