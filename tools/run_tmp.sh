cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python tools/scale_times.py C2 16 2>&1 | tail -6
