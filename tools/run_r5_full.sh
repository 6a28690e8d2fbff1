# the whole GPU suite, then the round's profile set (tools/profile_round.sh r05)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5full
mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
bash tools/profile_round.sh r05 > $O/profile_round.log 2>&1; echo "profile rc=$?"; tail -25 $O/profile_round.log
