# the whole GPU suite; the round's profile set (tools/profile_round.sh ${TAG:-r06}) only when the suite is green -- a GPU step that
# failed or was killed is followed by no further GPU step in the same call
cd $GRAFT_REPO_ROOT
O=gpurun_out/full
mkdir -p $O
timeout -k 10 ${PYTEST_LIMIT:-700} python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -3 $O/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
[ "${SKIP_PROFILE:-0}" = 1 ] && exit 0
bash tools/profile_round.sh ${TAG:-r06} > $O/profile_round.log 2>&1
rc=$?
echo "profile rc=$rc"; tail -25 $O/profile_round.log
exit $rc
