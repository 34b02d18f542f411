cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g9
timeout 900 python -m pytest tests -m gpu -x -q --timeout=120 --timeout-method=thread > gpurun_out/g9/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/g9/pytest.log
tail -4 gpurun_out/g9/pytest.log
