cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/suite
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/suite/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/suite/pytest.log
tail -8 gpurun_out/suite/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
