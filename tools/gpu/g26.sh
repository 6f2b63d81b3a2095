cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4/g26_smoke.txt 2>&1
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | grep -v Warning | tail -40 > gpurun_out/r4/g26_pytest.txt
echo done
