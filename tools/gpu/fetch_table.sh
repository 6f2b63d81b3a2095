cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf /tmp/pf /tmp/pw
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- python tools/pmc_step.py run > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw -- python tools/pmc_step.py run > /dev/null 2>&1
n=$(python -c "import json; print(len(json.load(open('gpurun_out/pmc_descs.json'))))")
python tools/pmc_step.py reduce /tmp/pf /tmp/pw gpurun_out/r05_pmc_per_shape.json $n | grep -v "^{"
