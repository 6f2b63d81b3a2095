cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
LEFTREFILL_AUTOTUNE=1 timeout 2400 python tools/tune_tiles.py --fresh --workloads single --out gpurun_out/r4/tile_table_single_r4.json > gpurun_out/r4/g20_tune.log 2>&1
python - > gpurun_out/r4/g20_diff.txt <<'PY'
import json
old = json.load(open("leftrefill_amd/tile_table.json"))
new = json.load(open("gpurun_out/r4/tile_table_single_r4.json"))
diff = {k: (old.get(k), v) for k, v in new.items() if old.get(k) != v}
print(len(new), "single shapes,", len(diff), "differ")
for k, (a, b) in sorted(diff.items()):
    print(k, a, "->", b)
old.update(new)
json.dump(old, open("gpurun_out/r4/tile_table_merged_r4.json", "w"), indent=0)
PY
b() { timeout 600 env $2 python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > gpurun_out/r4/g20_bench_$1.json 2> gpurun_out/r4/g20_bench_$1.err; }
b base X=1
b merged LEFTREFILL_TILE_TABLE_PATH=$GRAFT_REPO_ROOT/gpurun_out/r4/tile_table_merged_r4.json
b base2 X=1
b merged2 LEFTREFILL_TILE_TABLE_PATH=$GRAFT_REPO_ROOT/gpurun_out/r4/tile_table_merged_r4.json
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "out_block" 2>&1 | tail -3 > gpurun_out/r4/g20_pytest.txt
bash tools/kstats.sh r4d > gpurun_out/r4/g20_kstats.txt 2>&1
echo done
