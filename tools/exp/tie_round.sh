export RCN_EXPERIMENT=1   # engine.hip read_knobs: RCN_* switches are ignored without it
bash tools/exp/layer_prof.sh > gpurun_out/layer_prof2.txt 2>&1
python bench.py --steps 5 --warmup 1 --no-cpu --no-product --no-upload-leg --verify > gpurun_out/bench_tie.json 2> gpurun_out/bench_tie.err
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_band.py tests/test_gpu_fullsize.py -m gpu -q -x > gpurun_out/pytest_tie.log 2>&1; tail -5 gpurun_out/pytest_tie.log
