set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_conv_op.py tests/test_gpu_networks.py -m gpu -x -q -k "x3w8 or fp16x3" > gpurun_out/r2c_pytest.log 2>&1; echo "pytest rc=$?"
timeout 200 python tools/layer_profile.py sceneseg fp16x3 > gpurun_out/r2c_layers_sceneseg_x3.tsv 2>&1
timeout 200 python tools/layer_profile.py scene3d fp16x3 > gpurun_out/r2c_layers_scene3d_x3.tsv 2>&1
timeout 600 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r2c_bench.json 2> gpurun_out/r2c_bench.err; echo "bench rc=$?"
tail -3 gpurun_out/r2c_pytest.log; grep -E "decode_layer_[4-9]|graph replay|eager sum" gpurun_out/r2c_layers_sceneseg_x3.tsv gpurun_out/r2c_layers_scene3d_x3.tsv
cat gpurun_out/r2c_bench.json | cut -c1-900
