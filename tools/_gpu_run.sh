cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2m
timeout 600 python -m pytest tests/test_gpu_conv_op.py tests/test_gpu_networks.py tests/test_gpu_shared_heads.py -m gpu -x -q -k "x3w8 or fp16x3" > gpurun_out/r2m/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r2m/pytest.log
timeout 200 python tools/layer_profile.py sceneseg fp16x3 > gpurun_out/r2m/layers_sceneseg_fp16x3.tsv 2>&1
grep -E "context_layer_[4-6]|decode_layer_[0-9]|graph replay|eager sum" gpurun_out/r2m/layers_sceneseg_fp16x3.tsv
timeout 300 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r2m/bench.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r2m/bench.json')); print('value', d['value'], 'single', d['single_stream_fps'], 'p50', d['p50_ms'])"
