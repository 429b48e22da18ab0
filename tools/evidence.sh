#!/bin/bash
# The round's measured evidence in one call, from the repo root ON the GPU box:   bash tools/evidence.sh r06 [quick]
# Writes gpurun_out/<tag>_*; the files worth judging are then copied into profiles/ (tracked).  `quick` skips the GPU test suite.
#   <tag>_pytest_gpu.log                      python -m pytest tests -q -m gpu
#   <tag>_bench_default.json                  python bench.py   (the driver's N = 1 command)
#   <tag>_rocprofv3_kernel_stats_bench.csv    rocprofv3 --kernel-trace --stats of bench.py --no-secondary --no-cpu-baseline --no-fork --steps 100
#   <tag>_pmc_traffic.json                    separate --pmc FETCH_SIZE / WRITE_SIZE passes of tools/pmc_conv.py, calibrated in-run (tools/pmc_summarize.py)
#   <tag>_pmc_sq_conv.tsv                     SQ counters per dispatch of the matrix kernels (tools/pmc_sq_summarize.py)
#   <tag>_layers_<net>_<precision>.tsv        per-launch HIP-event tables (tools/layer_profile.py)
#   <tag>_trace_sceneseg_fp16x3.tsv           one SceneSeg frame inside the replayed graph (tools/trace_single_stream.py under --kernel-trace)
TAG=${1:-r06}
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
ROOT=$(pwd)
if [ "$2" != "quick" ]; then
  timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -30 > $O/${TAG}_pytest_gpu.log
  tail -3 $O/${TAG}_pytest_gpu.log
fi
timeout 900 python bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_err.log
python - <<PY
import json
d = json.loads(open("$O/${TAG}_bench_default.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "p50_ms", "single_stream_fps", "value_trained_like", "value_three_frames", "fp16_value", "fp16_p50_ms", "sceneseg_fp16_fps", "sceneseg_fp16_p50_ms",
          "sceneseg_fp16x3_fps", "sceneseg_fp16x3_p50_ms", "three_heads_fps", "autodrive_fps", "autodrive_p50_ms", "engine_create_s", "host_to_host_fps", "gather_fps"):
    print(k, d.get(k))
print(json.dumps(d["roofline"]["whole_frame"]))
print(d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["mfma_issue_frac"], d["roofline"]["avg_launch_us"], d["roofline"]["traffic"])
print(json.dumps(d.get("cpu_baseline")))
PY
# ---- rocprofv3 per-kernel statistics of the bench command (kernel trace only: no counters in this pass)
rm -rf /tmp/prof_stats
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $ROOT/bench.py --no-secondary --no-cpu-baseline --no-fork --steps 100 \
  > $ROOT/$O/${TAG}_bench_under_rocprofv3.json 2> /tmp/prof_stats.err)
f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $O/${TAG}_rocprofv3_kernel_stats_bench.csv && head -8 $O/${TAG}_rocprofv3_kernel_stats_bench.csv | cut -c1-200
# ---- HBM-side traffic per launch: two counter passes, each on its own (never with a trace domain beside --kernel-trace)
rm -rf /tmp/pmc_f /tmp/pmc_w
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -- python $ROOT/tools/pmc_conv.py fp16x3 > /dev/null 2> /tmp/pmc_f.err)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -- python $ROOT/tools/pmc_conv.py fp16x3 > /dev/null 2> /tmp/pmc_w.err)
python tools/pmc_summarize.py /tmp/pmc_f /tmp/pmc_w $O/${TAG}_pmc_traffic.json 2>&1 | tail -5
# ---- SQ counters of the matrix kernels (matrix-pipe busy share, clock, wait shares): one more counter pass of its own
rm -rf /tmp/pmc_sq
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE \
  --output-format csv -d /tmp/pmc_sq -- python $ROOT/tools/pmc_conv.py fp16x3 > /dev/null 2> /tmp/pmc_sq.err)
python tools/pmc_sq_summarize.py /tmp/pmc_sq > $O/${TAG}_pmc_sq_conv.tsv 2>/dev/null; head -20 $O/${TAG}_pmc_sq_conv.tsv | cut -c1-160
# ---- per-launch tables and the in-graph trace
for np in "sceneseg fp16x3" "scene3d fp16x3" "egolanes fp16x3" "domainseg fp16x3" "sceneseg fp16" "autodrive fp16"; do
  set -- $np
  timeout 300 python tools/layer_profile.py $1 $2 > $O/${TAG}_layers_$1_$2.tsv 2> /dev/null
done
rm -rf /tmp/trace_ss
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_ss -- python $ROOT/tools/trace_single_stream.py sceneseg fp16x3 20 > /dev/null 2>&1)
python tools/trace_single_stream.py --summarise /tmp/trace_ss > $O/${TAG}_trace_sceneseg_fp16x3.tsv 2> /dev/null
head -3 $O/${TAG}_trace_sceneseg_fp16x3.tsv | cut -c1-200
ls -la $O | grep ${TAG}_ | wc -l
