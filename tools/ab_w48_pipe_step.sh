#!/bin/bash
# The same A/B as ab_w48_pipe.sh on the whole denoise step: bench.py's line (ms per step, in-step attention ms) per stream variant,
# cross / step / cross on one box.
set -e
cd "$(dirname "$0")/.."
out=${1:-gpurun_out/ab_w48_step}
mkdir -p $out
gen() { for v in "" "--bias" "--prescaled" "--bias --prescaled"; do W48_PIPE=$1 python tools/gen_attn_w48.py $v > /dev/null; done; timeout 600 make -C ltx-video-swift-mlx_amd/csrc -j8 > /dev/null 2>&1; }
for leg in cross1 step cross2; do
  mode=${leg%[12]}
  gen $mode
  timeout -k 10 300 python bench.py > $out/$leg.json 2> $out/$leg.err
  python - $out/$leg.json $leg <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], "ms_per_step", d["ms_per_step"], "attention_ms", d["attention"]["ms_per_step"], "gemm_ms", d["roofline"]["gemm_ms_per_step"], flush=True)
PY
done
