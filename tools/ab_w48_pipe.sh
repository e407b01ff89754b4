#!/bin/bash
# A/B of the 48-query attention stream's read pipeline (W48_PIPE=step: reads issued behind the step barrier; default: across it) on
# one box: regenerate, rebuild attention.o, run the micro-benchmark; cross / step / cross to see the box's drift.
set -e
cd "$(dirname "$0")/.."
out=${1:-gpurun_out/ab_w48}
mkdir -p $out
gen() { for v in "" "--bias" "--prescaled" "--bias --prescaled"; do W48_PIPE=$1 python tools/gen_attn_w48.py $v > /dev/null; done; timeout 600 make -C ltx-video-swift-mlx_amd/csrc -j8 > /dev/null 2>&1; }
for leg in cross1 step cross2; do
  mode=${leg%[12]}
  gen $mode
  timeout -k 10 200 python tools/bench_attn.py --impls 4 > $out/$leg.txt 2>&1
  echo "== $leg"; cat $out/$leg.txt | grep -v amdgpu.ids
done
