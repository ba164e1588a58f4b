#!/bin/bash
# sweep kernel: one word of the row's bias record fetched in front of the chunk loop (GLX_TOUCH_BIAS), A/B on one box
for v in "-DGLX_TOUCH_BIAS=0" "-DGLX_TOUCH_BIAS=1" "-DGLX_TOUCH_BIAS=0" "-DGLX_TOUCH_BIAS=1"; do
  echo "== $v"
  GLX_CXXFLAGS="$v" python -m graphlearning_amd._build > /dev/null 2>&1
  GLX_CXXFLAGS="$v" python bench.py --no-traffic --no-scale 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['roofline']['avg_launch_us'], d.get('fp32', {}).get('avg_launch_us'))"
done
