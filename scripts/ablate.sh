#!/bin/bash
for a in ${ABL:-0}; do echo "ABLATE=$a L1=${L1:-24} L4=${L4:-96} SIGMA=${SIGMA:-all}"; env GLX_ABLATE=$a GLX_SELL_L1=${L1:-24} GLX_SELL_L4=${L4:-96} ${SIGMA:+GLX_SELL_SIGMA=$SIGMA} python scripts/sweep_probe.py 2>&1 | grep -E "symmetric k=10|regular L=16|Error|error" | cut -c1-200; done
