#!/bin/bash
for a in ${ABL:-0 2 4 7}; do echo "ABLATE=$a L1=${L1:-16} L4=${L4:-64}"; GLX_ABLATE=$a GLX_SELL_L1=${L1:-16} GLX_SELL_L4=${L4:-64} python scripts/sweep_probe.py 2>&1 | grep -E "symmetric k=10|regular L=4|regular L=16|regular L=64" | cut -c1-110; done
