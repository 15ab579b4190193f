#!/bin/bash
# attribution of the dense counting kernels' time: the same ingest with parts of the kernels switched off (PSK_DC_EXP)
for e in 0 1 3 4 12; do
  echo "== PSK_DC_EXP=$e"
  PSK_DC_EXP=$e PSK_PROBE_NOCHECK=1 tools/prof.sh attrib_$e tools/cold_probe.py 64 | grep "dc_\|kernel  "
done
