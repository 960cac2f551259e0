#!/bin/bash
# Scenes between the LDS limit and the large scene under the default choice of pipeline and chunks
# (profiles/r02_midsize_scenes.txt).  Run from the repository root on the GPU box.
for cfg in "100 2" "500 2" "500 3" "2500 4"; do
  for res in "1920 1080 256" "3840 2160 16"; do
    echo "== scene $cfg res $res"
    timeout -k 10 120 python3 -u tools/bigscene_spp.py $cfg $res 2>&1 | grep "pipe" || exit 1
  done
done
echo "== scene 100 2 res 1920 1080 2048 (a long render)"
PIPES=0,1 timeout -k 10 300 python3 -u tools/bigscene_spp.py 100 2 1920 1080 2048 2>&1 | grep "pipe" || exit 1
