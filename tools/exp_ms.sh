#!/bin/bash
# many streams on one device: evc_decode with 2 builder threads per worker (default) against 1
cd /root/repo; mkdir -p gpurun_out
for prof in base main; do
  for extra in "" "--builders 1"; do
    tag=$(echo "$prof$extra" | tr -d ' -')
    timeout 900 python tools/bench_multistream.py --streams 1,4,16,32 --profile $prof --extra "$extra" > gpurun_out/ms_$tag.json 2> gpurun_out/ms_$tag.err
    python -c "
import json; d = json.load(open('gpurun_out/ms_$tag.json')); print('$tag', {k: v['decode_only'] if isinstance(v, dict) else v for k, v in d['pictures_per_s'].items()})"
  done
done
