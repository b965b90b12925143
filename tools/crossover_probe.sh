# (experiment script of round 2: expects the variant library built beside the shipped one; restores nothing -- run on a throw-away GPU box copy only)
mkdir -p gpurun_out/r02y && export GBNNS_CACHE=/tmp/gbnns_cache
for lib in base h64; do
  if [ $lib == h64 ]; then cp gbnns_dim_red_amd/lib/libgbnns_hip_h64.so gbnns_dim_red_amd/lib/libgbnns_hip.so; fi
  for ef in 80 100 128; do
    python bench.py --config sift --ef $ef --steps 20 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r02y/sift_${ef}_$lib.json 2> gpurun_out/r02y/sift_${ef}_$lib.err
    echo "$lib sift ef=$ef $(grep -o '"kernel_ms": [0-9.]*' gpurun_out/r02y/sift_${ef}_$lib.json | head -1) $(grep -o '"kernel": "[^"]*"' gpurun_out/r02y/sift_${ef}_$lib.json | head -1)"
  done
  python bench.py --config glove-dot --ef 128 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r02y/glovedot_128_$lib.json 2> gpurun_out/r02y/glovedot_128_$lib.err
  echo "$lib glove-dot ef=128 $(grep -o '"kernel_ms": [0-9.]*' gpurun_out/r02y/glovedot_128_$lib.json | head -1) $(grep -o '"kernel": "[^"]*"' gpurun_out/r02y/glovedot_128_$lib.json | head -1)"
done
