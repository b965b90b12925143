# (experiment script of round 2: expects the variant library built beside the shipped one; restores nothing -- run on a throw-away GPU box copy only)
mkdir -p gpurun_out/r03a && export GBNNS_CACHE=/tmp/gbnns_cache
for lib in old new; do
  if [ $lib == new ]; then cp gbnns_dim_red_amd/lib/libgbnns_hip_new.so gbnns_dim_red_amd/lib/libgbnns_hip.so; fi
  for ce in "glove 600" "glove 1000" "gist 600" "gist 1000" "glove-dot 600"; do set -- $ce
    python bench.py --config $1 --ef $2 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r03a/$1_$2_$lib.json 2> gpurun_out/r03a/$1_$2_$lib.err
    echo "$lib $1 ef=$2 $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r03a/$1_$2_$lib.json | head -1) $(grep -o '"kernel_ms": [0-9.]*' gpurun_out/r03a/$1_$2_$lib.json | head -1) $(grep -o '"kernel": "[^"]*"' gpurun_out/r03a/$1_$2_$lib.json | head -1) $(grep -o '"rerank": [0-9.a-z]*' gpurun_out/r03a/$1_$2_$lib.json | head -1)"
  done
done
(for s in 61 62 63; do timeout -k 10 400 python tests/stress_rows128.py $s 80; done; for s in 71 72; do timeout -k 10 400 python tests/stress_rows128.py $s 80 -1 any; done) > gpurun_out/r03a/stress.log 2>&1; tail -5 gpurun_out/r03a/stress.log
python -m pytest tests -m gpu -x -q > gpurun_out/r03a/pytest.log 2>&1; tail -2 gpurun_out/r03a/pytest.log
