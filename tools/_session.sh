repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $out/r05k_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $out/r05k_smoke.log
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > $out/r05k_tests.log 2>&1; echo "tests rc=$?"; tail -6 $out/r05k_tests.log
timeout 600 python bench.py --steps 20 --warmup 3 --no-side > $out/r05k_bench.json 2> $out/r05k_bench.err; python tools/brief_line.py < $out/r05k_bench.json
