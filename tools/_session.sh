repo=$(pwd)
BN254_LIB_PATH=$repo/build_variants/lib_b_r0peel.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "shared_accumulator or config4_product" 2>&1 | grep -E "passed|failed|error" | tail -2
tools/ab_side.sh r05k_r0peel "product" 2
