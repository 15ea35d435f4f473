#!/bin/bash
# usage (under gpurun): tools/variant_check.sh <variant> -- the parity tests that pin the step kernels (tiled == gather
# bit for bit, engine vs oracle, the known answers) on rmf_crowdsim_amd/lib/variants/<variant>.so
CS_LIB_PATH=$PWD/rmf_crowdsim_amd/lib/variants/$1.so timeout -k 10 800 python -m pytest tests/test_gpu_parity.py tests/test_zanlungo_kats.py tests/test_gpu_tiles.py tests/test_native_mesh.py -m gpu -q -x --timeout 400 2>&1 | tail -5
