#!/bin/bash
# The `-theta` experiment of round 2 (run on the GPU box): rebuild the pairing translation unit with the UNARY negation stream at every
# fq2_neg site (line_add's l->c1 = -theta among them) into a scratch copy of the library and run the stage tests that pin the line table,
# the Miller loop and the final exponentiation against the oracle.
set -e
R=$PWD
T=/tmp/neg_unary; rm -rf $T; mkdir -p $T
cp -r $R/keaki_amd $R/oracle $R/tests $R/include $R/bench.py $T/
cd $T/keaki_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DKEAKI_NEG_UNARY -c pairing.hip -o build/pairing.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DKEAKI_NEG_UNARY -c selftest.hip -o build/selftest.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libkeaki_hip.so build/*.o
cd $T
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "line_table or miller or pairing or selftest or verify or golden" 2>&1 | tail -4
