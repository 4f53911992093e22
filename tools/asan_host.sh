#!/bin/bash
# AddressSanitizer + UBSan over the host-side code that runs without a GPU (GPU sanitizers are not available on this pool):
# the oracle's searches and CorePyExt's CPU tests.  Usage: tools/asan_host.sh   (from the repository root, CPU only)
set -e
cd "$(dirname "$0")/.."
make -C oracle asan > /dev/null
ASAN_LIBS="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so.6)"
mkdir -p /tmp/asan_ext
g++ -O1 -g -std=c++17 -fPIC -shared -fvisibility=hidden -ffp-contract=off -fsanitize=address,undefined \
    -I"$(python3 -c 'import pybind11; print(pybind11.get_include())')" -I"$(python3 -c 'import sysconfig; print(sysconfig.get_paths()["include"])')" \
    gomokuai_amd/csrc/core_pyext.cpp -o /tmp/asan_ext/CorePyExt.cpython-310-x86_64-linux-gnu.so -Lgomokuai_amd -lgomoku_hip -Wl,-rpath,"$PWD/gomokuai_amd"
cat > /tmp/asan_host.py <<'PY'
import ctypes as C, sys
sys.path.insert(0, '/tmp/asan_ext'); sys.path.insert(0, '.')
from oracle import oracle as O
O._SO = 'oracle/libgomoku_oracle_asan.so'
O.build = lambda force=False: O._SO
t = O.PoolRAVEMCTS(2.0, 0.0, seed=3, game_id=1)
moves = [112, 113, 97]
t.run(moves, 800); moves.append(t.step_forward()); t.set_noise(0.05, 0.25); t.run(moves, 300)
tr = O.TraditionalMCTS(5.0); tr.search([112, 113, 97, 98], 600)
m = O.MCTS(300, 5.0, 5, 7, 0); b = O.new_board()
for x in (112, 98): O.lib().go_board_apply(C.byref(b), x, 1)
m.run_playouts(b)
print("oracle searches clean:", t.root_visits, tr.root_visits, m.size)
import CorePyExt, pytest
assert '/tmp/asan_ext' in CorePyExt.__file__
sys.exit(pytest.main(['-x', '-q', 'tests/test_pyext.py', 'tests/test_interface.py', 'tests/test_oracle_golden.py', '-p', 'no:cacheprovider']))
PY
LD_PRELOAD="$ASAN_LIBS" ASAN_OPTIONS=detect_leaks=0 python3 /tmp/asan_host.py
