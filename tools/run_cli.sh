TIMEFORMAT="   -> %R s wall, %U user, %S sys"
python3 - <<'PY'
import pollen_amd as pa, os
g = pa.synth(1, 1_000_000, 1000, 100_000, "pangenome", False)
g.write_flatgfa("/dev/shm/cfgL.flatgfa")
PY
echo "== new (warm thread + populate + quick exit)"; for i in 1 2 3 4 5 6; do time pollen_amd/bin/fgfa -i /dev/shm/cfgL.flatgfa depth -d >/dev/null; done
echo "== no warm thread"; for i in 1 2 3 4; do time FLATGFA_NO_WARM=1 pollen_amd/bin/fgfa -i /dev/shm/cfgL.flatgfa depth -d >/dev/null; done
echo "== no warm thread, slow exit (as before)"; for i in 1 2 3 4; do time FLATGFA_NO_WARM=1 FLATGFA_SLOW_EXIT=1 pollen_amd/bin/fgfa -i /dev/shm/cfgL.flatgfa depth -d >/dev/null; done
echo "== new, with timing"; FLATGFA_TIMING=1 pollen_amd/bin/fgfa -i /dev/shm/cfgL.flatgfa depth -d 2>&1 >/dev/null | grep -v worker
echo "== cpu"; for i in 1 2 3; do time oracle/_build/fgfa_depth_cpu /dev/shm/cfgL.flatgfa -d > /dev/null; done
echo "== path depth new"; for i in 1 2 3; do time pollen_amd/bin/fgfa -i /dev/shm/cfgL.flatgfa depth >/dev/null; done
pollen_amd/bin/fgfa -i /dev/shm/cfgL.flatgfa depth -d | md5sum; oracle/_build/fgfa_depth_cpu /dev/shm/cfgL.flatgfa -d | md5sum
rm -f /dev/shm/cfgL.flatgfa
timeout 600 python3 -m pytest tests/test_gpu_depth.py tests/test_c_abi_example.py tests/test_gpu_next_rows.py -m gpu -x -q -k "cli or c_abi or byte" 2>&1 | tail -2
