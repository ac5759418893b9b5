#!/bin/bash
# Round-3 GPU sessions (run through gpurun from the repo root): tools/r03_gpu.sh <stage>.  Output under gpurun_out/r03_<stage>/.
set -o pipefail
stage=$1
out=gpurun_out/r03_$stage
mkdir -p $out
LIB=$PWD/cse168-raytracer_amd
case $stage in
  tests)
    python -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -n 6 $out/pytest.log ;;
  level_ab)
    for v in "" _w4; do
      echo "== lib$v" >> $out/level_ab.log
      MIRO_LIB=$LIB/lib$v/libmiro_hip.so python tools/level_probe.py >> $out/level_ab.log 2>&1
      MIRO_LIB=$LIB/lib$v/libmiro_hip.so python tools/level_probe.py --scene sponza --w 1920 --h 1080 --spp 4 >> $out/level_ab.log 2>&1
    done; cat $out/level_ab.log ;;
  primary)
    python bench.py --scene teapot --width 512 --height 512 --spp 1 --mode primary --steps 200 --no-cpu-baseline > $out/teapot_primary.json 2> $out/err.log
    python bench.py --mode primary --steps 20 --no-cpu-baseline > $out/sponza_primary.json 2>> $out/err.log
    cut -c1-300 $out/teapot_primary.json $out/sponza_primary.json ;;
  bench)
    python bench.py > $out/bench.json 2> $out/bench.err; cut -c1-700 $out/bench.json ;;
  photon)
    python -m pytest tests/test_photon.py -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -n 4 $out/pytest.log
    python bench.py --config photon > $out/photon.json 2> $out/photon.err; cat $out/photon.json; tail -n 3 $out/photon.err ;;
  photon_ab)
    for v in ${PHOTON_VARIANTS:-default _nopf default _nopf}; do
      d=lib$v; [ "$v" = default ] && d=lib
      echo "== $d" >> $out/photon_ab.log
      MIRO_LIB=$LIB/$d/libmiro_hip.so python tools/photon_probe.py $PHOTON_ARGS >> $out/photon_ab.log 2>&1
    done; grep -v amdgpu.ids $out/photon_ab.log ;;
  layouts)
    python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k layouts > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -n 4 $out/pytest.log
    python tools/layout_probe.py > $out/layout_probe.log 2>&1; grep -v amdgpu.ids $out/layout_probe.log ;;
  *) echo "unknown stage $stage"; exit 2 ;;
esac
