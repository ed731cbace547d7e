#!/bin/bash
# one diagnostic call: the toy probes, then the engine's smoke with stderr kept
cd $(dirname $0); mkdir -p ../../../gpurun_out/r05_u; O=../../../gpurun_out/r05_u/lds0.log
{ ./lds0_symbol; ./lds0_const; ./lds0_const_nosym; } > $O 2>&1
cd ../../..; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_u/smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/r05_u/smoke.log
cat gpurun_out/r05_u/lds0.log; tail -8 gpurun_out/r05_u/smoke.log
