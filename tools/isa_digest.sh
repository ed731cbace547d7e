#!/bin/bash
# sha256 of the gfx950 ISA of every kernel / device function in a built libur5sim.so (instruction text only): two builds with equal digests run the same machine code.
#   tools/isa_digest.sh [lib = mujoco_rl_ur5_amd/csrc/libur5sim.so]
lib=${1:-mujoco_rl_ur5_amd/csrc/libur5sim.so}
LLVM=/opt/rocm/lib/llvm/bin
D=$(mktemp -d /tmp/isa.XXXXXX)
cp "$lib" $D/lib.so
(cd $D && $LLVM/llvm-objdump --offloading lib.so > /dev/null 2>&1
 for f in *amdgcn*; do
   $LLVM/llvm-objdump -d --no-show-raw-insn $f | sed -e 's/^ *[0-9a-f]*: *//' -e 's/\/\/.*$//' | grep -v "file format\|^$\|Disassembly" | sha256sum | cut -c1-16 | sed "s/$/  $f/"
 done)
