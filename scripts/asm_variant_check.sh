#!/bin/bash
# build one knob variant (see asm_variants.sh) and run the parity check + phase times on it
name=$1; envs=$2
bash scripts/asm_variants.sh "$name:$envs" > /dev/null 2>&1
echo "== $name ($envs)"
DMPC_LIB=/tmp/var/lib_$name.so timeout -k 10 200 python tests/tools/asm_check.py 2>&1 | tail -1
bash -c "SCRIPT=scripts/asm_phase_times.py bash scripts/asm_variants.sh '${name}_t:GEN_TIMING=1 $envs'" 2>&1 | grep -E "backward|forward"
