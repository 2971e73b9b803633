#!/bin/bash
# usage: showloop.sh <mangled-prefix>   -- condensed view of the first depth-1 loop of a kernel in ../lib/asm/api.s
S=../lib/asm/api.s
N=$(grep -n "^$1.*:" $S | head -1 | cut -d: -f1)
[ -z "$N" ] && { echo "kernel not found"; exit 1; }
tail -n +$N $S | awk '/s_endpgm/{print; exit} {print}' > /tmp/kern.s
L=$(grep -n "Loop Header: Depth=1" /tmp/kern.s | head -1 | cut -d: -f1)
tail -n +$L /tmp/kern.s | head -${2:-300} | grep -vE "^\s*;|^\s*$" | awk '/v_mfma/{c++; next} {if(c){print "    <" c " mfma>"; c=0} print}' | cut -c1-90 | grep -E "mfma>|ds_read|global_load_lds|s_waitcnt|s_barrier|LBB"
