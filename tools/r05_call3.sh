#!/bin/bash
# round 5, GPU call 3: bisect of the device-side divergence under concurrent contexts (reproduced in call 2 in BOTH read-back arms: not a read-back problem)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out
O=gpurun_out/r05_c3
run() { name=$1; shift; ( env "$@" timeout 150 python tools/clone_stress.py 100000 mini 12 3 105 ) > ${O}_stress_$name.txt 2>&1; echo "rc $?" >> ${O}_stress_$name.txt; echo "== $name"; grep -c "coarse:\|semantic:" ${O}_stress_$name.txt; grep "GUARD\|clone_stress:\|errors" ${O}_stress_$name.txt | head -6; }
# every arm but the last on the routes that reproduced in call 2 (BARK_HIP_FEW_SLOTS=0: the lock-step routes of round 4)
run control_guard BARK_HIP_FEW_SLOTS=0 BARK_HIP_GUARD=1
run no_tail BARK_HIP_FEW_SLOTS=0 BARK_HIP_DIAG_NO_TAIL=1
run tail_inline BARK_HIP_FEW_SLOTS=0 BARK_HIP_TAIL_STREAM=0
run tail_normal_priority BARK_HIP_FEW_SLOTS=0 BARK_HIP_TAIL_PRIORITY=0
run eager BARK_HIP_FEW_SLOTS=0 BARK_HIP_GRAPH=0
run new_few_slot_route BARK_HIP_GUARD=0
