#!/bin/bash
# GPU-box run of the parity tests (every failure listed, not only the first).  usage: tools/gpu_tests.sh TAG [files...]
TAG=$1; shift
FILES=${@:-tests}
timeout -k 10 1000 python -u -m pytest $FILES -m gpu -q -rA --timeout 400 --timeout-method thread -p no:cacheprovider > gpurun_out/t_$TAG.log 2>&1
rc=$?; echo "tests exit $rc"; grep -E "^(PASSED|FAILED|ERROR)|passed|failed" gpurun_out/t_$TAG.log | grep -v "^PASSED" | tail -40
exit $rc
