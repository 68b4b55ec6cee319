#!/bin/bash
# bench A/B of alternative builds of the library: tools/run_libs.sh <dir-under-build> ...  ("product" = in-tree lib)
cd $GRAFT_REPO_ROOT
for D in "$@"; do
  if [ "$D" != "product" ]; then export VSP_LIB_PATH=$GRAFT_REPO_ROOT/build/$D/libvispeech_hip.so; else unset VSP_LIB_PATH; fi
  python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$D', round(d['ms_per_step'],2),'ms', round(d['value']/1e6,1),'M samples/s')"
done
