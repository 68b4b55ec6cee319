cd $GRAFT_REPO_ROOT
VSP_ATT_KSPLIT=1 VSP_ATT_KW=2 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "long or text_encoder or frame_prior" 2>&1 | tail -1
for rep in 1 2; do for V in 4 2; do
VSP_ATT_KW=$V python bench.py --workload C5 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 kw=$V', round(d['ms_per_step'],2),'ms', round(d['value']/1e6,1),'M samples/s')"
done; done
