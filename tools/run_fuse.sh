cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "fused or stage_generator or edge_lengths or infer_matches or streamed" 2>&1 | tail -3
for W in 1 2; do VSP_PAIR_WD=$W python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wd=$W', round(d['ms_per_step'],2),'ms', round(d['value']/1e6,1),'M samples/s', d['roofline']['launches'], round(d['roofline']['avg_launch_ms'],3))"; done
