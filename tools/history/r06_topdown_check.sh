timeout 300 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "bilinear" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_hip_model.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
for rep in 1 2; do for f in 0 1; do echo "GDL_FLAT=$f"; done; done > /dev/null
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-kernel-timer --no-input-stage --min-seconds 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('train', d['value'], 'infer', d.get('inference_tiles_per_s'))"
