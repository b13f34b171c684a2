O=gpurun_out/r06r; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_zz_decoder_probe.py -m gpu -q -p no:cacheprovider > $O/pytest_dec.log 2>&1; tail -3 $O/pytest_dec.log
timeout 600 python bench.py --no-traffic --no-cpu-baseline --no-one-process > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06r/bench.json').read().strip().splitlines()[-1])
print("decode", {k:d['decode'].get(k) for k in ('value','batched')} if 'decode' in d else None)
for n in ('C4','C5'):
    print(n, d['batch_configs'][n]['value'], d['batch_configs'][n].get('decode',{}).get('value'))
PY
tail -3 $O/bench.err
