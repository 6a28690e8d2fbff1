cd $GRAFT_REPO_ROOT
O=gpurun_out/r5h
mkdir -p $O
PCC_BENCH_SHARE_DEVICES=1 timeout -k 10 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu > $O/rehearsal_2ranks.json 2> $O/rehearsal_2ranks.err; echo "rc=$?"; tail -5 $O/rehearsal_2ranks.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5h/rehearsal_2ranks.json'))
print({k:d[k] for k in ('n_gpus','scaling','ms_per_step','value')}, d['config']['queries_total'], d['config']['queries_per_gpu'], d['config']['backend'], d['broadcast_bytes'], d['scaling_terms'])
print('extra keys', list(d['extra'].keys()), d['extra']['c3_weak']['queries_total'], d['extra']['c5']['queries_total'], d['extra']['c5']['queries_per_gpu'])
PY
