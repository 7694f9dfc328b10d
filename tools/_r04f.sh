set -u
O=gpurun_out/r04f; mkdir -p $O; export TMPDIR=/tmp
for i in 1 2; do
timeout 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-conv-events 2>/dev/null | grep '^{' > $O/a_$i.json; python -c "import json;d=json.load(open('$O/a_$i.json'));print('new',d['value'],d['ms_per_step'])"
timeout 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-conv-events --kernel-dbg 268435456 2>/dev/null | grep '^{' > $O/b_$i.json; python -c "import json;d=json.load(open('$O/b_$i.json'));print('no192',d['value'],d['ms_per_step'])"
done
