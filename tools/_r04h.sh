for i in 1 2; do
python tools/ab_bench.py base= --rounds 4 --block 20 2>&1 | tail -1
python tools/ab_bench.py base= --rounds 4 --block 20 --hi-main 2>&1 | tail -1
done
