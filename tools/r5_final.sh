#!/bin/bash
# round 5: the whole device suite, the smoke test, bench.py's refusal of more GPUs than the box has, then the round's measurement set
O=gpurun_out/r5f; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x -s > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -4 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py --gpus 8 --steps 2 --warmup 1 > $O/gpus8.out 2> $O/gpus8.err; echo "bench --gpus 8 on this box: rc $? : $(tail -1 $O/gpus8.err)" | tee $O/gpus8.txt
bash tools/run_profiles.sh r5
