#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p13; mkdir -p $O
timeout 300 python scripts/tall_panel.py > $O/tall_panel.txt 2>&1; grep -v amdgpu.ids $O/tall_panel.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
python bench.py --steps 5 --warmup 1 > $O/bench_default.json 2>$O/bench_default.err; tail -c 3000 $O/bench_default.json
