#!/bin/bash
# end-of-round check on the GPU box: the whole GPU suite, smoke(), the default bench line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03final; mkdir -p $O
cd $R
timeout 3000 python -m pytest tests/ -q -m gpu > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "rc=$?" >> $O/smoke.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
ls $O
