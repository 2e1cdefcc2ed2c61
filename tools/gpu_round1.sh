set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python __graft_entry__.py smoke 2>&1 | tail -5
python bench.py --steps 100 --warmup 10 > gpurun_out/bench1.json 2> gpurun_out/bench1.err; tail -3 gpurun_out/bench1.err; cat gpurun_out/bench1.json
rocprofv3 --kernel-trace --stats -d gpurun_out/prof1 -o r01 -- python bench.py --steps 30 --warmup 5 --no-cpu > gpurun_out/bench_prof.json 2> gpurun_out/prof1.err; tail -3 gpurun_out/prof1.err
ls -R gpurun_out/prof1 | head -20
python -c "
import torch, sys
sys.path.insert(0,'tests')
import multi_orb_slam_amd as m, numpy as np, oracle
from multi_orb_slam_amd import synth
print('torch', torch.__version__, torch.cuda.is_available())
mt = m.Matcher()
r = synth.descriptors(1000,1); q = synth.perturbed_queries(r)
a = mt.hamming_top2(q,r); b = oracle.bf_top2(q,r)
print('torch-first load parity:', all(np.array_equal(x,y) for x,y in zip(a,b)))
"
