cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" rocprofv3 --kernel-trace -d /tmp/rp_$RANDOM -o x -- python3 $R/bench.py --no-dropin --no-cpu --no-roofline --steps 50 --warmup 5 > /tmp/out.txt 2> /tmp/err.txt; echo "rc=$?"; grep -c SIGSEGV /tmp/err.txt; }
run A=1
run MORB_PYRAMID_PAIRS=0
run MORB_INLINE_MATCH=0
run MORB_CHAIN_GRAPH=0
run MORB_NO_BAR_STAGING=1
