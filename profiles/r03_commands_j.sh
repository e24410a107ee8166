# r03: host-side profile of the training step
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03v
rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-roofline --workload train"
python -c "
import cProfile, pstats, sys, io
sys.argv = ['bench.py', '--no-cpu-baseline', '--no-roofline', '--workload', 'train', '--steps', '10', '--warmup', '3']
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path('bench.py', run_name='__main__')
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(70)
open('$O/cprofile_cum.txt', 'w').write(s.getvalue())
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(45)
open('$O/cprofile_tot.txt', 'w').write(s.getvalue())
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_callers('method .to. of|_lib.py.*call|run_backward')
open('$O/cprofile_callers.txt', 'w').write(s.getvalue())
" > $O/bench_c.json 2>> $O/bench.err
