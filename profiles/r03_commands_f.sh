# r03: the whole per-batch sequence of adv_gen_rule_based.cfg (bench.py --workload full)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03f
timeout 600 python bench.py --workload full --steps 1 --warmup 0 > gpurun_out/r03f/full.json 2> gpurun_out/r03f/full.err
tail -5 gpurun_out/r03f/full.err
