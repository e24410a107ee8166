# r05 call C: the whole GPU suite with the new flip-aware / same-crop / config-4 / quarantine tests (prints kept)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05c
rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s --tb=short -p no:cacheprovider > $O/gpu_tests_full.log 2>&1 < /dev/null
grep -E "passed|failed|error" $O/gpu_tests_full.log | tail -5
grep -E "^(FAILED|ERROR)|crop differences|at the product latents|quarantine at|attacker per scene|same crops|closure at the product" $O/gpu_tests_full.log | head -80
