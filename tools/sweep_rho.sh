# operator parameter sweep on the bench workload (one GPU): prints ms/step and inner iterations
for RV in 0.1 0.3 1 3 25; do for CK in 10 25; do for AD in $CK 100; do
timeout -k 10 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-converge --op-rho-v $RV --op-check $CK --op-adapt $AD 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rho_v', $RV, 'check', $CK, 'adapt', $AD, round(d['ms_per_step'],3), d['breakdown']['operator_inner_iters_mean'])"
done; done; done
