cd "$GRAFT_REPO_ROOT"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(round(d["value"]), "apply_us", round(r["launches_per_step"]*r["mean_launch_us"],1), "launches", r["launches_per_step"], "MB", round(r["bytes_per_launch"]*r["launches_per_step"]/1e6,1), "frac", round(r["frac"],3))'
B="python bench.py --steps 1500 --warmup 50 --no-cpu-baseline --no-large-spmv --no-replicas --no-other-configs"
for i in 1 2; do
echo "uniform [2,2,2,2,2]: $(FC_ND_SHAPE=2,2,2,2,2 $B 2>/dev/null | python -c "$P")"
echo "default: $($B 2>/dev/null | python -c "$P")"
done
python scripts/batch_probe.py --skip-parity --steps 300 --ks 8,16,32 2>/dev/null | grep "k="
FC_ND_SHAPE=2,2,2,2,2 python scripts/batch_probe.py --skip-parity --steps 300 --ks 8,16,32 2>/dev/null | grep "k="
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
