export TMPDIR=/tmp
for i in 1 2 3; do
for f in 1 0; do
echo "no_fused=$f $(MMD_AMD_NO_FUSED_STEP=$f timeout 300 python bench.py --steps 8 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"]), "traj/s", round(d["ms_per_step"],2), "ms/round")')"
done; done > gpurun_out/r03_fused_step_ab.txt
