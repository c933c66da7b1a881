cd /root/repo
for rep in 1 2 3; do
for at in mlp_fwd composite_fw composite_bw hashgrid_fwd; do
    b=$(NGP_MARCH_AT=$at timeout 120 python bench.py --steps 20 --warmup 5 --timed-only 2>/dev/null | grep "^{" | python -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "at=$at rep=$rep plain=$b"
done; done
