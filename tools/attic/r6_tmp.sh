cd /tmp && export TMPDIR=/tmp
for s in max_gain max_gain_fresh; do
rm -rf /tmp/kt; STOP=$s NUP=8 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o k -- python $GRAFT_REPO_ROOT/tools/r6_tol.py 2>&1 | grep "ms per"
python $GRAFT_REPO_ROOT/tools/kstats2b.py /tmp/kt/k_kernel_trace.csv 10 | cut -c1-130
done
