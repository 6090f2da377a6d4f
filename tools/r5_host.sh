#!/bin/bash
# host side of the batch mode (round 5): what an ADMM iteration asks of the host with the pacer / blocking waits,
# eight ranks' host loops against one GPU, fit() with both Gauss-Newton updates
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/${ROUND:-r06}; R=${ROUND:-r06}; mkdir -p $OUT
for wl in tiny_32c3_thick2 cfg3_256c3_thick6z demo_181c3_thick4xyz; do WL=$wl python tools/host_time.py 2>/dev/null | tail -1; done > $OUT/${R}_host_time.jsonl
WL=cfg3_256c3_thick6z PACE=0 python tools/host_time.py 2>/dev/null | tail -1 > $OUT/${R}_host_time_nopace.jsonl
python tools/host_contention.py > $OUT/${R}_host_contention.jsonl 2>$OUT/contention.err
for wl in cfg3_256c3_thick6z demo_181c3_thick4xyz; do WL=$wl python tools/r4_fit.py 2>/dev/null | tail -1; done > $OUT/${R}_fit.jsonl
WL=cfg3_256c3_thick6z UNIRES_SET_REPEAT_DEVICE_SYNC=1 python tools/r4_fit.py 2>/dev/null | tail -1 > $OUT/${R}_fit_device_sync.jsonl
head -c 3000 $OUT/${R}_host_time.jsonl $OUT/${R}_host_time_nopace.jsonl $OUT/${R}_host_contention.jsonl $OUT/${R}_fit.jsonl $OUT/${R}_fit_device_sync.jsonl
