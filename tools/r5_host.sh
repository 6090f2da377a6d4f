#!/bin/bash
# host side of the batch mode (round 5): what an ADMM iteration asks of the host with the pacer / blocking waits,
# eight ranks' host loops against one GPU, fit() with both Gauss-Newton updates
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r05; mkdir -p $OUT
for wl in tiny_32c3_thick2 cfg3_256c3_thick6z demo_181c3_thick4xyz; do WL=$wl python tools/host_time.py 2>/dev/null | tail -1; done > $OUT/r05_host_time.jsonl
WL=cfg3_256c3_thick6z PACE=0 python tools/host_time.py 2>/dev/null | tail -1 > $OUT/r05_host_time_nopace.jsonl
python tools/host_contention.py > $OUT/r05_host_contention.jsonl 2>$OUT/contention.err
for wl in cfg3_256c3_thick6z demo_181c3_thick4xyz; do WL=$wl python tools/r4_fit.py 2>/dev/null | tail -1; done > $OUT/r05_fit.jsonl
WL=cfg3_256c3_thick6z UNIRES_SET_REPEAT_DEVICE_SYNC=1 python tools/r4_fit.py 2>/dev/null | tail -1 > $OUT/r05_fit_device_sync.jsonl
head -c 3000 $OUT/r05_host_time.jsonl $OUT/r05_host_time_nopace.jsonl $OUT/r05_host_contention.jsonl $OUT/r05_fit.jsonl $OUT/r05_fit_device_sync.jsonl
