#!/bin/bash
# development aid: the phases of cornetto_bgin_feed() (CORNETTO_TRACE, development build) while `cornetto_dev noboringbits` reads 2 x 1.7 GB of per-base bedgraph text
python3 - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from cornetto_amd import synth
dev = torch.device("cuda", 0)
for name, mq in (("t", False), ("q", True)):
    t = synth.make_bedgraph_text(torch, dev, 50_000_000, 11, mq)
    t.cpu().numpy().tofile("/dev/shm/bgtrace_%s.bg" % name)
PY
CORNETTO_CLI_TRACE_READS=1 CORNETTO_CLI_TRACE=1 CORNETTO_TRACE=1 cornetto_amd/cornetto_dev noboringbits /dev/shm/bgtrace_t.bg -q /dev/shm/bgtrace_q.bg > /dev/null 2> /dev/shm/bgtrace.err
grep "trace" /dev/shm/bgtrace.err | sed -n "60,110p"
rm -f /dev/shm/bgtrace_*.bg /dev/shm/bgtrace.err
