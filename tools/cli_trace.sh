#!/bin/bash
# development aid: the full CORNETTO_CLI_TRACE timeline of `cornetto sdust` / `telofind` on the bench assembly as a FASTA in /dev/shm
# (tools/perf_cli_ahead.py writes it and keeps it when KEEP=1)   bash tools/cli_trace.sh [out-prefix]
out=${1:-gpurun_out/r06_cli_trace}
python3 - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from cornetto_amd import synth
dev = torch.device("cuda", 0)
lens = synth.contig_lengths(0)
bases, offs = synth.make_assembly(torch, dev, lens, 0xC0FFEE)
hb = bases.cpu().numpy()
with open("/dev/shm/asm1.fa", "wb") as f:
    for i, (o, L) in enumerate(zip(offs, lens)):
        f.write(b">ptg%06dl\n" % i)
        f.write(memoryview(hb[int(o):int(o) + int(L)]))
        f.write(b"\n")
PY
for sub in sdust telofind; do
  for rep in 1 2; do
    t0=$(date +%s.%N)
    CORNETTO_CLI_TRACE=1 CORNETTO_TRACE=${LIBTRACE:-0} CORNETTO_SDUST_TRACE=${LIBTRACE:-0} ${CLI:-cornetto_amd/cornetto} $sub /dev/shm/asm1.fa > /dev/shm/out.$sub 2> $out.$sub.$rep.txt
    t1=$(date +%s.%N)
    echo "$sub wall $(echo "$t1 - $t0" | bc -l 2>/dev/null || python3 -c "print($t1 - $t0)") s" | tee -a $out.$sub.$rep.txt
  done
  md5sum /dev/shm/out.$sub
done
rm -f /dev/shm/asm1.fa /dev/shm/out.*
