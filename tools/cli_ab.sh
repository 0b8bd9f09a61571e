#!/bin/bash
# development aid: `cornetto sdust` / `telofind` on the bench assembly (FASTA in /dev/shm), the whole-file path against the piece loop, alternating on one box
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
wall() { local t0=$(date +%s.%N); "$@" > /dev/shm/out.txt 2> /dev/shm/err.txt; local t1=$(date +%s.%N); python3 -c "print('%.3f' % ($t1 - $t0))"; }
echo "version only (process start + exit, no GPU): $(wall cornetto_amd/cornetto --version) $(wall cornetto_amd/cornetto --version)"
for rep in 1 2 3; do
  for sub in sdust telofind; do
    for whole in 0 1; do
      t=$(CORNETTO_CLI_WHOLE=$whole CORNETTO_CLI_TRACE=1 wall cornetto_amd/cornetto $sub /dev/shm/asm1.fa)
      echo "$sub whole=$whole wall $t s; inside main: $(grep 'Real time' /dev/shm/err.txt | sed 's/.*Real time: //; s/;.*//'); md5 $(md5sum < /dev/shm/out.txt | cut -c1-8)"
      if [ $rep = 3 ]; then grep "cli trace" /dev/shm/err.txt | head -12; fi
    done
  done
done
rm -f /dev/shm/asm1.fa /dev/shm/out.txt /dev/shm/err.txt
