#!/bin/bash
# development aid: where the wall time outside main() goes (process start, exit) for `cornetto sdust` / `telofind` on the bench assembly
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
TIMEFORMAT='%R real %U user %S sys'
for rep in 1 2; do
for sub in sdust telofind; do
  echo "== $sub -> /dev/null"; time (cornetto_amd/cornetto $sub /dev/shm/asm1.fa > /dev/null 2> /dev/shm/err.txt); grep "Real time" /dev/shm/err.txt
  echo "== $sub -> /dev/shm file"; time (cornetto_amd/cornetto $sub /dev/shm/asm1.fa > /dev/shm/out.txt 2> /dev/shm/err.txt); grep "Real time" /dev/shm/err.txt
  echo "== $sub -> pipe to cat > /dev/null"; time (cornetto_amd/cornetto $sub /dev/shm/asm1.fa 2> /dev/shm/err.txt | cat > /dev/null); grep "Real time" /dev/shm/err.txt
done
done
rm -f /dev/shm/asm1.fa /dev/shm/out.txt /dev/shm/err.txt
