#!/bin/bash
# development aid (round 6): the FIRST run of the CLI over freshly written files in /dev/shm — copy out of a mapping (default) against pread() (CORNETTO_CLI_MMAP=0);
# every run gets files of its own (a second read of the same pages does not pay the first read's page-cache bookkeeping)
python3 - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from cornetto_amd import synth
dev = torch.device("cuda", 0)
lens = synth.contig_lengths(0)
bases, offs = synth.make_assembly(torch, dev, lens, 0xC0FFEE)
hb = bases.cpu().numpy()
with open("/dev/shm/fr_base.fa", "wb") as f:
    for i, (o, L) in enumerate(zip(offs, lens)):
        f.write(b">ptg%06dl\n" % i)
        f.write(memoryview(hb[int(o):int(o) + int(L)]))
        f.write(b"\n")
del bases, hb
for name, mq in (("t", False), ("q", True)):
    t = synth.make_bedgraph_text(torch, dev, 50_000_000, 11, mq)
    t.cpu().numpy().tofile("/dev/shm/fr_base_%s.bg" % name)
PY
wall() { local t0=$(date +%s.%N); "$@" > /dev/shm/fr_out.txt 2> /dev/shm/fr_err.txt; local t1=$(date +%s.%N); python3 -c "print('%.3f' % ($t1 - $t0))"; }
for k in 0 1 2 3 4 5; do
  m=$((k % 2))
  cp /dev/shm/fr_base.fa /dev/shm/fr_asm.fa                      # (a copy: pages that nobody has read)
  echo "sdust, first read of its file, mmap=$m: $(CORNETTO_CLI_MMAP=$m wall cornetto_amd/cornetto sdust /dev/shm/fr_asm.fa) s; again: $(CORNETTO_CLI_MMAP=$m wall cornetto_amd/cornetto sdust /dev/shm/fr_asm.fa) s; md5 $(md5sum < /dev/shm/fr_out.txt | cut -c1-8)"
  rm -f /dev/shm/fr_asm.fa
  cp /dev/shm/fr_base_t.bg /dev/shm/fr_t.bg; cp /dev/shm/fr_base_q.bg /dev/shm/fr_q.bg
  echo "noboringbits (2 x 1.7 GB), first read, mmap=$m: $(CORNETTO_CLI_MMAP=$m wall cornetto_amd/cornetto noboringbits /dev/shm/fr_t.bg -q /dev/shm/fr_q.bg) s; again: $(CORNETTO_CLI_MMAP=$m wall cornetto_amd/cornetto noboringbits /dev/shm/fr_t.bg -q /dev/shm/fr_q.bg) s; md5 $(md5sum < /dev/shm/fr_out.txt | cut -c1-8)"
  rm -f /dev/shm/fr_t.bg /dev/shm/fr_q.bg
done
rm -f /dev/shm/fr_*
