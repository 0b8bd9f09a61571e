import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np, json
import cornetto_amd
from cornetto_amd import synth
dev = torch.device("cuda", 0)
for prof in ("humanlike", "satellite", "uniform"):
    lens = synth.contig_lengths(1_000_000_000)
    bases, offs = synth.make_assembly(torch, dev, lens, 0xC0FFEE, prof)
    acc = cornetto_amd.Accel(0)
    asm = acc.asm_wrap(bases.data_ptr(), offs, np.array(lens, dtype=np.int64))
    print(prof, json.dumps(acc.sdust_stats(asm, 20, 64)))
    asm.close(); acc.close(); del bases
