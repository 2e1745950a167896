#!/usr/bin/env python3
"""Compare the (start,freq) pairs of the bucketed two-model pass with the streaming kernel's, position by position."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import divans_amd as da, workload
L = int(sys.argv[1]) if len(sys.argv) > 1 else 63
corpus = workload.load_corpus()
blocks = workload.make_blocks(corpus, 40, 70, block_len=L, perturb_per_block=L // 100)
d = torch.from_numpy(blocks).to("cuda:0")
res = []
for path in (1, 2, 2, 2):
    codec = da.LiteralCodec(da.config_simple() if os.environ.get("DBG_SIMPLE") else da.config_context_mixing(), L)
    codec.set_encode_path(path)
    pairs = codec.model_batch(d, blocks.shape[0], L)
    torch.cuda.synchronize()
    res.append(pairs.cpu().numpy().view(np.uint32)[:, :2 * L].copy())
    codec.close()
for k in (1, 2, 3):
    bad = np.argwhere(res[0] != res[k])
    print("run", k, "mismatches", len(bad), bad[:12].tolist())
    for (s, i) in bad[:3]:
        print("  stream", s, "nibble", i, "want %08x got %08x" % (res[0][s, i], res[k][s, i]), "byte", blocks[s, i // 2], "prev", blocks[s, i // 2 - 1] if i >= 2 else 0)
