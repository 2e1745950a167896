"""debug aid: bucketed (2) vs streaming (1) model pass outputs, fresh codec each time"""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch, divans_amd as da, workload
corpus = workload.load_corpus()
dev = torch.device('cuda', 0)
for N, L in ((70, 1), (70, 64), (3, 9000), (70, 65536), (3000, 9000)):
    blocks = workload.make_blocks(corpus, 40, N, block_len=L, perturb_per_block=L // 100)
    d_in = torch.from_numpy(blocks).to(dev)
    res = {}
    for path in (1, 2, 2):
        codec = da.LiteralCodec(da.config_simple(), L)
        codec.set_encode_path(path)
        pairs = codec.model_batch(d_in, N, L)
        torch.cuda.synchronize()
        res.setdefault(path, []).append(pairs.cpu().numpy()[:, :2 * L])
        codec.close()
    ref = res[1][0]
    for k, got in enumerate(res[2]):
        diff = got != ref
        print(f"N={N} L={L} run{k}: mismatching nibbles {int(diff.sum())} in {int(diff.any(axis=1).sum())} streams")
        for i in np.nonzero(diff.any(axis=1))[0][:3]:
            idx = np.nonzero(diff[i])[0]
            prevs = [int(blocks[i][j // 2 - 1]) if j >= 2 else 0 for j in idx[:12]]
            print("   stream", i, "n", idx.size, "nibble idx", idx[:12].tolist(), "prev bytes", prevs,
                  "got", [hex(int(x) & 0xffffffff) for x in got[i][idx[:4]]], "ref", [hex(int(x) & 0xffffffff) for x in ref[i][idx[:4]]])
