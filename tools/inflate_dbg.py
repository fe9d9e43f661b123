import sys, zlib, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from test_inflate import _corpus, _raw
from trgt_amd import _lib, ingest
ctx = _lib.Context(0)
streams, datas, tags = [], [], []
for ci, data in enumerate(_corpus()):
    data = data[:65536]
    for level in (0, 1, 2, 4, 6, 9):
        for si, strategy in enumerate((zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE)):
            streams.append(_raw(data, level, strategy)); datas.append(data); tags.append((ci, level, si))
got, status = ingest.inflate_blocks(ctx, streams, [len(d) for d in datas])
bad = 0
for g, d, st, tg, s in zip(got, datas, status, tags, streams):
    if st == 1 and g != d:
        bad += 1
        a = np.frombuffer(g, np.uint8); b = np.frombuffer(d, np.uint8)
        diff = np.nonzero(a != b)[0]
        if bad <= 12: print("MISMATCH corpus %d level %d strategy %d: len %d, comp %d, first diff at %d, n diff %d  got %r want %r" % (tg + (len(d), len(s), diff[0], len(diff), g[diff[0]-4:diff[0]+8], d[diff[0]-4:diff[0]+8])))
print("declined:", [tg + (len(d), len(s)) for d, st, tg, s in zip(datas, status, tags, streams) if st == 0])
print("bad", bad, "declined", int((status == 0).sum()), "of", len(streams))
