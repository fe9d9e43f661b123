"""Fuzzer for the device-side BGZF deflate (trgt_amd/csrc/deflate_dev.hip, trgt_deflate_blocks): blocks of random size and make-up --
BAM-like records, tandem repeats, runs, noise, and mixtures cut at random places -- must come back as raw DEFLATE streams that zlib (the
reference decoder) inflates to exactly the input, or be declined; room sizes are drawn too, and nothing may be written beyond a room.

    python tests/tools/deflate_fuzz.py [n_blocks=20000] [seed=1]      (GPU)"""
import os
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))


def make_block(rng):
    n = int(rng.choice([0, 1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 257, 258, 259, 1019, 1020, 1021, 4096, 65279, 65280, 65535, 65536])) if rng.random() < 0.15 else int(rng.integers(0, 65537))
    parts, have = [], 0
    while have < n:
        kind = int(rng.integers(0, 7))
        m = int(min(n - have, rng.integers(1, 6000)))
        if kind == 0:
            p = rng.integers(0, 256, m, dtype=np.uint8)
        elif kind == 1:
            unit = rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8)
            p = np.tile(unit, m // len(unit) + 1)[:m]
        elif kind == 2:
            p = np.full(m, int(rng.integers(0, 256)), np.uint8)
        elif kind == 3:   # quality-like: short runs of a few values
            v = rng.integers(33, 94, m // 5 + 1, dtype=np.uint8)
            p = np.repeat(v, rng.integers(1, 12, len(v)))[:m]
            if len(p) < m:
                p = np.concatenate([p, np.zeros(m - len(p), np.uint8)])
        elif kind == 4:   # 4-bit packed bases of a repeat with errors
            unit = rng.integers(0, 4, int(rng.integers(2, 30)))
            seq = np.tile(unit, 2 * m // len(unit) + 2)[:2 * m].copy()
            err = rng.random(2 * m) < 0.01
            seq[err] = rng.integers(0, 4, int(err.sum()))
            code = np.array([1, 2, 4, 8], np.uint8)[seq]
            p = (code[0::2] << 4 | code[1::2]).astype(np.uint8)[:m]
        elif kind == 5 and parts:   # a copy of something earlier, at a random distance
            src = np.concatenate(parts)
            a = int(rng.integers(0, len(src)))
            p = src[a:a + m]
            if len(p) < m:
                p = np.concatenate([p, rng.integers(0, 256, m - len(p), dtype=np.uint8)])
        else:
            p = np.frombuffer(("read/%d/ccs\0" % int(rng.integers(0, 10 ** 6))).encode() * (m // 8 + 1), np.uint8)[:m]
        parts.append(np.asarray(p, np.uint8))
        have += m
    return (np.concatenate(parts)[:n] if parts else np.zeros(0, np.uint8)).tobytes()


def main():
    from trgt_amd import _lib, writers
    n_blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    ctx = _lib.Context(0)
    done = declined = bad = 0
    in_bytes = out_bytes = z1_bytes = 0
    while done < n_blocks:
        k = int(min(n_blocks - done, rng.integers(1, 700)))
        datas = [make_block(rng) for _ in range(k)]
        cap = int(rng.choice([0xFF00, 0xFF00, 0xFF00, 0x10000, 4096, 300, 64]))
        got = writers.deflate_blocks(ctx, datas, cap=cap)
        for d, s in zip(datas, got):
            if s is None:
                declined += 1
                if cap >= 0xFF00 and len(d) * 9 // 8 + 8 < cap and len(zlib.compress(d, 1)) < 0.5 * len(d):
                    bad += 1
                    print("declined a compressible block of", len(d), "bytes with room", cap)
                continue
            dec = zlib.decompressobj(-15)
            try:
                ok = len(s) <= cap and dec.decompress(s) + dec.flush() == d and dec.eof and not dec.unused_data
            except zlib.error as e:
                ok = False
                print("zlib:", e)
            if not ok:
                bad += 1
                print("MISMATCH: block of", len(d), "bytes, stream of", len(s), "room", cap)
            elif cap >= 0xFF00:
                in_bytes += len(d); out_bytes += len(s); z1_bytes += len(zlib.compress(d, 1))
        done += k
    print("deflate_fuzz: %d blocks (seed %d), %d declined, %d mismatches; %.1f MB in, device streams %.1f MB, zlib level 1 %.1f MB"
          % (done, seed, declined, bad, in_bytes / 1e6, out_bytes / 1e6, z1_bytes / 1e6))
    ctx.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
